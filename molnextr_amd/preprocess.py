"""Host pre-processing in front of the device path.

  CropWhite(pad=50) -> Resize(384,384, bilinear) -> ToGray -> Normalize(ImageNet) -> CHW float32
  (reference MolNexTR/dataset.py:158-185 with augment=False, MolNexTR/data_aug.py:98-143, MolNexTR/model.py:104)

The reference runs these through albumentations 1.1.0 / OpenCV, neither of which is installed here; they are
restated from their documented behaviour: `cv2.resize(INTER_LINEAR)` on uint8 = half-pixel-centre bilinear with
11-bit fixed-point weights and OpenCV's two-pass integer arithmetic (its 2x-decimation special case, which
switches to area averaging for exact integer scale 2, is NOT restated); `cv2.cvtColor(RGB2GRAY)` = (R*4899 + G*9617 + B*1868 + 8192) >> 14; Normalize =
(x/255 - mean) / std. CropWhite and PadToSquare ARE pinned: tools/gen_golden.py drives the reference's own classes
(MolNexTR/data_aug.py) on ragged pages and tests/test_preprocess.py checks crop boxes, shapes and content hashes
(tests/golden/crop_pad.json). The OpenCV resize / gray arithmetic stays UNPINNED until a box with OpenCV can produce
fixtures.
"""
import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def crop_box(img: np.ndarray, value=(255, 255, 255)):
    """(crop_top, crop_bottom, crop_left, crop_right) as CropWhite.update_params reports them (reference
    data_aug.py:106-136): rows / columns to drop on each side so that the bounding box of every pixel that differs
    from `value` in ANY channel remains; all zero for a blank page. Pinned by tests/golden/crop_pad.json."""
    h, w, _ = img.shape
    ink = (img != np.asarray(value, dtype=img.dtype)).any(axis=2)
    rows, cols = np.flatnonzero(ink.any(axis=1)), np.flatnonzero(ink.any(axis=0))
    if rows.size == 0:
        return 0, 0, 0, 0
    return int(rows[0]), int(h - 1 - rows[-1]), int(cols[0]), int(w - 1 - cols[-1])


def _pad_white(img: np.ndarray, top: int, bottom: int, left: int, right: int, value=(255, 255, 255)) -> np.ndarray:
    h, w, c = img.shape
    out = np.empty((h + top + bottom, w + left + right, c), dtype=img.dtype)
    out[...] = np.asarray(value, dtype=img.dtype)
    out[top:top + h, left:left + w] = img
    return out


def crop_white(img: np.ndarray, pad: int = 50, value=(255, 255, 255)) -> np.ndarray:
    """CropWhite(pad): crop to the ink bounding box, then a white border of `pad` pixels (data_aug.py:138-143)."""
    h, w, _ = img.shape
    t, b, l, r = crop_box(img, value)
    return _pad_white(img[t:h - b, l:w - r], pad, pad, pad, pad, value)


def pad_to_square(img: np.ndarray, value=(255, 255, 255)) -> np.ndarray:
    """PadToSquare (reference data_aug.py:286-301, inserted after CropWhite for 'real/acs.csv' and 'real/UOB.csv',
    dataset.py:163-164): the shorter side grows to the longer one, diff//2 white pixels first, the rest after."""
    h, w, _ = img.shape
    diff = abs(h - w)
    p1, p2 = diff // 2, diff - diff // 2
    return _pad_white(img, p1, p2, 0, 0, value) if h <= w else _pad_white(img, 0, 0, p1, p2, value)


def _linear_coeffs(src: int, dst: int):
    """cv2.resize(INTER_LINEAR) sampling for 8-bit images: fx = (float)((dx+0.5)*scale-0.5) with a double `scale`
    = 1/(dst/src), source index clamped to [0, src-1] with the fraction zeroed at the borders, and BOTH weights
    rounded to 11 bits on their own (saturate_cast<short>(w * 2048), round-half-even)."""
    scale = 1.0 / (float(dst) / float(src))
    fx = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int64)
    frac = (fx - sx.astype(np.float32)).astype(np.float32)
    frac[sx < 0] = 0.0
    sx[sx < 0] = 0
    over = sx >= src - 1
    frac[over] = 0.0
    sx[over] = src - 1
    w0 = np.rint((np.float32(1.0) - frac) * np.float32(2048.0)).astype(np.int64)
    w1 = np.rint(frac * np.float32(2048.0)).astype(np.int64)
    return sx, np.minimum(sx + 1, src - 1), w0, w1


def resize_bilinear_u8(img: np.ndarray, size: int) -> np.ndarray:
    """Two-pass fixed point as OpenCV's 8-bit linear resize: horizontal sums keep 11 fractional bits, the vertical
    pass computes ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16), adds 2 and shifts by 2."""
    h, w, _ = img.shape
    y0, y1, wy0, wy1 = _linear_coeffs(h, size)
    x0, x1, wx0, wx1 = _linear_coeffs(w, size)
    im = img.astype(np.int64)
    r0 = im[y0][:, x0] * wx0[None, :, None] + im[y0][:, x1] * wx1[None, :, None]
    r1 = im[y1][:, x0] * wx0[None, :, None] + im[y1][:, x1] * wx1[None, :, None]
    out = (((wy0[:, None, None] * (r0 >> 4)) >> 16) + ((wy1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def to_gray_rgb(img: np.ndarray) -> np.ndarray:
    g = (img[..., 0].astype(np.int64) * 4899 + img[..., 1].astype(np.int64) * 9617 +
         img[..., 2].astype(np.int64) * 1868 + 8192) >> 14
    return np.repeat(g.astype(np.uint8)[..., None], 3, axis=2)


def transform_image(img: np.ndarray, input_size: int = 384, square: bool = False) -> np.ndarray:
    """HWC uint8 RGB -> CHW float32, normalised: the tensor the device path takes. square: PadToSquare after CropWhite."""
    if img.ndim == 2:
        img = np.repeat(img[..., None], 3, axis=2)
    img = np.ascontiguousarray(img[..., :3], dtype=np.uint8)
    page = crop_white(img, 50)
    if square:
        page = pad_to_square(page)
    x = to_gray_rgb(resize_bilinear_u8(page, input_size)).astype(np.float32)
    x = (x - MEAN * 255.0) * (1.0 / (STD * 255.0))
    return np.ascontiguousarray(x.transpose(2, 0, 1), dtype=np.float32)


def load_image_rgb(path: str) -> np.ndarray:
    """cv2.imread + BGR2RGB of the reference (model.py:177-178), via PIL (OpenCV is not installed here)."""
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)
