"""Host pre-processing in front of the device path ("next" row — restated, not yet pinned).

  CropWhite(pad=50) -> Resize(384,384, bilinear) -> ToGray -> Normalize(ImageNet) -> CHW float32
  (reference MolNexTR/dataset.py:158-185 with augment=False, MolNexTR/data_aug.py:98-143, MolNexTR/model.py:104)

The reference runs these through albumentations 1.1.0 / OpenCV, neither of which is installed here; they are
restated from their documented behaviour: `cv2.resize(INTER_LINEAR)` on uint8 = half-pixel-centre bilinear with
11-bit fixed-point weights and OpenCV's two-pass integer arithmetic (its 2x-decimation special case, which
switches to area averaging for exact integer scale 2, is NOT restated); `cv2.cvtColor(RGB2GRAY)` = (R*4899 + G*9617 + B*1868 + 8192) >> 14; Normalize =
(x/255 - mean) / std. PARITY UNPINNED until a box with OpenCV can produce fixtures.
"""
import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def crop_white(img: np.ndarray, pad: int = 50, value=(255, 255, 255)) -> np.ndarray:
    """Crop to the bounding box of non-white pixels, then pad `pad` white pixels on every side."""
    h, w, _ = img.shape
    ink = (img != np.array(value, dtype=img.dtype)).sum(axis=2)
    if ink.sum() != 0:
        rows, cols = ink.sum(axis=1), ink.sum(axis=0)
        top = 0
        while rows[top] == 0 and top + 1 < h:
            top += 1
        bottom = h
        while rows[bottom - 1] == 0 and bottom - 1 > top:
            bottom -= 1
        left = 0
        while cols[left] == 0 and left + 1 < w:
            left += 1
        right = w
        while cols[right - 1] == 0 and right - 1 > left:
            right -= 1
        img = img[top:bottom, left:right]
    return np.pad(img, ((pad, pad), (pad, pad), (0, 0)), mode="constant", constant_values=value[0])


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


def transform_image(img: np.ndarray, input_size: int = 384) -> np.ndarray:
    """HWC uint8 RGB -> CHW float32, normalised: the tensor the device path takes."""
    if img.ndim == 2:
        img = np.repeat(img[..., None], 3, axis=2)
    img = np.ascontiguousarray(img[..., :3], dtype=np.uint8)
    x = to_gray_rgb(resize_bilinear_u8(crop_white(img, 50), input_size)).astype(np.float32)
    x = (x - MEAN * 255.0) * (1.0 / (STD * 255.0))
    return np.ascontiguousarray(x.transpose(2, 0, 1), dtype=np.float32)


def load_image_rgb(path: str) -> np.ndarray:
    """cv2.imread + BGR2RGB of the reference (model.py:177-178), via PIL (OpenCV is not installed here)."""
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)
