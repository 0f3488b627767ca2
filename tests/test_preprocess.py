"""CPU: host pre-processing restated from albumentations/OpenCV semantics (unpinned: neither library is installed;
these tests pin our own documented behaviour and the invariants the reference transform has)."""
import numpy as np

from molnextr_amd.preprocess import crop_white, resize_bilinear_u8, to_gray_rgb, transform_image, MEAN, STD


def test_crop_white_bounding_box_and_pad():
    img = np.full((100, 120, 3), 255, np.uint8)
    img[30:40, 50:70] = 0
    out = crop_white(img, pad=50)
    assert out.shape == (10 + 100, 20 + 100, 3)
    assert (out[:50] == 255).all() and (out[50:60, 50:70] == 0).all()
    blank = np.full((20, 30, 3), 255, np.uint8)
    assert crop_white(blank, pad=5).shape == (30, 40, 3)          # all-white image: no crop, pad only


def test_resize_identity_and_constant():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(384, 384, 3), dtype=np.uint8)
    assert np.array_equal(resize_bilinear_u8(img, 384), img)
    const = np.full((123, 77, 3), 200, np.uint8)
    assert (resize_bilinear_u8(const, 384) == 200).all()
    up = resize_bilinear_u8(np.array([[[0, 0, 0], [255, 255, 255]]], np.uint8).repeat(2, 0), 4)
    assert up[0, 0, 0] == 0 and up[0, -1, 0] == 255 and (np.diff(up[0, :, 0].astype(int)) >= 0).all()


def test_gray_and_normalise():
    img = np.zeros((2, 2, 3), np.uint8)
    img[..., 0], img[..., 1], img[..., 2] = 255, 0, 0
    g = to_gray_rgb(img)
    assert (g == (255 * 4899 + 8192) >> 14).all() and g.shape == (2, 2, 3)       # 0.299 * 255 = 76
    white = np.full((60, 60, 3), 255, np.uint8)
    white[20:40, 20:40] = 0
    x = transform_image(white)
    assert x.shape == (3, 384, 384) and x.dtype == np.float32
    np.testing.assert_allclose(x[:, 0, 0], (1.0 - MEAN) / STD, rtol=1e-5)       # white corner
    np.testing.assert_allclose(x.min(axis=(1, 2)), (0.0 - MEAN) / STD, rtol=1e-5, atol=1e-6)


def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def test_crop_white_and_pad_to_square_vs_reference_golden(golden_dir):
    """CropWhite.update_params/apply and PadToSquare.apply of the reference's own data_aug.py (driven by
    tools/gen_golden.py through a minimal albumentations/cv2 stand-in) on ragged pages: crop parameters, shapes and
    content hashes. Pages are regenerated from their recipe (W.synthetic_page) and checked by hash first."""
    import json
    import os
    from molnextr_amd import weights as W
    from molnextr_amd.preprocess import crop_box, pad_to_square
    with open(os.path.join(golden_dir, "crop_pad.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) == len(W.PAGE_CASES)
    for c in cases:
        page = W.synthetic_page(c["case"])
        assert list(page.shape) == c["page_shape"] and _sha(page) == c["page_sha"], "page recipe drifted"
        assert list(crop_box(page)) == c["crop"], (c["case"], crop_box(page), c["crop"])
        out = crop_white(page, 50)
        assert list(out.shape) == c["cropped_shape"] and _sha(out) == c["cropped_sha"], c["case"]
        sq = pad_to_square(out)
        assert list(sq.shape) == c["square_shape"] and _sha(sq) == c["square_sha"], c["case"]
        assert sq.shape[0] == sq.shape[1]


def test_pad_to_square_orientation():
    from molnextr_amd.preprocess import pad_to_square
    wide = np.zeros((3, 8, 3), np.uint8)
    sq = pad_to_square(wide)
    assert sq.shape == (8, 8, 3) and (sq[:2] == 255).all() and (sq[2:5] == 0).all() and (sq[5:] == 255).all()   # 5 // 2 = 2 above
    tall = np.zeros((9, 2, 3), np.uint8)
    sq = pad_to_square(tall)
    assert sq.shape == (9, 9, 3) and (sq[:, :3] == 255).all() and (sq[:, 3:5] == 0).all() and (sq[:, 5:] == 255).all()
    x = transform_image(np.zeros((30, 90, 3), np.uint8), square=True)
    assert x.shape == (3, 384, 384)
