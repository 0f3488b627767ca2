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
