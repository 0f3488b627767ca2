import numpy as np

from . import functional, geometric, transforms  # noqa: F401


def pad_with_params(img, h_pad_top, h_pad_bottom, w_pad_left, w_pad_right, border_mode=0, value=None):
    """cv2.copyMakeBorder(img, top, bottom, left, right, BORDER_CONSTANT, value) — the only mode the reference uses."""
    assert border_mode == 0, "stand-in implements BORDER_CONSTANT only"
    v = np.asarray(value if value is not None else 0, dtype=img.dtype)
    h, w = img.shape[:2]
    out = np.empty((h + h_pad_top + h_pad_bottom, w + w_pad_left + w_pad_right) + img.shape[2:], dtype=img.dtype)
    out[...] = v
    out[h_pad_top:h_pad_top + h, w_pad_left:w_pad_left + w] = img
    return out
