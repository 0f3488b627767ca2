def _absent(*a, **k):
    raise NotImplementedError("not needed by CropWhite / PadToSquare")


safe_rotate_enlarged_img_size = _maybe_process_in_chunks = keypoint_rotate = resize = _absent
