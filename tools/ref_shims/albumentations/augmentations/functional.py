def grid_distortion(*a, **k):
    raise NotImplementedError
