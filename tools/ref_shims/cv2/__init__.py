"""Container-only stand-in for the `cv2` names MolNexTR/data_aug.py touches at IMPORT time (default arguments) and
the one call CropWhite / PadToSquare make through albumentations (constant-border padding). Not OpenCV: no resize,
no colour conversion — those stay unpinned."""
INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA = 0, 1, 2, 3
BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_WRAP, BORDER_REFLECT_101 = 0, 1, 2, 3, 4
FONT_HERSHEY_SIMPLEX = 0
COLOR_BGR2GRAY, COLOR_BGR2RGB = 6, 4


def _absent(*a, **k):
    raise NotImplementedError("cv2 is not installed; only constants exist in this stand-in")


getRotationMatrix2D = warpAffine = line = putText = rectangle = cvtColor = resize = imread = _absent
