"""Container-only stand-in for the albumentations 1.1.0 names MolNexTR/data_aug.py needs so that its CropWhite and
PadToSquare classes (pure numpy inside) can be imported and driven by tools/gen_golden.py. Restated from the
package's documented behaviour: BasicTransform.update_params records rows/cols of the image; pad_with_params with
BORDER_CONSTANT is a constant-colour border (cv2.copyMakeBorder)."""
from . import augmentations  # noqa: F401


class BasicTransform:
    def __init__(self, always_apply=False, p=0.5):
        self.always_apply, self.p = always_apply, p

    def update_params(self, params, **kwargs):
        if "image" in kwargs:
            params.update({"cols": kwargs["image"].shape[1], "rows": kwargs["image"].shape[0]})
        return params


class DualTransform(BasicTransform):
    pass


class ImageOnlyTransform(BasicTransform):
    pass


class SafeRotate(DualTransform):
    def __init__(self, limit=90, interpolation=1, border_mode=4, value=None, mask_value=None, always_apply=False, p=0.5):
        super().__init__(always_apply, p)
