"""Stand-in for timm 0.4.12's model factory (construction only)."""
_REGISTRY = {}


def create_model(name, pretrained=False, **kwargs):
    import MolNexTR.models.transformers  # noqa: F401  (registers swin_base)
    kwargs.pop('pretrained_strict', None)
    return _REGISTRY[name](pretrained=False, **kwargs)
