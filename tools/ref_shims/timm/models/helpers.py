def overlay_external_default_cfg(default_cfg, kwargs):
    kwargs.pop('external_default_cfg', None)


def build_model_with_cfg(model_cls, variant, pretrained, default_cfg=None,
                         pretrained_filter_fn=None, **kwargs):
    model = model_cls(**kwargs)
    model.default_cfg = default_cfg
    return model
