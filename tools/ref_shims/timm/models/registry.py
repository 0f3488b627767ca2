def register_model(fn):
    import timm
    timm._REGISTRY[fn.__name__] = fn
    return fn
