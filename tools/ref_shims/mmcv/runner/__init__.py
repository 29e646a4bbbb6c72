def load_checkpoint(model, filename, map_location=None, strict=False, logger=None, **kw):
    """No-op: no checkpoint files exist offline; weights are loaded via load_state_dict."""
    return {}
