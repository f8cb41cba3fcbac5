def save_image(*a, **k):
    raise RuntimeError("shim")
