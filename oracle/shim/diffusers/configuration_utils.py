"""Plumbing-only stand-ins for diffusers.configuration_utils (test shim)."""
import functools
import inspect


class FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class ConfigMixin:
    config_name = "config.json"

    def register_to_config(self, **kw):
        cur = dict(getattr(self, "_cfg", {}))
        cur.update(kw)
        self._cfg = cur

    @property
    def config(self):
        return FrozenDict(getattr(self, "_cfg", {}))


def register_to_config(init):
    @functools.wraps(init)
    def wrapped(self, *args, **kwargs):
        sig = inspect.signature(init)
        params = list(sig.parameters.values())[1:]
        cfg = {p.name: p.default for p in params if p.default is not inspect._empty}
        for p, a in zip(params, args):
            cfg[p.name] = a
        cfg.update(kwargs)
        ConfigMixin.register_to_config(self, **cfg)
        init(self, *args, **kwargs)

    return wrapped
