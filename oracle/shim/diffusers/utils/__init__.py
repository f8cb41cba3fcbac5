import logging as _logging
from collections import OrderedDict
from dataclasses import fields, is_dataclass


class BaseOutput(OrderedDict):
    def __post_init__(self):
        for f in fields(self):
            v = getattr(self, f.name)
            if v is not None:
                self[f.name] = v

    def __getitem__(self, k):
        if isinstance(k, str):
            return dict(self.items())[k]
        return list(self.values())[k]


class logging:  # noqa: N801
    @staticmethod
    def get_logger(name):
        return _logging.getLogger(name)


def _get_model_file(*a, **k):
    raise RuntimeError("shim: no hub access")


SAFETENSORS_WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"
WEIGHTS_NAME = "diffusion_pytorch_model.bin"
