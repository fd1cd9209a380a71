"""Enums of the reference's public API that the hot path consumes (clair_torch/common/enums.py)."""
from enum import Enum, auto

import torch


class InterpMode(Enum):
    """How ICRFModelBase.forward samples the LUT (reference enums.py:10-16)."""
    LOOKUP = auto()   # nearest sample, no gradient wrt the image
    LINEAR = auto()
    CATMULL = auto()


class MissingStdMode(Enum):
    """How a dataset derives a missing uncertainty image (reference enums.py:34-40)."""
    NONE = auto()
    CONSTANT = auto()
    MULTIPLIER = auto()


class ChannelOrder(Enum):
    RGB = auto()
    BGR = auto()
    ANY = auto()


class DimensionOrder(Enum):
    BCS = auto()  # batch, channel, spatial (PyTorch)
    BSC = auto()  # batch, spatial, channel (OpenCV)


class VarianceMode(Enum):
    POPULATION = auto()
    SAMPLE_FREQUENCY = auto()
    RELIABILITY_WEIGHTS = auto()


DTYPE_MAP = {"float16": torch.float16, "float32": torch.float32, "float64": torch.float64, "bfloat16": torch.bfloat16}
REVERSE_DTYPE_MAP = {v: k for k, v in DTYPE_MAP.items()}

# kernel-side names of the interpolation modes (include/clair_hip.h CT_INTERP_*)
INTERP_NAME = {InterpMode.LOOKUP: "lookup", InterpMode.LINEAR: "linear", InterpMode.CATMULL: "catmull"}
