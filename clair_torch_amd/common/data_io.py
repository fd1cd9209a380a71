"""ICRF curve text I/O with the reference's on-disk conventions (clair_torch/common/data_io.py:23-100): by default
the file holds the curve as (L, C) columns in BGR order; in memory the model wants (C, L) rows in RGB order.
Image / video file I/O (OpenCV) is out of scope of this package."""
from pathlib import Path

import numpy as np
import torch

from .enums import ChannelOrder, DimensionOrder
from .typecheck import expect


def _validate_input_txt(path: Path):
    # clair_torch/validation/io_checks.py: existence, file-ness and suffix, with the reference's error types
    if not path.exists():
        raise FileNotFoundError(f"File {path} doesn't exist.")
    if not path.is_file():
        raise ValueError(f"Expected a filepath, got {path}")
    if path.suffix != ".txt":
        raise ValueError(f"Expected .txt filetype, got {path.suffix}")


def load_icrf_txt(path, source_channel_order: ChannelOrder = ChannelOrder.BGR,
                  source_dimension_order: DimensionOrder = DimensionOrder.BSC) -> torch.Tensor:
    """Load an ICRF as a float32 (C, L) RGB tensor (reference data_io.py:23-62)."""
    expect(path, (str, Path), "path")
    expect(source_channel_order, ChannelOrder, "source_channel_order")
    expect(source_dimension_order, DimensionOrder, "source_dimension_order")
    path = Path(path)
    _validate_input_txt(path)
    try:
        data = torch.from_numpy(np.loadtxt(path)).float()
    except Exception as e:
        raise IOError(f"Failed to load NumPy array from {path}: {e}")
    if source_dimension_order == DimensionOrder.BSC:
        data = torch.transpose(data, 0, 1)
    if source_channel_order == ChannelOrder.BGR:
        data = data[[2, 1, 0], :]
    return data


def save_icrf_txt(icrf: torch.Tensor, path, target_channel_order: ChannelOrder = ChannelOrder.BGR,
                  target_dimension_order: DimensionOrder = DimensionOrder.BSC) -> None:
    """Save a (C, L) RGB ICRF tensor (reference data_io.py:65-100)."""
    expect(icrf, torch.Tensor, "icrf")
    expect(path, (str, Path), "path")
    expect(target_channel_order, ChannelOrder, "target_channel_order")
    expect(target_dimension_order, DimensionOrder, "target_dimension_order")
    path = Path(path)
    data = icrf.detach().cpu()
    if target_channel_order == ChannelOrder.BGR:
        data = data[[2, 1, 0], :]
    if target_dimension_order == DimensionOrder.BSC:
        data = torch.transpose(data, 0, 1)
    try:
        np.savetxt(path, data.numpy())
    except Exception:
        raise IOError(f"Couldn't save data to path {path}")
