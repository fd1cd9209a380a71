"""The two device-side transforms that define the uint -> [0,1] mapping (clair_torch/common/transforms.py:108-183).

``compute_hdr_image`` / ``linearize_dataset_generator`` recognise the pair ``[CastTo(float32), Normalize(max, 0)]`` in
``gpu_transforms`` applied to integer codes and fold it into the kernels' load stage; any other transform list is
executed with these classes' ``__call__`` (plain PyTorch ops on the device) before the float32 kernel variant runs.
"""
from typing import Optional

import torch

from .enums import DTYPE_MAP


class BaseTransform:
    def __call__(self, x: torch.Tensor) -> torch.Tensor:  # pragma: no cover - interface
        raise NotImplementedError


class CvToTorch(BaseTransform):
    """OpenCV (H, W[, C]) BGR -> PyTorch (C, H, W) RGB (reference transforms.py / general_functions.py:315-335).
    Applied to a collated batch (B, H, W, C) it acts per image.  As the first entry of ``gpu_transforms`` on raw
    integer frames it is folded into the kernels' load stage (layout "nhwc_bgr") instead of being executed."""

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if x.ndim == 2:
            return x.unsqueeze(0)
        if x.ndim in (3, 4) and x.shape[-1] == 3:
            # torch has no uint16 indexing / flip kernels: reverse the channels on a same-width signed view
            alias = {torch.uint16: torch.int16, torch.uint32: torch.int32}.get(x.dtype)
            y = (x.view(alias) if alias else x).flip(-1)
            y = y.view(x.dtype) if alias else y
            return y.permute(2, 0, 1) if x.ndim == 3 else y.permute(0, 3, 1, 2)
        raise ValueError(f"Unexpected image shape: {tuple(x.shape)}")


class CastTo(BaseTransform):
    def __init__(self, data_type=None, device=None):
        if isinstance(data_type, str):
            data_type = DTYPE_MAP[data_type]
        self.data_type = data_type
        self.device = torch.device(device) if isinstance(device, str) else device

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return x.to(dtype=self.data_type if self.data_type is not None else x.dtype,
                    device=self.device if self.device is not None else x.device)


class Normalize(BaseTransform):
    """(x - min) / (max - min) * span + min_t (reference general_functions.py:359-388).  Executed as a torch op on a
    GPU tensor the division by a scalar is a multiplication by its reciprocal (torch's GPU kernels), 1 ulp away from
    the CPU result for some codes; when the kernels fold this transform (integer codes, see
    ``fusable_code_normalisation``) they reproduce the CPU reference's correctly rounded division instead."""

    def __init__(self, max_val: Optional[float] = None, min_val: Optional[float] = None, target_range=(0.0, 1.0)):
        self.max_val, self.min_val, self.target_range = max_val, min_val, tuple(target_range)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        # clair_torch/common/general_functions.py:359-388
        max_val = x.max() if self.max_val is None else self.max_val
        min_val = x.min() if self.min_val is None else self.min_val
        den = max_val - min_val
        if den == 0:
            raise ValueError("Normalization range is zero (min == max); cannot normalize.")
        lo, hi = self.target_range
        return (x - min_val) / den * (hi - lo) + lo


def fusable_layout(images: torch.Tensor, transforms):
    """("nhwc_bgr", remaining transforms) when the list starts with CvToTorch on a (B,H,W,3) integer batch -- the
    channel reversal and the HWC->CHW transpose are then done by the kernel's addressing -- else ("nchw", transforms)."""
    ts = [t for t in transforms if t is not None]
    if ts and isinstance(ts[0], CvToTorch) and images.ndim == 4 and images.shape[3] == 3 and \
            images.dtype in (torch.uint8, torch.uint16):
        return "nhwc_bgr", ts[1:]
    return "nchw", ts


def fusable_code_normalisation(images: torch.Tensor, transforms):
    """If ``transforms`` applied to integer codes is exactly CastTo(float32)? + Normalize(max, 0, (0,1)),
    return max_code so the kernels can ingest the raw codes; otherwise None."""
    if images.dtype not in (torch.uint8, torch.uint16):
        return None
    ts = [t for t in transforms if t is not None]
    if ts and isinstance(ts[0], CastTo) and ts[0].data_type in (None, torch.float32) and ts[0].device is None:
        if ts[0].data_type is None:
            return None
        ts = ts[1:]
    else:
        return None
    if len(ts) == 1 and isinstance(ts[0], Normalize):
        n = ts[0]
        if n.max_val is not None and (n.min_val in (0, 0.0)) and tuple(n.target_range) == (0.0, 1.0):
            mc = float(n.max_val)
            if 1.0 <= mc <= 65535.0 and mc == int(mc):
                return mc
    return None
