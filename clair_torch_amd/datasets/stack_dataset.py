"""In-memory exposure stack that yields the reference's per-item 4-tuple
``(index, value image (C,H,W), std image | None, {'exposure_time': float})`` (clair_torch/datasets/base.py:59-139).

The reference's file-backed ImageMapDataset (cv2 decode, filename metadata) is out of scope; any Dataset that
yields this tuple -- the reference's own included -- can be fed to the entry points of this package.

Extension over the reference: when the uncertainty is *derived* from the value image (MissingStdMode CONSTANT or
MULTIPLIER, datasets/base.py:128-135) the dataset can keep ``materialize_std=False``: it then yields ``None`` for
the std image and exposes ``std_hint = (mode, value)``; compute_hdr_image & co. derive sigma inside the kernel
instead of streaming a second float32 stack from HBM.  With ``materialize_std=True`` the behaviour is the
reference's (explicit std tensors).
"""
import math
from typing import Optional, Sequence

import torch
from torch.utils.data import Dataset

from ..common.enums import MissingStdMode


class StackDataset(Dataset):
    def __init__(self, values: torch.Tensor, exposure_times: Sequence[float], stds: Optional[torch.Tensor] = None,
                 missing_std_mode: MissingStdMode = MissingStdMode.NONE, missing_std_value: float = 0.0,
                 materialize_std: bool = True):
        if values.ndim != 4:
            raise ValueError("values must be (N, C, H, W)")
        if len(exposure_times) != values.shape[0]:
            raise ValueError("one exposure time per image is required")
        self.values, self.stds = values, stds
        self.exposure_times = [float(t) for t in exposure_times]
        self.files = list(range(values.shape[0]))  # the reference indexes dataset.files (hdr_merge.py:82)
        self.missing_std_mode, self.missing_std_value = missing_std_mode, float(missing_std_value)
        self.materialize_std = materialize_std
        self.shared_std_tensor = torch.tensor(self.missing_std_value)
        self.std_hint = None
        if stds is None and not materialize_std and missing_std_mode != MissingStdMode.NONE:
            self.std_hint = ("constant" if missing_std_mode == MissingStdMode.CONSTANT else "multiplier",
                             self.missing_std_value)

    def __len__(self):
        return self.values.shape[0]

    def __getitem__(self, i):
        val = self.values[i]
        if self.stds is not None:
            std = self.stds[i]
        elif self.std_hint is not None or self.missing_std_mode == MissingStdMode.NONE:
            std = None
        elif not val.is_floating_point():
            raise TypeError("materialised CONSTANT/MULTIPLIER std needs float value images (normalise first)")
        elif self.missing_std_mode == MissingStdMode.CONSTANT:
            std = self.shared_std_tensor.expand_as(val)
        else:
            std = val * self.shared_std_tensor
        return i, val, std, {"exposure_time": self.exposure_times[i]}


class ArtefactStack:
    """In-memory stand-in for the reference's FlatFieldArtefactMapDataset (clair_torch/datasets/base.py:175-259):
    ``get_matching_artefact_images`` returns the collated 4-tuple ``(indices, value (B,C,H,W), std (B,C,H,W) | None,
    meta)`` with B = the number of frame settings asked for, every entry being the single calibration image it holds."""

    def __init__(self, value: torch.Tensor, std: Optional[torch.Tensor] = None):
        if value.ndim != 3:
            raise ValueError("value must be (C, H, W)")
        self.value, self.std = value, std

    def get_matching_artefact_images(self, reference_frame_settings_list):
        # the reference looks one artefact image up per requested frame and collates them (datasets/base.py:225-255):
        # a batch of B frames gets (B, C, H, W), here B views of the one image held
        n = max(1, len(reference_frame_settings_list))
        value = self.value.unsqueeze(0).expand(n, *self.value.shape)
        std = None if self.std is None else self.std.unsqueeze(0).expand(n, *self.std.shape)
        return torch.zeros(n, dtype=torch.int64), value, std, {}


def synthetic_exposure_stack(n: int, channels: int, height: int, width: int, bits: int = 16, stops_per_step: float = 0.25,
                             t0: float = 1e-3, seed: int = 1234, device="cpu", row_range=None):
    """Synthetic gamma-2.2 scene of SURVEY 8(d): irradiance E ~ U(0, 2/t_mid) per pixel-channel, exposures
    t_n = t0 * 2^(n * stops_per_step), code = round(clip(E t_n, 0, 1)^(1/2.2) * maxcode).
    Returns (codes (N,C,H,W) uint8/uint16, exposure list).  ``row_range=(r0, r1)`` generates only those rows of the
    global image, from a counter-based hash of the GLOBAL pixel coordinates, so tiles generated on different ranks
    assemble into exactly the image a single rank would generate."""
    maxcode = (1 << bits) - 1
    exposures = [t0 * 2.0 ** (k * stops_per_step) for k in range(n)]
    t_mid = math.sqrt(exposures[0] * exposures[-1])
    r0, r1 = (0, height) if row_range is None else row_range
    dev = torch.device(device)
    c_idx = torch.arange(channels, device=dev, dtype=torch.int64).view(channels, 1, 1)
    h_idx = torch.arange(r0, r1, device=dev, dtype=torch.int64).view(1, r1 - r0, 1)
    w_idx = torch.arange(width, device=dev, dtype=torch.int64).view(1, 1, width)
    key = ((c_idx * height + h_idx) * width + w_idx + seed * 1000003) & 0xFFFFFFFF
    # 32-bit integer hash (counter-based, no state) in int64 arithmetic
    key = ((key ^ (key >> 16)) * 0x45D9F3B) & 0xFFFFFFFF
    key = ((key ^ (key >> 16)) * 0x45D9F3B) & 0xFFFFFFFF
    key = key ^ (key >> 16)
    e = key.to(torch.float64) * (2.0 / t_mid / 4294967296.0)
    out_dtype = torch.uint8 if bits <= 8 else torch.uint16
    codes = torch.empty((n, channels, r1 - r0, width), dtype=out_dtype, device=dev)
    for k, t in enumerate(exposures):
        lin = (e * t).clamp_(0.0, 1.0)
        codes[k] = torch.round(lin.pow_(1.0 / 2.2) * maxcode).to(torch.int32).to(out_dtype)
    return codes, exposures
