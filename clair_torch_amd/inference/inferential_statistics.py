"""compute_video_mean_and_std with the reference's signature (clair_torch/inference/inferential_statistics.py:19-49).

Every batch of frames is one launch of ct_video_stats_batch (optional ICRF linearization fused with the unweighted
WBOMeanVar update); the running mean and m2 stay on the device.
"""
import math
from typing import Optional

import torch
from torch.utils.data import DataLoader

from .. import ops
from ..common.typecheck import expect
from ..models.base import ICRFModelBase
from ._staging import normalise_transform_list, resolve_device, stage_images


def compute_video_mean_and_std(dataloader: DataLoader, device, icrf_model: Optional[ICRFModelBase] = None,
                               gpu_transforms=None):
    """Mean and standard deviation of the mean over all frames served by ``dataloader``; returns float32
    ``(mean, std)`` squeezed like the reference.  ``gpu_transforms`` (extension): raw integer frames with
    [CastTo('float32'), Normalize(max, 0)] are normalised inside the kernel."""
    expect(dataloader, DataLoader, "dataloader")
    expect(device, (str, torch.device), "device")
    expect(icrf_model, ICRFModelBase, "icrf_model", allow_none=True)
    dev = resolve_device(device)
    transforms = normalise_transform_list(gpu_transforms)
    lut = interp = None
    if icrf_model is not None:
        lut, interp = icrf_model.icrf.detach().to(dev), icrf_model.interp_name
    mean = m2 = None
    n_frames = 0
    for _, val_batch, _std_batch, _meta in dataloader:
        frames, max_code, layout = stage_images(val_batch, dev, transforms, want_layout=True)
        if mean is None:
            f = frames.shape  # interleaved frames as decoded (CvToTorch folded into the kernel): the state is planar
            chw = tuple(f[1:]) if layout == "nchw" else (f[3], f[1], f[2])
            mean = torch.empty(chw, dtype=torch.float32, device=dev)
            m2 = torch.empty_like(mean)
        ops.video_stats_batch(frames, mean, m2, n_frames, lut=lut, interp=interp, max_code=max_code, layout=layout)
        n_frames += frames.shape[0]
    if mean is None:
        raise ValueError("dataloader yielded no batches")
    variance = m2 * (1 / (n_frames - 1)) if n_frames > 1 else m2 * float("inf")  # SAMPLE_FREQUENCY scale
    return mean.squeeze(), torch.sqrt(variance.squeeze()) / math.sqrt(n_frames)
