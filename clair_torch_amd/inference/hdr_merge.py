"""compute_hdr_image with the reference's signature (clair_torch/inference/hdr_merge.py:19-155).

Per batch the reference linearizes, divides by the exposure time, weights, updates a WBOMean, and back-propagates
through all of it to get the variance; here each batch is ONE fused kernel launch (ct_hdr_merge_batch) that updates
device-resident state, and the closed form of that variance is evaluated in registers.  Streaming semantics are the
reference's, including the batch-partition dependence of the variance (state detached per batch, hdr_merge.py:128).
"""
from typing import Callable, Optional

import torch
from torch.utils.data import DataLoader

from .. import ops
from ..common.typecheck import expect
from ..models.base import ICRFModelBase
from ._staging import normalise_transform_list, resolve_device, stage_images, std_arguments


def compute_hdr_image(dataloader: DataLoader, device, icrf_model: Optional[ICRFModelBase] = None,
                      weight_fn: Optional[Callable] = None, flat_field_dataset=None, gpu_transforms=None,
                      dark_field_dataset=None, tile: Optional[ops.TileGeometry] = None):
    """Merge the exposure stack served by ``dataloader`` into an HDR image and its standard uncertainty.

    Returns ``(mean float64 (C,H,W), std float32 (C,H,W) | None)`` on ``device`` (squeezed like the reference).
    ``weight_fn``: None = unit weights, anything else = Gaussian weights, scale 30 (hdr_merge.py:95).
    ``tile`` (extension): the rows this process holds of a taller global image (multi-GPU row bands).
    """
    expect(dataloader, DataLoader, "dataloader")
    expect(device, (str, torch.device), "device")
    expect(icrf_model, ICRFModelBase, "icrf_model", allow_none=True)
    if weight_fn is not None and not callable(weight_fn):
        expect(weight_fn, type(None), "weight_fn")
    if flat_field_dataset is not None or dark_field_dataset is not None:
        raise NotImplementedError("flat-field / dark-field corrections are not built yet (SURVEY 8f rows 1 and 4)")
    dev = resolve_device(device)
    transforms = normalise_transform_list(gpu_transforms)
    lut = interp = None
    if icrf_model is not None:
        lut, interp = icrf_model.icrf.detach().to(dev), icrf_model.interp_name

    state = result = None
    batches = iter(dataloader)
    pending = next(batches, None)
    if pending is None:
        raise ValueError("dataloader yielded no batches")
    while pending is not None:
        _, val_batch, std_batch, meta_batch = pending
        pending = next(batches, None)
        last = pending is None
        images, max_code = stage_images(val_batch, dev, transforms)
        std, std_mode, std_value = std_arguments(std_batch, dataloader.dataset, dev)
        if state is None and not last:
            state = ops.MergeState(tuple(images.shape[1:]), dev, with_variance=std_mode != "none")
        result = ops.hdr_merge_batch(images, meta_batch["exposure_time"], lut=lut, interp=interp,
                                     gaussian_weight=weight_fn is not None, std=std, std_mode=std_mode,
                                     std_value=std_value, max_code=max_code, state=state, finalize=last, tile=tile)
    mean, std = result
    return mean.squeeze(), (std.squeeze() if std is not None else None)
