"""Shared staging of a collated batch onto the device for the inference / training entry points."""
from typing import Iterable, Optional

import torch

from ..common.transforms import fusable_code_normalisation, fusable_layout


def resolve_device(device) -> torch.device:
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError(f"device={device!r}: clair_torch_amd computes on MI355X only (use 'cuda' / 'cuda:k'); "
                           "it has no CPU path -- the reference's CPU behaviour is reproduced by oracle/ for tests.")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def normalise_transform_list(gpu_transforms) -> list:
    if gpu_transforms is None:
        return []
    if isinstance(gpu_transforms, Iterable):
        return [t for t in gpu_transforms if t is not None]
    return [gpu_transforms]


def stage_images(val_batch: torch.Tensor, device: torch.device, transforms: list, want_layout: bool = False):
    """Move the value batch to the device and run / fuse the device transforms.

    Returns (images, max_code): integer codes with their max_code when the transform list is the
    CastTo(float32)+Normalize(max, 0) pair the kernels ingest directly, else float32 pixels and None.
    With ``want_layout`` a third value is returned: "nhwc_bgr" when the list additionally starts with CvToTorch
    on raw (B,H,W,3) frames (the kernel then reads the interleaved BGR frames as they are), else "nchw"."""
    images = val_batch.to(device=device, non_blocking=True)  # the ONE host-to-device copy of the batch (a plain DMA when pinned)
    if want_layout:
        layout, rest = fusable_layout(images, transforms)
        if layout != "nchw":
            max_code = fusable_code_normalisation(images, rest)
            if max_code is not None:
                return images, max_code, layout
        out = stage_images(images, device, transforms)  # already on the device: .to() is then the identity
        return out[0], out[1], "nchw"
    max_code = fusable_code_normalisation(images, transforms)
    if max_code is not None:
        return images, max_code
    for t in transforms:
        images = t(images)
    if images.dtype in (torch.uint8, torch.uint16):
        raise TypeError("integer images reached the kernel without a Normalize transform; pass "
                        "gpu_transforms=[CastTo('float32'), Normalize(max_val=<max code>, min_val=0)]")
    return images.to(torch.float32).contiguous(), None


def std_arguments(std_batch: Optional[torch.Tensor], dataset, device):
    """(explicit std tensor | None, std_mode, std_value): explicit tensors as the reference, or the dataset's
    ``std_hint`` for uncertainties derived in-kernel (see datasets/stack_dataset.py)."""
    if std_batch is not None:
        return std_batch.to(device=device, dtype=torch.float32, non_blocking=True), "explicit", 0.0
    hint = getattr(dataset, "std_hint", None)
    if hint is not None:
        return None, hint[0], float(hint[1])
    return None, "none", 0.0
