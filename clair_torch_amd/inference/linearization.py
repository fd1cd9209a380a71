"""linearize_dataset_generator with the reference's signature (clair_torch/inference/linearization.py:17-132).

The reference handles one frame per iteration: host-to-device copy (:62-63), forward + autograd.grad + square + sqrt
(:95-106), optional flat field (:118-130), and a blocking device-to-host copy of both results (:132).  The arithmetic
per frame is 8 us of HBM time; moving 12 MB in and 50 MB out over PCIe is ~1 ms, so end to end this path is a copy
pipeline (BASELINE config C4).  Here the per-frame yield order and the yielded values are the reference's -- value and
LINEAR / LOOKUP std bit for bit -- but the frames move through a three-stage pipeline:

    DataLoader (one frame per item, as the reference requires)
      -> host-to-device copies of a GROUP of frames on a copy stream (plain DMA when the dataset hands out pinned memory)
      -> ONE multi-frame ct_linearize_std launch (+ ct_flatfield_*) on the compute stream, into re-used ring slots
      -> device-to-host copies of (lin, std) into pinned host tensors on a second copy stream
      -> each frame is yielded once its group's copy event has completed; the next groups are already in flight.

Ownership: the yielded tensors are CPU tensors (pinned, from PyTorch's caching host allocator) that belong to the caller,
like the reference's ``.cpu()`` results; the pipeline never writes to them again.
"""
from collections import deque
from typing import Optional

import torch
from torch.utils.data import DataLoader

from .. import ops
from ..common.typecheck import expect
from ..models.base import ICRFModelBase
from ._staging import normalise_transform_list, resolve_device, stage_images, std_arguments
from ..common.transforms import fusable_code_normalisation, fusable_layout

_GROUP_BYTES = 256 << 20  # device-to-host bytes per group (two float32 planes per frame): ~5 frames of 1080p RGB
_SLOTS = 3                # ring slots = groups in flight (copy-in | compute | copy-out)


def linearize_dataset_generator(dataloader: DataLoader, device, icrf_model: ICRFModelBase, flatfield_dataset=None,
                                gpu_transforms=None, dark_field_dataset=None):
    expect(dataloader, DataLoader, "dataloader")
    expect(device, (str, torch.device), "device")
    expect(icrf_model, ICRFModelBase, "icrf_model")
    if not dataloader.batch_size == 1:
        raise ValueError("For linearization only batch_size of 1 is allowed.")
    dev = resolve_device(device)
    transforms = normalise_transform_list(gpu_transforms)
    lut, interp = icrf_model.icrf.detach().to(dev), icrf_model.interp_name
    flat = flat_std = None
    if flatfield_dataset is not None:  # linearization.py:48-57
        _, flat, flat_std, _ = flatfield_dataset.get_matching_artefact_images([dataloader.dataset.files[0]])
        flat = flat.to(dev)
        flat_std = flat_std.to(dev) if flat_std is not None else None
    dark = None
    if dark_field_dataset is not None:  # linearization.py:73-92: looked up per frame in the reference; one image here
        from .dark_field import DarkField
        dark = DarkField.from_dataset(dark_field_dataset, dataloader.dataset, dev)

    items = iter(dataloader)
    first = next(items, None)
    if first is None:
        return
    # Which route?  The pipeline ingests what the kernel ingests: raw integer codes whose normalisation (and OpenCV
    # layout) fold into the load, or float32 pixels with no device transform at all.  Anything else (arbitrary
    # gpu_transforms, a dark field: per-frame conditional blur) goes frame by frame through the generic staging.
    probe = first[1]
    layout, rest = fusable_layout(probe, transforms)
    max_code = fusable_code_normalisation(probe, rest)
    if max_code is None:
        layout = "nchw"
    streamable = dark is None and probe.ndim == 4 and ((max_code is not None) or
                                                       (not transforms and probe.dtype == torch.float32))
    if first[2] is not None and layout != "nchw":  # explicit uncertainty images are planar: generic route
        streamable = False
    if not streamable:
        yield from _frame_by_frame(first, items, dataloader, dev, transforms, lut, interp, flat, flat_std, dark)
        return
    yield from _pipelined(first, items, dataloader, dev, lut, interp, flat, flat_std, max_code, layout)


def _frame_by_frame(first, items, dataloader, dev, transforms, lut, interp, flat, flat_std, dark):
    item = first
    while item is not None:
        index_batch, val_batch, std_batch, meta_batch = item
        images, max_code, layout = stage_images(val_batch, dev, transforms, want_layout=True)
        std, std_mode, std_value = std_arguments(std_batch, dataloader.dataset, dev)
        if (std is not None or dark is not None) and layout != "nchw":  # explicit std / dark images are planar
            images, max_code, layout = stage_images(images, dev, transforms) + ("nchw",)
        if dark is not None:
            lin, lin_std = dark.linearize(index_batch, images, max_code, std, std_mode, std_value, lut, interp)
        else:
            lin, lin_std = ops.linearize_frames(images, lut, interp, std=std, std_mode=std_mode, std_value=std_value,
                                                max_code=max_code, want_std=True, layout=layout)
        if flat is not None:  # linearization.py:118-130: mean is a constant, the image term is not rescaled
            ops.flatfield_correct(lin, lin_std, flat, flat_std, input_is_variance=False, through_mean=False)
        yield lin.squeeze().cpu(), lin_std.squeeze().cpu(), meta_batch
        item = next(items, None)


class _Slot:
    """One ring slot: device buffers for a group of frames and the events that order its three stages."""

    def __init__(self, group, frame_shape, dtype, with_std, chw, dev):
        self.frames = torch.empty((group,) + tuple(frame_shape), dtype=dtype, device=dev)
        self.std = torch.empty((group,) + tuple(frame_shape), dtype=torch.float32, device=dev) if with_std else None
        self.lin_out = torch.empty((group,) + tuple(chw), dtype=torch.float32, device=dev)
        self.std_out = torch.empty_like(self.lin_out)
        self.copied_in, self.computed, self.copied_out = (torch.cuda.Event() for _ in range(3))
        self.busy = False


def _pipelined(first, items, dataloader, dev, lut, interp, flat, flat_std, max_code, layout):
    # Memory note for consumers: the yielded CPU tensors of one group (up to _GROUP_BYTES = 256 MB of output, 5 frames of
    # 1080p RGB) are views of two pinned host tensors, so keeping ONE frame alive keeps its group's pinned pages, and
    # list(generator) page-locks the whole output.  The reference hands out independent pageable .cpu() copies; copying
    # here would cost a 50 MB host memcpy per frame (a fifth of the pipeline's throughput), so a consumer that collects
    # frames should .clone() what it keeps.
    probe = first[1]
    frame_shape = tuple(probe.shape[1:])
    chw = frame_shape if layout == "nchw" else (frame_shape[2], frame_shape[0], frame_shape[1])
    out_bytes = 2 * 4 * chw[0] * chw[1] * chw[2]
    group = max(1, min(16, _GROUP_BYTES // out_bytes))
    std_probe, std_mode, std_value = std_arguments(first[2], dataloader.dataset, torch.device("cpu"))
    with_std = std_probe is not None
    slots = [_Slot(group, frame_shape, probe.dtype, with_std, chw, dev) for _ in range(_SLOTS)]
    compute = torch.cuda.current_stream(dev)
    h2d, d2h = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    in_flight = deque()  # (slot, n frames, pinned lin, pinned std, metas)

    def drain_one():
        slot, k, lin_host, std_host, metas = in_flight.popleft()
        slot.copied_out.synchronize()
        slot.busy = False
        for i in range(k):
            yield lin_host[i].squeeze(), std_host[i].squeeze(), metas[i]

    item, g = first, 0
    while item is not None:
        slot = slots[g % _SLOTS]
        if slot.busy:  # its previous group is the oldest in flight: hand those frames over first
            yield from drain_one()
        g += 1
        k, metas = 0, []
        with torch.cuda.stream(h2d):
            # the slot's input buffers were last read by the kernel of its previous group
            h2d.wait_event(slot.computed)
            while item is not None and k < group:
                _, val_batch, std_batch, meta_batch = item
                if tuple(val_batch.shape[1:]) != frame_shape or val_batch.dtype != probe.dtype:
                    raise ValueError("all frames of one linearization run must share shape and dtype")
                slot.frames[k].copy_(val_batch[0], non_blocking=True)
                if with_std != (std_batch is not None):
                    # the pipeline's buffers and kernel variant follow the FIRST frame; a stream that mixes frames with
                    # and without an uncertainty image is refused rather than silently ignoring some of them
                    raise ValueError("uncertainty images present for some frames only")
                if with_std:
                    slot.std[k].copy_(std_batch[0], non_blocking=True)
                metas.append(meta_batch)
                k += 1
                item = next(items, None)
            slot.copied_in.record(h2d)
        compute.wait_event(slot.copied_in)
        compute.wait_event(slot.copied_out)  # the slot's output buffers were last read by its previous copy-out
        lin, lin_std = ops.linearize_frames(slot.frames[:k], lut, interp, std=slot.std[:k] if with_std else None,
                                            std_mode=std_mode, std_value=std_value, max_code=max_code, want_std=True,
                                            layout=layout, out=(slot.lin_out[:k], slot.std_out[:k]))
        if flat is not None:  # linearization.py:118-130
            ops.flatfield_correct(lin, lin_std, flat, flat_std, input_is_variance=False, through_mean=False)
        slot.computed.record(compute)
        with torch.cuda.stream(d2h):
            d2h.wait_event(slot.computed)
            lin_host = torch.empty(lin.shape, dtype=torch.float32, pin_memory=True)
            std_host = torch.empty(lin.shape, dtype=torch.float32, pin_memory=True)
            lin_host.copy_(lin, non_blocking=True)
            std_host.copy_(lin_std, non_blocking=True)
            slot.copied_out.record(d2h)
        slot.busy = True
        in_flight.append((slot, k, lin_host, std_host, metas))
    while in_flight:
        yield from drain_one()
