"""Batch collation with the reference's semantics (clair_torch/datasets/collate.py:8-43):
samples are sorted by exposure time, a batch with any missing std image has std_batch = None, and the metadata
dict collates Python floats into float64 tensors.

One extension: when the per-sample images are equally spaced views of one tensor (what an in-memory, device-resident
``StackDataset`` hands out), the batch is returned as a strided VIEW of that tensor instead of a stacked copy -- at
BASELINE's C2 the copy alone (3.2 GB read + 3.2 GB written) takes longer than the merge kernel.  Consumers in this
package never write to the image batches."""
import torch
from torch.utils.data._utils.collate import default_collate


def _stack_or_view(tensors):
    first = tensors[0]
    if not isinstance(first, torch.Tensor):
        return default_collate(tensors)
    if len(tensors) == 1:
        # batch_size 1 (linearization): a view with a leading axis -- no 12-50 MB host copy per frame, and a pinned frame
        # stays pinned, so its host-to-device copy is a plain DMA
        return first.unsqueeze(0)
    base = first.untyped_storage().data_ptr()
    step = tensors[1].storage_offset() - first.storage_offset()
    if step < first.numel() or not first.is_contiguous():
        return default_collate(tensors)
    for k, t in enumerate(tensors):
        if (not isinstance(t, torch.Tensor) or t.dtype != first.dtype or t.shape != first.shape or
                t.stride() != first.stride() or t.device != first.device or t.untyped_storage().data_ptr() != base or
                t.storage_offset() != first.storage_offset() + k * step):
            return default_collate(tensors)
    return torch.as_strided(first, (len(tensors),) + tuple(first.shape), (step,) + tuple(first.stride()),
                            first.storage_offset())


def custom_collate(batch):
    ordered = sorted(batch, key=lambda item: item[3]["exposure_time"])
    indices, vals, stds, metas = zip(*ordered)
    std_batch = None if any(s is None for s in stds) else _stack_or_view(stds)
    return default_collate(indices), _stack_or_view(vals), std_batch, default_collate(metas)
