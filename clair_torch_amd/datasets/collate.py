"""Batch collation with the reference's semantics (clair_torch/datasets/collate.py:8-43):
samples are sorted by exposure time, a batch with any missing std image has std_batch = None, and the metadata
dict collates Python floats into float64 tensors."""
from torch.utils.data._utils.collate import default_collate


def custom_collate(batch):
    ordered = sorted(batch, key=lambda item: item[3]["exposure_time"])
    indices, vals, stds, metas = zip(*ordered)
    std_batch = None if any(s is None for s in stds) else default_collate(stds)
    return default_collate(indices), default_collate(vals), std_batch, default_collate(metas)
