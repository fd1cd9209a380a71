"""Dark-field correction of a batch before merging / linearizing (PARITY UNPINNED, see
csrc/ct_darkfield.hip).

The reference applies conditional_gaussian_blur with the batch's matched dark fields right after the device transforms
(clair_torch/inference/hdr_merge.py:76-92, linearization.py:73-92) and later adds the dark field's variance term by a
second autograd.grad (hdr_merge.py:117-126, linearization.py:108-116).  Here one kernel (ct_dark_field_blur) produces the
blurred batch and a per-sample uncertainty that carries both variance terms; the merge / linearize kernels then run
unchanged on (float32 pixels, explicit uncertainty).

Row bands: the 3x3 blur needs the rows just above and below a band.  ``exchange_halo`` gets them from the neighbouring
ranks (one send / receive pair per side over torch.distributed; 2 rows of B x C x W elements, latency-bound on xGMI).
"""
from typing import Optional

import torch
import torch.distributed as dist

from .. import ops


def exchange_halo(images: torch.Tensor, tile: Optional[ops.TileGeometry], group=None) -> Optional[torch.Tensor]:
    """(B, C, 2, W) rows [above, below] of this rank's band, from the ranks holding them; None for an untiled image.
    Bands are assumed stacked in rank order (rank r holds the rows right below rank r - 1), as bench.py / C5 lay them out."""
    if tile is None:
        return None
    b, c, h, w = images.shape
    halo = torch.zeros((b, c, 2, w), dtype=images.dtype, device=images.device)
    has_up, has_down = tile.row_offset > 0, tile.row_offset + h < tile.h_global
    if not (has_up or has_down):
        return halo
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("a row band that does not span the image needs its neighbours' rows: initialise "
                           "torch.distributed (one rank per band) or pass the whole image")
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    peers = dist.get_process_group_ranks(group) if group is not None else list(range(world))
    first_row, last_row = images[:, :, 0].contiguous(), images[:, :, h - 1].contiguous()
    up_buf = torch.empty_like(first_row) if has_up else None
    down_buf = torch.empty_like(last_row) if has_down else None
    ops_list = []
    if has_up:
        ops_list += [dist.P2POp(dist.isend, first_row, peers[rank - 1], group), dist.P2POp(dist.irecv, up_buf, peers[rank - 1], group)]
    if has_down:
        ops_list += [dist.P2POp(dist.isend, last_row, peers[rank + 1], group), dist.P2POp(dist.irecv, down_buf, peers[rank + 1], group)]
    for req in dist.batch_isend_irecv(ops_list):
        req.wait()
    if has_up:
        halo[:, :, 0] = up_buf
    if has_down:
        halo[:, :, 1] = down_buf
    return halo


class DarkField:
    """The dark-field dataset of one compute_hdr_image / linearize_dataset_generator call."""

    def __init__(self, dataset, main_dataset, device):
        self.dataset, self.main_dataset, self.device = dataset, main_dataset, device

    @classmethod
    def from_dataset(cls, dataset, main_dataset, device):
        return cls(dataset, main_dataset, device)

    def apply(self, index_batch, images, max_code, std, std_mode, std_value, tile=None, group=None):
        """-> (blurred float32 batch, explicit per-sample uncertainty | None), or (None, None) when no dark field matches
        this batch (MissingValMode.SKIP_BATCH: the reference then leaves the batch uncorrected, hdr_merge.py:84).

        Error behaviour follows the reference: the dark std is dereferenced unconditionally (AttributeError without it),
        and with a dark std but no image uncertainties autograd finds no path to the dark field (RuntimeError)."""
        frames = [self.main_dataset.files[int(i)] for i in index_batch]
        _, dark, dark_std, _ = self.dataset.get_matching_artefact_images(frames)
        if dark is None:
            return None, None
        if dark_std is None:
            raise AttributeError("'NoneType' object has no attribute 'to'")  # hdr_merge.py:87 / linearization.py:86
        if std is None and std_mode == "none":
            # hdr_merge.py:97-99,118: the ICRF runs under set_grad_enabled(stds is not None), so nothing connects the
            # running average to dark_field_val
            raise RuntimeError("element 0 of tensors does not require grad and does not have a grad_fn")
        halo = exchange_halo(images, tile, group)
        return ops.dark_field_blur(images, dark, dark_std, std=std, std_mode=std_mode, std_value=std_value,
                                   max_code=max_code, tile=tile, halo=halo)

    def linearize(self, index_batch, images, max_code, std, std_mode, std_value, lut, interp):
        xb, sig = self.apply(index_batch, images, max_code, std, std_mode, std_value)
        if xb is None:
            return ops.linearize_frames(images, lut, interp, std=std, std_mode=std_mode, std_value=std_value,
                                        max_code=max_code, want_std=True)
        return ops.linearize_frames(xb, lut, interp, std=sig, want_std=True)
