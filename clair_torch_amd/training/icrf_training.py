"""train_icrf with the reference's signature (clair_torch/training/icrf_training.py:18-186).

The loop, the per-channel optimisers, early stopping and schedulers are host logic and follow the reference step by
step; the per-pixel part of every step (pairs x pixels residual, weights, masks, spatial means, and its backward
into the LUT) is two HIP kernel launches (training/linearity.py).  The O(C*L) curve penalties stay torch ops.

Kept reference behaviours: the first step is a dead step (the model forwards through the ``_icrf`` buffer, which
only becomes a function of the parameters at the first ``update_icrf()``); ``batch_size == 1`` raises;
single-image batches are skipped; live plotting is a headless no-op.
"""
from typing import Optional

import torch
from torch.optim import Optimizer
from torch.utils.data import DataLoader

from .. import ops
from ..common.general_functions import get_valid_exposure_pairs
from ..common.typecheck import expect
from ..inference._staging import normalise_transform_list, resolve_device, stage_images, std_arguments
from ..models.base import ICRFModelBase
from .linearity import linearity_loss
from .losses import (compute_endpoint_penalty, compute_monotonicity_penalty, compute_range_penalty,
                     compute_smoothness_penalty)


def train_icrf(dataloader: DataLoader, batch_size: int, device, icrf_model: ICRFModelBase,
               optimizers: Optional[list] = None, schedulers: Optional[list] = None,
               use_relative_linearity_loss: bool = True, use_uncertainty_weighting: bool = True, epochs: int = 150,
               patience: int = 300, alpha: float = 1.0, beta: float = 1.0, gamma: float = 1.0, delta: float = 1.0,
               lower_valid_threshold: float = 1 / 255, upper_valid_threshold: float = 254 / 255,
               exposure_ratio_threshold: float = 0.1, gpu_transforms=None, tile=None, group=None, verbose: bool = True):
    """Returns the trained model.  Extensions over the reference: ``gpu_transforms`` (raw integer codes in the
    DataLoader), ``tile`` / ``group`` (row-band sharding over ranks with exact all-reduced statistics), ``verbose``."""
    expect(dataloader, DataLoader, "dataloader")
    expect(batch_size, int, "batch_size")
    expect(device, (str, torch.device), "device")
    expect(icrf_model, ICRFModelBase, "icrf_model")
    expect(optimizers, list, "optimizers", allow_none=True)
    expect(schedulers, list, "schedulers", allow_none=True)
    channels = icrf_model.channels
    if batch_size == 1:
        raise ValueError("Batch size must be larger than 1.")
    dev = resolve_device(device)
    if optimizers is None:
        # the reference's defaults (icrf_training.py:64-66); on the GPU the fused implementation does each update in one
        # kernel instead of seven (same update rule; 21 fewer launches per step with three channels)
        fused = all(p.is_cuda for c in range(channels) for p in icrf_model.channel_params(c))
        optimizers = [torch.optim.Adam(icrf_model.channel_params(c), lr=1e-3, amsgrad=False, **({"fused": True} if fused else {}))
                      for c in range(channels)]
    for opt in optimizers:
        expect(opt, Optimizer, "optimizers[...]")
    previous_lrs = [pg["lr"] for opt in optimizers for pg in opt.param_groups]
    if schedulers is None:
        schedulers = [None] * len(optimizers)
    if len(schedulers) != len(optimizers):
        raise ValueError(f"Mismatched number of optimizers: {len(optimizers)} and schedulers: {len(schedulers)}.")
    transforms = normalise_transform_list(gpu_transforms)
    best_losses = [float("inf")] * channels
    epochs_without_improvement = [0] * channels
    icrf_model.train()
    icrf_model.plot_icrf()
    pair_cache = {}

    # The reference reads the epoch's average loss on the host at the end of every epoch (icrf_training.py:161) and does
    # its bookkeeping there: early stopping, schedulers, the LR messages, the plot.  A blocking read leaves the GPU idle
    # from the end of the epoch's last kernel until the host has come back round to the next forward launch.  Here the
    # read is a non-blocking copy plus an event, and the bookkeeping of epoch e is SETTLED inside epoch e + 1, after that
    # epoch's forward and backward have been queued and before its first optimizer step -- the only point from which a
    # scheduler's new learning rate or an early stop can matter.  Same updates, same messages, same returned model: when
    # early stopping triggers, the extra forward / backward has touched nothing but the .grad buffers.
    pending = None
    host_loss = torch.empty(channels, dtype=torch.float64).pin_memory()

    def settle(entry) -> bool:
        """Bookkeeping of a finished epoch (icrf_training.py:161-186); True when early stopping triggers."""
        epoch, event = entry
        event.synchronize()
        avg_loss = host_loss.numpy().copy()
        if verbose:
            print(f"Epoch {epoch + 1} Loss: {avg_loss}")
        avg = avg_loss.reshape(-1)
        for c in range(channels):
            value = avg[c] if avg.size > 1 else avg[0]
            if value < best_losses[c]:
                best_losses[c] = value
                epochs_without_improvement[c] = 0
            else:
                epochs_without_improvement[c] += 1
        if all(epochs_without_improvement[c] >= patience for c in range(channels)):
            if verbose:
                print(f"Early stopping triggered for all channels (patience = {patience} epochs).")
            return True
        for c, scheduler in enumerate(schedulers):
            if scheduler is not None:
                scheduler.step(avg[c] if avg.size > 1 else avg[0])
        for i, optimizer in enumerate(optimizers):
            current_lr = optimizer.param_groups[0]["lr"]
            if current_lr != previous_lrs[i] and verbose:
                print(f"Optimizer {i} learning rate changed to: {current_lr}")
            previous_lrs[i] = current_lr
        if (epoch + 1) % 5 == 0:
            icrf_model.plot_icrf()
        return False

    stopped = False
    for epoch in range(epochs):
        running_loss = torch.zeros(channels, device=dev, dtype=torch.float64)
        for _, val_batch, std_batch, meta_batch in dataloader:
            images, max_code, layout = stage_images(val_batch, dev, transforms, want_layout=True)
            std, std_mode, std_value = std_arguments(std_batch, dataloader.dataset, dev)
            if std is not None and layout != "nchw":  # explicit uncertainty images are planar
                images, max_code, layout = stage_images(images, dev, transforms) + ("nchw",)
            if images.shape[0] < 2:
                print("Skipped batch due to single image.")
                continue
            exposures = meta_batch["exposure_time"].to(torch.float64)
            key = (tuple(exposures.tolist()), exposure_ratio_threshold)
            if key not in pair_cache:
                i_idx, j_idx, ratio = get_valid_exposure_pairs(exposures, exposure_ratio_threshold)
                pair_cache[key] = ops.PairList(i_idx, j_idx, ratio, images.shape[0], dev)
            pairs = pair_cache[key]

            for optimizer in optimizers:
                optimizer.zero_grad()
            icrf_curve = icrf_model.icrf
            lin_loss, _ = linearity_loss(icrf_curve, images, pairs, interp=icrf_model.interp_name,
                                         lower=lower_valid_threshold, upper=upper_valid_threshold,
                                         use_relative=use_relative_linearity_loss,
                                         use_unc_weight=use_uncertainty_weighting, std=std, std_mode=std_mode,
                                         std_value=std_value, max_code=max_code, tile=tile, group=group,
                                         layout=layout)
            loss = (lin_loss + alpha * compute_monotonicity_penalty(icrf_curve, per_channel=True)
                    + beta * compute_range_penalty(icrf_curve, per_channel=True)
                    + gamma * compute_endpoint_penalty(icrf_curve, per_channel=True)
                    + delta * compute_smoothness_penalty(icrf_curve, per_channel=True))
            if len(optimizers) == 1:
                loss = torch.sum(loss)
            if loss.requires_grad:  # False only in the reference's dead first step (curve still a buffer)
                if loss.ndim == 0:
                    loss.backward()
                else:
                    # The reference calls loss[c].backward(retain_graph=True) per channel (icrf_training.py:148-149); the
                    # gradients ACCUMULATE in the same parameter .grad tensors (every loss[c] reaches every LUT row through
                    # the p % C row rule), so what the optimisers see is d(sum_c loss[c]) / d parameters -- one backward
                    # of the sum: one ct_pair_residual_bwd launch instead of three whole-stack launches with two thirds
                    # of their coefficients zero.
                    loss[:len(optimizers)].sum().backward()
            if pending is not None:  # the previous epoch's bookkeeping, while this epoch's kernels run
                stopped, pending = settle(pending), None
                if stopped:
                    break
            for optimizer in optimizers:
                optimizer.step()
            icrf_model.update_icrf()
            running_loss += loss.detach()
        if stopped:
            break
        if pending is not None:  # an epoch whose batches were all skipped never reached the point above
            stopped, pending = settle(pending), None
            if stopped:
                break
        host_loss.copy_(running_loss / len(dataloader), non_blocking=True)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(dev))
        pending = (epoch, done)
    if pending is not None and not stopped:
        settle(pending)
    return icrf_model
