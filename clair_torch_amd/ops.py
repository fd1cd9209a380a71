"""Tensor-level front-end of the HIP kernels (thin: validation, pointer plumbing, stream selection).

PyTorch is used here only for device memory, streams and dtype bookkeeping; all arithmetic on image
data happens inside the C-ABI library (include/clair_hip.h).  Every function requires CUDA (ROCm) tensors
and raises otherwise -- there is no CPU implementation in this package.
"""
import ctypes
from dataclasses import dataclass
from typing import Optional

import torch

from . import _native as nv

_INTERP = {"lookup": nv.INTERP_LOOKUP, "linear": nv.INTERP_LINEAR, "catmull": nv.INTERP_CATMULL, None: nv.INTERP_NONE}
_STD = {"none": nv.STD_NONE, "constant": nv.STD_CONSTANT, "multiplier": nv.STD_MULTIPLIER, "explicit": nv.STD_EXPLICIT}
_DTYPE = {torch.uint8: nv.DTYPE_U8, torch.uint16: nv.DTYPE_U16, torch.float32: nv.DTYPE_F32}
# input stack layouts: planar (N,C,H,W) as the reference's tensors, or interleaved (N,H,W,C) as OpenCV decodes
# (optionally BGR: the kernels then fold cv_to_torch's channel reversal into the load).  Outputs are always (C,H,W).
_LAYOUT = {"nchw": nv.LAYOUT_NCHW, "nhwc": nv.LAYOUT_NHWC, "nhwc_bgr": nv.LAYOUT_NHWC_BGR}


@dataclass
class TileGeometry:
    """Rows [row_offset, row_offset + h_tile) of a global (C, h_global, W) image (include/clair_hip.h ct_geometry)."""
    h_global: int
    row_offset: int = 0


def _require_device(t: torch.Tensor, name: str):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: clair_torch_amd kernels run on MI355X (cuda/ROCm) tensors only; "
                           "there is no CPU path in this package")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _chw(stack: torch.Tensor, layout: str):
    if layout not in _LAYOUT:
        raise ValueError(f"unknown layout {layout!r} (nchw, nhwc, nhwc_bgr)")
    if layout == "nchw":
        return stack.shape[1], stack.shape[2], stack.shape[3]
    return stack.shape[3], stack.shape[1], stack.shape[2]


def _out_shape(stack: torch.Tensor, layout: str, out_layout: str):
    """Shape of the merge's state / outputs: planar (C,H,W), or the stack's own per-image shape with out_layout "input"."""
    if out_layout not in ("planar", "input"):
        raise ValueError(f"unknown out_layout {out_layout!r} (planar, input)")
    return tuple(stack.shape[1:]) if out_layout == "input" else tuple(_chw(stack, layout))


def _geometry(stack: torch.Tensor, tile: Optional[TileGeometry], layout: str = "nchw") -> nv.Geometry:
    c, h, w = _chw(stack, layout)
    hg, r0 = (h, 0) if tile is None else (tile.h_global, tile.row_offset)
    if r0 < 0 or r0 + h > hg:
        raise ValueError(f"tile rows [{r0}, {r0 + h}) do not fit a global height of {hg}")
    return nv.Geometry(channels=c, h_tile=h, width=w, h_global=hg, row_offset=r0, image_stride=stack.stride(0),
                       layout=_LAYOUT[layout])


def _check_stack(stack: torch.Tensor, name="stack"):
    _require_device(stack, name)
    if stack.ndim != 4:
        raise ValueError(f"{name} must be (N, C, H, W), got shape {tuple(stack.shape)}")
    if stack.dtype not in _DTYPE:
        raise TypeError(f"{name} dtype {stack.dtype} unsupported (uint8, uint16 codes or float32 pixels)")
    if stack.shape[0] > 0 and not stack[0].is_contiguous():
        raise ValueError(f"every image of {name} must be contiguous (C, H, W)")


def _icrf_struct(lut: Optional[torch.Tensor], interp, channels):
    if lut is None:
        return nv.Icrf(lut_dev=None, n_points=0, interp=nv.INTERP_NONE), None
    _require_device(lut, "lut")
    if lut.ndim != 2 or lut.shape[0] != channels:
        raise ValueError(f"lut must be (C={channels}, L), got {tuple(lut.shape)}")
    if interp not in _INTERP or interp is None:
        raise ValueError(f"Unknown interpolation mode {interp}")
    lut_c = lut.detach().to(torch.float32).contiguous()
    return nv.Icrf(lut_dev=lut_c.data_ptr(), n_points=lut_c.shape[1], interp=_INTERP[interp]), lut_c


class MergeState:
    """Device-resident WBOMean state + running variance of a streaming merge (one entry per output element)."""

    def __init__(self, shape, device, with_variance: bool):
        self.mean = torch.empty(shape, dtype=torch.float64, device=device)
        self.sumw = torch.empty(shape, dtype=torch.float32, device=device)
        self.var = torch.empty(shape, dtype=torch.float32, device=device) if with_variance else None
        self.batches = 0


def hdr_merge_batch(stack: torch.Tensor, exposures: torch.Tensor, *, lut: Optional[torch.Tensor] = None,
                    interp: Optional[str] = "linear", gaussian_weight: bool = True,
                    std: Optional[torch.Tensor] = None, std_mode: str = "none", std_value: float = 0.0,
                    max_code: Optional[float] = None, state: Optional[MergeState] = None, finalize: bool = True,
                    tile: Optional[TileGeometry] = None, mean_dtype: torch.dtype = torch.float64, layout: str = "nchw",
                    force_f64_moments: bool = False, reference_order: Optional[bool] = None, out_layout: str = "planar"):
    """One batch of the HDR merge (ct_hdr_merge_batch).  Returns (mean, std|None) when ``finalize`` else None.

    stack (B,C,H,W) uint8/uint16 codes (give ``max_code``) or float32 pixels; exposures (B) any float dtype.
    ``state`` carries the streaming state across batches (None = single-batch merge).
    ``layout`` "nhwc" / "nhwc_bgr": the stack is (B,H,W,C) as OpenCV decodes it (an explicit std stack likewise);
    outputs stay planar (C,H,W) -- unless ``out_layout="input"`` (extension, CT_MERGE_OUT_AS_INPUT): state and outputs
    then have the stack's own memory order, (H,W,C) in the input's channel order, which is what an OpenCV writer wants
    and lets the kernel store dense packets without regrouping (the MergeState must then be created with that shape).
    ``force_f64_moments`` (diagnostic, CT_MERGE_F64_MOMENTS): keep the float64-moment kernel where the pivoted
    float32 one would run (tests compare the two).
    ``reference_order``: True = evaluate the uncertainty in the reference's own float32 autograd order
    (CT_MERGE_REFERENCE_ORDER: two passes, slower, reproduces the reference's rounding); False = closed-form kernels in
    every mode (CT_MERGE_CLOSED_FORM); None = the library's default (reference order for LOOKUP / CATMULL with
    uncertainties, closed form otherwise).
    """
    _check_stack(stack)
    b = stack.shape[0]
    c, h, w = _chw(stack, layout)
    dev = stack.device
    if b < 1:
        raise ValueError("empty batch")
    if std is not None:
        std_mode = "explicit"
        _require_device(std, "std")
        if std.shape != stack.shape or std.dtype != torch.float32 or std.stride() != stack.stride():
            if std.shape != stack.shape:
                raise ValueError(f"std shape {tuple(std.shape)} != stack shape {tuple(stack.shape)}")
            std = std.to(torch.float32).contiguous()
            if std.stride() != stack.stride():
                stack = stack.contiguous()
    if std_mode not in _STD:
        raise ValueError(f"unknown std_mode {std_mode}")
    if stack.dtype != torch.float32 and max_code is None:
        max_code = 255.0 if stack.dtype == torch.uint8 else 65535.0
    if exposures.is_cuda or dev.type != "cuda":
        exposure_dev = exposures.to(device=dev, dtype=torch.float64).contiguous()
    else:
        # A copy from pageable host memory blocks the host until the stream reaches it -- i.e. until the previous merge
        # kernel has finished -- which serialises this call's host work with the device (measured: 1.18 ms per
        # compute_hdr_image call against a 0.95 ms kernel).  Staged through pinned memory the copy is asynchronous.
        exposure_dev = exposures.to(torch.float64).contiguous().pin_memory().to(dev, non_blocking=True)
    if exposure_dev.numel() != b:
        raise ValueError(f"{exposure_dev.numel()} exposure times for a batch of {b}")
    icrf, lut_keep = _icrf_struct(lut, interp, c)
    geom = _geometry(stack, tile, layout)
    has_std = std_mode != "none"
    first = state is None or state.batches == 0
    flags = (nv.MERGE_FIRST_BATCH if first else 0) | (nv.MERGE_FINALIZE if finalize else 0)
    if force_f64_moments:
        flags |= nv.MERGE_F64_MOMENTS
    if reference_order is not None:
        flags |= nv.MERGE_REFERENCE_ORDER if reference_order else nv.MERGE_CLOSED_FORM
    if mean_dtype == torch.float32:
        flags |= nv.MERGE_MEAN_OUT_F32
    elif mean_dtype != torch.float64:
        raise TypeError("mean_dtype must be float64 (reference) or float32")
    out_shape = _out_shape(stack, layout, out_layout)
    if out_layout == "input":
        flags |= nv.MERGE_OUT_AS_INPUT
    if state is None and not finalize:
        raise ValueError("a non-final batch needs a MergeState")
    if state is not None and has_std and state.var is None:
        raise ValueError("MergeState was created without a variance buffer")
    if state is not None and tuple(state.mean.shape) != out_shape:
        raise ValueError(f"MergeState has shape {tuple(state.mean.shape)}, this merge needs {out_shape}")
    mean_out = torch.empty(out_shape, dtype=mean_dtype, device=dev) if finalize else None
    std_out = torch.empty(out_shape, dtype=torch.float32, device=dev) if (finalize and has_std) else None
    with torch.cuda.device(dev):
        rc = nv.load().ct_hdr_merge_batch(
            _ptr(stack), _DTYPE[stack.dtype], float(max_code or 1.0), b, ctypes.byref(geom), _ptr(std), _STD[std_mode],
            float(std_value), _ptr(exposure_dev), ctypes.byref(icrf), nv.WEIGHT_GAUSS if gaussian_weight else nv.WEIGHT_NONE,
            _ptr(state.mean) if state else None, _ptr(state.sumw) if state else None,
            _ptr(state.var) if (state and state.var is not None) else None, _ptr(mean_out), _ptr(std_out), flags,
            _stream(dev))
    nv.check(rc, "ct_hdr_merge_batch")
    del lut_keep
    if state is not None:
        state.batches += 1
    return (mean_out, std_out) if finalize else None


MAX_MERGE_BATCHES = 16  # batches one ct_hdr_merge_batches call takes (the kernel's argument block holds 16 pointers)


def hdr_merge_batches(stacks, exposures, *, lut: Optional[torch.Tensor] = None, interp: Optional[str] = "linear",
                      gaussian_weight: bool = True, stds=None, std_mode: str = "none", std_value: float = 0.0,
                      max_code: Optional[float] = None, state: Optional[MergeState] = None, finalize: bool = True,
                      tile: Optional[TileGeometry] = None, mean_dtype: torch.dtype = torch.float64, layout: str = "nchw",
                      reference_order: Optional[bool] = None, require_one_launch: bool = False, out_layout: str = "planar"):
    """Several CONSECUTIVE batches of one merge in one call (ct_hdr_merge_batches): the same result, bit for bit, as
    hdr_merge_batch on each of them in turn with ``state`` carried along -- but where the pivoted code-domain kernel
    applies the streaming state stays in registers between the batches (one launch, no state traffic).

    ``stacks``: list of (B_k,C,H,W) device tensors of one dtype / geometry (each sorted by exposure like custom_collate);
    ``exposures``: list of (B_k) tensors; ``stds``: list of explicit std tensors or None.  At most MAX_MERGE_BATCHES.
    ``require_one_launch`` (tests): raise instead of falling back to one launch per batch.  ``out_layout``: as in hdr_merge_batch.
    Returns (mean, std|None) when ``finalize`` else None."""
    k = len(stacks)
    if k == 0 or k != len(exposures) or (stds is not None and len(stds) != k):
        raise ValueError("stacks / exposures / stds must be non-empty lists of equal length")
    if k > MAX_MERGE_BATCHES:
        raise ValueError(f"at most {MAX_MERGE_BATCHES} batches per call")
    if k == 1:
        return hdr_merge_batch(stacks[0], exposures[0], lut=lut, interp=interp, gaussian_weight=gaussian_weight,
                               std=None if stds is None else stds[0], std_mode=std_mode, std_value=std_value, max_code=max_code,
                               state=state, finalize=finalize, tile=tile, mean_dtype=mean_dtype, layout=layout,
                               reference_order=reference_order, out_layout=out_layout)
    for t in stacks:
        _check_stack(t)
        if t.dtype != stacks[0].dtype or t.shape[1:] != stacks[0].shape[1:] or t.device != stacks[0].device:
            raise ValueError("all batches of one call must share dtype, image shape and device")
    dev = stacks[0].device
    c, h, w = _chw(stacks[0], layout)
    if stds is not None:
        std_mode = "explicit"
        stds = [sd.to(device=dev, dtype=torch.float32).contiguous() for sd in stds]
        for sd, t in zip(stds, stacks):
            if sd.shape != t.shape:
                raise ValueError(f"std shape {tuple(sd.shape)} != stack shape {tuple(t.shape)}")
    if std_mode not in _STD:
        raise ValueError(f"unknown std_mode {std_mode}")
    stacks = [t.contiguous() for t in stacks]
    if stacks[0].dtype != torch.float32 and max_code is None:
        max_code = 255.0 if stacks[0].dtype == torch.uint8 else 65535.0
    sizes = [int(t.shape[0]) for t in stacks]
    for e, n in zip(exposures, sizes):
        if e.numel() != n:
            raise ValueError(f"{e.numel()} exposure times for a batch of {n}")
    host_exp = torch.cat([e.detach().to("cpu", torch.float64).reshape(-1) for e in exposures])
    exposure_dev = host_exp.pin_memory().to(dev, non_blocking=True) if dev.type == "cuda" else host_exp
    icrf, lut_keep = _icrf_struct(lut, interp, c)
    geom = _geometry(stacks[0], tile, layout)
    has_std = std_mode != "none"
    first = state is None or state.batches == 0
    flags = (nv.MERGE_FIRST_BATCH if first else 0) | (nv.MERGE_FINALIZE if finalize else 0)
    if reference_order is not None:
        flags |= nv.MERGE_REFERENCE_ORDER if reference_order else nv.MERGE_CLOSED_FORM
    if require_one_launch:
        flags |= nv.MERGE_REQUIRE_ONE_LAUNCH
    out_shape = _out_shape(stacks[0], layout, out_layout)
    if out_layout == "input":
        flags |= nv.MERGE_OUT_AS_INPUT
    if mean_dtype == torch.float32:
        flags |= nv.MERGE_MEAN_OUT_F32
    elif mean_dtype != torch.float64:
        raise TypeError("mean_dtype must be float64 (reference) or float32")
    if state is None and not finalize:
        raise ValueError("a non-final call needs a MergeState")
    if state is None:
        # several batches: whatever cannot run as one launch walks them with the state in memory
        state = MergeState(out_shape, dev, has_std)
    if has_std and state.var is None:
        raise ValueError("MergeState was created without a variance buffer")
    if tuple(state.mean.shape) != out_shape:
        raise ValueError(f"MergeState has shape {tuple(state.mean.shape)}, this merge needs {out_shape}")
    mean_out = torch.empty(out_shape, dtype=mean_dtype, device=dev) if finalize else None
    std_out = torch.empty(out_shape, dtype=torch.float32, device=dev) if (finalize and has_std) else None
    ptr_arr = (ctypes.c_void_p * k)(*[t.data_ptr() for t in stacks])
    std_arr = (ctypes.c_void_p * k)(*[sd.data_ptr() for sd in stds]) if stds is not None else None
    size_arr = (ctypes.c_int32 * k)(*sizes)
    with torch.cuda.device(dev):
        rc = nv.load().ct_hdr_merge_batches(
            ptr_arr, std_arr, size_arr, k, _DTYPE[stacks[0].dtype], float(max_code or 1.0), ctypes.byref(geom), _STD[std_mode],
            float(std_value), _ptr(exposure_dev), ctypes.byref(icrf), nv.WEIGHT_GAUSS if gaussian_weight else nv.WEIGHT_NONE,
            _ptr(state.mean), _ptr(state.sumw), _ptr(state.var) if state.var is not None else None, _ptr(mean_out),
            _ptr(std_out), flags, _stream(dev))
    nv.check(rc, "ct_hdr_merge_batches")
    del lut_keep
    state.batches += k
    return (mean_out, std_out) if finalize else None


def linearize_frames(frames: torch.Tensor, lut: torch.Tensor, interp: str = "linear", *,
                     std: Optional[torch.Tensor] = None, std_mode: str = "none", std_value: float = 0.0,
                     max_code: Optional[float] = None, want_std: bool = True, tile: Optional[TileGeometry] = None,
                     layout: str = "nchw", out=None):
    """ct_linearize_std on (F,C,H,W) frames -> (lin float32, std float32 | None); every frame is its own batch.
    ``layout`` "nhwc" / "nhwc_bgr": frames are (F,H,W,C); the outputs are planar (F,C,H,W).
    ``out`` = (lin, std | None): caller-owned contiguous float32 (F,C,H,W) device buffers to write into (the streamed
    pipeline re-uses its ring slots instead of allocating per launch)."""
    _check_stack(frames, "frames")
    f = frames.shape[0]
    c, h, w = _chw(frames, layout)
    dev = frames.device
    if std is not None:
        std_mode = "explicit"
        _require_device(std, "std")
        if std.shape != frames.shape:
            raise ValueError("std shape != frames shape")
        std = std.to(torch.float32).contiguous()
        frames = frames.contiguous()
    if frames.dtype != torch.float32 and max_code is None:
        max_code = 255.0 if frames.dtype == torch.uint8 else 65535.0
    icrf, lut_keep = _icrf_struct(lut, interp, c)
    frames = frames.contiguous()
    geom = _geometry(frames, tile, layout)
    if out is None:
        lin = torch.empty((f, c, h, w), dtype=torch.float32, device=dev)
        std_out = torch.empty_like(lin) if want_std else None
    else:
        lin, std_out = out
        for name, t in (("out[0]", lin), ("out[1]", std_out)):
            if t is None:
                continue
            _require_device(t, name)
            if t.dtype != torch.float32 or tuple(t.shape) != (f, c, h, w) or not t.is_contiguous():
                raise ValueError(f"{name} must be a contiguous float32 tensor of shape {(f, c, h, w)}")
        if want_std and std_out is None:
            raise ValueError("want_std needs out[1]")
        if not want_std:
            std_out = None
    with torch.cuda.device(dev):
        rc = nv.load().ct_linearize_std(_ptr(frames), _DTYPE[frames.dtype], float(max_code or 1.0), f, ctypes.byref(geom),
                                        _ptr(std), _STD[std_mode], float(std_value), ctypes.byref(icrf), _ptr(lin),
                                        _ptr(std_out), _stream(dev))
    nv.check(rc, "ct_linearize_std")
    del lut_keep
    return lin, std_out


def icrf_forward(x: torch.Tensor, lut: torch.Tensor, interp: str, tile: Optional[TileGeometry] = None):
    """ct_linearize_fwd: ICRFModelBase.forward on a float32 (N,C,H,W) device tensor."""
    _check_stack(x, "image")
    if x.dtype != torch.float32:
        raise TypeError("icrf_forward expects float32 pixel values")
    x = x.contiguous()
    icrf, lut_keep = _icrf_struct(lut, interp, x.shape[1])
    geom = _geometry(x, tile)
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = nv.load().ct_linearize_fwd(_ptr(x), x.shape[0], ctypes.byref(geom), ctypes.byref(icrf), _ptr(out),
                                        _stream(x.device))
    nv.check(rc, "ct_linearize_fwd")
    del lut_keep
    return out


def icrf_backward(x: torch.Tensor, grad_out: torch.Tensor, lut: torch.Tensor, interp: str, need_x: bool, need_lut: bool,
                  tile: Optional[TileGeometry] = None):
    """ct_linearize_bwd: (grad wrt image | None, grad wrt LUT (C,L) | None)."""
    _check_stack(x, "image")
    x = x.contiguous()
    grad_out = grad_out.to(torch.float32).contiguous()
    icrf, lut_keep = _icrf_struct(lut, interp, x.shape[1])
    geom = _geometry(x, tile)
    gx = torch.empty_like(x) if need_x else None
    gl = torch.zeros((x.shape[1], lut.shape[1]), dtype=torch.float32, device=x.device) if need_lut else None
    with torch.cuda.device(x.device):
        rc = nv.load().ct_linearize_bwd(_ptr(x), _ptr(grad_out), x.shape[0], ctypes.byref(geom), ctypes.byref(icrf),
                                        _ptr(gx), _ptr(gl), _stream(x.device))
    nv.check(rc, "ct_linearize_bwd")
    del lut_keep
    return gx, gl


# ---- exposure-pair linearity residual (training / measure_linearity) -----------------------------------------
def _pair_params(lower, upper, use_relative, use_unc_weight, std_mode, std_value, weight_scale=10.0, pair_band=0):
    return nv.PairParams(lower=float(lower), upper=float(upper), weight_scale=float(weight_scale),
                         use_relative=int(bool(use_relative)), use_uncertainty_weighting=int(bool(use_unc_weight)),
                         std_mode=_STD[std_mode], std_value=float(std_value), pair_band=int(pair_band))


class PairList:
    """Exposure pairs on the device: (i, j, ratio) in the reference's triu order plus the per-sample partner
    lists (CSR) the backward kernel walks.  Built from get_valid_exposure_pairs' outputs (tiny, host-side)."""

    def __init__(self, i_idx: torch.Tensor, j_idx: torch.Tensor, ratio: torch.Tensor, n_images: int, device):
        i_cpu, j_cpu = i_idx.to("cpu", torch.int64), j_idx.to("cpu", torch.int64)
        self.n_pairs, self.n_images = int(i_cpu.numel()), int(n_images)
        self.i = i_cpu.to(torch.int32).to(device)
        self.j = j_cpu.to(torch.int32).to(device)
        self.ratio = ratio.to("cpu", torch.float64).to(device)
        # band of the list: every pair has 0 < j - i <= band (0 when some pair has j <= i): the hint of ct_pair_params
        distance = j_cpu - i_cpu
        self.band = int(distance.max()) if self.n_pairs and int(distance.min()) > 0 else 0
        # CSR over samples, entries of a sample in ascending pair order: (partner, p) when the sample is the pair's first
        # image, (partner, ~p) when it is the second.  Vectorised: one stable sort of the 2P (owner, pair) keys.
        p_idx = torch.arange(self.n_pairs, dtype=torch.int64)
        owner = torch.cat([i_cpu, j_cpu])
        partner = torch.cat([j_cpu, i_cpu])
        code = torch.cat([p_idx, ~p_idx])
        order = torch.argsort(owner * max(self.n_pairs, 1) + torch.cat([p_idx, p_idx]), stable=True)
        counts = torch.bincount(owner, minlength=n_images) if self.n_pairs else torch.zeros(n_images, dtype=torch.int64)
        offsets = torch.zeros(n_images + 1, dtype=torch.int64)
        offsets[1:] = torch.cumsum(counts[:n_images], dim=0)
        samples, codes = partner[order], code[order]
        self.part_off = offsets.to(torch.int32).to(device)
        self.part_sample = (samples if self.n_pairs else torch.zeros(1, dtype=torch.int64)).to(torch.int32).to(device)
        self.part_pair = (codes if self.n_pairs else torch.zeros(1, dtype=torch.int64)).to(torch.int32).to(device)
        self._workspace = {}

    def workspace(self, channels: int) -> torch.Tensor:
        """Scratch for ct_pair_residual_bwd (its per-channel partner tables), allocated once per channel count."""
        ws = self._workspace.get(channels)
        if ws is None:
            nbytes = int(nv.load().ct_pair_residual_bwd_workspace(self.n_images, self.n_pairs, channels))
            ws = torch.empty(max(nbytes, 32), dtype=torch.uint8, device=self.i.device)
            self._workspace[channels] = ws
        return ws


def pair_residual_sums(stack: torch.Tensor, pairs: PairList, *, lut: Optional[torch.Tensor], interp: Optional[str],
                       lower: float, upper: float, use_relative: bool, use_unc_weight: bool,
                       std: Optional[torch.Tensor] = None, std_mode: str = "none", std_value: float = 0.0,
                       max_code: Optional[float] = None, level: int = 1, tile: Optional[TileGeometry] = None,
                       center: Optional[torch.Tensor] = None, layout: str = "nchw"):
    """ct_pair_residual_fwd -> (P, C, 5) float64 sums [sum w m, sum v w m, sum (v-center)^2 w m, sum err m, sum m].
    ``layout`` "nhwc" / "nhwc_bgr": the stack (and an explicit std stack) is (N,H,W,C) as OpenCV decodes it."""
    _check_stack(stack)
    n = stack.shape[0]
    c, _, _ = _chw(stack, layout)
    dev = stack.device
    if n != pairs.n_images:
        raise ValueError(f"pair list was built for {pairs.n_images} images, stack has {n}")
    if std is not None:
        std_mode = "explicit"
        _require_device(std, "std")
        if std.shape != stack.shape:
            raise ValueError("std shape != stack shape")
        std = std.to(torch.float32).contiguous()
        stack = stack.contiguous()
    if stack.dtype != torch.float32 and max_code is None:
        max_code = 255.0 if stack.dtype == torch.uint8 else 65535.0
    icrf, lut_keep = _icrf_struct(lut, interp, c)
    geom = _geometry(stack, tile, layout)
    prm = _pair_params(lower, upper, use_relative, use_unc_weight, std_mode, std_value)
    sums = torch.zeros((pairs.n_pairs, c, 5), dtype=torch.float64, device=dev)
    if center is not None:
        center = center.to(device=dev, dtype=torch.float64).contiguous()
        if center.shape != (pairs.n_pairs, c):
            raise ValueError(f"center must be (P={pairs.n_pairs}, C={c})")
    if pairs.n_pairs:
        with torch.cuda.device(dev):
            rc = nv.load().ct_pair_residual_fwd(_ptr(stack), _DTYPE[stack.dtype], float(max_code or 1.0), n,
                                                ctypes.byref(geom), _ptr(std), ctypes.byref(icrf), _ptr(pairs.i),
                                                _ptr(pairs.j), _ptr(pairs.ratio), pairs.n_pairs, ctypes.byref(prm),
                                                int(level), _ptr(center), _ptr(sums), _stream(dev))
        nv.check(rc, "ct_pair_residual_fwd")
    del lut_keep
    return sums


def pair_residual_lut_grad(stack: torch.Tensor, pairs: PairList, coef: torch.Tensor, *, lut: torch.Tensor, interp: str,
                           lower: float, upper: float, use_relative: bool, max_code: Optional[float] = None,
                           tile: Optional[TileGeometry] = None, use_unc_weight: bool = False,
                           std: Optional[torch.Tensor] = None, std_mode: str = "none", std_value: float = 0.0,
                           smean: Optional[torch.Tensor] = None, lane_kernel: bool = True, layout: str = "nchw"):
    """ct_pair_residual_bwd -> (C, L) float64 LUT gradient of sum_pc coef_pc * D_pc * mean_pc (coef = dL/dmean / D).
    With ``use_unc_weight`` and uncertainties the weights depend on the LUT and ``smean`` (P,C) is required.
    ``lane_kernel=False`` withholds the pair list's band hint, i.e. forces the generic backward kernel (tests).
    ``layout`` as in ``pair_residual_sums``."""
    _check_stack(stack)
    n = stack.shape[0]
    c, _, _ = _chw(stack, layout)
    dev = stack.device
    if std is not None:
        std_mode = "explicit"
        std = std.to(device=dev, dtype=torch.float32).contiguous()
        stack = stack.contiguous()
    if not use_unc_weight:
        std, std_mode = None, "none"
    if stack.dtype != torch.float32 and max_code is None:
        max_code = 255.0 if stack.dtype == torch.uint8 else 65535.0
    icrf, lut_keep = _icrf_struct(lut, interp, c)
    geom = _geometry(stack, tile, layout)
    prm = _pair_params(lower, upper, use_relative, use_unc_weight, std_mode, std_value,
                       pair_band=pairs.band if lane_kernel else 0)
    if std_mode != "none":
        if smean is None:
            raise ValueError("the uncertainty-weighted backward needs the forward's spatial means")
        smean = smean.to(device=dev, dtype=torch.float64).contiguous()
    else:
        smean = None
    coef = coef.to(device=dev, dtype=torch.float64).contiguous()
    if coef.shape != (pairs.n_pairs, c):
        raise ValueError(f"coef must be (P={pairs.n_pairs}, C={c}), got {tuple(coef.shape)}")
    grad = torch.zeros((c, lut.shape[1]), dtype=torch.float64, device=dev)
    if pairs.n_pairs:
        ws = pairs.workspace(c)
        with torch.cuda.device(dev):
            rc = nv.load().ct_pair_residual_bwd(_ptr(stack), _DTYPE[stack.dtype], float(max_code or 1.0), n,
                                                ctypes.byref(geom), _ptr(std), ctypes.byref(icrf), _ptr(pairs.ratio),
                                                pairs.n_pairs, _ptr(pairs.part_off), _ptr(pairs.part_sample),
                                                _ptr(pairs.part_pair), ctypes.byref(prm), _ptr(coef), _ptr(smean),
                                                _ptr(grad), _ptr(ws), ws.numel(), _stream(dev))
        nv.check(rc, "ct_pair_residual_bwd")
    del lut_keep
    return grad


# ---- per-band statistics (BASELINE configuration C5) -------------------------------------------------------------
def band_stats(mean: torch.Tensor, std: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ct_band_stats: (6, C) float64 = [min mean, max mean, sum mean, min std, max std, sum std] per channel of a merged
    (C, H_band, W) band in ONE pass (``std`` None: the std rows are zero).  min / max / sum combine over row bands."""
    _require_device(mean, "mean")
    if mean.dtype != torch.float64 or mean.dim() != 3:
        raise ValueError("mean must be a (C, H, W) float64 device tensor")
    mean = mean.contiguous()
    c = mean.shape[0]
    plane = mean.shape[1] * mean.shape[2]
    if std is not None:
        _require_device(std, "std")
        if std.dtype != torch.float32 or std.shape != mean.shape:
            raise ValueError("std must be float32 with the mean's shape")
        std = std.contiguous()
    lib = nv.load()
    ws_bytes = int(lib.ct_band_stats_workspace(c))
    ws = torch.empty((ws_bytes // 8,), dtype=torch.float64, device=mean.device)
    out = torch.empty((6, c), dtype=torch.float64, device=mean.device)
    with torch.cuda.device(mean.device):
        rc = lib.ct_band_stats(_ptr(mean), _ptr(std), c, plane, _ptr(ws), ws_bytes, _ptr(out), _stream(mean.device))
    nv.check(rc, "ct_band_stats")
    return out


# ---- flat-field correction epilogues ----------------------------------------------------------------------------
def flatfield_correct(value: torch.Tensor, var_or_std: Optional[torch.Tensor], flat: torch.Tensor,
                      flat_std: Optional[torch.Tensor], *, input_is_variance: bool, through_mean: bool,
                      global_pixels: Optional[int] = None, reduce=None):
    """In-place flat-field correction of ``value`` ((C,H,W) float64 merged mean or (F,C,H,W) float32 frames) and of
    its uncertainty (ct_flatfield_sums + ct_flatfield_apply).  ``through_mean``: the gradient also flows through the
    flat field's spatial mean (compute_hdr_image) or not (linearize).  ``reduce`` all-reduces the (C,2) sums across
    ranks holding row bands; ``global_pixels`` is then the pixel count of the whole image plane."""
    _require_device(value, "value")
    flat = flat.to(device=value.device, dtype=torch.float32).contiguous()
    if flat.ndim == 4:
        flat = flat[0]
    c, h, w = flat.shape
    plane = h * w
    if tuple(value.shape[-3:]) != (c, h, w):
        raise ValueError(f"flat field {tuple(flat.shape)} does not match the image {tuple(value.shape)}")
    if not value.is_contiguous():
        raise ValueError("value must be contiguous")
    frames = 1 if value.ndim == 3 else value.shape[0]
    is_f64 = value.dtype == torch.float64
    if value.dtype not in (torch.float64, torch.float32):
        raise TypeError("value must be float32 or float64")
    if flat_std is not None:
        flat_std = flat_std.to(device=value.device, dtype=torch.float32).contiguous()
        if flat_std.ndim == 4:
            flat_std = flat_std[0]
    dev = value.device
    sums = torch.zeros((c, 2), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = nv.load().ct_flatfield_sums(_ptr(value) if (through_mean and frames == 1) else None, int(is_f64), _ptr(flat),
                                         c, plane, _ptr(sums), _stream(dev))
    nv.check(rc, "ct_flatfield_sums")
    if reduce is not None:
        reduce(sums)
    n_px = float(global_pixels if global_pixels is not None else plane)
    flat_mean = (sums[:, 0] / n_px).to(torch.float32).contiguous()
    through = (sums[:, 1] / n_px).contiguous() if through_mean else None
    with torch.cuda.device(dev):
        rc = nv.load().ct_flatfield_apply(_ptr(value), int(is_f64), frames, _ptr(var_or_std), int(input_is_variance),
                                          _ptr(flat), _ptr(flat_std), _ptr(flat_mean), _ptr(through), c, plane, _stream(dev))
    nv.check(rc, "ct_flatfield_apply")
    return value, var_or_std


# ---- dark-field conditional blur -----------------------------------------------------------------------------------
def dark_field_blur(stack: torch.Tensor, dark: torch.Tensor, dark_std: Optional[torch.Tensor], *,
                    std: Optional[torch.Tensor] = None, std_mode: str = "none", std_value: float = 0.0,
                    max_code: Optional[float] = None, tile: Optional[TileGeometry] = None,
                    halo: Optional[torch.Tensor] = None, threshold: float = 0.05, alpha: float = 50.0):
    """ct_dark_field_blur: (xb float32 (B,C,H,W), sigma_eff float32 | None).  ``dark`` / ``dark_std`` are (1|B,C,H,W);
    ``halo`` (B,C,2,W) holds the global rows above / below a row band (see include/clair_hip.h)."""
    _check_stack(stack)
    b, c, h, w = stack.shape
    dev = stack.device
    if std is not None:
        std_mode = "explicit"
        _require_device(std, "std")
        if std.shape != stack.shape:
            raise ValueError("std shape != stack shape")
        std = std.to(torch.float32).contiguous()
    stack = stack.contiguous()
    if stack.dtype != torch.float32 and max_code is None:
        max_code = 255.0 if stack.dtype == torch.uint8 else 65535.0
    dark = dark.to(device=dev, dtype=torch.float32).contiguous()
    if dark.ndim != 4 or dark.shape[0] not in (1, b) or tuple(dark.shape[1:]) != (c, h, w):
        raise ValueError(f"mask_map batch dimension must be 1 or {b}, got shape {tuple(dark.shape)}")
    if dark_std is not None:
        dark_std = dark_std.to(device=dev, dtype=torch.float32).contiguous()
        if dark_std.shape != dark.shape:
            raise ValueError("dark_std shape != dark shape")
        if dark.shape[0] == 1 and b > 1:
            raise NotImplementedError(
                "one shared dark field for several frames with its uncertainty: the reference sums the dark-field gradient "
                "over the frames before squaring, which the per-frame effective sigma of this kernel cannot express; pass "
                "one (matched) dark field per frame, as get_matching_artefact_images does")
    if halo is not None:
        halo = halo.to(device=dev, dtype=stack.dtype).contiguous()
        if tuple(halo.shape) != (b, c, 2, w):
            raise ValueError(f"halo must be (B, C, 2, W) = {(b, c, 2, w)}, got {tuple(halo.shape)}")
    geom = _geometry(stack, tile)
    xb = torch.empty((b, c, h, w), dtype=torch.float32, device=dev)
    sig = torch.empty_like(xb) if dark_std is not None else None
    with torch.cuda.device(dev):
        rc = nv.load().ct_dark_field_blur(_ptr(stack), _DTYPE[stack.dtype], float(max_code or 1.0), b, ctypes.byref(geom),
                                          _ptr(halo), _ptr(std), _STD[std_mode], float(std_value), _ptr(dark),
                                          _ptr(dark_std), dark.shape[0], float(threshold), float(alpha), _ptr(xb), _ptr(sig),
                                          _stream(dev))
    nv.check(rc, "ct_dark_field_blur")
    return xb, sig


# ---- streaming video statistics -----------------------------------------------------------------------------------
def video_stats_batch(frames: torch.Tensor, mean_state: torch.Tensor, m2_state: torch.Tensor, frames_before: int, *,
                      lut: Optional[torch.Tensor] = None, interp: Optional[str] = None,
                      max_code: Optional[float] = None, tile: Optional[TileGeometry] = None, layout: str = "nchw"):
    """ct_video_stats_batch: merge one batch of frames into the running (mean, m2) float32 state in place.
    ``layout`` "nhwc" / "nhwc_bgr": frames are (F,H,W,C) as OpenCV decodes them; the state stays planar (C,H,W)."""
    _check_stack(frames, "frames")
    b = frames.shape[0]
    c, h, w = _chw(frames, layout)
    dev = frames.device
    frames = frames.contiguous()
    if frames.dtype != torch.float32 and max_code is None:
        max_code = 255.0 if frames.dtype == torch.uint8 else 65535.0
    for name, t in (("mean_state", mean_state), ("m2_state", m2_state)):
        _require_device(t, name)
        if t.dtype != torch.float32 or tuple(t.shape) != (c, h, w) or not t.is_contiguous():
            raise ValueError(f"{name} must be a contiguous float32 (C,H,W) tensor")
    icrf, lut_keep = _icrf_struct(lut, interp, c)
    geom = _geometry(frames, tile, layout)
    with torch.cuda.device(dev):
        rc = nv.load().ct_video_stats_batch(_ptr(frames), _DTYPE[frames.dtype], float(max_code or 1.0), b,
                                            ctypes.byref(geom), ctypes.byref(icrf), float(frames_before),
                                            _ptr(mean_state), _ptr(m2_state), _stream(dev))
    nv.check(rc, "ct_video_stats_batch")
    del lut_keep
