"""GPU parity tests of ICRF linearization: ct_linearize_std / ct_linearize_fwd / ct_linearize_bwd."""
import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

from _util import assert_parity, golden, std_for

pytestmark = pytest.mark.gpu
MODES = ("lookup", "linear", "catmull")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from clair_torch_amd import _native
    _native.load()
    return torch.device("cuda:0")


@pytest.mark.parametrize("case", ["a", "b", "c"])
@pytest.mark.parametrize("mode", MODES)
def test_model_forward_bit_exact(dev, case, mode):
    """ICRFModelDirect.forward equals the reference bit for bit in all three interpolation modes, incl. exact knots,
    0, 1 and out-of-range values, on shapes where H*W is not a multiple of C (row quirk)."""
    from clair_torch_amd.common.enums import InterpMode
    from clair_torch_amd.models import ICRFModelDirect
    g = golden("model_forward")
    model = ICRFModelDirect(icrf=torch.from_numpy(g["fwd_lut"]), interpolation_mode=InterpMode[mode.upper()]).to(dev)
    out = model(torch.from_numpy(g[f"fwd_{case}_x"]).to(dev))
    assert np.array_equal(out.cpu().numpy(), g[f"fwd_{case}_{mode}"])


@pytest.mark.parametrize("bits", [8, 16])
def test_every_code_bit_exact(dev, bits):
    """All 256 / 65536 integer codes through in-kernel normalisation + LUT: value bit-exact (uint LUT indexing)."""
    from clair_torch_amd import ops
    g = golden("model_forward")
    lut = torch.from_numpy(g["fwd_lut"]).to(dev)
    u = torch.arange(2 ** bits, dtype=torch.int32).to(torch.uint8 if bits == 8 else torch.uint16)
    frames = u.view(1, 1, 1, -1).repeat(1, 3, 1, 1).contiguous().to(dev)
    for mode in MODES:
        lin, _ = ops.linearize_frames(frames, lut, mode, want_std=False)
        assert np.array_equal(lin.cpu().numpy(), g[f"codes{bits}_{mode}"]), mode


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("sname", ["none", "multiplier", "explicit"])
def test_linearize_generator_vs_golden(dev, mode, sname):
    """linearize_dataset_generator: value bit-exact; std bit-exact for LINEAR (CATMULL: the reference's own
    float32 autograd noise bounds the comparison, see test_oracle_golden)."""
    from clair_torch_amd.common.enums import InterpMode, MissingStdMode
    from clair_torch_amd.common.transforms import CastTo, Normalize
    from clair_torch_amd.datasets import StackDataset, custom_collate
    from clair_torch_amd.inference import linearize_dataset_generator
    from clair_torch_amd.models import ICRFModelDirect
    from oracle import ct_oracle as oc
    g = golden("linearize")
    codes = torch.from_numpy(g["lin_codes"])
    model = ICRFModelDirect(icrf=torch.from_numpy(g["lin_lut"]), interpolation_mode=InterpMode[mode.upper()]).to(dev)
    if sname == "explicit":
        x = torch.from_numpy(oc.normalize_codes(g["lin_codes"]))
        ds = StackDataset(x, [0.01, 0.02, 0.04], stds=torch.from_numpy(g["lin_explicit_std"]))
        tf = None
    elif sname == "multiplier":
        ds = StackDataset(codes, [0.01, 0.02, 0.04], missing_std_mode=MissingStdMode.MULTIPLIER, missing_std_value=0.05,
                          materialize_std=False)
        tf = [CastTo("float32"), Normalize(max_val=65535, min_val=0)]
    else:
        ds = StackDataset(codes, [0.01, 0.02, 0.04])
        tf = [CastTo("float32"), Normalize(max_val=65535, min_val=0)]
    loader = DataLoader(ds, batch_size=1, shuffle=False, collate_fn=custom_collate)
    if mode == "lookup" and sname != "none":
        with pytest.raises(RuntimeError, match="does not require grad"):
            next(iter(linearize_dataset_generator(loader, "cuda", model, gpu_transforms=tf)))
        return
    outs = list(linearize_dataset_generator(loader, "cuda", model, gpu_transforms=tf))
    assert len(outs) == 3
    for f, (lin, sd, meta) in enumerate(outs):
        assert lin.device.type == "cpu" and sd.device.type == "cpu" and lin.dtype == torch.float32
        assert np.array_equal(lin.numpy(), g[f"lin_{mode}_{sname}_val"][f])
        # every mode bit for bit: CATMULL's derivative is evaluated in the reference's autograd order
        # (ct_device.hpp catmull_backward_ref; the emulation in oracle/eager_torch.py pins that order on the CPU)
        assert np.array_equal(sd.numpy(), g[f"lin_{mode}_{sname}_std"][f]), (mode, sname, f)
        assert float(meta["exposure_time"]) == [0.01, 0.02, 0.04][f]


def test_linearize_batch_size_must_be_one(dev):
    from clair_torch_amd.datasets import StackDataset, custom_collate
    from clair_torch_amd.inference import linearize_dataset_generator
    from clair_torch_amd.models import ICRFModelDirect
    ds = StackDataset(torch.rand(4, 3, 8, 8), [1, 2, 3, 4])
    loader = DataLoader(ds, batch_size=2, collate_fn=custom_collate)
    with pytest.raises(ValueError, match="batch_size of 1"):
        next(iter(linearize_dataset_generator(loader, "cuda", ICRFModelDirect().to(dev))))


@pytest.mark.parametrize("mode", MODES)
def test_backward_matches_eager_autograd(dev, mode):
    """Image gradient and (C,L) LUT gradient of the HIP forward/backward pair vs the eager-PyTorch restatement."""
    from clair_torch_amd.common.enums import InterpMode
    from clair_torch_amd.models import ICRFModelDirect
    from oracle import eager_torch as oe
    gen = torch.Generator().manual_seed(11)
    x = torch.rand((3, 3, 17, 13), generator=gen)
    x.view(-1)[:4] = torch.tensor([0.0, 1.0, 100 / 255, 1.25])
    go = torch.randn((3, 3, 17, 13), generator=gen)
    lut0 = torch.stack([torch.linspace(0, 1, 64) ** p for p in (1.5, 2.0, 2.5)])
    # oracle
    xo, lo = x.clone().requires_grad_(True), lut0.clone().requires_grad_(True)
    out_o = oe.icrf_forward(xo, lo, mode)
    grads = torch.autograd.grad(out_o, [lo] + ([xo] if mode != "lookup" else []), go)
    # HIP
    model = ICRFModelDirect(icrf=lut0.clone(), interpolation_mode=InterpMode[mode.upper()]).to(dev)
    with torch.no_grad():
        for c in range(3):
            model.direct_params[c].copy_(lut0[c])
    model.update_icrf()
    xd = x.to(dev).requires_grad_(True)
    out = model(xd)
    assert np.array_equal(out.detach().cpu().numpy(), out_o.detach().numpy())
    out.backward(go.to(dev))
    lut_grad = torch.stack([p.grad for p in model.direct_params]).cpu()
    assert_parity(lut_grad.numpy(), grads[0].numpy(), rtol=1e-5, norm_tol=1e-6, what="lut grad")
    if mode != "lookup":
        # the upstream gradient multiplies the LUT taps first, as in autograd: bit for bit (signed zeros aside)
        assert np.array_equal(xd.grad.cpu().numpy(), grads[1].numpy()), mode


def test_linearize_streamed_frames_vs_oracle(dev):
    """Config C4 shape (1920x1080x3 frames, uint8 and uint16, many frames per launch) against the C oracle."""
    from clair_torch_amd import ops
    from oracle import ct_oracle as oc
    rng = np.random.default_rng(5)
    lut = np.stack([np.linspace(0, 1, 256, dtype=np.float32) ** np.float32(p) for p in (2.2, 2.4, 2.6)])
    for dt, hi in ((np.uint8, 256), (np.uint16, 65536)):
        codes = rng.integers(0, hi, size=(4, 3, 1080, 1920)).astype(dt)
        lin, sd = ops.linearize_frames(torch.from_numpy(codes).to(dev), torch.from_numpy(lut).to(dev), "linear",
                                       std_mode="multiplier", std_value=0.05)
        x = oc.normalize_codes(codes)
        lin_o, sd_o = oc.linearize_std(x, x * np.float32(0.05), lut, "linear")
        assert np.array_equal(lin.cpu().numpy(), lin_o) and np.array_equal(sd.cpu().numpy(), sd_o)


def test_linearize_std_whose_square_underflows(dev):
    """linearization.py:106,132 forms sqrt((grad * std) ** 2) in float32: for 0 < |grad * std| < ~1e-19 the square is
    denormal or zero and the result is NOT |grad * std|.  The kernel takes |.| on the fast path and the correctly rounded
    sqrtf of the rounded square behind a wave-uniform branch; wavefronts that mix ordinary and underflowing samples, a
    constant sigma below the threshold and one far below it (square flushes to zero), bit for bit against the oracle."""
    from clair_torch_amd import ops
    from oracle import ct_oracle as oc
    rng = np.random.default_rng(18)
    lut = np.stack([np.linspace(0, 1, 256, dtype=np.float32) ** np.float32(p) for p in (2.2, 2.4, 2.6)])
    codes = rng.integers(0, 65536, size=(2, 3, 33, 64)).astype(np.uint16)
    x = oc.normalize_codes(codes)
    sigma = np.full(codes.shape, 0.01, dtype=np.float32)
    pick = rng.random(codes.shape)
    sigma[pick < 0.02] = np.float32(3e-20)   # square is a float32 denormal: precision lost, sqrt != |.|
    sigma[pick < 0.01] = np.float32(1e-30)   # square flushes to zero
    sigma[0, 0, :2] = np.float32(2e-19)      # whole rows (whole wavefronts) in the slow branch
    lin_o, sd_o = oc.linearize_std(x, sigma, lut, "linear")
    assert (sd_o != np.abs(sd_o)).sum() == 0 and ((sd_o == 0) & (sigma > 0) & (x > 0) & (x < 1)).any()
    lin, sd = ops.linearize_frames(torch.from_numpy(codes).to(dev), torch.from_numpy(lut).to(dev), "linear",
                                   std=torch.from_numpy(sigma).to(dev))
    assert np.array_equal(lin.cpu().numpy(), lin_o) and np.array_equal(sd.cpu().numpy(), sd_o)
    for value in (4e-20, 1e-30):
        const = np.full(codes.shape, value, dtype=np.float32)
        _, sd_o = oc.linearize_std(x, const, lut, "linear")
        _, sd = ops.linearize_frames(torch.from_numpy(codes).to(dev), torch.from_numpy(lut).to(dev), "linear",
                                     std_mode="constant", std_value=value)
        assert np.array_equal(sd.cpu().numpy(), sd_o)


@pytest.mark.parametrize("mode", MODES)
def test_linearize_codes_above_max_code(dev, mode):
    """uint16 codes above max_code = 4095 (x > 1): value clamps to the top of the LUT, the derivative (and with it the
    std) is zero there, exactly as the reference's clamp in the model (base.py:146,166,190) -- against the oracle."""
    from clair_torch_amd import ops
    from oracle import ct_oracle as oc
    rng = np.random.default_rng(40)
    codes = rng.integers(0, 4096, size=(2, 3, 9, 13)).astype(np.uint16)
    codes.reshape(-1)[::5] = rng.integers(4096, 9000, size=codes.reshape(-1)[::5].shape).astype(np.uint16)
    codes.reshape(-1)[:3] = [4095, 4096, 65535]
    x = (codes.astype(np.float32) / np.float32(4095.0)).astype(np.float32)
    lut = np.stack([np.linspace(0, 1, 256, dtype=np.float32) ** np.float32(p) for p in (1.7, 2.2, 2.7)])
    if mode == "lookup":  # no gradient path: the reference raises with uncertainties (linearization.py:100), value only
        lin_o, _ = oc.linearize_std(x, None, lut, mode)
        lin, _ = ops.linearize_frames(torch.from_numpy(codes).to(dev), torch.from_numpy(lut).to(dev), mode,
                                      want_std=False, max_code=4095.0)
        assert np.array_equal(lin.cpu().numpy(), lin_o)
        return
    lin_o, sd_o = oc.linearize_std(x, x * np.float32(0.05), lut, mode)
    lin, sd = ops.linearize_frames(torch.from_numpy(codes).to(dev), torch.from_numpy(lut).to(dev), mode,
                                   std_mode="multiplier", std_value=0.05, max_code=4095.0)
    assert np.array_equal(lin.cpu().numpy(), lin_o)
    assert np.array_equal(sd.cpu().numpy(), sd_o)


@pytest.mark.parametrize("pinned", [True, False])
@pytest.mark.parametrize("kind", ["u16", "f32_explicit_std", "u8_flat"])
def test_streamed_pipeline_is_bit_exact_and_ordered(dev, kind, pinned):
    """linearize_dataset_generator's pipelined route (groups of frames in flight on copy / compute / copy streams): every
    yielded frame equals the oracle bit for bit (value; LINEAR std), in dataset order with its own metadata, for more
    frames than ring slots x group size, a ragged last group, pinned and pageable sources, explicit uncertainty
    images, and the flat-field epilogue.  The frame-by-frame route (a non-fusable transform list) yields the same."""
    from clair_torch_amd.common.enums import InterpMode, MissingStdMode
    from clair_torch_amd.common.transforms import BaseTransform, CastTo, Normalize
    from clair_torch_amd.datasets import ArtefactStack, StackDataset, custom_collate
    from clair_torch_amd.inference import linearization, linearize_dataset_generator
    from clair_torch_amd.models import ICRFModelDirect
    from oracle import ct_oracle as oc
    rng = np.random.default_rng(99)
    n, c, h, w = 23, 3, 37, 41
    lut = np.stack([np.linspace(0, 1, 256, dtype=np.float32) ** np.float32(p) for p in (1.8, 2.2, 2.6)])
    model = ICRFModelDirect(icrf=torch.from_numpy(lut), interpolation_mode=InterpMode.LINEAR).to(dev)
    times = [float(k + 1) for k in range(n)]
    flat = flat_std = ff = None
    if kind == "u16":
        codes = rng.integers(0, 65536, size=(n, c, h, w)).astype(np.uint16)
        x = oc.normalize_codes(codes)
        sd = x * np.float32(0.05)
        vals = torch.from_numpy(codes)
        ds = StackDataset(vals.pin_memory() if pinned else vals, times, missing_std_mode=MissingStdMode.MULTIPLIER,
                          missing_std_value=0.05, materialize_std=False)
        tf = [CastTo("float32"), Normalize(65535, 0)]
    elif kind == "u8_flat":
        codes = rng.integers(0, 256, size=(n, c, h, w)).astype(np.uint8)
        x = oc.normalize_codes(codes)
        sd = x * np.float32(0.05)
        vals = torch.from_numpy(codes)
        ds = StackDataset(vals.pin_memory() if pinned else vals, times, missing_std_mode=MissingStdMode.MULTIPLIER,
                          missing_std_value=0.05, materialize_std=False)
        tf = [CastTo("float32"), Normalize(255, 0)]
        flat = (0.6 + 0.4 * rng.random((c, h, w))).astype(np.float32)
        flat_std = (0.01 * rng.random((c, h, w))).astype(np.float32)
        ff = ArtefactStack(torch.from_numpy(flat), torch.from_numpy(flat_std))
    else:
        x = rng.random((n, c, h, w), dtype=np.float32)
        sd = (0.001 + 0.02 * rng.random((n, c, h, w))).astype(np.float32)
        vx, vs = torch.from_numpy(x), torch.from_numpy(sd)
        ds = StackDataset(vx.pin_memory() if pinned else vx, times, stds=vs.pin_memory() if pinned else vs)
        tf = None
    lin_o, sd_o = oc.linearize_std(x, sd, lut, "linear")
    if flat is not None:
        lin_o, sd_o = oc.flatfield_linearize(lin_o, sd_o, flat, flat_std)
    old = linearization._GROUP_BYTES
    linearization._GROUP_BYTES = 2 * 4 * c * h * w * 3        # groups of 3 frames: 23 frames = 7 full groups + 2
    try:
        loader = DataLoader(ds, batch_size=1, shuffle=False, collate_fn=custom_collate)
        got = list(linearize_dataset_generator(loader, "cuda", model, flatfield_dataset=ff, gpu_transforms=tf))
    finally:
        linearization._GROUP_BYTES = old
    assert len(got) == n
    for k, (lin, sdv, meta) in enumerate(got):
        assert lin.device.type == "cpu" and lin.shape == (c, h, w) and float(meta["exposure_time"]) == times[k]
        assert np.array_equal(lin.numpy(), lin_o[k]), k
        assert np.array_equal(sdv.numpy(), sd_o[k]), k
    if kind == "u16":   # the generic frame-by-frame route gives the same frames
        class Identity(BaseTransform):
            def __call__(self, t):
                return t
        slow = list(linearize_dataset_generator(DataLoader(ds, batch_size=1, shuffle=False, collate_fn=custom_collate),
                                                "cuda", model, gpu_transforms=[Identity()] + tf))
        assert len(slow) == n
        # that route normalises with torch's GPU division (1 ulp off the CPU reference on some codes): value to rounding
        assert max(float((a[0] - b[0]).abs().max()) for a, b in zip(slow, got)) < 1e-6


def test_torch_library_ops_run_the_kernels(dev):
    """torch.ops.clair_hip.* (the dispatcher-registered form of the entry points) give the results of the ctypes
    front-end, and the registered autograd formula of icrf_forward returns the backward kernel's gradients."""
    from clair_torch_amd import ops, torch_ops  # noqa: F401
    rng = np.random.default_rng(8)
    x = torch.from_numpy(rng.random((2, 3, 9, 11), dtype=np.float32)).to(dev)
    lut = torch.stack([torch.linspace(0, 1, 64) ** p for p in (1.8, 2.2, 2.6)]).to(dev)
    assert torch.equal(torch.ops.clair_hip.icrf_forward(x, lut, "catmull"), ops.icrf_forward(x, lut, "catmull"))
    xg, lg = x.clone().requires_grad_(True), lut.clone().requires_grad_(True)
    out = torch.ops.clair_hip.icrf_forward(xg, lg, "linear")
    g = torch.from_numpy(rng.random(out.shape, dtype=np.float32)).to(dev)
    gx, gl = torch.autograd.grad(out, (xg, lg), g)
    rx, rl = ops.icrf_backward(x, g, lut, "linear", True, True)
    assert torch.equal(gx, rx) and torch.allclose(gl, rl, rtol=1e-5, atol=1e-7)   # LUT gradient: atomics, order-dependent
    codes = torch.from_numpy(rng.integers(0, 65536, size=(5, 3, 9, 11)).astype(np.uint16)).to(dev)
    t = torch.tensor([0.001 * 2.0 ** k for k in range(5)], dtype=torch.float64)
    lut256 = torch.stack([torch.linspace(0, 1, 256) ** p for p in (1.8, 2.2, 2.6)]).to(dev)
    m, s = torch.ops.clair_hip.hdr_merge(codes, t, lut256, "linear", True, None, "multiplier", 0.05, 65535.0)
    m2, s2 = ops.hdr_merge_batch(codes, t, lut=lut256, interp="linear", std_mode="multiplier", std_value=0.05)
    assert torch.equal(m, m2) and torch.equal(s, s2)
    lin, sd = torch.ops.clair_hip.linearize_std(codes, lut256, "linear", None, "multiplier", 0.05, 65535.0)
    lin2, sd2 = ops.linearize_frames(codes, lut256, "linear", std_mode="multiplier", std_value=0.05)
    assert torch.equal(lin, lin2) and torch.equal(sd, sd2)
