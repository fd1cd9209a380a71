"""GPU parity tests of the exposure-pair linearity kernels (ct_pair_residual_fwd / _bwd), measure_linearity and
train_icrf against vectors recorded from the reference (tests/golden/training.npz) and the eager oracle."""
import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

from _util import assert_parity, golden, rel_norm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from clair_torch_amd import _native
    _native.load()
    return torch.device("cuda:0")


def _inputs(g, dev, as_codes=True):
    from oracle import ct_oracle as oc
    codes = g["train_codes"]
    x = oc.normalize_codes(codes)
    stack = torch.from_numpy(codes if as_codes else x).to(dev)
    return stack, x


def _pairs(g, dev, thr):
    from clair_torch_amd import ops
    from clair_torch_amd.common.general_functions import get_valid_exposure_pairs
    t = torch.from_numpy(g["train_exposures"])
    i, j, r = get_valid_exposure_pairs(t, thr)
    return ops.PairList(i, j, r, t.numel(), dev), (i, j, r)


def test_exposure_pairs_match_reference(dev):
    g = golden("training")
    _, (i, j, r) = _pairs(g, dev, 0.25)
    assert np.array_equal(i.numpy(), g["train_i_idx"]) and np.array_equal(j.numpy(), g["train_j_idx"])
    assert np.array_equal(r.numpy(), g["train_ratio"])


@pytest.mark.parametrize("as_codes", [True, False])
@pytest.mark.parametrize("sname", ["none", "multiplier"])
@pytest.mark.parametrize("mode", ["linear", "catmull"])
def test_pair_sums_vs_golden(dev, as_codes, sname, mode):
    """Spatial mean / std / error and the mask popcount of every pair and channel, all four loss variants."""
    from clair_torch_amd import ops
    from clair_torch_amd.training import spatial_mean, spatial_statistics
    g = golden("training")
    stack, _ = _inputs(g, dev, as_codes)
    pairs, _ = _pairs(g, dev, 0.25)
    lut = torch.from_numpy(g["train_lut0"]).to(dev)
    for rel in (True, False):
        for unc in (True, False):
            kw = dict(std_mode="multiplier", std_value=0.05) if sname == "multiplier" else {}
            kw.update(lut=lut, interp=mode, lower=1 / 255, upper=254 / 255, use_relative=rel, use_unc_weight=unc, level=1)
            sums = ops.pair_residual_sums(stack, pairs, **kw)
            assert np.array_equal(sums[..., 4].cpu().numpy(), g[f"train_{sname}_mask_popcount"].astype(np.float64))
            centered = ops.pair_residual_sums(stack, pairs, center=spatial_mean(sums), **kw)
            mean, sd, err = spatial_statistics(sums, centered, sname != "none")
            key = f"train_{sname}_{mode}_{'rel' if rel else 'abs'}_{'unc' if unc else 'nounc'}"
            assert_parity(mean.cpu().numpy(), g[key + "_spatial"], rtol=1e-5, norm_tol=2e-6, what=key + " mean")
            assert_parity(sd.cpu().numpy(), g[key + "_spatial_std"], rtol=1e-5, norm_tol=2e-6, what=key + " std")
            if sname != "none":
                assert_parity(err.cpu().numpy(), g[key + "_spatial_err"], rtol=1e-5, norm_tol=2e-6, what=key + " err")


@pytest.mark.parametrize("n_points", [256, 16, 52])
@pytest.mark.parametrize("shape", [(3, 24, 64), (3, 7, 9)])
def test_code_domain_staging_equals_generic_staging(dev, n_points, shape):
    """uint16 stacks at full range with a whole-step LINEAR curve are staged in the code domain (interval by the round-down
    FMA, f = g[i] + slope (code - i step), mask as a code interval); the same data as normalised float32 goes through the
    generic staging (the reference's float32 order).  Forward sums, mask popcounts (exactly) and the LUT gradient must
    agree, codes ON the validity thresholds, one below / above them and on LUT knots included; vectorised and ragged tiles."""
    from clair_torch_amd import ops
    from clair_torch_amd.common.general_functions import get_valid_exposure_pairs
    from oracle import ct_oracle as oc
    rng = np.random.default_rng(n_points * 100 + shape[1])
    n = 9
    # random codes: residuals of order one.  (On a synthetic, perfectly consistent stack the residuals sit at the uint16
    # quantisation level and sign(residual) -- hence the gradient -- flips with ANY 1e-7 change of the linearized values,
    # in the reference's own float32 arithmetic as much as here.)
    codes = rng.integers(0, 65536, size=(n,) + shape).astype(np.uint16)
    exposures = 0.002 * 1.3 ** np.arange(n)
    step = 65535 // (n_points - 1)
    lo, hi = 1 / 255, 254 / 255
    u = np.arange(65536, dtype=np.float32) / np.float32(65535.0)
    c_lo, c_hi = int(np.argmax(u >= np.float32(lo))), int(65535 - np.argmax(u[::-1] <= np.float32(hi)))
    special = np.array([c_lo - 1, c_lo, c_lo + 1, c_hi - 1, c_hi, c_hi + 1, 0, 65535, step, step - 1, step + 1, 65535 - step,
                        (n_points // 2) * step, (n_points // 2) * step - 1], dtype=np.uint16)
    codes.reshape(-1)[:special.size] = special
    codes[3].reshape(-1)[:special.size] = special[::-1]
    x = oc.normalize_codes(codes)
    t = torch.tensor(exposures, dtype=torch.float64)
    i, j, r = get_valid_exposure_pairs(t, 0.25)
    pairs = ops.PairList(i, j, r, n, dev)
    lut = torch.stack([torch.linspace(0, 1, n_points) ** p for p in (1.8, 2.2, 2.6)]).to(dev)
    for rel in (True, False):
        kw = dict(lut=lut, interp="linear", lower=lo, upper=hi, use_relative=rel, use_unc_weight=False)
        s_code = ops.pair_residual_sums(torch.from_numpy(codes).to(dev), pairs, level=1, **kw)
        s_float = ops.pair_residual_sums(torch.from_numpy(x).to(dev), pairs, level=1, **kw)
        assert torch.equal(s_code[..., 4], s_float[..., 4]), "mask popcounts differ"
        assert_parity(s_code[..., :3].cpu().numpy(), s_float[..., :3].cpu().numpy(), rtol=1e-5, norm_tol=2e-6,
                      what=f"code-domain vs generic sums L={n_points} {'rel' if rel else 'abs'}")
        coef = torch.from_numpy(rng.uniform(0.5, 1.5, size=(pairs.n_pairs, 3))).to(dev)  # one sign: LUT bins do not cancel
        g_code = ops.pair_residual_lut_grad(torch.from_numpy(codes).to(dev), pairs, coef, **kw)
        g_float = ops.pair_residual_lut_grad(torch.from_numpy(x).to(dev), pairs, coef, **kw)
        # most LUT bins of so small an image are empty or hold a handful of +- terms: norm-wise, and element-wise
        # against the largest entry (the median the usual criterion scales by is ~0 here)
        gc, gf = g_code.cpu().numpy(), g_float.cpu().numpy()
        # (a residual within 1e-7 of zero changes sign between the two stagings and moves its +-weight: a few in 1e5 samples)
        assert rel_norm(gc, gf) <= 2e-5, (n_points, rel, rel_norm(gc, gf))
        assert np.abs(gc - gf).max() <= 1e-4 * np.abs(gf).max(), (n_points, rel, np.abs(gc - gf).max() / np.abs(gf).max())


@pytest.mark.parametrize("mode", ["linear", "catmull"])
@pytest.mark.parametrize("rel", [True, False])
def test_linearity_loss_and_lut_gradient(dev, mode, rel):
    """Per-channel linearity loss and its (C,L) LUT gradient: what loss[c].backward() deposits for that term."""
    from clair_torch_amd.training import linearity_loss
    g = golden("training")
    stack, _ = _inputs(g, dev)
    pairs, _ = _pairs(g, dev, 0.25)
    lut = torch.from_numpy(g["train_lut0"]).to(dev).requires_grad_(True)
    lin, spatial = linearity_loss(lut, stack, pairs, interp=mode, lower=1 / 255, upper=254 / 255, use_relative=rel,
                                  use_unc_weight=False)
    key = f"train_none_{mode}_{'rel' if rel else 'abs'}_nounc"
    assert lin.dtype == torch.float64
    assert_parity(lin.detach().cpu().numpy(), g[key + "_linloss"], rtol=1e-5, norm_tol=2e-6, what=key + " linloss")
    assert_parity(spatial.cpu().numpy(), g[key + "_spatial"], rtol=1e-5, norm_tol=2e-6, what=key + " spatial")
    grad = torch.autograd.grad(lin.sum(), lut)[0]
    assert grad.shape == lut.shape and grad.dtype == torch.float32
    # the reference accumulates this gradient in float32 through index_put (thread-order noise up to 2.5e-5 between two
    # runs of the reference itself on the arrays' smallest entries); observed against it here: 5.1e-7 element-wise
    assert_parity(grad.cpu().numpy(), g[key + "_lingrad"], norm_tol=2e-6, elem_tol=1e-5, what=key + " lingrad")
    # with uncertainty images but without uncertainty weighting the loss is unchanged (losses.py:93-100)
    lin2, _ = linearity_loss(lut, stack, pairs, interp=mode, lower=1 / 255, upper=254 / 255, use_relative=rel,
                             use_unc_weight=False, std_mode="multiplier", std_value=0.05)
    assert torch.equal(lin2.detach(), lin.detach())


@pytest.mark.parametrize("mode", ["linear", "catmull"])
@pytest.mark.parametrize("rel", [True, False])
def test_uncertainty_weighted_loss_and_gradient(dev, mode, rel):
    """use_uncertainty_weighting=True with uncertainty images (train_icrf's default): the weights 1/(err+1e-6)
    depend on the LUT for the relative loss; loss and LUT gradient against the reference's autograd."""
    from clair_torch_amd.training import linearity_loss
    g = golden("training")
    stack, _ = _inputs(g, dev)
    pairs, _ = _pairs(g, dev, 0.25)
    lut = torch.from_numpy(g["train_lut0"]).to(dev).requires_grad_(True)
    lin, spatial = linearity_loss(lut, stack, pairs, interp=mode, lower=1 / 255, upper=254 / 255, use_relative=rel,
                                  use_unc_weight=True, std_mode="multiplier", std_value=0.05)
    key = f"train_multiplier_{mode}_{'rel' if rel else 'abs'}_unc"
    assert_parity(lin.detach().cpu().numpy(), g[key + "_linloss"], rtol=1e-5, norm_tol=2e-6, what=key + " linloss")
    assert_parity(spatial.cpu().numpy(), g[key + "_spatial"], rtol=1e-5, norm_tol=2e-6, what=key + " spatial")
    grad = torch.autograd.grad(lin.sum(), lut)[0]
    assert_parity(grad.cpu().numpy(), g[key + "_lingrad"], norm_tol=2e-6, elem_tol=1e-5, what=key + " lingrad")


def test_train_icrf_uncertainty_weighted_run(dev):
    """Five epochs with use_uncertainty_weighting=True and MULTIPLIER uncertainties vs the reference's run."""
    from clair_torch_amd.common.enums import InterpMode, MissingStdMode
    from clair_torch_amd.datasets import StackDataset, custom_collate
    from clair_torch_amd.models import ICRFModelDirect
    from clair_torch_amd.training import train_icrf
    from oracle import ct_oracle as oc
    g = golden("training")
    x = torch.from_numpy(oc.normalize_codes(g["train_codes"]))
    ds = StackDataset(x, g["train_exposures"].tolist(), missing_std_mode=MissingStdMode.MULTIPLIER, missing_std_value=0.05)
    loader = DataLoader(ds, batch_size=x.shape[0], shuffle=False, collate_fn=custom_collate)
    model = ICRFModelDirect(n_points=256, channels=3, interpolation_mode=InterpMode.LINEAR, initial_power=2.5).to(dev)
    opts = [torch.optim.Adam(model.channel_params(c), lr=1e-3, amsgrad=False) for c in range(3)]
    train_icrf(loader, x.shape[0], "cuda", model, optimizers=opts, use_relative_linearity_loss=True,
               use_uncertainty_weighting=True, epochs=5, patience=200, alpha=10.0, exposure_ratio_threshold=0.25,
               verbose=False)
    ref = g["trainloop_multiplier_unc_icrf"]
    assert np.max(np.abs(model.icrf.detach().cpu().numpy() - ref)) < 5e-6


@pytest.mark.parametrize("h,w", [(61, 47), (60, 46), (64, 128)])
@pytest.mark.parametrize("kind", ["float", "u16", "u8"])
def test_lut_gradient_vs_eager_oracle_large(dev, h, w, kind):
    """Bigger, ragged cases (9 exposures, 64-sample LUT) against the eager float64-residual oracle, over the staging
    variants of the kernels: an odd plane (61x47: scalar staging), a 4-aligned plane that is not a multiple of the tile
    (60x46: vectorised staging with a partial last tile), whole tiles (64x128); float pixels, uint16 and uint8 codes.
    Also checks the tile decomposition: two row bands sum to the whole."""
    from clair_torch_amd import ops
    from clair_torch_amd.common.general_functions import get_valid_exposure_pairs
    from clair_torch_amd.training import linearity_loss
    from oracle import ct_oracle as oc
    from oracle import eager_torch as oe
    gen = torch.Generator().manual_seed(21)
    n, c = 9, 3
    t = torch.tensor([0.001 * 2.0 ** (k / 2.0) for k in range(n)], dtype=torch.float64)
    e = torch.rand((c, h, w), generator=gen, dtype=torch.float64) * (2.0 / float(torch.sqrt(t[0] * t[-1])))
    x = ((e.unsqueeze(0) * t.view(-1, 1, 1, 1)).clamp(0, 1) ** (1 / 2.2)).float()
    x = (x + 0.01 * torch.randn(x.shape, generator=gen)).clamp(0, 1)
    max_code = None
    if kind != "float":
        max_code = 65535 if kind == "u16" else 255
        codes = torch.round(x * max_code).to(torch.int32).numpy().astype(np.uint16 if kind == "u16" else np.uint8)
        x = torch.from_numpy(oc.normalize_codes(codes))   # what the reference sees after CastTo + Normalize
        dev_x = torch.from_numpy(codes).to(dev)
    else:
        dev_x = x.to(dev)
    lut0 = torch.stack([torch.linspace(0, 1, 64) ** p for p in (1.9, 2.2, 2.5)])
    # comparand: the eager chain with the LUT gradient scattered in float64 (deterministic; the reference's own float32
    # index_put accumulation depends on the CPU thread count)
    lin_o, sp_o, grad_o = oe.linearity_lut_grad_f64(x, None, t, lut0, "linear", 0.25, 1 / 255, 254 / 255, True, False)
    i, j, r = get_valid_exposure_pairs(t, 0.25)
    pairs = ops.PairList(i, j, r, n, dev)
    lut = lut0.to(dev).requires_grad_(True)
    lin, sp = linearity_loss(lut, dev_x, pairs, interp="linear", lower=1 / 255, upper=254 / 255, use_relative=True,
                             use_unc_weight=False, max_code=max_code)
    grad = torch.autograd.grad(lin.sum(), lut)[0]
    assert_parity(sp.cpu().numpy(), sp_o.detach().numpy(), rtol=1e-5, norm_tol=2e-6, what="spatial")
    assert_parity(lin.detach().cpu().numpy(), lin_o.detach().numpy(), rtol=1e-5, norm_tol=2e-6, what="lin loss")
    assert_parity(grad.cpu().numpy(), grad_o.numpy(), norm_tol=2e-6, elem_tol=1e-5, what="lut grad")
    # tiles: sums and LUT gradients are additive over row bands when the global geometry is passed
    kw = dict(lut=lut.detach(), interp="linear", lower=1 / 255, upper=254 / 255, use_relative=True, use_unc_weight=False,
              max_code=max_code)
    whole = ops.pair_residual_sums(dev_x, pairs, level=1, **kw)
    bands = ((0, 20), (20, h))
    parts = [ops.pair_residual_sums(dev_x[:, :, r0:r1].contiguous(), pairs, level=1,
                                    tile=ops.TileGeometry(h_global=h, row_offset=r0), **kw) for r0, r1 in bands]
    assert_parity((parts[0] + parts[1]).cpu().numpy(), whole.cpu().numpy(), rtol=1e-6, norm_tol=1e-6, what="tiles")
    coef = torch.rand((pairs.n_pairs, c), generator=gen, dtype=torch.float64).to(dev) * 1e-4
    g_whole = ops.pair_residual_lut_grad(dev_x, pairs, coef, **kw)
    g_parts = [ops.pair_residual_lut_grad(dev_x[:, :, r0:r1].contiguous(), pairs, coef,
                                          tile=ops.TileGeometry(h_global=h, row_offset=r0), **kw) for r0, r1 in bands]
    assert_parity((g_parts[0] + g_parts[1]).cpu().numpy(), g_whole.cpu().numpy(), rtol=1e-6, norm_tol=1e-6,
                  what="gradient tiles")


@pytest.mark.parametrize("max_code,n_points", [(4095, 256), (1023, 256), (16383, 64), (65535, 100), (4095, 33)])
def test_training_kernels_on_codes_below_full_range(dev, max_code, n_points):
    """10- / 12- / 14-bit data held in uint16 (Normalize(1023 / 4095 / 16383)) and LUT lengths whose step is not a whole number
    of codes: raw codes with ``max_code`` against the same data normalised with the reference's float32 division and fed
    as float32 pixels.  A fifth of the codes lie ABOVE max_code (the model clamps them to the top of the curve and the
    validity mask drops them), codes sit on the thresholds and on LUT knots; both backward kernels."""
    from clair_torch_amd import ops
    from clair_torch_amd.common.general_functions import get_valid_exposure_pairs
    rng = np.random.default_rng(max_code + n_points)
    n, shape = 9, (3, 24, 64)
    codes = rng.integers(0, max_code + 1, size=(n,) + shape).astype(np.uint16)
    flat = codes.reshape(-1)
    if max_code < 65535:
        flat[::5] = rng.integers(max_code + 1, min(65536, 2 * max_code + 2), size=flat[::5].shape).astype(np.uint16)
    lo, hi = 1 / 255, 254 / 255
    u = np.arange(max_code + 1, dtype=np.float32) / np.float32(max_code)
    c_lo, c_hi = int(np.argmax(u >= np.float32(lo))), int(max_code - np.argmax(u[::-1] <= np.float32(hi)))
    knots = np.ceil(np.arange(1, 6) * max_code / (n_points - 1)).astype(np.int64)
    special = np.concatenate([[c_lo - 1, c_lo, c_lo + 1, c_hi - 1, c_hi, c_hi + 1, 0, max_code, min(max_code + 1, 65535)],
                              knots, knots - 1]).astype(np.uint16)
    flat[1:1 + special.size] = special
    x = (codes.astype(np.float32) / np.float32(max_code)).astype(np.float32)  # CastTo + Normalize as the reference computes them
    t = torch.tensor(0.002 * 1.3 ** np.arange(n), dtype=torch.float64)
    i, j, r = get_valid_exposure_pairs(t, 0.25)
    pairs = ops.PairList(i, j, r, n, dev)
    lut = torch.stack([torch.linspace(0, 1, n_points) ** p for p in (1.8, 2.2, 2.6)]).to(dev)
    for rel in (True, False):
        kw = dict(lut=lut, interp="linear", lower=lo, upper=hi, use_relative=rel, use_unc_weight=False)
        s_code = ops.pair_residual_sums(torch.from_numpy(codes).to(dev), pairs, level=1, max_code=float(max_code), **kw)
        s_float = ops.pair_residual_sums(torch.from_numpy(x).to(dev), pairs, level=1, **kw)
        assert torch.equal(s_code[..., 4], s_float[..., 4]), "mask popcounts differ"
        assert float(s_code[..., 4].sum()) > 0
        assert_parity(s_code[..., :3].cpu().numpy(), s_float[..., :3].cpu().numpy(), rtol=1e-5, norm_tol=2e-6,
                      what=f"code-domain vs generic sums max {max_code} L={n_points} {'rel' if rel else 'abs'}")
        coef = torch.from_numpy(rng.uniform(0.5, 1.5, size=(pairs.n_pairs, 3))).to(dev)
        for lane in (True, False):
            g_code = ops.pair_residual_lut_grad(torch.from_numpy(codes).to(dev), pairs, coef, max_code=float(max_code),
                                                lane_kernel=lane, **kw)
            g_float = ops.pair_residual_lut_grad(torch.from_numpy(x).to(dev), pairs, coef, lane_kernel=lane, **kw)
            gc, gf = g_code.cpu().numpy(), g_float.cpu().numpy()
            assert rel_norm(gc, gf) <= 2e-5, (max_code, n_points, rel, rel_norm(gc, gf))
            assert np.abs(gc - gf).max() <= 1e-4 * np.abs(gf).max(), (max_code, n_points, rel)


@pytest.mark.parametrize("mode", ["linear", "lookup_fwd", "catmull"])
def test_many_exposures_narrow_tiles(dev, mode):
    """128 exposures: the kernels fall back to 32-column tiles (LDS budget) and walk the pair list in several launches;
    every pair passes the threshold-free list (8128 pairs).  Sums and LUT gradient against the eager oracle."""
    from clair_torch_amd import ops
    from clair_torch_amd.common.general_functions import get_valid_exposure_pairs
    from clair_torch_amd.training import linearity_loss
    from oracle import eager_torch as oe
    gen = torch.Generator().manual_seed(5)
    n, c, h, w = 128, 3, 12, 20
    t = torch.tensor([0.001 * 2.0 ** (k / 16.0) for k in range(n)], dtype=torch.float64)
    e = torch.rand((c, h, w), generator=gen, dtype=torch.float64) * (2.0 / float(torch.sqrt(t[0] * t[-1])))
    x = ((e.unsqueeze(0) * t.view(-1, 1, 1, 1)).clamp(0, 1) ** (1 / 2.2)).float()
    x = (x + 0.01 * torch.randn(x.shape, generator=gen)).clamp(0, 1)
    lut0 = torch.stack([torch.linspace(0, 1, 32) ** p for p in (1.9, 2.2, 2.5)])
    i, j, r = get_valid_exposure_pairs(t, None)
    assert i.numel() == n * (n - 1) // 2
    pairs = ops.PairList(i, j, r, n, dev)
    dev_x = x.to(dev)
    if mode == "lookup_fwd":   # LOOKUP has no LUT gradient in the reference either: forward statistics only
        _, sp_o, _, _ = oe.linearity_statistics(x, None, t, lut0, "lookup", None, 1 / 255, 254 / 255, True, False)
        sums = ops.pair_residual_sums(dev_x, pairs, lut=lut0.to(dev), interp="lookup", lower=1 / 255, upper=254 / 255,
                                      use_relative=True, use_unc_weight=False)
        sp = sums[..., 1] / sums[..., 0].clamp(min=1e-8)
        assert_parity(sp.cpu().numpy(), sp_o.numpy(), rtol=1e-5, norm_tol=2e-6, what="spatial (lookup)")
        return
    lo = lut0.clone().requires_grad_(True)
    _, lin_o, sp_o = oe.training_loss(x, None, t, lo, mode, None, 1 / 255, 254 / 255, True, False)
    grad_o = torch.autograd.grad(lin_o.sum(), lo)[0]
    lut = lut0.to(dev).requires_grad_(True)
    lin, sp = linearity_loss(lut, dev_x, pairs, interp=mode, lower=1 / 255, upper=254 / 255, use_relative=True,
                             use_unc_weight=False)
    grad = torch.autograd.grad(lin.sum(), lut)[0]
    assert_parity(sp.cpu().numpy(), sp_o.detach().numpy(), rtol=1e-5, norm_tol=2e-6, what="spatial")
    assert_parity(grad.cpu().numpy(), grad_o.numpy(), norm_tol=2e-6, elem_tol=1e-5, what="lut grad")


@pytest.mark.parametrize("sname", ["none", "multiplier"])
@pytest.mark.parametrize("mname", ["nomodel", "linear"])
def test_measure_linearity_api(dev, sname, mname):
    from clair_torch_amd.common.enums import InterpMode, MissingStdMode
    from clair_torch_amd.common.transforms import CastTo, Normalize
    from clair_torch_amd.datasets import StackDataset, custom_collate
    from clair_torch_amd.inference import measure_linearity
    from clair_torch_amd.models import ICRFModelDirect
    g = golden("training")
    codes = torch.from_numpy(g["train_codes"])
    mode = MissingStdMode.MULTIPLIER if sname == "multiplier" else MissingStdMode.NONE
    ds = StackDataset(codes, g["train_exposures"].tolist(), missing_std_mode=mode, missing_std_value=0.05,
                      materialize_std=False)
    loader = DataLoader(ds, batch_size=codes.shape[0], shuffle=False, collate_fn=custom_collate)
    model = None if mname == "nomodel" else ICRFModelDirect(icrf=torch.from_numpy(g["train_lut0"]),
                                                            interpolation_mode=InterpMode.LINEAR).to(dev)
    for rel in (True, False):
        for unc in (True, False):
            ratio, mean, sd, err = measure_linearity(loader, "cuda", unc, rel, model,
                                                     gpu_transforms=[CastTo("float32"), Normalize(255, 0)])
            key = f"meas_{sname}_{mname}_{'rel' if rel else 'abs'}_{'unc' if unc else 'nounc'}"
            assert np.array_equal(ratio.cpu().numpy(), g[key + "_ratio"])
            assert_parity(mean.cpu().numpy(), g[key + "_spatial"], rtol=1e-5, norm_tol=2e-6, what=key)
            assert_parity(sd.cpu().numpy(), g[key + "_spatial_std"], rtol=1e-5, norm_tol=2e-6, what=key + " std")
            if sname == "none":
                assert err is None
            else:
                assert_parity(err.cpu().numpy(), g[key + "_spatial_err"], rtol=1e-5, norm_tol=2e-6, what=key + " err")


@pytest.mark.parametrize("sname", ["none", "multiplier"])
def test_train_icrf_matches_reference_run(dev, sname):
    """Five epochs of train_icrf (per-channel Adam, lr 1e-3, alpha 10) reproduce the curve the reference reached,
    including its dead first step."""
    from clair_torch_amd.common.enums import InterpMode, MissingStdMode
    from clair_torch_amd.datasets import StackDataset, custom_collate
    from clair_torch_amd.models import ICRFModelDirect
    from clair_torch_amd.training import train_icrf
    from oracle import ct_oracle as oc
    g = golden("training")
    x = torch.from_numpy(oc.normalize_codes(g["train_codes"]))
    mode = MissingStdMode.MULTIPLIER if sname == "multiplier" else MissingStdMode.NONE
    ds = StackDataset(x, g["train_exposures"].tolist(), missing_std_mode=mode, missing_std_value=0.05)
    loader = DataLoader(ds, batch_size=x.shape[0], shuffle=False, collate_fn=custom_collate)
    model = ICRFModelDirect(n_points=256, channels=3, interpolation_mode=InterpMode.LINEAR, initial_power=2.5).to(dev)
    opts = [torch.optim.Adam(model.channel_params(c), lr=1e-3, amsgrad=False) for c in range(3)]
    out = train_icrf(loader, x.shape[0], "cuda", model, optimizers=opts, schedulers=None,
                     use_relative_linearity_loss=True, use_uncertainty_weighting=False, epochs=5, patience=200,
                     alpha=10.0, beta=1.0, gamma=1.0, delta=1.0, lower_valid_threshold=1 / 255,
                     upper_valid_threshold=254 / 255, exposure_ratio_threshold=0.25, verbose=False)
    assert out is model
    ref = g[f"trainloop_{sname}_nounc_icrf"]
    got = model.icrf.detach().cpu().numpy()
    # Adam's first steps move every bin by ~lr regardless of the gradient's size, so agreement to 2e-6 absolute
    # (0.2 % of one step) means every bin's gradient sign and relative size matched the reference
    assert np.max(np.abs(got - ref)) < 2e-6, np.max(np.abs(got - ref))
    assert sorted(model.state_dict().keys()) == sorted(
        ["_x_axis_datapoints", "_icrf", "direct_params.0", "direct_params.1", "direct_params.2"])


def test_train_icrf_deferred_bookkeeping_equals_the_blocking_loop(dev, capsys):
    """train_icrf settles an epoch's bookkeeping (loss read, early stopping, schedulers, messages) inside the NEXT epoch, behind
    that epoch's queued forward / backward.  Against a loop written in the reference's blocking order
    (icrf_training.py:96-186: read the loss, count patience, step the schedulers, then start the next epoch) it must give
    the same curve, stop at the same epoch and print the same messages -- with a scheduler that halves the rate every
    time the loss fails to improve and a patience that triggers early stopping."""
    from clair_torch_amd import ops
    from clair_torch_amd.common.enums import InterpMode
    from clair_torch_amd.common.general_functions import get_valid_exposure_pairs
    from clair_torch_amd.datasets import StackDataset, custom_collate
    from clair_torch_amd.models import ICRFModelDirect
    from clair_torch_amd.training import linearity_loss, train_icrf
    from clair_torch_amd.training.losses import (compute_endpoint_penalty, compute_monotonicity_penalty,
                                                 compute_range_penalty, compute_smoothness_penalty)
    from oracle import ct_oracle as oc
    g = golden("training")
    x = torch.from_numpy(oc.normalize_codes(g["train_codes"]))
    t = g["train_exposures"].tolist()
    n = x.shape[0]
    epochs, patience, lr = 40, 3, 0.02  # a rate at which the loss soon stops improving from epoch to epoch

    def make():
        model = ICRFModelDirect(n_points=64, channels=3, interpolation_mode=InterpMode.LINEAR, initial_power=2.5).to(dev)
        opts = [torch.optim.Adam(model.channel_params(c), lr=lr) for c in range(3)]
        scheds = [torch.optim.lr_scheduler.ReduceLROnPlateau(o, mode="min", factor=0.5, patience=0) for o in opts]
        return model, opts, scheds

    model, opts, scheds = make()
    loader = DataLoader(StackDataset(x, t), batch_size=n, shuffle=False, collate_fn=custom_collate)
    train_icrf(loader, n, "cuda", model, optimizers=opts, schedulers=scheds, epochs=epochs, patience=patience, alpha=10.0,
               exposure_ratio_threshold=0.25, use_uncertainty_weighting=False, verbose=True)
    printed = [l for l in capsys.readouterr().out.splitlines() if l.startswith(("Epoch", "Early", "Optimizer"))]

    # the reference's order, blocking
    ref, r_opts, r_scheds = make()
    ref.train()
    images = x.to(dev)
    i, j, r = get_valid_exposure_pairs(torch.tensor(t, dtype=torch.float64), 0.25)
    pairs = ops.PairList(i, j, r, n, dev)
    best, stale, lrs, lines = [float("inf")] * 3, [0] * 3, [lr] * 3, []
    for epoch in range(epochs):
        for o in r_opts:
            o.zero_grad()
        curve = ref.icrf
        lin, _ = linearity_loss(curve, images, pairs, interp="linear", lower=1 / 255, upper=254 / 255, use_relative=True,
                                use_unc_weight=False)
        loss = (lin + 10.0 * compute_monotonicity_penalty(curve, per_channel=True) + compute_range_penalty(curve, per_channel=True)
                + compute_endpoint_penalty(curve, per_channel=True) + compute_smoothness_penalty(curve, per_channel=True))
        if loss.requires_grad:
            loss.sum().backward()
        for o in r_opts:
            o.step()
        ref.update_icrf()
        avg = loss.detach().cpu().numpy()
        lines.append(f"Epoch {epoch + 1} Loss: {avg}")
        for c in range(3):
            if avg[c] < best[c]:
                best[c], stale[c] = avg[c], 0
            else:
                stale[c] += 1
        if all(s >= patience for s in stale):
            lines.append(f"Early stopping triggered for all channels (patience = {patience} epochs).")
            break
        for c, sch in enumerate(r_scheds):
            sch.step(avg[c])
        for k, o in enumerate(r_opts):
            cur = o.param_groups[0]["lr"]
            if cur != lrs[k]:
                lines.append(f"Optimizer {k} learning rate changed to: {cur}")
            lrs[k] = cur
    assert any(l.startswith("Early") for l in lines) and any(l.startswith("Optimizer") for l in lines), "the recipe must exercise both"
    assert len(printed) == len(lines)
    for a, b in zip(printed, lines):
        if a.startswith("Epoch"):  # the loss values agree to the float64 atomics' order
            va = np.array(a.split("Loss:")[1].strip(" []").split(), dtype=np.float64)
            vb = np.array(b.split("Loss:")[1].strip(" []").split(), dtype=np.float64)
            assert a.split("Loss:")[0] == b.split("Loss:")[0] and np.allclose(va, vb, rtol=1e-6, atol=0)
        else:
            assert a == b
    assert_parity(model.icrf.detach().cpu().numpy(), ref.icrf.detach().cpu().numpy(), rtol=1e-8, norm_tol=1e-9, what="curve")
    for o, q in zip(opts, r_opts):
        assert o.param_groups[0]["lr"] == q.param_groups[0]["lr"]


def test_train_icrf_argument_errors(dev):
    from clair_torch_amd.datasets import StackDataset, custom_collate
    from clair_torch_amd.models import ICRFModelDirect
    from clair_torch_amd.training import train_icrf
    ds = StackDataset(torch.rand(4, 3, 8, 8), [1, 2, 3, 4])
    loader = DataLoader(ds, batch_size=4, collate_fn=custom_collate)
    model = ICRFModelDirect().to(dev)
    with pytest.raises(ValueError, match="larger than 1"):
        train_icrf(loader, 1, "cuda", model)
    opts = [torch.optim.Adam(model.channel_params(c)) for c in range(3)]
    with pytest.raises(ValueError, match="Mismatched number"):
        train_icrf(loader, 4, "cuda", model, optimizers=opts, schedulers=[None])
    with pytest.raises(RuntimeError, match="MI355X"):
        train_icrf(loader, 4, "cpu", model)


def test_config_c3_full_shape(dev):
    """BASELINE config C3 at its real shape: 64 exposures x 2048 x 2048 x 3 uint16, threshold 0.25 -> 888 pairs.
    Exercises what the small fixtures cannot: the multi-round persistent grid, the float64 accumulation across 4 M
    pixels per channel, the channel skip of the workgroup order.
    (1) the five sums and the LUT gradient of two ragged row bands add up to the whole image's;
    (2) a full-width 8-row band cut out of the real stack (global geometry) equals the eager float64-residual oracle:
        spatial means, per-channel linearity loss, LUT gradient.  The band starts at a row that is a multiple of 3 and
        has 8 = 2048 (mod 3) rows, so the reference's LUT-row quirk (flat index % C) selects the same rows for the band
        alone (what the eager oracle sees) as for the band inside the full image (what the kernel is told)."""
    from clair_torch_amd import ops
    from clair_torch_amd.common.general_functions import get_valid_exposure_pairs
    from clair_torch_amd.datasets import synthetic_exposure_stack
    from clair_torch_amd.training import linearity_loss
    from oracle import ct_oracle as oc
    from oracle import eager_torch as oe
    n, c, h, w = 64, 3, 2048, 2048
    codes, exposures = synthetic_exposure_stack(n, c, h, w, bits=16, stops_per_step=0.125, seed=1237, device=dev)
    t = torch.tensor(exposures, dtype=torch.float64)
    i, j, r = get_valid_exposure_pairs(t, 0.25)
    assert i.numel() == 888
    pairs = ops.PairList(i, j, r, n, dev)
    lut0 = torch.stack([torch.linspace(0, 1, 256) ** p for p in (2.3, 2.5, 2.7)])
    kw = dict(lut=lut0.to(dev), interp="linear", lower=1 / 255, upper=254 / 255, use_relative=True, use_unc_weight=False)
    whole = ops.pair_residual_sums(codes, pairs, level=1, **kw)
    assert torch.isfinite(whole).all() and float(whole[..., 4].min()) > 0
    gen = torch.Generator().manual_seed(3)
    coef = (torch.rand((pairs.n_pairs, c), generator=gen, dtype=torch.float64) * 1e-6).to(dev)
    g_whole = ops.pair_residual_lut_grad(codes, pairs, coef, **kw)
    # C3's list is a band (16 of 64 samples, 888 of 1024 lane-steps filled): g_whole came from the lane <-> sample kernel;
    # the generic pair-walk kernel on the same launch arguments must agree (float32 vs float64 partner accumulation)
    assert pairs.band == 16
    g_generic = ops.pair_residual_lut_grad(codes, pairs, coef, lane_kernel=False, **kw)
    assert_parity(g_whole.cpu().numpy(), g_generic.cpu().numpy(), norm_tol=1e-7, elem_tol=1e-5, what="C3 LUT gradient: lane kernel = generic kernel")
    bands = ((0, 701), (701, h))
    parts = [ops.pair_residual_sums(codes[:, :, r0:r1].contiguous(), pairs, level=1,
                                    tile=ops.TileGeometry(h_global=h, row_offset=r0), **kw) for r0, r1 in bands]
    assert_parity((parts[0] + parts[1]).cpu().numpy(), whole.cpu().numpy(), rtol=1e-9, norm_tol=1e-12, what="C3 sums: bands = whole")
    g_parts = [ops.pair_residual_lut_grad(codes[:, :, r0:r1].contiguous(), pairs, coef,
                                          tile=ops.TileGeometry(h_global=h, row_offset=r0), **kw) for r0, r1 in bands]
    assert_parity((g_parts[0] + g_parts[1]).cpu().numpy(), g_whole.cpu().numpy(), rtol=1e-9, norm_tol=1e-12,
                  what="C3 LUT gradient: bands = whole")
    r0, rows = 1023, 8
    band = codes[:, :, r0:r0 + rows].contiguous()
    x = torch.from_numpy(oc.normalize_codes(band.cpu().numpy()))
    lo = lut0.clone().requires_grad_(True)
    _, lin_o, sp_o = oe.training_loss(x, None, t, lo, "linear", 0.25, 1 / 255, 254 / 255, True, False)
    grad_o = torch.autograd.grad(lin_o.sum(), lo)[0]
    lut = lut0.to(dev).requires_grad_(True)
    lin, sp = linearity_loss(lut, band, pairs, interp="linear", lower=1 / 255, upper=254 / 255, use_relative=True,
                             use_unc_weight=False, tile=ops.TileGeometry(h_global=h, row_offset=r0), group=False)
    grad = torch.autograd.grad(lin.sum(), lut)[0]
    assert_parity(sp.cpu().numpy(), sp_o.detach().numpy(), rtol=1e-5, norm_tol=2e-6, what="C3 band spatial means")
    assert_parity(lin.detach().cpu().numpy(), lin_o.detach().numpy(), rtol=1e-5, norm_tol=2e-6, what="C3 band linearity loss")
    assert_parity(grad.cpu().numpy(), grad_o.numpy(), norm_tol=2e-6, elem_tol=1e-5, what="C3 band LUT gradient")


@pytest.mark.parametrize("interp", ["linear", "catmull", "lookup"])
@pytest.mark.parametrize("relative", [True, False])
@pytest.mark.parametrize("n,stops,h,w,kind", [(64, 0.125, 24, 64, "u16"), (64, 0.125, 13, 37, "float"),
                                              (57, 0.2, 16, 48, "u8"), (64, 0.15, 10, 128, "u16")])
def test_lane_backward_kernel(dev, interp, relative, n, stops, h, w, kind):
    """The lane <-> sample backward (band pair lists, N <= 64) against the eager float64 oracle and against the generic
    pair-walk kernel: whole-tile / ragged (scalar staging) / uint8 / odd band (stops 0.15 -> band 13, rounded up with an
    all-zero step), every interpolation mode, relative and absolute residual."""
    from clair_torch_amd import ops
    from clair_torch_amd.common.general_functions import get_valid_exposure_pairs
    from oracle import ct_oracle as oc
    from oracle import eager_torch as oe
    gen = torch.Generator().manual_seed(n + h)
    c = 3
    t = torch.tensor([0.001 * 2.0 ** (k * stops) for k in range(n)], dtype=torch.float64)
    e = torch.rand((c, h, w), generator=gen, dtype=torch.float64) * (2.0 / float(torch.sqrt(t[0] * t[-1])))
    x = ((e.unsqueeze(0) * t.view(-1, 1, 1, 1)).clamp(0, 1) ** (1 / 2.2)).float()
    x = (x + 0.01 * torch.randn(x.shape, generator=gen)).clamp(0, 1)
    max_code = None
    if kind != "float":
        max_code = 65535 if kind == "u16" else 255
        codes = torch.round(x * max_code).to(torch.int32).numpy().astype(np.uint16 if kind == "u16" else np.uint8)
        x = torch.from_numpy(oc.normalize_codes(codes))
        dev_x = torch.from_numpy(codes).to(dev)
    else:
        dev_x = x.to(dev)
    i, j, r = get_valid_exposure_pairs(t, 0.25)
    pairs = ops.PairList(i, j, r, n, dev)
    assert pairs.band >= 1 and pairs.n_pairs >= 0.65 * ((pairs.band + 1) // 2 * 2) * 64, "case must be lane-eligible"
    lut0 = torch.stack([torch.linspace(0, 1, 64) ** p for p in (1.9, 2.2, 2.5)])
    coef = (torch.rand((pairs.n_pairs, c), generator=gen, dtype=torch.float64) * 1e-4)
    kw = dict(lut=lut0.to(dev), interp=interp, lower=1 / 255, upper=254 / 255, use_relative=relative,
              use_unc_weight=False, max_code=max_code)
    g_lane = ops.pair_residual_lut_grad(dev_x, pairs, coef.to(dev), **kw)
    g_generic = ops.pair_residual_lut_grad(dev_x, pairs, coef.to(dev), lane_kernel=False, **kw)
    assert_parity(g_lane.cpu().numpy(), g_generic.cpu().numpy(), norm_tol=5e-7, elem_tol=1e-5, what="lane = generic")
    if interp == "lookup":  # no LUT gradient in the reference for LOOKUP (index lookup): kernel-against-kernel only
        return
    # the oracle, through the same autograd route train_icrf takes (linearity_loss -> ct_pair_residual_bwd)
    from clair_torch_amd.training import linearity_loss
    # comparand: oracle/eager_torch.linearity_lut_grad_f64 -- the eager chain with the per-sample tap gradients scattered
    # into the (C, L) bins in float64.  (Round 2 compared with the float32 index_put of eager autograd, whose accumulation
    # order depends on the CPU thread count -- 2.6e-6 norm-wise between 3 and 8 threads -- and had to budget 5e-6 for it.)
    _, sp_o, grad_o = oe.linearity_lut_grad_f64(x, None, t, lut0, interp, 0.25, 1 / 255, 254 / 255, relative, False)
    lut = lut0.to(dev).requires_grad_(True)
    lin, sp = linearity_loss(lut, dev_x, pairs, interp=interp, lower=1 / 255, upper=254 / 255, use_relative=relative,
                             use_unc_weight=False, max_code=max_code)
    grad = torch.autograd.grad(lin.sum(), lut)[0]
    assert_parity(sp.cpu().numpy(), sp_o.detach().numpy(), rtol=1e-5, norm_tol=2e-6, what="lane: spatial means")
    assert_parity(grad.cpu().numpy(), grad_o.numpy(), norm_tol=2e-6, elem_tol=2e-5, what="lane: LUT gradient vs the float64-scatter oracle")


def test_lane_backward_broken_promise_falls_back(dev):
    """A band hint smaller than the list's real band (or a list with a pair repeated) must not change the result: the
    entries kernel detects it and the generic kernel does the work."""
    from clair_torch_amd import ops
    from clair_torch_amd.common.general_functions import get_valid_exposure_pairs
    gen = torch.Generator().manual_seed(9)
    n, c, h, w = 64, 3, 8, 64
    t = torch.tensor([0.001 * 2.0 ** (k * 0.125) for k in range(n)], dtype=torch.float64)
    x = torch.rand((n, c, h, w), generator=gen).to(dev)
    i, j, r = get_valid_exposure_pairs(t, 0.25)
    pairs = ops.PairList(i, j, r, n, dev)
    lut = torch.stack([torch.linspace(0, 1, 64) ** p for p in (1.9, 2.2, 2.5)]).to(dev)
    coef = (torch.rand((pairs.n_pairs, c), generator=gen, dtype=torch.float64) * 1e-4).to(dev)
    kw = dict(lut=lut, interp="linear", lower=1 / 255, upper=254 / 255, use_relative=True, use_unc_weight=False)
    good = ops.pair_residual_lut_grad(x, pairs, coef, lane_kernel=False, **kw)
    pairs.band = 14  # a lie: the list's band is 16
    lied = ops.pair_residual_lut_grad(x, pairs, coef, **kw)
    # the generic kernel's float64 atomics land in a run-dependent order (~1e-16); the lane kernel differs from it by
    # ~1e-9 (float32 partner sums), so 1e-12 also proves WHICH kernel produced the result
    assert_parity(lied.cpu().numpy(), good.cpu().numpy(), rtol=1e-9, norm_tol=1e-12, what="broken band promise -> generic kernel")
    pairs.band = 16
    lane = ops.pair_residual_lut_grad(x, pairs, coef, **kw)
    assert not torch.equal(lane, good) and float((lane - good).norm() / good.norm()) < 1e-6


def test_lane_backward_narrow_tiles(dev):
    """A 512-entry LUT leaves room for 32-column tiles only beside the lane kernel's constants (two workgroups per CU):
    the narrow-tile route of the lane kernel against the generic kernel and the eager oracle."""
    from clair_torch_amd import ops
    from clair_torch_amd.common.general_functions import get_valid_exposure_pairs
    from clair_torch_amd.training import linearity_loss
    from oracle import eager_torch as oe
    gen = torch.Generator().manual_seed(77)
    n, c, h, w = 64, 3, 9, 52
    t = torch.tensor([0.001 * 2.0 ** (k * 0.125) for k in range(n)], dtype=torch.float64)
    e = torch.rand((c, h, w), generator=gen, dtype=torch.float64) * (2.0 / float(torch.sqrt(t[0] * t[-1])))
    x = ((e.unsqueeze(0) * t.view(-1, 1, 1, 1)).clamp(0, 1) ** (1 / 2.2)).float()
    x = (x + 0.01 * torch.randn(x.shape, generator=gen)).clamp(0, 1)
    i, j, r = get_valid_exposure_pairs(t, 0.25)
    pairs = ops.PairList(i, j, r, n, dev)
    lut0 = torch.stack([torch.linspace(0, 1, 512) ** p for p in (1.9, 2.2, 2.5)])
    coef = (torch.rand((pairs.n_pairs, c), generator=gen, dtype=torch.float64) * 1e-4).to(dev)
    kw = dict(lut=lut0.to(dev), interp="linear", lower=1 / 255, upper=254 / 255, use_relative=True, use_unc_weight=False)
    g_lane = ops.pair_residual_lut_grad(x.to(dev), pairs, coef, **kw)
    g_generic = ops.pair_residual_lut_grad(x.to(dev), pairs, coef, lane_kernel=False, **kw)
    assert not torch.equal(g_lane, g_generic)  # two different kernels ...
    assert_parity(g_lane.cpu().numpy(), g_generic.cpu().numpy(), norm_tol=5e-7, elem_tol=1e-5, what="narrow tiles: lane = generic")
    _, _, grad_o = oe.linearity_lut_grad_f64(x, None, t, lut0, "linear", 0.25, 1 / 255, 254 / 255, True, False)
    lut = lut0.to(dev).requires_grad_(True)
    lin, _ = linearity_loss(lut, x.to(dev), pairs, interp="linear", lower=1 / 255, upper=254 / 255, use_relative=True,
                            use_unc_weight=False)
    grad = torch.autograd.grad(lin.sum(), lut)[0]
    assert_parity(grad.cpu().numpy(), grad_o.numpy(), norm_tol=2e-6, elem_tol=2e-5, what="narrow tiles: LUT gradient vs oracle")
