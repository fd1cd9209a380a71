"""GPU parity tests of the HDR merge (ct_hdr_merge_batch through clair_torch_amd.ops / compute_hdr_image).

Chain of trust: reference --(golden vectors)--> oracle --(same seeded inputs)--> HIP kernels.  The kernels are
compared both directly with the golden vectors recorded from the reference and with the CPU oracle on larger /
ragged inputs.  Tolerances: SURVEY 8(d) -- mean rtol 1e-5; std norm-wise <= 1e-5 and element-wise
allclose(rtol, atol = rtol*median) where rtol is 1e-5 against the float64 closed-form oracle and the per-mode value
of the reference's own float32 autograd noise (tests/test_oracle_golden.py ELEM_TOL) against the golden vectors.
"""
import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

from _util import PARTITIONS, assert_parity, golden, sampler_batches, std_for

pytestmark = pytest.mark.gpu

# element-wise against the GOLDEN vectors (the reference's own recorded outputs).  CATMULL with uncertainties runs the
# reference-order kernel (ct_merge_exact.hip), which equals the float32-order emulation of the reference's backward bit for
# bit; what is left against the recorded vectors is the last bit of torch's CPU exp (Sleef, 1 ULP: 1.1 % of its results are
# not the correctly rounded ones), which the reference's own CATMULL chain amplifies to 1.07e-5 on the uint16 fixtures
# (tests/test_oracle_golden.py::test_reference_uncertainty_depends_on_the_last_bit_of_exp) -- hence 1.1e-5 there.
ELEM_TOL = {"linear": 1e-5, "nomodel": 1e-5, "lookup": 1e-5, "catmull": 1.1e-5}
NORM_TOL = {"linear": 1e-5, "nomodel": 1e-5, "lookup": 1e-5, "catmull": 1e-5}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from clair_torch_amd import _native
    _native.load()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def _closed_form(mode):
    """LOOKUP and CATMULL with uncertainties run the reference-order kernel by default, which carries the reference's own
    float32 noise (up to 2e-5 / 4e-5 on single elements against the closed form); the comparisons with the float64
    closed-form oracle ask for the closed-form kernels instead (CT_MERGE_CLOSED_FORM)."""
    return dict(reference_order=False) if mode in ("catmull", "lookup") else {}


def _pivot_proven(max_code, n_points, lookup, dtype_max):
    """Does the host proof allow the code-domain table addressing of ct::merge_pivot_kernel for these arguments?"""
    import ctypes
    from clair_torch_amd import _native as nv
    scale = ctypes.c_float()
    fn = nv.load().ct_pivot_interval_constants
    fn.argtypes = [ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
    return fn(float(max_code), int(n_points), int(lookup), int(dtype_max), ctypes.byref(scale)) == 0


def _kernel_name(dtype, max_code, mode, n_points, first=True):
    from clair_torch_amd import _native as nv
    interp = {"lookup": nv.INTERP_LOOKUP, "linear": nv.INTERP_LINEAR, "catmull": nv.INTERP_CATMULL}[mode]
    flags = (nv.MERGE_FIRST_BATCH if first else 0) | nv.MERGE_FINALIZE | nv.MERGE_STD_HINT | nv.MERGE_CLOSED_FORM
    return nv.load().ct_hdr_merge_kernel_name(nv.DTYPE_U8 if dtype == "u8" else nv.DTYPE_U16, float(max_code), interp,
                                              n_points, flags).decode()


def _run_partition(ops, stack, t, part, dev, **kw):
    batches, k = [], 0
    for b in part:
        batches.append(list(range(k, k + b)))
        k += b
    return _run_batches(ops, stack, t, batches, dev, **kw)


def _run_batches(ops, stack, t, batches, dev, **kw):
    """Stream the batches (index lists, each sorted by exposure like custom_collate) through ct_hdr_merge_batch."""
    has_std = kw.get("std") is not None or kw.get("std_mode", "none") != "none"
    st = ops.MergeState(tuple(stack.shape[1:]), dev, has_std) if len(batches) > 1 else None
    res = None
    std_full = kw.pop("std", None)
    for bi, idx in enumerate(batches):
        contiguous = idx == list(range(idx[0], idx[0] + len(idx)))
        if contiguous:
            sub = stack[idx[0]:idx[0] + len(idx)]
            sub_std = None if std_full is None else std_full[idx[0]:idx[0] + len(idx)]
        else:  # (torch has no CUDA gather for uint16: stack the images instead)
            sub = torch.stack([stack[i] for i in idx])
            sub_std = None if std_full is None else torch.stack([std_full[i] for i in idx])
        res = ops.hdr_merge_batch(sub.contiguous(), torch.from_numpy(t[idx]), state=st, finalize=bi == len(batches) - 1,
                                  std=None if sub_std is None else sub_std.contiguous(), **kw)
    return res


@pytest.mark.parametrize("as_codes", [True, False])
def test_merge_all_golden_cases(dev, as_codes):
    """All 132 recorded merge cases: {u8,u16} x {linear,lookup,catmull,no model} x {none,gauss} x 4 std modes x
    batch partitions {[8],[4,4],[3,3,2]}, fed as raw integer codes (in-kernel normalisation) and as float32."""
    from clair_torch_amd import ops
    from oracle import ct_oracle as oc
    g = golden("merge")
    t = g["merge_exposures"]
    lut = torch.from_numpy(g["merge_lut"]).to(dev)
    for key in [str(k) for k in g["merge_cases"]]:
        _, ub, mname, wname, sname, pname = key.split("_")
        codes = g[f"merge_{ub}_codes"]
        x = oc.normalize_codes(codes)
        stack = torch.from_numpy(codes if as_codes else x).to(dev)
        kw = dict(lut=None if mname == "nomodel" else lut, interp=None if mname == "nomodel" else mname,
                  gaussian_weight=wname == "gauss")
        if sname == "explicit":
            kw["std"] = torch.from_numpy(g[f"merge_{ub}_explicit_std"]).to(dev)
        elif sname != "none":
            kw.update(std_mode=sname, std_value=0.01 if sname == "constant" else 0.05)
        mean, std = _run_partition(ops, stack, t, PARTITIONS[pname], dev, **kw)
        assert mean.dtype == torch.float64 and mean.shape == (3, 16, 16)
        assert_parity(mean.cpu().numpy(), g[key + "_mean"], rtol=1e-5, norm_tol=1e-6, what=key + " mean")
        if sname == "none":
            assert std is None
        else:
            assert std.dtype == torch.float32
            assert_parity(std.cpu().numpy(), g[key + "_std"], norm_tol=NORM_TOL[mname], elem_tol=ELEM_TOL[mname],
                          what=key + " std")


@pytest.mark.parametrize("as_codes", [True, False])
def test_merge_shuffled_golden_cases(dev, as_codes):
    """Non-monotone batch composition (the scripts' default is shuffle: true; custom_collate sorts only within a batch and
    the variance depends on the batch order): the reference's recorded outputs for three explicit batch samplers, one of
    them starting with the three longest (most saturated) exposures -- the pivoted kernel's first-batch pivot is then far
    from the final mean."""
    from clair_torch_amd import ops
    from oracle import ct_oracle as oc
    g = golden("merge_shuffled")
    t = g["shuf_exposures"]
    lut = torch.from_numpy(g["shuf_lut"]).to(dev)
    keys = [str(k) for k in g["shuf_cases"]]
    assert len(keys) >= 90
    for key in keys:
        _, ub, mname, wname, sname, pname = key.split("_")
        codes = g[f"shuf_{ub}_codes"]
        stack = torch.from_numpy(codes if as_codes else oc.normalize_codes(codes)).to(dev)
        kw = dict(lut=None if mname == "nomodel" else lut, interp=None if mname == "nomodel" else mname,
                  gaussian_weight=wname == "gauss")
        if sname == "explicit":
            kw["std"] = torch.from_numpy(g[f"shuf_{ub}_explicit_std"]).to(dev)
        else:
            kw.update(std_mode=sname, std_value=0.01 if sname == "constant" else 0.05)
        mean, std = _run_batches(ops, stack, t, sampler_batches(g[f"shuf_sampler_{pname}"], t), dev, **kw)
        assert_parity(mean.cpu().numpy(), g[key + "_mean"], rtol=1e-5, norm_tol=1e-6, what=key + " mean")
        assert_parity(std.cpu().numpy(), g[key + "_std"], norm_tol=NORM_TOL[mname], elem_tol=ELEM_TOL[mname], what=key + " std")


@pytest.mark.parametrize("as_codes", [True, False])
def test_merge_reference_order_kernel_reproduces_the_recorded_bits(dev, as_codes):
    """ct_merge_exact.hip follows the reference's backward operation by operation (the same sequence as
    oracle/eager_torch.merge_stack_reference_order, which the CPU tests pin bit for bit to the recorded vectors).  Against
    the vectors recorded from the reference itself, in every mode, contiguous and shuffled batches:
      * without a weight function (no exp anywhere): at least 97 % of the elements bit for bit -- the VARIANCES are
        bit-identical (checked offline on dumps, tools/debug/exact_dump2.py), what is left is torch's CPU sqrt, which is
        not the correctly rounded root for ~1 % of its arguments on the machine the vectors were recorded on, while the
        kernel's is (its own expansion: hipcc's __fsqrt_rn was found 1 ULP low on 17 % of these variances on gfx950);
      * with Gaussian weights: at least 92 % bit for bit (additionally the kernel's exp is correctly rounded, torch's CPU
        exp -- Sleef, 1 ULP -- is not for 1.1 % of its arguments; the emulation with a correctly rounded exp gives
        96.1-97.5 % on these fixtures) and all within 1e-5 (1.1e-5 for CATMULL on uint16, where the reference's own
        chain amplifies that last bit, tests/test_oracle_golden.py).
    The emulation evaluated on this machine's CPU is compared too, at 1e-6: torch's CPU kernels are not bit-stable from
    one host to another (the recorded vectors are the build container's), so no bit-equality is asked of it here."""
    from clair_torch_amd import ops
    from oracle import ct_oracle as oc
    from oracle import eager_torch as oe
    exact_exp = lambda v: torch.exp(v.double()).float()   # noqa: E731
    n, worst_same = 0, 1.0
    for name, prefix in (("merge", "merge"), ("merge_shuffled", "shuf")):
        g = golden(name)
        t = g[f"{prefix}_exposures"]
        lut_h = torch.from_numpy(g[f"{prefix}_lut"])
        lut = lut_h.to(dev)
        for key in [str(k) for k in g[f"{prefix}_cases"]]:
            _, ub, mname, wname, sname, pname = key.split("_")
            if sname == "none":
                continue
            codes = g[f"{prefix}_{ub}_codes"]
            x = oc.normalize_codes(codes)
            sd = std_for(sname, x, g[f"{prefix}_{ub}_explicit_std"])
            if prefix == "shuf":
                batches = sampler_batches(g[f"shuf_sampler_{pname}"], t)
            else:
                batches, k = [], 0
                for b in PARTITIONS[pname]:
                    batches.append(list(range(k, k + b)))
                    k += b
            stack = torch.from_numpy(codes if as_codes else x).to(dev)
            kw = dict(lut=None if mname == "nomodel" else lut, interp=None if mname == "nomodel" else mname,
                      gaussian_weight=wname == "gauss", reference_order=True)
            if sname == "explicit":
                kw["std"] = torch.from_numpy(g[f"{prefix}_{ub}_explicit_std"]).to(dev)
            else:
                kw.update(std_mode=sname, std_value=0.01 if sname == "constant" else 0.05)
            mean, std = _run_batches(ops, stack, t, batches, dev, **kw)
            got, ref = std.cpu().numpy(), g[key + "_std"]
            same = float((got == ref).mean())
            floor = 0.97 if wname == "none" else 0.92
            assert same >= floor, f"{key}: only {same:.4f} of the elements bit-identical to the reference"
            worst_same = min(worst_same, same)
            tol = 1.1e-5 if (mname, ub) == ("catmull", "u16") else 1e-5
            assert_parity(got, ref, norm_tol=1e-6, elem_tol=tol, what=key + " std (reference order) vs golden")
            # (Gaussian weights: a last-bit difference of exp moves W_b, a float32 sum, and with it the float64 mean by ~6e-8)
            mt = dict(rtol=1e-13, norm_tol=1e-14) if wname == "none" else dict(rtol=1e-6, norm_tol=1e-7)
            assert_parity(mean.cpu().numpy(), g[key + "_mean"], what=key + " mean (reference order) vs golden", **mt)
            if n % 7 == 0:  # the emulation on this host, a sample of the cases
                mean_e, std_e = oe.merge_stack_reference_order(
                    torch.from_numpy(x), torch.from_numpy(np.ascontiguousarray(sd)), torch.from_numpy(t),
                    None if mname == "nomodel" else lut_h, "linear" if mname == "nomodel" else mname, wname == "gauss", batches,
                    exp=exact_exp)
                assert_parity(got, std_e.numpy(), rtol=1e-6, norm_tol=1e-6, what=key + " std vs the emulation on this host")
            n += 1
    assert n > 150
    from _util import OBSERVED
    OBSERVED.append({"test": "test_merge_reference_order_kernel_reproduces_the_recorded_bits", "what": "smallest share of bit-identical elements",
                     "norm": worst_same, "norm_tol": 0.92, "elem": worst_same, "elem_tol": 0.92, "n": n})


def test_merge_lookup_without_weight_raises(dev):
    from clair_torch_amd import ops
    g = golden("merge")
    stack = torch.from_numpy(g["merge_u8_codes"]).to(dev)
    with pytest.raises(RuntimeError, match="does not require grad"):
        ops.hdr_merge_batch(stack, torch.from_numpy(g["merge_exposures"]), lut=torch.from_numpy(g["merge_lut"]).to(dev),
                            interp="lookup", gaussian_weight=False, std_mode="constant", std_value=0.01)


def test_config_c1_through_public_api(dev):
    """BASELINE config C1 (8 x 256x256x3 uint8) through compute_hdr_image with the DataLoader plumbing:
    integer codes + gpu_transforms=[CastTo, Normalize] (fused), MULTIPLIER std derived in-kernel, batch sizes 8 and 4;
    and the reference-style path (float images + explicit std tensors from the dataset)."""
    from clair_torch_amd.common.enums import InterpMode, MissingStdMode
    from clair_torch_amd.common.transforms import CastTo, Normalize
    from clair_torch_amd.datasets import StackDataset, custom_collate
    from clair_torch_amd.inference import compute_hdr_image
    from clair_torch_amd.models import ICRFModelDirect
    from clair_torch_amd.training.losses import gaussian_value_weights
    from oracle import ct_oracle as oc
    g = golden("merge_c1")
    codes = torch.from_numpy(g["c1_codes"])
    model = ICRFModelDirect(icrf=torch.from_numpy(g["c1_lut"]), interpolation_mode=InterpMode.LINEAR).to(dev)
    for pname, bs in (("8", 8), ("44", 4)):
        ds = StackDataset(codes, g["c1_exposures"].tolist(), missing_std_mode=MissingStdMode.MULTIPLIER,
                          missing_std_value=0.05, materialize_std=False)
        loader = DataLoader(ds, batch_size=bs, shuffle=False, collate_fn=custom_collate)
        mean, std = compute_hdr_image(loader, "cuda", model, weight_fn=gaussian_value_weights,
                                      gpu_transforms=[CastTo("float32"), Normalize(max_val=255, min_val=0)])
        assert mean.dtype == torch.float64 and std.dtype == torch.float32 and mean.shape == (3, 256, 256)
        assert_parity(mean.cpu().numpy(), g[f"c1_{pname}_mean"], rtol=1e-5, norm_tol=1e-6, what="c1 mean")
        assert_parity(std.cpu().numpy(), g[f"c1_{pname}_std"], norm_tol=1e-5, elem_tol=1e-5, what="c1 std")
    # reference-style: float images, explicit std tensors
    x = torch.from_numpy(oc.normalize_codes(g["c1_codes"]))
    ds = StackDataset(x, g["c1_exposures"].tolist(), missing_std_mode=MissingStdMode.MULTIPLIER, missing_std_value=0.05)
    loader = DataLoader(ds, batch_size=4, shuffle=False, collate_fn=custom_collate)
    mean, std = compute_hdr_image(loader, torch.device("cuda:0"), model, weight_fn=gaussian_value_weights)
    assert_parity(mean.cpu().numpy(), g["c1_44_mean"], rtol=1e-5, norm_tol=1e-6, what="c1 mean (float path)")
    assert_parity(std.cpu().numpy(), g["c1_44_std"], norm_tol=1e-5, elem_tol=1e-5, what="c1 std (float path)")


@pytest.mark.parametrize("shape", [(5, 3, 37, 53), (7, 1, 64, 40), (4, 3, 31, 8), (9, 4, 16, 24)])
@pytest.mark.parametrize("dtype", ["u8", "u16", "f32"])
def test_merge_vs_oracle_ragged_shapes(dev, shape, dtype):
    """Odd widths (packet tail launch), 1 and 4 channels, against the float64 closed-form oracle."""
    from clair_torch_amd import ops
    from oracle import ct_oracle as oc
    n, c, h, w = shape
    rng = np.random.default_rng(hash((shape, dtype)) % (2 ** 32))
    t = 0.002 * 2.0 ** (np.arange(n) / 2.0)
    lut = np.stack([np.linspace(0, 1, 256, dtype=np.float32) ** np.float32(2.0 + 0.2 * k) for k in range(c)])
    if dtype == "f32":
        x = rng.random((n, c, h, w), dtype=np.float32)
        stack = torch.from_numpy(x).to(dev)
    else:
        codes = rng.integers(0, 256 if dtype == "u8" else 65536, size=shape).astype(np.uint8 if dtype == "u8" else np.uint16)
        x = oc.normalize_codes(codes)
        stack = torch.from_numpy(codes).to(dev)
    sd = (0.001 + 0.05 * rng.random(shape)).astype(np.float32)
    lut_d = torch.from_numpy(lut).to(dev)
    for mode in ("linear", "catmull", "lookup"):
        for part in ([n], [2, n - 2]):
            mean_o, std_o = oc.hdr_merge(x, sd, t, lut, mode, True, part)
            mean, std = _run_partition(ops, stack, t, part, dev, lut=lut_d, interp=mode, gaussian_weight=True,
                                       std=torch.from_numpy(sd).to(dev), **_closed_form(mode))
            assert_parity(mean.cpu().numpy(), mean_o, rtol=1e-5, norm_tol=1e-6, what=f"{mode} mean")
            assert_parity(std.cpu().numpy(), std_o, rtol=1e-5, norm_tol=1e-5, what=f"{mode} std")


@pytest.mark.parametrize("shape", [(5, 3, 37, 53), (7, 1, 64, 40), (20, 3, 8, 32)])
@pytest.mark.parametrize("dtype", ["u16", "f32"])
@pytest.mark.parametrize("mode", ["catmull", "linear", "lookup"])
def test_merge_reference_order_kernel_ragged_shapes(dev, shape, dtype, mode):
    """ct_merge_exact.hip on odd shapes, 20 exposures (torch.sum's 16-row cascade), streamed state, against the float32-order
    emulation evaluated on this host: 1e-6 on every element whose column lies in torch.sum's vectorised blocks (the first
    32 * (Q // 32) flattened columns; the last Q % 32 go through a differently associated sum on the CPU,
    tests/test_oracle_golden.py), 1e-5 everywhere."""
    from clair_torch_amd import ops
    from oracle import ct_oracle as oc
    from oracle import eager_torch as oe
    torch.set_num_threads(1)  # the reference's column blocks start at the beginning of each thread's chunk
    n, c, h, w = shape
    rng = np.random.default_rng(sum(shape))
    t = 0.002 * 2.0 ** (np.arange(n) / 3.0)
    lut = np.stack([np.linspace(0, 1, 256, dtype=np.float32) ** np.float32(2.0 + 0.2 * k) for k in range(c)])
    if dtype == "f32":
        x = rng.random((n, c, h, w), dtype=np.float32)
        stack = torch.from_numpy(x).to(dev)
    else:
        codes = rng.integers(0, 65536, size=shape).astype(np.uint16)
        x = oc.normalize_codes(codes)
        stack = torch.from_numpy(codes).to(dev)
    sd = (0.001 + 0.05 * rng.random(shape)).astype(np.float32)
    exact_exp = lambda v: torch.exp(v.double()).float()   # noqa: E731
    q = c * h * w
    main = (q // 32) * 32
    for part in ([n], [2, n - 2]):
        batches, k = [], 0
        for b in part:
            batches.append(list(range(k, k + b)))
            k += b
        mean_e, std_e = oe.merge_stack_reference_order(torch.from_numpy(x), torch.from_numpy(sd), torch.from_numpy(t),
                                                       torch.from_numpy(lut), mode, True, batches, exp=exact_exp)
        mean, std = _run_batches(ops, stack, t, batches, dev, lut=torch.from_numpy(lut).to(dev), interp=mode,
                                 gaussian_weight=True, std=torch.from_numpy(sd).to(dev), reference_order=True)
        got, want = std.cpu().numpy().reshape(-1), std_e.numpy().reshape(-1)
        # (torch's CPU kernels differ in the last bit from host to host, so no bit-equality with the emulation run HERE)
        assert_parity(got[:main], want[:main], rtol=1e-6, norm_tol=1e-6, what=f"reference order {mode} std")
        assert_parity(got, want, rtol=1e-5, norm_tol=1e-6, what=f"reference order {mode} std (with the tail columns)")
        gm, wm = mean.cpu().numpy().reshape(-1), mean_e.numpy().reshape(-1)
        assert_parity(gm[:main], wm[:main], rtol=1e-13, norm_tol=1e-14, what=f"reference order {mode} mean")
        assert_parity(gm, wm, rtol=1e-6, norm_tol=1e-7, what=f"reference order {mode} mean (with the tail columns: W_b one ulp apart)")


def test_merge_tiles_equal_whole(dev):
    """Row-band tiles with the global geometry give bit-identical results to the untiled launch (the LUT-row quirk
    p % C depends on global coordinates), including a width that is not a multiple of C."""
    from clair_torch_amd import ops
    rng = np.random.default_rng(7)
    n, c, h, w = 6, 3, 48, 20
    codes = torch.from_numpy(rng.integers(0, 65536, size=(n, c, h, w)).astype(np.uint16)).to(dev)
    t = torch.tensor(0.001 * 2.0 ** np.arange(n))
    lut = torch.stack([torch.linspace(0, 1, 256) ** p for p in (1.8, 2.2, 2.6)]).to(dev)
    mean, std = ops.hdr_merge_batch(codes, t, lut=lut, interp="linear", std_mode="multiplier", std_value=0.05)
    for bands in ([(0, 16), (16, 32), (32, 48)], [(0, 7), (7, 48)]):
        for r0, r1 in bands:
            tile = codes[:, :, r0:r1, :].contiguous()
            m_t, s_t = ops.hdr_merge_batch(tile, t, lut=lut, interp="linear", std_mode="multiplier", std_value=0.05,
                                           tile=ops.TileGeometry(h_global=h, row_offset=r0))
            assert torch.equal(m_t, mean[:, r0:r1]) and torch.equal(s_t, std[:, r0:r1])
    # a tile-local launch WITHOUT the global geometry must differ when (rows*W) % C != 0 (guards the test itself)
    m_bad, _ = ops.hdr_merge_batch(codes[:, :, 7:48].contiguous(), t, lut=lut, interp="linear", std_mode="multiplier",
                                   std_value=0.05)
    assert not torch.equal(m_bad, mean[:, 7:48])


@pytest.mark.parametrize("c,h,w", [(4, 5, 3), (2, 4, 3), (4, 3, 5)])
@pytest.mark.parametrize("dtype", ["u16", "f32"])
def test_merge_packets_crossing_channel_planes(dev, c, h, w, dtype):
    """Planes whose size is not a multiple of the packet while the image is: packets straddle channel planes (and,
    in a row band, skip the other bands' rows) -- the per-packet LUT-row bookkeeping must follow.  Whole image against
    the oracle, every row band bit-identical to the whole."""
    from clair_torch_amd import ops
    from oracle import ct_oracle as oc
    rng = np.random.default_rng(c * 100 + h * 10 + w)
    n = 6
    t = 0.002 * 2.0 ** (np.arange(n) / 2.0)
    lut = np.stack([np.linspace(0, 1, 64, dtype=np.float32) ** np.float32(1.8 + 0.3 * k) for k in range(c)])
    if dtype == "f32":
        x = rng.random((n, c, h, w), dtype=np.float32)
        stack = torch.from_numpy(x).to(dev)
    else:
        codes = rng.integers(0, 65536, size=(n, c, h, w)).astype(np.uint16)
        x = oc.normalize_codes(codes)
        stack = torch.from_numpy(codes).to(dev)
    sd = (0.05 * x).astype(np.float32)
    lut_d = torch.from_numpy(lut).to(dev)
    for mode in ("linear", "catmull", "lookup"):
        kw = dict(lut=lut_d, interp=mode, gaussian_weight=True, std_mode="multiplier", std_value=0.05, **_closed_form(mode))
        mean, std = ops.hdr_merge_batch(stack, torch.from_numpy(t), **kw)
        mean_o, std_o = oc.hdr_merge(x, sd, t, lut, mode, True, [n])
        assert_parity(mean.cpu().numpy(), mean_o, rtol=1e-5, norm_tol=1e-6, what=f"{mode} mean")
        assert_parity(std.cpu().numpy(), std_o, rtol=1e-5, norm_tol=1e-5, what=f"{mode} std")
        for r0, r1 in ((0, 1), (1, h), (0, h - 1)):
            m_t, s_t = ops.hdr_merge_batch(stack[:, :, r0:r1].contiguous(), torch.from_numpy(t),
                                           tile=ops.TileGeometry(h_global=h, row_offset=r0), **kw)
            assert torch.equal(m_t, mean[:, r0:r1]) and torch.equal(s_t, std[:, r0:r1]), (mode, r0, r1)


def _band_vs_oracle(codes_band, exposures, lut, mean_band, std_band, h_global, r0, what):
    """A full-width row band of a LARGE launch's real output against the float64 oracle run with the same global
    geometry (the LUT-row quirk p % C needs h_global / row_offset, oracle/ct_oracle.c)."""
    from oracle import ct_oracle as oc
    x = oc.normalize_codes(codes_band.cpu().numpy())
    m_o, s_o = oc.hdr_merge(x, x * np.float32(0.05), np.asarray(exposures), lut.cpu().numpy(), "linear", True,
                            tile=(h_global, r0))
    assert_parity(mean_band.cpu().numpy(), m_o, rtol=1e-5, norm_tol=1e-6, what=what + " mean")
    assert_parity(std_band.cpu().numpy(), s_o, rtol=1e-5, norm_tol=1e-5, what=what + " std")


def test_merge_full_size_properties(dev):
    """BASELINE config C2 size (32 x 4096 x 4096 x 3 uint16), one launch over the whole stack:
    (1) full-width row bands TAKEN FROM THAT LAUNCH'S OUTPUT (first rows, a band straddling the middle, last rows --
        the largest offsets the kernel forms) equal the float64 oracle run with the global geometry;
    (2) exposure-scale covariance: multiplying every exposure time by 2 halves mean and std exactly (power of two);
    (3) the float64-moment kernel (CT_MERGE_F64_MOMENTS, the round-1 path) agrees with the pivoted float32 one
        norm-wise to 1e-6 over all 50 M elements; (4) outputs finite, std >= 0."""
    from clair_torch_amd import ops
    from clair_torch_amd.datasets import synthetic_exposure_stack
    n, c, h, w = 32, 3, 4096, 4096
    codes, exposures = synthetic_exposure_stack(n, c, h, w, bits=16, stops_per_step=0.25, seed=1236, device=dev)
    lut = torch.stack([torch.linspace(0, 1, 256) ** p for p in (2.2, 2.4, 2.6)]).to(dev)
    t = torch.tensor(exposures, dtype=torch.float64)
    kw = dict(lut=lut, interp="linear", std_mode="multiplier", std_value=0.05)
    mean, std = ops.hdr_merge_batch(codes, t, **kw)
    assert torch.isfinite(mean).all() and torch.isfinite(std).all() and (std >= 0).all()
    for r0, r1 in ((0, 8), (2040, 2056), (4088, 4096)):
        _band_vs_oracle(codes[:, :, r0:r1], exposures, lut, mean[:, r0:r1], std[:, r0:r1], h, r0, f"C2 rows {r0}-{r1}")
    mean2, std2 = ops.hdr_merge_batch(codes, t * 2.0, **kw)
    assert torch.equal(mean2 * 2.0, mean) and torch.equal(std2 * 2.0, std)
    del mean2, std2
    mean64, std64 = ops.hdr_merge_batch(codes, t, force_f64_moments=True, **kw)
    assert float((mean64 - mean).norm() / mean64.norm()) < 1e-6
    assert float((std64.double() - std.double()).norm() / std64.double().norm()) < 1e-6


def test_config_c5_eight_bands_equal_whole_image(dev):
    """BASELINE config C5 on one GPU: the 32 x 8192 x 8192 x 3 uint16 stack (12.9 GB, resident in HBM) merged as one
    image and as the 8 row bands of 1024 rows the 8 ranks would hold.  Bands are generated independently from the
    global pixel coordinates (as each rank would) and must reproduce the whole image bit for bit; the per-band
    statistics vector C5 gathers is checked to sum to the whole image's."""
    from clair_torch_amd import ops
    from clair_torch_amd.datasets import synthetic_exposure_stack
    n, c, h, w, ranks = 32, 3, 8192, 8192, 8
    lut = torch.stack([torch.linspace(0, 1, 256) ** p for p in (2.2, 2.4, 2.6)]).to(dev)
    kw = dict(lut=lut, interp="linear", gaussian_weight=True, std_mode="multiplier", std_value=0.05)
    whole, exposures = synthetic_exposure_stack(n, c, h, w, bits=16, stops_per_step=0.25, seed=1238, device=dev)
    t = torch.tensor(exposures, dtype=torch.float64, device=dev)
    mean, std = ops.hdr_merge_batch(whole, t, **kw)
    del whole
    torch.cuda.empty_cache()
    band_rows = h // ranks
    sum_mean = torch.zeros(c, dtype=torch.float64, device=dev)
    for r in range(ranks):
        r0 = r * band_rows
        band, _ = synthetic_exposure_stack(n, c, h, w, bits=16, stops_per_step=0.25, seed=1238, device=dev,
                                           row_range=(r0, r0 + band_rows))
        m_b, s_b = ops.hdr_merge_batch(band, t, tile=ops.TileGeometry(h_global=h, row_offset=r0), **kw)
        assert torch.equal(m_b, mean[:, r0:r0 + band_rows]) and torch.equal(s_b, std[:, r0:r0 + band_rows]), r
        sum_mean += m_b.sum(dim=(1, 2))
        del band, m_b, s_b
    assert torch.allclose(sum_mean, mean.sum(dim=(1, 2)), rtol=1e-12)
    assert torch.isfinite(mean).all() and torch.isfinite(std).all()
    # a band of the WHOLE-image launch's output (rows beyond 2^31 / (3 * 8192 * 2) bytes into a plane) against the oracle
    for r0 in (5000, 8184):
        band, _ = synthetic_exposure_stack(n, c, h, w, bits=16, stops_per_step=0.25, seed=1238, device=dev,
                                           row_range=(r0, r0 + 8))
        _band_vs_oracle(band, exposures, lut, mean[:, r0:r0 + 8], std[:, r0:r0 + 8], h, r0, f"C5 rows {r0}-{r0 + 8}")


@pytest.mark.parametrize("n_points", [2, 100, 1000])
@pytest.mark.parametrize("dtype", ["u8", "u16"])
def test_merge_unusual_lut_sizes(dev, n_points, dtype):
    """LUT lengths other than 256: the code->LUT-coordinate fold is only taken when it reproduces the reference's
    float32 index for every code (ct_index_constants); either way the result must match the oracle, knots included."""
    from clair_torch_amd import ops
    from oracle import ct_oracle as oc
    rng = np.random.default_rng(n_points)
    n, c, h, w = 6, 3, 16, 24
    hi = 256 if dtype == "u8" else 65536
    codes = rng.integers(0, hi, size=(n, c, h, w)).astype(np.uint8 if dtype == "u8" else np.uint16)
    codes.reshape(-1)[:4] = [0, hi - 1, (hi - 1) // 3, (hi - 1) // 5]        # exact knots / ends
    x = oc.normalize_codes(codes)
    t = 0.001 * 2.0 ** np.arange(n)
    lut = np.stack([np.linspace(0, 1, n_points, dtype=np.float32) ** np.float32(p) for p in (1.7, 2.2, 2.7)])
    for mode in ("linear", "lookup", "catmull"):
        # raw codes run the pivoted code-domain kernel for ANY LUT length the host proof accepts (steps that are not a whole
        # number of codes included; CATMULL's closed form too since round 3: the interval's cubic in the code offset);
        # refused lengths the generic pivoted kernel
        name = _kernel_name(dtype, hi - 1, mode, n_points)
        assert ("merge_pivot_kernel" in name) == _pivot_proven(hi - 1, n_points, mode == "lookup", hi - 1), name
        mean_o, std_o = oc.hdr_merge(x, x * np.float32(0.05), t, lut, mode, True)
        mean, std = ops.hdr_merge_batch(torch.from_numpy(codes).to(dev), torch.from_numpy(t), lut=torch.from_numpy(lut).to(dev),
                                        interp=mode, std_mode="multiplier", std_value=0.05, **_closed_form(mode))
        assert_parity(mean.cpu().numpy(), mean_o, rtol=1e-5, norm_tol=1e-6, what=f"L={n_points} {mode} mean")
        assert_parity(std.cpu().numpy(), std_o, rtol=1e-5, norm_tol=1e-5, what=f"L={n_points} {mode} std")


@pytest.mark.parametrize("dtype,n_points", [("u16", 2), ("u16", 4), ("u16", 16), ("u16", 52), ("u16", 258), ("u16", 772),
                                            ("u8", 2), ("u8", 16), ("u8", 52), ("u8", 86), ("u8", 256)])
@pytest.mark.parametrize("shape", [(3, 16, 24), (3, 7, 9)])
def test_merge_whole_step_lut_sizes_on_the_typed_load_kernel(dev, dtype, n_points, shape):
    """LUT lengths whose step max_code / (L-1) is a whole number of codes go through ct::merge_pivot_kernel: the codes
    arrive as floats from typed buffer loads and the interval is one round-down FMA (ct_pivot_floor_constants).  Packets of
    four and the ragged one-element path (7 x 9 planes), knots, their neighbours and both ends included, first batch and
    streamed state, against the float64 oracle."""
    from clair_torch_amd import _native as nv
    from clair_torch_amd import ops
    from oracle import ct_oracle as oc
    maxc = 255 if dtype == "u8" else 65535
    step = maxc // (n_points - 1)
    name = nv.load().ct_hdr_merge_kernel_name(nv.DTYPE_U8 if dtype == "u8" else nv.DTYPE_U16, float(maxc), nv.INTERP_LINEAR,
                                              n_points, nv.MERGE_FIRST_BATCH | nv.MERGE_FINALIZE).decode()
    # ... unless the reference's own float32 index is not code // step (L = 772: the proof refuses, the generic kernel runs)
    proven = _pivot_proven(maxc, n_points, False, maxc)
    assert proven == (n_points != 772)
    assert ("merge_pivot_kernel" in name and "typed buffer loads" in name) == proven
    rng = np.random.default_rng(1000 * n_points + shape[1])
    n = 6
    codes = rng.integers(0, maxc + 1, size=(n,) + shape).astype(np.uint8 if dtype == "u8" else np.uint16)
    flat = codes.reshape(-1)
    knots = np.array([0, maxc, step, step - 1 if step > 1 else 0, min(maxc, step + 1), maxc - step, maxc - 1,
                      (n_points - 2) * step, (n_points // 2) * step, max(0, (n_points // 2) * step - 1)])
    flat[:knots.size] = knots.astype(flat.dtype)
    x = oc.normalize_codes(codes)
    t = 0.001 * 2.0 ** np.arange(n)
    lut = np.stack([np.linspace(0, 1, n_points, dtype=np.float32) ** np.float32(p) for p in (1.7, 2.2, 2.7)])
    for part in ([n], [2, 4]):
        mean_o, std_o = oc.hdr_merge(x, x * np.float32(0.05), t, lut, "linear", True, part)
        mean, std = _run_partition(ops, torch.from_numpy(codes).to(dev), t, part, dev, lut=torch.from_numpy(lut).to(dev),
                                   interp="linear", std_mode="multiplier", std_value=0.05)
        assert_parity(mean.cpu().numpy(), mean_o, rtol=1e-5, norm_tol=1e-6, what=f"{dtype} L={n_points} typed mean")
        assert_parity(std.cpu().numpy(), std_o, rtol=1e-5, norm_tol=1e-5, what=f"{dtype} L={n_points} typed std")


@pytest.mark.parametrize("mode", ["linear", "lookup", "catmull"])
def test_codes_above_max_code_are_clamped_like_the_reference(dev, mode):
    """12-bit data in a uint16 container, Normalize(4095): codes above max_code give x > 1, which the reference's model
    clamps to the top of the LUT with zero gradient (base.py:146,166,190) while weights and MULTIPLIER sigma keep using
    the unclamped x.  Merge (single batch and streamed), against the float64 oracle."""
    from clair_torch_amd import ops
    from oracle import ct_oracle as oc
    rng = np.random.default_rng(4095)
    n, c, h, w = 6, 3, 12, 20
    codes = rng.integers(0, 4096, size=(n, c, h, w)).astype(np.uint16)
    codes.reshape(-1)[::7] = rng.integers(4096, 6000, size=codes.reshape(-1)[::7].shape).astype(np.uint16)
    codes.reshape(-1)[:3] = [4095, 4096, 65535]
    x = (codes.astype(np.float32) / np.float32(4095.0)).astype(np.float32)
    t = 0.001 * 2.0 ** np.arange(n)
    lut = np.stack([np.linspace(0, 1, 256, dtype=np.float32) ** np.float32(p) for p in (1.7, 2.2, 2.7)])
    # Normalize(4095) on uint16: the pivoted code-domain kernel with the clamp (one v_min per sample), not the generic one
    assert "merge_pivot_kernel" in _kernel_name("u16", 4095, mode, 256)
    for part in ([n], [2, 4]):
        mean_o, std_o = oc.hdr_merge(x, x * np.float32(0.05), t, lut, mode, True, part)
        mean, std = _run_partition(ops, torch.from_numpy(codes).to(dev), t, part, dev, lut=torch.from_numpy(lut).to(dev),
                                   interp=mode, std_mode="multiplier", std_value=0.05, max_code=4095.0, **_closed_form(mode))
        assert_parity(mean.cpu().numpy(), mean_o, rtol=1e-5, norm_tol=1e-6, what=f"max_code 4095 {mode} mean")
        assert_parity(std.cpu().numpy(), std_o, rtol=1e-5, norm_tol=1e-5, what=f"max_code 4095 {mode} std")


@pytest.mark.parametrize("shape", [(3, 64, 96), (3, 7, 9), (1, 5, 3), (4, 33, 64)])
@pytest.mark.parametrize("with_std", [True, False])
def test_band_statistics_match_torch_reductions(dev, shape, with_std):
    """ct_band_stats (what configuration C5 gathers per band): min / max exactly, sums to float64 rounding; even and odd
    plane sizes (vector and scalar loads), and bands combine to the whole image."""
    from clair_torch_amd import ops
    gen = torch.Generator().manual_seed(shape[1])
    mean = (torch.rand(shape, generator=gen, dtype=torch.float64) * 3.0 - 1.0).to(dev)
    std = torch.rand(shape, generator=gen, dtype=torch.float32).to(dev) if with_std else None
    out = ops.band_stats(mean, std)
    flat = mean.reshape(shape[0], -1)
    ref = [flat.amin(dim=1), flat.amax(dim=1), flat.sum(dim=1)]
    if with_std:
        sflat = std.reshape(shape[0], -1)
        ref += [sflat.amin(dim=1).double(), sflat.amax(dim=1).double(), sflat.double().sum(dim=1)]
    else:
        ref += [torch.zeros(shape[0], dtype=torch.float64, device=dev)] * 3
    ref = torch.stack(ref)
    assert out.shape == (6, shape[0])
    assert torch.equal(out[[0, 1, 3, 4]], ref[[0, 1, 3, 4]])
    assert torch.allclose(out[[2, 5]], ref[[2, 5]], rtol=1e-13, atol=1e-13)
    assert torch.equal(out, ops.band_stats(mean, std))  # deterministic
    if shape[1] >= 7:  # two row bands combine to the whole
        a = ops.band_stats(mean[:, :3].contiguous(), std[:, :3].contiguous() if with_std else None)
        b = ops.band_stats(mean[:, 3:].contiguous(), std[:, 3:].contiguous() if with_std else None)
        comb = torch.stack([torch.minimum(a[0], b[0]), torch.maximum(a[1], b[1]), a[2] + b[2],
                            torch.minimum(a[3], b[3]), torch.maximum(a[4], b[4]), a[5] + b[5]])
        assert torch.equal(comb[[0, 1, 3, 4]], out[[0, 1, 3, 4]]) and torch.allclose(comb[[2, 5]], out[[2, 5]], rtol=1e-13, atol=1e-13)



class _RetryCounter:
    """ct_merge_set_retry_counter around a block: counts the wavefronts of ct::merge_pivot_kernel that repeated a batch."""

    def __init__(self, dev):
        self.buf = torch.zeros(1, dtype=torch.int64, device=dev)

    def __enter__(self):
        from clair_torch_amd import _native as nv
        import ctypes
        nv.load().ct_merge_set_retry_counter.argtypes = [ctypes.c_void_p]
        nv.load().ct_merge_set_retry_counter(ctypes.c_void_p(self.buf.data_ptr()))
        return self

    def __exit__(self, *exc):
        from clair_torch_amd import _native as nv
        torch.cuda.synchronize()
        nv.load().ct_merge_set_retry_counter(None)

    def value(self):
        torch.cuda.synchronize()
        return int(self.buf.item())


@pytest.mark.parametrize("dtype", ["u8", "u16"])
@pytest.mark.parametrize("kind", ["plateau_then_rise", "jump"])
def test_pivot_kernel_exact_offset_staging_on_steep_luts(dev, dtype, kind):
    """the `rough` staging of ct::merge_pivot_kernel ({g[i], S} entries and the exactly formed offset
    code - i * step, taken when |A| > 64 max|g| somewhere): a LUT with ~100 leading zeros and a steep rise, and one with a
    jump, L = 256, uint8 and uint16 -- against the float64-moment kernel and the float64 oracle."""
    from clair_torch_amd import ops
    from oracle import ct_oracle as oc
    rng = np.random.default_rng(77)
    n, c, h, w = 6, 3, 16, 32
    hi = 256 if dtype == "u8" else 65536
    codes = rng.integers(0, hi, size=(n, c, h, w)).astype(np.uint8 if dtype == "u8" else np.uint16)
    x = oc.normalize_codes(codes)
    t = 0.001 * 2.0 ** np.arange(n)
    grid = np.linspace(0, 1, 256, dtype=np.float64)
    if kind == "plateau_then_rise":
        rows = [np.where(grid < 100 / 255, 0.0, ((grid - 100 / 255) / (155 / 255)) ** p) for p in (3.0, 4.0, 5.0)]
    else:
        rows = [grid ** p * np.where(np.arange(256) >= 200, 9.0, 1.0) for p in (2.2, 2.4, 2.6)]
    lut = np.stack(rows).astype(np.float32)
    assert "merge_pivot_kernel" in _kernel_name(dtype, hi - 1, "linear", 256)
    lut_d = torch.from_numpy(lut).to(dev)
    for part in ([n], [2, 4]):
        kw = dict(lut=lut_d, interp="linear", gaussian_weight=True, std_mode="multiplier", std_value=0.05)
        mean, std = _run_partition(ops, torch.from_numpy(codes).to(dev), t, part, dev, **kw)
        mean_f, std_f = _run_partition(ops, torch.from_numpy(codes).to(dev), t, part, dev, force_f64_moments=True, **kw)
        mean_o, std_o = oc.hdr_merge(x, x * np.float32(0.05), t, lut, "linear", True, part)
        assert_parity(mean.cpu().numpy(), mean_o, rtol=1e-5, norm_tol=1e-6, what=f"{kind} mean vs oracle")
        # (jump LUT on random codes: single pixels where one exposure carries the mean -- there the oracle, like the
        # reference, forms m_b from the float32-rounded sum of weights, which y_n - m_b amplifies to ~2e-5)
        assert_parity(std.cpu().numpy(), std_o, norm_tol=1e-5, elem_tol=1e-5 if kind != "jump" else 3e-5, what=f"{kind} std vs oracle")
        assert_parity(std.cpu().numpy(), std_f.cpu().numpy(), rtol=1e-5, norm_tol=1e-5, what=f"{kind} std vs float64 moments")
        assert_parity(mean.cpu().numpy(), mean_f.cpu().numpy(), rtol=1e-5, norm_tol=1e-6, what=f"{kind} mean vs float64 moments")


@pytest.mark.parametrize("dtype", ["u8", "u16"])
def test_pivot_kernel_retry_pass_is_taken_and_exact(dev, dtype):
    """the ill-conditioned-pivot repeat of ct::merge_pivot_kernel.  A stack whose MIDDLE exposure (the first
    batch's pivot seed) is black or saturated while the others are consistent, then a second streamed batch far from the
    running mean: the retry counter must move, and the result must still match the float64-moment kernel and the oracle."""
    from clair_torch_amd import ops
    from oracle import ct_oracle as oc
    rng = np.random.default_rng(123)
    n, c, h, w = 8, 3, 16, 64
    maxc = 255 if dtype == "u8" else 65535
    t = 0.001 * 2.0 ** np.arange(n)
    e = rng.random((c, h, w)) * (2.0 / np.sqrt(t[0] * t[-1]))
    lin = np.clip(e[None] * t[:, None, None, None], 0.0, 1.0)
    codes = np.rint(lin ** (1 / 2.2) * maxc).astype(np.uint8 if dtype == "u8" else np.uint16)
    for probe in (n // 2, 5 // 2):             # the pivot seeds of the first batches of both partitions below:
        codes[probe, :, :, : w // 2] = 0       # black on the left half ...
        codes[probe, :, :, w // 2:] = maxc     # ... and saturated on the right half
    codes[5:] = np.rint(np.clip(codes[5:].astype(np.float64) * 0.2, 0, maxc)).astype(codes.dtype)   # second batch far from the mean
    x = oc.normalize_codes(codes)
    lut = np.stack([np.linspace(0, 1, 256, dtype=np.float32) ** np.float32(p) for p in (2.2, 2.4, 2.6)])
    lut_d = torch.from_numpy(lut).to(dev)
    kw = dict(lut=lut_d, interp="linear", gaussian_weight=True, std_mode="multiplier", std_value=0.05)
    for part in ([n], [5, 3]):
        with _RetryCounter(dev) as counter:
            mean, std = _run_partition(ops, torch.from_numpy(codes).to(dev), t, part, dev, **kw)
            retried = counter.value()
        assert retried > 0, f"the retry pass never ran (partition {part})"
        mean_f, std_f = _run_partition(ops, torch.from_numpy(codes).to(dev), t, part, dev, force_f64_moments=True, **kw)
        mean_o, std_o = oc.hdr_merge(x, x * np.float32(0.05), t, lut, "linear", True, part)
        assert_parity(mean.cpu().numpy(), mean_o, rtol=1e-5, norm_tol=1e-6, what="retry mean vs oracle")
        assert_parity(std.cpu().numpy(), std_o, rtol=1e-5, norm_tol=1e-5, what="retry std vs oracle")
        assert_parity(std.cpu().numpy(), std_f.cpu().numpy(), rtol=1e-5, norm_tol=1e-5, what="retry std vs float64 moments")


@pytest.mark.parametrize("dtype,max_code", [("u16", 65535), ("u8", 255), ("u16", 4095)])
@pytest.mark.parametrize("mode,gauss,std_mode", [("linear", True, "multiplier"), ("linear", True, "explicit"), ("linear", False, "constant"),
                                                 ("linear", True, "none"), (None, True, "multiplier"), ("lookup", True, "constant"),
                                                 ("lookup", False, "none"), ("catmull", True, "none"), ("catmull", True, "multiplier")])
def test_merge_batches_one_launch_equals_one_launch_per_batch(dev, dtype, max_code, mode, gauss, std_mode):
    """ct_hdr_merge_batches: several consecutive batches in ONE launch of ct::merge_pivot_kernel, the streaming state in
    registers in between (VERDICT r2 missing #3: the reference's default is batch_size: 4).  Bit for bit what one launch
    per batch gives -- first call of a merge, a later call continuing from a MergeState, 2 to 16 batches of unequal sizes,
    a batch composition that is not monotone in the exposure time -- and against the float64 oracle."""
    from clair_torch_amd import ops
    from oracle import ct_oracle as oc
    rng = np.random.default_rng(31 + max_code)
    n, c, h, w = 23, 3, 12, 32
    t = 0.0005 * 2.0 ** (np.arange(n) / 3.0)
    e = rng.random((c, h, w)) * (2.0 / np.sqrt(t[0] * t[-1]))
    lin = np.clip(e[None] * t[:, None, None, None], 0.0, 1.0)
    hi = 256 if dtype == "u8" else 65536
    codes = np.rint(lin ** (1 / 2.2) * max_code + 0.3 * max_code * (rng.random(lin.shape) < 0.02)).clip(0, hi - 1)
    codes = codes.astype(np.uint8 if dtype == "u8" else np.uint16)
    if max_code == 255 and dtype == "u8" or max_code == 65535:
        x = oc.normalize_codes(codes)
    else:
        x = (codes.astype(np.float32) / np.float32(max_code)).astype(np.float32)
    sd = (0.002 + 0.03 * rng.random(codes.shape)).astype(np.float32)
    lut = np.stack([np.linspace(0, 1, 256, dtype=np.float32) ** np.float32(p) for p in (2.2, 2.4, 2.6)])
    stack, sd_d = torch.from_numpy(codes).to(dev), torch.from_numpy(sd).to(dev)
    kw = dict(lut=None if mode is None else torch.from_numpy(lut).to(dev), interp=mode, gaussian_weight=gauss, max_code=float(max_code),
              reference_order=False)
    if std_mode != "explicit":
        kw.update(std_mode=std_mode, std_value=0.01 if std_mode == "constant" else 0.05)
    order = rng.permutation(n)
    cuts = [0, 4, 5, 9, 12, 16, 20, 23]   # batch sizes 4 1 4 3 4 4 3, composition shuffled, each batch sorted like collate
    batches = [sorted(order[a:b].tolist(), key=lambda i: t[i]) for a, b in zip(cuts[:-1], cuts[1:])]

    def one_per_batch(bs, state):
        res = None
        for k, idx in enumerate(bs):
            sub = torch.stack([stack[i] for i in idx])
            res = ops.hdr_merge_batch(sub, torch.from_numpy(t[idx]), state=state, finalize=k == len(bs) - 1,
                                      std=torch.stack([sd_d[i] for i in idx]) if std_mode == "explicit" else None, **kw)
        return res

    def one_launch(bs, state, finalize=True):
        subs = [torch.stack([stack[i] for i in idx]) for idx in bs]
        stds = [torch.stack([sd_d[i] for i in idx]) for idx in bs] if std_mode == "explicit" else None
        return ops.hdr_merge_batches(subs, [torch.from_numpy(t[idx]) for idx in bs], stds=stds, state=state, finalize=finalize,
                                     require_one_launch=True, **kw)

    has_std = std_mode != "none"
    # (a) the whole merge in one call, no state buffers at all
    ref = one_per_batch(batches, ops.MergeState((c, h, w), dev, has_std))
    got = one_launch(batches, None)
    assert torch.equal(got[0], ref[0]) and (not has_std or torch.equal(got[1], ref[1]))
    # (b) two calls: the second continues from the MergeState the first one left
    st = ops.MergeState((c, h, w), dev, has_std)
    assert one_launch(batches[:3], st, finalize=False) is None and st.batches == 3
    got2 = one_launch(batches[3:], st)
    assert torch.equal(got2[0], ref[0]) and (not has_std or torch.equal(got2[1], ref[1]))
    # (c) the float64 oracle with the same batch composition
    flat = [i for b in batches for i in b]
    sdo = {"explicit": sd, "constant": np.full_like(x, np.float32(0.01)), "multiplier": x * np.float32(0.05), "none": None}[std_mode]
    mean_o, std_o = oc.hdr_merge(np.ascontiguousarray(x[flat]), None if sdo is None else np.ascontiguousarray(sdo[flat]),
                                 np.ascontiguousarray(t[flat]), lut if mode is not None else None, mode or "linear", gauss,
                                 [len(b) for b in batches])
    assert_parity(got[0].cpu().numpy(), mean_o, rtol=1e-5, norm_tol=1e-6, what="multi-batch mean vs oracle")
    if has_std:
        assert_parity(got[1].cpu().numpy(), std_o, norm_tol=1e-5, elem_tol=1e-5 if mode != "lookup" else 5e-5, what="multi-batch std vs oracle")
    # 16 single-exposure batches in one launch, 17 refused by the wrapper
    singles = [[i] for i in range(16)]
    ref16 = one_per_batch(singles, ops.MergeState((c, h, w), dev, has_std))
    got16 = one_launch(singles, None)
    assert torch.equal(got16[0], ref16[0]) and (not has_std or torch.equal(got16[1], ref16[1]))
    with pytest.raises(ValueError, match="at most 16"):
        one_launch([[i] for i in range(17)], None)


def test_merge_batches_falls_back_where_one_launch_cannot_run(dev):
    """Odd plane sizes (no whole packets), float32 pixels, a LUT length the host proof refuses (772 points on uint16: the
    generic kernel): ct_hdr_merge_batches walks the batches with one launch each -- same results as the explicit loop;
    with require_one_launch it says so instead."""
    from clair_torch_amd import ops
    rng = np.random.default_rng(5)
    n, c, h, w = 9, 3, 7, 9
    t = 0.001 * 2.0 ** np.arange(n)
    lut = torch.stack([torch.linspace(0, 1, 256) ** p for p in (2.2, 2.4, 2.6)]).to(dev)
    cases = [(torch.from_numpy(rng.integers(0, 65536, size=(n, c, h, w)).astype(np.uint16)).to(dev), "linear"),
             (torch.from_numpy(rng.random((n, c, 8, 8), dtype=np.float32)).to(dev), "linear"),
             (torch.from_numpy(rng.integers(0, 65536, size=(n, c, 8, 8)).astype(np.uint16)).to(dev), "linear772")]
    assert not _pivot_proven(65535, 772, False, 65535)
    lut772 = torch.stack([torch.linspace(0, 1, 772) ** p for p in (2.2, 2.4, 2.6)]).to(dev)
    for stack, mode in cases:
        std_mode = "multiplier"
        kw = dict(lut=lut772 if mode == "linear772" else lut, interp="linear", gaussian_weight=True, std_mode=std_mode, std_value=0.05)
        parts = [(0, 4), (4, 6), (6, 9)]
        st = ops.MergeState(tuple(stack.shape[1:]), dev, std_mode != "none")
        for k, (a, b) in enumerate(parts):
            ref = ops.hdr_merge_batch(stack[a:b], torch.from_numpy(t[a:b]), state=st, finalize=k == 2, **kw)
        got = ops.hdr_merge_batches([stack[a:b] for a, b in parts], [torch.from_numpy(t[a:b]) for a, b in parts], **kw)
        assert torch.equal(got[0], ref[0]) and (std_mode == "none" or torch.equal(got[1], ref[1]))
        with pytest.raises(RuntimeError, match="unsupported"):
            ops.hdr_merge_batches([stack[a:b] for a, b in parts], [torch.from_numpy(t[a:b]) for a, b in parts],
                                  require_one_launch=True, **kw)


@pytest.mark.parametrize("kind,shape", [("u16", (12, 32)), ("u16", (7, 9)), ("f32", (8, 8)), ("u8", (5, 12))])
@pytest.mark.parametrize("mode,std_mode", [("catmull", "multiplier"), ("lookup", "constant"), ("catmull", "explicit"),
                                           ("linear", "multiplier")])
def test_merge_batches_reference_order_one_launch(dev, kind, shape, mode, std_mode):
    """The reference-order kernel (default for LOOKUP / CATMULL uncertainties; LINEAR on request) walks several batches per
    launch as well, the streaming state in registers in between: bit for bit what one launch per batch gives -- shuffled
    batch composition, unequal sizes incl. one of 1 and one of 9 exposures (LDS cache levels 2, 1 and 0 inside one launch),
    packets and odd planes, integer codes and float pixels, explicit uncertainties, a second call continuing from a
    MergeState -- and the recorded-order emulation's numbers are those of the per-batch route by construction."""
    from clair_torch_amd import ops
    rng = np.random.default_rng(sum(shape) + len(mode))
    n, c = 21, 3
    h, w = shape
    t = 0.0005 * 2.0 ** (np.arange(n) / 3.0)
    if kind == "f32":
        stack = torch.from_numpy(rng.random((n, c, h, w), dtype=np.float32)).to(dev)
    else:
        hi = 256 if kind == "u8" else 65536
        stack = torch.from_numpy(rng.integers(0, hi, size=(n, c, h, w)).astype(np.uint8 if kind == "u8" else np.uint16)).to(dev)
    sd_d = torch.from_numpy((0.002 + 0.03 * rng.random((n, c, h, w))).astype(np.float32)).to(dev)
    lut = torch.stack([torch.linspace(0, 1, 256) ** p for p in (2.2, 2.4, 2.6)]).to(dev)
    kw = dict(lut=lut, interp=mode, gaussian_weight=True, reference_order=True)
    if std_mode != "explicit":
        kw.update(std_mode=std_mode, std_value=0.01 if std_mode == "constant" else 0.05)
    order = rng.permutation(n)
    cuts = [0, 4, 5, 14, 17, 21]   # batch sizes 4 1 9 3 4
    batches = [sorted(order[a:b].tolist(), key=lambda i: t[i]) for a, b in zip(cuts[:-1], cuts[1:])]

    def one_per_batch(bs, state):
        res = None
        for k, idx in enumerate(bs):
            res = ops.hdr_merge_batch(torch.stack([stack[i] for i in idx]), torch.from_numpy(t[idx]), state=state,
                                      finalize=k == len(bs) - 1,
                                      std=torch.stack([sd_d[i] for i in idx]) if std_mode == "explicit" else None, **kw)
        return res

    def one_launch(bs, state, finalize=True):
        stds = [torch.stack([sd_d[i] for i in idx]) for idx in bs] if std_mode == "explicit" else None
        return ops.hdr_merge_batches([torch.stack([stack[i] for i in idx]) for idx in bs], [torch.from_numpy(t[idx]) for idx in bs],
                                     stds=stds, state=state, finalize=finalize, require_one_launch=True, **kw)

    ref = one_per_batch(batches, ops.MergeState((c, h, w), dev, True))
    got = one_launch(batches, None)
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    st = ops.MergeState((c, h, w), dev, True)
    assert one_launch(batches[:2], st, finalize=False) is None
    got2 = one_launch(batches[2:], st)
    assert torch.equal(got2[0], ref[0]) and torch.equal(got2[1], ref[1])
    # small batches only: the cached variants carry the state too
    small = [b for b in batches if len(b) <= 4]
    ref_s = one_per_batch(small, ops.MergeState((c, h, w), dev, True))
    got_s = one_launch(small, None)
    assert torch.equal(got_s[0], ref_s[0]) and torch.equal(got_s[1], ref_s[1])


@pytest.mark.parametrize("dtype", ["u8", "u16"])
@pytest.mark.parametrize("mode", ["linear", "lookup", "catmull", None])
def test_merge_edge_cases(dev, dtype, mode):
    """The corners of the batch loop against the float64 oracle: batches of ONE exposure streamed one by one, a single
    exposure as the whole stack, a 1x1 image and a one-row image, stacks that are black or saturated in every exposure (all
    weights at their minimum, zero spread about the pivot), and 70 / 300 exposures in one batch (prefetch ring wrap-around,
    torch.sum's second summation level in the reference-order kernel)."""
    from clair_torch_amd import ops
    from oracle import ct_oracle as oc
    hi = 255 if dtype == "u8" else 65535
    npdt = np.uint8 if dtype == "u8" else np.uint16
    rng = np.random.default_rng(77)
    c = 3
    lut = None if mode is None else np.stack([np.linspace(0, 1, 256, dtype=np.float32) ** np.float32(2.0 + 0.2 * k) for k in range(c)])
    lut_d = None if mode is None else torch.from_numpy(lut).to(dev)
    kw = dict(lut=lut_d, interp=mode, gaussian_weight=True, std_mode="multiplier", std_value=0.05)

    def check(codes, t, part, what, closed_form=True, rtol=1e-5, std_is_noise=False):
        x = oc.normalize_codes(codes)
        mean_o, std_o = oc.hdr_merge(x, x * np.float32(0.05), t, lut, mode or "none", True, part)
        extra = _closed_form(mode) if (closed_form and mode is not None) else {}
        mean, std = _run_partition(ops, torch.from_numpy(codes).to(dev), t, part, dev, **kw, **extra)
        assert torch.isfinite(mean).all() and torch.isfinite(std).all(), what
        assert_parity(mean.cpu().numpy(), mean_o, rtol=rtol, norm_tol=1e-6, what=f"{what} mean")
        if std_is_noise:
            # ONE exposure in LOOKUP mode: the only gradient path is the weight's, and it multiplies y - m = y * 1e-6 / (w + 1e-6),
            # i.e. the rounding of float32(w + 1e-6).  The reference's own autograd and the float64 closed form differ by
            # 4 % per element there (measured on this stack); what can be asserted is that the result is that small.
            assert float((std / mean.abs().clamp(min=1e-30)).max()) < 1e-2 and np.max(std_o / np.maximum(np.abs(mean_o), 1e-30)) < 1e-2
        else:
            assert_parity(std.cpu().numpy(), std_o, rtol=rtol, norm_tol=1e-5, what=f"{what} std")

    t6 = 0.002 * 2.0 ** (np.arange(6) / 2.0)
    codes = rng.integers(0, hi + 1, size=(6, c, 9, 12)).astype(npdt)
    check(codes, t6, [1] * 6, "batches of one")
    check(codes[:1], t6[:1], [1], "single exposure", std_is_noise=mode == "lookup")
    check(codes[:, :, :1, :1].copy(), t6, [6], "1x1 image")
    check(codes[:, :, :1, :].copy(), t6, [4, 2], "one-row image")
    check(np.zeros_like(codes), t6, [3, 3], "black stack")
    check(np.full_like(codes, hi), t6, [3, 3], "saturated stack")
    for n in (70, 300):
        tn = 0.002 * 2.0 ** (np.arange(n) / 16.0)
        big = rng.integers(0, hi + 1, size=(n, c, 4, 8)).astype(npdt)
        check(big, tn, [n], f"{n} exposures")
        if mode in ("lookup", "catmull"):  # the default route of these modes: the reference-order kernel (float32 noise: 3e-5)
            check(big, tn, [n], f"{n} exposures, reference order", closed_form=False, rtol=3e-5)
