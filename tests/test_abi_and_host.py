"""CPU tests: the C-ABI library loads and exports every symbol of include/clair_hip.h, argument validation that
returns before any launch, the host-side code-normalisation proofs, and the host logic around the kernels
(collation, datasets, transform fusion, pair lists, loud failure without a GPU)."""
import ctypes
import os
import sys
import re

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from clair_torch_amd import build, _native
    build.build()
    return _native.load()


def test_every_declared_symbol_is_exported(lib):
    header = open(os.path.join(ROOT, "include", "clair_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(ct_[a-z_0-9]+)\s*\(", header))
    assert {"ct_hdr_merge_batch", "ct_linearize_std", "ct_linearize_fwd", "ct_linearize_bwd", "ct_pair_residual_fwd",
            "ct_pair_residual_bwd", "ct_pair_residual_bwd_workspace", "ct_abi_version", "ct_error_string"} <= declared
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in clair_hip.h but not exported"
    assert lib.ct_abi_version() == 3
    assert lib.ct_error_string(0) == b"ok" and b"gradient" in lib.ct_error_string(-4)


def test_argument_validation_returns_before_launch(lib):
    from clair_torch_amd import _native as nv
    g = nv.Geometry(channels=3, h_tile=4, width=4, h_global=4, row_offset=0, image_stride=48)
    icrf = nv.Icrf(lut_dev=None, n_points=0, interp=nv.INTERP_NONE)
    fake = ctypes.c_void_p(0x1000)  # never dereferenced: validation fails first
    rc = lib.ct_hdr_merge_batch(None, nv.DTYPE_F32, 1.0, 4, ctypes.byref(g), None, nv.STD_NONE, 0.0, fake,
                                ctypes.byref(icrf), nv.WEIGHT_NONE, None, None, None, fake, None, 3, None)
    assert rc == -1
    lut = nv.Icrf(lut_dev=0x1000, n_points=256, interp=nv.INTERP_LOOKUP)
    rc = lib.ct_hdr_merge_batch(fake, nv.DTYPE_F32, 1.0, 4, ctypes.byref(g), None, nv.STD_CONSTANT, 0.1, fake,
                                ctypes.byref(lut), nv.WEIGHT_NONE, None, None, None, fake, fake, 3, None)
    assert rc == nv.ERR_NO_GRADIENT_PATH  # hdr_merge.py:107-113 raises in the reference
    bad = nv.Geometry(channels=3, h_tile=4, width=4, h_global=2, row_offset=0, image_stride=48)
    rc = lib.ct_linearize_std(fake, nv.DTYPE_F32, 1.0, 1, ctypes.byref(bad), None, nv.STD_NONE, 0.0, ctypes.byref(lut),
                              fake, None, None)
    assert rc == -1
    with pytest.raises(RuntimeError, match="does not require grad"):
        nv.check(nv.ERR_NO_GRADIENT_PATH, "x")
    with pytest.raises(ValueError):
        nv.check(-1, "x")


@pytest.mark.parametrize("max_code", [255, 1023, 4095, 65535])
def test_code_normalisation_constants_are_exact(lib, max_code):
    """fma(u, hi, u*lo) == float32(u) / float32(max_code) for every code (the reference's Normalize)."""
    hi, lo = ctypes.c_float(), ctypes.c_float()
    lib.ct_norm_constants.argtypes = [ctypes.c_float, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    assert lib.ct_norm_constants(float(max_code), ctypes.byref(hi), ctypes.byref(lo)) == 0
    u = np.arange(max_code + 1, dtype=np.float32)
    ref = u / np.float32(max_code)
    got = (u.astype(np.float64) * np.float64(hi.value) + (u * np.float32(lo.value)).astype(np.float64)).astype(np.float32)
    assert np.array_equal(got, ref)
    lib.ct_index_constants.argtypes = [ctypes.c_float, ctypes.c_int, ctypes.POINTER(ctypes.c_float),
                                       ctypes.POINTER(ctypes.c_float)]
    rc = lib.ct_index_constants(float(max_code), 256, ctypes.byref(hi), ctypes.byref(lo))
    if rc == 0:  # the folded LUT coordinate must select the reference's interval for every code
        s_new = (u.astype(np.float64) * np.float64(hi.value) + (u * np.float32(lo.value)).astype(np.float64)).astype(np.float32)
        s_ref = ref * np.float32(255.0)
        assert np.array_equal(np.floor(s_new), np.floor(s_ref))
    assert lib.ct_norm_constants(0.5, ctypes.byref(hi), ctypes.byref(lo)) != 0


@pytest.mark.parametrize("max_code,n_points,ok", [(65535, 256, True), (65535, 2, True), (65535, 4, True), (65535, 16, True),
                                                 (65535, 52, True), (65535, 258, True), (65535, 772, False), (65535, 65536, True),
                                                 (255, 256, True), (255, 2, True), (255, 16, True), (255, 52, True), (255, 86, True),
                                                 (65535, 100, False), (65535, 1000, False), (255, 100, False), (4095, 256, False)])
def test_round_down_fma_interval_constants(lib, max_code, n_points, ok):
    """ct_pivot_floor_constants: the LUT interval the typed-load kernels form from the code held as a float --
    floor(code * r) with r = 1 / step rounded UP (what one FMA under round-toward-minus-infinity against 1.5 * 2^23 leaves
    in the mantissa) -- equals the reference's float32 interval floor(fl(fl(u / max) * (L - 1))) (base.py:166-168) for
    every code; refused when the step is not a whole number of codes, and for L = 772 (step 85), where the reference's own
    two roundings put exact multiples of the step one interval low."""
    r = ctypes.c_float()
    lib.ct_pivot_floor_constants.argtypes = [ctypes.c_float, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
    rc = lib.ct_pivot_floor_constants(float(max_code), n_points, ctypes.byref(r))
    assert (rc == 0) == ok
    u = np.arange(max_code + 1)
    ref = np.floor((u.astype(np.float32) / np.float32(max_code)) * np.float32(n_points - 1))
    if not ok:
        if max_code % (n_points - 1) == 0:  # a whole step, refused because the reference's index is not u // step
            assert not np.array_equal(ref, u // (max_code // (n_points - 1)))
        return
    step = max_code // (n_points - 1)
    assert step * (n_points - 1) == max_code and np.float64(r.value) >= 1.0 / step
    got = np.floor(u.astype(np.float64) * np.float64(r.value))  # exact: 16-bit code times a 24-bit significand
    assert np.array_equal(got, ref) and np.array_equal(got, u // step) and got.max() == n_points - 1


@pytest.mark.parametrize("max_code,n_points,dtype_max", [(65535, 256, 65535), (4095, 256, 65535), (16383, 256, 65535),
                                                         (1023, 256, 65535), (65535, 100, 65535), (65535, 1000, 65535),
                                                         (65535, 772, 65535), (4095, 1024, 65535), (255, 256, 255),
                                                         (255, 100, 255), (255, 64, 255), (255, 1024, 255)])
@pytest.mark.parametrize("lookup", [0, 1])
def test_pivot_interval_constants_every_code(lib, max_code, n_points, dtype_max, lookup):
    """ct_pivot_interval_constants: min(floor(u * scale), last) -- one round-down FMA on the code -- is the reference's
    LINEAR interval / LOOKUP sample for EVERY code the container can hold, also above max_code (12- and 14-bit data in a
    uint16 container, which the reference clamps) and for LUT steps that are not a whole number of codes.  When the
    function says yes the claim is re-checked here in numpy; a refusal only sends the merge to the generic kernel."""
    r = ctypes.c_float()
    lib.ct_pivot_interval_constants.argtypes = [ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
    rc = lib.ct_pivot_interval_constants(float(max_code), n_points, lookup, dtype_max, ctypes.byref(r))
    top = n_points - 1
    u = np.arange(dtype_max + 1)
    s_ref = (u.astype(np.float32) / np.float32(max_code)) * np.float32(top)
    if rc != 0:  # refusals happen where the reference's own two roundings move a code across an entry boundary
        assert (max_code, n_points) not in ((65535, 256), (4095, 256), (16383, 256), (1023, 256), (255, 256))
        return
    got = np.minimum(np.floor(u.astype(np.float64) * np.float64(r.value)), 2 * top if lookup else top)  # exact product
    if lookup:
        assert np.array_equal((got + 1) // 2, np.clip(np.rint(s_ref), 0, top))
    else:
        assert np.array_equal(got, np.floor(np.clip(s_ref, 0, top)))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from clair_torch_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native._build, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_native.NativeLibraryError, match="no CPU fallback"):
        _native.load()


def test_cpu_device_is_refused_loudly():
    from clair_torch_amd.datasets import StackDataset, custom_collate
    from clair_torch_amd.inference import compute_hdr_image, linearize_dataset_generator
    from clair_torch_amd.models import ICRFModelDirect
    from clair_torch_amd import ops
    ds = StackDataset(torch.rand(4, 3, 8, 8), [1, 2, 3, 4])
    loader = DataLoader(ds, batch_size=2, collate_fn=custom_collate)
    with pytest.raises(RuntimeError, match="MI355X"):
        compute_hdr_image(loader, "cpu", ICRFModelDirect())
    with pytest.raises(RuntimeError, match="MI355X"):
        next(iter(linearize_dataset_generator(DataLoader(ds, batch_size=1, collate_fn=custom_collate), "cpu",
                                              ICRFModelDirect())))
    with pytest.raises(RuntimeError, match="cuda"):
        ICRFModelDirect()(torch.rand(1, 3, 4, 4))
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.hdr_merge_batch(torch.rand(2, 3, 4, 4), torch.tensor([1.0, 2.0]))


def test_entry_point_type_errors():
    from clair_torch_amd.common.typecheck import TypeCheckError
    from clair_torch_amd.inference import compute_hdr_image
    from clair_torch_amd.models import ICRFModelDirect
    from clair_torch_amd.common.enums import InterpMode
    with pytest.raises(TypeCheckError):
        compute_hdr_image([1, 2, 3], "cuda")
    with pytest.raises(TypeCheckError):
        ICRFModelDirect(interpolation_mode="linear")
    assert ICRFModelDirect(interpolation_mode=InterpMode.CATMULL).interp_name == "catmull"


def test_custom_collate_semantics():
    """Sorted by exposure time, float64 exposure tensor, std batch None if any std is missing (collate.py:8-43)."""
    from clair_torch_amd.datasets import custom_collate
    items = [(0, torch.full((1, 2, 2), 3.0), torch.ones(1, 2, 2), {"exposure_time": 0.3}),
             (1, torch.full((1, 2, 2), 1.0), torch.ones(1, 2, 2), {"exposure_time": 0.1}),
             (2, torch.full((1, 2, 2), 2.0), None, {"exposure_time": 0.2})]
    idx, val, std, meta = custom_collate(items)
    assert idx.tolist() == [1, 2, 0] and val[:, 0, 0, 0].tolist() == [1.0, 2.0, 3.0]
    assert std is None and meta["exposure_time"].dtype == torch.float64
    _, _, std2, _ = custom_collate(items[:2])
    assert std2.shape == (2, 1, 2, 2)


def test_stack_dataset_std_modes_and_uint16_collation():
    from clair_torch_amd.common.enums import MissingStdMode
    from clair_torch_amd.datasets import StackDataset, custom_collate
    x = torch.rand(3, 3, 4, 4)
    ds = StackDataset(x, [1, 2, 3], missing_std_mode=MissingStdMode.MULTIPLIER, missing_std_value=0.05)
    _, v, s, m = ds[1]
    assert torch.equal(s, v * torch.tensor(0.05)) and m == {"exposure_time": 2.0} and ds.std_hint is None
    ds = StackDataset(x, [1, 2, 3], missing_std_mode=MissingStdMode.CONSTANT, missing_std_value=0.01)
    assert torch.equal(ds[0][2], torch.full_like(x[0], 0.01))
    codes = torch.randint(0, 65535, (3, 3, 4, 4), dtype=torch.int32).to(torch.uint16)
    ds = StackDataset(codes, [3, 1, 2], missing_std_mode=MissingStdMode.MULTIPLIER, missing_std_value=0.05,
                      materialize_std=False)
    assert ds.std_hint == ("multiplier", 0.05) and ds[0][2] is None
    _, val, std, meta = next(iter(DataLoader(ds, batch_size=3, collate_fn=custom_collate)))
    assert val.dtype == torch.uint16 and std is None and meta["exposure_time"].tolist() == [1.0, 2.0, 3.0]
    assert torch.equal(val[0].to(torch.int32), codes[1].to(torch.int32))
    with pytest.raises(ValueError):
        StackDataset(x, [1, 2])


def test_transform_fusion_detection():
    from clair_torch_amd.common.transforms import CastTo, Normalize, fusable_code_normalisation
    codes = torch.zeros((1, 3, 2, 2), dtype=torch.uint16)
    assert fusable_code_normalisation(codes, [CastTo("float32"), Normalize(max_val=65535, min_val=0)]) == 65535.0
    assert fusable_code_normalisation(codes, [CastTo("float32"), Normalize(4095, 0)]) == 4095.0
    assert fusable_code_normalisation(codes, [Normalize(65535, 0)]) is None          # no cast: not the reference's chain
    assert fusable_code_normalisation(codes, [CastTo("float64"), Normalize(65535, 0)]) is None
    assert fusable_code_normalisation(codes, [CastTo("float32"), Normalize(65535, 1)]) is None
    assert fusable_code_normalisation(codes, [CastTo("float32"), Normalize(65535, 0, (0, 2))]) is None
    assert fusable_code_normalisation(codes.float(), [CastTo("float32"), Normalize(65535, 0)]) is None
    x = torch.tensor([0.0, 128.0, 255.0])
    assert torch.equal(Normalize(255, 0)(x), x / 255)
    with pytest.raises(ValueError):
        Normalize()(torch.ones(3))


def test_exposure_pairs_and_partner_lists():
    from clair_torch_amd import ops
    from clair_torch_amd.common.general_functions import get_pairwise_valid_pixel_mask, get_valid_exposure_pairs
    i, j, r = get_valid_exposure_pairs(torch.tensor([1.0, 2.0, 4.0]), 0.4)   # reference test_general_functions.py:290-327
    assert i.tolist() == [0, 1] and j.tolist() == [1, 2] and r.tolist() == [0.5, 0.5]
    i, j, r = get_valid_exposure_pairs(torch.tensor([1.0, 2.0, 4.0]))
    assert i.tolist() == [0, 0, 1] and j.tolist() == [1, 2, 2] and r.tolist() == [0.5, 0.25, 0.5]
    pl = ops.PairList(i, j, r, 3, "cpu")
    assert pl.part_off.tolist() == [0, 2, 4, 6]
    assert pl.part_sample.tolist() == [1, 2, 0, 2, 0, 1]
    assert pl.part_pair.tolist() == [0, 1, ~0, 2, ~1, ~2]
    stack = torch.tensor([0.0, 0.5, 1.0]).view(3, 1, 1, 1)
    m = get_pairwise_valid_pixel_mask(stack, i, j, val_lower=0.1, val_upper=0.9)
    assert m.flatten().tolist() == [False, False, False]
    with pytest.raises(ValueError):
        get_pairwise_valid_pixel_mask(stack, i, j, val_lower=1.0, val_upper=0.0)


def test_penalties_match_eager_oracle_and_gaussian_weights():
    from clair_torch_amd.training import losses
    from oracle import eager_torch as oe
    curve = torch.stack([torch.linspace(0, 1, 32) ** 2, torch.linspace(-0.1, 1.2, 32), torch.linspace(1, 0, 32)])
    mono, rng, endp, smooth = oe.curve_penalties(curve)
    assert torch.equal(losses.compute_monotonicity_penalty(curve, per_channel=True), mono)
    assert torch.equal(losses.compute_range_penalty(curve, per_channel=True), rng)
    assert torch.equal(losses.compute_endpoint_penalty(curve, per_channel=True), endp)
    assert torch.equal(losses.compute_smoothness_penalty(curve, per_channel=True), smooth)
    assert torch.equal(losses.compute_smoothness_penalty(curve), smooth.sum())
    x = torch.linspace(0, 1, 11)
    w = losses.gaussian_value_weights(x)                       # reference test_losses.py:10-60
    assert w[5] == 1.0 and torch.allclose(w, w.flip(0)) and (w <= 1).all() and (w > 0).all()
    assert (losses.gaussian_value_weights(x, 10.0) >= w).all()
    pw = losses.combined_gaussian_pair_weights(x.view(-1, 1), torch.tensor([0, 1]), torch.tensor([2, 3]), 10.0)
    assert torch.equal(pw[0], losses.gaussian_value_weights(x[0:1], 10.0) + losses.gaussian_value_weights(x[2:3], 10.0))


def test_model_state_dict_and_default_curve():
    from clair_torch_amd.models import ICRFModelDirect
    m = ICRFModelDirect(n_points=16, channels=2, initial_power=2.0)
    assert sorted(m.state_dict()) == ["_icrf", "_x_axis_datapoints", "direct_params.0", "direct_params.1"]
    assert torch.equal(m.icrf, (torch.linspace(0, 1, 16) ** 2.0).repeat(2, 1))
    assert m.channel_params(1)[0] is m.direct_params[1]
    m.update_icrf()
    assert m.icrf.requires_grad and m.icrf.shape == (2, 16)
    given = torch.rand(3, 8)
    m2 = ICRFModelDirect(icrf=given.clone())
    assert m2.channels == 3 and m2.n_points == 8 and torch.equal(m2.icrf, given)
    m2.load_state_dict(ICRFModelDirect(icrf=given * 0.5).state_dict())
    assert torch.equal(m2.icrf, given * 0.5)
    assert m.plot_icrf() is None


def test_synthetic_stack_bands_assemble_to_the_whole():
    from clair_torch_amd.datasets import synthetic_exposure_stack
    whole, t = synthetic_exposure_stack(4, 3, 24, 10, bits=16, seed=5)
    parts = [synthetic_exposure_stack(4, 3, 24, 10, bits=16, seed=5, row_range=r)[0] for r in ((0, 7), (7, 24))]
    assert torch.equal(torch.cat(parts, dim=2).to(torch.int32), whole.to(torch.int32))
    assert whole.dtype == torch.uint16 and t == [1e-3 * 2 ** (k * 0.25) for k in range(4)]
    u8, _ = synthetic_exposure_stack(3, 1, 8, 8, bits=8)
    assert u8.dtype == torch.uint8 and 0 < u8.float().mean() < 255


def test_reference_import_paths_resolve():
    import clair_torch  # noqa: F401
    from clair_torch.inference.hdr_merge import compute_hdr_image
    from clair_torch.inference.linearization import linearize_dataset_generator
    from clair_torch.inference.measure_linearity import measure_linearity
    from clair_torch.models.icrf_model import ICRFModelDirect
    from clair_torch.training.icrf_training import train_icrf
    from clair_torch.training.losses import gaussian_value_weights
    from clair_torch.training import pixelwise_linearity_loss, compute_spatial_linearity_loss  # training/__init__.py:6
    from clair_torch.common.general_functions import weighted_mean_and_std, flat_field_mean, flatfield_correction
    from clair_torch.datasets.collate import custom_collate
    assert all(callable(f) for f in (pixelwise_linearity_loss, compute_spatial_linearity_loss, weighted_mean_and_std,
                                     flat_field_mean, flatfield_correction))
    from clair_torch.common.enums import InterpMode
    assert all(callable(f) for f in (compute_hdr_image, linearize_dataset_generator, measure_linearity, train_icrf,
                                     gaussian_value_weights, custom_collate))
    assert ICRFModelDirect and InterpMode.LINEAR


def test_oracle_tiles_equal_whole():
    """The oracle's own tile geometry (used by the sharding tests): bands with global geometry == whole image."""
    from oracle import ct_oracle as oc
    rng = np.random.default_rng(3)
    x = rng.random((4, 3, 9, 7), dtype=np.float32)
    lut = np.stack([np.linspace(0, 1, 32, dtype=np.float32) ** np.float32(p) for p in (1.5, 2.0, 2.5)])
    t = np.array([1.0, 2.0, 4.0, 8.0])
    sd = x * np.float32(0.05)
    mean, std = oc.hdr_merge(x, sd, t, lut, "linear", True)
    for r0, r1 in ((0, 4), (4, 9)):
        m_t, s_t = oc.hdr_merge(np.ascontiguousarray(x[:, :, r0:r1]), np.ascontiguousarray(sd[:, :, r0:r1]), t, lut,
                                "linear", True, tile=(9, r0))
        assert np.array_equal(m_t, mean[:, r0:r1]) and np.array_equal(s_t, std[:, r0:r1])
    m_bad, _ = oc.hdr_merge(np.ascontiguousarray(x[:, :, 4:9]), np.ascontiguousarray(sd[:, :, 4:9]), t, lut, "linear", True)
    assert not np.array_equal(m_bad, mean[:, 4:9])


def test_wbo_accumulators_match_reference_vectors():
    """WBOMean / WBOMeanVar (API utilities) against values recorded from the reference's classes, all three variance
    modes, weighted and unweighted, ragged batches; plus the reference tests' plain-mean property
    (tests/unit/common/test_statistics.py:31-78)."""
    from _util import golden
    from clair_torch_amd.common import VarianceMode, WBOMean, WBOMeanVar
    g = golden("video_stats")
    vals, wts = torch.from_numpy(g["wbv_vals"]), torch.from_numpy(g["wbv_wts"])
    for mode in (VarianceMode.POPULATION, VarianceMode.SAMPLE_FREQUENCY, VarianceMode.RELIABILITY_WEIGHTS):
        for weighted in (True, False):
            h = WBOMeanVar(dim=0, variance_mode=mode)
            k = 0
            for b in (4, 3, 3):
                h.update_values(vals[k:k + b], wts[k:k + b] if weighted else None)
                k += b
            tag = f"wbv_{mode.name.lower()}_{'w' if weighted else 'u'}"
            assert np.allclose(h.mean.numpy(), g[tag + "_mean"], rtol=1e-14, atol=0)
            assert np.allclose(h.variance().numpy(), g[tag + "_var"], rtol=1e-12, atol=0)
    gh = golden("helpers")
    v, w = torch.from_numpy(gh["wbo_vals"]), torch.from_numpy(gh["wbo_wts"])
    for weighted in (True, False):
        h = WBOMean(dim=0)
        k = 0
        for b in (4, 3, 2):
            m = h.update_values(v[k:k + b], w[k:k + b] if weighted else None)
            k += b
        assert np.allclose(m.numpy(), gh[f"wbo_mean_{'w' if weighted else 'u'}"], rtol=1e-14, atol=0)
    h = WBOMean(dim=0)
    data = torch.arange(12, dtype=torch.float64).view(6, 2)
    h.update_values(data[:4])
    assert torch.allclose(h.update_values(data[4:]).squeeze(0), data.mean(dim=0), atol=1e-8)
    with pytest.raises(TypeError):
        WBOMean(dim=(0, 1))


def test_icrf_txt_io_layouts_and_errors(tmp_path):
    """load_icrf_txt / save_icrf_txt: on-disk (L, C) BGR by default <-> in-memory (C, L) RGB; error types as the
    reference's tests expect (tests/unit/common/test_data_io.py:18-66)."""
    from clair_torch_amd.common import ChannelOrder, DimensionOrder, load_icrf_txt, save_icrf_txt
    from clair_torch_amd.common.typecheck import TypeCheckError
    with pytest.raises(FileNotFoundError, match="doesn't exist"):
        load_icrf_txt(tmp_path / "missing.txt")
    with pytest.raises(ValueError, match="Expected a filepath"):
        load_icrf_txt(tmp_path)
    csv = tmp_path / "data.csv"
    csv.write_text("0.0 0.1 0.2")
    with pytest.raises(ValueError, match="Expected .txt filetype"):
        load_icrf_txt(csv)
    bad = tmp_path / "bad.txt"
    bad.write_text("this is not numeric data")
    with pytest.raises(IOError, match="Failed to load NumPy array"):
        load_icrf_txt(bad)
    good = tmp_path / "icrf.txt"
    np.savetxt(good, np.tile([0.1, 0.2, 0.3], (256, 1)).astype(np.float32))
    with pytest.raises(TypeCheckError):
        load_icrf_txt(good, source_channel_order="invalid channel order")
    base = torch.from_numpy(np.tile([0.1, 0.2, 0.3], (256, 1))).float()
    for co in (ChannelOrder.RGB, ChannelOrder.BGR, ChannelOrder.ANY):
        for do in (DimensionOrder.BSC, DimensionOrder.BCS):
            ref = base
            if do == DimensionOrder.BSC:
                ref = ref.transpose(0, 1)
            if co == ChannelOrder.BGR:
                ref = ref[[2, 1, 0], :]
            got = load_icrf_txt(good, source_channel_order=co, source_dimension_order=do)
            assert got.shape == ref.shape and torch.allclose(got, ref)
    curve = torch.stack([torch.linspace(0, 1, 16) ** p for p in (1.5, 2.0, 2.5)])
    out = tmp_path / "saved.txt"
    save_icrf_txt(curve, out)
    on_disk = np.loadtxt(out)
    assert on_disk.shape == (16, 3) and np.allclose(on_disk[:, 0], curve[2].numpy())   # column 0 = blue
    assert torch.allclose(load_icrf_txt(out), curve)
    save_icrf_txt(curve, out, ChannelOrder.RGB, DimensionOrder.BCS)
    assert torch.allclose(load_icrf_txt(out, ChannelOrder.RGB, DimensionOrder.BCS), curve)


def test_custom_collate_returns_views_of_equally_spaced_images():
    """Batches of an in-memory stack alias it (no 2 x stack-size copy per batch); anything irregular is stacked."""
    from clair_torch_amd.datasets import StackDataset, custom_collate
    x = torch.arange(5 * 3 * 4 * 6, dtype=torch.float32).reshape(5, 3, 4, 6)
    ds = StackDataset(x, [1.0, 2.0, 3.0, 4.0, 5.0], stds=0.1 * x)
    _, vals, stds, meta = custom_collate([ds[i] for i in (3, 1, 2)])  # sorted by exposure: images 1, 2, 3
    assert vals.data_ptr() == x[1].data_ptr() and torch.equal(vals, x[1:4])
    assert stds.data_ptr() == ds.stds[1].data_ptr() and stds.stride() == vals.stride()
    assert meta["exposure_time"].tolist() == [2.0, 3.0, 4.0]
    _, vals, _, _ = custom_collate([ds[i] for i in (0, 2, 4)])  # equally spaced: one strided view
    assert vals.data_ptr() == x.data_ptr() and vals.stride(0) == 2 * x.stride(0) and torch.equal(vals, x[0::2])
    _, vals, _, _ = custom_collate([ds[i] for i in (0, 1, 3)])  # irregular spacing: a stacked copy
    assert vals.data_ptr() != x.data_ptr() and torch.equal(vals, x[[0, 1, 3]])
    rev = StackDataset(x, [5.0, 4.0, 3.0, 2.0, 1.0])  # sorting reverses the storage order: a copy
    _, vals, _, _ = custom_collate([rev[i] for i in (0, 1, 2)])
    assert torch.equal(vals, x[[2, 1, 0]])


def test_pair_backward_workspace_contract(lib):
    """ct_pair_residual_bwd validates its caller-owned workspace before anything is launched."""
    from clair_torch_amd import _native as nv
    n, p, c = 6, 9, 3
    need = lib.ct_pair_residual_bwd_workspace(n, p, c)
    assert need >= (n + 1 + c) * 4 + p * c * 32 and need % 32 == 0
    assert lib.ct_pair_residual_bwd_workspace(-1, p, c) == 0
    g = nv.Geometry(channels=c, h_tile=8, width=8, h_global=8, row_offset=0, image_stride=c * 64)
    lut = nv.Icrf(lut_dev=0x1000, n_points=256, interp=nv.INTERP_LINEAR)
    prm = nv.PairParams(lower=0.0, upper=1.0, weight_scale=10.0, use_relative=1, use_uncertainty_weighting=0,
                        std_mode=nv.STD_NONE, std_value=0.0)
    fake = ctypes.c_void_p(0x1000)  # never dereferenced: validation fails first

    def call(ws, ws_bytes):
        return lib.ct_pair_residual_bwd(fake, nv.DTYPE_F32, 1.0, n, ctypes.byref(g), None, ctypes.byref(lut), fake, p,
                                        fake, fake, fake, ctypes.byref(prm), fake, None, fake, ws, ws_bytes, None)

    assert call(None, 0) == -1                              # no workspace
    assert call(ctypes.c_void_p(0x2000), need - 1) == -1    # too small
    assert call(ctypes.c_void_p(0x2004), need) == -1        # not 32-byte aligned
    assert call(ctypes.c_void_p(0x2000), -5) == -1


def test_public_tensor_helpers_match_reference_vectors():
    """The reference's public helper names that the fused kernels replace on the hot path, kept as plain torch fronts:
    weighted_mean_and_std / flat_field_mean / flatfield_correction against vectors recorded from the reference
    (helpers.npz), and pixelwise_linearity_loss -> compute_spatial_linearity_loss chained exactly as train_icrf chains
    them (icrf_training.py:105-133) against the recorded (P, C) spatial statistics of every loss variant (training.npz)."""
    from _util import assert_parity, golden
    from clair_torch_amd.common.general_functions import (flat_field_mean, flatfield_correction,
                                                          get_pairwise_valid_pixel_mask, get_valid_exposure_pairs,
                                                          weighted_mean_and_std)
    from clair_torch_amd.training import (combined_gaussian_pair_weights, compute_spatial_linearity_loss,
                                          pixelwise_linearity_loss)
    from oracle import ct_oracle as oc
    gh = golden("helpers")
    m, sd = weighted_mean_and_std(torch.from_numpy(gh["wms_v"]), weights=torch.from_numpy(gh["wms_w"]),
                                  mask=torch.from_numpy(gh["wms_mask"]), dim=(2, 3))
    assert np.allclose(m.numpy(), gh["wms_mean"], rtol=1e-14, atol=0) and np.allclose(sd.numpy(), gh["wms_std"], rtol=1e-13, atol=0)
    assert float(m[0, 0]) == 0.0 and float(sd[0, 0]) == 0.0                      # fully masked -> zeros
    v = torch.arange(24, dtype=torch.float64).view(2, 3, 4)
    m0, s0 = weighted_mean_and_std(v, dim=2)                                        # reference test_general_functions.py:109-174
    assert torch.allclose(m0, v.mean(dim=2)) and torch.allclose(s0, v.std(dim=2, unbiased=False))
    ff = torch.from_numpy(gh["ff_flat"])
    assert np.array_equal(flat_field_mean(ff, 1.0).numpy(), gh["ff_mean"])
    assert np.array_equal(flat_field_mean(ff, 0.5).numpy(), gh["ff_mean_half"])
    with pytest.raises(ValueError):
        flat_field_mean(ff, 1.5)
    corrected = flatfield_correction(torch.from_numpy(gh["ff_img"]), ff, torch.from_numpy(gh["ff_mean"]))
    assert np.array_equal(corrected.numpy(), gh["ff_corrected"])
    with pytest.raises(ValueError):
        flatfield_correction(torch.zeros(2, 3, 4, 4), torch.zeros(2, 3, 5, 4), torch.zeros(1, 3, 1, 1))

    g = golden("training")
    x = torch.from_numpy(oc.normalize_codes(g["train_codes"]))
    t = torch.from_numpy(g["train_exposures"])
    i, j, r = get_valid_exposure_pairs(t, 0.25)
    mask = get_pairwise_valid_pixel_mask(x, i, j, None, 1 / 255, 254 / 255)
    assert np.array_equal(mask.sum(dim=(2, 3)).numpy(), g["train_none_mask_popcount"])
    gw = combined_gaussian_pair_weights(x, i, j)
    lin, dlin = oc.icrf_forward(x.numpy(), g["train_lut0"], "linear", want_derivative=True)
    lin = torch.from_numpy(lin)
    lin_std = (torch.from_numpy(dlin) * (x * 0.05)).abs()                          # icrf_training.py:117-124 with sigma = 0.05 x
    for sname, stds in (("none", None), ("multiplier", lin_std)):
        for rel in (True, False):
            for unc in (True, False):
                loss, err = pixelwise_linearity_loss(lin, i, j, r, stds, rel)
                assert loss.dtype == torch.float64 and (err is None) == (stds is None)
                sp, sp_std, sp_err = compute_spatial_linearity_loss(loss, err, gw, mask, unc)
                key = f"train_{sname}_linear_{'rel' if rel else 'abs'}_{'unc' if unc else 'nounc'}"
                assert_parity(sp.numpy(), g[key + "_spatial"], rtol=1e-5, norm_tol=2e-6, what=key + " spatial (helpers)")
                assert_parity(sp_std.numpy(), g[key + "_spatial_std"], rtol=2e-5, norm_tol=5e-6, what=key + " std (helpers)")
                if stds is not None:
                    assert_parity(sp_err.numpy(), g[key + "_spatial_err"], rtol=1e-5, norm_tol=2e-6, what=key + " err (helpers)")


def test_alias_package_overlays_a_reference_install(tmp_path, monkeypatch):
    """clair_torch/__init__.py: modules this build does not provide (file settings, parameters, metadata ...) resolve to
    a reference install when one is importable, attributes missing from a provided module (data_io.load_image) are
    taken from the reference's module of the same name, and the reference's own `import clair_torch.<hot path>` lands
    on the MI355X implementation.  The reference install is faked in tmp_path: nothing here reads /root/reference."""
    import importlib
    ref = tmp_path / "site" / "clair_torch"
    (ref / "common").mkdir(parents=True)
    (ref / "metadata").mkdir()
    (ref / "__init__.py").write_text("")
    (ref / "common" / "__init__.py").write_text("raise RuntimeError('the reference common/__init__ must not run: it is aliased')")
    (ref / "common" / "parameters.py").write_text(
        "from clair_torch.common.enums import InterpMode\nfrom clair_torch.models.icrf_model import ICRFModelDirect\n"
        "class Parameters:\n    mode = InterpMode.LINEAR\n    model = ICRFModelDirect\n")
    (ref / "common" / "data_io.py").write_text("def load_image(path):\n    return ('reference load_image', path)\n")
    (ref / "metadata" / "__init__.py").write_text("from clair_torch.metadata.imaging import parse\n")
    (ref / "metadata" / "imaging.py").write_text("def parse(name):\n    return name.split()\n")
    import clair_torch
    for name in [n for n in sys.modules if n.startswith(("clair_torch.metadata", "clair_torch.common.parameters",
                                                          "_clair_torch_reference"))]:
        del sys.modules[name]
    # without a reference install: a clear failure
    with pytest.raises(ImportError):
        importlib.import_module("clair_torch.common.parameters")
    from clair_torch.common import data_io
    with pytest.raises(AttributeError, match="no reference install"):
        data_io.load_image
    monkeypatch.setenv("CLAIR_TORCH_REFERENCE", str(ref))
    importlib.invalidate_caches()
    params = importlib.import_module("clair_torch.common.parameters")
    from clair_torch_amd.common.enums import InterpMode
    from clair_torch_amd.models import ICRFModelDirect
    assert params.Parameters.mode is InterpMode.LINEAR and params.Parameters.model is ICRFModelDirect
    assert importlib.import_module("clair_torch.metadata").parse("10ms 50x") == ["10ms", "50x"]
    assert data_io.load_image("x.tif") == ("reference load_image", "x.tif")
    assert callable(data_io.load_icrf_txt) and data_io.load_icrf_txt.__module__.startswith("clair_torch_amd")
    for name in [n for n in sys.modules if n.startswith(("clair_torch.metadata", "clair_torch.common.parameters",
                                                          "_clair_torch_reference"))]:
        del sys.modules[name]


def test_torch_library_ops_are_registered_with_schemas_and_fake_kernels():
    """the entry points are torch.library custom ops (clair_hip::*): schemas, fake (meta) kernels that
    propagate shapes / dtypes under FakeTensorMode (what torch.compile / export trace through), and the loud refusal
    of CPU tensors."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from clair_torch_amd import torch_ops  # noqa: F401  (registers the ops)
    names = ["icrf_forward", "icrf_backward", "hdr_merge", "linearize_std", "pair_residual_sums", "pair_residual_lut_grad",
             "band_stats"]
    for n in names:
        assert hasattr(torch.ops.clair_hip, n), n
    assert "Tensor? lut" in str(torch.ops.clair_hip.hdr_merge.default._schema)
    with FakeTensorMode():
        lut = torch.empty((3, 256))
        y = torch.ops.clair_hip.icrf_forward(torch.empty((2, 3, 8, 9)), lut, "linear")
        assert tuple(y.shape) == (2, 3, 8, 9) and y.dtype == torch.float32
        m, s = torch.ops.clair_hip.hdr_merge(torch.empty((4, 8, 9, 3), dtype=torch.uint16), torch.empty(4, dtype=torch.float64),
                                             lut, "linear", True, None, "multiplier", 0.05, 65535.0, 0, 0, "nhwc_bgr")
        assert tuple(m.shape) == (3, 8, 9) and m.dtype == torch.float64 and s.dtype == torch.float32
        m2, s2 = torch.ops.clair_hip.hdr_merge(torch.empty((4, 3, 8, 9)), torch.empty(4, dtype=torch.float64), None, "linear",
                                               False, None, "none", 0.0, 0.0)
        assert tuple(m2.shape) == (3, 8, 9) and s2.numel() == 0
        lin, sd = torch.ops.clair_hip.linearize_std(torch.empty((5, 3, 8, 9), dtype=torch.uint8), lut, "linear", None,
                                                    "multiplier", 0.05, 255.0)
        assert tuple(lin.shape) == (5, 3, 8, 9) and sd.shape == lin.shape
        sums = torch.ops.clair_hip.pair_residual_sums(torch.empty((6, 3, 8, 9)), torch.empty(7, dtype=torch.int64),
                                                      torch.empty(7, dtype=torch.int64), torch.empty(7, dtype=torch.float64), lut,
                                                      "linear", 0.0, 1.0, True, False, "none", 0.0, 0.0, 1)
        assert tuple(sums.shape) == (7, 3, 5) and sums.dtype == torch.float64
        st = torch.ops.clair_hip.band_stats(torch.empty((3, 8, 9), dtype=torch.float64), torch.empty((3, 8, 9)))
        assert tuple(st.shape) == (6, 3) and st.dtype == torch.float64
    with pytest.raises(RuntimeError, match="MI355X"):
        torch.ops.clair_hip.icrf_forward(torch.zeros((1, 3, 4, 4)), torch.zeros((3, 16)), "linear")
