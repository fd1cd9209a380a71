"""ctypes front-end of oracle/ct_oracle.c (closed-form CPU restatement) -- TEST INFRASTRUCTURE ONLY.

See the header of ct_oracle.c for scope and citations.  Parity status: pinned by tests/test_oracle_golden.py
against vectors recorded from the reference (tests/golden/make_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libct_oracle.so")
MODES = {"lookup": 0, "linear": 1, "catmull": 2, "nomodel": 3}
_lib = None


def build(force=False):
    """Compile the C oracle with gcc (no-op when up to date)."""
    src = os.path.join(_HERE, "ct_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s", "libct_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        i64, i32, f32, f64, vp = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_void_p
        L.cto_normalize_u8.argtypes = [vp, i64, vp]
        L.cto_normalize_u16.argtypes = [vp, i64, vp]
        L.cto_icrf_forward.argtypes = [vp, i64, i32, i64, i64, vp, i32, i32, vp, vp, i64, i64]
        L.cto_hdr_merge_batch.argtypes = [vp, vp, vp, i64, i32, i64, i64, vp, i32, i32, i32, vp, vp, vp, i32, i64, i64]
        L.cto_hdr_merge_batch.restype = i32
        L.cto_linearize_std.argtypes = [vp, vp, i64, i32, i64, i64, vp, i32, i32, vp, vp, i64, i64]
        L.cto_linearize_std.restype = i32
        L.cto_flatfield_merge.argtypes = [vp, vp, vp, vp, i32, i64]
        L.cto_flatfield_linearize.argtypes = [vp, vp, vp, vp, i64, i32, i64]
        L.cto_pair_sums.argtypes = [vp, vp, vp, i64, i32, i64, i64, vp, vp, vp, i64, f32, f32, i32, i32, vp]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def normalize_codes(u):
    """uint8/uint16 codes -> float32 in [0,1] exactly as Normalize(0, maxcode) on a float32 tensor."""
    u = np.ascontiguousarray(u)
    x = np.empty(u.shape, dtype=np.float32)
    if u.dtype == np.uint8:
        lib().cto_normalize_u8(_p(u), u.size, _p(x))
    elif u.dtype == np.uint16:
        lib().cto_normalize_u16(_p(u), u.size, _p(x))
    else:
        raise TypeError(u.dtype)
    return x


def _geom(tile):
    """tile = None (whole image) or (h_global, row_offset) for a band of rows of a taller image."""
    return (0, 0) if tile is None else (int(tile[0]), int(tile[1]))


def icrf_forward(x, lut, mode, want_derivative=False, tile=None):
    x = _f32(x)
    n, c, h, w = x.shape
    lut = _f32(lut)
    out = np.empty_like(x)
    dout = np.empty_like(x) if want_derivative else None
    lib().cto_icrf_forward(_p(x), n, c, h, w, _p(lut), lut.shape[1], MODES[mode], _p(out), _p(dout), *_geom(tile))
    return (out, dout) if want_derivative else out


class MergeState:
    """Running (mean f64, sum-of-weights f32, variance f32) of the streaming merge."""

    def __init__(self, c, h, w):
        self.shape = (c, h, w)
        self.mean = np.zeros(self.shape, dtype=np.float64)
        self.sumw = np.zeros(self.shape, dtype=np.float32)
        self.var = np.zeros(self.shape, dtype=np.float32)
        self.first = True
        self.has_var = False


def hdr_merge_batch(state, x, sd, exposures, lut, mode, use_gauss, tile=None):
    x = _f32(x)
    sd = _f32(sd)
    b, c, h, w = x.shape
    t = np.ascontiguousarray(exposures, dtype=np.float64)
    lut_c = _f32(lut)
    L = 0 if lut_c is None else lut_c.shape[1]
    m = MODES["nomodel"] if lut_c is None else MODES[mode]
    rc = lib().cto_hdr_merge_batch(_p(x), _p(sd), _p(t), b, c, h, w, _p(lut_c), L, m, int(bool(use_gauss)),
                                   _p(state.mean), _p(state.sumw), _p(state.var), int(state.first), *_geom(tile))
    if rc != 0:
        raise RuntimeError("element 0 of tensors does not require grad and does not have a grad_fn")
    state.first = False
    state.has_var = state.has_var or sd is not None
    return state


def hdr_merge(x, sd, exposures, lut, mode="linear", use_gauss=True, partition=None, tile=None):
    """Whole compute_hdr_image for an in-memory stack; ``partition`` = batch sizes (default one batch)."""
    n, c, h, w = x.shape
    partition = [n] if partition is None else list(partition)
    st, k = MergeState(c, h, w), 0
    for b in partition:
        hdr_merge_batch(st, x[k:k + b], None if sd is None else sd[k:k + b], exposures[k:k + b], lut, mode, use_gauss, tile)
        k += b
    return st.mean, (np.sqrt(st.var) if st.has_var else None)


def linearize_std(x, sd, lut, mode="linear", tile=None):
    x = _f32(x)
    sd = _f32(sd)
    f, c, h, w = x.shape
    lut = _f32(lut)
    lin, so = np.empty_like(x), np.empty_like(x)
    rc = lib().cto_linearize_std(_p(x), _p(sd), f, c, h, w, _p(lut), lut.shape[1], MODES[mode], _p(lin), _p(so), *_geom(tile))
    if rc != 0:
        raise RuntimeError("element 0 of tensors does not require grad and does not have a grad_fn")
    return lin, so


def exposure_pairs(exposures, threshold):
    """get_valid_exposure_pairs (general_functions.py:242-272): all i<j in triu order with t_i/t_j >= thr."""
    t = np.asarray(exposures, dtype=np.float64)
    n = t.shape[0]
    i, j = np.triu_indices(n, k=1)
    r = t[i] / t[j]
    if threshold is not None:
        keep = r >= threshold
        i, j, r = i[keep], j[keep], r[keep]
    return i.astype(np.int64), j.astype(np.int64), r


def pair_sums(lin, x, lsd, i_idx, j_idx, ratio, lo, hi, use_relative, use_unc_weight):
    lin, x, lsd = _f32(lin), _f32(x), _f32(lsd)
    n, c, h, w = x.shape
    i_idx = np.ascontiguousarray(i_idx, dtype=np.int64)
    j_idx = np.ascontiguousarray(j_idx, dtype=np.int64)
    ratio = np.ascontiguousarray(ratio, dtype=np.float64)
    sums = np.zeros((len(ratio), c, 6), dtype=np.float64)
    lib().cto_pair_sums(_p(lin), _p(x), _p(lsd), n, c, h, w, _p(i_idx), _p(j_idx), _p(ratio), len(ratio),
                        np.float32(lo), np.float32(hi), int(use_relative), int(use_unc_weight), _p(sums))
    return sums


def spatial_stats(sums, have_err):
    """(P,C,6) sums -> (spatial mean, spatial std, spatial error|None), general_functions.py:149-170."""
    s0, s1, s2, s3, s4 = (sums[..., k] for k in range(5))
    den = np.maximum(s0, 1e-8)
    mean = s1 / den
    var = (s2 - 2.0 * mean * s1 + mean * mean * s0) / den
    std = np.sqrt(np.maximum(var, 0.0))
    err = s3 / np.maximum(s4, 1e-8) if have_err else None
    return mean, std, err


def flatfield_merge(mean, std, flat, flat_std):
    """Flat-field epilogue of compute_hdr_image on a merged (mean f64, std f32): returns corrected (mean, std)."""
    c = mean.shape[0]
    p = int(np.prod(mean.shape[1:]))
    m = np.ascontiguousarray(mean, dtype=np.float64).copy()
    var = (np.ascontiguousarray(std, dtype=np.float32) ** 2).astype(np.float32)
    flat, flat_std = _f32(flat), _f32(flat_std)
    lib().cto_flatfield_merge(_p(m), _p(var), _p(flat), _p(flat_std), c, p)
    return m, np.sqrt(var)


def flatfield_linearize(lin, std, flat, flat_std):
    """Flat-field epilogue of linearize_dataset_generator for (F,C,H,W) frames: corrected (lin, std) float32."""
    f, c = lin.shape[:2]
    p = int(np.prod(lin.shape[2:]))
    lo, so = _f32(lin).copy(), _f32(std).copy()
    flat, flat_std = _f32(flat), _f32(flat_std)
    lib().cto_flatfield_linearize(_p(lo), _p(so), _p(flat), _p(flat_std), f, c, p)
    return lo, so
