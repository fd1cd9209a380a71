"""Shared helpers for the parity tests: golden-fixture access and the parity criteria."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PARTITIONS = {"8": [8], "44": [4, 4], "332": [3, 3, 2]}


def sampler_batches(sampler, exposures):
    """Batches of a recorded batch_sampler (rows padded with -1), each sorted by exposure time like custom_collate
    (clair_torch/datasets/collate.py:23; Python's sort is stable)."""
    out = []
    for row in np.asarray(sampler):
        idx = [int(i) for i in row if i >= 0]
        out.append(sorted(idx, key=lambda i: exposures[i]))
    return out


def golden(name):
    return np.load(os.path.join(GOLDEN, f"{name}.npz"))


def std_for(mode, x, explicit):
    """Rebuild the std stack of a golden merge/linearize case from its mode name (make_golden.make_stds)."""
    if mode == "none":
        return None
    if mode == "constant":
        return np.full_like(x, np.float32(0.01))
    if mode == "multiplier":
        return x * np.float32(0.05)
    if mode == "explicit":
        return explicit
    raise ValueError(mode)


def rel_norm(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


OBSERVED = []  # (what, norm-wise error, worst element error in units of the element tolerance, tolerances): conftest dumps it


def assert_parity(got, ref, rtol=1e-5, norm_tol=1e-5, elem_tol=None, what=""):
    """SURVEY 8(d) parity criterion: norm-wise relative error <= norm_tol and
    allclose(rtol, atol = rtol * median|ref|) element-wise (elem_tol overrides rtol for the element test).
    Every call records what it observed (OBSERVED; written to parity_observed.json at the end of the session), so the
    share of each tolerance a kernel actually uses is on record."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    assert np.all(np.isfinite(got) == np.isfinite(ref)), f"{what}: finiteness differs"
    fin = np.isfinite(ref)
    got, ref = got[fin], ref[fin]
    if ref.size == 0:
        return
    rn = rel_norm(got, ref)
    et = rtol if elem_tol is None else elem_tol
    med = float(np.median(np.abs(ref)))
    atol = et * med
    # worst element error expressed as the rtol it would need under the same "rtol |ref| + rtol median" form
    worst = float(np.max(np.abs(got - ref) / (np.abs(ref) + med))) if med > 0 or np.any(ref != 0) else float(np.max(np.abs(got - ref)))
    test = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0].split("::")[-1].split("[")[0]
    OBSERVED.append({"test": test, "what": what, "norm": rn, "norm_tol": norm_tol, "elem": worst, "elem_tol": et, "n": int(ref.size)})
    assert rn <= norm_tol, f"{what}: norm-wise rel error {rn:.3e} > {norm_tol:.1e}"
    bad = np.abs(got - ref) > et * np.abs(ref) + atol
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{ref.size} elements outside rtol={et:.1e}; "
                           f"worst {np.max(np.abs(got - ref) / (np.abs(ref) + atol)):.3e}")
