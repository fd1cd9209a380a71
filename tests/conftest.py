import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    from oracle import ct_oracle
    ct_oracle.build()


def pytest_sessionfinish(session, exitstatus):
    """Write the parity errors every assert_parity call observed (worst per label) next to the other GPU-run artefacts."""
    try:
        import json
        from _util import OBSERVED
        if not OBSERVED:
            return
        worst = {}
        for o in OBSERVED:
            w = worst.setdefault((o["test"], o["what"]), dict(o, calls=0))
            w["calls"] += 1
            w["norm"], w["elem"] = max(w["norm"], o["norm"]), max(w["elem"], o["elem"])
        out_dir = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        import torch
        name = "parity_observed.json" if torch.cuda.is_available() else "parity_observed_cpu.json"
        with open(os.path.join(out_dir, name), "w") as fh:
            json.dump(sorted(worst.values(), key=lambda o: -o["elem"] / o["elem_tol"]), fh, indent=1)
    except Exception as exc:  # never fail a run over the report
        print(f"parity report not written: {exc}", file=sys.stderr)
