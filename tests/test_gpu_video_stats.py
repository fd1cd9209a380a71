"""GPU parity of the streaming video mean / std (ct_video_stats_batch)."""
import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

from _util import assert_parity, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from clair_torch_amd import _native
    _native.load()
    return torch.device("cuda:0")


@pytest.mark.parametrize("mname", ["nomodel", "linear", "catmull"])
@pytest.mark.parametrize("bname,bs", [("b4", 4), ("b11", 11), ("b1", 1)])
@pytest.mark.parametrize("as_codes", [True, False])
def test_compute_video_mean_and_std(dev, mname, bname, bs, as_codes):
    from clair_torch_amd.common.enums import InterpMode
    from clair_torch_amd.common.transforms import CastTo, Normalize
    from clair_torch_amd.datasets import StackDataset, custom_collate
    from clair_torch_amd.inference import compute_video_mean_and_std
    from clair_torch_amd.models import ICRFModelDirect
    from oracle import ct_oracle as oc
    g = golden("video_stats")
    codes = g["vid_codes"]
    frames = torch.from_numpy(codes if as_codes else oc.normalize_codes(codes))
    ds = StackDataset(frames, [1.0] * frames.shape[0])
    loader = DataLoader(ds, batch_size=bs, shuffle=False, collate_fn=custom_collate)
    model = None if mname == "nomodel" else ICRFModelDirect(icrf=torch.from_numpy(g["vid_lut"]),
                                                            interpolation_mode=InterpMode[mname.upper()]).to(dev)
    tf = [CastTo("float32"), Normalize(255, 0)] if as_codes else None
    mean, std = compute_video_mean_and_std(loader, "cuda", model, gpu_transforms=tf)
    assert mean.dtype == torch.float32 and std.dtype == torch.float32 and mean.shape == (3, 9, 14)
    assert_parity(mean.cpu().numpy(), g[f"vid_{mname}_{bname}_mean"], rtol=1e-6, norm_tol=1e-7, what="video mean")
    assert_parity(std.cpu().numpy(), g[f"vid_{mname}_{bname}_std"], rtol=1e-5, norm_tol=1e-6, what="video std")


def test_video_stats_large_vs_eager_oracle(dev):
    """1080p-sized uint16 frames, 3 ragged batches, against the eager oracle."""
    from clair_torch_amd import ops
    from oracle import ct_oracle as oc
    from oracle import eager_torch as oe
    rng = np.random.default_rng(12)
    codes = rng.integers(20000, 40000, size=(7, 3, 270, 481)).astype(np.uint16)
    lut = torch.stack([torch.linspace(0, 1, 256) ** p for p in (2.2, 2.4, 2.6)])
    x = torch.from_numpy(oc.normalize_codes(codes))
    mean_o, std_o = oe.video_mean_std(x, lut, "linear", [3, 3, 1])
    mean = torch.empty((3, 270, 481), dtype=torch.float32, device=dev)
    m2 = torch.empty_like(mean)
    k = 0
    for b in (3, 3, 1):
        ops.video_stats_batch(torch.from_numpy(codes[k:k + b]).to(dev), mean, m2, k, lut=lut.to(dev), interp="linear")
        k += b
    std = torch.sqrt(m2 / (k - 1)) / k ** 0.5
    assert_parity(mean.cpu().numpy(), mean_o.numpy(), rtol=1e-6, norm_tol=1e-7, what="mean")
    assert_parity(std.cpu().numpy(), std_o.numpy(), rtol=1e-4, norm_tol=1e-6, what="std")


@pytest.mark.parametrize("mode", ["linear", "catmull"])
def test_video_stats_codes_above_max_code(dev, mode):
    """uint16 codes above max_code = 4095: the in-kernel linearization clamps like the reference's model."""
    from clair_torch_amd import ops
    from oracle import eager_torch as oe
    rng = np.random.default_rng(41)
    codes = rng.integers(0, 4096, size=(6, 3, 10, 14)).astype(np.uint16)
    codes.reshape(-1)[::4] = rng.integers(4096, 7000, size=codes.reshape(-1)[::4].shape).astype(np.uint16)
    lut = torch.stack([torch.linspace(0, 1, 256) ** p for p in (2.2, 2.4, 2.6)])
    x = torch.from_numpy((codes.astype(np.float32) / np.float32(4095.0)).astype(np.float32))
    mean_o, std_o = oe.video_mean_std(x, lut, mode, [4, 2])
    mean = torch.empty((3, 10, 14), dtype=torch.float32, device=dev)
    m2 = torch.empty_like(mean)
    k = 0
    for b in (4, 2):
        ops.video_stats_batch(torch.from_numpy(codes[k:k + b]).to(dev), mean, m2, k, lut=lut.to(dev), interp=mode,
                              max_code=4095.0)
        k += b
    std = torch.sqrt(m2 / (k - 1)) / k ** 0.5
    assert_parity(mean.cpu().numpy(), mean_o.numpy(), rtol=1e-6, norm_tol=1e-7, what="max_code 4095 video mean")
    assert_parity(std.cpu().numpy(), std_o.numpy(), rtol=1e-4, norm_tol=1e-6, what="max_code 4095 video std")
