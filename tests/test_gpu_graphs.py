"""The C-ABI entry points only enqueue work on the stream they are given (no allocation, no synchronisation), so whole
sequences of them can be captured into a HIP graph and replayed: merge, linearize and one training forward+backward.
The replay must reproduce the eager results bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from clair_torch_amd import _native
    _native.load()
    return torch.device("cuda:0")


def _stack(dev, n=6, h=48, w=64, seed=3):
    from clair_torch_amd.datasets import synthetic_exposure_stack
    codes, exposures = synthetic_exposure_stack(n, 3, h, w, bits=16, stops_per_step=0.5, seed=seed, device=dev)
    return codes, torch.tensor(exposures, dtype=torch.float64, device=dev)


def test_merge_and_linearize_replay_from_a_graph(dev):
    from clair_torch_amd import ops
    codes, t = _stack(dev)
    lut = torch.stack([torch.linspace(0, 1, 256, device=dev) ** p for p in (2.2, 2.4, 2.6)]).contiguous()
    kw = dict(lut=lut, interp="linear", gaussian_weight=True, std_mode="multiplier", std_value=0.05)
    eager_mean, eager_std = ops.hdr_merge_batch(codes, t, **kw)
    eager_lin, eager_lsd = ops.linearize_frames(codes, lut, "linear", std_mode="multiplier", std_value=0.05)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        ops.hdr_merge_batch(codes, t, **kw)  # warm-up on the capture stream (constant caches, allocator pools)
        ops.linearize_frames(codes, lut, "linear", std_mode="multiplier", std_value=0.05)
    side.synchronize()
    with torch.cuda.graph(graph, stream=side):
        g_mean, g_std = ops.hdr_merge_batch(codes, t, **kw)
        g_lin, g_lsd = ops.linearize_frames(codes, lut, "linear", std_mode="multiplier", std_value=0.05)
    for out in (g_mean, g_std, g_lin, g_lsd):
        out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(g_mean, eager_mean) and torch.equal(g_std, eager_std)
    assert torch.equal(g_lin, eager_lin) and torch.equal(g_lsd, eager_lsd)
    # new data in the same buffers, replay again
    codes2, _ = _stack(dev, seed=11)
    codes.copy_(codes2)
    ref_mean, ref_std = ops.hdr_merge_batch(codes, t, **kw)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(g_mean, ref_mean) and torch.equal(g_std, ref_std)


def test_training_kernels_replay_from_a_graph(dev):
    from clair_torch_amd import ops
    from clair_torch_amd.common.general_functions import get_valid_exposure_pairs
    codes, t = _stack(dev, n=8)
    i, j, r = get_valid_exposure_pairs(t.cpu(), 0.1)
    pairs = ops.PairList(i, j, r, 8, dev)
    lut = torch.stack([torch.linspace(0, 1, 256, device=dev) ** p for p in (2.2, 2.4, 2.6)]).contiguous()
    kw = dict(lut=lut, interp="linear", lower=1 / 255, upper=254 / 255, use_relative=True, use_unc_weight=False)
    coef = torch.full((pairs.n_pairs, 3), 1e-5, dtype=torch.float64, device=dev)
    eager_sums = ops.pair_residual_sums(codes, pairs, **kw)
    eager_grad = ops.pair_residual_lut_grad(codes, pairs, coef, **kw)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        ops.pair_residual_sums(codes, pairs, **kw)
        ops.pair_residual_lut_grad(codes, pairs, coef, **kw)
    side.synchronize()
    with torch.cuda.graph(graph, stream=side):
        g_sums = ops.pair_residual_sums(codes, pairs, **kw)
        g_grad = ops.pair_residual_lut_grad(codes, pairs, coef, **kw)
    graph.replay()
    torch.cuda.synchronize()
    # the sums and the gradient are accumulated with float64 atomics whose order is not fixed: equal to rounding
    assert torch.allclose(g_sums, eager_sums, rtol=1e-12, atol=0)
    assert torch.allclose(g_grad, eager_grad, rtol=1e-9, atol=1e-18)
