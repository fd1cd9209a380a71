"""world_size = 2 on CPU (gloo): the sharded (row-band) form of the training statistics is exact.

The HIP kernels need a GPU, so here the two ops the distributed code calls are replaced by CPU stand-ins built on the
oracle (tests may use the oracle as the checker); what is under test is the product's host logic in
clair_torch_amd/training/linearity.py: per-band sums -> all_reduce -> identical loss on every rank -> per-band LUT
gradient -> all_reduce, and bench.py's max-over-ranks / stats gather.  Bands are deliberately ragged and
(rows * W) % C != 0 so the global-geometry handling of the LUT-row quirk is exercised.
"""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _scene():
    gen = torch.Generator().manual_seed(33)
    n, c, h, w = 5, 3, 29, 10
    t = torch.tensor([0.002 * 2.0 ** k for k in range(n)], dtype=torch.float64)
    e = torch.rand((c, h, w), generator=gen, dtype=torch.float64) * (2.0 / float(torch.sqrt(t[0] * t[-1])))
    x = ((e.unsqueeze(0) * t.view(-1, 1, 1, 1)).clamp(0, 1) ** (1 / 2.2)).float()
    lut = torch.stack([torch.linspace(0, 1, 48) ** p for p in (1.8, 2.2, 2.6)])
    return x, t, lut


def _install_stand_ins(whole_x):
    from clair_torch_amd import ops
    from oracle import ct_oracle as oc
    from oracle import eager_torch as oe

    def fake_sums(stack, pairs, *, lut, interp, lower, upper, use_relative, use_unc_weight, std=None, std_mode="none",
                  std_value=0.0, max_code=None, level=1, tile=None, center=None):
        geom = None if tile is None else (tile.h_global, tile.row_offset)
        x = stack.numpy()
        lin = oc.icrf_forward(x, lut.detach().numpy(), interp, tile=geom)
        sums = oc.pair_sums(lin, x, None, pairs.i.numpy(), pairs.j.numpy(), pairs.ratio.numpy(), lower, upper,
                            use_relative, use_unc_weight)
        return torch.from_numpy(np.ascontiguousarray(sums[..., :5]))

    @torch.enable_grad()  # called from inside autograd.Function.backward, where grad mode is off
    def fake_grad(stack, pairs, coef, *, lut, interp, lower, upper, use_relative, max_code=None, tile=None, **_unused):
        lut_t = lut.detach().clone().requires_grad_(True)
        r0 = 0 if tile is None else tile.row_offset
        lin = oe.icrf_forward(whole_x, lut_t, interp)[:, :, r0:r0 + stack.shape[2]]  # rows picked on the whole image
        i, j = pairs.i.long(), pairs.j.long()
        xi, xj = stack[i], stack[j]
        m = ((xi >= lower) & (xi <= upper) & (xj >= lower) & (xj <= upper)).double()
        gw = (oe.gaussian_weight(xi, 10.0) + oe.gaussian_weight(xj, 10.0)).double()
        expected = lin[j] * pairs.ratio.view(-1, 1, 1, 1)
        diff = lin[i] - expected
        if use_relative:
            diff = diff / (expected + 1e-6)
        s1 = (diff.abs() * m * gw).sum(dim=(2, 3))
        return torch.autograd.grad((coef * s1).sum(), lut_t)[0].double()

    ops.pair_residual_sums = fake_sums
    ops.pair_residual_lut_grad = fake_grad


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from clair_torch_amd import ops
        from clair_torch_amd.common.general_functions import get_valid_exposure_pairs
        from clair_torch_amd.training import linearity_loss
        import bench
        x, t, lut0 = _scene()
        _install_stand_ins(x)
        i, j, r = get_valid_exposure_pairs(t, 0.2)
        pairs = ops.PairList(i, j, r, x.shape[0], "cpu")
        bands = [(0, 13), (13, 29)]
        r0, r1 = bands[rank]
        kw = dict(interp="linear", lower=1 / 255, upper=254 / 255, use_relative=True, use_unc_weight=False)
        # sharded: each rank sees only its band, statistics all-reduced over the default group
        lut = lut0.clone().requires_grad_(True)
        lin, spatial = linearity_loss(lut, x[:, :, r0:r1].contiguous(), pairs,
                                      tile=ops.TileGeometry(h_global=x.shape[2], row_offset=r0), **kw)
        grad = torch.autograd.grad(lin.sum(), lut)[0]
        # unsharded reference in the same process (group=False suppresses the collectives)
        lut_w = lut0.clone().requires_grad_(True)
        lin_w, spatial_w = linearity_loss(lut_w, x, pairs, group=False, **kw)
        grad_w = torch.autograd.grad(lin_w.sum(), lut_w)[0]
        assert torch.allclose(lin, lin_w, rtol=1e-12, atol=0), (lin, lin_w)
        assert torch.allclose(spatial, spatial_w, rtol=1e-12, atol=1e-300)
        assert torch.allclose(grad, grad_w, rtol=1e-5, atol=1e-9), (grad - grad_w).abs().max()
        # every rank holds the same loss and gradient (identical Adam steps follow)
        both = [torch.empty_like(grad) for _ in range(world)]
        dist.all_gather(both, grad.contiguous())
        assert torch.equal(both[0], both[1])
        # and the eager oracle on the whole image agrees
        from oracle import eager_torch as oe
        lo = lut0.clone().requires_grad_(True)
        _, lin_o, _ = oe.training_loss(x, None, t, lo, "linear", 0.2, 1 / 255, 254 / 255, True, False)
        assert torch.allclose(lin, lin_o.detach(), rtol=1e-6)
        assert torch.allclose(grad, torch.autograd.grad(lin_o.sum(), lo)[0], rtol=1e-4, atol=1e-8)
        # bench.py's collectives: max over ranks and the per-band stats gather
        assert bench.max_over_ranks(1.0 + rank, world, "cpu") == 2.0
        g = bench.gather_stats(torch.full((6, 3), float(rank), dtype=torch.float64), world)
        assert g.shape == (2, 6, 3) and g[1].eq(1).all() and g[0].eq(0).all()
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_sharded_linearity_statistics_two_ranks(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
