"""Dark-field correction: ct_dark_field_blur + the unchanged merge / linearize kernels against the
eager restatement of the reference's op sequence (oracle/eager_torch.py).

PARITY UNPINNED: the 3x3 blur is torchvision's GaussianBlur(3, sigma=1) -- not vendored by the reference, absent from
this image, covered by no vector recorded from the reference -- so the comparand restates its published algorithm and
these tests pin the kernels to that restatement only."""
import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

from _util import assert_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from clair_torch_amd import _native
    _native.load()
    return torch.device("cuda:0")


def _scene(seed, n, c, h, w):
    gen = torch.Generator().manual_seed(seed)
    t = torch.tensor([0.002 * 2.0 ** k for k in range(n)], dtype=torch.float64)
    e = torch.rand((c, h, w), generator=gen, dtype=torch.float64) * (2.0 / float(torch.sqrt(t[0] * t[-1])))
    x = ((e.unsqueeze(0) * t.view(-1, 1, 1, 1)).clamp(0, 1) ** (1 / 2.2)).float()
    sd = (0.002 + 0.03 * torch.rand(x.shape, generator=gen)).float()
    dark = (0.1 * torch.rand((c, h, w), generator=gen)).float()          # straddles the 0.05 threshold
    dark_std = (0.002 + 0.01 * torch.rand((c, h, w), generator=gen)).float()
    lut = torch.stack([torch.linspace(0, 1, 256) ** p for p in (1.8, 2.2, 2.6)])
    return x, sd, t, dark, dark_std, lut


def test_blur_kernel_against_restated_torchvision(dev):
    """xb and the effective uncertainty, float32 pixels and raw codes, shared and per-frame dark fields."""
    from clair_torch_amd import ops
    from oracle import ct_oracle as oc
    from oracle import eager_torch as oe
    x, sd, _, dark, dark_std, _ = _scene(3, 4, 3, 13, 9)
    ref = oe.conditional_gaussian_blur(x, dark.unsqueeze(0))
    nb = x.shape[0]   # one matched dark field per frame (get_matching_artefact_images collates B copies)
    xb, sig = ops.dark_field_blur(x.to(dev), dark.expand(nb, *dark.shape).to(dev), dark_std.expand(nb, *dark.shape).to(dev), std=sd.to(dev))
    # ONE shared dark field for several frames with its uncertainty is a different quantity in the reference
    # ((sum_n g_n)^2 sigma_D^2): refused, not approximated
    with pytest.raises(NotImplementedError, match="shared dark field"):
        ops.dark_field_blur(x.to(dev), dark.unsqueeze(0).to(dev), dark_std.unsqueeze(0).to(dev), std=sd.to(dev))
    xb_shared, none_sig = ops.dark_field_blur(x.to(dev), dark.unsqueeze(0).to(dev), None, std=None)   # without a dark std: fine
    assert none_sig is None and torch.equal(xb_shared, xb)
    assert_parity(xb.cpu().numpy(), ref.numpy(), rtol=1e-6, norm_tol=1e-7, what="blurred batch")
    m = torch.sigmoid((dark - 0.05) * 50.0)
    dterm = (oe.gaussian_blur3(x) - x) * (50.0 * m * (1 - m))
    sig_ref = torch.sqrt(sd ** 2 + (dterm * dark_std) ** 2)
    assert_parity(sig.cpu().numpy(), sig_ref.numpy(), rtol=1e-5, norm_tol=1e-6, what="effective sigma")
    per_frame = torch.stack([dark * (0.5 + 0.25 * k) for k in range(4)])
    ref2 = oe.conditional_gaussian_blur(x, per_frame)
    xb2, _ = ops.dark_field_blur(x.to(dev), per_frame.to(dev), None, std=None)
    assert_parity(xb2.cpu().numpy(), ref2.numpy(), rtol=1e-6, norm_tol=1e-7, what="per-frame dark fields")
    codes = torch.round(x * 65535).to(torch.int32).numpy().astype(np.uint16)
    xc = torch.from_numpy(oc.normalize_codes(codes))
    xb3, sig3 = ops.dark_field_blur(torch.from_numpy(codes).to(dev), dark.expand(nb, *dark.shape).to(dev),
                                    dark_std.expand(nb, *dark.shape).to(dev), std_mode="multiplier", std_value=0.05)
    assert_parity(xb3.cpu().numpy(), oe.conditional_gaussian_blur(xc, dark.unsqueeze(0)).numpy(), rtol=1e-6, norm_tol=1e-7,
                  what="blurred batch from codes")
    dterm3 = (oe.gaussian_blur3(xc) - xc) * (50.0 * m * (1 - m))
    assert_parity(sig3.cpu().numpy(), torch.sqrt((0.05 * xc) ** 2 + (dterm3 * dark_std) ** 2).numpy(), rtol=1e-5,
                  norm_tol=1e-6, what="effective sigma from codes")


def test_bands_with_halo_equal_whole(dev):
    """Row bands with the neighbouring rows as halo reproduce the untiled blur bit for bit."""
    from clair_torch_amd import ops
    x, sd, _, dark, dark_std, _ = _scene(4, 3, 3, 17, 8)
    nb = x.shape[0]
    xd, sdd = x.to(dev), sd.to(dev)
    dd, dsd = dark.expand(nb, *dark.shape).contiguous().to(dev), dark_std.expand(nb, *dark.shape).contiguous().to(dev)
    xb, sig = ops.dark_field_blur(xd, dd, dsd, std=sdd)
    h = x.shape[2]
    for r0, r1 in ((0, 6), (6, 11), (11, 17)):
        halo = torch.zeros((3, 3, 2, 8), device=dev)
        if r0 > 0:
            halo[:, :, 0] = xd[:, :, r0 - 1]
        if r1 < h:
            halo[:, :, 1] = xd[:, :, r1]
        xt, st = ops.dark_field_blur(xd[:, :, r0:r1].contiguous(), dd[:, :, r0:r1].contiguous(), dsd[:, :, r0:r1].contiguous(),
                                     std=sdd[:, :, r0:r1].contiguous(), tile=ops.TileGeometry(h_global=h, row_offset=r0),
                                     halo=halo)
        assert torch.equal(xt, xb[:, :, r0:r1]) and torch.equal(st, sig[:, :, r0:r1])
    with pytest.raises(ValueError):     # a band without its halo is refused (CT_ERR_INVALID_ARGUMENT)
        ops.dark_field_blur(xd[:, :, 6:11].contiguous(), dd[:, :, 6:11].contiguous(), None,
                            tile=ops.TileGeometry(h_global=h, row_offset=6))


@pytest.mark.parametrize("mode", ["linear", "catmull"])
def test_compute_hdr_image_with_dark_field(dev, mode):
    """Through the public API, streamed in two batches, against the eager restatement (both autograd variance terms)."""
    from clair_torch_amd.common.enums import InterpMode
    from clair_torch_amd.datasets import ArtefactStack, StackDataset, custom_collate
    from clair_torch_amd.inference import compute_hdr_image
    from clair_torch_amd.models import ICRFModelDirect
    from clair_torch_amd.training.losses import gaussian_value_weights
    from oracle import eager_torch as oe
    x, sd, t, dark, dark_std, lut = _scene(5, 6, 3, 12, 10)
    model = ICRFModelDirect(icrf=lut, interpolation_mode=InterpMode[mode.upper()]).to(dev)
    ds = StackDataset(x, t.tolist(), stds=sd)
    loader = DataLoader(ds, batch_size=4, shuffle=False, collate_fn=custom_collate)
    mean, std = compute_hdr_image(loader, "cuda", model, weight_fn=gaussian_value_weights,
                                  dark_field_dataset=ArtefactStack(dark, dark_std))
    mean_o, std_o = oe.merge_stack_dark(x, sd, t, lut, dark, dark_std, mode, True, [4, 2])
    assert_parity(mean.cpu().numpy(), mean_o.numpy(), rtol=1e-5, norm_tol=1e-6, what=f"dark-field merge mean {mode}")
    assert_parity(std.cpu().numpy(), std_o.numpy(), norm_tol=1e-5, elem_tol=4e-5 if mode == "catmull" else 1e-5,
                  what=f"dark-field merge std {mode}")
    # the correction matters on this scene (guards the test)
    plain_mean, _ = compute_hdr_image(loader, "cuda", model, weight_fn=gaussian_value_weights)
    assert float((plain_mean - mean).abs().max() / mean.abs().max()) > 1e-3
    # reference error behaviour: dark std missing -> AttributeError; no image uncertainties -> RuntimeError
    with pytest.raises(AttributeError):
        compute_hdr_image(loader, "cuda", model, weight_fn=gaussian_value_weights, dark_field_dataset=ArtefactStack(dark))
    bare = DataLoader(StackDataset(x, t.tolist()), batch_size=4, shuffle=False, collate_fn=custom_collate)
    with pytest.raises(RuntimeError, match="does not require grad"):
        compute_hdr_image(bare, "cuda", model, weight_fn=gaussian_value_weights,
                          dark_field_dataset=ArtefactStack(dark, dark_std))


def test_linearize_generator_with_dark_field(dev):
    from clair_torch_amd.common.enums import InterpMode
    from clair_torch_amd.datasets import ArtefactStack, StackDataset, custom_collate
    from clair_torch_amd.inference import linearize_dataset_generator
    from clair_torch_amd.models import ICRFModelDirect
    from oracle import eager_torch as oe
    x, sd, t, dark, dark_std, lut = _scene(6, 5, 3, 11, 14)
    model = ICRFModelDirect(icrf=lut, interpolation_mode=InterpMode.LINEAR).to(dev)
    loader = DataLoader(StackDataset(x, t.tolist(), stds=sd), batch_size=1, shuffle=False, collate_fn=custom_collate)
    got = list(linearize_dataset_generator(loader, "cuda", model, dark_field_dataset=ArtefactStack(dark, dark_std)))
    assert len(got) == 5
    for k, (lin, lsd, _) in enumerate(got):
        lin_o, sd_o = oe.linearize_frame_dark(x[k], sd[k], lut, dark, dark_std, "linear")
        assert_parity(lin.numpy(), lin_o.numpy(), rtol=1e-5, norm_tol=1e-6, what="dark-field linearize value")
        assert_parity(lsd.numpy(), sd_o.numpy(), rtol=1e-5, norm_tol=1e-6, what="dark-field linearize std")
