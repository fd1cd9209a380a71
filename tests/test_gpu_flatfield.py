"""GPU parity of the flat-field correction epilogues against vectors recorded from the reference."""
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


def _setup(g, dev, codes, exposures, std_mode):
    from clair_torch_amd.common.enums import InterpMode, MissingStdMode
    from clair_torch_amd.datasets import StackDataset
    from clair_torch_amd.models import ICRFModelDirect
    mode = MissingStdMode.MULTIPLIER if std_mode == "multiplier" else MissingStdMode.NONE
    ds = StackDataset(codes, exposures, missing_std_mode=mode, missing_std_value=0.05, materialize_std=False)
    model = ICRFModelDirect(icrf=torch.from_numpy(g["ff_lut"]), interpolation_mode=InterpMode.LINEAR).to(dev)
    return ds, model


@pytest.mark.parametrize("pname,bs", [("6", 6), ("33", 3)])
def test_merge_with_flat_field(dev, pname, bs):
    from clair_torch_amd.common.transforms import CastTo, Normalize
    from clair_torch_amd.datasets import ArtefactStack, custom_collate
    from clair_torch_amd.inference import compute_hdr_image
    from clair_torch_amd.training.losses import gaussian_value_weights
    g = golden("flatfield")
    ds, model = _setup(g, dev, torch.from_numpy(g["ff_codes"]), g["ff_exposures"].tolist(), "multiplier")
    loader = DataLoader(ds, batch_size=bs, shuffle=False, collate_fn=custom_collate)
    ff = ArtefactStack(torch.from_numpy(g["ff_flat"]), torch.from_numpy(g["ff_flat_std"]))
    tf = [CastTo("float32"), Normalize(65535, 0)]
    mean, std = compute_hdr_image(loader, "cuda", model, weight_fn=gaussian_value_weights, flat_field_dataset=ff,
                                  gpu_transforms=tf)
    assert mean.dtype == torch.float64 and std.dtype == torch.float32
    assert_parity(mean.cpu().numpy(), g[f"ffmerge_ffstd_{pname}_mean"], rtol=1e-5, norm_tol=1e-6, what="ff mean")
    assert_parity(std.cpu().numpy(), g[f"ffmerge_ffstd_{pname}_std"], norm_tol=1e-5, elem_tol=1e-5, what="ff std")
    # reference behaviours at the edges: no flat-field std -> AttributeError (hdr_merge.py:134);
    # no image uncertainties -> TypeError (None + tensor, hdr_merge.py:151)
    with pytest.raises(AttributeError):
        compute_hdr_image(loader, "cuda", model, weight_fn=gaussian_value_weights, gpu_transforms=tf,
                          flat_field_dataset=ArtefactStack(torch.from_numpy(g["ff_flat"])))
    ds2, _ = _setup(g, dev, torch.from_numpy(g["ff_codes"]), g["ff_exposures"].tolist(), "none")
    with pytest.raises(TypeError):
        compute_hdr_image(DataLoader(ds2, batch_size=bs, collate_fn=custom_collate), "cuda", model, gpu_transforms=tf,
                          weight_fn=gaussian_value_weights, flat_field_dataset=ff)


@pytest.mark.parametrize("fsname", ["ffstd", "noffstd"])
@pytest.mark.parametrize("sname", ["none", "multiplier"])
def test_linearize_with_flat_field(dev, fsname, sname):
    from clair_torch_amd.common.transforms import CastTo, Normalize
    from clair_torch_amd.datasets import ArtefactStack, custom_collate
    from clair_torch_amd.inference import linearize_dataset_generator
    g = golden("flatfield")
    ds, model = _setup(g, dev, torch.from_numpy(g["ff_codes"][:3]), g["ff_exposures"][:3].tolist(), sname)
    loader = DataLoader(ds, batch_size=1, shuffle=False, collate_fn=custom_collate)
    ff = ArtefactStack(torch.from_numpy(g["ff_flat"]), torch.from_numpy(g["ff_flat_std"]) if fsname == "ffstd" else None)
    outs = list(linearize_dataset_generator(loader, "cuda", model, flatfield_dataset=ff,
                                            gpu_transforms=[CastTo("float32"), Normalize(65535, 0)]))
    lin = np.stack([o[0].numpy() for o in outs])
    sd = np.stack([o[1].numpy() for o in outs])
    assert_parity(lin, g[f"fflin_{fsname}_{sname}_val"], rtol=2e-7, norm_tol=1e-7, what="ff lin")
    assert_parity(sd, g[f"fflin_{fsname}_{sname}_std"], rtol=1e-6, norm_tol=1e-6, what="ff lin std")


def test_flat_field_row_bands_equal_whole(dev):
    """The spatial sums are additive over row bands: a band corrected with the whole image's sums equals the whole."""
    from clair_torch_amd import ops
    gen = torch.Generator().manual_seed(9)
    c, h, w = 3, 20, 12
    mean = torch.rand((c, h, w), generator=gen, dtype=torch.float64).to(dev)
    var = (0.01 * torch.rand((c, h, w), generator=gen)).to(dev)
    flat = (0.5 + 0.5 * torch.rand((c, h, w), generator=gen)).to(dev)
    fstd = (0.01 * torch.rand((c, h, w), generator=gen)).to(dev)
    m_w, s_w = ops.flatfield_correct(mean.clone(), var.clone(), flat, fstd, input_is_variance=True, through_mean=True)
    bands = [(0, 7), (7, 20)]
    # emulate the all-reduce: every band's local sums are replaced by the sum over bands
    locals_ = []

    def collect(t):
        locals_.append(t.clone())
    for r0, r1 in bands:
        ops.flatfield_correct(mean[:, r0:r1].contiguous(), var[:, r0:r1].contiguous(), flat[:, r0:r1].contiguous(),
                              fstd[:, r0:r1].contiguous(), input_is_variance=True, through_mean=True, reduce=collect,
                              global_pixels=h * w)
    total = locals_[0] + locals_[1]
    for r0, r1 in bands:
        m_b, s_b = ops.flatfield_correct(mean[:, r0:r1].contiguous(), var[:, r0:r1].contiguous(),
                                         flat[:, r0:r1].contiguous(), fstd[:, r0:r1].contiguous(), input_is_variance=True,
                                         through_mean=True, reduce=lambda t: t.copy_(total), global_pixels=h * w)
        assert torch.allclose(m_b, m_w[:, r0:r1], rtol=1e-12, atol=0)
        assert torch.allclose(s_b, s_w[:, r0:r1], rtol=1e-6, atol=0)
