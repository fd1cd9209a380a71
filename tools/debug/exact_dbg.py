"""debug: reference-order kernel vs emulation on the first golden case"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from _util import golden, std_for
from oracle import ct_oracle as oc, eager_torch as oe
from clair_torch_amd import ops
dev = torch.device("cuda:0")
g = golden("merge"); t = g["merge_exposures"]; lut_h = torch.from_numpy(g["merge_lut"]); lut = lut_h.to(dev)
exact_exp = lambda v: torch.exp(v.double()).float()
for key in ["merge_u8_linear_none_constant_8", "merge_u8_linear_gauss_constant_8", "merge_u16_catmull_gauss_multiplier_8", "merge_u8_nomodel_none_constant_8", "merge_u8_lookup_gauss_constant_8"]:
    _, ub, mname, wname, sname, pname = key.split("_")
    codes = g[f"merge_{ub}_codes"]; x = oc.normalize_codes(codes)
    sd = std_for(sname, x, g[f"merge_{ub}_explicit_std"])
    mean_e, std_e = oe.merge_stack_reference_order(torch.from_numpy(x), torch.from_numpy(np.ascontiguousarray(sd)), torch.from_numpy(t),
                                                   None if mname == "nomodel" else lut_h, "linear" if mname == "nomodel" else mname, wname == "gauss", None, exp=exact_exp)
    kw = dict(lut=None if mname == "nomodel" else lut, interp=None if mname == "nomodel" else mname, gaussian_weight=wname == "gauss", reference_order=True,
              std_mode=sname, std_value=0.01 if sname == "constant" else 0.05)
    mean, std = ops.hdr_merge_batch(torch.from_numpy(x).to(dev), torch.from_numpy(t), **kw)
    a, b = std.cpu().numpy(), std_e.numpy()
    d = np.abs(a.astype(np.float64) - b)
    print(key, "differing", int((a != b).sum()), "of", a.size, "max rel", float((d / np.abs(b)).max()), "mean rel", float(np.abs(mean.cpu().numpy() - mean_e.numpy()).max() / np.abs(mean_e.numpy()).max()))
    idx = np.argmax(d / np.abs(b)); print("   worst at", np.unravel_index(idx, a.shape), a.reshape(-1)[idx], b.reshape(-1)[idx], "golden", g[key + "_std"].reshape(-1)[idx])
