"""debug: dump the reference-order kernel's outputs for a few golden cases (analysed offline against the emulation)"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from _util import golden, std_for
from oracle import ct_oracle as oc
from clair_torch_amd import ops
dev = torch.device("cuda:0")
g = golden("merge"); t = g["merge_exposures"]; lut = torch.from_numpy(g["merge_lut"]).to(dev)
out = {}
for key in ["merge_u8_linear_none_constant_8", "merge_u16_linear_none_constant_8", "merge_u8_linear_none_multiplier_8", "merge_u8_catmull_none_constant_8",
            "merge_u8_nomodel_none_constant_8", "merge_u8_linear_gauss_constant_8", "merge_u8_nomodel_gauss_constant_8"]:
    _, ub, mname, wname, sname, pname = key.split("_")
    x = oc.normalize_codes(g[f"merge_{ub}_codes"])
    kw = dict(lut=None if mname == "nomodel" else lut, interp=None if mname == "nomodel" else mname, gaussian_weight=wname == "gauss", reference_order=True,
              std_mode=sname, std_value=0.01 if sname == "constant" else 0.05)
    mean, std = ops.hdr_merge_batch(torch.from_numpy(x).to(dev), torch.from_numpy(t), **kw)
    out[key + "_std"] = std.cpu().numpy(); out[key + "_mean"] = mean.cpu().numpy()
    print(key, float((out[key + "_std"] == g[key + "_std"]).mean()))
os.makedirs("gpurun_out/s5", exist_ok=True)
np.savez("gpurun_out/s5/exact_dump.npz", **out)
