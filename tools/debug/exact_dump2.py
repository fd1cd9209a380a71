import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from _util import golden
from oracle import ct_oracle as oc
from clair_torch_amd import ops
dev = torch.device("cuda:0")
g = golden("merge"); t = g["merge_exposures"]; lut = torch.from_numpy(g["merge_lut"]).to(dev)
out = {}
for key in ["merge_u8_linear_none_constant_8", "merge_u8_nomodel_gauss_constant_8"]:
    _, ub, mname, wname, sname, pname = key.split("_")
    x = oc.normalize_codes(g[f"merge_{ub}_codes"])
    kw = dict(lut=None if mname == "nomodel" else lut, interp=None if mname == "nomodel" else mname, gaussian_weight=wname == "gauss", reference_order=True,
              std_mode=sname, std_value=0.01)
    st = ops.MergeState((3, 16, 16), dev, True)
    mean, std = ops.hdr_merge_batch(torch.from_numpy(x).to(dev), torch.from_numpy(t), state=st, finalize=True, **kw)
    out[key + "_std"] = std.cpu().numpy(); out[key + "_var"] = st.var.cpu().numpy(); out[key + "_sumw"] = st.sumw.cpu().numpy()
    print(key, "torch.sqrt(var)==std", float((torch.sqrt(st.var) == std).float().mean()), "np", float((np.sqrt(out[key + "_var"]) == out[key + "_std"]).mean()))
np.savez("gpurun_out/s5/exact_dump2.npz", **out)
