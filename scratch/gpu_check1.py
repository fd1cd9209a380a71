"""Scratch: first on-GPU check of the merge kernel against the C oracle + a timing at C2 size."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from clair_torch_amd import ops
from oracle import ct_oracle as oc
sys.path.insert(0, 'tests')
from _util import golden, std_for, PARTITIONS, rel_norm

dev = torch.device('cuda:0')
print(torch.cuda.get_device_name(0))
g = golden('merge'); t = g['merge_exposures']; lut = torch.from_numpy(g['merge_lut']).to(dev)
worst = {}
for key in [str(k) for k in g['merge_cases']]:
    _, ub, mname, wname, sname, pname = key.split('_')
    codes = g[f'merge_{ub}_codes']
    x = oc.normalize_codes(codes)
    sd = std_for(sname, x, g[f'merge_{ub}_explicit_std'])
    part = PARTITIONS[pname]
    for as_codes in (True, False):
        stack = torch.from_numpy(codes if as_codes else x).to(dev)
        has = sname != 'none'
        st = ops.MergeState((3, 16, 16), dev, has) if len(part) > 1 else None
        k = 0; res = None
        for bi, b in enumerate(part):
            last = bi == len(part) - 1
            kw = dict(lut=None if mname == 'nomodel' else lut, interp=None if mname == 'nomodel' else mname,
                      gaussian_weight=wname == 'gauss', state=st, finalize=last)
            if sname == 'explicit': kw['std'] = torch.from_numpy(sd[k:k+b]).to(dev)
            elif has: kw.update(std_mode=sname, std_value=0.01 if sname == 'constant' else 0.05)
            res = ops.hdr_merge_batch(stack[k:k+b], torch.from_numpy(t[k:k+b]), **kw)
            k += b
        mean, std = res
        em = rel_norm(mean.cpu().numpy(), g[key + '_mean'])
        es = rel_norm(std.cpu().numpy(), g[key + '_std']) if has else 0.0
        kk = (mname, wname, sname, 'codes' if as_codes else 'f32')
        w = worst.get(kk, (0, 0)); worst[kk] = (max(w[0], em), max(w[1], es))
for k, v in sorted(worst.items()): print(k, 'mean %.2e std %.2e' % v)

# C2-size timing
N, C, H, W = 32, 3, 4096, 4096
gen = torch.Generator(device=dev).manual_seed(0)
stack = torch.randint(0, 65536, (N, C, H, W), device=dev, dtype=torch.int32, generator=gen).to(torch.uint16)
expo = torch.tensor([0.001 * 2 ** (n / 4) for n in range(N)], dtype=torch.float64)
lut3 = torch.stack([torch.linspace(0, 1, 256) ** p for p in (2.2, 2.4, 2.6)]).to(dev)
for _ in range(3): ops.hdr_merge_batch(stack, expo, lut=lut3, interp='linear', std_mode='multiplier', std_value=0.05)
torch.cuda.synchronize()
t0 = time.time(); K = 10
for _ in range(K): ops.hdr_merge_batch(stack, expo, lut=lut3, interp='linear', std_mode='multiplier', std_value=0.05)
torch.cuda.synchronize(); dt = (time.time() - t0) / K
byts = N*C*H*W*2 + C*H*W*12
print('C2 merge: %.3f ms  %.1f MPix/s  %.2f TB/s (%.1f%% of 8 TB/s)' % (dt*1e3, H*W/dt/1e6, byts/dt/1e12, byts/dt/8e12*100))
