#!/usr/bin/env python3
"""bench.py -- throughput of the hot path on MI355X, one JSON line on stdout (rank 0).

Default workload = BASELINE.json config C2: HDR merge + propagated uncertainty of a 32-exposure 4096x4096x3 uint16
stack (LINEAR ICRF, Gaussian weights, sigma = 0.05 * x derived in-kernel), inputs resident in HBM.
A "step" is one ct_hdr_merge_batch launch over the whole stack.  With --gpus N every rank merges its own C2-sized row
band of a (4096*N) x 4096 global image (weak scaling, no data-path collective: pixels are independent); a small RCCL
all_gather of per-band statistics runs once after the timed region.  `--scaling strong` is BASELINE config C5 as
stated: ONE 8192x8192x3 image cut into N row bands of 8192/N rows, with the per-band statistics all_gather inside
every timed step (fixed total work: the N-GPU value over the 1-GPU value is the speed-up).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

The JSON carries `roofline` (HBM: algorithmic bytes / mean kernel time from device events on the launch stream, against
the 8 TB/s spec peak) and `cpu_baseline` (the reference-equivalent eager-PyTorch path of oracle/eager_torch.py timed on
the host cores over a bounded sample; N = 1, rank 0 only).  Other workloads (--workload linearize|train|flatfield|video) are for
development and print the same shape of line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
POWERS = (2.2, 2.4, 2.6)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def host_cores():
    """Cores this process may use: the scheduler affinity, capped at the 16-core share a one-GPU box grants
    (os.cpu_count() reports the whole host, 256 logical CPUs, and oversubscribing them is 10x slower)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("CLAIR_BENCH_CORES", "16"))))


def make_lut(device):
    return torch.stack([torch.linspace(0, 1, 256) ** p for p in POWERS]).to(device)


def cpu_baseline_merge(n_exp, stops, seconds=12.0, tile=512):
    """Eager-PyTorch restatement (the reference's op sequence incl. autograd) on host cores, 512x512 tiles."""
    from clair_torch_amd.datasets import synthetic_exposure_stack
    from oracle import eager_torch as oe
    threads = host_cores()
    torch.set_num_threads(threads)
    lut = make_lut("cpu")
    codes, exposures = synthetic_exposure_stack(n_exp, 3, tile, tile, bits=16, stops_per_step=stops, seed=1236)
    x = codes.to(torch.int32).to(torch.float32) / 65535.0
    sd = x * torch.tensor(0.05)
    t = torch.tensor(exposures, dtype=torch.float64)
    oe.merge_stack(x[:, :, :64, :64], sd[:, :, :64, :64], t, lut, "linear", True)  # warm the op caches
    done, t0 = 0, time.perf_counter()
    while True:
        oe.merge_stack(x, sd, t, lut, "linear", True)
        done += 1
        el = time.perf_counter() - t0
        if el >= seconds or done >= 64:
            break
    return {"value": round(done * tile * tile / el / 1e6, 4), "unit": "MPix/s", "cores": threads, "kind": "port",
            "sample": f"{done} tile(s) of {n_exp}x{tile}x{tile}x3 float32 (same synthetic scene), single batch, "
                      f"oracle/eager_torch.merge_stack incl. autograd variance, {el:.1f} s"}


def _sync(dev):
    if torch.device(dev).type == "cuda":
        torch.cuda.synchronize()


def c5_strong_block(args, rank, world, dev, ops=None, make_stack=None):
    """BASELINE config C5 as stated, inside the driver's `--gpus N` line: ONE --global-size^2 x 3 image (default 8192)
    cut into `world` row bands, every timed step = merge + uncertainty of the rank's band (ct_hdr_merge_batch with the
    global geometry), the per-band statistics in one fused pass (ct_band_stats) and their all_gather over RCCL.  Fixed
    total work: value(N) / value(1) is the speed-up.  Barrier + synchronize on both sides, max over ranks, per-rank times
    gathered.  `ops` / `make_stack`: injection points for the 2-rank gloo test (tests/test_distributed_gloo.py), which
    drives exactly this function with CPU stand-ins for the two kernels."""
    if ops is None:
        from clair_torch_amd import ops
    if make_stack is None:
        from clair_torch_amd.datasets import synthetic_exposure_stack as make_stack
    n_exp, c = args.exposures, 3
    h_global = w = args.global_size
    if h_global % world:
        return {"skipped": f"{h_global} rows do not split into {world} equal bands"}
    h = h_global // world
    lut = make_lut(dev)
    codes, exposures = make_stack(n_exp, c, h_global, w, bits=16, stops_per_step=0.25, seed=1240, device=dev,
                                  row_range=(rank * h, (rank + 1) * h))
    t_dev = torch.tensor(exposures, dtype=torch.float64, device=dev)
    tile = ops.TileGeometry(h_global=h_global, row_offset=rank * h)
    kw = dict(lut=lut, interp="linear", gaussian_weight=True, std_mode="multiplier", std_value=0.05, tile=tile)

    def step():
        mean, std = ops.hdr_merge_batch(codes, t_dev, **kw)
        return gather_stats(ops.band_stats(mean, std), world)

    for _ in range(args.warmup):
        gathered = step()
    barrier(world)
    _sync(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gathered = step()
    _sync(dev)
    mine = time.perf_counter() - t0
    barrier(world)
    elapsed = max_over_ranks(time.perf_counter() - t0, world, dev)
    per_rank = [mine]
    if _dist_on():
        box = [None] * world
        torch.distributed.all_gather_object(box, mine)
        per_rank = [float(v) for v in box]
    px = h_global * w
    bytes_alg = (n_exp * c * px * 2 + c * px * 12 + c * px * 12) / world  # per GPU: stack read, outputs written, outputs re-read by the statistics
    return {
        "workload": f"C5: {n_exp}-exposure {h_global}x{w}x3 uint16 image in {world} row band(s) of {h} rows, merge+uncertainty, "
                    "per-band statistics (ct_band_stats) all_gathered inside every timed step",
        "scaling": "strong", "value": round(px * args.steps / elapsed / 1e6, 1), "unit": "MPix/s",
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step_per_rank": {"min": round(min(per_rank) / args.steps * 1e3, 4), "max": round(max(per_rank) / args.steps * 1e3, 4)},
        "world_seen": torch.distributed.get_world_size() if _dist_on() else 1,
        "bands_gathered": int(gathered.shape[0]), "finite": bool(torch.isfinite(gathered).all()),
        "hbm_frac_per_gpu": round(bytes_alg / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
    }


def run_merge(args, rank, world, dev):
    """C2 per GPU (weak scaling, the default) or C5's own geometry (--scaling strong: one global 8192x8192x3 image cut
    into `world` row bands of 8192 / world rows, per-band statistics gathered over RCCL inside every timed step)."""
    from clair_torch_amd import _native as nv
    from clair_torch_amd import ops
    from clair_torch_amd.datasets import synthetic_exposure_stack
    n_exp, c = args.exposures, 3
    strong = args.scaling == "strong"
    if strong:
        h_global, w = args.global_size, args.global_size
        if h_global % world:
            raise SystemExit(f"--scaling strong: {h_global} rows do not split into {world} equal bands")
        h = h_global // world
    else:
        h, w = args.height, args.width
        h_global = h * world
    lut = make_lut(dev)
    # One-time initialisation first, on an 8 x 3 x 8 x 8 stack: loading the 12 MB code object, the host-side constant
    # proofs (65 536-code loops) and the occupancy query take ~20 ms of host time on the first call.  Done after the
    # stack is generated they leave the GPU idle between the generation kernels and the warm-up launches, and the power
    # controller then meets the merge kernel from idle: launches 6..25 average 0.965 ms instead of 0.893 ms
    # (tools/transient_probe.py, both orders measured back to back on one box; no GPU work is added either way).
    init_codes, init_exp = synthetic_exposure_stack(8, c, 8, 8, bits=16, stops_per_step=0.25, seed=1, device=dev)
    ops.hdr_merge_batch(init_codes, torch.tensor(init_exp, dtype=torch.float64, device=dev), lut=lut, interp="linear",
                        gaussian_weight=True, std_mode="multiplier", std_value=0.05)
    bits = {65535: 16, 16383: 14, 4095: 12, 1023: 10}[args.max_code]   # e.g. 12-bit camera data in a uint16 container
    codes, exposures = synthetic_exposure_stack(n_exp, c, h_global, w, bits=bits, stops_per_step=0.25, seed=1236,
                                                device=dev, row_range=(rank * h, (rank + 1) * h))
    if bits < 16:
        codes = codes.to(torch.uint16) if codes.dtype != torch.uint16 else codes
    t_dev = torch.tensor(exposures, dtype=torch.float64, device=dev)
    tile = ops.TileGeometry(h_global=h_global, row_offset=rank * h) if world > 1 else None
    kw = dict(lut=lut, interp=args.interp, gaussian_weight=True, std_mode="multiplier", std_value=0.05, tile=tile,
              max_code=float(args.max_code))
    in_bytes = 2
    if args.input == "f32":  # what the reference's own DataLoader delivers: normalised float32 pixels
        codes = codes.to(torch.int32).to(torch.float32) / float(args.max_code)
        kw.pop("max_code")
        in_bytes = 4
    if args.std == "explicit":
        kw.update(std=(codes.to(torch.float32) * (0.05 / (float(args.max_code) if args.input == "u16" else 1.0))), std_mode="explicit")
        in_bytes += 4
    elif args.std == "none":
        kw.update(std_mode="none")
    if args.f64_moments:
        kw.update(force_f64_moments=True)
    if args.layout != "nchw":
        # the layout north_star names: (N, H, W, C) as OpenCV decodes it (BGR: channel order reversed on the fly)
        codes = (codes.flip(1) if args.layout == "nhwc_bgr" else codes).permute(0, 2, 3, 1).contiguous()
        if "std" in kw:
            kw["std"] = (kw["std"].flip(1) if args.layout == "nhwc_bgr" else kw["std"]).permute(0, 2, 3, 1).contiguous()
        kw.update(layout=args.layout, out_layout=args.out_layout)

    def band_stats(mean, std):
        # per channel: min, max, sum of the mean and of the std -- one fused pass (ct_band_stats; torch's reductions took
        # six passes, 1.3 ms for the whole 8192^2 image: tools/stats_timing.py)
        return ops.band_stats(mean, std)

    def step():
        mean, std = ops.hdr_merge_batch(codes, t_dev, **kw)
        if strong:  # C5: the per-band statistics gather is part of the job
            return mean, std, gather_stats(band_stats(mean, std), world)
        return mean, std, None

    for _ in range(args.warmup):
        step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier(world)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        mean, std, gathered = step()
        ev[k][1].record()
    torch.cuda.synchronize()
    barrier(world)
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, world, dev)
    per_launch = sorted(a.elapsed_time(b) for a, b in ev)  # device events on the launch stream (torch's current stream)
    kernel_ms = sum(per_launch) / args.steps
    # The first ~40 launches after any pause run inside a clock / power transient (fast, then 30 % slower, then settling:
    # tools/dvfs_transient.py, profiles/r02_dvfs_transient.log), so a short timed region reads 10-15 % above the steady
    # state.  The contract's numbers above are what they are; the steady state is measured AFTER the timed region and
    # reported beside them.
    steady = None
    if not strong:
        lead = max(0, 60 - (args.warmup + args.steps))
        for _ in range(lead):
            step()
        ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
        for a, b in ev2:
            a.record()
            step()
            b.record()
        torch.cuda.synchronize()
        steady_ms = sum(a.elapsed_time(b) for a, b in ev2) / len(ev2)
        steady = (steady_ms, args.warmup + args.steps + lead)
    if gathered is None:  # weak mode: per-band statistics gathered once, after the timed region
        gathered = gather_stats(band_stats(mean, std), world)
    px = h * w
    # stack (+ explicit std) read + float64 mean (+ float32 std) written
    bytes_alg = n_exp * c * px * in_bytes + c * px * (8 + (0 if args.std == "none" else 4))
    dtype_code = nv.DTYPE_U16 if args.input == "u16" else nv.DTYPE_F32
    flags = nv.MERGE_FIRST_BATCH | nv.MERGE_FINALIZE | (nv.MERGE_F64_MOMENTS if args.f64_moments else 0) | \
        (0 if args.std == "none" else nv.MERGE_STD_HINT)
    interp_code = {"linear": nv.INTERP_LINEAR, "lookup": nv.INTERP_LOOKUP, "catmull": nv.INTERP_CATMULL}[args.interp]
    kernel = nv.load().ct_hdr_merge_kernel_name(dtype_code, float(args.max_code), interp_code, 256, flags).decode()
    traffic = measured_traffic("merge_c2") if (args.input, args.std, h, w, n_exp, args.max_code, args.interp, args.layout) == \
        ("u16", "multiplier", 4096, 4096, 32, 65535, "linear", "nchw") else None
    name = "C5" if strong else "C2"
    out = {
        "metric": "MPix/s HDR-merged (+uncertainty) at N=32 4K RGB", "value": round(world * px * args.steps / elapsed / 1e6, 1),
        "unit": "MPix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{name}: {n_exp}-exposure {h}x{w}x3 {'uint16' if args.input == 'u16' else 'float32'} stack per GPU"
                               f"{'' if args.layout == 'nchw' else ' in ' + args.layout.upper() + ' memory order' + (' (outputs in the same order)' if args.out_layout == 'input' else '')}, "
                               f"merge{'' if args.std == 'none' else '+uncertainty'} ({args.interp.upper()} ICRF 3x256"
                               f"{'' if args.max_code == 65535 else ', Normalize(' + str(args.max_code) + ')'}, Gaussian weights, "
                               f"sigma: {args.std}, float64 mean + float32 std out)"
                               + (", per-band statistics all_gather (RCCL) inside every step" if strong else ""),
                   "global_image": f"{h_global}x{w}x3 in {world} row band(s)", "kernel": kernel,
                   "finite": bool(torch.isfinite(gathered).all())},
        "roofline": {"bound": "hbm", "achieved": round(bytes_alg / (kernel_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(bytes_alg / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "traffic": traffic,
                     "traffic_source": "profiles/traffic.json (builder's rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass on this "
                                       "workload, not measured by this run)" if traffic is not None else None,
                     "bytes_per_launch": bytes_alg,
                     "kernel_ms": round(kernel_ms, 4), "kernel_ms_min": round(per_launch[0], 4),
                     "kernel_ms_median": round(per_launch[len(per_launch) // 2], 4),
                     "kernel_ms_max": round(per_launch[-1], 4),
                     "timing": "hipEvent pairs around each launch on the launch stream"
                               + (" (includes the statistics reduction and all_gather)" if strong else "")},
    }
    if steady is not None:
        out["roofline"]["steady_state"] = {
            "kernel_ms": round(steady[0], 4), "frac": round(bytes_alg / (steady[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "launches": 100, "after_launches": steady[1],
            "note": "same launches, measured after the timed region once the clock transient of the first ~40 launches "
                    "has passed; not part of value / ms_per_step / frac"}
    if world > 1 and not strong:
        # the driver's multi-GPU line is weak scaling (above); BASELINE's configuration C5 -- one 8192^2 image in N bands,
        # statistics gather inside every step -- rides in the same JSON line
        codes = mean = std = gathered = None   # (the stack is captured by step(): rebind, then release)
        torch.cuda.empty_cache()
        block = c5_strong_block(args, rank, world, dev)
        out["c5_strong"] = block
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        codes = None
        torch.cuda.empty_cache()
        out["cpu_baseline"] = cpu_baseline_merge(n_exp, 0.25, args.cpu_seconds)
    return out


def run_linearize(args, rank, world, dev):
    """C4.  Default: kernel-only, `--frames` 1920x1080x3 uint16 frames resident in HBM, one ct_linearize_std launch per
    step.  --streamed: BASELINE's configuration as stated -- 1024 frames streamed from (pinned) host memory through the
    drop-in linearize_dataset_generator, results back on the host, end to end."""
    from clair_torch_amd import ops
    if args.streamed:
        return run_linearize_streamed(args, dev)
    frames = torch.randint(0, 65536, (args.frames, 3, 1080, 1920), device=dev, dtype=torch.int32).to(torch.uint16)
    if args.layout != "nchw":
        frames = (frames.flip(1) if args.layout == "nhwc_bgr" else frames).permute(0, 2, 3, 1).contiguous()
    lut = make_lut(dev)
    step = lambda: ops.linearize_frames(frames, lut, "linear", std_mode="multiplier", std_value=0.05, layout=args.layout)  # noqa: E731
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = e0.elapsed_time(e1) / args.steps
    px = args.frames * 1080 * 1920
    bytes_alg = px * 3 * (2 + 8)
    return {"metric": "frames/s linearized (+uncertainty), 1920x1080x3 uint16, kernel-only", "unit": "frames/s",
            "value": round(args.frames * args.steps / elapsed, 1), "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C4 (kernel-only): {args.frames} resident 1920x1080x3 uint16 frames per launch"
                                   f"{'' if args.layout == 'nchw' else ' in ' + args.layout.upper() + ' memory order'}, ct_linearize_std"},
            "roofline": {"bound": "hbm", "achieved": round(bytes_alg / (kernel_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(bytes_alg / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": measured_traffic("linearize_c4") if args.frames == 64 else None},
            **({} if args.no_cpu_baseline else {"cpu_baseline": cpu_baseline_linearize(min(args.cpu_seconds, 10.0))})}


def run_linearize_streamed(args, dev):
    """1024 frames (a pool of distinct pinned host frames, cycled) -> linearize_dataset_generator -> host.  One step =
    the whole stream.  The roofline here is the host link, not HBM: 49.8 MB per frame come back over PCIe (63 GB/s spec,
    the measured pinned copy rate of this box is reported next to it)."""
    from torch.utils.data import DataLoader
    from clair_torch_amd.common.enums import InterpMode, MissingStdMode
    from clair_torch_amd.common.transforms import CastTo, Normalize
    from clair_torch_amd.datasets import StackDataset, custom_collate
    from clair_torch_amd.inference import linearize_dataset_generator
    from clair_torch_amd.models import ICRFModelDirect
    n, pool, c, h, w = args.stream_frames, min(args.stream_frames, 128), 3, 1080, 1920
    gen = torch.Generator().manual_seed(1239)
    host = torch.randint(0, 65536, (pool, c, h, w), generator=gen, dtype=torch.int32).to(torch.uint16).pin_memory()

    class Cycled(StackDataset):  # frame i of the stream is pool frame i % pool: every one crosses PCIe on its own
        def __len__(self):
            return n

        def __getitem__(self, i):
            return i, self.values[i % pool], None, {"exposure_time": 1.0}

    ds = Cycled(host, [1.0] * pool, missing_std_mode=MissingStdMode.MULTIPLIER, missing_std_value=0.05, materialize_std=False)
    ds.files = list(range(n))
    model = ICRFModelDirect(icrf=make_lut("cpu"), interpolation_mode=InterpMode.LINEAR).to(dev)
    tf = [CastTo("float32"), Normalize(max_val=65535, min_val=0)]

    def stream():
        done, checksum = 0, 0.0
        for lin, sd, _meta in linearize_dataset_generator(DataLoader(ds, batch_size=1, shuffle=False, collate_fn=custom_collate),
                                                          dev, model, gpu_transforms=tf):
            done += 1
            if done % 256 == 0:
                checksum += float(lin[0, 0, 0]) + float(sd[0, 0, 0])
        return done, checksum

    # this box's pinned device->host copy rate (warmed up; the same transfer size the pipeline uses) and the kernel-only
    # rate on resident frames, reported next to the end-to-end figure
    from clair_torch_amd import ops
    d_buf = torch.empty((10, c, h, w), dtype=torch.float32, device=dev)
    h_buf = torch.empty((10, c, h, w), dtype=torch.float32, pin_memory=True)
    h_buf.copy_(d_buf, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(16):
        h_buf.copy_(d_buf, non_blocking=True)
    torch.cuda.synchronize()
    d2h_gbs = 16 * d_buf.numel() * 4 / (time.perf_counter() - t0) / 1e9
    del d_buf, h_buf
    resident = host[:64].to(dev)
    lut_dev = make_lut(dev)
    for _ in range(3):
        ops.linearize_frames(resident, lut_dev, "linear", std_mode="multiplier", std_value=0.05)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ops.linearize_frames(resident, lut_dev, "linear", std_mode="multiplier", std_value=0.05)
    torch.cuda.synchronize()
    kernel_only_fps = 10 * resident.shape[0] / (time.perf_counter() - t0)
    del resident
    torch.cuda.empty_cache()
    for _ in range(max(1, args.warmup // 3)):
        stream()
    torch.cuda.synchronize()
    times = []
    for _ in range(max(1, args.steps // 10)):
        t0 = time.perf_counter()
        done, _ = stream()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        assert done == n
    best, mean_t = min(times), sum(times) / len(times)
    out_bytes, in_bytes = 2 * c * h * w * 4, c * h * w * 2
    floor_spec = out_bytes / 63e9
    return {"metric": "frames/s linearized (+uncertainty), 1920x1080x3 uint16, streamed host -> MI355X -> host", "unit": "frames/s",
            "value": round(n / mean_t, 1), "n_gpus": 1, "steps": len(times), "warmup": max(1, args.warmup // 3),
            "ms_per_step": round(mean_t * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C4 (end to end): {n} frames 1920x1080x3 uint16 from pinned host memory through "
                                   "linearize_dataset_generator (LINEAR ICRF, sigma = 0.05 x in-kernel), value + std float32 "
                                   "back on the host",
                       "best_frames_per_s": round(n / best, 1),
                       "kernel_only_frames_per_s": round(kernel_only_fps, 1),
                       "bytes_per_frame": {"host_to_device": in_bytes, "device_to_host": out_bytes}},
            "roofline": {"bound": "pcie", "achieved": round(out_bytes * n / mean_t / 1e9, 2), "peak": 63.0, "unit": "GB/s",
                         "frac": round(out_bytes * n / mean_t / 1e9 / 63.0, 4), "traffic": None,
                         "floor_frames_per_s_at_spec": round(1.0 / floor_spec, 1),
                         "measured_pinned_d2h_GBps": round(d2h_gbs, 2),
                         "frac_of_measured_d2h": round(out_bytes * n / mean_t / 1e9 / d2h_gbs, 4)}}


def _timed_kernel(step, args):
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, e0.elapsed_time(e1) / args.steps


def _hbm_roofline(bytes_alg, kernel_ms):
    gbs = bytes_alg / (kernel_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None}


def run_flatfield(args, rank, world, dev):
    """flat-field epilogue of compute_hdr_image on a merged 4096x4096x3 image (float64 mean, float32
    variance, float32 flat field + its std): one ct_flatfield_sums + one ct_flatfield_apply per step."""
    from clair_torch_amd import ops
    c, h, w = 3, 4096, 4096
    gen = torch.Generator(device=dev).manual_seed(3)
    mean0 = torch.rand((c, h, w), generator=gen, device=dev, dtype=torch.float64)
    var0 = torch.rand((c, h, w), generator=gen, device=dev, dtype=torch.float32) * 1e-4
    flat = 0.5 + torch.rand((c, h, w), generator=gen, device=dev, dtype=torch.float32)
    flat_std = 0.01 * flat
    mean, var = mean0.clone(), var0.clone()

    def step():  # in place; the values drift but the work per step is identical
        ops.flatfield_correct(mean, var, flat, flat_std, input_is_variance=True, through_mean=True)

    elapsed, kernel_ms = _timed_kernel(step, args)
    q = c * h * w
    # sums: value 8 + flat 4 read; apply: value 8 r + 8 w, variance 4 r + 4 w, flat 4, flat std 4
    bytes_alg = q * ((8 + 4) + (16 + 8 + 4 + 4))
    return {"metric": "MPix/s flat-field corrected (+variance term), 4096x4096x3", "unit": "MPix/s",
            "value": round(h * w * args.steps / elapsed / 1e6, 1), "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "8f-1: flat-field epilogue of the merge, 4096x4096x3, ct_flatfield_sums + ct_flatfield_apply"},
            "roofline": _hbm_roofline(bytes_alg, kernel_ms)}


def run_video(args, rank, world, dev):
    """compute_video_mean_and_std, batches of 32 resident 1920x1080x3 uint16 frames, linearized in-kernel."""
    from clair_torch_amd import ops
    b, c, h, w = 32, 3, 1080, 1920
    frames = torch.randint(0, 65536, (b, c, h, w), device=dev, dtype=torch.int32).to(torch.uint16)
    lut = make_lut(dev)
    mean = torch.zeros((c, h, w), device=dev, dtype=torch.float32)
    m2 = torch.zeros_like(mean)
    seen = [0]

    def step():
        ops.video_stats_batch(frames, mean, m2, seen[0], lut=lut, interp="linear")
        seen[0] += b

    elapsed, kernel_ms = _timed_kernel(step, args)
    q = c * h * w
    bytes_alg = q * (b * 2 + 16)  # every frame once + (mean, m2) float32 state read and written
    return {"metric": "frames/s video mean/std (WBOMeanVar), 1920x1080x3 uint16, kernel-only", "unit": "frames/s",
            "value": round(b * args.steps / elapsed, 1), "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"8f-2: {b} resident 1920x1080x3 uint16 frames per batch, ct_video_stats_batch (LINEAR ICRF)"},
            "roofline": _hbm_roofline(bytes_alg, kernel_ms)}


def cpu_baseline_train(seconds=12.0, n_exp=8, size=96):
    """One training step of the eager-PyTorch restatement (forward, float64 residuals over all pairs, autograd into the
    LUT, penalties) on host cores: SURVEY 8(d) times C3's CPU baseline at batch_size 8 (P <= 28) because the eager path
    needs > 100 B per pair-pixel."""
    from clair_torch_amd.datasets import synthetic_exposure_stack
    from oracle import eager_torch as oe
    threads = host_cores()
    torch.set_num_threads(threads)
    codes, exposures = synthetic_exposure_stack(n_exp, 3, size, size, bits=16, stops_per_step=0.125, seed=1237)
    x = codes.to(torch.int32).to(torch.float32) / 65535.0
    t = torch.tensor(exposures, dtype=torch.float64)
    lut = torch.stack([torch.linspace(0, 1, 256) ** 2.5 for _ in range(3)]).requires_grad_(True)
    done, t0 = 0, time.perf_counter()
    while True:
        loss, _, sp = oe.training_loss(x, None, t, lut, "linear", 0.25, 1 / 255, 254 / 255, True, False, 10.0, 1.0, 1.0, 1.0)
        torch.autograd.grad(loss.sum(), lut)
        done += 1
        el = time.perf_counter() - t0
        if el >= seconds or done >= 400:
            break
    pairs = int(sp.shape[0])
    return {"value": round(done / el, 3), "unit": "iters/s", "cores": threads, "kind": "port",
            "pair_pixels_per_s": round(done * pairs * 3 * size * size / el / 1e6, 2),
            "sample": f"{done} step(s) of oracle/eager_torch.training_loss + autograd on a {n_exp}x{size}x{size}x3 float32 stack "
                      f"({pairs} pairs; the full C3 shape has 888 pairs x 12.6 M elements and does not fit the eager path), {el:.1f} s"}


def cpu_baseline_linearize(seconds=10.0):
    """oracle/eager_torch.linearize_frame (forward + autograd.grad per frame, as the reference) on 1920x1080x3 frames."""
    from oracle import eager_torch as oe
    threads = host_cores()
    torch.set_num_threads(threads)
    gen = torch.Generator().manual_seed(5)
    frame = torch.randint(0, 65536, (3, 1080, 1920), generator=gen, dtype=torch.int32).to(torch.float32) / 65535.0
    lut = make_lut("cpu")
    sd = frame * 0.05
    done, t0 = 0, time.perf_counter()
    while True:
        oe.linearize_frame(frame, sd, lut, "linear")
        done += 1
        el = time.perf_counter() - t0
        if el >= seconds or done >= 200:
            break
    return {"value": round(done / el, 3), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{done} frame(s) 1920x1080x3 float32 through oracle/eager_torch.linearize_frame (forward + autograd.grad), {el:.1f} s"}


def run_train_through_api(args, rank, world, dev):
    """BASELINE configuration C3 AS STATED: `--steps` epochs (500 in BASELINE) of train_icrf itself on the device-resident
    64 x 2048 x 2048 x 3 stack -- DataLoader + custom_collate, per-channel fused Adam (lr 1e-3), ReduceLROnPlateau(0.5, 50),
    alpha, beta, gamma, delta = 10, 1, 1, 1, relative loss, no uncertainty weighting, ratio threshold 0.25
    (scripts/run_icrf_model_training.py:49-69, scripts/config.yaml:31-34).  Also: the same call on a down-scaled copy
    against the eager oracle trained the same way on the CPU (final LUT)."""
    from torch.utils.data import DataLoader
    from clair_torch_amd.common.enums import InterpMode
    from clair_torch_amd.common.transforms import CastTo, Normalize
    from clair_torch_amd.datasets import StackDataset, custom_collate, synthetic_exposure_stack
    from clair_torch_amd.models import ICRFModelDirect
    from clair_torch_amd.training import train_icrf
    n_exp, size = args.train_exposures, args.train_size
    tf = [CastTo("float32"), Normalize(max_val=65535, min_val=0)]

    def fit(codes, exposures, epochs, device):
        model = ICRFModelDirect(n_points=256, channels=3, interpolation_mode=InterpMode.LINEAR, initial_power=2.5).to(device)
        opts = [torch.optim.Adam(model.channel_params(c), lr=1e-3, fused=True) for c in range(3)]
        scheds = [torch.optim.lr_scheduler.ReduceLROnPlateau(o, mode="min", factor=0.5, patience=50) for o in opts]
        ds = StackDataset(codes, exposures)
        loader = DataLoader(ds, batch_size=len(exposures), shuffle=False, collate_fn=custom_collate)
        train_icrf(loader, len(exposures), device, model, optimizers=opts, schedulers=scheds, use_relative_linearity_loss=True,
                   use_uncertainty_weighting=False, epochs=epochs, patience=10 ** 9, alpha=10.0, beta=1.0, gamma=1.0, delta=1.0,
                   lower_valid_threshold=1 / 255, upper_valid_threshold=254 / 255, exposure_ratio_threshold=0.25,
                   gpu_transforms=tf, verbose=False)
        return model

    codes, exposures = synthetic_exposure_stack(n_exp, 3, size, size, bits=16, stops_per_step=0.125, seed=1237, device=dev)
    fit(codes, exposures, max(2, args.warmup), dev)  # code objects, pair tables, allocator
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fit(codes, exposures, args.steps, dev)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # kernel share: the two launches of a step timed back to back without the host loop around them
    from clair_torch_amd import ops
    from clair_torch_amd.common.general_functions import get_valid_exposure_pairs
    from clair_torch_amd.training import linearity_loss
    t = torch.tensor(exposures, dtype=torch.float64)
    i, j, r = get_valid_exposure_pairs(t, 0.25)
    pairs = ops.PairList(i, j, r, n_exp, dev)
    lut = torch.stack([torch.linspace(0, 1, 256) ** 2.5 for _ in range(3)]).to(dev).requires_grad_(True)

    def kernels():
        lin, _ = linearity_loss(lut, codes, pairs, interp="linear", lower=1 / 255, upper=254 / 255, use_relative=True, use_unc_weight=False)
        torch.autograd.grad(lin.sum(), lut)

    for _ in range(3):
        kernels()
    torch.cuda.synchronize()
    k0 = time.perf_counter()
    for _ in range(20):
        kernels()
    torch.cuda.synchronize()
    kernel_ms = (time.perf_counter() - k0) / 20 * 1e3
    del codes
    torch.cuda.empty_cache()
    # final LUT against the eager oracle on a down-scaled copy (same optimiser / scheduler recipe on the CPU)
    small, exp_s = synthetic_exposure_stack(16, 3, 48, 48, bits=16, stops_per_step=0.5, seed=1241)
    ep = 40
    got = fit(small.to(dev), exp_s, ep, dev).icrf.detach().cpu()
    want = _oracle_training_run(small, exp_s, ep)
    out = {"metric": "ICRF training iterations/s (train_icrf through the public API)", "unit": "iters/s",
           "value": round(args.steps / elapsed, 3), "n_gpus": 1, "steps": args.steps, "warmup": max(2, args.warmup),
           "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"C3 as stated: {args.steps} epochs of train_icrf, {n_exp}-exposure {size}x{size}x3 uint16 stack resident in HBM, "
                                  f"{pairs.n_pairs} pairs, relative loss, per-channel fused Adam + ReduceLROnPlateau(0.5, 50), "
                                  "alpha, beta, gamma, delta = 10, 1, 1, 1",
                      "kernel_ms_per_step": round(kernel_ms, 3),
                      "host_and_small_kernels_ms_per_step": round(elapsed / args.steps * 1e3 - kernel_ms, 3),
                      "final_lut_vs_eager_oracle": {"stack": "16x48x48x3 uint16", "epochs": ep,
                                                    "max_abs_diff": float((got - want).abs().max()),
                                                    "mean_abs_diff": float((got - want).abs().mean()),
                                                    "bins_off_by_more_than_1e-4": int(((got - want).abs() > 1e-4).sum()),
                                                    "max_abs_change_from_start": float((want - torch.stack([torch.linspace(0, 1, 256) ** 2.5] * 3)).abs().max())}},
           "roofline": _train_roofline(pairs.n_pairs, 3 * size * size, elapsed / args.steps)}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_train(args.cpu_seconds)
    return out


def _oracle_training_run(codes, exposures, epochs):
    """The reference's training recipe on the eager oracle (CPU): per-channel Adam (lr 1e-3), ReduceLROnPlateau(0.5, 50),
    alpha 10; first step dead as in the reference (the curve only becomes a function of the parameters after the first
    update_icrf, icrf_training.py:92-156)."""
    from oracle import eager_torch as oe
    x = codes.to(torch.int32).to(torch.float32) / 65535.0
    t = torch.tensor(exposures, dtype=torch.float64)
    params = [torch.nn.Parameter(torch.linspace(0, 1, 256) ** 2.5) for _ in range(3)]
    opts = [torch.optim.Adam([p], lr=1e-3) for p in params]
    scheds = [torch.optim.lr_scheduler.ReduceLROnPlateau(o, mode="min", factor=0.5, patience=50) for o in opts]
    for epoch in range(epochs):
        for o in opts:
            o.zero_grad()
        curve = torch.stack(list(params))
        loss, _, _ = oe.training_loss(x, None, t, curve, "linear", 0.25, 1 / 255, 254 / 255, True, False, 10.0, 1.0, 1.0, 1.0)
        if epoch > 0:  # the dead first step: no gradient reaches the parameters, Adam steps on None grads do nothing
            for c in range(3):
                loss[c].backward(retain_graph=True)
            for o in opts:
                o.step()
        for c, sc in enumerate(scheds):
            sc.step(float(loss[c].detach()))
    return torch.stack([p.detach() for p in params])


def run_train(args, rank, world, dev):
    """C3: one train_icrf optimizer step (forward sums + LUT gradient + Adam) on a 64-exposure 2048x2048x3 stack."""
    if args.through_api:
        return run_train_through_api(args, rank, world, dev)
    from clair_torch_amd import ops
    from clair_torch_amd.common.general_functions import get_valid_exposure_pairs
    from clair_torch_amd.datasets import synthetic_exposure_stack
    from clair_torch_amd.training import linearity_loss
    n_exp, h, w = args.train_exposures, args.train_size, args.train_size
    codes, exposures = synthetic_exposure_stack(n_exp, 3, h, w, bits=16, stops_per_step=0.125, seed=1237, device=dev)
    t = torch.tensor(exposures, dtype=torch.float64)
    max_code = float(args.max_code)
    if args.max_code != 65535:  # 10- / 12- / 14-bit camera data held in uint16, normalised by its own maximum
        shift = {16383: 2, 4095: 4, 1023: 6}[args.max_code]
        codes = (codes.to(torch.int32) >> shift).to(torch.uint16)
    if args.layout != "nchw":  # the stack as decoded: (N, H, W, C), BGR order for nhwc_bgr; the staging gathers it
        codes = (codes.flip(1) if args.layout == "nhwc_bgr" else codes).permute(0, 2, 3, 1).contiguous()
    i, j, r = get_valid_exposure_pairs(t, 0.25)
    pairs = ops.PairList(i, j, r, n_exp, dev)
    params = [torch.nn.Parameter((torch.linspace(0, 1, 256) ** 2.5).to(dev)) for _ in range(3)]
    opts = [torch.optim.Adam([p], lr=1e-3, fused=True) for p in params]  # as train_icrf builds them on the GPU

    def step():
        for o in opts:
            o.zero_grad()
        lut = torch.stack(params)
        lin, _ = linearity_loss(lut, codes, pairs, interp="linear", lower=1 / 255, upper=254 / 255, use_relative=True,
                                use_unc_weight=False, layout=args.layout, max_code=max_code)
        lin.sum().backward()
        for o in opts:
            o.step()
        return lin

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    return {"metric": "ICRF training iterations/s (linearity term fwd+bwd + Adam)", "unit": "iters/s",
            "value": round(args.steps / elapsed, 3), "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C3: {n_exp}-exposure {h}x{w}x3 uint16 stack"
                                   f"{'' if args.layout == 'nchw' else ' in ' + args.layout.upper() + ' memory order'}"
                                   f"{'' if args.max_code == 65535 else f' with codes 0..{args.max_code}'}, "
                                   f"{pairs.n_pairs} pairs, relative loss"},
            "roofline": _train_roofline(pairs.n_pairs, 3 * h * w, elapsed / args.steps),
            **({} if args.no_cpu_baseline else {"cpu_baseline": cpu_baseline_train(args.cpu_seconds)})}


def _train_roofline(n_pairs, elements, seconds):
    """SURVEY 8(d) convention for C3: ~12 flop per pair and pixel-channel forward and the same again backward, against
    the 157.3 TFLOP/s float32 vector peak.  Whole optimizer step (both kernels + Adam), host clock."""
    flops = 2.0 * 12.0 * n_pairs * elements
    achieved = flops / seconds / 1e12
    return {"bound": "valu", "achieved": round(achieved, 2), "peak": 157.3, "unit": "TFLOP/s",
            "frac": round(achieved / 157.3, 4), "traffic": measured_traffic("train_c3") if (n_pairs, elements) == (888, 3 * 2048 * 2048) else None}


def measured_traffic(key):
    """HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 FETCH_SIZE/WRITE_SIZE, gfx950
    correction applied there); None when no profile for this workload exists."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            return json.load(fh).get(key, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def _dist_on():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def barrier(world):
    if _dist_on():
        torch.distributed.barrier()


def max_over_ranks(value, world, dev):
    if not _dist_on():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


def gather_stats(stats, world):
    if not _dist_on():
        return stats.unsqueeze(0)
    out = [torch.empty_like(stats) for _ in range(world)]
    torch.distributed.all_gather(out, stats.contiguous())
    return torch.stack(out)


def main():
    # Exactly ONE line may reach stdout (the JSON).  RCCL prints a version banner to fd 1 when its communicator
    # comes up, so fd 1 is pointed at stderr for the whole run and the JSON is written to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="merge", choices=["merge", "linearize", "train", "flatfield", "video"])
    ap.add_argument("--input", default="u16", choices=["u16", "f32"], help="merge: stack element type (default = C2)")
    ap.add_argument("--std", default="multiplier", choices=["multiplier", "explicit", "none"],
                    help="merge: uncertainty source (default = C2: sigma = 0.05 x derived in-kernel)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="merge with --gpus N: weak = one C2-sized band per GPU (default); strong = C5, a fixed "
                         "--global-size^2 image cut into N row bands with the per-band statistics gather in every step")
    ap.add_argument("--global-size", type=int, default=8192, help="--scaling strong: rows = columns of the global image")
    ap.add_argument("--f64-moments", action="store_true", help="merge: time the float64-moment kernel (round-1 path)")
    ap.add_argument("--max-code", type=int, default=65535, choices=[65535, 16383, 4095, 1023],
                    help="merge: what Normalize divides the uint16 codes by (12-bit camera data: 4095)")
    ap.add_argument("--interp", default="linear", choices=["linear", "lookup", "catmull"], help="merge: ICRF interpolation mode")
    ap.add_argument("--out-layout", default="planar", choices=["planar", "input"],
                    help="merge with an interleaved --layout: outputs planar (C,H,W) like the reference, or in the input's own (H,W,C) order (CT_MERGE_OUT_AS_INPUT)")
    ap.add_argument("--layout", default="nchw", choices=["nchw", "nhwc", "nhwc_bgr"],
                    help="merge / linearize: memory order of the stack (nhwc = as decoded, nhwc_bgr = OpenCV's channel order)")
    ap.add_argument("--exposures", type=int, default=32)
    ap.add_argument("--height", type=int, default=4096)
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--streamed", action="store_true",
                    help="linearize: end to end, --stream-frames frames from pinned host memory through the drop-in generator")
    ap.add_argument("--stream-frames", type=int, default=1024)
    ap.add_argument("--through-api", action="store_true",
                    help="train: BASELINE C3 as stated -- --steps epochs of train_icrf itself (DataLoader, per-channel fused Adam, "
                         "ReduceLROnPlateau, penalties) instead of the two kernels + Adam")
    ap.add_argument("--train-exposures", type=int, default=64)
    ap.add_argument("--train-size", type=int, default=2048)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} "
                             f"(WORLD_SIZE is {world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ  # under torch.distributed.run
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", device_id=dev)
    from clair_torch_amd import _native
    _native.load()
    fn = {"merge": run_merge, "linearize": run_linearize, "train": run_train, "flatfield": run_flatfield,
          "video": run_video}[args.workload]
    out = fn(args, rank, world, dev)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if _dist_on():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
