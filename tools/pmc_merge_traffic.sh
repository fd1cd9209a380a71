#!/bin/bash
# HBM traffic of the C2 merge launch: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (TCC has 4 slots:
# FETCH_SIZE takes 3, WRITE_SIZE 2), --kernel-trace only.  The harness runs the product kernel next to a streaming kernel of
# the same access pattern whose bytes are known exactly (stream_only<4>: 8 B per lane per exposure, 12 B written per
# element): the gfx950 FETCH_SIZE under-count of wide coalesced reads is calibrated on it, as the microarchitecture guide
# prescribes, and applied to the merge kernel.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_merge_traffic
rm -rf $O && mkdir -p $O
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/$c --output-format csv -- ./tools/merge_bench 32 4096 4096 3 "stream V4|pivot V4 mult|f64 V4 PF2 mult" > $O/$c.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % c, recursive=True):
        agg = collections.defaultdict(float); cnt = collections.Counter(); seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            agg[k] += float(r["Counter_Value"])
            key = (r["Dispatch_Id"], k)
            if key not in seen:
                seen.add(key); cnt[k] += 1
        for k in agg:
            if "stream_only" in k or "merge" in k:
                res.setdefault(k, {})[c] = agg[k] / cnt[k]
                print(c, k[-70:], "dispatches", cnt[k], "KB per dispatch %.1f" % (agg[k] / cnt[k]))
S, Q = 32 * 3 * 4096 * 4096, 3 * 4096 * 4096
stream = [v for k, v in res.items() if "stream_only<4>" in k][0]
corr = S * 2 / (stream["FETCH_SIZE"] * 1024)
print("calibration on stream_only<4>: %.4f GB read -> FETCH_SIZE correction x%.4f; WRITE_SIZE %.4f GB for %.4f GB written"
      % (S * 2 / 1e9, corr, stream["WRITE_SIZE"] * 1024 / 1e9, Q * 12 / 1e9))
for k, v in res.items():
    if "merge_pivot_kernel<unsigned short, 4, 1, 1, 2, true" in k or "merge_kernel<unsigned short, 4, 1, 1, 2" in k:
        rd, wr = v["FETCH_SIZE"] * 1024 * corr, v["WRITE_SIZE"] * 1024
        print(k[-64:], "read %.4f GB + written %.4f GB = %.4f GB (algorithmic %.4f GB)" % (rd / 1e9, wr / 1e9, (rd + wr) / 1e9, (S * 2 + Q * 12) / 1e9))
        json.dump({"kernel": k, "fetch_size_kb": v["FETCH_SIZE"], "write_size_kb": v["WRITE_SIZE"], "fetch_correction": corr,
                   "hbm_bytes_per_launch": int(rd + wr)}, open("$O/" + ("pivot" if "pivot" in k else "f64") + ".json", "w"), indent=1)
PY
