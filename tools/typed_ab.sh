#!/bin/bash
# A/B of merge_pivot_kernel harness builds, each sustained ALONE (the card sits on its power limit, so interleaved
# variants would share one clock state).  usage: typed_ab.sh <suffix> ...   for tools/merge_bench_<suffix>, built with
#   -DCT_PIVOT_TYPED_LOAD={0,1} [-DCT_PIVOT_KERNEL_ATTR=__attribute__((amdgpu_waves_per_eu(W,8)))]
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/typed_ab
mkdir -p $O
: > $O/ab.log
if [ -x $R/tools/typed_load_probe ]; then timeout -k 10 120 $R/tools/typed_load_probe > $O/probe.log 2>&1; tail -12 $O/probe.log >> $O/ab.log; fi
for rep in 1 2; do
  for b in "$@"; do
    [ -x $R/tools/merge_bench_$b ] || continue
    echo "== $b (run $rep)" >> $O/ab.log
    timeout -k 10 60 $R/tools/merge_bench_$b 32 4096 4096 3000 "pivot V4 mult" 2>&1 | grep -v "^std\[\|^== data" >> $O/ab.log || exit 1
  done
done
# parity on all three data sets (uniform random codes exercise the fallback) with the last build named
last="${@: -1}"
echo "== $last full (parity on three data sets)" >> $O/ab.log
timeout -k 10 120 $R/tools/merge_bench_$last 32 4096 4096 5 2>&1 | grep "^== data\|pivot vs f64\|pivot V4 mult" >> $O/ab.log
cat $O/ab.log
