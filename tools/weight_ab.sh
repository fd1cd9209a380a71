#!/bin/bash
# A/B of the weight-evaluation variants of merge_pivot_kernel (CT_PIVOT_WEIGHT), each sustained ALONE, on the bench scene
# (independent pixels) and on the smooth ramp (spatially correlated codes).  usage: weight_ab.sh <suffix> ...
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/weight_ab
mkdir -p $O
: > $O/ab.log
for rep in 1 2; do
  for b in "$@"; do
    [ -x $R/tools/merge_bench_$b ] || continue
    echo "== $b (run $rep)" >> $O/ab.log
    timeout -k 10 60 $R/tools/merge_bench_$b 32 4096 4096 2500 "pivot V4 mult" 2>&1 | grep -v "^std\[" >> $O/ab.log || exit 1
  done
done
for b in "$@"; do
  [ -x $R/tools/merge_bench_$b ] || continue
  echo "== $b smooth ramp (spatially correlated codes), sustained alone" >> $O/ab.log
  MERGE_BENCH_MODE=2 timeout -k 10 60 $R/tools/merge_bench_$b 32 4096 4096 1500 "pivot V4 mult" 2>&1 | grep -v "^std\[" >> $O/ab.log || exit 1
done
for b in "$@"; do
  echo "== $b full (parity on three data sets, interleaved with the yardsticks)" >> $O/ab.log
  timeout -k 10 120 $R/tools/merge_bench_$b 32 4096 4096 7 2>&1 | grep "^== data\|pivot vs f64\|pivot V4 mult\|stream V4" >> $O/ab.log
done
cat $O/ab.log
