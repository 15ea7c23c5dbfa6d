#!/bin/bash
# contact scan: box-pair prefetch depth 1 / 2 / 4 (build_ab/pf1, product, build_ab/pf4), same box, interleaved
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
out=gpurun_out/r06j_contact_prefetch_ab.txt; : > $out
for rep in 1 2 3; do
  for v in pf1 product pf4; do
    if [ $v = product ]; then unset INTERDIFF_HIP_LIB; else export INTERDIFF_HIP_LIB=$PWD/build_ab/$v/libinterdiff_hip.so; fi
    echo "$v: $(python tools/corr_bench.py --only scan_order 2>/dev/null | tail -1)" >> $out
  done
done
unset INTERDIFF_HIP_LIB
cat $out | cut -c1-200
