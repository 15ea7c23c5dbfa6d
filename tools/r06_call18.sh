#!/bin/bash
# hook with the predictor inside the scan's launch vs after it, inside the bench (same box, interleaved)
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
out=gpurun_out/r06l_hook_fused_ab.txt; : > $out
for rep in 1 2 3; do
  for v in 0 1; do
    r=$(INTERDIFF_HOOK_ONE_STREAM=$v python bench.py --no-cpu-baseline --no-kernel-profile --no-postopt --no-extra-configs 2>/dev/null | tail -1 | python -c "import sys,json; b=json.loads(sys.stdin.read()); print(b['ms_per_step'])")
    echo "predictor after the scan=$v: $r" >> $out
  done
done
cat $out
