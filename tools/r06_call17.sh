#!/bin/bash
# hook with the predictor beside the contact scan: parity (overlapped == one-stream, eager and captured), the hook tests, timing
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -x -q -m gpu -k "predictor_inside or correction_edge or objproj" 2>&1 | tail -12
python tools/corr_bench.py --only scan_order > gpurun_out/r06k_corr_bench.txt 2>&1; tail -1 gpurun_out/r06k_corr_bench.txt
python bench.py --no-cpu-baseline --no-kernel-profile --no-postopt --no-extra-configs 2>/dev/null | tail -1 | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('ms_per_step', b['ms_per_step'], b['value'])"
