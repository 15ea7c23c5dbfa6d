#!/bin/bash
# whole GPU suite + rocprofv3 kernel stats of the bench command (one-chain form) on one box
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
INTERDIFF_CHAINS=1 IDF_STEP_MARKER="ln_linear_h2_kernel<1" tools/gpu_prof.sh r06b_bench python bench.py --no-cpu-baseline --no-kernel-profile --no-postopt --no-extra-configs > /dev/null 2>&1
head -30 gpurun_out/r06b_bench_kernel_stats.txt
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06b_gpu_suite.txt
cat gpurun_out/r06b_gpu_suite.txt
