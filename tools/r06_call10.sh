#!/bin/bash
# attention with the staged out-projection stores: parity tests + in-situ kernel stats
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "attention or mdm_forward or any_memory_length or chained or fallback" 2>&1 | tail -6 > gpurun_out/r06i_attn_tests.txt
cat gpurun_out/r06i_attn_tests.txt
INTERDIFF_CHAINS=1 tools/gpu_prof.sh r06i_bench python bench.py --no-cpu-baseline --no-kernel-profile --no-postopt --no-extra-configs > /dev/null 2>&1
head -12 gpurun_out/r06i_bench_kernel_stats.txt
