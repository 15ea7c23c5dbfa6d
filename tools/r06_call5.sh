#!/bin/bash
# hook after the marker-loop rewrite: exactness tests, then timing (corr_bench, contact_probe)
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "contact or denoised_fn or correction or corr32 or full_size or end_to_end or optimize" 2>&1 | tail -8 > gpurun_out/r06e_hook_tests.txt
timeout 200 python tools/corr_bench.py > gpurun_out/r06e_corr_bench.txt 2>&1
timeout 200 python tools/contact_probe.py > gpurun_out/r06e_contact_probe.txt 2>&1
cat gpurun_out/r06e_hook_tests.txt gpurun_out/r06e_corr_bench.txt gpurun_out/r06e_contact_probe.txt
