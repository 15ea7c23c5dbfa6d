#!/bin/bash
# round-6 fallback tests (debug deny list, mixed arithmetic) + the exclusive-CU test on one box
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "fallback or mixed or exclusive_cu" 2>&1 | tail -25 > gpurun_out/r06c_fallback_tests.txt
cat gpurun_out/r06c_fallback_tests.txt
