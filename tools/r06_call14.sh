#!/bin/bash
# contact scan ranked by the fused distance: parity tests that touch it, then the hook's timing
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -x -q -m gpu -k "contact or correction or optimize or hook or golden or sample" 2>&1 | tail -8
python tools/corr_bench.py > gpurun_out/r06i_corr_bench.txt 2>&1; cat gpurun_out/r06i_corr_bench.txt | tail -2
python tools/contact_probe.py > gpurun_out/r06i_contact_probe.txt 2>&1; tail -40 gpurun_out/r06i_contact_probe.txt
