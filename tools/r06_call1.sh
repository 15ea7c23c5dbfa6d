#!/bin/bash
# round-6 baseline on one box: phase stamps of the shipped feed-forward kernel + whole-sample timing of the product library
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 120 build_tools/ffn_h2_probe 1600 > gpurun_out/r06a_ffn_probe.txt 2>&1
R05_LABEL="product" timeout 400 python tools/r05_ab.py once 2>&1 | grep -E "^sample|Error|error" > gpurun_out/r06a_once.txt
cat gpurun_out/r06a_ffn_probe.txt gpurun_out/r06a_once.txt
