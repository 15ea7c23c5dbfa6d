#!/bin/bash
# chunk-pipelined feed-forward kernel vs the shipped one: bits, back-to-back time, half-tick stamps
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out
timeout 120 build_tools/ffn_h2f_probe 1600 > gpurun_out/r06b_ffn_h2f_probe.txt 2>&1
cat gpurun_out/r06b_ffn_h2f_probe.txt
