#!/bin/bash
# shipped 32-row feed-forward kernel with eight loader waves (16-wave workgroup) against the 8-wave form: bits, time, stamps
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out
timeout 200 build_tools/ffn_h2_probe 1600 > gpurun_out/r06g_ffn_loader_waves.txt 2>&1
head -24 gpurun_out/r06g_ffn_loader_waves.txt
