#!/bin/bash
# balanced phase-1 unit map of the 32-row feed-forward kernel: old vs new probe (stamps), bits of the new kernel against the experiment kernel (which equals the old kernel bit for bit)
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out
(echo "== old (whole-tile map)"; timeout 120 build_tools/ffn_h2_probe_old 1600 | head -9; echo "== new (balanced units)"; timeout 120 build_tools/ffn_h2_probe 1600 | head -9; echo "== old again"; timeout 120 build_tools/ffn_h2_probe_old 1600 | head -9; echo "== new again"; timeout 120 build_tools/ffn_h2_probe 1600 | head -9; timeout 120 build_tools/ffn_h2f_probe 1600 | grep "bit-for-bit") > gpurun_out/r06f_ffn_balanced_ab.txt 2>&1
cat gpurun_out/r06f_ffn_balanced_ab.txt
