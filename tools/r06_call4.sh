#!/bin/bash
# go / no-go measurement of the persistent QaN layer's seam (tools/experiments/handoff_probe.hip)
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out
(timeout 120 build_tools/handoff_probe 11000 16000; timeout 120 build_tools/handoff_probe 2000 2000; timeout 120 build_tools/handoff_probe 11000 2000) > gpurun_out/r06d_handoff_probe.txt 2>&1
cat gpurun_out/r06d_handoff_probe.txt
