#!/bin/bash
# whole GPU suite on the final build + the tiled self-attention kernel timed at T = 240 / 300
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python tools/tiled_attn_time.py > gpurun_out/r06_tiled_attention_time.txt 2>&1
cat gpurun_out/r06_tiled_attention_time.txt
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06c_gpu_suite.txt
cat gpurun_out/r06c_gpu_suite.txt
