#!/bin/bash
# LDS conflict share of the row block after a layout change: SQ counter pass on the kernel-level bench + parity + bit-identity of the token forms + library A/B vs build_ab/head
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
python tools/rb_tokens_diff.py 16 100 | head -2
timeout 600 python -m pytest tests -m gpu -q -x -k "mdm_forward or edge_sizes or memory_length or timed_route_equals_eager" 2>&1 | tail -2
tools/gpu_pmc.sh r05g_sq_kbench "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" python tools/kbench.py --reps 3 > /dev/null 2>&1
grep -E "rowblock8_kernel|self_attn_kernel<true, 1>" gpurun_out/r05g_sq_kbench_pmc.txt | grep -E "LDS" | cut -c1-160
bash tools/r05_call5.sh head
