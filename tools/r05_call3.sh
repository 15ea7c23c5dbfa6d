#!/bin/bash
# Round 5, third GPU call: the eight-wave row block with 8 tokens per workgroup -- parity (incl. 8-token shards == 16-token unsharded batch, bit for bit), phase stamps, A/B.
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out; O=gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -x -q -k "mdm_forward or edge_sizes or memory_length or longer_memory or emulated_ranks or timed_route_equals_eager or chained_plain or two_chain or forward_step_matches or exclusive_cu" > $O/r05c_pytest.log 2>&1; echo "pytest rc $?" >> $O/r05c_pytest.log
( echo "== 8 tokens per workgroup"; timeout 60 build_tools/rowblock_probe 16 100 1 0 8; echo "== 16 tokens per workgroup"; timeout 60 build_tools/rowblock_probe 16 100 1 0 16 ) > $O/r05c_rowblock_probe.txt 2>&1
timeout 600 python tools/r05_ab.py rb_tokens 2 > $O/r05c_rb_tokens_ab.txt 2>&1
R05_CLIPS=32 timeout 600 python tools/r05_ab.py rb_tokens 2 > $O/r05c_rb_tokens_ab_B32.txt 2>&1
tail -3 $O/r05c_pytest.log; grep -E "total|==" $O/r05c_rowblock_probe.txt; grep sample $O/r05c_rb_tokens_ab.txt
