#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out; O=gpurun_out
export PYTHONUNBUFFERED=1
timeout 120 python tools/rb_tokens_diff.py 16 100 > $O/r05d_rb_tokens_diff.txt 2>&1
timeout 120 python tools/rb_tokens_diff.py 3 35 >> $O/r05d_rb_tokens_diff.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -k "mdm_forward or edge_sizes or memory_length or longer_memory or emulated_ranks or timed_route_equals_eager or chained_plain or two_chain or forward_step_matches" > $O/r05d_pytest.log 2>&1; echo "pytest rc $?" >> $O/r05d_pytest.log
cat $O/r05d_rb_tokens_diff.txt; tail -4 $O/r05d_pytest.log
