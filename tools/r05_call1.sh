#!/bin/bash
# Round 5, first GPU call: verdicts of the exclusive-CU check, the GPU suite (minus the fixtures still being generated), the co-residency victim variants,
# the two cheap A/Bs (step-tail order, LDS stride sets) and a short bench line.  Everything lands under gpurun_out/r05a_*.
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out; O=gpurun_out
export PYTHONUNBUFFERED=1
python -c "
from interdiff_amd import _lib
t, bad = _lib.exclusive_cu_report(); print(t); print('not exclusive:', bad)" > $O/r05a_exclusive_cu.txt 2>&1
# the eight-wave row block first, on its own: if it is wrong, the rest of the call runs on round 4's four-wave kernel so that every other result stays usable
timeout 600 python -m pytest tests -m gpu -x -q -k "mdm_forward_golden or mdm_forward_bench_shape or mdm_forward_split_f16_vs_exact or edge_sizes or memory_length" > $O/r05a_pytest_rowblock8.log 2>&1
if ! tail -1 $O/r05a_pytest_rowblock8.log | grep -q " passed" || grep -q "failed" $O/r05a_pytest_rowblock8.log; then export INTERDIFF_ROWBLOCK_WAVES=4; echo "rowblock8 FAILED: the rest of this call uses INTERDIFF_ROWBLOCK_WAVES=4" >> $O/r05a_pytest_rowblock8.log; fi
timeout 900 python -m pytest tests -m gpu -x -q -k "not well_conditioned" > $O/r05a_pytest.log 2>&1; echo "pytest rc $?" >> $O/r05a_pytest.log
( echo "== build_tools/coresidency_repro (stand-alone, default flags)"; timeout 120 build_tools/coresidency_repro 2 5000 1024
  echo "== build_tools/coresidency_repro_noslp (-fno-slp-vectorize: no packed-fp32 instruction in the victim)"; timeout 120 build_tools/coresidency_repro_noslp 2 5000 1024
  echo "== build_tools/coresidency_probe 20000 1024 1"; timeout 300 build_tools/coresidency_probe 20000 1024 1
  echo "== build_tools/coresidency_probe_noslp 20000 1024 1"; timeout 300 build_tools/coresidency_probe_noslp 20000 1024 1 ) > $O/r05a_coresidency_victim_variants.txt 2>&1
( echo "== product library"; timeout 300 python tools/hook_stage_probe.py aggr 4
  echo "== library built with -fno-slp-vectorize"; INTERDIFF_HIP_LIB=build_ab/noslp/libinterdiff_hip.so timeout 300 python tools/hook_stage_probe.py aggr 4 ) > $O/r05a_hook_stage_victims.txt 2>&1
[ -z "$INTERDIFF_ROWBLOCK_WAVES" ] && timeout 600 python tools/r05_ab.py rowblock_waves 3 > $O/r05a_rowblock_waves_ab.txt 2>&1
timeout 600 python tools/r05_ab.py tail_order 2 > $O/r05a_tail_order_ab.txt 2>&1
( for i in 1 2; do R05_LABEL="lds strides round 5" timeout 200 python tools/r05_ab.py once; R05_LABEL="lds strides round 4" INTERDIFF_HIP_LIB=build_ab/strides4/libinterdiff_hip.so timeout 200 python tools/r05_ab.py once; done ) > $O/r05a_lds_strides_ab.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-postopt 2> $O/r05a_bench.log | tail -1 > $O/r05a_bench.json
tail -3 $O/r05a_pytest_rowblock8.log; tail -3 $O/r05a_pytest.log; tail -12 $O/r05a_exclusive_cu.txt; grep -c sample $O/r05a_tail_order_ab.txt
