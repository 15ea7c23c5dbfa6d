#!/bin/bash
# Round 5, second GPU call: phase stamps of the eight-wave row block, rocprofv3 kernel stats + the SQ counter pass on the current build, the three LDS stride sets,
# the hook stages beside a LONG f16-MFMA aggressor, the GPU suite without -x.  -> gpurun_out/r05b_*
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out; O=gpurun_out
export PYTHONUNBUFFERED=1
( echo "== eight waves (rowblock8_kernel)"; timeout 60 build_tools/rowblock_probe 16 100 1; echo "== four waves (round 4's kernel)"; timeout 60 build_tools/rowblock_probe 16 100 1 1 ) > $O/r05b_rowblock_probe.txt 2>&1
INTERDIFF_CHAINS=1 IDF_STEP_MARKER="ln_linear_h2_kernel<1>" tools/gpu_prof.sh r05b_bench python bench.py --no-cpu-baseline --no-kernel-profile --no-postopt --no-extra-configs > /dev/null 2>&1
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
tools/gpu_pmc.sh r05b_sq_kbench "$SQ" python tools/kbench.py --reps 3 > /dev/null 2>&1
( for i in 1 2; do for v in 5 4 6; do lib=build_ab/strides$v/libinterdiff_hip.so; [ $v = 5 ] && lib=interdiff_amd/csrc/libinterdiff_hip.so; R05_LABEL="lds stride set $v" INTERDIFF_HIP_LIB=$lib timeout 200 python tools/r05_ab.py once; done; done ) > $O/r05b_lds_strides_ab.txt 2>&1
timeout 300 python tools/hook_stage_probe.py aggr 4 > $O/r05b_hook_stage_victims.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q -k "not well_conditioned" > $O/r05b_pytest.log 2>&1; echo "pytest rc $?" >> $O/r05b_pytest.log
tail -5 $O/r05b_pytest.log; cat $O/r05b_rowblock_probe.txt | head -30
