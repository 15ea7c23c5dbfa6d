#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round_r06.sh <round-tag>   -> gpurun_out/<tag>_*
# Round 6's measurement set: rocprofv3 kernel stats of the bench command (INTERDIFF_CHAINS=1: the per-launch table of one chain; and the default form), the counter passes on the kernel-level
# bench and on the correction hook (each its own rocprofv3 run: --kernel-trace + --pmc only), profiles/traffic.json from them (tools/make_traffic_json.py) and THEN the bench line, which reads
# that file back as its recorded values: the line and the profile it quotes come from one box and one build.
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"; mkdir -p gpurun_out
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
SQV="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
INTERDIFF_CHAINS=1 IDF_STEP_MARKER="ln_linear_h2_kernel<1," tools/gpu_prof.sh ${tag}_bench python bench.py --no-cpu-baseline --no-kernel-profile --no-postopt --no-extra-configs > /dev/null
IDF_STEP_MARKER="ln_linear_h2_kernel<1," tools/gpu_prof.sh ${tag}_bench_two_chains python bench.py --no-cpu-baseline --no-kernel-profile --no-postopt --no-extra-configs > /dev/null
tools/gpu_pmc.sh ${tag}_fetch_size_kbench FETCH_SIZE python tools/kbench.py --reps 3 > /dev/null
tools/gpu_pmc.sh ${tag}_write_size_kbench WRITE_SIZE python tools/kbench.py --reps 3 > /dev/null
tools/gpu_pmc.sh ${tag}_sq_kbench "$SQ" python tools/kbench.py --reps 3 > /dev/null
tools/gpu_pmc.sh ${tag}_sq_corr "$SQV" python tools/corr_bench.py > /dev/null
python tools/make_traffic_json.py ${tag} > gpurun_out/${tag}_traffic.log 2>&1
timeout 900 python bench.py 2> gpurun_out/${tag}_bench.log | tail -1 > gpurun_out/${tag}_bench.json
timeout 120 python tools/corr_bench.py > gpurun_out/${tag}_corr_bench.txt 2>&1
timeout 120 python tools/contact_probe.py > gpurun_out/${tag}_contact_probe.txt 2>&1
python -c "
from interdiff_amd import _lib
t, bad = _lib.exclusive_cu_report(); print(t); print('not exclusive:', bad)" > gpurun_out/${tag}_exclusive_cu_report.txt 2>&1
ls -la gpurun_out | tail -30
