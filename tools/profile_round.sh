#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <round-tag>   -> gpurun_out/<tag>_*
# The measurement set behind DESIGN.md / profiles/: the bench line, rocprofv3 kernel stats of the same command, counter passes
# (each its own rocprofv3 run: --kernel-trace + --pmc only) on the kernel-level bench, the SMPL probe, the correction bench and the
# post-optimisation bench.
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"; mkdir -p gpurun_out
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
SQV="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
timeout 900 python bench.py 2> gpurun_out/${tag}_bench.log | tail -1 > gpurun_out/${tag}_bench.json
IDF_STEP_MARKER="gemm_kernel<32, 64" tools/gpu_prof.sh ${tag}_bench python bench.py --no-cpu-baseline --no-kernel-profile --no-postopt --no-extra-configs > /dev/null    # marker = the embedding GEMM, first launch of a step
tools/gpu_pmc.sh ${tag}_fetch_size_kbench FETCH_SIZE python tools/kbench.py --reps 3 > /dev/null
tools/gpu_pmc.sh ${tag}_write_size_kbench WRITE_SIZE python tools/kbench.py --reps 3 > /dev/null
tools/gpu_pmc.sh ${tag}_sq_kbench "$SQ" python tools/kbench.py --reps 3 > /dev/null
tools/gpu_pmc.sh ${tag}_fetch_size_smpl FETCH_SIZE build_tools/smpl_probe 1600 > /dev/null
tools/gpu_pmc.sh ${tag}_write_size_smpl WRITE_SIZE build_tools/smpl_probe 1600 > /dev/null
tools/gpu_pmc.sh ${tag}_sq_corr "$SQV" python tools/corr_bench.py > /dev/null
tools/gpu_prof.sh ${tag}_postopt python tools/opt_bench.py --reps 1 > /dev/null
tools/gpu_pmc.sh ${tag}_sq_postopt "$SQV" python tools/opt_bench.py --reps 1 > /dev/null
timeout 120 python tools/corr_bench.py > gpurun_out/${tag}_corr_bench.txt 2>&1
timeout 60 build_tools/smpl_probe 1600 > gpurun_out/${tag}_smpl_probe.txt 2>&1
ls -la gpurun_out | tail -30
