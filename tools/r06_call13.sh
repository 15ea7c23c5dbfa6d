#!/bin/bash
# contact scan: what bounds it?  instruction / LDS counters of the scan-order and the brute-force (identity-order) forms, separately
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
C1="SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
C2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES"
for v in scan_order identity_order; do
  tools/gpu_pmc.sh r06h_${v}_c1 "$C1" python tools/corr_bench.py --only $v > /dev/null
  tools/gpu_pmc.sh r06h_${v}_c2 "$C2" python tools/corr_bench.py --only $v > /dev/null
  grep "corr_contact" gpurun_out/r06h_${v}_c1_pmc.txt gpurun_out/r06h_${v}_c2_pmc.txt | cut -c1-200
done
python tools/corr_bench.py
