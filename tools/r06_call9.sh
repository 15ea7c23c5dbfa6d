#!/bin/bash
# QKV kernels: sixteen waves (product build) vs eight waves (build_ab/qkv8), rocprofv3 in-situ averages of the same short bench + whole-sample timing, alternating
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out; O=gpurun_out/r06h_qkv_waves_ab.txt
export PYTHONUNBUFFERED=1
: > $O
for rep in 1 2; do
  for V in product qkv8; do
    if [ $V = product ]; then unset INTERDIFF_HIP_LIB; else export INTERDIFF_HIP_LIB=$PWD/build_ab/$V/libinterdiff_hip.so; fi
    INTERDIFF_CHAINS=1 tools/gpu_prof.sh r06h_$V python bench.py --no-cpu-baseline --no-kernel-profile --no-postopt --no-extra-configs > /dev/null 2>&1
    echo "== $V (rep $rep)" >> $O
    grep "ln_linear_h2_kernel\|self_attn_h2\|ffn_h2_kernel<2" gpurun_out/r06h_${V}_kernel_stats.txt >> $O
    R05_LABEL="$V" timeout 300 python tools/r05_ab.py once 2>&1 | grep -E "^sample|Error|error" >> $O
  done
done
unset INTERDIFF_HIP_LIB
cat $O
