#!/bin/bash
# where the runtime puts kernel arguments: HIP_FORCE_DEV_KERNARG 0 / 1 / unset, same box, alternating
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out; O=gpurun_out/r05f_dev_kernarg_ab.txt
export PYTHONUNBUFFERED=1
: > $O
for rep in 1 2; do
  for v in unset 0 1; do
    if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
    R05_LABEL="HIP_FORCE_DEV_KERNARG=$v" timeout 300 python tools/r05_ab.py once 2>&1 | grep -E "^sample|Error|error" >> $O
  done
done
unset HIP_FORCE_DEV_KERNARG
cat $O
echo "== probe, graph replay, HIP_FORCE_DEV_KERNARG=1"; HIP_FORCE_DEV_KERNARG=1 build_tools/rowblock_probe 16 100 1 0 8 1 | sed -n 2,14p
echo "== probe, graph replay, HIP_FORCE_DEV_KERNARG=0"; HIP_FORCE_DEV_KERNARG=0 build_tools/rowblock_probe 16 100 1 0 8 1 | sed -n 2,14p
