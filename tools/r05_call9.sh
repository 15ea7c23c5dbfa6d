#!/bin/bash
# library-level A/B of several variants under build_ab/ against the product library, one box, alternating (2 rounds)
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out; O=gpurun_out/r05i_lib_ab_multi.txt
export PYTHONUNBUFFERED=1
: > $O
for rep in 1 2; do
  R05_LABEL="product" timeout 300 python tools/r05_ab.py once 2>&1 | grep -E "^sample|Error|error" >> $O
  for V in "$@"; do
    R05_LABEL="$V" INTERDIFF_HIP_LIB=$PWD/build_ab/$V/libinterdiff_hip.so timeout 300 python tools/r05_ab.py once 2>&1 | grep -E "^sample|Error|error" >> $O
  done
done
cat $O
