#!/bin/bash
# library-level A/B on one box: product library vs a variant under build_ab/<name>/ (alternating, 3 rounds)
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out; O=gpurun_out
export PYTHONUNBUFFERED=1
V=${1:-preload}
: > $O/r05e_lib_ab_$V.txt
for rep in 1 2 3; do
  R05_LABEL="product" timeout 300 python tools/r05_ab.py once 2>&1 | grep -E "^sample|Error|error" >> $O/r05e_lib_ab_$V.txt
  R05_LABEL="$V" INTERDIFF_HIP_LIB=$PWD/build_ab/$V/libinterdiff_hip.so timeout 300 python tools/r05_ab.py once 2>&1 | grep -E "^sample|Error|error" >> $O/r05e_lib_ab_$V.txt
done
cat $O/r05e_lib_ab_$V.txt
