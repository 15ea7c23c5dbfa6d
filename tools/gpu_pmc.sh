#!/bin/bash
# usage: tools/gpu_pmc.sh <tag> <counter-list> <command...>  (on the GPU box) -> gpurun_out/<tag>_pmc.db
# one rocprofv3 counter pass (kernel trace + --pmc only, as the pool requires)
tag=$1; ctr=$2; shift; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$root/gpurun_out"
export TMPDIR=/tmp
# rocprofv3 runs from /tmp: make repo-relative arguments (tools/x.py, build_tools/x, bench.py) absolute
args=()
for a in "$@"; do if [ -e "$root/$a" ] && [ "${a#/}" = "$a" ]; then args+=("$root/$a"); else args+=("$a"); fi; done
set -- "${args[@]}"
rm -rf /tmp/pmc_$tag
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$tag -o $tag -- "$@" ) > "$root/gpurun_out/${tag}_pmc_run.log" 2>&1
db=$(find /tmp/pmc_$tag -name "*.db" | head -1)
python "$root/tools/pmc_summary.py" "$db" > "$root/gpurun_out/${tag}_pmc.txt" 2>&1
head -40 "$root/gpurun_out/${tag}_pmc.txt"
