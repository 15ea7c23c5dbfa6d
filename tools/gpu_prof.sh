#!/bin/bash
# usage: tools/gpu_prof.sh <tag> <command...>   (on the GPU box)  -> gpurun_out/<tag>_kernel_stats.txt
# rocprofv3 kernel trace of <command>, summarised by tools/rocprof_summary.py (IDF_STEP_MARKER=<kernel name part> adds the per-position table)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$root/gpurun_out"
export TMPDIR=/tmp
# rocprofv3 runs from /tmp: make repo-relative arguments (tools/x.py, build_tools/x, bench.py) absolute
args=()
for a in "$@"; do if [ -e "$root/$a" ] && [ "${a#/}" = "$a" ]; then args+=("$root/$a"); else args+=("$a"); fi; done
set -- "${args[@]}"
rm -rf /tmp/prof_$tag
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o $tag -- "$@" ) > "$root/gpurun_out/${tag}_run.log" 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python "$root/tools/rocprof_summary.py" "$db" ${IDF_STEP_MARKER:+"$IDF_STEP_MARKER"} > "$root/gpurun_out/${tag}_kernel_stats.txt" 2>&1
head -40 "$root/gpurun_out/${tag}_kernel_stats.txt"
