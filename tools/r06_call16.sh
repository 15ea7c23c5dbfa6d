#!/bin/bash
# final build: the round-6 measurement set (tools/profile_round_r06.sh) and the whole GPU suite on one box
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
tools/profile_round_r06.sh r06 > gpurun_out/r06_profile_round.log 2>&1
tail -3 gpurun_out/r06_profile_round.log
python -c "
import json
b=json.load(open('gpurun_out/r06_bench.json'))
print('ms_per_step', b['ms_per_step'], 'value', b['value'], 'frac', b['roofline']['frac'], 'step frac', b['step_roofline']['frac'], 'ffn us', b['roofline']['us_per_launch'])
print('cpu', {k: b['cpu_baseline'][k] for k in ('value','cores','kind')}, b['cpu_baseline'].get('headline_gpu_over_cpu'), b['cpu_baseline'].get('port_vs_reference'))
"
head -16 gpurun_out/r06_bench_kernel_stats.txt
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06_gpu_suite.txt
cat gpurun_out/r06_gpu_suite.txt
