#!/bin/bash
# copies the judged summaries of one measurement set (tools/profile_round_r06.sh on the GPU box -> gpurun_out/<tag>_*) into profiles/ under their committed names, then renders the tables
tag=${1:-r06}; cd "$(dirname "$0")/.."
g=gpurun_out; p=profiles
cp $g/${tag}_bench.json $p/${tag}_bench.json
cp $g/${tag}_bench_kernel_stats.txt $p/${tag}_kernel_stats_bench.txt
cp $g/${tag}_bench_two_chains_kernel_stats.txt $p/${tag}_kernel_stats_bench_two_chains.txt
for c in fetch_size_kbench write_size_kbench sq_kbench sq_corr; do cp $g/${tag}_${c}_pmc.txt $p/${tag}_pmc_${c}.txt; done
grep -v amdgpu.ids $g/${tag}_corr_bench.txt > $p/${tag}_corr_bench.txt
grep -v amdgpu.ids $g/${tag}_contact_probe.txt > $p/${tag}_contact_probe.txt
grep -v amdgpu.ids $g/${tag}_exclusive_cu_report.txt > $p/${tag}_exclusive_cu_report.txt
cp $g/${tag}_traffic.json $p/traffic.json
[ -f $g/${tag}_gpu_suite.txt ] && cp $g/${tag}_gpu_suite.txt $p/${tag}_gpu_suite.txt
[ -f $g/parity_${tag}.json ] && cp $g/parity_${tag}.json $p/parity_${tag}.json
python tools/render_tables.py && python tools/render_tables.py --check
