#!/bin/bash
# phase stamps of the row block on the final build: 8 / 16 tokens per workgroup, round 4's four-wave kernel
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; mkdir -p gpurun_out; O=gpurun_out/r05_rowblock_probe.txt
{ echo "tools/rowblock_probe.hip built with the library's flags (kernel-argument preloading included); stamps of thread 0, mean over the workgroups of the forward's last QaN row block."
  echo "== 8 tokens per workgroup (shipped at this shape), REPLAYED AS A GRAPH"; build_tools/rowblock_probe 16 100 1 0 8 1
  echo "== 8 tokens per workgroup, plain launches"; build_tools/rowblock_probe 16 100 1 0 8
  echo "== 16 tokens per workgroup"; build_tools/rowblock_probe 16 100 1 0 16
  echo "== TIMING EXPERIMENT (wrong results): 8 tokens, graph replay, the lo planes of the folded-score and P.VW fragments not loaded (56 of ~200 KB of operands per workgroup)"; build_tools/rowblock_probe_skiplo 16 100 1 0 8 1
  echo "== four waves (round 4's kernel)"; build_tools/rowblock_probe 16 100 1 1 16; } > $O 2>&1
cat $O | cut -c1-250
