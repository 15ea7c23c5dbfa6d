#!/bin/bash
# Store scope of the rows / slabs one kernel hands to the next (csrc/common.h IDF_WT_MODE): 0 system-scope write-through (shipped), 1 agent scope, 2 plain stores.
# Rebuilds the library on the GPU box for each mode and runs the same short bench; restores mode 0 at the end.  Output -> profiles/r04_wt_mode_ab.txt
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
line() { python bench.py --no-cpu-baseline --no-kernel-profile --no-postopt --no-extra-configs 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1',d['ms_per_step'],d['ms_per_step_samples']['all'],'forward_us',round(d['denoiser_forward']['us'],1) if 'denoiser_forward' in d else '')"; }
for mode in 0 2 1 0 2 1; do
  IDF_EXTRA_HIPCC_FLAGS="-DIDF_WT_MODE=$mode" python -m interdiff_amd.csrc.build --force > /dev/null 2>&1 || { echo "build failed mode $mode"; break; }
  line "wt_mode=$mode"
done
python -m interdiff_amd.csrc.build --force > /dev/null 2>&1
