"""How much do the end-to-end pose errors of BASELINE config #2 move when only the ROUNDING of the denoiser changes?  (test infrastructure, run as
``python -m tests.pose_error_spread`` on the GPU box; uses the oracle's fp64 twin like tests/test_hip_parity.py::test_full_size_end_to_end_golden.)

Every variant below computes the same denoiser to fp32 grade (forward 4-5e-7 from the fp64 answer, test_mdm_forward_split_f16_vs_exact_and_fp64); they differ
in the order / form of some roundings: the exact-fp32 feed-forward kernel with its 32-row tile and with its 16-row tile (two rounding classes, csrc/ffn.h),
split-f16 feed-forward + QKV with the exact row block, and split-f16 everywhere.  After 1000 steps, 11 corrections and the rot6d -> matrix -> SMPL
post-processing the distance of the final poses from the fp64 answer is a chaotic function of those roundings: this prints its realisations next to the
REFERENCE's own distance (the yardstick of the test's gates)."""
import json
import sys

import torch

from tests import fixtures as fx
from tests import test_hip_parity as tp


def main():
    from interdiff_amd import _lib
    from interdiff_amd.mdm import MDM
    from interdiff_amd.smpl import SMPL_Layer
    _lib.load()
    smpl = SMPL_Layer(fx.smpl_model(), device=tp.DEV)
    variants = [('exact, 32-row tile', dict(ffn_math='exact', ffn_rows=32)), ('exact, 16-row tile', dict(ffn_math='exact', ffn_rows=16)),
                ('exact, 64-row tile', dict(ffn_math='exact', ffn_rows=64)),
                ('split feed-forward + QKV, exact row block', dict(ffn_math='split', rowblock_math='exact')),
                ('split everywhere', dict(ffn_math='split', rowblock_math='split'))]
    keys = ('body_rotations', 'markers', 'joints', 'obj_rotation', 'obj_translation')
    out = []
    for name, kw in variants:
        m = MDM(fx.mdm_weights(), device=tp.DEV)
        for k, v in kw.items():
            setattr(m, k, v)
        rep, per_dump, fin, fin_same, rot_anchor = tp._full_size_report(m, smpl)
        row = dict(variant=name, final_sample_vs_fp64=per_dump['999']['hip_vs_fp64'], final_sample_vs_reference=per_dump['999']['hip_vs_reference'],
                   flips=rep['condition_flips_vs_reference'] + rep['contact_marker_flips_vs_reference'],
                   vs_fp64={k: rep['final_outputs_hip_vs_fp64'][k] for k in keys}, vs_reference={k: fin[k] for k in keys})
        out.append(row)
        print(json.dumps(row), flush=True)
    print(json.dumps(dict(variant='REFERENCE (its own fp32 run; the yardstick)', final_sample_vs_fp64=per_dump['999']['reference_vs_fp64'],
                          vs_fp64={k: rep['final_outputs_reference_vs_fp64'][k] for k in keys})), flush=True)


if __name__ == '__main__':
    torch.set_grad_enabled(False)
    sys.exit(main())
