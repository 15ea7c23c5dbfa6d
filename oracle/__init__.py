"""CPU oracle for the InterDiff denoising-sampler hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``interdiff_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / reported baseline.

Every function restates (in plain torch-CPU / numpy arithmetic, fp32 with an
fp64 twin selected by the dtype of the inputs) what the reference computes on
the path ``eval_smpl_short.py`` drives, citing the reference file:line it
follows.  The restatement is pinned in two ways:

* against the reference's OWN source, imported read-only from
  ``/root/reference/interdiff`` through ``sys.modules`` shims by
  ``tests/golden/make_golden.py`` -> committed fixtures ``tests/golden/*.npz``
  (``tests/test_oracle_golden.py`` replays them without the reference);
* against closed-form identities and the schedule known-answers of
  SURVEY.md §8(c).

Third-party arithmetic that is NOT under /root/reference (pytorch3d 0.7.2
transforms, local-attention (unpinned), chamfer_distance (unlisted),
torchvision.ops.stochastic_depth) is restated from the published algorithms
in ``rotations.py`` / ``local_attn.py`` / ``geometry.py``: for those pieces
parity is UNPINNED (no reference test, no golden vector exists upstream) and
the restatement defines the contract; each sits behind a named switch.
"""
