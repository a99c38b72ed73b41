"""CPU oracle for the SMPL-to-marker fitting hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / the timed CPU baseline.  ``uuo_mocap_amd`` never imports
this package; its operators raise if the HIP library is missing.

What is restated here: the reference is pure Python whose arithmetic
lives in un-vendored third-party packages that are absent from this container and
from the GPU box --

* ``smplx`` (git submodule, URL only, no pinned SHA: reference .gitmodules:4-6)
  -> :mod:`oracle.smpl_ref` restates ``smplx.lbs.lbs`` / ``batch_rigid_transform`` /
  ``SMPL.forward`` from the published algorithm, anchored on the reference call
  sites utils/smpl.py:22-27,39-45;
* ``pytorch3d`` (unpinned conda build for pytorch 2.0.1, 0.7.3+ for the
  ``single_directional`` kwarg: reference install.sh:5-7)
  -> :mod:`oracle.p3d_ref` restates ``knn_points`` (K=1, CPU loop semantics),
  ``chamfer_distance`` and the rotation transforms, anchored on
  losses/chamfer_distance.py:15-20, optimization.py:7-8,66-74,662-679;
* ``torch.optim.LBFGS`` is importable here and is used directly.

:mod:`oracle.stages_ref` restates the reference's own stage solvers
(optimization.py, markers/markers_utils.py, multimodal.py) on top of those.

PARITY PINNING.  The reference ships no tests, fixtures or golden vectors
(SURVEY.md F14), so the third-party numerics are *parity unpinned* beyond the
known-answer values of SURVEY.md section 4 (checked in tests/test_oracle_kats.py).
The *orchestration* (loss weights, parameter packing, L-BFGS options, stage order)
is pinned by golden fixtures captured from the reference's own Python modules
executed in the build container over these restated primitives
(``oracle/make_golden.py`` -> ``tests/golden/*.npz``).
"""
