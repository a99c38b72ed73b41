"""Trajectory-independent pins of the end-to-end fixtures (run with -m gpu on the MI355X box).

Two fp32 L-BFGS runs of one problem part after a few dozen evaluations (a flipped line-search branch),
so comparing what the reference's own ``multimodal_video_mocap`` converged to with what the HIP fit converged to needs wide
bands.  The OBJECTIVE at a given point does not depend on any trajectory.  ``oracle/make_golden_e2e.py`` records, for
every ``torch.optim.LBFGS.step`` of the reference's run (BASELINE ``configs[0]``, ``[1]``, ``[2]``, the reference side of
``[4]``, and the metric's own workload ``video_mocap.yaml`` at 300 x 50), the parameters the solve ended on, the loss of
the reference's closure AT that point, the nearest-vertex indices of that evaluation and the closure's constants.  Here:

* **forward**: the fused HIP closure evaluated at the reference's recorded final parameters reproduces its recorded loss
  (rtol 2e-5) and its nearest-vertex assignment (exactly; a difference is only accepted on a near-tie of the oracle's own
  vertices, as everywhere in this suite);
* **backward**: the HIP solver started from the reference's starting point of the same solve (checked: its first loss IS
  the reference's first loss) runs to ITS converged point, and the oracle's closure (the reference's dense formulation
  under torch autograd, CPU) evaluated at that point reproduces the loss the HIP solve ended on (rtol 2e-5);
* the two converged losses of every solve are reported (junit properties) and bounded.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import p3d_ref, stages_ref  # noqa: E402
from uuo_mocap_amd.config import packaged_config  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["e2e_config0", "e2e_config1", "e2e_hmr_part", "e2e_mht_rotation", "e2e_headline_300x50"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def smpl(tables, dev):
    from uuo_mocap_amd.smpl import SmplInference

    return SmplInference(dev, tables=tables)


def _load(case):
    path = os.path.join(GOLDEN, case + ".npz")
    if not os.path.isfile(path):
        pytest.skip("%s.npz not generated yet (oracle/make_golden_e2e.py)" % case)
    d = np.load(path, allow_pickle=False)
    if "loss_at_final" not in d.files:
        pytest.skip("%s.npz has no converged-point records" % case)
    return d


from converged_records import Solve, _solves, _starts  # noqa: E402


def _near_tie_only(nn_hip, nn_ref, queries, verts):
    """Assignments that differ from the reference's must be near-ties on the oracle's vertices (the HIP closure searched its
    own MFMA-computed vertices, <= 2e-5 m away)."""
    flips = np.argwhere(nn_hip != nn_ref)
    for f, m in flips:
        da = float(np.sum((queries[f, m] - verts[f, nn_hip[f, m]]).astype(np.float64) ** 2))
        db = float(np.sum((queries[f, m] - verts[f, nn_ref[f, m]]).astype(np.float64) ** 2))
        assert abs(da - db) <= 4.0 * np.sqrt(max(da, db)) * 4e-5 + 1e-9, (f, m, da, db)
    return len(flips)


@pytest.mark.parametrize("case", CASES)
def test_hip_closure_at_the_references_converged_points(case, smpl, oracle_smpl, dev, record_property):
    d = _load(case)
    cfg = packaged_config(str(d["yaml"]))
    solves = _solves(d, oracle_smpl)
    worst, flips_total, pairs_total = 0.0, 0, 0
    for s in solves:
        prob = s.problem(smpl, cfg, dev)
        x = prob.pack(*[p.to(dev) for p in s.final])
        loss, _, nn = prob.evaluate(x)
        rel = abs(loss - s.loss_at_final) / max(abs(s.loss_at_final), 1e-30)
        worst = max(worst, rel)
        assert rel <= 2e-5, (case, s.k, s.stage, loss, s.loss_at_final)
        if s.nn is not None:
            queries = (s.markers if s.stage == "chamfer" else s.msub).numpy()
            nn_ref = s.nn.reshape(queries.shape[0], queries.shape[1])
            nn_hip = nn.cpu().numpy()
            if not np.array_equal(nn_hip, nn_ref):
                lo, verts = s.oracle(s.final, oracle_smpl, cfg)       # (also: the oracle AT the reference's point)
                np.testing.assert_allclose(lo, s.loss_at_final, rtol=2e-5)
                flips_total += _near_tie_only(nn_hip, nn_ref, queries, verts)
            pairs_total += nn_ref.size
    record_property("converged_forward_worst_rel_%s" % case, worst)
    record_property("converged_forward_nn_flips_%s" % case, "%d of %d" % (flips_total, pairs_total))
    print("%s: %d solves, HIP closure at the reference's final points: worst relative loss difference %.2e, "
          "assignments differing %d of %d" % (case, len(solves), worst, flips_total, pairs_total))
    assert flips_total <= 1e-3 * max(pairs_total, 1)


@pytest.mark.parametrize("case", CASES)
def test_oracle_closure_at_the_hip_fits_converged_points(case, smpl, oracle_smpl, dev, record_property):
    d = _load(case)
    cfg = packaged_config(str(d["yaml"]))
    solves = _solves(d, oracle_smpl)
    starts = _starts(solves)
    opt = cfg["optimizer"]
    pick = solves if len(solves) <= 12 else solves[::5]   # hmr_part: every fifth of its 46 candidates (CPU oracle time)
    ratios = []
    for s in pick:
        prob = s.problem(smpl, cfg, dev)
        x = prob.pack(*[p.to(dev) for p in starts[s.k]])
        first, _, _ = prob.evaluate(x, want_nn=False)
        np.testing.assert_allclose(first, s.first_loss, rtol=2e-5, err_msg="%s solve %d: not the reference's start" % (case, s.k))
        st = prob.solve(x, max_iter=cfg["stages"][s.stage]["num_iters"], lr=s.lr(cfg), tolerance_grad=opt["tolerance_grad"],
                        tolerance_change=opt["tolerance_change"])
        assert st["stop_reason"] not in ("max_iter", "max_eval"), st
        final = [p.cpu() for p in prob.unpack(x)]
        at_final, _, _ = prob.evaluate(x, want_nn=False)           # the loss AT the point the solve returned
        lo, _ = s.oracle(final, oracle_smpl, cfg)
        assert abs(lo - at_final) <= 2e-5 * abs(at_final), (case, s.k, s.stage, lo, at_final, st)
        ratios.append((s.k, s.stage, at_final, s.loss_at_final, st["n_eval"], int(d["n_evals"][s.k])))
    for k, stage, a, b, ne, ne_ref in ratios:
        print("%s solve %d (%s): HIP converged at %.6g in %d evaluations, the reference at %.6g in %d" % (case, k, stage, a, ne, b, ne_ref))
    record_property("converged_losses_%s" % case, "; ".join("%d:%s %.6g/%.6g" % (k, st_, a, b) for k, st_, a, b, _, _ in ratios))
    # same start, same objective (pinned both ways above), both stopped on a tolerance: two local searches of a piecewise-smooth
    # objective may still stop on different plateaus (hard assignment); the data terms of the early stages agree closely
    # (marker solves are not bounded: the reference's own marker solve of a poor hypothesis can stop after a dozen evaluations
    # on a flat line search -- e2e_config0's solve 2: 0.137 after 13 evaluations, 0.044 here -- which says nothing about parity)
    for k, stage, a, b, _, _ in ratios:
        if stage in ("part", "chamfer"):
            assert a == pytest.approx(b, rel=0.1 if stage == "part" else 0.25), (case, k, stage, a, b)
