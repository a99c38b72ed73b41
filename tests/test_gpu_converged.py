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


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


class Solve:
    """One recorded ``LBFGS.step`` of the reference's run: constants, final parameters, losses."""

    def __init__(self, d, k, oracle_smpl):
        self.k, self.stage = k, str(d["solve_stage"][k])
        self.F, self.M = d["markers"].shape[:2]
        g = lambda name: d["s%d_%s" % (k, name)]  # noqa: E731
        has = lambda name: ("s%d_%s" % (k, name)) in d.files  # noqa: E731
        self.markers = _t(np.nan_to_num(d["markers"]).astype(np.float32))
        self.hmr_pose = _t(d["hmr_pose_body"])
        self.o_betas = _t(g("o_betas"))
        self.final = [_t(d["s%d_p%d" % (k, j)]) for j in range(4 if self.stage != "part" else 3)]
        self.loss_at_final, self.first_loss = float(d["loss_at_final"][k]), float(d["first_losses"][k])
        self.nn = g("nn").astype(np.int64) if has("nn") else None
        if self.stage in ("chamfer", "marker"):
            self.o_pose = self.hmr_pose if has("o_pose_body_is_hmr") else _t(g("o_pose_body"))
        if self.stage == "chamfer":
            self.root = _t(g("root_orient"))
        elif self.stage == "marker":
            self.placement = _t(g("placement_idx")).long()
            self.repeat = int(g("repeat"))
        else:
            assert has("pose_body_is_hmr")
            self.root = _t(g("root_orient"))
            self.marker_indices = _t(g("marker_indices")).long()
            vlabels = torch.argmax(oracle_smpl.get_lbs_weights(), dim=-1)
            self.vidx = torch.cat([(vlabels == int(j)).nonzero(as_tuple=True)[0] for j in g("subtree")], dim=0)
            assert self.vidx.numel() == int(g("n_vertex_indices"))
            self.msub = self.markers[:, self.marker_indices].contiguous()

    def lr(self, cfg):
        return 0.1 if self.stage == "chamfer" else 1.0   # optimization.py:181 / :324, markers_utils.py:433

    def problem(self, smpl, cfg, dev):
        from uuo_mocap_amd.engine import ChamferProblem, MarkerProblem, PartProblem

        if self.stage == "chamfer":
            return ChamferProblem(smpl, self.markers.to(dev), self.o_pose.to(dev), self.o_betas.to(dev), self.root.to(dev), cfg)
        if self.stage == "marker":
            return MarkerProblem(smpl, self.markers.to(dev), self.o_pose.to(dev), self.o_betas.to(dev),
                                 self.placement.to(dev), cfg)
        return PartProblem(smpl, self.msub.to(dev), self.hmr_pose.to(dev), self.o_betas.to(dev), self.root.to(dev),
                           self.vidx.to(dev), cfg)

    def oracle(self, params, oracle_smpl, cfg):
        """(loss, vertices the search ran on or None) of the reference-faithful dense closure at `params` (CPU)."""
        p = [q.clone() for q in params]
        with torch.no_grad():
            if self.stage == "chamfer":
                lo, out = stages_ref.chamfer_stage_loss(self.markers, p[3], self.o_pose, p[2], self.o_betas, self.root, p[0],
                                                        p[1], oracle_smpl, cfg)
                return float(lo), out["vertices"].numpy()
            if self.stage == "marker":
                one_hot = torch.zeros(self.M, 6890)
                one_hot[torch.arange(self.M), self.placement] = 1.0
                lo, _ = stages_ref.marker_stage_loss(self.markers, p[0], self.o_pose, p[1], self.o_betas, p[2], p[3], one_hot,
                                                     oracle_smpl, cfg)
                return float(lo), None
            lo, out, _ = stages_ref.part_stage_loss(self.msub, self.hmr_pose, p[2], self.o_betas, self.root, p[1], p[0],
                                                    self.vidx, oracle_smpl, cfg)
            return float(lo), out["vertices"][:, self.vidx].numpy()


def _solves(d, oracle_smpl):
    return [Solve(d, k, oracle_smpl) for k in range(int(d["n_solves"]))]


def _starts(solves):
    """The point every recorded solve started from, rebuilt from the records the way the reference's orchestrator hands it
    on (multimodal.py:462-574,609-677): chamfer / part solves start on the HMR pose, the marker median and the mean HMR
    shape (every fixture with chamfer solves is full-body, where the part fit's translation / shape are discarded,
    :372-375); a hypothesis' marker solve continues from its chamfer solve (root <- Rz(z) root, optimization.py:280-285); the
    final marker solve from the winning hypothesis' result, 6D-normalised, which is also its pose prior."""
    starts = {}
    last_chamfer = None
    for s in solves:
        med = torch.median(s.markers, dim=1)[0]
        if s.stage == "part":
            starts[s.k] = [torch.zeros(1, 1, 1), med.clone(), s.o_betas.clone()]
        elif s.stage == "chamfer":
            starts[s.k] = [med.clone(), torch.zeros(s.F, 1, 1), s.o_betas.clone(), s.hmr_pose.clone()]
            last_chamfer = s
        elif s.repeat == 0:
            c = last_chamfer
            root = stages_ref.compute_root_orient_z(c.final[1]) @ c.root
            starts[s.k] = [c.final[3].clone(), c.final[2].clone(), root, c.final[0].clone()]
        else:
            best = [m for m in solves if m.stage == "marker" and m.repeat == 0 and
                    torch.equal(stages_ref.normalize_rot(m.final[0]), s.o_pose)]
            assert len(best) >= 1, "no hypothesis' marker result matches the final stage's pose prior"
            m = best[0]
            starts[s.k] = [stages_ref.normalize_rot(m.final[0]), m.final[1].clone(), stages_ref.normalize_rot(m.final[2]),
                           m.final[3].clone()]
    return starts


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
    for k, stage, a, b, _, _ in ratios:
        if stage in ("part", "chamfer"):
            assert a == pytest.approx(b, rel=0.1), (case, k, stage, a, b)
