"""Records of the reference's solves inside the end-to-end fixtures (oracle/make_golden_e2e.py): shared by the CPU tests
(oracle at the recorded points) and the GPU tests (HIP closure at the recorded points, oracle at the HIP solver's)."""
import numpy as np
import torch

from oracle import stages_ref


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


