"""Cross-checks of the oracle's third-party restatements that share NO code with them.

The oracle's LBS / K=1 search / rotation conversions restate smplx and pytorch3d (absent here: SURVEY.md 8c) and the
golden fixtures were captured over those same restatements, so a transcription slip in ``oracle/smpl_ref.py`` or
``oracle/p3d_ref.py`` would be invisible to every other test.  Here each leaf is recomputed by a different route:

* LBS: float64 numpy written straight from the SMPL equations (Loper et al. 2015, eqs. 2-8), one vertex at a time,
  4x4 homogeneous joint transforms composed by explicit matrix products -- no einsum, no torch, no shared helper;
* K=1 nearest neighbour: ``torch.cdist`` + ``argmin`` in float64 on tie-free data, and scipy's cKDTree;
* rotation conversions: ``scipy.spatial.transform.Rotation`` and ``numpy.linalg.qr`` (Gram-Schmidt = thin QR with a
  positive diagonal).
"""
import numpy as np
import torch
from scipy.spatial import cKDTree
from scipy.spatial.transform import Rotation

from oracle import p3d_ref
from oracle.smpl_ref import SmplInferenceRef


def _rot(rng, *shape):
    return Rotation.from_rotvec(rng.normal(scale=0.7, size=shape + (3,)).reshape(-1, 3)).as_matrix().reshape(shape + (3, 3))


def _smpl_equations_f64(tables, rot, betas, trans, vertex_ids):
    """SMPL forward for the listed vertices of one frame, float64, per-vertex loop.
    rot [24,3,3] (root first), betas [10], trans [3] -> (verts [len(vertex_ids),3], joints24 [24,3])."""
    vt = tables.v_template.astype(np.float64)
    S = tables.shapedirs.astype(np.float64)              # [V,3,10]
    P = tables.posedirs.astype(np.float64)               # [207, V*3]
    Jr = tables.J_regressor.astype(np.float64)           # [24,V]
    W = tables.lbs_weights.astype(np.float64)            # [V,24]
    parents = [int(p) for p in tables.parents]
    V = vt.shape[0]
    # eq. 8/9: shape blend, then rest joints regressed from the shaped template (every vertex is needed for J)
    v_shaped = np.empty((V, 3))
    for v in range(V):
        v_shaped[v] = vt[v] + S[v] @ betas
    J = np.zeros((24, 3))
    for j in range(24):
        nz = np.nonzero(Jr[j])[0]
        for v in nz:
            J[j] += Jr[j, v] * v_shaped[v]
    # world transforms of the kinematic chain, 4x4 homogeneous
    G = [None] * 24
    for j in range(24):
        local = np.eye(4)
        local[:3, :3] = rot[j]
        local[:3, 3] = J[j] if j == 0 else J[j] - J[parents[j]]
        G[j] = local if j == 0 else G[parents[j]] @ local
    joints = np.stack([G[j][:3, 3] for j in range(24)]) + trans
    # remove the rest pose: G'_j = G_j * T(-J_j)
    Gp = []
    for j in range(24):
        rest = np.eye(4)
        rest[:3, 3] = -J[j]
        Gp.append(G[j] @ rest)
    # pose feature: vec(R_1 - I, ..., R_23 - I), row-major per joint
    pf = np.concatenate([(rot[j] - np.eye(3)).reshape(-1) for j in range(1, 24)])
    out = np.empty((len(vertex_ids), 3))
    for n, v in enumerate(vertex_ids):
        offset = np.array([pf @ P[:, 3 * v + c] for c in range(3)])
        vp = np.append(v_shaped[v] + offset, 1.0)
        T = np.zeros((4, 4))
        for j in range(24):
            if W[v, j] != 0.0:
                T += W[v, j] * Gp[j]
        out[n] = (T @ vp)[:3] + trans
    return out, joints


def test_lbs_matches_per_vertex_float64_equations(tables):
    rng = np.random.default_rng(42)
    F = 2
    rot = _rot(rng, F, 24)
    betas = rng.normal(size=(F, 10))
    trans = rng.normal(size=(F, 3))
    oracle = SmplInferenceRef(tables)
    out = oracle(poses=torch.from_numpy(rot[:, 1:]).float(), betas=torch.from_numpy(betas).float(),
                 root_orient=torch.from_numpy(rot[:, :1]).float(), trans=torch.from_numpy(trans).float())
    vids = np.concatenate([np.arange(0, 6890, 37), np.asarray(tables.extra_joint_vids)])
    for f in range(F):
        # the float32 inputs the oracle saw, promoted: the comparison isolates the arithmetic, not the rounding of inputs
        v64, j64 = _smpl_equations_f64(tables, rot[f].astype(np.float32).astype(np.float64),
                                       betas[f].astype(np.float32).astype(np.float64),
                                       trans[f].astype(np.float32).astype(np.float64), vids)
        np.testing.assert_allclose(out["vertices"][f, vids].numpy(), v64, atol=5e-6, rtol=0)
        np.testing.assert_allclose(out["joints"][f, :24].numpy(), j64, atol=5e-6, rtol=0)
        # the 21 extra joints are the listed vertices (VertexJointSelector), already translated
        np.testing.assert_allclose(out["joints"][f, 24:].numpy(), v64[-21:], atol=5e-6, rtol=0)


def test_generator_lbs_agrees_with_per_vertex_equations(tables):
    """uuo_mocap_amd.synthetic.lbs_f64 produces the ground truth of every synthetic sequence (and of bench.py's error
    figures): check that third implementation against the same equations."""
    from uuo_mocap_amd.synthetic import lbs_f64

    rng = np.random.default_rng(7)
    rot = _rot(rng, 1, 24)
    betas = rng.normal(size=(1, 10))
    trans = rng.normal(size=(1, 3))
    verts, joints, _ = lbs_f64(tables, rot, betas, trans)
    vids = np.arange(5, 6890, 111)
    v64, j64 = _smpl_equations_f64(tables, rot[0], betas[0], trans[0], vids)
    np.testing.assert_allclose(verts[0, vids], v64, atol=1e-12)
    np.testing.assert_allclose(joints[0], j64, atol=1e-12)


def test_knn_matches_cdist_and_kdtree_on_tie_free_data():
    rng = np.random.default_rng(3)
    N, P1, P2 = 6, 41, 6890
    x = rng.normal(size=(N, P1, 3)).astype(np.float32)
    y = rng.normal(size=(N, P2, 3)).astype(np.float32)
    d_loop, i_loop = p3d_ref.knn1_loop(x, y)
    nn = p3d_ref.knn_points(torch.from_numpy(x), torch.from_numpy(y), K=1)
    d64 = torch.cdist(torch.from_numpy(x).double(), torch.from_numpy(y).double())
    i_cdist = torch.argmin(d64, dim=-1).numpy()
    # tie-free in float64 by a margin far above fp32 rounding: the answers must coincide exactly
    srt = torch.sort(d64 ** 2, dim=-1)[0]
    assert float((srt[..., 1] - srt[..., 0]).min()) > 1e-5
    np.testing.assert_array_equal(i_loop, i_cdist)
    np.testing.assert_array_equal(nn.idx[..., 0].numpy(), i_cdist)
    np.testing.assert_allclose(d_loop, (d64 ** 2).min(dim=-1)[0].numpy(), rtol=2e-6)
    for n in range(N):
        dist, idx = cKDTree(y[n].astype(np.float64)).query(x[n].astype(np.float64), k=1)
        np.testing.assert_array_equal(i_loop[n], idx)
        np.testing.assert_allclose(d_loop[n], dist ** 2, rtol=2e-6)


def test_chamfer_distance_matches_direct_formula():
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.normal(size=(7, 9, 3)).astype(np.float32))
    y = torch.from_numpy(rng.normal(size=(7, 33, 3)).astype(np.float32))
    w = torch.from_numpy((rng.random(7) > 0.3).astype(np.float32))
    d2 = torch.cdist(x.double(), y.double()) ** 2
    fwd = (d2.min(dim=2)[0] * w[:, None].double()).sum(1) / 9
    bwd = (d2.min(dim=1)[0] * w[:, None].double()).sum(1) / 33
    single, _ = p3d_ref.chamfer_distance(x, y, weights=w, single_directional=True)
    both, _ = p3d_ref.chamfer_distance(x, y, weights=w, single_directional=False)
    np.testing.assert_allclose(single.item(), (fwd.sum() / w.sum()).item(), rtol=1e-5)
    np.testing.assert_allclose(both.item(), ((fwd.sum() + bwd.sum()) / w.sum()).item(), rtol=1e-5)
    plain, _ = p3d_ref.chamfer_distance(x, y, single_directional=True)
    np.testing.assert_allclose(plain.item(), (d2.min(dim=2)[0].sum(1) / 9).mean().item(), rtol=1e-5)


def test_axis_angle_and_quaternion_conversions_match_scipy():
    rng = np.random.default_rng(11)
    aa = rng.normal(scale=1.2, size=(200, 3))
    aa[0] = 0.0                      # small-angle branch
    aa[1] = [0.0, 0.0, 3e-7]
    aa[2] = [0.0, 0.0, 2.5]          # the reference's use: yaw about z (optimization.py:672-679)
    R_ref = Rotation.from_rotvec(aa).as_matrix()
    R = p3d_ref.axis_angle_to_matrix(torch.from_numpy(aa).float()).numpy()
    np.testing.assert_allclose(R, R_ref, atol=2e-6)
    # quaternion -> matrix for non-unit quaternions (two_s = 2 / |q|^2): scipy normalises, same rotation
    q = rng.normal(size=(100, 4))
    Rq = p3d_ref.quaternion_to_matrix(torch.from_numpy(q).float()).numpy()
    np.testing.assert_allclose(Rq, Rotation.from_quat(q[:, [1, 2, 3, 0]]).as_matrix(), atol=3e-6)
    # matrix -> quaternion (real part first), up to sign
    Rm = Rotation.from_rotvec(rng.normal(scale=2.0, size=(300, 3)))
    q_ours = p3d_ref.matrix_to_quaternion(torch.from_numpy(Rm.as_matrix()).float()).numpy()
    q_sp = Rm.as_quat()[:, [3, 0, 1, 2]]
    sign = np.sign(np.sum(q_ours * q_sp, axis=1, keepdims=True))
    np.testing.assert_allclose(q_ours * sign, q_sp, atol=3e-6)


def test_rotation_6d_is_gram_schmidt_of_the_first_two_rows():
    rng = np.random.default_rng(13)
    M = rng.normal(size=(150, 3, 3))
    ours = p3d_ref.rotation_6d_to_matrix(p3d_ref.matrix_to_rotation_6d(torch.from_numpy(M).float())).numpy()
    for n in range(M.shape[0]):
        q, r = np.linalg.qr(M[n, :2].T)          # columns = the two rows
        q = q * np.sign(np.diag(r))[None, :]     # Gram-Schmidt keeps the directions of the inputs
        b3 = np.cross(q[:, 0], q[:, 1])
        np.testing.assert_allclose(ours[n], np.stack([q[:, 0], q[:, 1], b3]), atol=2e-5)
    # a rotation is a fixed point
    Rm = Rotation.from_rotvec(rng.normal(size=(50, 3))).as_matrix()
    back = p3d_ref.rotation_6d_to_matrix(p3d_ref.matrix_to_rotation_6d(torch.from_numpy(Rm).float())).numpy()
    np.testing.assert_allclose(back, Rm, atol=2e-6)


def test_lbfgs_oracle_is_the_installed_torch_optimizer():
    """A9 is pinned by the real thing: the oracle drives its stages with torch.optim.LBFGS itself."""
    import inspect

    from oracle import stages_ref

    src = inspect.getsource(stages_ref._lbfgs)
    assert "torch.optim.LBFGS" in src and "strong_wolfe" in src
