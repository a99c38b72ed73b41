"""Pins the oracle against the known-answer values of SURVEY.md section 4 and the golden fixtures captured
from the reference's own leaf modules (oracle/make_golden.py)."""
import json
import os

import numpy as np
import torch

from oracle import p3d_ref, stages_ref
from oracle.stages_ref import MARKER_DISTANCE


def test_model_checksum_matches_golden(tables):
    from conftest import GOLDEN

    meta = json.load(open(os.path.join(GOLDEN, "meta.json")))
    assert str(tables.checksum()) == meta["model_checksum"]


def test_kat_a_weighted_chamfer(golden):
    g = golden("kats.npz")
    x = torch.arange(0, 16 * 7 * 3).float().reshape(16, 7, 3)
    y = torch.arange(0, 16 * 19 * 3).float().reshape(16, 19, 3)
    w = torch.ones(16, 7)
    w[:, ::2] = 0
    loss, aux = stages_ref.weighted_chamfer_distance(x, y, w)
    assert aux is None
    assert loss.item() == 287035.3125
    assert loss.item() == float(g["ka_loss"])
    nn = p3d_ref.knn_points(x, y, K=1)
    assert nn.idx[0, :, 0].tolist() == [0, 1, 2, 3, 4, 5, 6]
    assert nn.idx[15, :, 0].tolist() == [0] * 7


def test_kat_b_marker_loss(golden):
    g = golden("kats.npz")
    m = torch.tensor([[[0, 0, 0], [1, 2, 2]], [[0, 3, 4], [0, 0, 0]]]).float()
    vm = torch.tensor([[[1, 0, 0], [1, 2, 2.0095]], [[0, 0, 0], [5, 5, 5]]]).float()
    out = stages_ref.marker_loss(m, vm, stages_ref.get_marker_mask(m), MARKER_DISTANCE)
    np.testing.assert_array_equal(out.numpy(), g["kb"])
    assert out.mean().item() == 6.2262725830078125


def test_knn_vectorised_equals_c_loop():
    rng = np.random.default_rng(0)
    p1 = rng.standard_normal((5, 17, 3)).astype(np.float32)
    p2 = rng.standard_normal((5, 301, 3)).astype(np.float32)
    p2[:, 100] = p2[:, 7]  # exact duplicates -> first index must win
    p1[:, 0] = p2[:, 7]
    d_c, i_c = p3d_ref.knn1_loop(p1, p2)
    nn = p3d_ref.knn_points(torch.from_numpy(p1), torch.from_numpy(p2), K=1)
    np.testing.assert_array_equal(nn.idx[..., 0].numpy(), i_c)
    np.testing.assert_array_equal(nn.dists[..., 0].numpy(), d_c)
    assert (i_c[:, 0] == 7).all()


def test_knn_backward_matches_autograd_of_gather():
    torch.manual_seed(0)
    p1 = torch.randn(3, 5, 3, requires_grad=True)
    p2 = torch.randn(3, 40, 3, requires_grad=True)
    nn = p3d_ref.knn_points(p1, p2, K=1)
    (nn.dists.sum() * 1.5).backward()
    g1, g2 = p1.grad.clone(), p2.grad.clone()
    p1.grad = p2.grad = None
    idx = nn.idx[..., 0]
    sel = torch.gather(p2, 1, idx[..., None].expand(-1, -1, 3))
    (((p1 - sel) ** 2).sum() * 1.5).backward()
    torch.testing.assert_close(g1, p1.grad)
    torch.testing.assert_close(g2, p2.grad)


def test_rotation_6d_roundtrip_and_rz():
    torch.manual_seed(1)
    r = p3d_ref.rotation_6d_to_matrix(torch.randn(10, 6))
    eye = torch.eye(3).expand(10, 3, 3)
    torch.testing.assert_close(r @ r.transpose(1, 2), eye, atol=1e-5, rtol=0)
    torch.testing.assert_close(p3d_ref.rotation_6d_to_matrix(p3d_ref.matrix_to_rotation_6d(r)), r, atol=1e-6, rtol=0)
    th = torch.tensor([[0.0], [0.3], [-2.0], [1e-7]])
    rz = stages_ref.compute_root_orient_z(th)
    expect = torch.stack([torch.stack([torch.cos(th[:, 0]), -torch.sin(th[:, 0]), torch.zeros(4)], -1),
                          torch.stack([torch.sin(th[:, 0]), torch.cos(th[:, 0]), torch.zeros(4)], -1),
                          torch.tensor([[0.0, 0.0, 1.0]]).expand(4, 3)], -2)
    torch.testing.assert_close(rz, expect, atol=1e-6, rtol=0)


def test_numpy_semantics_k_e():
    """np.mean(axis=0) == sequential fp32 accumulate / F ; np.linalg.norm == sqrt((x0^2+x1^2)+x2^2) (bit-equal)."""
    rng = np.random.default_rng(3)
    a = rng.standard_normal((37, 5, 64, 3)).astype(np.float32)
    n = np.linalg.norm(a, axis=-1)
    manual = np.sqrt((a[..., 0] * a[..., 0] + a[..., 1] * a[..., 1]) + a[..., 2] * a[..., 2])
    np.testing.assert_array_equal(n, manual)
    acc = np.zeros(n.shape[1:], dtype=np.float32)
    for f in range(n.shape[0]):
        acc = acc + n[f]
    np.testing.assert_array_equal(np.mean(n, axis=0), acc / np.float32(n.shape[0]))


def test_sub_hierarchies_match_reference(tables, golden):
    g = golden("kats.npz")
    for k, n_all, n_kept in g["sub_counts"]:
        s = stages_ref.get_sub_hierarchies(tables.parents, int(k))
        assert len(s) == n_all
        assert len(stages_ref.remove_approximately_redundant_hierarchies(s, 0.9)) == n_kept
    s5 = stages_ref.get_sub_hierarchies(tables.parents, 5)
    np.testing.assert_array_equal(np.array(s5), g["sub5"])
    np.testing.assert_array_equal(np.array(stages_ref.remove_approximately_redundant_hierarchies(s5)), g["sub5_pruned"])


def test_config_loader_matches_reference():
    from conftest import GOLDEN
    from uuo_mocap_amd.config import packaged_config

    ref = json.load(open(os.path.join(GOLDEN, "configs.json")))
    for name, cfg in ref.items():
        assert packaged_config(name) == cfg, name
