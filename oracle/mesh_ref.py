"""CPU restatement of the two third-party mesh primitives behind the reference's barycentric marker placement.
TEST INFRASTRUCTURE (see oracle/__init__.py).  **Parity unpinned for this file**: neither `igl` nor `trimesh` is
installed here and the reference holds no fixture for them, so these follow the packages' published
algorithms; the reference's own window / granularity / scatter logic around them IS pinned, by executing the
reference's compute_nearest_points over these functions (oracle/make_golden_placement.py).

* `signed_distance`  -- `igl.signed_distance(P, V, F)` as the reference consumes it (optimization.py:494-500: the
  absolute value of S, the face index I, the closest point C).  libigl's AABB query evaluates
  `point_simplex_squared_distance` per candidate triangle, which is the region test of C. Ericson, "Real-Time
  Collision Detection", section 5.1.5; here every triangle is a candidate (brute force, float64).  The sign is not
  reproduced (the reference discards it); S is returned non-negative.
* `points_to_barycentric` -- `trimesh.triangles.points_to_barycentric(triangles, points)` (method "cramer").
"""
from __future__ import annotations

import numpy as np


def closest_on_triangles(p: np.ndarray, tri: np.ndarray):
    """p [3], tri [T,3,3] -> (closest [T,3], squared distance [T]) by Ericson's region test, vectorised over T."""
    a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
    ab, ac, ap = b - a, c - a, p[None] - a
    d1, d2 = np.einsum("ij,ij->i", ab, ap), np.einsum("ij,ij->i", ac, ap)
    bp = p[None] - b
    d3, d4 = np.einsum("ij,ij->i", ab, bp), np.einsum("ij,ij->i", ac, bp)
    cp = p[None] - c
    d5, d6 = np.einsum("ij,ij->i", ab, cp), np.einsum("ij,ij->i", ac, cp)
    vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
    with np.errstate(divide="ignore", invalid="ignore"):
        denom = 1.0 / (va + vb + vc)
        v, w = vb * denom, vc * denom                                    # interior
        e_bc = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        e_ac = d2 / (d2 - d6)
        e_ab = d1 / (d1 - d3)
    # later assignments take precedence: apply the regions in REVERSE of Ericson's if-chain
    m = (va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0)
    v, w = np.where(m, 1.0 - e_bc, v), np.where(m, e_bc, w)
    m = (vb <= 0) & (d2 >= 0) & (d6 <= 0)
    v, w = np.where(m, 0.0, v), np.where(m, e_ac, w)
    m = (d6 >= 0) & (d5 <= d6)
    v, w = np.where(m, 0.0, v), np.where(m, 1.0, w)
    m = (vc <= 0) & (d1 >= 0) & (d3 <= 0)
    v, w = np.where(m, e_ab, v), np.where(m, 0.0, w)
    m = (d3 >= 0) & (d4 <= d3)
    v, w = np.where(m, 1.0, v), np.where(m, 0.0, w)
    m = (d1 <= 0) & (d2 <= 0)
    v, w = np.where(m, 0.0, v), np.where(m, 0.0, w)
    q = a + ab * v[:, None] + ac * w[:, None]
    d2q = np.sum((p[None] - q) ** 2, axis=-1)
    return q, np.where(np.isnan(d2q), np.inf, d2q)


def signed_distance(P, V, F):
    """(|S| [N], I [N] int32, C [N,3]) for query points P [N,3] against the mesh (V [nv,3], F [nf,3])."""
    P, V = np.asarray(P, np.float64), np.asarray(V, np.float64)
    tri = V[np.asarray(F, np.int64)]
    S = np.zeros(P.shape[0])
    I = np.zeros(P.shape[0], dtype=np.int32)
    C = np.zeros((P.shape[0], 3))
    for n in range(P.shape[0]):
        q, d2 = closest_on_triangles(P[n], tri)
        I[n] = int(np.argmin(d2))  # first minimum
        S[n] = np.sqrt(d2[I[n]])
        C[n] = q[I[n]]
    return S, I, C


def points_to_barycentric(triangles, points):
    """trimesh.triangles.points_to_barycentric, method "cramer": triangles [N,3,3], points [N,3] -> [N,3]."""
    triangles = np.asarray(triangles, np.float64)
    points = np.asarray(points, np.float64)
    edge = triangles[:, 1:] - triangles[:, :1]
    w = points - triangles[:, 0].reshape((-1, 3))
    dot00 = np.einsum("ij,ij->i", edge[:, 0], edge[:, 0])
    dot01 = np.einsum("ij,ij->i", edge[:, 0], edge[:, 1])
    dot02 = np.einsum("ij,ij->i", edge[:, 0], w)
    dot11 = np.einsum("ij,ij->i", edge[:, 1], edge[:, 1])
    dot12 = np.einsum("ij,ij->i", edge[:, 1], w)
    inv = 1.0 / (dot00 * dot11 - dot01 * dot01)
    out = np.zeros((len(triangles), 3))
    out[:, 2] = (dot00 * dot12 - dot01 * dot02) * inv
    out[:, 1] = (dot11 * dot02 - dot01 * dot12) * inv
    out[:, 0] = 1 - out[:, 1] - out[:, 2]
    return out


class TrimeshRef:
    """The two attributes of trimesh.Trimesh the reference touches (optimization.py:487-491,519)."""

    def __init__(self, vertices=None, faces=None, process=False, **kw):
        self.vertices = np.asarray(vertices)
        self.faces = np.asarray(faces)

    @property
    def triangles(self):
        return self.vertices[self.faces.astype(np.int64)]
