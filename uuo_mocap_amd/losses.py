"""Loss operators with the reference's signatures (reference src/video_mocap/losses/chamfer_distance.py:5-21,
losses/losses.py:43-51)."""
from __future__ import annotations

import torch

from . import _lib
from .engine import _f32, _ptr, check, current_stream


class _Knn1(torch.autograd.Function):
    """K=1 nearest neighbour on the GPU (uuo_nn_argmin).  Backward = pytorch3d's knn backward:
    g = 2*grad*(x - y[idx]); +g to x, -g scattered to y[idx]."""

    @staticmethod
    def forward(ctx, x, y):
        lib = _lib.load()
        xd, yd = _f32(x, "x"), _f32(y, "y")
        N, P1, P2 = xd.shape[0], xd.shape[1], yd.shape[1]
        dist = torch.empty((N, P1), dtype=torch.float32, device=xd.device)
        idx = torch.empty((N, P1), dtype=torch.int32, device=xd.device)
        ws = torch.empty((max(N * P1, 1),), dtype=torch.int64, device=xd.device)
        with torch.cuda.device(xd.device):
            check(lib.uuo_nn_argmin(current_stream(xd.device), N, P1, P2, _ptr(xd), _ptr(yd), None, 0, _ptr(dist),
                                    _ptr(idx), _ptr(ws)), "uuo_nn_argmin")
        idx64 = idx.long()
        ctx.save_for_backward(xd, yd, idx64)
        ctx.mark_non_differentiable(idx64)
        return dist, idx64

    @staticmethod
    def backward(ctx, grad_dist, _):
        x, y, idx = ctx.saved_tensors
        gidx = idx[..., None].expand(-1, -1, 3)
        g = 2.0 * grad_dist[..., None] * (x - torch.gather(y, 1, gidx))
        # accumulate with index_put_ (sort-based on the GPU, fixed summation order) rather than scatter_add_ (float
        # atomics): several x points may share a nearest y point, and the solves built on this operator must be
        # reproducible run to run
        rows = (torch.arange(y.shape[0], device=y.device)[:, None] * y.shape[1] + idx).reshape(-1)
        gy = torch.zeros((y.shape[0] * y.shape[1], 3), dtype=y.dtype, device=y.device)
        gy.index_put_((rows,), -g.reshape(-1, 3), accumulate=True)
        return g, gy.view_as(y)


def knn_points_k1(x: torch.Tensor, y: torch.Tensor):
    """(squared distances [N,P1], indices [N,P1] int64) of each x point's nearest y point (first index on ties)."""
    return _Knn1.apply(x, y)


def chamfer_distance(x, y, weights=None, single_directional: bool = False):
    """pytorch3d.loss.chamfer_distance with its defaults (mean/mean, squared L2), as the reference calls it at
    markers/markers_utils.py:471-475,575-579."""
    N = x.shape[0]

    def one_way(a, b):
        d, _ = knn_points_k1(a, b)
        if weights is not None:
            if weights.sum() == 0.0:
                return (a.sum((1, 2)) * weights).sum() * 0.0
            d = d * weights.view(N, 1)
        d = d.sum(1) / float(max(a.shape[1], 1))
        div = weights.sum() if weights is not None else max(N, 1)
        return d.sum() / div

    cham = one_way(x, y)
    if not single_directional:
        cham = cham + one_way(y, x)
    return cham, None


class _SoftMin(torch.autograd.Function):
    """EXTENSION (not in the reference): soft-min of the squared distances to a cloud, uuo_soft_nn_forward/backward."""

    @staticmethod
    def forward(ctx, x, y, tau):
        lib = _lib.load()
        xd, yd = _f32(x, "x"), _f32(y, "y")
        N, P1, P2 = xd.shape[0], xd.shape[1], yd.shape[1]
        soft = torch.empty((N, P1), dtype=torch.float32, device=xd.device)
        dmin = torch.empty_like(soft)
        sumexp = torch.empty_like(soft)
        ws = torch.empty((max(N * P1, 1),), dtype=torch.int64, device=xd.device)
        with torch.cuda.device(xd.device):
            check(lib.uuo_soft_nn_forward(current_stream(xd.device), N, P1, P2, _ptr(xd), _ptr(yd), float(tau), _ptr(soft),
                                          _ptr(dmin), _ptr(sumexp), _ptr(ws)), "uuo_soft_nn_forward")
        ctx.save_for_backward(xd, yd, dmin, sumexp)
        ctx.tau = float(tau)
        return soft

    @staticmethod
    def backward(ctx, grad_soft):
        x, y, dmin, sumexp = ctx.saved_tensors
        lib = _lib.load()
        g = _f32(grad_soft, "grad")
        N, P1, P2 = x.shape[0], x.shape[1], y.shape[1]
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(x.device):
            check(lib.uuo_soft_nn_backward(current_stream(x.device), N, P1, P2, _ptr(x), _ptr(y), ctx.tau, _ptr(dmin),
                                           _ptr(sumexp), _ptr(g), _ptr(gx), _ptr(gy)), "uuo_soft_nn_backward")
        return gx, gy, None


def soft_weighted_chamfer_distance(x: torch.Tensor, y: torch.Tensor, x_weights: torch.Tensor, tau: float):
    """EXTENSION, not reference behaviour: `weighted_chamfer_distance` with the hard minimum over the vertices replaced
    by the soft minimum  -tau log sum_j exp(-|x_i - y_j|^2 / tau)  (-> the hard term as tau -> 0).  Same normalisation."""
    d = _SoftMin.apply(x, y, tau)
    w = x_weights.to(d.dtype) if x_weights.dtype != d.dtype else x_weights
    wsum = x_weights.sum()
    if wsum == 0.0:
        return (x.sum() * 0.0), None
    return (d * w).sum() / wsum, None


def soft_chamfer_distance(x: torch.Tensor, y: torch.Tensor, tau: float):
    """EXTENSION, not reference behaviour: the one-directional `chamfer_distance(x, y, single_directional=True)` of the part
    stage (markers_utils.py:471-475: mean over the clouds of the mean over the points, unmasked) with the hard minimum over
    the candidate vertices replaced by the soft minimum -tau log sum_j exp(-|x_i - y_j|^2 / tau)."""
    d = _SoftMin.apply(x, y, tau)                      # [N, P1]
    return d.sum(1).div(float(max(x.shape[1], 1))).sum() / float(max(x.shape[0], 1)), None


def weighted_chamfer_distance(x: torch.Tensor, y: torch.Tensor, x_weights: torch.Tensor,
                              single_directional: bool = False):
    """sum_{n,i} w[n,i] * min_j |x[n,i]-y[n,j]|^2 / sum(w)  (the reference flattens to one cloud per marker and
    repeats y per marker -- 1.24 GB at F=300, M=50; here y is read in place).  Like the reference
    (chamfer_distance.py:19) the `single_directional` argument is ignored: always marker -> vertex."""
    d, _ = knn_points_k1(x, y)
    w = x_weights.to(d.dtype) if x_weights.dtype != d.dtype else x_weights
    wsum = x_weights.sum()
    if wsum == 0.0:
        return (x.sum() * 0.0), None
    return (d * w).sum() / wsum, None


def MarkerLoss(markers, virtual_markers, marker_weights, marker_distance):
    """[F,M] squared deviation of the marker-to-skin distance from `marker_distance`, masked."""
    gap = torch.norm(markers - virtual_markers, dim=-1) - marker_distance
    return gap ** 2 * marker_weights
