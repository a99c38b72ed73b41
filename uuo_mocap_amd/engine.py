"""Thin host layer over libuuo_hip.so: device model, per-sequence workspace and the three stage problems.

torch is used for device memory and the current HIP stream only; every numerical step of the fitted path
is a kernel behind the C ABI (include/uuo_hip.h).
"""
from __future__ import annotations

import ctypes
import itertools
import threading
from collections import OrderedDict
from ctypes import byref, c_float, c_void_p
from typing import Callable, Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import (EVAL_CALLBACK, UUO_STAGE_CHAMFER, UUO_STAGE_MARKER, UUO_STAGE_PART, UuoLbfgsOptions,
                   UuoLbfgsStats, UuoProblem, UuoReprojectionProblem, check)
from .body_model import SmplTables

MARKER_DISTANCE = 0.0095  # reference utils/settings.py:1

STOP_REASONS = ["max_iter", "max_eval", "tolerance_grad", "tolerance_change(step)", "tolerance_change(loss)",
                "directional_derivative", "initial_tolerance_grad"]


_tls = threading.local()


WORKSPACE_GROUP_STRIDE = 64  # slots per group


def set_workspace_slot(slot: int):
    """Selects which per-(F, M) workspace the calling thread uses.  Independent solves that run concurrently
    (one host thread + one HIP stream each, e.g. the yaw hypotheses of multimodal_video_mocap) must use
    different slots; a workspace is never shared by two solves in flight."""
    _tls.slot = int(slot)


def set_workspace_group(group: int):
    """Selects the block of workspace slots the calling thread's slots index into.  One group per sequence in
    flight (parallel.fit_many): two sequences fitted concurrently never meet in a workspace.  Worker threads do not
    inherit it: the code that spawns them passes `workspace_group()` on."""
    _tls.group = int(group)


def workspace_group() -> int:
    return getattr(_tls, "group", 0)


def workspace_slot() -> int:
    return workspace_group() * WORKSPACE_GROUP_STRIDE + getattr(_tls, "slot", 0)


_worker_streams: Dict = {}
_worker_streams_lock = threading.Lock()


def worker_streams(device, count: int, role: str = "") -> list:
    """`count` persistent side streams of `device` for the calling thread's workspace group and a role ("hypothesis",
    "subtree", "sequence").  Created once per process and reused by every fit: torch hands out stream handles round-robin
    from a pool of 32 and keeps a BLAS workspace (76 MB on this build) alive per stream it has seen, so fresh
    `torch.cuda.Stream()` objects per fit grow the process by up to 2.4 GB and let two live worker threads meet on one
    handle."""
    device = torch.device(device)
    out = []
    with _worker_streams_lock:
        for i in range(count):
            key = (device.index, workspace_group(), role, i)
            st = _worker_streams.get(key)
            if st is None:
                st = torch.cuda.Stream(device=device)
                _worker_streams[key] = st
            out.append(st)
    return out


_worker_pools: Dict = {}


def worker_pool(count: int, role: str = ""):
    """A persistent ThreadPoolExecutor of `count` threads for the calling thread's workspace group and a role.  Worker
    threads are kept for the life of the process: every NEW host thread that touches the GPU costs a BLAS handle (and its
    device workspace) and the runtime's per-thread state, which a dataset run would otherwise pay once per stage per
    sequence."""
    from concurrent.futures import ThreadPoolExecutor

    key = (workspace_group(), role, int(count))
    with _worker_streams_lock:
        pool = _worker_pools.get(key)
        if pool is None:
            pool = ThreadPoolExecutor(max_workers=int(count), thread_name_prefix="uuo-%s-g%d" % (role, key[0]))
            _worker_pools[key] = pool
    return pool


def _require_cuda(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU: the fitted path has no CPU implementation" % name)


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    _require_cuda(t, name)
    return t.detach().to(torch.float32).contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def current_stream(device) -> c_void_p:
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _FitHandle:
    """Owner of one uuo_fit_t.  Problems hold a reference for as long as they may launch on the workspace; the
    workspace is destroyed (uuo_fit_destroy: hipFree waits for the device) when the last reference goes."""

    live = 0  # workspaces currently allocated in this process (diagnostics / tests)

    def __init__(self, lib, ptr: c_void_p, device):
        self._lib, self.ptr, self._device = lib, ptr, device
        _FitHandle.live += 1

    def __del__(self):
        try:
            if self.ptr:
                with torch.cuda.device(self._device):
                    self._lib.uuo_fit_destroy(self.ptr)
                self.ptr = None
                _FitHandle.live -= 1
        except Exception:
            pass


class _BatchHandle:
    """Owner of one uuo_batch_t (B workspaces + the staging of a lock-step batch of independent solves)."""

    def __init__(self, lib, ptr: c_void_p, device, shape, capacity: int):
        self._lib, self.ptr, self._device, self.shape, self.capacity = lib, ptr, device, shape, capacity

    def __del__(self):
        try:
            if self.ptr:
                with torch.cuda.device(self._device):
                    self._lib.uuo_batch_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


def solve_batch(problems, xs, max_iter: int, lr: float = 1.0, tolerance_grad: float = 1e-7,
                tolerance_change: float = 1e-9, history_size: int = 100):
    """uuo_batch_solve: the independent problems `problems` (same stage, F, M; e.g. the candidate body parts of
    find_best_part_fits or the yaw hypotheses) solved in lock-step, one launch per kernel and round for all of them.
    Each xs[i] is updated in place; returns one statistics dict per problem.  Results are bit-identical to
    `problems[i].solve(xs[i], ...)` one after the other."""
    assert len(problems) == len(xs) and len(problems) > 0
    p0 = problems[0]
    model = p0.model
    for p, x in zip(problems, xs):
        assert p.stage == p0.stage and p.F == p0.F and p.M == p0.M and p.model is model
        assert x.is_cuda and x.dtype == torch.float32 and x.numel() == p.n and x.is_contiguous()
    nb = len(problems)
    handle = model.batch(p0.stage, p0.F, p0.M, nb)
    probs = (UuoProblem * nb)()
    for i, p in enumerate(problems):
        ctypes.memmove(ctypes.byref(probs[i]), ctypes.byref(p.problem), ctypes.sizeof(UuoProblem))
    ptrs = (c_void_p * nb)(*[x.data_ptr() for x in xs])
    opt = UuoLbfgsOptions(int(max_iter), int(history_size), float(lr), float(tolerance_grad), float(tolerance_change), 0, 0)
    stats = (UuoLbfgsStats * nb)()
    with torch.cuda.device(model.device):
        check(model.lib.uuo_batch_solve(handle.ptr, current_stream(model.device), probs, ptrs, nb, byref(opt), stats),
              "uuo_batch_solve")
    return [{"n_iter": st.n_iter, "n_eval": st.n_eval, "first_loss": st.first_loss, "final_loss": st.final_loss,
             "stop_reason": STOP_REASONS[st.stop_reason], "device_ms": st.device_ms, "driver": "batch"} for st in stats]


def part_scores_batch(problems, xs):
    """uuo_batch_part_scores: two-directional chamfer distance between the markers and each candidate's vertices at
    xs[i] -- the ranking score of find_best_part_fits -- for all candidates in one batched forward + one score kernel.
    Call after solve_batch of the same problems (same batch, shared pose-blend cache).  Returns a list of floats."""
    p0 = problems[0]
    model = p0.model
    nb = len(problems)
    handle = model.batch(p0.stage, p0.F, p0.M, nb)
    probs = (UuoProblem * nb)()
    for i, p in enumerate(problems):
        ctypes.memmove(ctypes.byref(probs[i]), ctypes.byref(p.problem), ctypes.sizeof(UuoProblem))
    ptrs = (c_void_p * nb)(*[x.data_ptr() for x in xs])
    scores = (c_float * nb)()
    with torch.cuda.device(model.device):
        check(model.lib.uuo_batch_part_scores(handle.ptr, current_stream(model.device), probs, ptrs, nb, scores),
              "uuo_batch_part_scores")
    return [float(v) for v in scores]


class DeviceModel:
    """Owns a uuo_model_t (device copies of the SMPL tables) and the per-(F, M) fit workspaces."""

    def __init__(self, tables: SmplTables, device: torch.device):
        if torch.device(device).type != "cuda":
            raise RuntimeError("DeviceModel needs a CUDA/HIP device (got %s); there is no CPU path" % device)
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.tables = tables
        self.V = int(tables.v_template.shape[0])
        handle = c_void_p()
        arrs = [np.ascontiguousarray(tables.v_template, np.float32), np.ascontiguousarray(tables.shapedirs, np.float32),
                np.ascontiguousarray(tables.posedirs, np.float32), np.ascontiguousarray(tables.J_regressor, np.float32),
                np.ascontiguousarray(tables.lbs_weights, np.float32), np.ascontiguousarray(tables.parents, np.int64),
                np.ascontiguousarray(tables.extra_joint_vids, np.int64)]
        with torch.cuda.device(self.device):
            check(self.lib.uuo_model_create(*[a.ctypes.data for a in arrs], self.V, byref(handle)), "uuo_model_create")
        self.handle = handle
        self._fits: "OrderedDict" = OrderedDict()   # (slot, F, M) -> _FitHandle, least recently used first
        self._batches: Dict = {}                    # (group, stage) -> _BatchHandle
        self._fits_lock = threading.Lock()

    #: sequence shapes (F, M) whose workspaces stay cached per slot.  A dataset of sequences of many different lengths
    #: fitted through one SmplInference would otherwise keep every (F, M) workspace it ever met (F*V*3 vertex floats +
    #: the optimiser's history each) until hipMalloc fails.
    MAX_SHAPES_PER_SLOT = 2

    def fit(self, F: int, M: int) -> "_FitHandle":
        """The calling thread's workspace for sequences of F frames x M markers (created on first use).  The returned
        handle keeps the workspace alive: the cache holds the MAX_SHAPES_PER_SLOT most recently used shapes of every
        slot, an evicted workspace is destroyed when the last problem that uses it is gone."""
        slot = workspace_slot()
        key = (slot, int(F), int(M))
        with self._fits_lock:
            h = self._fits.get(key)
            if h is not None:
                self._fits.move_to_end(key)
                return h
            ptr = c_void_p()
            with torch.cuda.device(self.device):
                check(self.lib.uuo_fit_create(self.handle, key[1], key[2], byref(ptr)), "uuo_fit_create")
            h = _FitHandle(self.lib, ptr, self.device)
            self._fits[key] = h
            same_slot = [k for k in self._fits if k[0] == slot]
            for k in same_slot[:max(0, len(same_slot) - self.MAX_SHAPES_PER_SLOT)]:
                del self._fits[k]  # the handle frees the workspace once no problem references it any more
            return h

    #: device memory a lock-step batch may reserve beyond what its problems need (the growth head-room of the part stage)
    BATCH_FLOOR_BYTES = 8 << 30

    def batch(self, stage: int, F: int, M: int, count: int) -> "_BatchHandle":
        """A lock-step batch (uuo_batch_t) able to step `count` problems of (stage, F, M) together; one per workspace group
        and stage is kept (re-created when the shape changes or more problems are needed)."""
        key = (workspace_group(), int(stage))
        with self._fits_lock:
            h = self._batches.get(key)
            if h is not None and h.shape == (int(F), int(M)) and h.capacity >= count:
                return h
            # small batches (the yaw hypotheses) exactly.  Candidate lists of the part stage: the count is data-dependent
            # (89 .. 202 for a 10-marker limb) and re-creating a batch costs ~100 ms (one pinned allocation, two events and
            # a 25-MB workspace per member), so a batch starts at 256 members and only ever grows, by halves
            if count <= 8:
                cap = max(int(count), 4)
            else:
                # ... within a memory budget: a member's workspace is ~83 KB per frame (the vertex buffer), so the
                # 256-member floor alone would reserve tens of GB for a handful of candidates of a 3000-frame sequence
                per_member = int(F) * (self.V * 12 + 1024) + (1 << 20)
                floor = max(64, min(256, self.BATCH_FLOOR_BYTES // per_member // 64 * 64))
                prev = h.capacity if h is not None and h.shape == (int(F), int(M)) else 0
                cap = max(floor, (int(count) + 63) // 64 * 64, (prev * 3 // 2 + 63) // 64 * 64 if prev else 0)
            if h is not None:
                del self._batches[key]
                h = None
            ptr = c_void_p()
            with torch.cuda.device(self.device):
                check(self.lib.uuo_batch_create(self.handle, int(stage), int(F), int(M), cap, byref(ptr)), "uuo_batch_create")
            h = _BatchHandle(self.lib, ptr, self.device, (int(F), int(M)), cap)
            self._batches[key] = h
            return h

    def cached_workspaces(self) -> int:
        with self._fits_lock:
            return len(self._fits)

    def close(self):
        with self._fits_lock:
            self._fits.clear()
            self._batches.clear()
        if self.handle:
            self.lib.uuo_model_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- operators -------------------------------------------------------------------------------
    def smpl_forward(self, poses, betas, root_orient, trans, want_joints: bool = True):
        """uuo_smpl_forward: poses [F,23,3,3], betas [1|F,10], root_orient [F,1,3,3], trans [F,3]|None."""
        poses = _f32(poses, "poses")
        betas = _f32(betas, "betas")
        root = _f32(root_orient, "root_orient")
        F = poses.shape[0]
        tr = _f32(trans, "trans") if trans is not None else None
        verts = torch.empty((F, self.V, 3), dtype=torch.float32, device=self.device)
        joints = torch.empty((F, 45, 3), dtype=torch.float32, device=self.device) if want_joints else None
        with torch.cuda.device(self.device):
            check(self.lib.uuo_smpl_forward(self.handle, current_stream(self.device), F, _ptr(poses), _ptr(betas),
                                            int(betas.shape[0]), _ptr(root), _ptr(tr), _ptr(verts), _ptr(joints)),
                  "uuo_smpl_forward")
        return verts, joints

    def smpl_backward(self, poses, betas, root_orient, trans, d_verts, d_joints):
        """uuo_smpl_backward: gradients of uuo_smpl_forward's outputs' upstream (d_verts [F,V,3] / d_joints [F,45,3],
        either may be None) with respect to poses [F,23,3,3], betas (its own shape), root_orient [F,1,3,3], trans."""
        poses = _f32(poses, "poses")
        betas_c = _f32(betas, "betas")
        root = _f32(root_orient, "root_orient")
        F = poses.shape[0]
        tr = _f32(trans, "trans") if trans is not None else None
        dv = _f32(d_verts, "d_verts") if d_verts is not None else None
        dj = _f32(d_joints, "d_joints") if d_joints is not None else None
        g_poses = torch.empty((F, 23, 3, 3), dtype=torch.float32, device=self.device)
        g_betas = torch.empty((F, 10), dtype=torch.float32, device=self.device)
        g_root = torch.empty((F, 1, 3, 3), dtype=torch.float32, device=self.device)
        g_trans = torch.empty((F, 3), dtype=torch.float32, device=self.device)
        scratch = torch.empty((F * 24,), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.uuo_smpl_backward(self.handle, current_stream(self.device), F, _ptr(poses), _ptr(betas_c),
                                             int(betas_c.shape[0]), _ptr(root), _ptr(tr), _ptr(dv), _ptr(dj),
                                             _ptr(g_poses), _ptr(g_betas), _ptr(g_root), _ptr(g_trans), _ptr(scratch)),
                  "uuo_smpl_backward")
        if betas_c.shape[0] == 1:
            g_betas = g_betas.sum(dim=0, keepdim=True)
        return g_poses, g_betas, g_root, g_trans

    def nn_argmin(self, x, y, y_subset=None):
        """uuo_nn_argmin: x [N,P1,3], y [N,P2,3] -> (dist [N,P1] fp32, idx [N,P1] int32)."""
        x = _f32(x, "x")
        y = _f32(y, "y")
        N, P1, P2 = x.shape[0], x.shape[1], y.shape[1]
        sub = y_subset.to(torch.int32).contiguous() if y_subset is not None else None
        dist = torch.empty((N, P1), dtype=torch.float32, device=self.device)
        idx = torch.empty((N, P1), dtype=torch.int32, device=self.device)
        ws = torch.empty((max(N * P1, 1),), dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.uuo_nn_argmin(current_stream(self.device), N, P1, P2, _ptr(x), _ptr(y), _ptr(sub),
                                         int(sub.numel()) if sub is not None else 0, _ptr(dist), _ptr(idx), _ptr(ws)),
                  "uuo_nn_argmin")
        return dist, idx

    def assign_mean_argmin(self, verts, markers, valid):
        """uuo_assign_mean_argmin -> idx [M] int32."""
        verts = _f32(verts, "verts")
        markers = _f32(markers, "markers")
        valid_u8 = valid.to(device=self.device, dtype=torch.uint8).contiguous()
        F, V, M = verts.shape[0], verts.shape[1], markers.shape[1]
        idx = torch.empty((M,), dtype=torch.int32, device=self.device)
        ws = torch.empty((M,), dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.uuo_assign_mean_argmin(current_stream(self.device), F, M, V, _ptr(verts), _ptr(markers),
                                                  _ptr(valid_u8), _ptr(idx), _ptr(ws)), "uuo_assign_mean_argmin")
        return idx


    def mesh_closest_points(self, verts, faces, points):
        """uuo_mesh_closest_points (see the module-level function; it needs no model tables)."""
        return mesh_closest_points(verts, faces, points)


def mesh_closest_points(verts, faces, points):
    """uuo_mesh_closest_points: closest point of points[f,m] on the mesh (verts[f], faces) ->
    (dist [F,M], face [F,M] int32, closest [F,M,3], barycentric [F,M,3])."""
    lib = _lib.load()
    verts = _f32(verts, "verts")
    points = _f32(points, "points")
    device = verts.device
    if device.type != "cuda":
        raise RuntimeError("mesh_closest_points needs a CUDA/HIP device (got %s); there is no CPU path" % device)
    faces = faces.to(device=device, dtype=torch.int32).contiguous()
    if verts.dim() != 3 or points.dim() != 3 or points.shape[0] != verts.shape[0] or faces.dim() != 2 or faces.shape[1] != 3:
        raise ValueError("mesh_closest_points: verts [F,V,3], faces [NF,3], points [F,M,3] expected")
    F, V, M, NF = verts.shape[0], verts.shape[1], points.shape[1], faces.shape[0]
    dist = torch.empty((F, M), dtype=torch.float32, device=device)
    face = torch.empty((F, M), dtype=torch.int32, device=device)
    closest = torch.empty((F, M, 3), dtype=torch.float32, device=device)
    bary = torch.empty((F, M, 3), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        check(lib.uuo_mesh_closest_points(current_stream(device), F, M, V, NF, _ptr(verts), _ptr(faces), _ptr(points),
                                          _ptr(dist), _ptr(face), _ptr(closest), _ptr(bary)), "uuo_mesh_closest_points")
    return dist, face, closest, bary


_POSE_CACHE_IDS = itertools.count(1)  # unique per PartProblem (thread-safe: itertools.count.__next__ is atomic in CPython)


class _StageProblem:
    """One L-BFGS problem of the fit on a flat device vector in the reference's parameter packing."""

    stage = -1

    def __init__(self, model: DeviceModel, markers, o_pose, o_betas, root, w_data, w_pose, w_betas,
                 assign=None, subset=None, own_workspace: bool = True):
        self.model = model
        self.lib = model.lib
        self.device = model.device
        self.markers = _f32(markers, "markers")
        self.F, self.M = int(self.markers.shape[0]), int(self.markers.shape[1])
        self.o_pose = _f32(o_pose, "o_pose_body").reshape(self.F, 23, 9)
        self.o_betas = _f32(o_betas, "o_betas").reshape(-1)[:10].contiguous()
        self.root = _f32(root, "root_orient").reshape(self.F, 9) if root is not None else None
        self.assign = assign.to(device=self.device, dtype=torch.int32).contiguous() if assign is not None else None
        self.subset = subset.to(device=self.device, dtype=torch.int32).contiguous() if subset is not None else None
        # the calling thread's (F, M) workspace, kept alive while this problem exists; a problem that is only ever solved
        # as a member of a lock-step batch (solve_batch: the batch owns its workspaces) does not need one
        self._fit_ref = model.fit(self.F, self.M) if own_workspace else None
        self.fit = self._fit_ref.ptr if own_workspace else None
        p = UuoProblem()
        p.stage, p.F, p.M = self.stage, self.F, self.M
        p.d_markers = self.markers.data_ptr()
        p.d_o_pose = self.o_pose.data_ptr()
        p.d_o_betas = self.o_betas.data_ptr()
        p.d_root = self.root.data_ptr() if self.root is not None else None
        p.d_assign = self.assign.data_ptr() if self.assign is not None else None
        p.d_subset = self.subset.data_ptr() if self.subset is not None else None
        p.n_subset = int(self.subset.numel()) if self.subset is not None else 0
        p.w_data, p.w_pose, p.w_betas = float(w_data), float(w_pose), float(w_betas)
        p.marker_distance = MARKER_DISTANCE
        self.problem = p
        self.n = int(self.lib.uuo_problem_num_params(byref(p)))

    def _need_workspace(self):
        if self.fit is None:  # created for a lock-step batch only: give it the thread's workspace on first standalone use
            self._fit_ref = self.model.fit(self.F, self.M)
            self.fit = self._fit_ref.ptr

    def evaluate(self, x: torch.Tensor, want_nn: bool = True):
        """One closure evaluation: (loss, flat grad, nn index [F,M] int32 | None)."""
        assert x.is_cuda and x.dtype == torch.float32 and x.numel() == self.n and x.is_contiguous()
        self._need_workspace()
        loss = torch.empty((1,), dtype=torch.float32, device=self.device)
        grad = torch.empty((self.n,), dtype=torch.float32, device=self.device)
        nn = None
        if want_nn and self.stage != UUO_STAGE_MARKER:
            nn = torch.empty((self.F, self.M), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.uuo_closure_eval(self.fit, current_stream(self.device), byref(self.problem), _ptr(x),
                                            _ptr(loss), _ptr(grad), _ptr(nn)), "uuo_closure_eval")
        return float(loss.item()), grad, nn

    def solve(self, x: torch.Tensor, max_iter: int, lr: float = 1.0, tolerance_grad: float = 1e-7,
              tolerance_change: float = 1e-9, history_size: int = 100,
              callback: Optional[Callable[[int, float], None]] = None,
              point_callback: Optional[Callable[[int, float, torch.Tensor], None]] = None) -> Dict:
        """torch.optim.LBFGS(..., line_search_fn="strong_wolfe").step(closure) on the device; x updated in place.
        `callback(i, loss)` runs after every closure evaluation; `point_callback(i, loss, x_eval)` additionally gets a
        host copy of the evaluated parameter vector (one device -> host copy per evaluation: iter_fn support)."""
        assert x.is_cuda and x.dtype == torch.float32 and x.numel() == self.n and x.is_contiguous()
        self._need_workspace()
        opt = UuoLbfgsOptions(int(max_iter), int(history_size), float(lr), float(tolerance_grad),
                              float(tolerance_change), 0, 0)
        stats = UuoLbfgsStats()
        stream = current_stream(self.device)

        def on_eval(user, i, loss, d_x_eval):
            if callback is not None:
                callback(i, loss)
            if point_callback is not None:
                host = torch.empty((self.n,), dtype=torch.float32)
                check(self.lib.uuo_copy_to_host(stream, d_x_eval, host.data_ptr(), self.n), "uuo_copy_to_host")
                point_callback(i, loss, host)

        cb = EVAL_CALLBACK(on_eval) if (callback is not None or point_callback is not None) else None
        with torch.cuda.device(self.device):
            check(self.lib.uuo_lbfgs_solve(self.fit, current_stream(self.device), byref(self.problem), _ptr(x),
                                           byref(opt), byref(stats), ctypes.cast(cb, c_void_p) if cb else None, None),
                  "uuo_lbfgs_solve")
        return {"n_iter": stats.n_iter, "n_eval": stats.n_eval, "first_loss": stats.first_loss,
                "final_loss": stats.final_loss, "stop_reason": STOP_REASONS[stats.stop_reason],
                "device_ms": stats.device_ms}

    def solve_shared(self, x: torch.Tensor, reducer, max_iter: int, lr: float = 1.0, tolerance_grad: float = 1e-7,
                     tolerance_change: float = 1e-9, history_size: int = 100,
                     callback: Optional[Callable[[int, float], None]] = None) -> Dict:
        """EXTENSION (BASELINE configs[3], not reference behaviour): this stage's problem on every rank of `reducer`'s
        process group solved as ONE joint L-BFGS problem whose shape vector (the 10 betas) is shared by all ranks' sequences,
        ON THE DEVICE SOLVER (uuo_lbfgs_solve_shared: the same driver, kernels and closures as `solve`).  What crosses the
        ranks goes through `reducer.gather_array` (one all_gather of 16 doubles per closure evaluation, one of 627 per
        iteration, one of 10 at the start -- rank 0's betas win); every rank reduces the gathered tables in rank order, so
        all ranks take the same decisions and end with bit-identical betas.  x is updated in place."""
        from ._lib import GATHER_FN, UuoShared

        assert x.is_cuda and x.dtype == torch.float32 and x.numel() == self.n and x.is_contiguous()
        self._need_workspace()
        opt = UuoLbfgsOptions(int(max_iter), int(history_size), float(lr), float(tolerance_grad),
                              float(tolerance_change), 0, 0)
        stats = UuoLbfgsStats()
        world = int(reducer.world)
        failure = []

        def gather(user, mine, n, out):
            try:
                reducer.gather_array(np.ctypeslib.as_array(mine, shape=(n,)), np.ctypeslib.as_array(out, shape=(world, n)))
                return 0
            except BaseException as exc:  # an exception must not unwind through the C driver
                failure.append(exc)
                return 5

        if hasattr(reducer, "native"):   # the node-local mailbox: the driver calls it directly, no Python on the path
            gather_c, user = reducer.native()
        else:
            gather_c, user = GATHER_FN(gather), None
        shared = UuoShared(gather_c, user, int(reducer.rank), world)
        cb = EVAL_CALLBACK(lambda user, i, loss, d_x_eval: callback(i, loss)) if callback is not None else None
        with torch.cuda.device(self.device):
            rc = self.lib.uuo_lbfgs_solve_shared(self.fit, current_stream(self.device), byref(self.problem), _ptr(x),
                                                 byref(opt), byref(stats), byref(shared),
                                                 ctypes.cast(cb, c_void_p) if cb else None, None)
        if failure:
            raise failure[0]
        check(rc, "uuo_lbfgs_solve_shared")
        return {"n_iter": stats.n_iter, "n_eval": stats.n_eval, "first_loss": stats.first_loss,
                "final_loss": stats.final_loss, "stop_reason": STOP_REASONS[stats.stop_reason],
                "device_ms": stats.device_ms, "driver": "device-lbfgs(shared betas, world=%d)" % world}

    def solve_shared_reference(self, x: torch.Tensor, reducer, max_iter: int, lr: float = 1.0, tolerance_grad: float = 1e-7,
                               tolerance_change: float = 1e-9, history_size: int = 100) -> Dict:
        """The joint problem of `solve_shared` on dist_lbfgs.ShardedLBFGS (a Python L-BFGS in coefficient space over the
        fused HIP closure of each rank; round 2's driver): kept as the CHECKER of the device route."""
        from .dist_lbfgs import ShardedLBFGS

        assert x.is_cuda and x.dtype == torch.float32 and x.numel() == self.n and x.is_contiguous()
        F = self.F
        off = {UUO_STAGE_CHAMFER: 4 * F, UUO_STAGE_MARKER: 207 * F, UUO_STAGE_PART: 3 * F + 1}[self.stage]
        idx = torch.arange(self.n, device=self.device)
        perm = torch.cat([idx[:off], idx[off + 10:], idx[off:off + 10]])  # own parameters first, the shared betas last
        xs = x[perm].clone()
        full = torch.empty_like(x)

        def evaluate(xl):
            full[perm] = xl
            loss, grad, _ = self.evaluate(full, want_nn=False)
            return loss, grad[perm]

        st = ShardedLBFGS(xs, 10, evaluate, reducer=reducer, lr=lr, max_iter=max_iter, tolerance_grad=tolerance_grad,
                          tolerance_change=tolerance_change, history_size=history_size).solve()
        x[perm] = xs
        st.update(device_ms=0.0, driver="sharded-lbfgs(world=%d)" % reducer.world)
        return st

    def solve_adam(self, x: torch.Tensor, num_steps: int, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
                   callback: Optional[Callable[[int, float], None]] = None) -> Dict:
        """EXTENSION, not reference behaviour (the reference only drives its closures with L-BFGS; BASELINE's north star
        also names Adam): `num_steps` steps of torch.optim.Adam on the flat parameter vector, every gradient from the
        fused HIP closure (uuo_closure_eval), the update itself a handful of element-wise device ops; x updated in place.
        No host synchronisation inside the loop unless a callback asks for the loss."""
        assert x.is_cuda and x.dtype == torch.float32 and x.numel() == self.n and x.is_contiguous()
        self._need_workspace()
        p = x.detach().requires_grad_(True)
        opt = torch.optim.Adam([p], lr=lr, betas=betas, eps=eps)
        loss = torch.empty((1,), dtype=torch.float32, device=self.device)
        grad = torch.empty((self.n,), dtype=torch.float32, device=self.device)
        first = last = None
        stream = current_stream(self.device)
        for i in range(int(num_steps)):
            with torch.cuda.device(self.device):
                check(self.lib.uuo_closure_eval(self.fit, stream, byref(self.problem), _ptr(p), _ptr(loss), _ptr(grad),
                                                None), "uuo_closure_eval")
            if callback is not None or i == 0 or i == num_steps - 1:
                last = float(loss.item())
                first = last if first is None else first
                if callback is not None:
                    callback(i, last)
            p.grad = grad
            opt.step()
        return {"n_iter": int(num_steps), "n_eval": int(num_steps), "first_loss": first, "final_loss": last,
                "stop_reason": "num_steps", "device_ms": 0.0, "driver": "adam"}

    def time_closure(self, x: torch.Tensor, iters: int = 20, dominant_only=False) -> float:
        """Device ms per closure evaluation (HIP events); dominant_only 1 / True: the fp32 skinning kernel alone (k_skin2), 2: the
        fp16-split skinning kernel of the chamfer closure's search (k_skin3)."""
        self._need_workspace()
        ms = c_float(0.0)
        with torch.cuda.device(self.device):
            check(self.lib.uuo_time_closure(self.fit, current_stream(self.device), byref(self.problem), _ptr(x),
                                            int(iters), int(dominant_only), byref(ms)), "uuo_time_closure")
        return float(ms.value)


def _cfg_weights(losses: Dict, data_key: str):
    return (float(losses.get(data_key, 0.0)), float(losses.get("reg_pose_body", 0.0)),
            float(losses.get("reg_betas", 0.0)))


class ChamferProblem(_StageProblem):
    """closure_stage_chamfer (reference optimization.py:187-275); x = [trans 3F | z F | betas 10 | pose 207F]."""

    stage = UUO_STAGE_CHAMFER

    def __init__(self, smpl_inference, markers, o_pose_body, o_betas, root_orient, config):
        losses = config["stages"]["chamfer"]["losses"]
        unsupported = set(losses) - {"full_chamfer", "reg_pose_body", "reg_betas", "soft_chamfer"}
        if unsupported:
            raise NotImplementedError("chamfer-stage losses outside the shipped configs: %s" % sorted(unsupported))
        if not config["stages"]["chamfer"]["yaw_lock"]:
            raise NotImplementedError("stages.chamfer.yaw_lock False is not a shipped configuration")
        wd, wp, wb = _cfg_weights(losses, "full_chamfer")
        super().__init__(smpl_inference.device_model, markers, o_pose_body, o_betas, root_orient, wd, wp, wb)
        # EXTENSION (not in the reference): soft assignment of every marker to the body's vertices, fused closure with the dense
        # backward on the matrix pipe (csrc/dense_bwd.hip); not available inside lock-step batches
        w_soft = float(losses.get("soft_chamfer", 0.0))
        if w_soft != 0.0:
            self.problem.w_soft = w_soft
            self.problem.soft_tau = float(config["stages"]["chamfer"].get("soft_tau", 1e-3))
            if not self.problem.soft_tau > 0.0:
                raise ValueError("stages.chamfer.soft_tau must be positive")

    def pack(self, trans, z_angle, betas, pose_body):
        return torch.cat([_f32(trans, "trans").reshape(-1), _f32(z_angle, "z").reshape(-1),
                          _f32(betas, "betas").reshape(-1), _f32(pose_body, "pose").reshape(-1)]).contiguous()

    def unpack(self, x):
        F = self.F
        return (x[:3 * F].reshape(F, 3), x[3 * F:4 * F].reshape(F, 1, 1), x[4 * F:4 * F + 10].reshape(1, 10),
                x[4 * F + 10:].reshape(F, 23, 3, 3))


class MarkerProblem(_StageProblem):
    """closure_stage_marker_pose (reference optimization.py:329-394); x = [pose 207F | betas 10 | root 9F | trans 3F]."""

    stage = UUO_STAGE_MARKER

    def __init__(self, smpl_inference, markers, o_pose_body, o_betas, assign, config, bary=None):
        """`assign` [M] vertex ids (the one-hot placement of the shipped configs), or -- with `bary` [M, 3] -- [M, 3] corner
        vertex ids of a three-corner (barycentric) placement: virtual marker m = sum_k bary[m, k] v[assign[m, k]]."""
        st = config["stages"]["marker"]
        unsupported = set(st["losses"]) - {"marker", "reg_pose_body", "reg_betas"}
        if unsupported:
            raise NotImplementedError("marker-stage losses outside the shipped configs: %s" % sorted(unsupported))
        if st.get("use_sdf"):
            raise NotImplementedError("stages.marker.use_sdf is off in every shipped config")
        wd, wp, wb = _cfg_weights(st["losses"], "marker")
        super().__init__(smpl_inference.device_model, markers, o_pose_body, o_betas, None, wd, wp, wb, assign=assign)
        if bary is not None:
            if self.assign.dim() != 2 or tuple(self.assign.shape) != (self.M, 3) or tuple(bary.shape) != (self.M, 3):
                raise ValueError("a three-corner placement takes assign [M, 3] and bary [M, 3]")
            self.bary = _f32(bary, "bary").to(self.device).contiguous()
            self.problem.n_corners = 3
            self.problem.d_bary = self.bary.data_ptr()
        elif self.assign.dim() != 1 or self.assign.numel() != self.M:
            raise ValueError("a one-hot placement takes assign [M]")

    def pack(self, pose_body, betas, root_orient, trans):
        return torch.cat([_f32(pose_body, "pose").reshape(-1), _f32(betas, "betas").reshape(-1),
                          _f32(root_orient, "root").reshape(-1), _f32(trans, "trans").reshape(-1)]).contiguous()

    def unpack(self, x):
        F = self.F
        return (x[:207 * F].reshape(F, 23, 3, 3), x[207 * F:207 * F + 10].reshape(1, 10),
                x[207 * F + 10:216 * F + 10].reshape(F, 1, 3, 3), x[216 * F + 10:].reshape(F, 3))


#: most markers per frame the fused soft-assignment part closure (k_part_soft) is instantiated for
PART_SOFT_MAX_MARKERS = 16


class PartProblem(_StageProblem):
    """closure_fit_subtree (reference markers/markers_utils.py:454-562); x = [z 1 | trans 3F | betas 10]."""

    stage = UUO_STAGE_PART

    def __init__(self, smpl_inference, markers, pose_body, o_betas, root_orient, vertex_indices, config,
                 own_workspace: bool = True):
        st = config["stages"]["part"]
        losses = st["losses"]
        unsupported = set(losses) - {"chamfer", "reg_betas", "soft_chamfer"}
        if unsupported:
            raise NotImplementedError("part-stage losses outside the shipped configs: %s" % sorted(unsupported))
        super().__init__(smpl_inference.device_model, markers, pose_body, o_betas, root_orient,
                         float(losses.get("chamfer", 0.0)), 0.0, float(losses.get("reg_betas", 0.0)),
                         subset=vertex_indices, own_workspace=own_workspace)
        # the body pose is a constant of this problem: let the library compute its pose-corrective blend once
        self.problem.pose_cache_id = next(_POSE_CACHE_IDS)
        # EXTENSION (not in the reference): soft assignment of every marker to the candidate's vertices, fused (k_part_soft)
        w_soft = float(losses.get("soft_chamfer", 0.0))
        if w_soft != 0.0:
            if self.M > PART_SOFT_MAX_MARKERS:
                raise NotImplementedError("the fused soft-assignment part closure takes at most %d markers per frame"
                                          % PART_SOFT_MAX_MARKERS)
            self.problem.w_soft = w_soft
            self.problem.soft_tau = float(st.get("soft_tau", 2.5e-4))
            if not self.problem.soft_tau > 0.0:
                raise ValueError("stages.part.soft_tau must be positive")

    def pack(self, z_angle, trans, betas):
        return torch.cat([_f32(z_angle, "z").reshape(-1), _f32(trans, "trans").reshape(-1),
                          _f32(betas, "betas").reshape(-1)]).contiguous()

    def unpack(self, x):
        F = self.F
        return x[:1].reshape(1, 1, 1), x[1:3 * F + 1].reshape(F, 3), x[3 * F + 1:].reshape(1, 10)


class ReprojectionProblem:
    """One yaw hypothesis of the 2D-prior fit (reference utils/hmr_utils.py:170-425) as the library's fused closure
    (uuo_reprojection_*, csrc/reprojection.hip).  x = [yaw 1 | body translation 3F (HMR axes) | camera translation 3 |
    betas 10 (detached in the reference: they never move)].  `joints0` [F,J,3] / `verts0` [F,V,3] come from ONE forward of
    the HMR pose with the solve's betas, the HMR root orientation and zero translation: nothing else of the body changes
    while this stage runs."""

    def __init__(self, markers: torch.Tensor, joints0: torch.Tensor, verts0: torch.Tensor, kp_target: torch.Tensor,
                 mask: torch.Tensor, focal, center, w_reprojection: float, w_chamfer: float):
        self.lib = _lib.load()
        self.device = markers.device
        # the handle stores raw pointers: the tensors live as long as the problem does
        self._keep = [_f32(t, n) for t, n in ((markers, "markers"), (joints0, "joints0"), (verts0, "verts0"),
                                               (kp_target, "kp_target"), (mask, "mask"))]
        mk, j0, v0, kp, ms = self._keep
        self.F, self.M, self.V, self.J = int(mk.shape[0]), int(mk.shape[1]), int(v0.shape[1]), int(j0.shape[1])
        if j0.shape != (self.F, self.J, 3) or v0.shape != (self.F, self.V, 3) or kp.shape != (self.F, self.J, 2) \
                or ms.shape != (self.F,) or mk.shape != (self.F, self.M, 3):
            raise ValueError("ReprojectionProblem: inconsistent shapes")
        p = UuoReprojectionProblem()
        p.F, p.M, p.V, p.J = self.F, self.M, self.V, self.J
        p.d_markers, p.d_joints0, p.d_verts0 = mk.data_ptr(), j0.data_ptr(), v0.data_ptr()
        p.d_kp_target, p.d_mask = kp.data_ptr(), ms.data_ptr()
        p.focal[0], p.focal[1] = float(focal[0]), float(focal[1])
        p.center[0], p.center[1] = float(center[0]), float(center[1])
        p.w_reprojection, p.w_chamfer = float(w_reprojection), float(w_chamfer)
        self.problem = p
        self.n = int(self.lib.uuo_reprojection_num_params(byref(p)))
        h = c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.uuo_reprojection_create(byref(p), byref(h)), "uuo_reprojection_create")
        self.handle = h

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            self.lib.uuo_reprojection_destroy(h)

    def evaluate(self, x: torch.Tensor, want_kp: bool = False, want_nn: bool = False):
        """One closure evaluation: (loss, flat gradient [3F+14], key points [F,J,2] | None, nearest vertex [F,M] | None)."""
        assert x.is_cuda and x.dtype == torch.float32 and x.numel() == self.n and x.is_contiguous()
        loss = torch.empty((1,), dtype=torch.float32, device=self.device)
        grad = torch.empty((self.n,), dtype=torch.float32, device=self.device)
        kp = torch.empty((self.F, self.J, 2), dtype=torch.float32, device=self.device) if want_kp else None
        nn = torch.empty((self.F, self.M), dtype=torch.int32, device=self.device) if want_nn else None
        with torch.cuda.device(self.device):
            check(self.lib.uuo_reprojection_eval(self.handle, current_stream(self.device), _ptr(x), _ptr(loss), _ptr(grad),
                                                 _ptr(kp), _ptr(nn)), "uuo_reprojection_eval")
        return float(loss.item()), grad, kp, nn

    def solve(self, x: torch.Tensor, max_iter: int, lr: float = 1.0, tolerance_grad: float = 1e-7,
              tolerance_change: float = 1e-9, history_size: int = 100,
              callback: Optional[Callable[[int, float], None]] = None,
              point_callback: Optional[Callable[[int, float, torch.Tensor], None]] = None) -> Dict:
        """torch.optim.LBFGS(..., line_search_fn="strong_wolfe").step(closure) at hmr_utils.py:367 on the device; x updated
        in place.  Returns the solver statistics plus `x_last` / `kp_last`: the parameter vector and the key points of the
        LAST closure evaluation (what the reference's outputs are derived from, :383-425)."""
        assert x.is_cuda and x.dtype == torch.float32 and x.numel() == self.n and x.is_contiguous()
        opt = UuoLbfgsOptions(int(max_iter), int(history_size), float(lr), float(tolerance_grad),
                              float(tolerance_change), 0, 0)
        stats = UuoLbfgsStats()
        x_last = x.clone()
        kp_last = torch.zeros((self.F, self.J, 2), dtype=torch.float32, device=self.device)
        stream = current_stream(self.device)

        def on_eval(user, i, loss, d_x_eval):
            if callback is not None:
                callback(i, loss)
            if point_callback is not None:
                host = torch.empty((self.n,), dtype=torch.float32)
                check(self.lib.uuo_copy_to_host(stream, d_x_eval, host.data_ptr(), self.n), "uuo_copy_to_host")
                point_callback(i, loss, host)

        cb = EVAL_CALLBACK(on_eval) if (callback is not None or point_callback is not None) else None
        with torch.cuda.device(self.device):
            check(self.lib.uuo_reprojection_solve(self.handle, stream, _ptr(x), byref(opt), byref(stats), _ptr(x_last),
                                                  _ptr(kp_last), ctypes.cast(cb, c_void_p) if cb else None, None),
                  "uuo_reprojection_solve")
        return {"n_iter": stats.n_iter, "n_eval": stats.n_eval, "first_loss": stats.first_loss,
                "final_loss": stats.final_loss, "stop_reason": STOP_REASONS[stats.stop_reason],
                "device_ms": stats.device_ms, "driver": "device-lbfgs(fused reprojection closure)",
                "x_last": x_last, "kp_last": kp_last}
