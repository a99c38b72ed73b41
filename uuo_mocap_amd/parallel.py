"""One process per GPU: sequences are independent in the reference (one multimodal_video_mocap call per sequence
with its own betas: reference test/test.py:57-112), so a node shards *sequences* over ranks with no data-path
collective.  torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests) is used only for the
timing barrier, the max-over-ranks of the elapsed time and the gather of per-sequence results."""
from __future__ import annotations

import os
import threading
import time
from typing import Callable, Dict, List, Sequence

import torch


def world_info():
    """(rank, world size, local rank): from the initialised default process group when there is one, else from the
    launcher's environment (torchrun)."""
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size(), int(os.environ.get("LOCAL_RANK", str(dist.get_rank())))
    except Exception:
        pass
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def host_cpu_budget() -> int:
    """CPUs this process may actually use: the smaller of the scheduler affinity and the cgroup CPU quota (cpu.max).  A GPU
    box hands a process a QUOTA (16 CPUs per GPU here) while `nproc` still shows every core of the host (256)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                txt = fh.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) // int(txt[1]))))
            else:
                quota = int(txt[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh2:
                        n = min(n, max(1, quota // int(fh2.read().split()[0])))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def limit_host_threads(reserve: int = 4) -> int:
    """Caps torch's intra-op CPU thread pool at the process' CPU budget minus `reserve` (the fitting threads busy-poll the
    GPU's evaluation reports; the HIP runtime has threads of its own).  Found in round 3: torch sizes its OpenMP pool by
    the HOST's core count (256), any CPU tensor op of a fit (a dtype conversion of the marker array is enough) wakes all of
    them, they spin, the cgroup quota of 16 CPUs is burnt within milliseconds and the kernel throttles the whole process
    for the rest of the 100 ms period -- a 50 ms stall in every third `hmr_full` fit.  Call once from the application (the
    runner and bench.py do); a library does not change global thread settings on import.  Returns the thread count set."""
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))  # torchrun: the ranks of this host share the quota
    n = max(1, host_cpu_budget() // local_world - max(0, int(reserve)))
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return torch.get_num_threads()


_wait_lock = threading.Lock()
_wait_app_policy = (None, 20.0)  # what the APPLICATION asked for last (spin_us, sleep_us); fit_many("auto") restores it
_wait_auto_users = 0             # fit_many("auto") calls in progress (they share one temporary policy)


def set_wait_policy(spin_us: float = None, sleep_us: float = 20.0) -> None:
    """How the solver's host threads wait for the GPU's evaluation reports (uuo_set_wait_policy).  `spin_us=None`: spin
    (the default: lowest latency, one CPU per solve in flight -- twelve with three sequences in flight).  Otherwise a wait
    spins for about `spin_us` microseconds and then sleeps `sleep_us` at a time: for hosts whose CPU quota is smaller than
    the number of solves in flight.  Process-wide; call from the application.  While a `fit_many(wait_policy="auto")` is in
    progress the new policy takes effect when the last such call ends."""
    global _wait_app_policy
    with _wait_lock:
        _wait_app_policy = (spin_us, sleep_us)
        if _wait_auto_users == 0:
            _apply_wait_policy(spin_us, sleep_us)


def get_wait_policy():
    """(spin_us, sleep_us) the application set last (spin_us None = pure spinning)."""
    return _wait_app_policy


def _apply_wait_policy(spin_us, sleep_us) -> None:
    from . import _lib

    lib = _lib.load()
    if spin_us is None:
        _lib.check(lib.uuo_set_wait_policy(-1, 0), "uuo_set_wait_policy")
    else:
        # one poll = a load + `pause` ~ 40 ns on current x86 hosts
        _lib.check(lib.uuo_set_wait_policy(int(max(0.0, spin_us) * 25), int(max(1.0, sleep_us) * 1000)), "uuo_set_wait_policy")


def ensure_process_group(device=None):
    """The default torch.distributed process group of a multi-rank launch (torchrun's RANK / WORLD_SIZE / MASTER_*), created
    here when the application has not done it: "nccl" (= RCCL) bound to `device`, or "gloo" when the ranks of this host
    share a GPU (UUO_SHARE_GPU=1 or fewer devices than local ranks -- RCCL refuses two ranks on one device).  Must run
    before the first HIP call of a rank that is going to use RCCL.  Returns (rank, world).  A one-rank launch needs no group.
    Raises when a multi-rank environment cannot be brought up -- the collective modes never degrade to every rank fitting
    everything on its own."""
    import torch.distributed as dist

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env <= 1:
        return (0, 1)
    if not dist.is_available():
        raise RuntimeError("WORLD_SIZE=%d but torch.distributed is not available" % world_env)
    if not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world_env))))
        share = os.environ.get("UUO_SHARE_GPU", os.environ.get("UUO_BENCH_SHARE_GPU", "0")) == "1" or \
            (torch.cuda.is_available() and torch.cuda.device_count() < local_world)
        if share or device is None or torch.device(device).type != "cuda":
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device(device))
    if dist.get_world_size() != world_env:
        raise RuntimeError("process group has %d ranks, the launcher started %d" % (dist.get_world_size(), world_env))
    return (dist.get_rank(), dist.get_world_size())


def shard_indices(num_items: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment of sequence ids to ranks (balanced to within one item)."""
    return list(range(rank, num_items, world))


def _dist():
    import torch.distributed as dist

    return dist if dist.is_available() and dist.is_initialized() else None


def barrier(device=None):
    dist = _dist()
    if dist is not None:
        dist.barrier()
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(value: float, device=None) -> float:
    dist = _dist()
    if dist is None:
        return float(value)
    dev = device if (device is not None and dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def fit_sharded(sequence_ids: Sequence[int], fit_fn: Callable[[int], Dict], device=None):
    """Every rank fits its share of `sequence_ids` with `fit_fn(seq_id) -> result`; returns
    (results gathered on rank 0 as {seq_id: result} else None, elapsed seconds = max over ranks)."""
    rank, world, _ = world_info()
    mine = [sequence_ids[i] for i in shard_indices(len(sequence_ids), rank, world)]
    barrier(device)
    t0 = time.perf_counter()
    local = {sid: fit_fn(sid) for sid in mine}
    barrier(device)
    elapsed = max_over_ranks(time.perf_counter() - t0, device)
    dist = _dist()
    if dist is None:
        return local, elapsed
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(local, gathered, dst=0)
    if rank != 0:
        return None, elapsed
    merged: Dict = {}
    for part in gathered:
        merged.update(part)
    return merged, elapsed


def auto_wait_sleep_us() -> float:
    """How long the sleeping waits of `fit_many(wait_policy="auto")` sleep at a time: 20 us where the rank has four or more CPUs of
    its own (16 waiting threads then cost ~2 CPUs), 50 / 100 us on tighter budgets -- eight ranks inside ONE 16-CPU quota have two
    CPUs each, and a throttled cgroup stalls every thread of every rank for the rest of its 100-ms period."""
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    per_rank = host_cpu_budget() / local_world
    return 20.0 if per_rank >= 4 else (50.0 if per_rank >= 3 else 100.0)


def fit_many(items: Sequence, fit_fn: Callable, inflight: int = 1, device=None, wait_policy: str = "auto") -> List:
    """Fits independent sequences `fit_fn(item)` on ONE GPU with up to `inflight` of them in progress at a time
    (each on its own host thread, HIP stream and workspace group).  Sequences are independent in the reference
    (test/test.py:57-112 loops over them); overlapping them fills the GPU while another sequence is in a phase
    with fewer than four hypotheses alive (part stage, the hypotheses' ragged ends, the final marker stage).
    Results come back in the order of `items`.

    `wait_policy` "auto": while several sequences are in flight the solver's host threads (one per yaw hypothesis of every
    sequence: twelve with three in flight) sleep 20 us at a time instead of spinning while they wait for the GPU's reports
    (set_wait_policy): the GPU is the bottleneck then, throughput is the same (217.6-220.5 vs 217.8 ms per fit) and the
    process burns 1.7 CPUs instead of 9.8 -- inside a 16-CPU quota, or eight ranks on one host, that is the difference
    between running and being throttled.  Spinning is restored afterwards.  "keep": leave the process' policy alone."""
    from concurrent.futures import ThreadPoolExecutor

    from .engine import set_workspace_group, set_workspace_slot, worker_streams

    if inflight <= 1 or len(items) <= 1:
        return [fit_fn(it) for it in items]
    if wait_policy == "auto":
        # a temporary, process-wide policy shared by every "auto" call in progress; the last one to finish restores what the
        # application had set (it used to restore pure spinning unconditionally, and overlapping calls raced)
        global _wait_auto_users
        with _wait_lock:
            if _wait_auto_users == 0:
                _apply_wait_policy(10.0, auto_wait_sleep_us())
            _wait_auto_users += 1
        try:
            return fit_many(items, fit_fn, inflight, device, wait_policy="keep")
        finally:
            with _wait_lock:
                _wait_auto_users -= 1
                if _wait_auto_users == 0:
                    _apply_wait_policy(*_wait_app_policy)
    use_cuda = device is not None and torch.device(device).type == "cuda"
    main = torch.cuda.current_stream(device) if use_cuda else None
    free_groups = list(range(1, inflight + 1))  # group 0 stays with the calling thread
    import threading

    lock = threading.Lock()

    def run(it):
        with lock:
            g = free_groups.pop()
        try:
            set_workspace_group(g)
            set_workspace_slot(0)
            if use_cuda:
                st = worker_streams(device, 1, "sequence")[0]  # persistent, one per workspace group
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    out = fit_fn(it)
                st.synchronize()
                return out
            return fit_fn(it)
        finally:
            with lock:
                free_groups.append(g)

    with ThreadPoolExecutor(max_workers=inflight) as pool:
        return list(pool.map(run, items))


# ---------------------------------------------------------------------------------------------------------------------
# Beyond "one sequence per rank": the two other decompositions of SURVEY.md 8e.  Both are switched on for the calling thread
# with a context manager, so the operator surface (multimodal_video_mocap, optim_chamfer, optim_markers) stays the reference's.
# ---------------------------------------------------------------------------------------------------------------------
import contextlib

_ctx = threading.local()


def shared_betas_reducer():
    """The reducer of the active `shared_betas(...)` context of this thread, or None."""
    return getattr(_ctx, "shared", None)


_reducer_cache: Dict = {}


def _default_reducer(group, device, transport: str = "auto"):
    """One reducer per (group, device, transport) and process: its lanes are shared-memory tables / process groups, which are
    not to be created per fit.  COLLECTIVE on first use.  `transport`: "shm" = the node-local mailbox (dist_lbfgs.ShmReducer:
    what the ranks of one node use -- the exchanged blocks are produced in and consumed from HOST memory, so host shared
    memory is their shortest path); "rccl" / "gloo" = torch.distributed all_gather on the group itself (DistReducer: device
    buffers over RCCL when the group's backend is nccl; ranks on several nodes); "auto" = shm when every rank of the group
    runs on this host, else the group's own backend."""
    from .dist_lbfgs import DistReducer, LocalReducer, ShmReducer

    dist = _dist()
    if dist is None:
        return LocalReducer()
    if transport not in ("auto", "shm", "rccl", "gloo", "group"):
        raise ValueError("unknown collective transport %r" % transport)
    key = (id(group) if group is not None else 0, str(device), transport)
    if key not in _reducer_cache:
        use_shm = transport == "shm"
        if transport == "auto":
            import socket

            hosts = [None] * dist.get_world_size(group)
            dist.all_gather_object(hosts, socket.gethostname(), group=group)
            use_shm = len(set(hosts)) == 1
        if use_shm:
            import secrets

            token = [secrets.token_hex(6) if dist.get_rank(group) == 0 else None]
            dist.broadcast_object_list(token, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            _reducer_cache[key] = ShmReducer("/uuo_mb_%s" % token[0], dist.get_rank(group), dist.get_world_size(group))
        else:
            _reducer_cache[key] = DistReducer(group, device)
    return _reducer_cache[key]


def collective_lanes(count: int):
    """For multimodal_video_mocap: how the `count` yaw hypotheses of a fit may run CONCURRENTLY inside a `shared_betas` /
    `shard_frames` context -- a list of `count` context managers (one per hypothesis index, to be entered on the thread that
    fits it), or None when the context was opened without lanes (then the hypotheses run one after the other)."""
    lanes = getattr(_ctx, "lanes", None)
    if not lanes or len(lanes) < count:
        return None
    kind = getattr(_ctx, "lanes_kind", None)

    def enter(i):
        @contextlib.contextmanager
        def cm():
            prev_s, prev_f = getattr(_ctx, "shared", None), getattr(_ctx, "frames", None)
            if kind == "shared":
                _ctx.shared = lanes[i]
            else:
                _ctx.frames = FrameShard(lanes[i], getattr(prev_f, "active", False))
            try:
                yield
            finally:
                _ctx.shared, _ctx.frames = prev_s, prev_f
        return cm()

    return [lambda i=i: enter(i) for i in range(count)]


@contextlib.contextmanager
def shared_betas(group=None, device=None, reducer=None, lanes: int = 0, transport: str = "auto"):
    """EXTENSION (BASELINE configs[3]; not reference behaviour -- the reference fits every sequence with its own betas,
    SURVEY.md F12): inside this context every chamfer / marker stage solve is ONE joint L-BFGS problem over the ranks of
    `group`, the sequences of the ranks (same subject) sharing a single shape vector.  All ranks must run the same stages in
    the same order (they do: the iteration counts follow from all-reduced scalars only); the yaw hypotheses therefore run
    one after the other instead of on concurrent threads -- unless `lanes` >= their number: then every hypothesis index gets
    a process group of its own (DistReducer.fork; created once per process, COLLECTIVELY at the first entry) and the
    hypotheses run on their threads as usual, each ordering only its own lane.  Without an initialised process group the
    context is the one-rank case of the same driver."""
    if reducer is None:
        reducer = _default_reducer(group, device, transport)
    prev = getattr(_ctx, "shared", None), getattr(_ctx, "lanes", None), getattr(_ctx, "lanes_kind", None)
    _ctx.shared = reducer
    _ctx.lanes = reducer.fork(lanes) if lanes > 0 else None
    _ctx.lanes_kind = "shared"
    try:
        yield reducer
    finally:
        _ctx.shared, _ctx.lanes, _ctx.lanes_kind = prev


class HypothesisShard:
    """the yaw hypotheses of ONE sequence (reference multimodal.py:462-574: independent solves, best one
    picked by 4 scalars, :576-599) spread over the ranks of a process group -- rank r fits hypotheses r, r + world, ...;
    the per-hypothesis results (the parameters of each stage as host arrays, ~263 KB each at F = 300, and the solver
    statistics) are exchanged with one all_gather_object; every rank then holds all of them and continues identically."""

    def __init__(self, group=None, rank: int = None, world: int = None):
        dist = _dist()
        self.group = group
        self.rank = rank if rank is not None else (dist.get_rank(group) if dist is not None else 0)
        self.world = world if world is not None else (dist.get_world_size(group) if dist is not None else 1)

    def mine(self, count: int) -> List[int]:
        return list(range(self.rank, count, self.world))

    def exchange(self, local: Dict[int, Dict], count: int) -> List[Dict]:
        dist = _dist()
        if dist is None or self.world == 1:
            parts = [local]
        else:
            parts = [None] * self.world
            dist.all_gather_object(parts, local, group=self.group)
        merged: Dict[int, Dict] = {}
        for part in parts:
            merged.update(part)
        missing = [i for i in range(count) if i not in merged]
        if missing:
            raise RuntimeError("hypotheses %s were fitted by no rank" % missing)
        return [merged[i] for i in range(count)]


def hypothesis_shard():
    return getattr(_ctx, "hyp", None)


@contextlib.contextmanager
def shard_hypotheses(group=None, shard: HypothesisShard = None):
    """Inside this context multimodal_video_mocap fits only this rank's share of the yaw hypotheses and exchanges the
    results (HypothesisShard): one sequence uses up to `num_root_orient_angles` GPUs.  Every rank returns the full,
    identical output dictionary."""
    prev = getattr(_ctx, "hyp", None)
    _ctx.hyp = shard if shard is not None else HypothesisShard(group)
    try:
        yield _ctx.hyp
    finally:
        _ctx.hyp = prev


class FrameShard:
    """ONE solve spread over the ranks by contiguous blocks of frames.  Every stage problem is separable over
    frames except for the shape vector, so rank r solves the frames of its block with their own per-frame parameters, the 10
    betas are the shared tail of the joint problem (engine.solve_shared -> uuo_lbfgs_solve_shared: one all_gather of 16
    doubles per closure evaluation, one of 627 per iteration) and the per-rank loss weights carry the GLOBAL normalisers, so the
    joint objective is the one-GPU objective term by term (the fp32 summation order differs: converged quantities agree,
    trajectories need not).  After a solve the per-frame results of all blocks are exchanged and every rank holds the full
    tensors, so everything around the solves (segmentation, candidate search, placement, scoring) runs replicated and
    unchanged."""

    def __init__(self, reducer, joint_with_one_rank: bool = False):
        self.reducer = reducer
        self.rank, self.world = int(reducer.rank), int(reducer.world)
        # one rank: the plain solver IS the joint solve of one block; `joint_with_one_rank` sends it through the joint driver
        # and its exchanges all the same (tests: the collective code of a one-rank RCCL group on a one-GPU box)
        self.active = self.world > 1 or bool(joint_with_one_rank)

    def bounds(self, num_frames: int) -> List[int]:
        base, rem = divmod(int(num_frames), self.world)
        edges = [0]
        for r in range(self.world):
            edges.append(edges[-1] + base + (1 if r < rem else 0))
        return edges

    def block(self, num_frames: int):
        e = self.bounds(num_frames)
        if e[self.rank + 1] == e[self.rank]:
            raise ValueError("frame sharding: %d frames leave rank %d of %d without any" % (num_frames, self.rank, self.world))
        return e[self.rank], e[self.rank + 1]

    def gather_frames(self, local: torch.Tensor, num_frames: int) -> torch.Tensor:
        """[F_r, ...] blocks of all ranks -> [F, ...] on the caller's device, identical on every rank."""
        import numpy as np

        e = self.bounds(num_frames)
        per = int(np.prod(local.shape[1:])) if local.dim() > 1 else 1
        longest = max(e[r + 1] - e[r] for r in range(self.world))
        mine = np.zeros(longest * per, dtype=np.float64)
        mine[:local.numel()] = local.detach().reshape(-1).double().cpu().numpy()
        table = np.zeros((self.world, longest * per), dtype=np.float64)
        self.reducer.gather_array(mine, table)
        parts = [torch.from_numpy(table[r, :(e[r + 1] - e[r]) * per].copy()).to(device=local.device, dtype=local.dtype)
                 .reshape((e[r + 1] - e[r],) + tuple(local.shape[1:])) for r in range(self.world)]
        return torch.cat(parts, dim=0)


def frame_shard():
    """The FrameShard of the active `shard_frames(...)` context of this thread, or None."""
    return getattr(_ctx, "frames", None)


@contextlib.contextmanager
def shard_frames(group=None, device=None, reducer=None, lanes: int = 0, transport: str = "auto",
                 joint_with_one_rank: bool = False):
    """Inside this context the chamfer and marker stage solves of a fit (optim_chamfer / optim_markers on their fused
    closures) are spread over the ranks of `group` by frame blocks (FrameShard): one sequence uses all the
    GPUs of the group.  All ranks must call the fit with the same inputs; every rank returns the full, identical result.  The
    yaw hypotheses run one after the other (every solve is a collective) unless `lanes` >= their number (see
    shared_betas).  Without an initialised process group the context is the one-rank case."""
    if reducer is None:
        reducer = _default_reducer(group, device, transport)
    prev = getattr(_ctx, "frames", None), getattr(_ctx, "lanes", None), getattr(_ctx, "lanes_kind", None)
    _ctx.frames = FrameShard(reducer, joint_with_one_rank)
    _ctx.lanes = reducer.fork(lanes) if lanes > 0 else None
    _ctx.lanes_kind = "frames"
    try:
        yield _ctx.frames
    finally:
        _ctx.frames, _ctx.lanes, _ctx.lanes_kind = prev
