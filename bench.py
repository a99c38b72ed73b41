#!/usr/bin/env python3
"""Headline benchmark: mocap frames/sec fitted (300-frame sequence, 50 markers) on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one complete fit (multimodal_video_mocap) of one synthetic 300-frame x 50-marker sequence.
Sequences are independent in the reference (one call per sequence, own betas), so ranks shard sequences
with no data-path collective (weak scaling: every rank fits K sequences); the only collectives are the
timing barrier and a max-reduce of the elapsed time.  One JSON line is printed by rank 0.
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP16_TFLOPS = 2500.0  # same guide: dense F16/BF16 matrix peak
# k_skin3 at F=300, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB per launch; profiles/r4_pmc_summary.json)
SKIN16_FETCH_KB, SKIN16_WRITE_KB = 13514.6, 30858.6
PEAK_FP32_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: FP32 matrix (v_mfma_f32_32x32x2_f32) = vector peak
# algorithmic FLOPs of the skin kernel per frame: pose blend 2*207*20670 + skinning 2*6890*24*12 + apply 6890*24
SKIN_FLOPS_PER_FRAME = 2 * 207 * 20670 + 2 * 6890 * 24 * 12 + 6890 * 24
NN_FLOPS_PER_FRAME_MARKER = 6890 * 8
# what the kernel's matrix pipe usefully executes per frame: the [pose feature | betas] x [posedirs | shapedirs] blend
# (the <=4-weight skinning runs on the VALU, not as the dense 24-joint product SURVEY 8d's figure credits)
SKIN_MFMA_FLOPS_PER_FRAME = 2 * 217 * 20670
# what k_skin2 EXECUTES per frame: the 217-row blend on the matrix pipe + the <=4-weight skinning (4 weights x 12 matrix
# entries x 6890 vertices, multiply-add) + applying the blended 3x4 to the posed vertex: 9.80 MFLOP -- the roofline's `frac`
# (the SURVEY 8d figure above credits a dense 24-joint skinning product the kernel rightly does not do: `frac_survey_flops`)
SKIN_EXECUTED_FLOPS_PER_FRAME = 2 * 217 * 20670 + 2 * 4 * 12 * 6890 + 6890 * 24
# fused algorithmic bytes of one chamfer closure: tables once (18 688 848 B) + 2.75 KB per frame
FUSED_ALGORITHMIC_BYTES = lambda F: 18688848 + F * 2750  # noqa: E731
# algorithmic FLOPs of one frame through one closure (forward + backward)
CHAMFER_FLOPS_PER_FRAME_EVAL = 15.5e6
MARKER_FLOPS_PER_FRAME_EVAL = 0.28e6
CHAMFER_CEILING_FRAME_EVALS_PER_S = PEAK_FP32_TFLOPS * 1e12 / CHAMFER_FLOPS_PER_FRAME_EVAL  # 10.1 M/s


def part_flops_per_frame_eval(n_subset, n_markers):
    """part-stage closure with the pose-corrective blend cached per solve: shape blend + skinning of the candidate
    part's vertices + the K=1 search over them."""
    return n_subset * (2 * 10 * 3 + 2 * 24 * 12 + 24) + 8.0 * n_markers * n_subset


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)  # two full rounds of --inflight 4
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--markers", type=int, default=50)
    ap.add_argument("--config", default="video_mocap", help="video_mocap | hmr_full | hmr_part | mht_rotation | hmr_part_soft | video_mocap_soft")
    ap.add_argument("--inflight", type=int, default=4,
                    help="sequences fitted concurrently per GPU (parallel.fit_many): independent sequences overlap on one "
                         "device -- each on its own host thread, stream and workspaces -- which is how a dataset is run; "
                         "1 = one sequence at a time (latency of a single fit)")
    ap.add_argument("--wait-sleep-us", type=float, default=-1.0,
                    help="-1 (default): parallel.fit_many's own policy (sleeping waits while several sequences are in "
                         "flight); 0: the solver's host threads always spin while they wait for the GPU's reports; > 0: they "
                         "spin --wait-spin-us and then sleep this long at a time (parallel.set_wait_policy)")
    ap.add_argument("--wait-spin-us", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-only", action="store_true",
                    help="only the timing loops of the roofline section (one warm-up fit): the command profiled into "
                         "profiles/r1_roofline_kernel_stats.csv")
    ap.add_argument("--cpu-evals", type=int, default=20, help="closure evaluations per stage type timed on the CPU")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the hmr_full / hmr_part / mht_rotation legs")
    ap.add_argument("--soft-operator-fit", action="store_true",
                    help="also time ONE hmr_part_soft fit on the operator-composed closure (the fused soft closure's checker; "
                         "~20 s): other_configs.hmr_part_soft.operator_composed_ms_per_step")
    ap.add_argument("--mode", default="sequences", choices=["sequences", "hypotheses", "shared_betas", "frames"],
                    help="how N ranks share the work (SURVEY.md 8e): sequences = independent sequences per rank, no data-path "
                         "collective (default; weak scaling); hypotheses = every step is ONE sequence whose yaw hypotheses "
                         "are spread over the ranks (strong scaling, useful up to num_root_orient_angles ranks); "
                         "shared_betas = one sequence per rank, one shape vector for all of them (extension: joint L-BFGS, "
                         "one small all_gather per evaluation); frames = every step is ONE sequence whose chamfer / marker solves "
                         "are spread over the ranks by frame blocks (SURVEY 8e.3: the joint problem of shared_betas with global "
                         "normalisers; strong scaling)")
    ap.add_argument("--collective-lanes", type=int, default=4,
                    help="modes shared_betas / frames: process groups (gloo) for the yaw hypotheses, so that the hypotheses of a "
                         "fit run concurrently although every solve is a collective (one lane per hypothesis index); 0 = one "
                         "hypothesis after the other on the default group")
    ap.add_argument("--collective-transport", default="auto", choices=["auto", "shm", "gloo", "rccl"],
                    help="modes shared_betas / frames: what carries the 17 doubles a closure evaluation exchanges -- shm = the "
                         "node-local shared-memory mailbox (csrc/mailbox.hip; auto picks it when all ranks are on one host), "
                         "gloo / rccl = torch.distributed all_gather on process groups of that backend")
    ap.add_argument("--history-size", type=int, default=0,
                    help="EXPERIMENT ONLY (not the metric's workload): L-BFGS history of the chamfer / marker solves instead of "
                         "torch's default 100 -- halves the bytes of the two history passes; used to measure what those passes "
                         "cost a fit (compare frame_evals_per_s, the solves take other trajectories)")
    ap.add_argument("--hypothesis-lockstep", action="store_true",
                    help="step the yaw hypotheses as one lock-step batch instead of one host thread + stream each "
                         "(multimodal_video_mocap(execution={'hypothesis_lockstep': True}); same results)")
    return ap.parse_args()


MODE = "sequences"
LANES = 0
TRANSPORT = "auto"
REDUCERS = []


def fit_once(smpl, seq, cfg, dev, execution=None):
    import contextlib

    from uuo_mocap_amd import parallel
    from uuo_mocap_amd.multimodal import last_run_stats, multimodal_video_mocap

    ctx = contextlib.nullcontext()
    if MODE == "hypotheses":
        ctx = parallel.shard_hypotheses()
    elif MODE == "shared_betas":
        ctx = parallel.shared_betas(device=dev, lanes=LANES, transport=TRANSPORT)
    elif MODE == "frames":
        ctx = parallel.shard_frames(device=dev, lanes=LANES, transport=TRANSPORT)
    with ctx as handle:
        red = getattr(handle, "reducer", handle)
        if red is not None and hasattr(red, "stats") and red not in REDUCERS:
            REDUCERS.append(red)
        out = multimodal_video_mocap(seq.img_smpl, copy.deepcopy(seq.markers), dev, cfg, offset=0, print_options=[],
                                     save_stages=False, smpl_inference=smpl, execution=execution)
    return out, copy.deepcopy(dict(last_run_stats()))


def eval_counts(stats):
    n = {"part": 0, "chamfer": 0, "marker": 0}
    for s in stats.get("part", []):
        n["part"] += s["n_eval"]
    for s in stats.get("chamfer", []):
        n["chamfer"] += s["n_eval"]
    for key in ("marker", "marker_final"):
        for s in stats.get(key, []):
            n["marker"] += s["n_eval"]
    return n


def stage_summary(stats):
    """Final loss, stop reason and work of every solve of one fit, by stage (what the fit converged to, not only how
    fast it ran)."""
    out = {}
    for key in ("part", "chamfer", "marker", "marker_final"):
        solves = stats.get(key, [])
        if not solves:
            continue
        entry = {"solves": len(solves),
                 "n_iter": [int(s["n_iter"]) for s in solves][:8], "n_eval": [int(s["n_eval"]) for s in solves][:8],
                 "final_loss": [float(s["final_loss"]) for s in solves][:8],
                 "stop_reason": sorted(set(str(s["stop_reason"]) for s in solves))}
        if len(solves) > 8:
            entry["final_loss_min"] = float(min(s["final_loss"] for s in solves))
            entry["n_eval_total"] = int(sum(s["n_eval"] for s in solves))
        out[key] = entry
    if "yaw_scores" in stats:
        out["yaw_scores"] = [float(v) for v in stats["yaw_scores"]]
        out["best_angle"] = float(stats["best_angle"]) if stats.get("best_angle") is not None else None
    return out


def fit_quality(smpl, seq, out, dev):
    """Error of a fit against the synthetic ground truth (uuo_mocap_amd/synthetic.py: float64 SMPL equations): mean
    per-vertex and per-joint Euclidean error in millimetres (V2V / MPJPE over the 24 skeleton joints, reference
    evaluation/metrics.py definitions), the shape error and the masked marker-to-nearest-vertex chamfer score."""
    from uuo_mocap_amd.optimization import get_marker_mask, weighted_chamfer_distance

    with torch.no_grad():
        res = smpl(out["pose_body"].to(dev), out["betas"].to(dev), out["root_orient"].to(dev), out["trans"].to(dev))
        gt_v = torch.from_numpy(seq.gt["verts"]).to(dev)
        gt_j = torch.from_numpy(seq.gt["joints"]).to(dev)
        v2v = (res["vertices"] - gt_v).norm(dim=-1).mean().item()
        mpjpe = (res["joints"][:, :24] - gt_j).norm(dim=-1).mean().item()
        markers = torch.from_numpy(seq.markers.get_points()).float().to(dev)
        score = weighted_chamfer_distance(markers, res["vertices"], get_marker_mask(markers))[0].item()
    beta_err = float(np.abs(out["betas"][0].numpy() - seq.gt["betas"][0]).mean())
    return {"v2v_mm": 1e3 * v2v, "mpjpe_mm": 1e3 * mpjpe, "betas_mean_abs_err": beta_err,
            "marker_chamfer_m2": score}


def cpu_baseline(tables, seq, cfg, n_eval, n_cpu_evals, cpu_quota=None):
    """Reference-faithful CPU path (oracle 'port', same dense materialisations, torch autograd) timed per closure
    type on the host cores and scaled by the closure counts of the GPU fit of the same sequence."""
    from oracle import stages_ref
    from oracle.smpl_ref import SmplInferenceRef

    smpl = SmplInferenceRef(tables)
    markers = torch.from_numpy(seq.markers.get_points()).float()
    F, M = markers.shape[0], markers.shape[1]
    o_pose = seq.img_smpl.pose_body.clone()
    o_betas = (seq.img_smpl.betas.sum(0, keepdim=True) / seq.img_smpl.img_mask.sum()).clone()
    root = seq.img_smpl.root_orient.clone()
    trans0 = torch.median(markers, dim=1)[0].clone()
    full = packaged_cfg_full()
    per_eval = {}

    def timed(fn, leaves):
        fn()  # warm-up (allocator, threads)
        t0 = time.perf_counter()
        for _ in range(n_cpu_evals):
            for p in leaves:
                p.grad = None
            loss = fn()
            loss.backward()
        return (time.perf_counter() - t0) / n_cpu_evals

    if n_eval["chamfer"]:
        leaves = [trans0.clone().requires_grad_(True), torch.zeros(F, 1, 1, requires_grad=True),
                  o_betas.clone().requires_grad_(True), o_pose.clone().requires_grad_(True)]
        per_eval["chamfer"] = timed(lambda: stages_ref.chamfer_stage_loss(
            markers, leaves[3], o_pose, leaves[2], o_betas, root, leaves[0], leaves[1], smpl, full)[0], leaves)
    if n_eval["marker"]:
        one_hot = torch.zeros(M, 6890)
        one_hot[torch.arange(M), torch.from_numpy(seq.gt["marker_vids"]).long()] = 1.0
        leaves = [o_pose.clone().requires_grad_(True), o_betas.clone().requires_grad_(True),
                  root.clone().requires_grad_(True), trans0.clone().requires_grad_(True)]
        per_eval["marker"] = timed(lambda: stages_ref.marker_stage_loss(
            markers, leaves[0], o_pose, leaves[1], o_betas, leaves[2], leaves[3], one_hot, smpl, full)[0], leaves)
    if n_eval["part"]:
        vlabels = torch.argmax(smpl.get_lbs_weights(), dim=-1)
        vidx = torch.cat([(vlabels == j).nonzero(as_tuple=True)[0] for j in range(24)], dim=0)
        leaves = [torch.zeros(1, 1, 1, requires_grad=True), trans0.clone().requires_grad_(True),
                  o_betas.clone().requires_grad_(True)]
        per_eval["part"] = timed(lambda: stages_ref.part_stage_loss(
            markers, o_pose, leaves[2], o_betas, root, leaves[1], leaves[0], vidx, smpl, full)[0], leaves)
    total = sum(per_eval[k] * n_eval[k] for k in per_eval)
    return {
        "value": F / total if total > 0 else None,
        "unit": "frames/s",
        "cores": torch.get_num_threads(),
        "cpu_quota": cpu_quota,
        "kind": "port",
        "sample": "oracle (reference-faithful torch-CPU port with the reference's dense materialisations) timed for %d "
                  "closure evaluations (forward+backward) per stage type at F=%d, M=%d: %s s/eval; extrapolated by the "
                  "GPU fit's closure counts %s (L-BFGS vector work and marker placement excluded -> optimistic for "
                  "the CPU); %d torch threads = the process' CPU quota (the host shows %d cores)"
                  % (n_cpu_evals, F, M, {k: round(v, 3) for k, v in per_eval.items()}, n_eval, torch.get_num_threads(),
                     os.cpu_count()),
        "seconds_per_eval": per_eval,
    }


def cgroup_cpu_stat():
    """cpu.stat of this process' cgroup (v2) as a dict of integers, {} where there is none."""
    try:
        with open("/sys/fs/cgroup/cpu.stat") as fh:
            return {k: int(v) for k, v in (line.split() for line in fh if len(line.split()) == 2)}
    except (OSError, ValueError):
        return {}


def measure_roofline(smpl, seq, dev, F, iters=200):
    from uuo_mocap_amd.config import packaged_config
    from uuo_mocap_amd.engine import ChamferProblem

    # ---- roofline of the dominant kernel (k_skin: MFMA blend + skinning), HIP events on the launch stream
    markers = torch.from_numpy(seq.markers.get_points()).float().to(dev)
    o_betas = (seq.img_smpl.betas.sum(0, keepdim=True) / seq.img_smpl.img_mask.sum()).to(dev)
    full_cfg = packaged_config("video_mocap")
    prob = ChamferProblem(smpl, markers, seq.img_smpl.pose_body.to(dev), o_betas, seq.img_smpl.root_orient.to(dev),
                          full_cfg)
    x = prob.pack(torch.median(markers, dim=1)[0], torch.zeros(F, 1, 1, device=dev), o_betas,
                  seq.img_smpl.pose_body.to(dev))
    skin32_ms = prob.time_closure(x, iters=iters, dominant_only=1)   # k_skin2: the fp32 matrix pipe (operators, other closures)
    skin_ms = prob.time_closure(x, iters=iters, dominant_only=2)     # k_skin3: the chamfer closure's own skinning kernel
    closure_ms = prob.time_closure(x, iters=iters, dominant_only=False)
    skin_flops = SKIN_EXECUTED_FLOPS_PER_FRAME * F
    achieved = skin_flops / (skin_ms * 1e-3) / 1e12                       # executed useful FLOPs: what `frac` is made of
    survey = SKIN_FLOPS_PER_FRAME * F / (skin_ms * 1e-3) / 1e12           # SURVEY 8d's count (dense skinning credited)
    # HBM traffic of one k_skin3 launch: separate rocprofv3 --pmc passes (profiles/r4_pmc_summary.json):
    # FETCH_SIZE 13 515 KB x2 (gfx950 correction for 16-B/lane coalesced reads) + WRITE_SIZE 30 859 KB, at F=300
    # (k_skin2: 15 188 x2 + 30 396).
    # An OFFLINE figure (a PMC pass cannot run inside this process): null at any other size.
    traffic = int((SKIN16_FETCH_KB * 2 + SKIN16_WRITE_KB) * 1024) if (F == 300) else None
    mfma_useful = SKIN_MFMA_FLOPS_PER_FRAME * F / (skin_ms * 1e-3) / 1e12
    closure_rate = F / (closure_ms * 1e-3)
    roofline = {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_TFLOPS, "traffic": traffic,
                "frac_basis": "useful FLOPs of k_skin3 at their fp32 count -- 9.80 MFLOP per frame (217 x 20670 blend, <=4-weight "
                              "skinning and the 3x4 apply on the VALU) -- against the FP32 matrix peak, the precision class of its "
                              "result; the blend itself runs on the fp16 pipe: both operands split into hi/lo fp16 planes, three "
                              "products of 224 x 20670 (v_mfma_f32_16x16x32_f16, fp32 accumulation): frac_of_fp16_peak",
                "frac_of_fp16_peak": (3 * 2 * 224 * 20670 * F / (skin_ms * 1e-3) / 1e12) / PEAK_FP16_TFLOPS,
                "fp32_kernel": "k_skin2<true,0>", "fp32_kernel_ms": skin32_ms,
                "fp32_kernel_frac": (skin_flops / (skin32_ms * 1e-3) / 1e12) / PEAK_FP32_TFLOPS,
                "frac_survey_flops": survey / PEAK_FP32_TFLOPS, "achieved_survey_flops": survey,
                "flops_per_launch_survey": SKIN_FLOPS_PER_FRAME * F,
                # PMC bytes of a launch against what a fully fused closure would have to move (tables once +
                # 2.75 KB per frame = 19.5 MB at F = 300) and against this design's own budget (vertices and unit boxes are
                # materialised for the pruned search: 46.6 MB)
                "traffic_vs_algorithmic": (traffic / FUSED_ALGORITHMIC_BYTES(F)) if traffic else None,
                "traffic_vs_unfused_budget": (traffic / (18688848 + F * (6890 * 12 + 431 * 24))) if traffic else None,
                "traffic_source": "OFFLINE: separate rocprofv3 --pmc passes of this kernel (k_skin3) at F=300 "
                                  "(profiles/r4_pmc_summary.json), not measured by this run",
                "kernel": "k_skin3<true>",
                # frac credits SURVEY 8d's dense 24-joint skinning product; the kernel does that part as <=4-weight VALU
                # work, so the matrix pipe's own useful rate is the blend contraction alone:
                "frac_mfma_useful": mfma_useful / PEAK_FP32_TFLOPS, "achieved_mfma_useful": mfma_useful,
                # one whole chamfer closure (forward + backward, five kernels) alone on the GPU against SURVEY 8d's
                # FP32 ceiling of 157.3 TF / 15.5 MFLOP = 10.1 M frame-evals/s
                "closure_frac": closure_rate / CHAMFER_CEILING_FRAME_EVALS_PER_S,
                "kernel_ms": skin_ms, "flops_per_launch": skin_flops,
                # SURVEY 8d asks for the HBM fraction as well: algorithmic bytes of the launch (blend basis once +
                # vertices + unit boxes written) over its duration against the 8 TB/s spec; small by construction
                # (the kernel is FP32-bound at ~80 FLOP/B)
                "algorithmic_bytes": 18688848 + F * (6890 * 12 + 431 * 24),
                "achieved_hbm_GBps": (18688848 + F * (6890 * 12 + 431 * 24)) / (skin_ms * 1e-3) / 1e9,
                "achieved_hbm_frac": (18688848 + F * (6890 * 12 + 431 * 24)) / (skin_ms * 1e-3) / 8.0e12,
                "chamfer_closure_ms": closure_ms, "closure_frame_evals_per_s": F / (closure_ms * 1e-3)}
    return roofline


def packaged_cfg_full():
    from uuo_mocap_amd.config import packaged_config

    return packaged_config("video_mocap")


def main():
    global MODE, LANES, TRANSPORT
    args = parse()
    MODE = args.mode
    LANES = args.collective_lanes
    TRANSPORT = args.collective_transport
    if TRANSPORT in ("gloo", "rccl"):
        TRANSPORT = "group"   # torch.distributed all_gather on the group's own backend (nccl = RCCL unless the ranks share a GPU)
    if MODE != "sequences":
        args.inflight = 1  # both modes issue collectives from the fitting thread: one sequence at a time per rank
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rehearsal on a box with fewer GPUs than ranks (UUO_BENCH_SHARE_GPU=1): ranks share the devices and rendezvous
        # over gloo -- RCCL refuses two ranks on one device; the timed path is the same
        share = os.environ.get("UUO_BENCH_SHARE_GPU") == "1"
        if share:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the fitted path has no CPU fallback)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from uuo_mocap_amd.body_model import synthetic_smpl
    from uuo_mocap_amd.config import packaged_config
    from uuo_mocap_amd.engine import ChamferProblem
    from uuo_mocap_amd.parallel import auto_wait_sleep_us, host_cpu_budget, limit_host_threads
    from uuo_mocap_amd.smpl import SmplInference
    from uuo_mocap_amd.synthetic import make_sequence

    # torch's CPU pool follows the process' CPU quota, not the host's core count (parallel.limit_host_threads explains)
    host_threads = limit_host_threads()
    from uuo_mocap_amd.parallel import set_wait_policy
    fit_wait = "auto"  # parallel.fit_many: sleeping waits while several sequences are in flight, spinning otherwise
    if args.wait_sleep_us >= 0:
        fit_wait = "keep"
        if args.wait_sleep_us > 0:
            set_wait_policy(spin_us=args.wait_spin_us, sleep_us=args.wait_sleep_us)
    tables = synthetic_smpl(0)
    cfg = packaged_config(args.config)
    if args.hypothesis_lockstep:
        cfg["execution"] = {"hypothesis_lockstep": True}
    if args.history_size > 0:
        cfg["optimizer"]["history_size"] = args.history_size
    smpl = SmplInference(dev, tables=tables)
    F, M = args.frames, args.markers
    n_seq = args.warmup + args.steps
    limb = args.config in ("hmr_part", "hmr_part_soft")
    # distinct seeds for every warm-up and timed sequence of every rank (the solves stop on tolerances, so time depends on
    # the data: a timed step must not repeat a warm-up step)
    seed_base = 0 if MODE in ("hypotheses", "frames") else rank * n_seq   # these modes: all ranks work on the SAME sequences
    # shared_betas mode: step i of every rank is a different sequence of ONE subject (same ground-truth shape)
    seqs = [make_sequence(tables, seed=seed_base + i, num_frames=F, num_markers=10 if limb else M, limb_only=limb,
                          subject_seed=(5000 + i) if MODE == "shared_betas" else None)
            for i in range(n_seq)]

    if args.roofline_only:
        # no fit: the dominant kernel and the chamfer closure alone on an idle GPU (the command behind
        # profiles/r1_roofline_kernel_stats.csv, whose k_skin2 average must agree with roofline.kernel_ms)
        if rank == 0:
            print(json.dumps({"roofline": measure_roofline(smpl, seqs[-1], dev, F, iters=500)}), flush=True)
        return

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    from uuo_mocap_amd.parallel import fit_many

    import contextlib
    import io

    # the reference prints stage banners unconditionally; sys.stdout is process-wide, so it is redirected once around
    # the fits (not per fit: with sequences in flight the per-fit redirections would restore each other's streams)
    with contextlib.redirect_stdout(io.StringIO()):
        # initialisation, not a step: a fit of the workload's size with a handful of iterations loads every code object,
        # allocates the per-hypothesis workspaces and pays torch's lazy initialisations (2 s in a fresh process) even when
        # --warmup is 0
        cfg_init = copy.deepcopy(cfg)
        for k in ("part", "chamfer", "marker"):
            if cfg_init["stages"][k]["num_iters"] > 0:
                cfg_init["stages"][k]["num_iters"] = 3
        fit_once(smpl, seqs[0], cfg_init, dev)
        fit_many(seqs[:args.warmup], lambda sq: fit_once(smpl, sq, cfg, dev), inflight=args.inflight, device=dev,
                 wait_policy=fit_wait)
        barrier()
        coll0 = [r_.stats() for r_ in REDUCERS]
        cg0 = cgroup_cpu_stat()
        t0 = time.perf_counter()
        fits = fit_many(seqs[args.warmup:n_seq], lambda sq: fit_once(smpl, sq, cfg, dev), inflight=args.inflight,
                        device=dev, wait_policy=fit_wait)
        barrier()
        elapsed = time.perf_counter() - t0
        cg1 = cgroup_cpu_stat()
        coll1 = [r_.stats() for r_ in REDUCERS]
    all_stats = [st for _, st in fits]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # fit quality of every timed step of this rank against the synthetic ground truth (outside the timed region)
    quality = [fit_quality(smpl, sq, out, dev) for sq, (out, _) in zip(seqs[args.warmup:n_seq], fits)]
    q_mean = {k: float(np.mean([q[k] for q in quality])) for k in quality[0]}
    q_worst = {k: float(np.max([q[k] for q in quality])) for k in quality[0]}
    if world > 1:
        # every rank fitted different sequences: report the mean / worst over the whole job
        gathered = [None] * world
        dist.all_gather_object(gathered, (q_mean, q_worst))
        q_mean = {k: float(np.mean([g[0][k] for g in gathered])) for k in q_mean}
        q_worst = {k: float(np.max([g[1][k] for g in gathered])) for k in q_worst}

    # host side of every rank's timed region (its own cgroup view; ranks of one box share one quota)
    host_mine = {"rank": rank, "torch_threads": host_threads,
                 "cpu_seconds_timed": (cg1.get("usage_usec", 0) - cg0.get("usage_usec", 0)) / 1e6 if cg0 else None,
                 "nr_throttled_timed": cg1.get("nr_throttled", 0) - cg0.get("nr_throttled", 0) if cg0 else None}
    host_all = [host_mine]
    if world > 1:
        host_all = [None] * world
        dist.all_gather_object(host_all, host_mine)
    if rank == 0:
        n_eval = eval_counts(all_stats[-1])
        total_evals = sum(sum(eval_counts(s).values()) for s in all_stats)
        frames = (1 if MODE in ("hypotheses", "frames") else world) * args.steps * F
        value = frames / elapsed
        roofline = measure_roofline(smpl, seqs[-1], dev, F)
        # whole-fit arithmetic rate: SURVEY 8d's algorithmic FLOPs of every closure evaluation the timed fits executed
        # (rank 0's counts; all ranks run the same workload) over the wall time, against the FP32 peak
        n_sub = int(tables.v_template.shape[0])
        fit_flops = 0.0
        for st in all_stats:
            c = eval_counts(st)
            fit_flops += F * (c["chamfer"] * CHAMFER_FLOPS_PER_FRAME_EVAL + c["marker"] * MARKER_FLOPS_PER_FRAME_EVAL)
            for ps in st.get("part", []):
                fit_flops += F * ps["n_eval"] * part_flops_per_frame_eval(ps.get("n_subset", n_sub),
                                                                           ps.get("n_markers", M))
        roofline["fit_frac"] = fit_flops / elapsed / (PEAK_FP32_TFLOPS * 1e12)
        roofline["fit_achieved"] = fit_flops / elapsed / 1e12
        result = {
            "metric": "mocap frames/sec fitted (300-frame seq, 50 markers)",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if MODE in ("hypotheses", "frames") else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s.yaml full fit, F=%d frames x M=%d markers, synthetic SMPL-shaped model, one "
                                   "sequence per step per GPU" % (args.config, F, M if not limb else 10),
                       "frames": F, "markers": M, "sequences_per_gpu": args.steps,
                       "sequences_in_flight": args.inflight, "rank_mode": MODE,
                       **({"EXPERIMENT_history_size": args.history_size} if args.history_size > 0 else {})},
            "closure_evals_per_step": total_evals / max(args.steps, 1), "closure_evals_last_step": n_eval,
            "frame_evals_per_s": world * total_evals * F / elapsed if world == 1 else None,
            "fit_quality": {"mean": q_mean, "worst": q_worst, "steps": len(quality) * world,
                            "against": "synthetic ground truth (float64 SMPL equations, uuo_mocap_amd/synthetic.py); "
                                       "HMR stand-in starts 100 deg off in yaw with 0.1 rad pose / 0.5 shape noise"},
            "stages_last_step": stage_summary(all_stats[-1]),
            "stage_ms_last": {l: round(1e3 * (t - p_), 2) for (l, t), p_ in
                              zip(all_stats[-1]["timeline"], [0.0] + [t for _, t in all_stats[-1]["timeline"][:-1]])},
            "roofline": roofline,
            # host side of the timed region (this process' cgroup): CPU seconds burnt by the polling / orchestrating threads
            # and whether the CPU quota throttled them (parallel.limit_host_threads explains why that matters)
            "host": {"cpu_quota": host_cpu_budget(), "torch_threads": host_threads,
                     "wait_policy": ("sleep %g us after 10 us of spinning while sequences are in flight (parallel.fit_many)"
                                     % auto_wait_sleep_us() if fit_wait == "auto" and args.inflight > 1 else
                                     "spin" if args.wait_sleep_us <= 0 else "sleep %g us" % args.wait_sleep_us),
                     "cpu_seconds_timed": (cg1.get("usage_usec", 0) - cg0.get("usage_usec", 0)) / 1e6 if cg0 else None,
                     "nr_throttled_timed": cg1.get("nr_throttled", 0) - cg0.get("nr_throttled", 0) if cg0 else None,
                     "throttled_seconds_timed": (cg1.get("throttled_usec", 0) - cg0.get("throttled_usec", 0)) / 1e6
                     if cg0 else None,
                     "per_rank": host_all if world > 1 else None},
        }
        if MODE in ("shared_betas", "frames"):
            n_g = sum(b_["gathers"] - a_["gathers"] for a_, b_ in zip(coll0, coll1))
            t_g = sum(b_["seconds"] - a_["seconds"] for a_, b_ in zip(coll0, coll1))
            result["collective"] = {
                "transport": type(REDUCERS[0]).__name__ if REDUCERS else None, "lanes": LANES,
                "gathers_timed": n_g, "mean_gather_us": 1e6 * t_g / n_g if n_g else None,
                "note": "rank 0's time inside gathers (message copy + waiting for the slowest rank) over the timed fits"}
        if world == 1 and MODE == "sequences":
            # latency of ONE sequence alone on the GPU (--inflight 1) beside the throughput headline (three in flight)
            n_lat = min(3, args.steps)
            seqs_l = [make_sequence(tables, seed=3000 + i, num_frames=F, num_markers=10 if limb else M, limb_only=limb)
                      for i in range(n_lat)]
            with contextlib.redirect_stdout(io.StringIO()):
                torch.cuda.synchronize(dev)
                t3 = time.perf_counter()
                fits_l = [fit_once(smpl, sq, cfg, dev) for sq in seqs_l]
                torch.cuda.synchronize(dev)
                dt3 = time.perf_counter() - t3
            result["latency_one_sequence"] = {
                "sequences_in_flight": 1, "steps": n_lat, "ms_per_fit": 1e3 * dt3 / n_lat, "value": n_lat * F / dt3,
                "unit": "frames/s",
                "closure_evals_per_step": sum(sum(eval_counts(st_).values()) for _, st_ in fits_l) / n_lat}
        if world == 1 and args.config == "video_mocap" and not args.no_other_configs:
            # the other shipped configurations on sequences of the same size, beside the headline (not the metric's
            # workload): hmr_full.yaml = BASELINE configs[1] as written (part stage on the full skeleton only: SURVEY F9),
            # hmr_part.yaml = configs[2] (10 markers on one limb, sub-tree search), mht_rotation.yaml = configs[4]'s
            # config (one yaw hypothesis)
            result["other_configs"] = {}
            n_other = min(args.steps, 4)
            for name in ("hmr_full", "hmr_part", "mht_rotation", "hmr_part_soft", "video_mocap_soft"):
                cfg_o = packaged_config(name)
                limb_o = name in ("hmr_part", "hmr_part_soft")
                seqs_o = [make_sequence(tables, seed=1000 + i, num_frames=F, num_markers=10 if limb_o else M,
                                        limb_only=limb_o) for i in range(n_other + 1)]
                with contextlib.redirect_stdout(io.StringIO()):
                    fit_once(smpl, seqs_o[0], cfg_o, dev)
                    torch.cuda.synchronize(dev)
                    t1 = time.perf_counter()
                    fits_o = [fit_once(smpl, sq, cfg_o, dev) for sq in seqs_o[1:]]
                    torch.cuda.synchronize(dev)
                    dt = time.perf_counter() - t1
                q_o = [fit_quality(smpl, sq, out, dev) for sq, (out, _) in zip(seqs_o[1:], fits_o)]
                st_o = [st for _, st in fits_o]
                # ... and as a dataset is run (parallel.fit_many, the headline's `--inflight`): throughput, not latency
                flight = None
                if args.inflight > 1:
                    n_fl = 2 * args.inflight
                    seqs_f = [make_sequence(tables, seed=2000 + i, num_frames=F, num_markers=10 if limb_o else M,
                                            limb_only=limb_o) for i in range(args.inflight + n_fl)]
                    with contextlib.redirect_stdout(io.StringIO()):
                        fit_many(seqs_f[:args.inflight], lambda sq: fit_once(smpl, sq, cfg_o, dev), inflight=args.inflight,
                                 device=dev, wait_policy=fit_wait)  # (the workspaces of the other workspace groups)
                        torch.cuda.synchronize(dev)
                        t2 = time.perf_counter()
                        fit_many(seqs_f[args.inflight:], lambda sq: fit_once(smpl, sq, cfg_o, dev), inflight=args.inflight,
                                 device=dev, wait_policy=fit_wait)
                        torch.cuda.synchronize(dev)
                        dt2 = time.perf_counter() - t2
                    flight = {"sequences_in_flight": args.inflight, "steps": n_fl, "value": n_fl * F / dt2,
                              "unit": "frames/s", "ms_per_step": 1e3 * dt2 / n_fl}
                result["other_configs"][name] = {
                    "value": n_other * F / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt / n_other, "steps": n_other,
                    "sequences_in_flight": 1, "in_flight": flight,
                    "markers": 10 if limb_o else M,
                    "closure_evals_per_step": sum(sum(eval_counts(s_).values()) for s_ in st_o) / max(n_other, 1),
                    "stage_ms_last": {l: round(1e3 * (t - p_), 2) for (l, t), p_ in
                                      zip(st_o[-1]["timeline"], [0.0] + [t for _, t in st_o[-1]["timeline"][:-1]])},
                    "fit_quality_mean": {k: float(np.mean([q[k] for q in q_o])) for k in q_o[0]},
                    "stages_last_step": stage_summary(st_o[-1]),
                }
            result["other_configs"]["hmr_full"]["note"] = \
                "hmr_full.yaml: part stage only (stages.chamfer / stages.marker num_iters 0)"
            result["other_configs"]["hmr_part_soft"]["note"] = \
                "EXTENSION, not a reference configuration (BASELINE configs[2] names a soft-assignment path): hmr_part.yaml " \
                "with the soft-min data term (stages.part.losses.soft_chamfer 10, soft_tau 2.5e-4 m^2) on the FUSED closure " \
                "(k_part_soft + k_bwd_part in pre mode), all candidates in one lock-step batch like hmr_part"
            result["other_configs"]["video_mocap_soft"]["note"] = \
                "EXTENSION, not a reference configuration (the north star names a soft-assignment Chamfer distance): " \
                "video_mocap.yaml with the chamfer stage's data term soft over all 6 890 vertices (soft_chamfer 10, soft_tau 1e-3 " \
                "m^2) on the fused closure: box-pruned soft-min kernels + the dense backward on the matrix pipe (csrc/dense_bwd.hip)"
            if args.soft_operator_fit:
                # the fused closure's checker as a timing: the same fit with closures composed from the differentiable HIP
                # operators, one candidate after the other (round 4's first route: ~23 s per fit)
                seq_s = make_sequence(tables, seed=1001, num_frames=F, num_markers=10, limb_only=True)
                with contextlib.redirect_stdout(io.StringIO()):
                    torch.cuda.synchronize(dev)
                    t1 = time.perf_counter()
                    fit_once(smpl, seq_s, packaged_config("hmr_part_soft"), dev, execution={"part_soft_fused": False})
                    torch.cuda.synchronize(dev)
                    dt = time.perf_counter() - t1
                result["other_configs"]["hmr_part_soft"]["operator_composed_ms_per_step"] = 1e3 * dt
        if world == 1 and not args.no_cpu_baseline:
            torch.set_num_threads(host_cpu_budget())  # the CPU leg gets every CPU of the quota (nothing else runs now)
            result["cpu_baseline"] = cpu_baseline(tables, seqs[-1], cfg, n_eval, args.cpu_evals, host_cpu_budget())
            torch.set_num_threads(host_threads)
            if result["cpu_baseline"]["value"]:
                result["gpu_over_cpu"] = value / result["cpu_baseline"]["value"]
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
