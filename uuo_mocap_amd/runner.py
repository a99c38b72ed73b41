"""Batch runner: the counterpart of the reference's `src/video_mocap/test/test.py`.

Same command line (argument names, `--print_options` default), same directory conventions
(`<input_dir>/<dataset>/mocap[_parts___P | _synthetic___S]/<subject>/<sequence>`, results under
`<input_dir>/<dataset>/results/<config name>/<subject>/<sequence>_stageii.npz`), the same skip-if-exists rule and the
same output files: `betas[10], trans[F,3], poses[F,72] (axis-angle, root first), mocap_frame_rate,
mocap_markers[F,M,3], gender="neutral"` plus one `_stageii.<stage>.npz` per saved stage (test.py:115-143).

Inputs: the reference's own triple -- `<sequence>.c3d` (uuo_mocap_amd.ingest.Markers: a self-contained C3D point reader
in place of ezc3d), the 4D-Humans `demo_<sequence>.pkl` (ingest.ImgSmpl: gap filling by lerp / slerp, img_mask) and the
`.avi`'s frame rate (RIFF headers in place of OpenCV) -- or a `<sequence>.npz` bundle holding
`markers[F,M,3], mocap_frame_rate, pose_body[F,23,3,3], root_orient[F,1,3,3], betas[F,10], img_mask[F], video_frame_rate`
(`write_sequence_npz` writes one).

One process per GPU: with `torchrun` (WORLD_SIZE > 1) the sequences are sharded round-robin over the ranks
(`parallel.shard_indices`), no collective on the data path; `--inflight N` overlaps N sequences per GPU; `--rank_mode
hypotheses | frames` puts ALL ranks on every sequence instead (its yaw hypotheses / the frame blocks of its solves)."""
from __future__ import annotations

import argparse
import os
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from .config import load_config
from .parallel import fit_many, shard_indices, world_info
from .synthetic import SyntheticImgSmpl, SyntheticMarkers
from .transforms import matrix_to_axis_angle

CAMERAS = {  # test.py:170-178
    "umpm": "l", "cmu_kitchen_pilot": "7151062", "cmu_kitchen_pilot_rb": "7151062", "moyo_train": None,
    "moyo_val": None, "bmlmovi_train": None, "bmlmovi_val": None,
}


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", type=str, help="configuration file", required=True)
    parser.add_argument("--cpu_only", action="store_true", help="only use the CPU")
    parser.add_argument("--dataset", type=str, help="dataset", required=True)
    parser.add_argument("--input_dir", type=str, help="input directory", required=True)
    parser.add_argument("--gpu", type=int, help="GPU ID")
    parser.add_argument("--num_files", type=int, help="number of files", default=None)
    parser.add_argument("--sequences", nargs="+", type=str, help="sequence names", default=None)
    parser.add_argument("--subjects", nargs="+", type=str, help="subject names", default=None)
    parser.add_argument("--synthetic", action="store_true", help="use synthetic mocap")
    parser.add_argument("--synthetic_list", nargs="+", default=[])
    parser.add_argument("--parts", action="store_true", help="use part mocap")
    parser.add_argument("--parts_list", nargs="+", default=[])
    parser.add_argument("--print_options", type=str, nargs="*", default=["loss", "progress"])
    parser.add_argument("--inflight", type=int, default=1, help="sequences fitted concurrently per GPU (not in the reference)")
    parser.add_argument("--rank_mode", choices=["sequences", "hypotheses", "frames"], default="sequences",
                        help="under torchrun (not in the reference): sequences = every rank fits its own share of the "
                             "sequences (default, no data-path collective); hypotheses / frames = all ranks work on every "
                             "sequence together, its yaw hypotheses or the frame blocks of its solves spread over them "
                             "(parallel.shard_hypotheses / parallel.shard_frames); rank 0 writes the outputs")
    parser.add_argument("--video_fps", type=float, default=None,
                        help="video frame rate when the .avi is absent or not an AVI container (not in the reference: it asks OpenCV)")
    return parser


def write_sequence_npz(path: str, markers: np.ndarray, mocap_frame_rate: float, pose_body, root_orient, betas,
                       img_mask=None, video_frame_rate: Optional[float] = None):
    """The bundle `load_sequence` reads (stand-in for the .c3d + 4D-Humans .pkl + .avi triple)."""
    F = markers.shape[0]
    np.savez(path, markers=np.asarray(markers, np.float32), mocap_frame_rate=float(mocap_frame_rate),
             pose_body=np.asarray(pose_body, np.float32), root_orient=np.asarray(root_orient, np.float32),
             betas=np.asarray(betas, np.float32),
             img_mask=np.ones(F, np.float32) if img_mask is None else np.asarray(img_mask, np.float32),
             video_frame_rate=float(mocap_frame_rate if video_frame_rate is None else video_frame_rate))


def load_dataset_sequence(filename_base: str, dataset: str, input_dir: str, video_fps: Optional[float] = None):
    """The reference's own inputs (test.py:76-102): `<sequence>.c3d` markers, the 4D-Humans result
    `<input_dir>/<dataset>/comparisons/4d_humans/<subject>/<sequence>[.<camera>]/results/demo_<sequence>.pkl` (joblib) and the
    frame rate of `<input_dir>/<dataset>/videos/<subject>/<sequence>[.<camera>].avi` (or `video_fps`).  Returns None when
    the HMR result is missing (the reference prints "Skipping" and moves on, test.py:91-93)."""
    from . import ingest

    subject = os.path.basename(os.path.dirname(filename_base))
    name = os.path.basename(filename_base)
    camera = CAMERAS.get(dataset)
    video_name = name if camera is None else name + "." + camera
    pkl = os.path.join(input_dir, dataset, "comparisons", "4d_humans", subject, video_name, "results", "demo_" + name + ".pkl")
    if not os.path.isfile(pkl):
        print("Skipping", pkl)
        return None
    if video_fps is None:
        video_fps = ingest.video_frame_rate(os.path.join(input_dir, dataset, "videos", subject, video_name + ".avi"))
    img_smpl = ingest.load_hmr_pkl(pkl, video_fps)
    markers = ingest.Markers(filename_base + ".c3d")
    points = ingest.cleanup_markers(np.nan_to_num(markers.get_points(), nan=0.0))  # test.py:98-101
    markers.set_points(points)
    return img_smpl, markers


def load_sequence(filename_base: str, dataset: Optional[str] = None, input_dir: Optional[str] = None,
                  video_fps: Optional[float] = None):
    """-> (img_smpl, markers) with the attributes multimodal_video_mocap reads (reference test.py:88-101): from the
    `<sequence>.npz` bundle if there is one, else from the reference's `.c3d` / 4D-Humans `.pkl` / `.avi` triple."""
    npz = filename_base + ".npz"
    if not os.path.isfile(npz):
        if os.path.isfile(filename_base + ".c3d") and dataset is not None and input_dir is not None:
            return load_dataset_sequence(filename_base, dataset, input_dir, video_fps)
        raise FileNotFoundError(npz)
    d = np.load(npz)
    points = np.nan_to_num(np.asarray(d["markers"], np.float32), nan=0.0)  # test.py:98-99
    markers = SyntheticMarkers(points, float(d["mocap_frame_rate"]))
    F = points.shape[0]
    root = torch.from_numpy(np.asarray(d["root_orient"], np.float32))
    img = SyntheticImgSmpl(
        trans=torch.from_numpy(np.asarray(d["trans"], np.float32)) if "trans" in d.files else torch.zeros(F, 3),
        root_orient=root, hmr_root_orient=root.clone(),
        pose_body=torch.from_numpy(np.asarray(d["pose_body"], np.float32)),
        betas=torch.from_numpy(np.asarray(d["betas"], np.float32)),
        foot_contacts=torch.zeros(F, 2), camera_bbox=torch.zeros(F, 3), center=torch.zeros(F, 2),
        scale=torch.zeros(F, 1), size=torch.zeros(F, 2),
        img_mask=torch.from_numpy(np.asarray(d["img_mask"], np.float32)) > 0.5,
        freq=float(d["video_frame_rate"]))
    return img, markers


def list_jobs(input_dir: str, output_dir: str, dataset: str, part: Optional[str], synthetic: Optional[str],
              sequences: Optional[List[str]], subjects: Optional[List[str]]):
    """(sequence file base, output file base) pairs in the reference's order, skip-if-exists applied (test.py:34-74)."""
    if part:
        mocap_dir = os.path.join(input_dir, dataset, "mocap_parts___" + part)
    elif synthetic:
        mocap_dir = os.path.join(input_dir, dataset, "mocap_synthetic___" + synthetic)
    else:
        mocap_dir = os.path.join(input_dir, dataset, "mocap")
    if subjects is None:
        subjects = sorted(os.listdir(mocap_dir))
    jobs = []
    for subject in subjects:
        if sequences is None:
            names = sorted({os.path.splitext(x)[0] for x in os.listdir(os.path.join(mocap_dir, subject))
                            if x.endswith(".c3d") or x.endswith(".npz")})
        else:
            names = list(sequences)
        for name in names:
            if synthetic:
                out = os.path.join(output_dir, subject, "synthetic_" + synthetic, name + "_stageii")
            else:
                out = os.path.join(output_dir, subject, name + "_stageii")
            os.makedirs(os.path.dirname(out), exist_ok=True)
            if os.path.exists(out + ".npz"):
                print("Skipping", out)
                continue
            jobs.append((os.path.join(mocap_dir, subject, name), out))
    return jobs


def save_outputs(output_filename: str, result: Dict):
    """test.py:115-143."""
    def poses72(root_orient, pose_body):
        rot = torch.cat((torch.as_tensor(root_orient), torch.as_tensor(pose_body)), dim=1)
        return torch.flatten(matrix_to_axis_angle(rot), start_dim=1, end_dim=-1).detach().cpu().numpy()

    def to_np(v):
        return v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v

    out = {
        "betas": to_np(result["betas"][0]), "trans": to_np(result["trans"]),
        "poses": poses72(result["root_orient"], result["pose_body"]),
        "mocap_frame_rate": result["mocap_frame_rate"], "mocap_markers": result["mocap_markers"].get_points(),
        "gender": "neutral",
    }
    np.savez(output_filename, **out)
    for stage, st in result.get("stages", {}).items():
        out["trans"] = to_np(st["trans"])
        out["betas"] = to_np(st["betas"])
        out["poses"] = poses72(st["root_orient"], st["pose_body"])
        np.savez(output_filename.replace("_stageii", "_stageii." + stage), **out)


def run(args, fit_fn: Optional[Callable] = None) -> int:
    """Fits every listed sequence; returns the number written by this rank."""
    config = load_config(args.config)
    output_dir = os.path.join(args.input_dir, args.dataset, "results", config["name"])
    rank, world, local_rank = world_info()
    if fit_fn is None:
        if args.cpu_only or not torch.cuda.is_available():
            raise RuntimeError("uuo_mocap_amd fits on the GPU only: --cpu_only / a machine without a HIP device cannot "
                               "run the accelerated path (use the reference for CPU runs)")
        n_dev = max(torch.cuda.device_count(), 1)
        device = torch.device("cuda:%d" % (args.gpu if args.gpu is not None else local_rank % n_dev))
        if world > 1 and getattr(args, "rank_mode", "sequences") != "sequences":
            # the collective modes need the process group BEFORE the first HIP call (RCCL binds to the device); without it every
            # rank would silently fit the whole sequence on its own
            from .parallel import ensure_process_group

            ensure_process_group(device)
        torch.cuda.set_device(device)
        from .multimodal import multimodal_video_mocap
        from .smpl import SmplInference

        smpl = SmplInference(device)

        def fit_fn(img_smpl, markers):
            return multimodal_video_mocap(img_smpl, markers, device, config=config, offset=0,
                                          print_options=args.print_options, save_stages=True, smpl_inference=smpl)
    else:
        device = None

    variants = [(None, None)]
    if args.parts:
        dirs = [x for x in os.listdir(os.path.join(args.input_dir, args.dataset)) if x.startswith("mocap_parts")]
        parts = [x.split("___")[-1] for x in dirs]
        if args.parts_list:
            parts = [x for x in parts if x in args.parts_list]
        variants = [(x, None) for x in parts]
    elif args.synthetic:
        dirs = [x for x in os.listdir(os.path.join(args.input_dir, args.dataset)) if x.startswith("mocap_synthetic")]
        syn = [x.split("___")[-1] for x in dirs]
        if args.synthetic_list:
            syn = [x for x in syn if x in args.synthetic_list]
        variants = [(None, x) for x in syn]
    written = 0
    for part, synthetic in variants:
        jobs = list_jobs(args.input_dir, output_dir, args.dataset, part, synthetic, args.sequences, args.subjects)
        if args.num_files is not None:
            jobs = jobs[:args.num_files + 1]  # the reference stops after file_count > num_files (test.py:145-147)
        rank_mode = getattr(args, "rank_mode", "sequences")
        together = rank_mode != "sequences" and world > 1
        if together:
            from .parallel import _dist

            d_ = _dist()
            if d_ is None or d_.get_world_size() != world:
                raise RuntimeError("--rank_mode %s with WORLD_SIZE=%d needs an initialised torch.distributed process group of "
                                   "that size (parallel.ensure_process_group)" % (rank_mode, world))
        mine = jobs if together else [jobs[i] for i in shard_indices(len(jobs), rank, world)]

        def one(job):
            base, out = job
            loaded = load_sequence(base, args.dataset, args.input_dir, getattr(args, "video_fps", None))
            if loaded is None:
                return None
            img_smpl, markers = loaded
            if together:  # every rank takes part in every fit and ends with the same result; one of them writes it
                from . import parallel

                ctx = parallel.shard_hypotheses() if rank_mode == "hypotheses" else \
                    parallel.shard_frames(device=device, lanes=int(config.get("num_root_orient_angles", 4)))
                with ctx:
                    result = fit_fn(img_smpl, markers)
                if rank != 0:
                    return None
            else:
                result = fit_fn(img_smpl, markers)
            save_outputs(out, result)
            return out

        written += sum(1 for r in fit_many(mine, one, inflight=1 if together else args.inflight, device=device)
                       if r is not None)
    return written


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.dataset not in CAMERAS:
        raise KeyError(args.dataset)
    from .parallel import limit_host_threads

    limit_host_threads()  # torch's CPU pool follows the process' CPU quota, not the host's core count
    n = run(args)
    print("wrote", n, "sequence(s)")


if __name__ == "__main__":
    main()
