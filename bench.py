#!/usr/bin/env python3
"""bench.py -- Gauss-Newton tracking throughput on MI355X (BASELINE.json metric: GN iterations/s and aligned
Mpix/s on 1920x1080 3-level pyramids).

Workload "R1080p-dense-se3" (default): the reference-faithful tracker (OdometerClass::TrackPose semantics,
6-DoF SE(3), odometer.cpp:257-426) on 1920x1080 frames with a 3-level pyramid (lv_f=2, lv_l=0), 8x8 patches on
an 8-px grid that tiles the frame exactly (240 x 135 = 32 400 points => 2 073 600 aligned pixels per GN
iteration, the pixel count of a full 1080p frame), maxiter=10 with normdp_ratio=0 (fixed iteration count, so
every launch does full work), donorm=0, dopatchnorm=0. A "step" is one coarse-to-fine tracking (SetPose
projection + 3 levels x [setup + 10 GN iterations]) of a batch of B independent frame pairs (default 32: each
pair has its own pyramids and patch buffers, so one GN iteration streams B x 33 MB = 1.06 GB, far beyond the 256 MB
Infinity Cache; measured: 16 pairs 238, 32 pairs 266, 64 pairs 270 Gpix/s -- fixed launch/tail costs amortise).
Inputs (pyramids, 3-D points) are resident in HBM before the timed region; every step ends with its poses on the
host. Single-GPU mode (since r03): the B pairs of a step are ONE engine on one stream in the resident-iteration form --
per level one setup launch (k_ref8) and ONE launch for all iterations (k_level_resident: the templates of four frame
pairs stay in registers / LDS, B pairs in ceil(B / 4) rounds) -- and two step holders alternate, so that the host
prepares step i+1 while the GPU runs step i (--no-pipeline: one holder). --variant 2097152 selects the streaming
kernels (a launch pair per iteration), which run as S = 2 engines of B/2 pairs on two HIP streams (--streams).

  value        = aligned pixels / s  (pixels entering the residual, all levels, all problems, all ranks) in Mpix/s
  roofline     = the dominant kernel, k_level_resident: algorithmic bytes (16 B per patch pixel and GN
                 iteration -- T, Gx, Gy, one current-frame texel -- x maxiter x the launch's pixels) per launch / mean
                 launch duration (HIP events on the kernel's stream around every launch; one engine, one stream: nothing
                 overlaps), vs 8 TB/s HBM3E. The kernel does not MOVE those bytes (templates cross HBM once per level),
                 so frac exceeds 1: the distance past the streaming formulation's roofline. roofline.resident_bounds
                 holds what bounds the kernel itself (HBM floor of the bytes it must move; the iteration chain),
                 roofline.traffic the PMC-measured bytes per launch, roofline.streaming_kernel the streaming kernel
                 k_iter8 timed on its own after the run (the sharded mode's kernel), roofline.setup_kernel k_ref8,
                 roofline.end_to_end all algorithmic bytes of a step (16 B/px x maxiter + 24 B/px setup) over ms_per_step.
                 profiles/recompute_roofline.py derives frac from the rocprofv3 kernel trace of the same command.
  cpu_baseline = the oracle (C restatement of the reference, one thread, -O3 -msse4 -mavx) timed on the same
                 workload for one frame pair (a bounded sample), on this host's cores.

N > 1 (one rank per GPU): the points of every frame pair are sharded over the ranks (default: each rank owns 32 400
points per pair => weak scaling; --strong: ONE set of 32 400 points per pair split over the ranks), frames are
replicated, and per pair the ranks sum ONE 27-float record per GN iteration (21 floats of H, filled by a level's first
iteration, + 6 floats of b). Paths: torch.distributed / RCCL driving the streaming form's phase kernels
(invcompcamtrack_amd/dist.py ShardedTracker; the baseline that has run on real nodes), and the sharded RESIDENT form
(dist.ResidentShardedTracker: the single-GPU kernels on every rank, the sums exchanged inside the k_level_resident
launches through hipIpc-mapped mailboxes), which is tried as a tuning candidate behind three all-rank gates and used
when it passes them and is faster (--no-resident-p2p: never). `python bench.py --gpus N` starts the N ranks itself
(torch.distributed.run as a child process, before this process touches torch or the GPU) and relays rank 0's JSON line;
under torchrun (WORLD_SIZE set) it is a rank.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="independent frame pairs per step (per rank)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--levels", type=int, default=3)
    ap.add_argument("--psz", type=int, default=8)
    ap.add_argument("--maxiter", type=int, default=10)
    ap.add_argument("--variant", type=int, default=0, help="kernel-selection bits for A/B runs (include/ictr.h, ictr_odometer_set_variant)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-events", action="store_true", help="skip the in-stream HIP-event kernel timing")
    ap.add_argument("--streams", type=int, default=0,
                    help="single-GPU mode: split the B pairs over this many engines on their own HIP streams and run "
                         "them concurrently. 0 (default) = 1 with the resident-iteration form (its launches fill the "
                         "chip one at a time), 2 with the streaming kernels (variant bit 21: one engine's latency-bound "
                         "setup kernel, tails and launch gaps overlap the other's HBM-bound iteration kernel)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="one engine, host and GPU strictly alternate (the pre-pipelining behaviour)")
    ap.add_argument("--groups", type=int, default=0,
                    help="sharded mode: force this many groups of pairs per rank (0 = choose by timing)")
    ap.add_argument("--no-tune-groups", dest="tune_groups", action="store_false",
                    help="sharded mode: do not time the two-group candidates (two groups of B/2 pairs per rank whose "
                         "collectives hide behind each other's compute); by default every candidate is timed for two "
                         "steps before the warm-up and the fastest is used")
    ap.add_argument("--rccl-direct", action="store_true",
                    help="sharded mode: also try the library's own in-stream RCCL communicator (dist.RcclDirect)")
    ap.add_argument("--p2p", action="store_true",
                    help="sharded mode: also try the one-shot peer-to-peer exchange of the 27-float records (dist.P2PDirect)")
    ap.add_argument("--strong", action="store_true",
                    help="N > 1: STRONG scaling -- ONE set of 32 400 points per frame pair, split over the ranks in "
                         "contiguous blocks (dist.shard_slices; the north star's partition: the patches of one frame "
                         "shard across the GPUs), so the per-iteration all-reduce is exposed instead of amortised; "
                         "value keeps its unit (whole-job aligned Mpix/s), \"scaling\": \"strong\"")
    ap.add_argument("--no-resident-p2p", dest="auto_resident", action="store_false",
                    help="N > 1 on real GPUs: do NOT try the sharded RESIDENT form as a candidate (by default it is tried "
                         "behind three gates -- the mailboxes' lock-step set-up and self-test succeed on every rank, its "
                         "poses equal the torch.distributed candidate's to 1e-4 on every rank, and it is faster -- and "
                         "dropped on every rank together otherwise)")
    ap.add_argument("--resident-p2p", action="store_true",
                    help="sharded mode: also try the sharded RESIDENT form (dist.ResidentShardedTracker): one setup launch + "
                         "ONE k_level_resident launch per level and rank, the ranks' H / b sums exchanged inside the launch "
                         "through hipIpc-mapped mailboxes -- the single-GPU headline kernel on every rank, no collective "
                         "call between two iterations. Opt-in like --p2p: the link path has only been rehearsed with "
                         "several processes on one GPU")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the N>1 code path (sharded phases + collectives) with a world of 1 (testing)")
    ap.add_argument("--no-secondary", dest="secondary", action="store_false",
                    help="skip the short secondary workloads (psz 4, full-frame affine / homography, patch flow, "
                         "pose-sample batch, small-problem latency) reported under \"secondary\"")
    ap.add_argument("--secondary-seconds", type=float, default=1.0, help="time budget of each secondary record")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="ranks only join a gloo group and all-reduce one number (CPU test of the --gpus N launcher)")
    ap.add_argument("--rehearse-p2p", action="store_true",
                    help="N>1 rehearsal on ONE GPU: all ranks use cuda:0, gloo group for the set-up, the 27*B floats "
                         "exchanged by the one-shot peer-to-peer kernel through hipIpc-mapped mailboxes (not a benchmark)")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="N>1 rehearsal on ONE GPU: all ranks use cuda:0 and the 27*B floats are all-reduced through "
                         "host memory with gloo (same kernels, same phase sequence; numbers are not a benchmark)")
    return ap.parse_args()


def launch_ranks(args, argv):
    """`bench.py --gpus N` from a plain shell: become the launcher. Nothing here imports torch or touches HIP (a
    process that has initialised the GPU must not be replaced or forked into ranks); the ranks are fresh children of
    `python -m torch.distributed.run`. Rank 0 writes the JSON line to its stdout, which is relayed; the exit code is
    the children's."""
    import socket
    import subprocess
    with socket.socket() as s:  # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    print("[bench] launching ranks: " + " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    if proc.returncode != 0:
        print(f"[bench] ranks failed with exit code {proc.returncode}", file=sys.stderr)
        return proc.returncode
    if line is None:
        print("[bench] ranks finished without a result line", file=sys.stderr)
        return 1
    return 0


def launcher_selftest(args):
    """--selftest-launcher: the rank side of a launcher check that needs no GPU (tests/test_bench_launcher_cpu.py):
    gloo process group, one all-reduce, rank 0 prints a JSON line shaped like the real one."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if os.environ.get("ICTR_SELFTEST_FAIL_RANK") == str(rank):  # the test of "a failing rank fails the launcher"
        raise SystemExit(7)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"metric": "launcher-selftest", "n_gpus": world, "value": float(t.item()),
                          "gpus_arg": args.gpus, "steps": args.steps, "warmup": args.warmup}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


def build_inputs(args, rank, world):
    """Synthetic frames + points. Two distinct scenes are rendered; the B problems alternate between them with
    their own copies of the pyramids (own HBM), own start pose and own jittered point grid."""
    import invcompcamtrack_amd as ic
    from invcompcamtrack_amd import synth
    w, h, P = args.width, args.height, args.psz
    lv_f = args.levels - 1
    scenes = []
    for s in range(2):
        scenes.append(synth.make_scene(w, h, grid_step=P, margin=P / 2.0, jitter=0.35, seed=100 + s,
                                       tex_seed=1234 + s,
                                       dp_gt=np.array([0.02, -0.015, 0.03, 0.003, -0.002, 0.004]) * (1 + 0.5 * s)))
    n_pts = scenes[0]["pts3d"].shape[1]
    lo, hi = 0, n_pts
    if args.strong:  # this rank's block of every pair's points (the same points on every rank before the split)
        from invcompcamtrack_amd.dist import shard_slices
        lo, hi = shard_slices(n_pts, world)[rank]
        n_pts = hi - lo
    op = ic.optparam(lv_f, 0, P, args.maxiter, 0.0, 0, 0, max(n_pts, 1))
    cam = ic.CamClass(lv_f + 1, scenes[0]["fc"], scenes[0]["cc"], scenes[0]["wh"], P)
    pyrs, points = [], []
    rng = np.random.default_rng(7 + (0 if args.strong else rank))
    for b in range(args.batch):
        sc = scenes[b % 2]
        pyrs.append((ic.Pyramid(sc["img_a"], lv_f, P), ic.Pyramid(sc["img_b"], lv_f, P)))
        pts = sc["pts3d"].copy()
        if (world > 1 and not args.strong) or b >= 2:  # every rank / problem owns a different jittered sample of the same plane
            pts = pts + rng.normal(0, 1e-3, pts.shape) * np.array([[1.0], [1.0], [0.0]])
        points.append(np.ascontiguousarray(pts[:, lo:hi]))

    def make_engines(n_parts, split, variant=None):
        """split=False: n_parts engines that each hold all B pairs (they take the steps in turn);
        split=True: the B pairs divided into n_parts groups (sharded mode)."""
        variant = args.variant if variant is None else variant
        per = args.batch // n_parts if split else args.batch
        engs = [ic.TrackBatch(cam, op, per) for _ in range(n_parts)]
        for b in range(args.batch):
            if split:
                engs[b // per].Set3Dpoints(b % per, points[b].copy())
            else:
                for e in engs:
                    e.Set3Dpoints(b, points[b].copy())
        for e in engs:
            e.set_variant(variant)
            if not args.no_events:
                e.set_timing(True)
        return engs

    return dict(ic=ic, op=op, cam=cam, make_engines=make_engines, pyrs=pyrs, scenes=scenes, n_pts=n_pts)


def cpu_baseline(args, scene, n_pts):
    """Oracle (port of the reference's CPU path) on one frame pair of the same workload, single thread."""
    from oracle import oracle as O
    lv_f, P = args.levels - 1, args.psz
    op = O.make_op(lv_f, 0, P, args.maxiter, 0.0, 0, 0, n_pts)
    pa, pb = O.Pyramid(scene["img_a"], lv_f, P), O.Pyramid(scene["img_b"], lv_f, P)
    tr = O.Tracker(op, scene["fc"], scene["cc"], scene["wh"])
    pix_per_run = args.levels * args.maxiter * n_pts * P * P
    runs, t_used = 0, 0.0
    t_end = time.perf_counter() + args.cpu_seconds
    while True:
        pts = scene["pts3d"].copy()
        t0 = time.perf_counter()
        tr.set3dpoints(pts)
        tr.setpose(scene["p_a"], pa, pb)
        p_cpu = np.array(tr.trackpose(), np.float64)
        t_used += time.perf_counter() - t0
        runs += 1
        if time.perf_counter() > t_end or runs >= 1000:
            break
    cpu_trace = tr.trace()  # per-iteration H, b, delta_p, p of the last tracking (same inputs every time)
    # once more with the oracle's whole-buffer sums accumulated in float64 (oracle/ictr_oracle.c, orc_set_sum_mode):
    # what H, b and delta_p are when the summation order does not matter -- the yardstick for BOTH float32 paths
    O.lib().orc_set_sum_mode(1)
    try:
        tr.set3dpoints(scene["pts3d"].copy())
        tr.setpose(scene["p_a"], pa, pb)
        tr.trackpose()
        cpu_trace = (cpu_trace, tr.trace())
    finally:
        O.lib().orc_set_sum_mode(0)
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return p_cpu, cpu_trace, {"value": pix_per_run * runs / t_used / 1e6, "unit": "Mpix/s", "cores": 1, "kind": "port",
            "cpu_model": cpu_model, "host_cores": os.cpu_count(),
            "gn_iters_per_s": args.levels * args.maxiter * runs / t_used,
            "sample": f"{runs} full trackings of one {args.width}x{args.height} frame pair ({n_pts} points, "
                      f"{args.levels} levels x {args.maxiter} iterations), {t_used:.1f} s; oracle/libictr_oracle.so "
                      f"(C restatement of the reference, gcc -O3 -msse4 -mavx, 1 thread of {os.cpu_count()})"}


def cpu_track(scene, lv_f, psz, maxiter, ratio, n_pts, seconds=0.3, p_start=None, img_pair=None, max_runs=200):
    """cpu_baseline leg for the secondary records: the oracle (one thread) on ONE tracking of `scene` -> (pose, ms per
    tracking, runs). Pyramids are built outside the timed part, like the GPU records' device pyramids."""
    from oracle import oracle as O
    op = O.make_op(lv_f, 0, psz, maxiter, ratio, 0, 0, n_pts)
    ia, ib = img_pair if img_pair is not None else (scene["img_a"], scene["img_b"])
    pa, pb = O.Pyramid(ia, lv_f, psz), O.Pyramid(ib, lv_f, psz)
    tr = O.Tracker(op, scene["fc"], scene["cc"], scene["wh"])
    p0 = scene["p_a"] if p_start is None else p_start
    runs, t_used, p = 0, 0.0, None
    while runs < 1 or (t_used < seconds and runs < max_runs):
        pts = scene["pts3d"].copy()
        t0 = time.perf_counter()
        tr.set3dpoints(pts)
        tr.setpose(p0, pa, pb)
        p = np.array(tr.trackpose(), np.float64)
        t_used += time.perf_counter() - t0
        runs += 1
    return p, t_used / runs * 1e3, runs


def dp_vs_cpu(ic, args, scene, n_pts, cpu_trace):
    """The delta_p updates of the benchmark problem, iteration by iteration, HIP path vs CPU path (north star: "match
    the reference CPU path's delta_p updates and final pose"): one more tracking of scene 0 with the device trace on,
    outside the timed region. The first iteration starts from bit-identical inputs; later ones start from poses that
    already differ by rounding, and Gauss-Newton answers a pose offset e with an update offset ~e."""
    lv_f, P = args.levels - 1, args.psz
    op = ic.optparam(lv_f, 0, P, args.maxiter, 0.0, 0, 0, n_pts)
    cam = ic.CamClass(lv_f + 1, scene["fc"], scene["cc"], scene["wh"], P)
    odo = ic.OdometerClass(ic.PoseClass(cam, op), op)
    odo.enable_trace(True)
    odo.Set3Dpoints(scene["pts3d"].copy())
    odo.SetPose(scene["p_a"], ic.Pyramid(scene["img_a"], lv_f, P), ic.Pyramid(scene["img_b"], lv_f, P))
    odo.TrackPose()
    g = odo.trace()
    cpu_trace, cpu64 = cpu_trace
    n = min(len(g), len(cpu_trace))
    same = [(a["level"], a["iter"]) for a in g[:n]] == [(a["level"], a["iter"]) for a in cpu_trace[:n]]
    rel = lambda a, b: float(np.abs(a.astype(np.float64) - b).max() / max(np.abs(b).max(), 1e-30))
    per_it = [rel(g[i]["dp"], cpu_trace[i]["dp"]) for i in range(n)]
    traj = [float(np.abs(g[i]["p"].astype(np.float64) - cpu_trace[i]["p"]).max()) for i in range(n)]
    out = {"iterations_compared": n, "same_level_iteration_sequence": bool(same and len(g) == len(cpu_trace)),
           "max_pose_trajectory_abs_diff": max(traj) if n else None,
           "first_iteration_dp_rel": per_it[0] if n else None}
    if n:
        # Where the first update's difference comes from: both paths solve THEIR OWN 6x6 system; H and b are float32
        # sums over ~2 M pixel products (the CPU path adds them one after the other in float32, the HIP path as per-lane
        # partials, wave reductions and a fixed-order float64 tail), and the solve amplifies their relative difference
        # by up to cond(H). Solving both systems again in float64 reproduces the observed difference.
        Hg, Hc = g[0]["H"].astype(np.float64), cpu_trace[0]["H"].astype(np.float64)
        bg, bc = g[0]["b"].astype(np.float64), cpu_trace[0]["b"].astype(np.float64)
        xg, xc = np.linalg.solve(Hg, bg), np.linalg.solve(Hc, bc)
        out.update({"first_iteration_H_rel": rel(g[0]["H"], cpu_trace[0]["H"]),
                    "first_iteration_b_rel": rel(g[0]["b"], cpu_trace[0]["b"]),
                    "first_iteration_cond_H": float(np.linalg.cond(Hc)),
                    "first_iteration_dp_rel_predicted_from_H_b": float(np.abs(xg - xc).max() / np.abs(xc).max()),
                    "first_iteration_solver_rel_hip": float(np.abs(g[0]["dp"] - xg).max() / np.abs(xg).max()),
                    "first_iteration_solver_rel_cpu": float(np.abs(cpu_trace[0]["dp"] - xc).max() / np.abs(xc).max()),
                    # against the CPU path with float64 accumulation of the same float32 products (summation-order free)
                    "first_iteration_vs_f64_sums": {
                        "H_rel_hip": rel(g[0]["H"], cpu64[0]["H"]), "H_rel_cpu": rel(cpu_trace[0]["H"], cpu64[0]["H"]),
                        "b_rel_hip": rel(g[0]["b"], cpu64[0]["b"]), "b_rel_cpu": rel(cpu_trace[0]["b"], cpu64[0]["b"]),
                        "dp_rel_hip": rel(g[0]["dp"], cpu64[0]["dp"]),
                        "dp_rel_cpu": rel(cpu_trace[0]["dp"], cpu64[0]["dp"])}})
        m = min(n, len(cpu64))
        out["all_iterations_vs_f64_sums"] = {
            "iterations_compared": m,
            "max_dp_abs_diff_hip": max(float(np.abs(g[i]["dp"].astype(np.float64) - cpu64[i]["dp"]).max()) for i in range(m)),
            "max_dp_abs_diff_cpu": max(float(np.abs(cpu_trace[i]["dp"].astype(np.float64) - cpu64[i]["dp"]).max())
                                       for i in range(m)),
            "max_abs_dp_first_iteration": float(np.abs(cpu64[0]["dp"]).max()),
            "max_pose_trajectory_abs_diff_hip": max(float(np.abs(g[i]["p"].astype(np.float64) - cpu64[i]["p"]).max())
                                                    for i in range(m)),
            "final_pose_abs_diff_hip": float(np.abs(g[m - 1]["p"].astype(np.float64) - cpu64[m - 1]["p"]).max())}
    out["note"] = ("rel = |x_hip - x_cpu|_inf / |x_cpu|_inf on problem 0 of the timed steps (same frame pair, points, "
                   "options). The first iteration starts from bit-identical inputs: H and b differ by the order of "
                   "float32 summation only, delta_p by that difference times up to cond(H) (predicted_from_H_b: both "
                   "systems re-solved in float64; solver_rel_*: each path's own float32 full-pivot LU against the "
                   "float64 solve of its own system; *_vs_f64_sums: the CPU path re-run with its whole-buffer sums accumulated "
                   "in float64, orc_set_sum_mode -- the summation-order-free answer both float32 paths approximate). "
                   "Later iterations start from poses that differ by rounding, so "
                   "the trajectory (max_pose_trajectory_abs_diff) and the final pose (pose_err_vs_cpu) are the "
                   "quantities with a bar (1e-4).")
    return out


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))  # before anything touches torch.cuda / HIP in this process
    if args.selftest_launcher:
        sys.exit(launcher_selftest(args))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: the launcher's world wins",
              file=sys.stderr, flush=True)
    # The contract is ONE JSON line on stdout. Native libraries write there too (RCCL prints a five-line version
    # banner on its first collective), so everything that goes to fd 1 during the run is sent to stderr and the
    # result line is written to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    if args.rehearse_p2p:
        args.p2p = True
    one_gpu = args.rehearse_gloo or args.rehearse_p2p   # every rank on cuda:0, host-side group = gloo
    if world > 1 and one_gpu:
        import torch.distributed as dist
        local_rank = 0
        torch.cuda.set_device(0)
        dist.init_process_group("gloo")
    elif world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif args.force_sharded:
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    else:
        dist = None
        torch.cuda.set_device(0)
    from invcompcamtrack_amd import _lib
    _lib.check(_lib.load().ictr_set_device(local_rank if world > 1 else 0))

    # yardstick: what a plain streaming read reaches on this box (2 GiB, far beyond the Infinity Cache)
    import ctypes
    stream_gbps = ctypes.c_double(0.0)
    _lib.check(_lib.load().ictr_stream_read_bandwidth(2 << 30, 5, ctypes.byref(stream_gbps)))
    inp = build_inputs(args, rank, world)
    op, pyrs, scenes = inp["op"], inp["pyrs"], inp["scenes"]
    B, P, n_pts = args.batch, args.psz, inp["n_pts"]
    sharded = world > 1 or args.force_sharded
    tracker = None      # sharded mode: the ShardedTracker chosen below
    holders = None      # single-GPU mode: D step holders (D = 2: the host prepares step i+1 while the GPU runs step i)
    n_streams = 1

    class StepGroup:
        """The engines that together hold one step (B frame pairs): one engine of B / S pairs per HIP stream. With
        S = 2 the two engines run concurrently: one's latency-bound setup kernel, tails and launch gaps overlap the
        other's HBM-bound iteration kernel."""

        def __init__(self, engs):
            self.engs, self.per = engs, B // len(engs)

        def setpose_all(self):
            for b in range(B):
                pa, pb = pyrs[b]
                self.engs[b // self.per].SetPose(b % self.per, scenes[b % 2]["p_a"], pa, pb)

        def track(self):
            for e in self.engs:
                e.track_async()

        def poses(self):
            return np.concatenate([e.poses() for e in self.engs], 0)

    if sharded:
        from invcompcamtrack_amd.dist import ShardedTracker
    else:
        n_streams = args.streams if args.streams > 0 else (2 if (args.variant & (1 << 21)) else 1)
        if B % n_streams:
            raise SystemExit("--streams must divide --batch")
        depth = 1 if args.no_pipeline else 2
        streams = [torch.cuda.Stream() for _ in range(n_streams)] if n_streams > 1 else [None]
        holders = []
        for d in range(depth):
            engs = inp["make_engines"](n_streams, split=True)
            for e, st in zip(engs, streams):
                if st is not None:
                    e.set_stream(st.cuda_stream)
            holders.append(StepGroup(engs))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ev_setup = np.zeros(args.levels)
    ev_iters = np.zeros(args.levels)
    ev_kernel = np.zeros(args.levels)
    ev_first = np.zeros(args.levels)
    intervals = []      # (start_ms, end_ms) of every accumulate launch of the timed steps, all engines
    other_intervals = []  # ... and of the setup launches (the other big kernel that shares the GPU with them)

    host_t = {"setpose": 0.0, "enqueue": 0.0, "wait_poses": 0.0, "read_events": 0.0}  # host-side phases (stderr)

    def step_engines(h):
        return h.batches if sharded else h.engs

    def enqueue(i):
        """Hand step i (one batch of B frame pairs) to the GPU; returns the holder of the step."""
        h = tracker if sharded else holders[i % len(holders)]
        t_a = time.perf_counter()
        if sharded:
            engs = h.batches
            per = B // len(engs)
            for b in range(B):
                pa, pb = pyrs[b]
                engs[b // per].SetPose(b % per, scenes[b % 2]["p_a"], pa, pb)
        else:
            h.setpose_all()
        t_b = time.perf_counter()
        h.track()
        host_t["setpose"] += t_b - t_a
        host_t["enqueue"] += time.perf_counter() - t_b
        return h

    def collect(h, timed):
        """Wait for that step's tracking and fetch its poses (and, in the timed region, its event timings)."""
        nonlocal ev_setup, ev_iters, ev_kernel, ev_first
        t_a = time.perf_counter()
        poses = h.poses()
        t_b = time.perf_counter()
        if timed and not args.no_events:
            for e_ in step_engines(h):
                a, b_ = e_.level_times()  # zeros in the sharded (phase-driven) mode
                ev_setup += a
                ev_iters += b_
                ev_kernel += e_.kernel_times()
                ev_first += e_.first_iter_times()
                st_, en_ = e_.kernel_intervals()
                intervals.extend((float(x), float(y)) for x, y in zip(st_.ravel(), en_.ravel()) if y > x)
                if not sharded:
                    st_, en_ = e_.setup_intervals()
                    other_intervals.extend((float(x), float(y)) for x, y in zip(st_, en_) if y > x)
        host_t["wait_poses"] += t_b - t_a
        host_t["read_events"] += time.perf_counter() - t_b
        return poses

    def run(nsteps, timed):
        poses, pending = None, None
        for i in range(nsteps):
            h = enqueue(i)
            if sharded or len(holders) == 1:
                poses = collect(h, timed)
            else:
                if pending is not None:
                    poses = collect(pending, timed)
                pending = h
        if pending is not None:
            poses = collect(pending, timed)
        return poses

    tuning = None
    if sharded:
        # Sharded mode: one group of B pairs exposes every all-reduce; two groups of B/2 hide each other's
        # collectives behind compute (dist.sharded_program) but pay the per-iteration tail/finish launches twice.
        # Which wins depends on the collective's latency on this node, so both are timed (2 steps each, max over
        # ranks) before the warm-up and the faster one is used.
        # Candidates: (collective path, groups). "direct" = RCCL called in-stream with the library's own communicator
        # (no cross-stream event hand-off); with two groups each group has its own stream and communicator, so one
        # group's collective (link latency) overlaps the other group's kernels. "torch" = torch.distributed on its
        # own stream, with one group (collective exposed) or two (hidden behind the other group's compute).
        two = (B % 2 == 0 and B >= 2)
        # "torch" = torch.distributed (RCCL on its own stream): the default, the only path that has run on real
        # multi-GPU nodes. --rccl-direct adds the in-stream communicator as a candidate (it joins only if its
        # lock-step set-up and self-test succeed on every rank), --p2p the one-shot peer-to-peer exchange.
        auto_res = args.auto_resident and world > 1 and not one_gpu and P == 8
        modes = (["torch"] + (["direct"] if args.rccl_direct else []) + (["p2p"] if args.p2p else [])
                 + (["resident"] if (args.resident_p2p or auto_res) else []))
        if args.groups:
            cands = [(m, args.groups) for m in modes]
        else:
            # default: one group, like the single-GPU default (kernels never overlap, clean per-launch durations);
            # --tune-groups adds the two-group candidates (overlapping launches, higher throughput)
            cands = [(m, 1) for m in modes]
            if args.tune_groups and two:
                cands += [(m, 2) for m in modes if m not in ("p2p", "resident")]
        if args.rehearse_gloo:
            cands = [c for c in cands if c[0] in ("torch", "resident")]
        if args.rehearse_p2p:
            cands = [c for c in cands if c[0] in ("p2p", "resident")]
            if args.resident_p2p:
                cands = [c for c in cands if c[0] == "resident"]
        built, tuning, tune_poses = {}, {}, {}
        for mode, g in cands:
            engines = inp["make_engines"](g, split=True)
            if mode == "resident":
                from invcompcamtrack_amd.dist import ResidentShardedTracker
                tracker = ResidentShardedTracker(engines[0])
                tracker.batches, tracker.direct, tracker.p2p, tracker.resident = engines, None, None, True
                have = tracker.ok
            else:
                tracker = ShardedTracker(engines, staged=args.rehearse_gloo, direct=(mode == "direct"), p2p=(mode == "p2p"))
                have = (mode == "torch" or (mode == "direct" and tracker.direct is not None)
                        or (mode == "p2p" and tracker.p2p is not None))
            ok = torch.tensor([1.0 if have else 0.0], dtype=torch.float64,
                              device="cpu" if one_gpu else "cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)   # a path is a candidate only if every rank has it
            if float(ok.item()) < 1.0:
                continue
            key = f"{mode}-{g}"
            built[key] = (engines, tracker)
            if len(cands) > 1:
                try:
                    run(1, False)
                    barrier()
                    t_a = time.perf_counter()
                    tune_poses[key] = run(2, False)
                    barrier()
                except RuntimeError as exc:
                    # an in-launch / P2P exchange timed out: the trackers raise on EVERY rank together (dist.py), so all
                    # ranks drop the candidate here and stay in step
                    if rank == 0:
                        print(f"[bench] candidate {key} failed ({exc}); dropped", file=sys.stderr, flush=True)
                    del built[key]
                    if hasattr(tracker, "close"):
                        tracker.close()
                    continue
                t_g = torch.tensor([time.perf_counter() - t_a], dtype=torch.float64,
                                   device="cpu" if one_gpu else "cuda")
                dist.all_reduce(t_g, op=dist.ReduceOp.MAX)
                tuning[key] = float(t_g.item()) / 2 * 1e3
        if "resident-1" in tuning and "torch-1" in tune_poses:
            # gate 2: the in-launch exchange must give the collective's poses (summation order differs: float noise)
            d_ = float(np.abs(np.asarray(tune_poses["resident-1"]) - np.asarray(tune_poses["torch-1"])).max())
            okp = torch.tensor([1.0 if d_ <= 1e-4 else 0.0], dtype=torch.float64, device="cpu" if one_gpu else "cuda")
            dist.all_reduce(okp, op=dist.ReduceOp.MIN)
            if float(okp.item()) < 1.0:
                if rank == 0:
                    print(f"[bench] candidate resident-1 dropped: poses differ from torch-1 by {d_:.3e}", file=sys.stderr, flush=True)
                del tuning["resident-1"]
                built.pop("resident-1")[1].close()
        best = min(tuning, key=tuning.get) if tuning else next(iter(built))
        engines, tracker = built[best]
        built.clear()
        if rank == 0 and tuning:
            print(f"[bench] sharded-mode tuning (ms/step): {tuning} -> {best}", file=sys.stderr, flush=True)

    run(args.warmup, False)
    barrier()
    for k in host_t:
        host_t[k] = 0.0
    # a generation-2 collection of the Python GC costs ~40 ms with torch imported and used to land inside the timed
    # steps (one 42 ms stall of the SetPose loop per run): collect now, keep the collector off while timing
    import gc
    gc.collect()
    gc.disable()
    if not args.no_events:
        inp["ic"].timebase_mark()   # launch intervals of the timed steps are measured from here (device idle: barrier above)
    t0 = time.perf_counter()
    poses = run(args.steps, True)
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if rank == 0:
        print("[bench] host ms per step: " + ", ".join(f"{k} {v / args.steps * 1e3:.3f}" for k, v in host_t.items()),
              file=sys.stderr, flush=True)
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_gpu else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    iters_per_step = args.levels * args.maxiter * B           # per rank
    pix_per_iter = n_pts * P * P                               # per problem per rank
    total_pix = iters_per_step * pix_per_iter * args.steps * world
    err = float(max(np.abs(poses[b] - scenes[b % 2]["p_b"]).max() for b in range(B)))

    if rank == 0:
        out = {
            "metric": "aligned_Mpix_per_s (IC-GN, 1920x1080, 3-level pyramid)",
            "value": total_pix / dt / 1e6,
            "unit": "Mpix/s",
            "gn_iters_per_s": iters_per_step * args.steps * world / dt,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if (args.strong and sharded) else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "R1080p-dense-se3: reference-faithful 6-DoF SE(3) tracker, "
                                   f"{args.width}x{args.height}, {args.levels}-level pyramid, {n_pts} 8x8 patches "
                                   "tiling the frame, maxiter 10 fixed",
                       "frame_pairs_per_step_per_gpu": B, "points_per_pair_per_gpu": n_pts,
                       "points_per_pair_whole_job": n_pts * world if (args.strong and sharded) else None, "psz": P,
                       "levels": args.levels, "maxiter": args.maxiter, "normdp_ratio": 0.0,
                       "pixels_per_gn_iteration": pix_per_iter, "kernel_variant": args.variant,
                       "host_pipeline": ("2 step holders alternate, host one step ahead"
                                         if (not sharded and len(holders) == 2) else "none"),
                       "collective": ((("sharded RESIDENT form: H / b summed over the ranks inside the k_level_resident launches "
                                        "(one-hop mailbox exchange, no collective call)" if getattr(tracker, "resident", False)
                                        else "RCCL in-stream (own communicator)" if tracker.direct is not None
                                        else "one-shot P2P mailbox exchange" if getattr(tracker, "p2p", None) is not None
                                        else "torch.distributed (own stream)")
                                       + f", {len(tracker.batches)} group(s) of pairs per rank; tuning ms/step: {tuning}")
                                      if sharded else "n/a"),
                       "concurrent_streams": n_streams,
                       "parallelism": "single GPU" if (world == 1 and not sharded) else f"points sharded x{world}, RCCL all-reduce of "
                                                                      "one 27-float record (H 21 + b 6) per pair per iteration"},
            "pose_err_vs_ground_truth": err,
        }
        if not args.no_events:
            eng_list = tracker.batches if sharded else holders[0].engs
            pairs_per_launch = eng_list[0].B           # pairs per engine: B / streams (B / groups in the sharded mode)
            n_eng_step = len(eng_list)
            overlapping = n_eng_step > 1               # launches of different engines share the GPU
            nl = args.steps * n_eng_step * args.maxiter
            n_all = nl * args.levels
            n_first = args.steps * n_eng_step * args.levels
            # resident-iteration form: a level's iterations are ONE k_level_resident launch (no per-iteration events)
            resident_form = float(ev_kernel.sum()) == 0.0 and float(ev_iters.sum()) > 0.0
            alg_iter = 16.0 * pix_per_iter * pairs_per_launch   # one GN iteration of the engine's pairs
            # unique bytes: T, Gx, Gy are 12 B per patch pixel at every level; the current-frame texel is unique only
            # at level 0 (8-px grid = patch size). At level l the centres are 8/2^l px apart, so the 8x8 patches
            # overlap 4^l-fold and a launch touches 4 / 4^l B of distinct frame bytes per patch pixel.
            uniq_bpp = [12.0 + 4.0 / (4.0 ** l) for l in range(args.levels)]

            def load_traffic(kernel_key):
                for tname in ("traffic_r03.json", "traffic_r02.json"):
                    tpath = os.path.join(ROOT, "profiles", tname)
                    if not os.path.exists(tpath):
                        continue
                    try:
                        tj = json.load(open(tpath))
                        if tj.get("kernel", "k_iter8") == kernel_key and tj.get("batch") == pairs_per_launch and tj.get("points") == n_pts:
                            return tj.get("hbm_bytes_per_launch_mean"), (
                                f"profiles/{tname}: rocprofv3 --pmc passes of this command collected in their own runs "
                                "(FETCH_SIZE x 2 + WRITE_SIZE); static, NOT measured in this run")
                    except Exception:
                        pass
                return None, None

            def streaming_leg(engs, stream_ptr, steps):
                """k_iter8 on its own: `engs` (per-iteration launches) run on ONE stream behind a k_stream_read marker;
                mean HIP-event duration (the dispatch's own start / end stamps) of every k_iter8 launch."""
                gb = ctypes.c_double(0.0)
                _lib.check(_lib.load().ictr_stream_read_bandwidth(1 << 30, 2, ctypes.byref(gb)))
                per = engs[0].B
                tot, first, lv = 0.0, 0.0, np.zeros(args.levels)
                for st_ in range(steps + 1):
                    for k_, e_ in enumerate(engs):
                        e_.set_stream(stream_ptr)
                        for b_ in range(per):
                            pa_, pb_ = pyrs[k_ * per + b_]
                            e_.SetPose(b_, scenes[(k_ * per + b_) % 2]["p_a"], pa_, pb_)
                    for e_ in engs:
                        e_.track_async()
                    for e_ in engs:
                        e_.poses()
                        if st_ == 0:
                            continue  # warm-up
                        kt_ = e_.kernel_times()
                        lv += kt_
                        tot += float(kt_.sum())
                        first += float(e_.first_iter_times().sum())
                n = steps * len(engs) * args.maxiter * args.levels
                t = tot / max(n, 1) * 1e-3
                a_ = 16.0 * pix_per_iter * per
                return {"kernel": "k_iter8 (GN iteration: steps 7-9a), every launch of every level, running alone",
                        "pairs_per_launch": per, "algorithmic_bytes_per_launch": a_, "us_per_launch": t * 1e6,
                        "launches_timed": n, "achieved": a_ / t / 1e9, "frac": a_ / t / 1e9 / 8000.0,
                        "regular_launch_us": (tot - first) / max(steps * len(engs) * (args.maxiter - 1) * args.levels, 1) * 1e3,
                        "first_iteration_launch_us": first / max(steps * len(engs) * args.levels, 1) * 1e3,
                        "per_level_kernel_us": [float(x) / (steps * len(engs) * args.maxiter) * 1e3 for x in lv],
                        "measured_stream_read_GBps_before_leg": gb.value,
                        "basis": "same engines / kernels / grids as a tracking step, on ONE stream behind a k_stream_read "
                                 "marker (nothing overlaps): mean HIP-event duration (hipExtLaunchKernel start / stop stamps = "
                                 "the dispatch's own begin and end) of every k_iter8 launch; profiles/recompute_roofline.py "
                                 "derives the same figure from the rocprofv3 kernel trace of the same command"}

            fair = None
            if resident_form:
                # ---- headline kernel: k_level_resident, one launch per level and engine
                n_launch = args.steps * n_eng_step * args.levels
                t_launch = float(ev_iters.sum()) / max(n_launch, 1) * 1e-3
                alg = alg_iter * args.maxiter
                traffic, traffic_source = load_traffic("k_level_resident")
                roof = {"kernel": "k_level_resident: ALL Gauss-Newton iterations of a pyramid level in one launch, templates "
                                  "resident in registers / LDS (one launch per level; steps 7-10 of odometer.cpp:344-418)",
                        "algorithmic_bytes_per_launch": alg, "us_per_launch": t_launch * 1e6, "launches_timed": n_launch,
                        "achieved": alg / t_launch / 1e9, "frac": alg / t_launch / 1e9 / 8000.0,
                        "duration_basis": "HIP events on the engine's stream around each k_level_resident launch of the timed "
                                          "steps (nothing overlaps: one engine, one stream)",
                        "note": "algorithmic bytes = SURVEY.md 8d's 16 B per patch pixel and iteration x maxiter x the launch's "
                                "pixels (the contract's figure). This kernel does NOT move them -- T / Gx / Gy cross HBM once "
                                "per level and stay on the chip, an iteration reads only the current frame's windows -- so "
                                "`achieved` is an equivalent rate, exceeds the HBM peak, and says how far the form is past "
                                "the streaming formulation's roofline, not how close it is to its own bound. Its own "
                                "bounds are in `resident_bounds`: the bytes it must move (a small fraction of HBM time) and "
                                "the per-iteration dependency chain that actually sets its pace.",
                        "per_level_launch_us": [float(x) / (args.steps * n_eng_step) * 1e3 for x in ev_iters],
                        "us_per_iteration_equivalent": t_launch * 1e6 / args.maxiter}
                # what bounds THIS kernel. (1) HBM floor: T, Gx, Gy once per level (12 B per patch pixel) + every distinct
                # current-frame texel once (4 B / 4^level per patch pixel). (2) The chain: a pair's iterations are serial,
                # pairs_in_flight (the chip holds four 1080p pairs' templates) overlap: launch >= rounds x maxiter x
                # chain; chain = profiles/r03_notes.md's stamped phases of one iteration with the pair alone on the chip.
                min_bytes = float(np.mean([(12.0 + 4.0 / 4.0 ** l) * pix_per_iter * pairs_per_launch for l in range(args.levels)]))
                in_flight = 4
                chain_us = 8.2
                rounds = -(-pairs_per_launch // in_flight)
                roof["resident_bounds"] = {
                    "hbm_floor_bytes_per_launch": min_bytes, "hbm_floor_us": min_bytes / 8e12 * 1e6,
                    "hbm_floor_frac": min_bytes / 8e12 / t_launch,
                    "pairs_in_flight": in_flight, "chain_us_per_iteration_alone": chain_us,
                    "chain_floor_us_per_launch": rounds * args.maxiter * chain_us,
                    "chain_floor_frac": rounds * args.maxiter * chain_us * 1e-6 / t_launch,
                    "basis": "hbm floor: (12 + 4 / 4^level) B per patch pixel, mean over the levels, at 8 TB/s; chain floor: "
                             "ceil(pairs / 4 in flight) x maxiter x 8.2 us (stage 1 + 2 of one wave 3.5, wave reduction 0.4, "
                             "gather hop incl. the skew of 254 workgroups 2.2, solver turn 1.4, broadcast hop 0.9; "
                             "tools/restrace.py, profiles/r03_notes.md), template prologues not counted"}
                # the streaming form beside it (the kernel every other configuration runs: sharded multi-GPU mode, small
                # patch sizes, robustness options): one 16-pair engine with variant bit 21, its launches alone
                try:
                    sub = inp["make_engines"](2 if B % 2 == 0 else 1, split=True, variant=args.variant | (1 << 21))[:1]
                    leg = streaming_leg(sub, 0, 2)
                    leg["traffic"], leg["traffic_source"] = None, None
                    for tname in ("traffic_r02.json",):
                        tpath = os.path.join(ROOT, "profiles", tname)
                        if os.path.exists(tpath):
                            tj = json.load(open(tpath))
                            if tj.get("batch") == sub[0].B and tj.get("points") == n_pts:
                                leg["traffic"] = tj.get("hbm_bytes_per_launch_mean")
                                leg["traffic_source"] = f"profiles/{tname} (PMC passes, static)"
                    roof["streaming_kernel"] = leg
                    del sub
                except Exception as exc:  # never take the headline down
                    roof["streaming_kernel"] = {"error": repr(exc)}
            else:
                # ---- headline kernel: k_iter8 (per-iteration launches)
                t_all = float(ev_kernel.sum()) / max(n_all, 1) * 1e-3          # s per launch, EVERY k_iter8 launch (wall)
                traffic, traffic_source = load_traffic("k_iter8")
                if overlapping and not sharded and streams[0] is not None:
                    # Two engines on two streams: a launch's wall-clock duration includes the time it shared the GPU.
                    # roofline.frac = the UN-OVERLAPPED figure (the same engines once more on ONE stream right after the
                    # timed region); the fair-share figure of the overlapped region stays as roofline.fair_share: every
                    # launch is charged the integral of 1 / (big kernels in flight) over its interval.
                    charged = 0.0
                    if intervals:
                        evs = []
                        for x, y in intervals:
                            evs += [(x, 1, 0), (y, -1, 0)]
                        for x, y in other_intervals:
                            evs += [(x, 0, 1), (y, 0, -1)]
                        evs.sort()
                        n_it = n_ot = 0
                        t_prev = evs[0][0]
                        for t, di, do in evs:
                            if n_it > 0:
                                charged += (t - t_prev) * n_it / (n_it + n_ot)
                            t_prev = t
                            n_it += di
                            n_ot += do
                        t_fair = charged * 1e-3 / max(len(intervals), 1)
                        fair = {"achieved": alg_iter / t_fair / 1e9, "frac": alg_iter / t_fair / 1e9 / 8000.0,
                                "us_per_launch": t_fair * 1e6, "wall_us_per_launch": t_all * 1e6,
                                "overlap_factor": float(ev_kernel.sum()) / charged if charged > 0 else 1.0,
                                "launches_timed": n_all,
                                "basis": "timed region, two engines on two streams: every launch is charged the integral of "
                                         "1 / (kernels in flight) over its interval (HIP events of all k_iter8 and setup "
                                         "launches on a common time base); wall_us_per_launch is what rocprofv3 --stats "
                                         "averages over the overlapped launches"}
                    roof = streaming_leg(holders[0].engs, streams[0].cuda_stream, 4)
                    for e_, st in zip(holders[0].engs, streams):
                        e_.set_stream(st.cuda_stream)
                    roof["duration_basis"] = roof.pop("basis")
                else:
                    t_first = float(ev_first.sum()) / max(n_first, 1) * 1e-3
                    roof = {"kernel": "k_iter8 (GN iteration: steps 7-9a), every launch of every level",
                            "algorithmic_bytes_per_launch": alg_iter, "us_per_launch": t_all * 1e6, "launches_timed": n_all,
                            "achieved": alg_iter / t_all / 1e9, "frac": alg_iter / t_all / 1e9 / 8000.0,
                            "duration_basis": "mean launch duration (HIP events around each launch; nothing overlaps)",
                            "regular_launch_us": float(ev_kernel.sum() - ev_first.sum()) / max(n_all - n_first, 1) * 1e3,
                            "first_iteration_launch_us": t_first * 1e6,
                            "per_level_kernel_us": [float(x) / nl * 1e3 for x in ev_kernel]}
            # end to end: ALL algorithmic bytes of a step (16 B per patch pixel and iteration + the level setup's 12 B read
            # + 12 B written per patch pixel) over the step's wall time -- gaps, tails and host included
            e2e_bytes = float(B) * args.levels * pix_per_iter * (16.0 * args.maxiter + 24.0)
            e2e_rate = e2e_bytes / (dt / args.steps) / 1e9
            setup_us = [float(x) / (args.steps * n_eng_step) * 1e3 for x in ev_setup] if not sharded else None
            out["roofline"] = {"bound": "hbm", "achieved": roof.pop("achieved"), "peak": 8000.0, "unit": "GB/s",
                               "frac": roof.pop("frac"), "traffic": traffic, "traffic_source": traffic_source, **roof,
                               "end_to_end": {"algorithmic_bytes_per_step": e2e_bytes, "achieved": e2e_rate,
                                              "frac": e2e_rate / 8000.0,
                                              "basis": "frame pairs x levels x patch pixels x (16 B x maxiter + 24 B setup) "
                                                       "/ ms_per_step / 8 TB/s (per GPU)"},
                               "fair_share": fair,
                               "measured_stream_read_GBps": stream_gbps.value,
                               "per_level_bytes_unique_per_px": uniq_bpp,
                               "per_level_us_per_iteration_incl_tail_and_gaps":
                                   [float(x) / nl * 1e3 for x in ev_iters] if not sharded else None,
                               "per_level_setup_us": setup_us,
                               "setup_kernel": ({"kernel": "k_ref8 + k_level_tail (steps 4-6: reference patches, sd coefficients, "
                                                           "H partials from three sums per patch)",
                                                 "algorithmic_bytes_per_launch": 24.0 * pix_per_iter * pairs_per_launch,
                                                 "us_per_launch": float(np.mean(setup_us)),
                                                 "frac": 24.0 * pix_per_iter * pairs_per_launch / (float(np.mean(setup_us)) * 1e-6) / 8e12,
                                                 "basis": "HIP events around the level's setup launches in the timed steps"
                                                          + (" (two streams: they overlap the other engine's iterations)"
                                                             if overlapping else "")}
                                                if setup_us and float(np.mean(setup_us)) > 0 else None)}
        pose_fail = False
        if args.cpu_seconds > 0 and world == 1 and not sharded:
            # the same tracking on the CPU path (the oracle: checker and baseline, never the product): problem 0 is
            # scene 0 with its unmodified points, so its pose is directly comparable (north star: <= 1e-4)
            p_cpu, cpu_trace, out["cpu_baseline"] = cpu_baseline(args, scenes[0], n_pts)
            out["pose_err_vs_cpu"] = float(np.abs(np.asarray(poses[0], np.float64) - p_cpu).max())
            out["pose_err_vs_cpu_bar"] = 1e-4
            pose_fail = not (out["pose_err_vs_cpu"] <= 1e-4)
            out["dp_vs_cpu"] = dp_vs_cpu(inp["ic"], args, scenes[0], n_pts, cpu_trace)
        else:
            out["cpu_baseline"] = None  # measured on rank 0 at N=1 only
            out["pose_err_vs_cpu"] = None
        if args.secondary and world == 1 and not sharded:
            import gc as _gc
            holders = tracker = None
            _gc.collect()
            from tools import secondary as sec
            out["secondary"] = sec.run_all(args.secondary_seconds,
                                           cpu_track=cpu_track if args.cpu_seconds > 0 else None)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
        if pose_fail:
            print(f"[bench] FAIL: pose differs from the CPU path by {out['pose_err_vs_cpu']:.3e} > 1e-4",
                  file=sys.stderr, flush=True)
            sys.exit(3)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
