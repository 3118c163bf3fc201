"""Launcher of the row-band sharded full-frame alignment (BASELINE config 5: 8-parameter homography, 4K frames,
4-level pyramid, N GPUs; also config 3's 1080p affine) -- ``icgn.run_sharded`` behind a command line.

    python -m invcompcamtrack_amd.run_align_sharded --gpus N [--config c5|c3|small] [--steps K] [--rehearse-gloo]

Every rank holds both frames (its own pyramids), owns the band ``icgn.shard_rows(...)[rank]`` of template rows and
contributes its partial H (once per level) and b (once per iteration); the ranks all-reduce ONE 44-float record per
problem (36 H + 8 b) between the accumulate and the finish phase, then every rank solves and composes the same warp
(no broadcast). With --gpus N > 1 the N ranks are started here (torch.distributed.run as a child process, before this
process touches torch or HIP; 127.0.0.1 rendezvous). --rehearse-gloo: all ranks on cuda:0 and the records all-reduced
through host memory (one-GPU boxes; the numbers are then no benchmark). Rank 0 prints one JSON line: ms per alignment,
aligned Gpix-iterations/s of the whole job, corner error against the ground-truth warp and against the unsharded engine.
Extension (the reference has no 2-D warp): parity unpinned by the reference; oracle ``oracle/np_icgn.py``.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

CONFIGS = {
    "c5": dict(w=3840, h=2160, model="homography", p=[0.002, -0.001, 2e-6, 0.002, -0.002, -3e-6, 3.1, -2.2], lv_f=3, B=4),
    "c3": dict(w=1920, h=1080, model="affine", p=[0.003, -0.002, 0.004, -0.003, 3.1, -2.2], lv_f=2, B=8),
    "small": dict(w=320, h=256, model="homography", p=[0.004, -0.003, 1e-5, 0.003, -0.002, -1e-5, 1.5, -1.0], lv_f=2, B=2),
}


def _launch(args, argv):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(args.port), "-m", "invcompcamtrack_amd.run_align_sharded"] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def corner_err(M, Mgt, w, h):
    c4 = np.array([[0, 0, 1], [w, 0, 1], [0, h, 1], [w, h, 1.0]]).T
    x, y = M @ c4, Mgt @ c4
    return float(np.abs(x[:2] / x[2] - y[:2] / y[2]).max())


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", default="c5", choices=sorted(CONFIGS))
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--maxiter", type=int, default=10)
    ap.add_argument("--pad", type=int, default=16)
    ap.add_argument("--port", type=int, default=29561)
    ap.add_argument("--rehearse-gloo", action="store_true")
    args = ap.parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return _launch(args, argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if args.rehearse_gloo else int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.rehearse_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import invcompcamtrack_amd as ic
    from invcompcamtrack_amd import _lib, icgn
    _lib.check(_lib.load().ictr_set_device(local_rank))
    c = CONFIGS[args.config]
    w, h, lv_f, B = c["w"], c["h"], c["lv_f"], c["B"]
    Cn = np.array([[1, 0, w / 2], [0, 1, h / 2], [0, 0, 1.0]])
    Mgt = Cn @ icgn.warp_matrix(c["model"], c["p"]) @ np.linalg.inv(Cn)
    eng = icgn.AlignBatch(c["model"], w, h, lv_f, 0, args.maxiter, 0.0, None, B)
    ref = icgn.AlignBatch(c["model"], w, h, lv_f, 0, args.maxiter, 0.0, None, B) if rank == 0 else None
    keep = []
    for k in range(B):
        a, b = icgn.make_warped_pair(w, h, Mgt, seed=100 + k)
        pa, pb = ic.Pyramid(a, lv_f, args.pad), ic.Pyramid(b, lv_f, args.pad, getgrad=False)
        eng.set_frames(k, pa, pb)
        if ref is not None:
            ref.set_frames(k, pa, pb)
        keep.append((pa, pb))
    lo, hi = icgn.shard_rows(2, h - 2, world)[rank]
    eng.set_rows(lo, hi)
    red = torch.zeros(B * icgn.RED_STRIDE, dtype=torch.float32, device="cuda")
    eng.enable_sharding(red.data_ptr())
    host = torch.zeros(B * icgn.RED_STRIDE, dtype=torch.float32) if (world > 1 and args.rehearse_gloo) else None

    def allreduce():
        if world == 1:
            return
        torch.cuda.synchronize()          # the engine's launches run on its own stream
        if host is not None:
            host.copy_(red)
            dist.all_reduce(host)
            red.copy_(host)
        else:
            dist.all_reduce(red)
        torch.cuda.synchronize()

    def one():
        for k in range(B):
            eng.set_warp(k, None)
        icgn.run_sharded(eng, lv_f, 0, args.maxiter, allreduce)
        return eng.results()

    M, it, _ = one()                      # warm-up + the result that is checked
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = (time.perf_counter() - t0) / max(args.steps, 1)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.rehearse_gloo else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # every rank must hold the same warps (same bits: same sums, same solve)
        mine = torch.from_numpy(M.reshape(-1).copy())
        mine = mine if args.rehearse_gloo else mine.cuda()
        allm = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allm, mine)
        same = all(bool(torch.equal(allm[0], q)) for q in allm)
    else:
        same = True
    rc = 0
    if rank == 0:
        ref.run_async()
        Mref, itref, _ = ref.results()
        npx = [((w - 4) >> l) * ((h - 4) >> l) for l in range(lv_f + 1)]
        out = {"config": args.config, "model": c["model"], "frame": [w, h], "levels": lv_f + 1, "pairs": B,
               "n_gpus": world, "maxiter": args.maxiter, "row_band_of_rank0": [lo, hi],
               "collective": ("none" if world == 1 else "gloo through host memory (one-GPU rehearsal)" if args.rehearse_gloo
                              else "RCCL all-reduce of 44 floats per pair, once per level + once per iteration"),
               "ms_per_alignment_batch": dt * 1e3,
               "gpix_iterations_per_s": B * sum(npx) * args.maxiter / dt / 1e9,
               "iterations": int(it[0]), "ranks_agree_bitwise": bool(same),
               "corner_err_px_vs_ground_truth": max(corner_err(M[k], Mgt, w, h) for k in range(B)),
               "corner_err_px_vs_unsharded": max(corner_err(M[k], Mref[k], w, h) for k in range(B)),
               "parity": "unpinned by the reference (extension); unsharded engine vs oracle/np_icgn.py in tests/test_gpu_icgn.py"}
        print(json.dumps(out), flush=True)
        if not (same and out["corner_err_px_vs_unsharded"] < 1e-3):
            rc = 3
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
