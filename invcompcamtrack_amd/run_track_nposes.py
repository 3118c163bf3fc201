"""Many-poses driver, file-compatible with the reference's run_track_nposes (run_track_nposes.cpp:133-454):

  python -m invcompcamtrack_amd.run_track_nposes input.txt output.txt [--gpus N]

For every RANSAC pose sample: gather its inlier 3-D points, track the pose forward over nFwd frame pairs and
backward over nBack frame pairs (SetPose/TrackPose chains, run_track_nposes.cpp:232-258), reproject with the final
poses and score each point by the patch NCC between the reference, the last forward and the last backward frame
(run_track_nposes.cpp:271-355). Output: per sample `noimages` lines of 6 pose coefficients and one line of per-point
correlations (io_formats.write_nposes_result).

The samples are mutually independent, so they run as ONE batch of problems per frame pair on the GPU (TrackBatch)
instead of the reference's serial loop; per problem the call order -- and therefore the state that persists between
calls (patches of points that leave the view) -- is the reference's. One reference quirk is order dependent and is
kept: run_track_nposes.cpp:281 sets op.dopatchnorm = true for the NCC patches of the FIRST sample and never resets
it, so every later sample is tracked with patch normalisation on. Hence sample 0 runs alone first when the input
asks for dopatchnorm = 0.

--gpus N: the samples are split into N contiguous ranges, one process per GPU tracks and scores its range, rank 0
gathers the results (a host-side gather of a few KB: no GPU collective on this axis) and writes the file.
"""
from __future__ import annotations

import sys

import numpy as np

from . import io_formats as iof
from .tracker import CamClass, TrackBatch, ncc_score, optparam, util_constructpyramide

__all__ = ["run", "main"]


def _track_samples(cam, op, pyrs, sample_ids, inp, out_pose, pts2d):
    """Forward + backward chains for the given samples as one batch. Fills out_pose[sid] and pts2d[sid]."""
    nback, nfwd = inp["fbframes"]
    B = len(sample_ids)
    batch = TrackBatch(cam, op, B)
    M = op.maxpttrack
    npts = []
    for k, sid in enumerate(sample_ids):
        ids = inp["inlids"][sid] - 1  # 1-based in the file (run_track_nposes.cpp:212)
        pts = np.ascontiguousarray(inp["pt3d"][ids].T)  # SoA X.. Y.. Z..
        npts.append(min(len(ids), M))
        batch.Set3Dpoints(k, pts)

    def reproject(poses):
        out = []
        batch.SetPoseAll(poses, pyrs[0], pyrs[0])  # images are dummies here (run_track_nposes.cpp:219,241,260)
        batch.begin()
        for k in range(B):
            p2 = batch.Get2DPoints(k)
            out.append(np.stack([p2[:npts[k]], p2[M:M + npts[k]]], 1))
        return out

    start = np.array([inp["poses"][sid] for sid in sample_ids], np.float64)
    refe = reproject(start)
    for k, sid in enumerate(sample_ids):
        out_pose[sid][nback] = start[k]
    # forward track (run_track_nposes.cpp:229-239)
    cpos = start.copy()
    for fr in range(nfwd):
        fr_t = fr + nback
        batch.SetPoseAll(cpos, pyrs[fr_t], pyrs[fr_t + 1])
        batch.track_async()
        cpos = batch.poses()
        for k, sid in enumerate(sample_ids):
            out_pose[sid][fr_t + 1] = cpos[k]
    forw = reproject(cpos)
    # backward track (run_track_nposes.cpp:249-258)
    cpos = start.copy()
    for fr in range(nback):
        fr_t = nback - fr
        batch.SetPoseAll(cpos, pyrs[fr_t], pyrs[fr_t - 1])
        batch.track_async()
        cpos = batch.poses()
        for k, sid in enumerate(sample_ids):
            out_pose[sid][fr_t - 1] = cpos[k]
    back = reproject(cpos)
    for k, sid in enumerate(sample_ids):
        pts2d[sid] = (back[k], refe[k], forw[k])


def _ncc(cam, op_ncc, pyrs, inp, pts2d, nsamples):
    """Patch correlation of every point of every sample (run_track_nposes.cpp:271-355): one device launch for all
    samples (k_ncc: fetch, zero-mean, unit norm, the two dot products and the weighting per point in one wave); only
    the correlations come back."""
    nback, nfwd = inp["fbframes"]
    if nsamples == 0:
        return []
    counts = [len(pts2d[sid][1]) for sid in range(nsamples)]
    back = np.concatenate([np.asarray(pts2d[sid][0], np.float32).reshape(-1, 2) for sid in range(nsamples)], 0)
    refe = np.concatenate([np.asarray(pts2d[sid][1], np.float32).reshape(-1, 2) for sid in range(nsamples)], 0)
    forw = np.concatenate([np.asarray(pts2d[sid][2], np.float32).reshape(-1, 2) for sid in range(nsamples)], 0)
    corr = ncc_score(pyrs[0], pyrs[nback], pyrs[len(pyrs) - 1], op_ncc.lv_l, back, refe, forw, op_ncc.psz,
                     nback * nback, nfwd * nfwd)  # backward-most, reference, forward-most frame; weights :321,:332
    out, lo = [], 0
    for c in counts:
        out.append(corr[lo:lo + c].astype(np.float64))
        lo += c
    return out


def partition_samples(nsamples, world):
    """Contiguous, balanced ranges of pose samples: rank r owns [lo, hi). The samples are mutually independent
    (run_track_nposes.cpp:193), so this axis needs no collective in the data path."""
    base, rem = divmod(nsamples, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def _run_local(inp, images, sample_ids):
    """Track and score the given samples on this process's GPU. Returns {sid: (corr, poses)}."""
    o = inp["op"]
    op = optparam(o["lv_f"], o["lv_l"], o["psz"], o["maxiter"], o["normdp_ratio"], o["donorm"], o["dopatchnorm"],
                  o["maxpttrack"], o["verbosity"])
    cam = CamClass(op.lv_f + 1, inp["fc"], inp["cc"], inp["wh"], op.psz)
    if images is None:
        images = [iof.read_image_gray(fn) for fn in inp["filenames"]]
    pyrs = [util_constructpyramide(im, op.lv_f, 1, op.psz) for im in images]  # run_track_nposes.cpp:160-181
    noimages = len(pyrs)
    S = len(inp["poses"])
    out_pose = [np.zeros((noimages, 6)) for _ in range(S)]
    pts2d = [None] * S
    ids = list(sample_ids)
    if ids:
        # run_track_nposes.cpp:281: dopatchnorm is switched on while sample 0 is scored and never reset, so sample 0
        # is tracked with the input's setting and EVERY later sample with patch normalisation on -- a property of
        # the sample's index, not of the rank that owns it
        first = [sid for sid in ids if sid == 0] if not op.dopatchnorm else ids
        if first:
            _track_samples(cam, op, pyrs, first, inp, out_pose, pts2d)
        rest = [sid for sid in ids if sid not in first]
        op.dopatchnorm = True
        for lo in range(0, len(rest), 4096):
            _track_samples(cam, op, pyrs, rest[lo:lo + 4096], inp, out_pose, pts2d)
    op.dopatchnorm = True
    corr = _ncc(cam, op, pyrs, inp, [pts2d[sid] for sid in ids], len(ids))
    return {sid: (corr[k], out_pose[sid]) for k, sid in enumerate(ids)}


def merge_results(parts, nsamples):
    """parts: per-rank dicts {sid: (corr, poses)} -> (out_corr, out_pose) in sample order, as the reference writes."""
    merged = {}
    for part in parts:
        for sid, v in part.items():
            if sid in merged:
                raise ValueError(f"sample {sid} was tracked by two ranks")
            merged[sid] = v
    missing = [sid for sid in range(nsamples) if sid not in merged]
    if missing:
        raise ValueError(f"samples {missing[:8]} were tracked by no rank")
    return [merged[sid][0] for sid in range(nsamples)], [merged[sid][1] for sid in range(nsamples)]


def run(inp, images=None, dist=None):
    """inp: dict from io_formats.read_nposes_input. images: optional list of grey float32 arrays (else the files are
    read). Returns (out_corr, out_pose) as the reference writes them.
    dist: an initialised torch.distributed module -> the pose samples are split over the ranks in contiguous ranges
    (one process per GPU, no collective in the data path); rank 0 gathers and returns the merged result, the other
    ranks return (None, None)."""
    S = len(inp["poses"])
    if dist is None:
        return merge_results([_run_local(inp, images, range(S))], S)
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = partition_samples(S, world)[rank]
    mine = _run_local(inp, images, range(lo, hi))
    parts = [None] * world if rank == 0 else None
    dist.gather_object(mine, parts, dst=0)
    if rank != 0:
        return None, None
    return merge_results(parts, S)


def _launch_ranks(ngpus, argv):
    """`--gpus N` from a plain shell: start N ranks (one per GPU) with torch.distributed.run before this process
    touches the GPU; rank 0 writes the output file."""
    import os
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), "-m", "invcompcamtrack_amd.run_track_nposes"] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main(argv=None):
    import os
    a = sys.argv[1:] if argv is None else list(argv)
    ngpus = 1
    if "--gpus" in a:
        i = a.index("--gpus")
        ngpus = int(a[i + 1])
        a = a[:i] + a[i + 2:]
    if len(a) != 2:
        print(__doc__)
        return 2
    if ngpus > 1 and "WORLD_SIZE" not in os.environ:
        return _launch_ranks(ngpus, a + ["--gpus", str(ngpus)])
    dist = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        from . import _lib
        # the host-side gather is tiny (poses + correlations): gloo; the GPUs never talk to each other on this axis
        dist.init_process_group("gloo")
        _lib.check(_lib.load().ictr_set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, _lib.load().ictr_device_count())))
    inp = iof.read_nposes_input(a[0])
    out_corr, out_pose = run(inp, dist=dist)
    if out_corr is not None:
        iof.write_nposes_result(a[1], out_corr, out_pose)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
