"""Many-poses driver, file-compatible with the reference's run_track_nposes (run_track_nposes.cpp:133-454):

  python -m invcompcamtrack_amd.run_track_nposes input.txt output.txt

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
        for k in range(B):
            batch.SetPose(k, poses[k], pyrs[0], pyrs[0])  # images are dummies here (run_track_nposes.cpp:219,241,260)
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
        for k in range(B):
            batch.SetPose(k, cpos[k], pyrs[fr_t], pyrs[fr_t + 1])
        batch.track_async()
        cpos = batch.poses()
        for k, sid in enumerate(sample_ids):
            out_pose[sid][fr_t + 1] = cpos[k]
    forw = reproject(cpos)
    # backward track (run_track_nposes.cpp:249-258)
    cpos = start.copy()
    for fr in range(nback):
        fr_t = nback - fr
        for k in range(B):
            batch.SetPose(k, cpos[k], pyrs[fr_t], pyrs[fr_t - 1])
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


def run(inp, images=None):
    """inp: dict from io_formats.read_nposes_input. images: optional list of grey float32 arrays (else the files are
    read). Returns (out_corr, out_pose) as the reference writes them."""
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
    if S > 0:
        first = [0] if not op.dopatchnorm else list(range(S))
        _track_samples(cam, op, pyrs, first, inp, out_pose, pts2d)
        rest = [s for s in range(S) if s not in first]
        op.dopatchnorm = True  # run_track_nposes.cpp:281 -- stays on for every later sample
        for lo in range(0, len(rest), 4096):
            _track_samples(cam, op, pyrs, rest[lo:lo + 4096], inp, out_pose, pts2d)
    op.dopatchnorm = True
    out_corr = _ncc(cam, op, pyrs, inp, pts2d, S)
    return out_corr, out_pose


def main(argv=None):
    a = sys.argv[1:] if argv is None else list(argv)
    if len(a) != 2:
        print(__doc__)
        return 2
    inp = iof.read_nposes_input(a[0])
    out_corr, out_pose = run(inp)
    iof.write_nposes_result(a[1], out_corr, out_pose)
    return 0


if __name__ == "__main__":
    sys.exit(main())
