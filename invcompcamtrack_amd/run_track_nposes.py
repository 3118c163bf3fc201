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
from .tracker import CamClass, TrackBatch, optparam, util_constructpyramide, util_getPatch

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
    """Patch correlation of every point of every sample (run_track_nposes.cpp:271-355), patches fetched on the GPU."""
    nback, nfwd = inp["fbframes"]
    lv = op_ncc.lv_l
    swo, sho = cam.getswo(lv), cam.getsho(lv)
    frames = (0, nback, len(pyrs) - 1)  # backward-most, reference, forward-most frame
    out = []
    for sid in range(nsamples):
        pats, vals = [], []
        for which in range(3):
            mids = np.asarray(pts2d[sid][which], np.float32).reshape(-1, 2)
            ok = (mids[:, 0] > 0) & (mids[:, 1] > 0) & (mids[:, 0] < swo) & (mids[:, 1] < sho)  # strict, :290,:297,:305
            p = np.zeros((len(mids), op_ncc.novals), np.float32)
            if ok.any():
                p[ok] = util_getPatch(pyrs[frames[which]], lv, mids[ok], op_ncc)
            with np.errstate(invalid="ignore", divide="ignore"):
                p = p / np.sqrt(np.sum(p * p, axis=1, dtype=np.float32))[:, None]
            pats.append(p)
            vals.append(ok)
        pb, pr, pf = pats
        bv, rv, fv = vals
        w0 = np.where(bv, np.float32(nback * nback), np.float32(0))
        w1 = np.where(fv, np.float32(nfwd * nfwd), np.float32(0))
        cbr = np.where(bv, np.maximum(np.float32(0), np.nan_to_num(np.sum(pb * pr, 1, dtype=np.float32))), np.float32(-1))
        crf = np.where(fv, np.maximum(np.float32(0), np.nan_to_num(np.sum(pr * pf, 1, dtype=np.float32))), np.float32(-1))
        with np.errstate(invalid="ignore", divide="ignore"):
            c = (cbr * w0 + crf * w1) / (w0 + w1)
        c = np.where(np.isnan(c), np.float32(0), np.maximum(np.float32(0), c))  # std::max(0.0f, NaN) == 0.0f
        out.append(np.where(rv, c, np.float32(-1)).astype(np.float64))
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
