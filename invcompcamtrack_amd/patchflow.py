"""Flow producer for the reference's ``misc_src/run_*OF*`` drivers.

Those drivers get their displacement fields from an external binary of the author's OF_DIS repository
(``run_OF_RGB`` / ``run_DE_RGB``, ``misc_src/run_test_OF_track.py:90-108``, ``run_OF_point_track.py.ipynb`` cell 2)
which is not part of the reference. This module replaces it with the in-tree HIP point tracker (``ictr_patchflow``:
per-patch translation inverse-compositional Lucas-Kanade, pyramidal, one wave64 per patch) and re-creates the
notebook's loop: corners -> forward / backward flow -> ``oftrack.addframe`` -> ``savetofile``.

Build-defined (no reference implementation exists to pin it): oracle = ``oracle/np_patchflow.py``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, fp
from .classoftrack import oftrack
from .tracker import Pyramid

__all__ = ["track_points", "partition_patches", "dense_flow", "good_features", "run_OF_point_track", "last_kernel_ms"]


def last_kernel_ms():
    """Duration of the k_patchflow launch of the last track_points call (HIP events), or None."""
    ms = float(_lib.load().ictr_patchflow_last_kernel_ms())
    return ms if ms >= 0 else None


def partition_patches(npatches, world):
    """Contiguous, balanced patch ranges: rank r owns [lo, hi). The patches are mutually independent (each owns its two
    parameters: SURVEY.md §8e, BASELINE config 4), so this axis needs no collective in the data path."""
    base, rem = divmod(npatches, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def track_points(pyr_a, pyr_b, pts, psz=15, lv_f=None, lv_l=0, maxiter=10, eps=0.01, dist=None, group=None, _local=None):
    """Track K points from frame A to frame B. pts: (K,2) level-0 pixel coordinates (x,y).
    Returns (new_pts (K,2) float32 with NaN rows for lost points, status (K,) bool, iters (K,) int32).

    dist (torch.distributed, initialised; optional `group`): BASELINE config 4's multi-GPU form -- every rank holds both
    frames (its own pyramids on its own GPU) and tracks the contiguous range partition_patches(K, world)[rank] of the K
    patches; rank 0 gathers the ranges in rank order and returns the full arrays, the other ranks return None. The
    result on rank 0 equals the single-process result bit for bit (a patch's arithmetic does not depend on what else
    is in its launch). `_local` replaces the GPU call in the CPU tests of the partition / gather logic."""
    pts = np.atleast_2d(np.asarray(pts, np.float32))
    if dist is not None:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        lo, hi = partition_patches(pts.shape[0], world)[rank]
        local = _local if _local is not None else (
            lambda q: track_points(pyr_a, pyr_b, q, psz, lv_f, lv_l, maxiter, eps) if len(q) else
            (np.zeros((0, 2), np.float32), np.zeros(0, bool), np.zeros(0, np.int32)))
        mine = local(pts[lo:hi])
        parts = [None] * world if rank == 0 else None
        dist.gather_object(mine, parts, dst=0, group=group)
        if rank != 0:
            return None
        return (np.concatenate([q[0] for q in parts], 0), np.concatenate([q[1] for q in parts], 0),
                np.concatenate([q[2] for q in parts], 0))
    K = pts.shape[0]
    lv_f = pyr_a.lv_f if lv_f is None else lv_f
    soa = np.ascontiguousarray(pts.T)
    out = np.empty((2, K), np.float32)
    status = np.zeros(K, np.int32)
    iters = np.zeros(K, np.int32)
    check(_lib.load().ictr_patchflow(pyr_a._h, pyr_b._h, fp(soa), K, psz, lv_f, lv_l, maxiter, float(eps), fp(out),
                                     status.ctypes.data_as(_lib.IP), iters.ctypes.data_as(_lib.IP)))
    return np.ascontiguousarray(out.T), status.astype(bool), iters


def dense_flow(pyr_a, pyr_b, step=4, psz=15, lv_f=None, maxiter=10, eps=0.01):
    """(H, W, 2) float32 displacement field A -> B: patches tracked on a `step`-pixel grid, bilinearly up-sampled;
    lost grid points take the displacement of the nearest tracked one in their row/column scan (0 if none)."""
    w, h = pyr_a.w, pyr_a.h
    xs = np.arange(step // 2, w, step, dtype=np.float32)
    ys = np.arange(step // 2, h, step, dtype=np.float32)
    gx, gy = np.meshgrid(xs, ys)
    pts = np.stack([gx.ravel(), gy.ravel()], 1)
    new, ok, _ = track_points(pyr_a, pyr_b, pts, psz=psz, lv_f=lv_f, maxiter=maxiter, eps=eps)
    d = (new - pts).reshape(len(ys), len(xs), 2)
    bad = ~ok.reshape(len(ys), len(xs))
    if bad.any():
        d[bad] = np.nan
        for axis in (1, 0):  # forward/backward fill along rows, then columns
            for rev in (False, True):
                v = d[:, ::-1] if (rev and axis == 1) else d[::-1] if rev else d
                idx = np.isnan(v[..., 0])
                pos = np.where(~idx, np.arange(v.shape[axis]).reshape((-1, 1) if axis == 0 else (1, -1)), 0)
                np.maximum.accumulate(pos, axis=axis, out=pos)
                filled = np.take_along_axis(v, pos[..., None].repeat(2, -1), axis=axis)
                v[idx] = filled[idx]
        d[np.isnan(d)] = 0.0
    # bilinear up-sampling to every pixel (grid nodes sit at step//2 + k*step)
    fx = np.clip((np.arange(w) - step // 2) / step, 0, len(xs) - 1)
    fy = np.clip((np.arange(h) - step // 2) / step, 0, len(ys) - 1)
    x0, y0 = np.floor(fx).astype(int), np.floor(fy).astype(int)
    x1, y1 = np.minimum(x0 + 1, len(xs) - 1), np.minimum(y0 + 1, len(ys) - 1)
    ax, ay = (fx - x0)[None, :, None], (fy - y0)[:, None, None]
    top = d[y0][:, x0] * (1 - ax) + d[y0][:, x1] * ax
    bot = d[y1][:, x0] * (1 - ax) + d[y1][:, x1] * ax
    return (top * (1 - ay) + bot * ay).astype(np.float32)


def good_features(img, maxcorners=1000, quality=0.001, mindist=5, win=3):
    """Shi-Tomasi style corner picker standing in for cv2.goodFeaturesToTrack(gray, 1000, 0.001, 5) of the reference's
    notebook (cv2 is not available here; the reference does not pin the detector): minimum eigenvalue of
    the windowed structure tensor, strongest first, greedy minimum distance. Returns (K,2) float32 (x,y)."""
    img = np.asarray(img, np.float64)
    gx = np.zeros_like(img)
    gy = np.zeros_like(img)
    gx[:, 1:-1] = img[:, 2:] - img[:, :-2]
    gy[1:-1, :] = img[2:, :] - img[:-2, :]

    def box(a):
        c = np.cumsum(np.cumsum(np.pad(a, ((win + 1, win), (win + 1, win))), 0), 1)
        k = 2 * win + 1
        return c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]

    sxx, sxy, syy = box(gx * gx), box(gx * gy), box(gy * gy)
    lam = 0.5 * (sxx + syy) - np.sqrt(0.25 * (sxx - syy) ** 2 + sxy * sxy)
    lam[:mindist], lam[-mindist:], lam[:, :mindist], lam[:, -mindist:] = 0, 0, 0, 0
    thr = quality * lam.max()
    # local maxima in a 3x3 neighbourhood
    p = np.pad(lam, 1, constant_values=-1)
    ismax = np.ones_like(lam, bool)
    for dy in (0, 1, 2):
        for dx in (0, 1, 2):
            if dy != 1 or dx != 1:
                ismax &= lam >= p[dy:dy + lam.shape[0], dx:dx + lam.shape[1]]
    ys, xs = np.nonzero(ismax & (lam > thr))
    order = np.argsort(-lam[ys, xs], kind="stable")
    taken = np.zeros((lam.shape[0] // mindist + 2, lam.shape[1] // mindist + 2), bool)
    out = []
    for i in order:
        cy, cx = ys[i] // mindist, xs[i] // mindist
        if taken[cy, cx]:
            continue
        taken[cy, cx] = True
        out.append((xs[i], ys[i]))
        if len(out) >= maxcorners:
            break
    return np.array(out, np.float32).reshape(-1, 2)


def run_OF_point_track(frames, bsize=10, psz=15, lv_f=3, step=4, maxcorners=1000, th_flowvalid_ratio=0.2,
                       th_flowvalid_abs=1.0, savefile=None):
    """The loop of misc_src/run_OF_point_track.py.ipynb cell 2 with the external flow binary replaced by dense_flow():
    per consecutive frame pair: forward and backward flow, new corners on the first frame of the pair, addframe.
    frames: list of grey float images. Returns the oftrack object."""
    h, w = np.asarray(frames[0]).shape
    tracker = oftrack(bsize, w, h, th_flowvalid_ratio, th_flowvalid_abs)
    pyrs = [Pyramid(np.asarray(f, np.float32), lv_f, psz, True) for f in frames]
    for k in range(len(frames) - 1):
        of_forw = dense_flow(pyrs[k], pyrs[k + 1], step=step, psz=psz, lv_f=lv_f)
        of_back = dense_flow(pyrs[k + 1], pyrs[k], step=step, psz=psz, lv_f=lv_f)
        corners = good_features(frames[k], maxcorners, 0.001, 5)
        tracker.addframe(of_forw, of_back, corners if len(corners) else None)
    if savefile is not None:
        tracker.savetofile(savefile)
    return tracker
