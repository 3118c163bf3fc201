"""NumPy restatement of the per-patch translation IC-LK of invcompcamtrack_amd/csrc/ictr_patchflow.hip.

TEST INFRASTRUCTURE ONLY. The algorithm is build-defined (the reference obtains flow from an external binary that is
not in its repository), so this oracle pins the HIP kernel against an independent implementation, not against the
reference: "parity unpinned by the reference". Sampling = np_oracle.patches (util_getPatch's convention)."""
from __future__ import annotations

import numpy as np

from . import np_oracle as N

f32 = np.float32


def track_points(pyr_a, pyr_b, pts, psz, lv_f, lv_l=0, maxiter=10, eps=0.01, min_det=1e-4):
    """pyr_*: oracle.Pyramid (host planes). pts (K,2). Returns (new (K,2) f32 with NaN, status, iters)."""
    pts = np.asarray(pts, f32)
    K = len(pts)
    out = np.full((K, 2), np.nan, f32)
    status = np.zeros(K, bool)
    iters = np.zeros(K, np.int32)
    for k in range(K):
        x0, y0 = pts[k]
        if not (np.isfinite(x0) and np.isfinite(y0)):
            continue
        p = np.zeros(2, f32)
        ok, nit = True, 0
        for l in range(lv_f, lv_l - 1, -1):
            if l != lv_f:
                p = p * f32(2)
            sc = f32(0.5 ** l)
            wl, hl = pyr_a.img[l].shape[1] - 2 * pyr_a.pad, pyr_a.img[l].shape[0] - 2 * pyr_a.pad
            xl, yl = f32(x0 * sc), f32(y0 * sc)
            if not (0 <= xl <= wl and 0 <= yl <= hl):
                ok = False
                break
            mx, my = np.array([xl], f32), np.array([yl], f32)
            T = N.patches(pyr_a.img[l], mx, my, psz)[0].astype(np.float64)
            Gx = N.patches(pyr_a.dx[l], mx, my, psz)[0].astype(np.float64)
            Gy = N.patches(pyr_a.dy[l], mx, my, psz)[0].astype(np.float64)
            hxx, hxy, hyy = (Gx * Gx).sum(), (Gx * Gy).sum(), (Gy * Gy).sum()
            det, tr = hxx * hyy - hxy * hxy, hxx + hyy
            if not (det > min_det * tr * tr) or not (tr > 0):
                ok = False
                break
            for _ in range(maxiter):
                cx, cy = f32(xl + p[0]), f32(yl + p[1])
                if not (0 <= cx <= wl and 0 <= cy <= hl):
                    ok = False
                    break
                I = N.patches(pyr_b.img[l], np.array([cx], f32), np.array([cy], f32), psz)[0].astype(np.float64)
                r = T - I
                bx, by = (Gx * r).sum(), (Gy * r).sum()
                dx, dy = (hyy * bx - hxy * by) / det, (hxx * by - hxy * bx) / det
                p = (p + np.array([dx, dy])).astype(f32)
                nit += 1
                if dx * dx + dy * dy < eps * eps:
                    break
            if not ok:
                break
        if ok:
            s = f32(2.0 ** lv_l)
            out[k] = (x0 + p[0] * s, y0 + p[1] * s)
        status[k], iters[k] = ok, nit
    return out, status, iters
