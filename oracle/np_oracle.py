"""Independent NumPy-float32 restatement of the reference tracker.

TEST INFRASTRUCTURE ONLY (same rules as oracle.py). Written from the reference's formulas without looking at
ictr_oracle.c's structure: vectorised over points and patch pixels, no materialised sd planes. Used to
cross-check the C oracle: element-wise quantities (projections, patches, coefficients) must agree bit for
bit, sums to float tolerance. Does not reproduce the stale-patch quirk (all test points stay in view).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def exp_se3(p, dtype=np.float32):
    """utilities.h:84-145."""
    p = np.asarray(p, dtype)
    one = dtype(1)
    w = p[3:]
    q = w * w
    sig = np.sqrt(q[0] + q[1] + q[2], dtype=dtype)
    s2, s3 = sig * sig, sig * sig * sig
    if sig > 1e-4:
        sa = np.sin(sig, dtype=dtype) / sig
        sb = (one - np.cos(sig, dtype=dtype)) / s2
        sc = (sig - np.sin(sig, dtype=dtype)) / s3
    else:
        sa = one - s2 / dtype(6) * (one - s2 / dtype(20) * (one - s2 / dtype(42)))
        sb = dtype(0.5) * (one - s2 / dtype(12) * (one - s2 / dtype(30) * (one - s2 / dtype(56))))
        sc = (one - s2 / dtype(20) * (one - s2 / dtype(42) * (one - s2 / dtype(72)))) / dtype(6)
    G = np.zeros(12, dtype)
    G[0] = one - q[1] * sb - q[2] * sb
    G[1] = w[0] * w[1] * sb - w[2] * sa
    G[2] = w[1] * sa + w[0] * w[2] * sb
    G[4] = w[2] * sa + w[0] * w[1] * sb
    G[5] = one - q[0] * sb - q[2] * sb
    G[6] = w[1] * w[2] * sb - w[0] * sa
    G[8] = w[0] * w[2] * sb - w[1] * sa
    G[9] = w[0] * sa + w[1] * w[2] * sb
    G[10] = one - q[0] * sb - q[1] * sb
    t1, t2, t3 = w[2] * sb, w[0] * w[1] * sc, w[1] * sb
    t4, t5, t6 = w[0] * w[2] * sc, w[0] * sb, w[1] * w[2] * sc
    G[3] = (one - (q[1] + q[2]) * sc) * p[0] + (t2 - t1) * p[1] + (t3 + t4) * p[2]
    G[7] = (t1 + t2) * p[0] + (one - (q[0] + q[2]) * sc) * p[1] + (t6 - t5) * p[2]
    G[11] = (t4 - t3) * p[0] + (t5 + t6) * p[1] + (one - (q[0] + q[1]) * sc) * p[2]
    return G


def project(G, X, Y, Z, fx, fy, cx, cy):
    """pose.cpp:384-391, float32, left-to-right sums."""
    tx = ((G[0] * X + G[1] * Y) + G[2] * Z) + G[3]
    ty = ((G[4] * X + G[5] * Y) + G[6] * Z) + G[7]
    tz = ((G[8] * X + G[9] * Y) + G[10] * Z) + G[11]
    return (tx / tz) * fx + cx, (ty / tz) * fy + cy, tx, ty, tz


def patches(plane, mx, my, psz):
    """utilities.cpp:55-113 for K centres at once: returns (K, psz, psz) float32."""
    mx, my = mx.astype(f32), my.astype(f32)
    p0 = np.ceil(mx + f32(.00001)).astype(np.int64)
    p1 = np.ceil(my + f32(.00001)).astype(np.int64)
    r0 = mx - np.floor(mx)
    r1 = my - np.floor(my)
    w = [r0 * r1, (f32(1) - r0) * r1, r0 * (f32(1) - r1), (f32(1) - r0) * (f32(1) - r1)]
    ii = np.arange(psz)
    col = (p0 + psz // 2)[:, None, None] + ii[None, None, :]
    row = (p1 + psz // 2)[:, None, None] + ii[None, :, None]
    a, b, c, d = plane[row, col], plane[row, col - 1], plane[row - 1, col], plane[row - 1, col - 1]
    wv = [x[:, None, None] for x in w]
    return ((wv[0] * a + wv[1] * b) + wv[2] * c) + wv[3] * d


def sd_coefs(X, Y, Z, fx, fy):
    """odometer.cpp:313-326; (K,6) x-coefficients and y-coefficients, the '1.0 +' terms in float64."""
    zsq = Z * Z
    cx = np.zeros((len(X), 6), f32)
    cy = np.zeros((len(X), 6), f32)
    cx[:, 0] = fx / Z
    cy[:, 1] = fy / Z
    cx[:, 2] = -X / zsq * fx
    cy[:, 2] = -Y / zsq * fy
    cx[:, 3] = -X * Y / zsq * fx
    cy[:, 3] = ((-(1.0 + (Y * Y / zsq).astype(np.float64))) * np.float64(fy)).astype(f32)
    cx[:, 4] = ((1.0 + (X * X / zsq).astype(np.float64)) * np.float64(fx)).astype(f32)
    cy[:, 4] = X * Y / zsq * fy
    cx[:, 5] = -Y / Z * fx
    cy[:, 5] = X / Z * fy
    return cx, cy


def track(pts3d, p_in, pyr_ref, pyr_new, cam, lv_f, lv_l, psz, maxiter, solve, compose=None, huber_k=0.0):
    """No-normalisation TrackPose (odometer.cpp:257-426) for points that stay in view.
    pyr_*: object with .img/.dx/.dy lists of padded planes; cam(which, level) -> float; solve(H,b) -> dp.
    Returns (p float32[6], trace list of dict(level, iter, H, b, dp)).
    Options of the build's robustness extension (not reference behaviour): compose(p, dp) -> p_new replaces the
    additive update (the test passes log(exp(dp) exp(p))); huber_k > 0 weights residuals min(1, k/|r|) in b."""
    X, Y, Z = (pts3d[k].astype(f32) for k in range(3))
    p = np.asarray(p_in, np.float64).astype(f32)
    G0 = exp_se3(p)
    trace = []
    _, _, Xc, Yc, Zc = project(G0, X, Y, Z, f32(1), f32(1), f32(0), f32(0))
    for sl in range(lv_f, lv_l - 1, -1):
        fx, fy, cx_, cy_ = (f32(cam(k, sl)) for k in range(4))
        mx, my, _, _, _ = project(G0, X, Y, Z, fx, fy, cx_, cy_)
        T = patches(pyr_ref.img[sl], mx, my, psz)
        Gx = patches(pyr_ref.dx[sl], mx, my, psz)
        Gy = patches(pyr_ref.dy[sl], mx, my, psz)
        cxk, cyk = sd_coefs(Xc, Yc, Zc, fx, fy)
        sd = Gx[:, None] * cxk[:, :, None, None] + Gy[:, None] * cyk[:, :, None, None]  # (K,6,P,P)
        sd[:, 0] = Gx * cxk[:, 0, None, None]
        sd[:, 1] = Gy * cyk[:, 1, None, None]
        sdf = sd.reshape(len(X), 6, -1).astype(np.float64)
        H = np.einsum("kip,kjp->ij", sdf, sdf).astype(f32)
        for it in range(maxiter):
            G = exp_se3(p)
            nx, ny, _, _, _ = project(G, X, Y, Z, fx, fy, cx_, cy_)
            I = patches(pyr_new.img[sl], nx, ny, psz)
            r = T - I
            if huber_k > 0:
                ar = np.abs(r)
                r = np.where(ar > f32(huber_k), r * (f32(huber_k) / np.where(ar > 0, ar, f32(1))), r).astype(f32)
            b = (sd * r[:, None]).reshape(len(X), 6, -1).astype(np.float64).sum((0, 2)).astype(f32)
            dp = solve(H, b)
            p = (p + dp) if compose is None else np.asarray(compose(p, dp), f32)
            trace.append(dict(level=sl, iter=it, H=H, b=b, dp=dp, p=p.copy()))
    return p, trace
