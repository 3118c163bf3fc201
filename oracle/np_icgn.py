"""NumPy (float64) restatement of the full-frame inverse-compositional alignment of csrc/ictr_icgn.hip.

TEST INFRASTRUCTURE ONLY. The warp models (translation / SE(2) / affine / homography) do not exist in the reference
(its only warp is the SE(3) reprojection, odometer.cpp:193-300), so this oracle pins the HIP engine against an
independent implementation of the same build-defined algorithm, not against the reference: "parity unpinned by the
reference". Inputs are padded pyramid planes (oracle.Pyramid or downloads of the GPU pyramid)."""
from __future__ import annotations

import numpy as np

NP = {0: 2, 1: 3, 2: 6, 3: 8}


def param_matrix(model, p):
    W = np.eye(3)
    if model == 0:
        W[0, 2], W[1, 2] = p
    elif model == 1:
        c, s = np.cos(p[0]), np.sin(p[0])
        W[:2, :2] = [[c, -s], [s, c]]
        W[0, 2], W[1, 2] = p[1], p[2]
    elif model == 2:
        W[0, 0] += p[0]; W[1, 0] = p[1]; W[0, 1] = p[2]; W[1, 1] += p[3]; W[0, 2] = p[4]; W[1, 2] = p[5]
    else:
        W[0, 0] += p[0]; W[1, 0] = p[1]; W[2, 0] = p[2]; W[0, 1] = p[3]; W[1, 1] += p[4]; W[2, 1] = p[5]
        W[0, 2] = p[6]; W[1, 2] = p[7]
    return W


def sd_images(model, gx, gy, x, y):
    if model == 0:
        return [gx, gy]
    if model == 1:
        return [gy * x - gx * y, gx, gy]
    if model == 2:
        return [gx * x, gy * x, gx * y, gy * y, gx, gy]
    q = -(gx * x + gy * y)
    return [gx * x, gy * x, q * x, gx * y, gy * y, q * y, gx, gy]


def level_geometry(w, h, level):
    s = 0.5 ** level
    return s * (w / 2.0 + 0.5) - 0.5, s * (h / 2.0 + 0.5) - 0.5, s * max(w, h) / 2.0


def region_at(region, level, rows=None):
    x0, y0, rw, rh = region
    x1, y1 = x0 + rw, y0 + rh
    if rows is not None:
        y0, y1 = max(y0, rows[0]), min(y1, rows[1])
    s = 1 << level
    lx0, lx1 = -(-x0 // s), x1 // s
    ly0, ly1 = -(-y0 // s), -(-y1 // s)
    return lx0, ly0, max(lx1 - lx0, 0), max(ly1 - ly0, 0)


def K_matrix(w, h):
    f = max(w, h) / 2.0
    return np.array([[f, 0, w / 2.0], [0, f, h / 2.0], [0, 0, 1.0]])


class NpEngine:
    """The phase API of the HIP engine (begin / hess_accumulate / hess_finish / iter_accumulate / iter_finish) for one
    problem; ``rows`` restricts the template to a band (row-band sharding), ``red`` is the 44-float reduction record
    the caller all-reduces between accumulate and finish."""

    def __init__(self, planes_a, planes_b, pad, w, h, model, maxiter=10, eps=0.0, region=None, M0_px=None, rows=None,
                 trace=None):
        self.pa, self.pb, self.pad, self.w, self.h, self.model = planes_a, planes_b, pad, w, h, model
        self.n, self.maxiter, self.eps, self.rows, self.trace = NP[model], maxiter, eps, rows, trace
        self.region = (2, 2, w - 4, h - 4) if region is None else tuple(region)
        self.K = K_matrix(w, h)
        self.Ki = np.linalg.inv(self.K)
        self.M0 = np.eye(3) if M0_px is None else self.Ki @ np.asarray(M0_px, np.float64) @ self.K
        self.red = np.zeros(44)

    def begin(self):
        M = self.M0 / self.M0[2, 2]
        self.M = M.astype(np.float32).astype(np.float64)
        self.total = 0

    def _tri(self):
        return np.triu_indices(self.n)

    def hess_accumulate(self, l):
        pad = self.pad
        img, dx, dy = (np.asarray(a, np.float64) for a in self.pa[l])
        self.cur = np.asarray(self.pb[l], np.float64)
        self.wl, self.hl = img.shape[1] - 2 * pad, img.shape[0] - 2 * pad
        self.cx, self.cy, self.f = level_geometry(self.w, self.h, l)
        x0, y0, rw, rh = region_at(self.region, l, self.rows)
        ys, xs = np.mgrid[y0:y0 + rh, x0:x0 + rw]
        self.T = img[ys + pad, xs + pad]
        gx, gy = dx[ys + pad, xs + pad], dy[ys + pad, xs + pad]
        self.nx, self.ny = (xs - self.cx) / self.f, (ys - self.cy) / self.f
        self.sd = np.stack(sd_images(self.model, gx, gy, self.nx, self.ny), 0).reshape(self.n, -1)
        H = self.sd @ self.sd.T
        self.red[:] = 0
        self.red[:len(self._tri()[0])] = H[self._tri()]

    def hess_finish(self, l):
        H = np.zeros((self.n, self.n))
        H[self._tri()] = self.red[:len(self._tri()[0])]
        self.H = H + np.triu(H, 1).T
        self.it, self.active = 0, self.maxiter > 0
        self.red[:] = 0

    def iter_accumulate(self, l):
        if not self.active:
            return
        M, nx, ny, f, pad, cur = self.M, self.nx, self.ny, self.f, self.pad, self.cur
        u = M[0, 0] * nx + M[0, 1] * ny + M[0, 2]
        v = M[1, 0] * nx + M[1, 1] * ny + M[1, 2]
        q = M[2, 0] * nx + M[2, 1] * ny + M[2, 2]
        px, py = u / q * f + self.cx, v / q * f + self.cy
        ok = (px >= 0) & (py >= 0) & (px <= self.wl - 1) & (py <= self.hl - 1)
        pxs, pys = np.where(ok, px, 0.0), np.where(ok, py, 0.0)
        fx, fy = np.floor(pxs), np.floor(pys)
        ax, ay = pxs - fx, pys - fy
        ix, iy = fx.astype(int) + pad, fy.astype(int) + pad
        iv = (cur[iy, ix] * (1 - ax) + cur[iy, ix + 1] * ax) * (1 - ay) + \
             (cur[iy + 1, ix] * (1 - ax) + cur[iy + 1, ix + 1] * ax) * ay
        r = np.where(ok, iv - self.T, 0.0).reshape(-1)
        self.red[36:36 + self.n] = self.sd @ r

    def iter_finish(self, l):
        if not self.active:
            return
        b = self.red[36:36 + self.n].copy()
        self.red[:] = 0
        dp = np.linalg.solve(self.H, b) * (2.0 / self.f)
        if self.trace is not None:
            self.trace.append((l, self.it, self.H.copy(), b, dp.copy()))
        M = self.M @ np.linalg.inv(param_matrix(self.model, dp))
        self.M = (M / M[2, 2]).astype(np.float32).astype(np.float64)
        self.it += 1
        self.total += 1
        self.active = self.it < self.maxiter and dp @ dp > self.eps * self.eps

    def result(self):
        Mp = self.K @ self.M @ self.Ki
        return Mp / Mp[2, 2], self.total


def align(planes_a, planes_b, pad, w, h, model, lv_f, lv_l=0, maxiter=10, eps=0.0, region=None, M0_px=None,
          trace=None):
    """planes_a: list over levels of (img, dx, dy) padded arrays; planes_b: list of padded images.
    Returns (M_px (3,3), iterations). trace (list) receives (level, it, H, b, dp) tuples."""
    e = NpEngine(planes_a, planes_b, pad, w, h, model, maxiter, eps, region, M0_px, None, trace)
    e.begin()
    for l in range(lv_f, lv_l - 1, -1):
        e.hess_accumulate(l)
        e.hess_finish(l)
        for _ in range(maxiter):
            e.iter_accumulate(l)
            e.iter_finish(l)
    return e.result()
