"""Full-frame parametric alignment (translation / SE(2) / affine / homography) -- host mirror of ``ictr_icgn_*``.

Extension: the reference has no parametric 2-D warp (its only warp is the SE(3) reprojection of 3-D points,
``odometer.cpp:193-300``); BASELINE.json's configs 1, 2, 3 and 5 name these models, so they are built on the tracker's
Gauss-Newton skeleton with the Baker-Matthews inverse-compositional update. Oracle: ``oracle/np_icgn.py``
("parity unpinned by the reference"). No CPU fallback: everything here calls the HIP library.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, fp

__all__ = ["MODELS", "AlignBatch", "align", "warp_matrix", "make_warped_pair", "shard_rows", "run_sharded"]

MODELS = {"translation": 0, "se2": 1, "affine": 2, "homography": 3}
NPARAMS = {0: 2, 1: 3, 2: 6, 3: 8}
RED_STRIDE = 44  # per problem: 36 (upper triangle of the 8x8 H, row-major over the model's N) + 8 (b)


def warp_matrix(model, p):
    """3x3 matrix of the parametrisation the engine linearises around the identity (ic_param_matrix in
    csrc/ictr_icgn.hip): translation (tx,ty); se2 (theta,tx,ty); affine column-major a11,a21,a12,a22,tx,ty (as
    deviations from I); homography h11,h21,h31,h12,h22,h32,tx,ty (deviations from I, h33 = 1)."""
    m = MODELS[model] if isinstance(model, str) else model
    p = np.asarray(p, np.float64)
    W = np.eye(3)
    if m == 0:
        W[0, 2], W[1, 2] = p
    elif m == 1:
        c, s = np.cos(p[0]), np.sin(p[0])
        W[:2, :2] = [[c, -s], [s, c]]
        W[0, 2], W[1, 2] = p[1], p[2]
    elif m == 2:
        W[0, 0] += p[0]; W[1, 0] = p[1]; W[0, 1] = p[2]; W[1, 1] += p[3]; W[0, 2] = p[4]; W[1, 2] = p[5]
    else:
        W[0, 0] += p[0]; W[1, 0] = p[1]; W[2, 0] = p[2]; W[0, 1] = p[3]; W[1, 1] += p[4]; W[2, 1] = p[5]
        W[0, 2] = p[6]; W[1, 2] = p[7]
    return W


def make_warped_pair(w, h, M_px, seed=1234):
    """Synthetic pair with A(x) = B(M x): A samples the analytic texture at the pixel grid, B samples it at
    M^-1 x' (exact re-rendering, not a resampled A). M_px: 3x3 in level-0 pixel coordinates."""
    from .synth import texture
    tex = texture(seed)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    a = tex(xs, ys).astype(np.float32)
    Mi = np.linalg.inv(np.asarray(M_px, np.float64))
    u = Mi[0, 0] * xs + Mi[0, 1] * ys + Mi[0, 2]
    v = Mi[1, 0] * xs + Mi[1, 1] * ys + Mi[1, 2]
    q = Mi[2, 0] * xs + Mi[2, 1] * ys + Mi[2, 2]
    b = tex(u / q, v / q).astype(np.float32)
    return a, b


class AlignBatch:
    """``nproblems`` independent frame pairs aligned in the same launches (the batch axis is blockIdx.y)."""

    def __init__(self, model, w, h, lv_f, lv_l=0, maxiter=10, eps=0.0, region=None, nproblems=1, stream=None):
        self.model = MODELS[model] if isinstance(model, str) else int(model)
        self.n = NPARAMS[self.model]
        self.w, self.h, self.lv_f, self.lv_l, self.maxiter, self.B = w, h, lv_f, lv_l, maxiter, nproblems
        self._h = C.c_void_p()
        reg = None
        if region is not None:
            self._reg = np.asarray(region, np.int32).copy()
            reg = self._reg.ctypes.data_as(_lib.IP)
        check(_lib.load().ictr_icgn_create(C.byref(self._h), self.model, w, h, lv_f, lv_l, maxiter, float(eps), reg,
                                           nproblems))
        self._keep = {}
        if stream:
            check(_lib.load().ictr_icgn_set_stream(self._h, C.c_void_p(stream)))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None and getattr(_lib, "load", None):
            _lib.load().ictr_icgn_destroy(self._h)
            self._h = None

    def set_frames(self, problem, pyr_tmpl, pyr_cur):
        check(_lib.load().ictr_icgn_set_frames(self._h, problem, pyr_tmpl._h, pyr_cur._h))
        self._keep[problem] = (pyr_tmpl, pyr_cur)

    def set_warp(self, problem, M_px=None):
        m = None if M_px is None else np.ascontiguousarray(M_px, np.float64).ravel()
        check(_lib.load().ictr_icgn_set_warp(self._h, problem, None if m is None else m.ctypes.data_as(_lib.DP)))

    def set_timing(self, on=True):
        check(_lib.load().ictr_icgn_set_timing(self._h, int(on)))

    def run_async(self):
        check(_lib.load().ictr_icgn_run_async(self._h))

    def results(self):
        """(warps (B,3,3) float64 in level-0 pixels, iterations (B,), last dp (B,n))."""
        M = np.empty((self.B, 9), np.float64)
        it = np.zeros(self.B, np.int32)
        dp = np.zeros((self.B, 8), np.float32)
        check(_lib.load().ictr_icgn_get_results(self._h, M.ctypes.data_as(_lib.DP), it.ctypes.data_as(_lib.IP), fp(dp)))
        return M.reshape(self.B, 3, 3), it, dp[:, :self.n]

    def kernel_times(self):
        ms = np.zeros(self.lv_f + 1, np.float32)
        check(_lib.load().ictr_icgn_get_kernel_times(self._h, fp(ms)))
        return ms

    # -- phase API for the row-band sharded form
    def set_rows(self, lo, hi):
        check(_lib.load().ictr_icgn_set_rows(self._h, int(lo), int(hi)))

    def enable_sharding(self, red_dev_ptr):
        check(_lib.load().ictr_icgn_enable_sharding(self._h, 1, C.c_void_p(red_dev_ptr)))

    def begin(self):
        check(_lib.load().ictr_icgn_begin(self._h))

    def hess_accumulate(self, level):
        check(_lib.load().ictr_icgn_hess_accumulate(self._h, level))

    def hess_finish(self, level):
        check(_lib.load().ictr_icgn_hess_finish(self._h, level))

    def iter_accumulate(self, level):
        check(_lib.load().ictr_icgn_iter_accumulate(self._h, level))

    def iter_finish(self, level):
        check(_lib.load().ictr_icgn_iter_finish(self._h, level))


def align(img_a, img_b, model="affine", lv_f=2, maxiter=10, eps=0.0, region=None, M0=None, pad=4):
    """One-pair convenience wrapper: returns (M (3,3) level-0 pixels, iterations)."""
    from .tracker import Pyramid
    h, w = np.asarray(img_a).shape
    pa, pb = Pyramid(img_a, lv_f, pad), Pyramid(img_b, lv_f, pad, getgrad=False)
    eng = AlignBatch(model, w, h, lv_f, 0, maxiter, eps, region, 1)
    eng.set_frames(0, pa, pb)
    if M0 is not None:
        eng.set_warp(0, M0)
    eng.run_async()
    M, it, _ = eng.results()
    return M[0], int(it[0])


def shard_rows(y0, y1, world):
    """Contiguous, balanced row bands of the template region: rank r owns rows [lo, hi) at level 0."""
    base, rem = divmod(y1 - y0, world)
    out, lo = [], y0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def run_sharded(engine, lv_f, lv_l, maxiter, allreduce):
    """Coarse-to-fine loop in its row-band sharded form (BASELINE config 5): every rank holds both frames, owns a
    band of template rows and contributes a partial H / b; ``allreduce()`` sums the 44-float records over ranks.
    ``engine``: an AlignBatch with sharding enabled, or a test double with the same five methods."""
    engine.begin()
    for level in range(lv_f, lv_l - 1, -1):
        engine.hess_accumulate(level)
        allreduce()
        engine.hess_finish(level)
        for _ in range(maxiter):
            engine.iter_accumulate(level)
            allreduce()
            engine.iter_finish(level)
