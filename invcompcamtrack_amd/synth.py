"""Seeded synthetic frame pairs + 3-D points for the tracker (tests and bench.py).

Nothing here comes from the reference's data (it ships none): the scene recipe follows
SURVEY.md §8d -- an analytic texture ``f(x,y) = 128 + sum_k a_k sin(2 pi (u_k x + v_k y) + phi_k)``
(32 components, wavelengths 8..256 px, amplitude ~ wavelength^0.5, clipped to [0,255]) painted on a
plane that is fronto-parallel to camera A. Frame A samples the plane through pose ``p_a``; frame B samples
the *same* plane through pose ``p_b`` by exact ray/plane intersection, so B is an exact re-rendering and
not a resampled A. Camera intrinsics default to the reference's own synthetic recipe
(``run_io_test.m:18-22``: fc=(1000,1200), cc=(20,30)+wh/2).
"""
from __future__ import annotations

import numpy as np

__all__ = ["se3_exp", "texture", "make_scene", "make_sequence", "grid_points"]


def se3_exp(p):
    """Closed-form exp map se(3)->[R|Vt] in float64 (same parametrisation as utilities.h:84-145:
    p = (t, omega), G row-major 3x4)."""
    p = np.asarray(p, dtype=np.float64)
    w = p[3:]
    th = np.linalg.norm(w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th > 1e-8:
        sa, sb, sc = np.sin(th) / th, (1 - np.cos(th)) / th**2, (th - np.sin(th)) / th**3
    else:
        sa, sb, sc = 1.0, 0.5, 1.0 / 6.0
    R = np.eye(3) + sa * W + sb * W @ W
    V = np.eye(3) + sb * W + sc * W @ W
    return np.hstack([R, (V @ p[:3])[:, None]])


class _Texture:
    def __init__(self, seed=1234, ncomp=32):
        rng = np.random.default_rng(seed)
        lam = np.exp(rng.uniform(np.log(8.0), np.log(256.0), ncomp))
        ang = rng.uniform(0, 2 * np.pi, ncomp)
        self.u = np.cos(ang) / lam
        self.v = np.sin(ang) / lam
        self.phi = rng.uniform(0, 2 * np.pi, ncomp)
        a = lam**0.5
        self.a = a * (32.0 / np.sqrt(0.5 * np.sum(a * a)))  # texture std = 32 grey levels around 128

    def __call__(self, x, y):
        out = np.full(np.broadcast(x, y).shape, 128.0)
        for k in range(len(self.a)):
            out += self.a[k] * np.sin(2 * np.pi * (self.u[k] * x + self.v[k] * y) + self.phi[k])
        return np.clip(out, 0.0, 255.0)


def texture(seed=1234):
    return _Texture(seed)


def grid_points(w, h, step, margin, rng=None, jitter=0.0):
    """Pixel centres on a regular grid (optionally jittered), returned as (N,2) float64 in image coords."""
    xs = np.arange(margin, w - margin + 1e-9, step)
    ys = np.arange(margin, h - margin + 1e-9, step)
    gx, gy = np.meshgrid(xs, ys)
    pts = np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float64)
    if jitter > 0 and rng is not None:
        pts += rng.uniform(-jitter, jitter, pts.shape)
    return pts


def make_scene(w, h, n_points=None, *, grid_step=None, seed=42, depth=10.0, p_a=None, dp_gt=None, fc=None, cc=None,
               tex_seed=1234, quantize=False, margin=24.0, jitter=0.35, dtype=np.float32):
    """Returns dict(img_a, img_b [h,w] f32, pts3d [3,N] f64 world coords (SoA like the reference),
    p_a, p_b (6,) f64 se(3) coefficients of world->camera A / B, fc, cc, wh).

    The tracker is called with ``SetPose(p_a, pyr(img_a), .., pyr(img_b))`` and should return ~``p_b``.
    """
    rng = np.random.default_rng(seed)
    fc = np.array([1000.0, 1200.0]) * (w / 1280.0) if fc is None else np.asarray(fc, np.float64)
    cc = np.array([20.0, 30.0]) * (w / 1280.0) + np.array([w, h]) / 2.0 if cc is None else np.asarray(cc, np.float64)
    p_a = np.array([0.3, -0.2, 0.5, 0.02, -0.03, 0.01]) if p_a is None else np.asarray(p_a, np.float64)
    dp_gt = np.array([0.03, -0.02, 0.04, 0.004, -0.003, 0.005]) if dp_gt is None else np.asarray(dp_gt, np.float64)
    p_b = p_a + dp_gt  # the reference's update is additive in se(3) coordinates (pose.cpp:116-129)
    Ga, Gb = se3_exp(p_a), se3_exp(p_b)
    Ra, ta = Ga[:, :3], Ga[:, 3]
    Rb, tb = Gb[:, :3], Gb[:, 3]
    tex = _Texture(tex_seed)

    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img_a = tex(xx, yy)
    # frame B: ray through B's pixel -> plane Z_A = depth (in camera-A coordinates) -> A's pixel -> texture
    R = Rb @ Ra.T            # X_B = R X_A + t
    t = tb - R @ ta
    rx, ry = (xx - cc[0]) / fc[0], (yy - cc[1]) / fc[1]
    n_rt = R[:, 2]           # (R^T)^T n with n = e_z  ->  third column of R^T^T... = R[:,2] dotted with X_B
    num = depth + n_rt @ t
    den = n_rt[0] * rx + n_rt[1] * ry + n_rt[2]
    lam = num / den
    XB = np.stack([lam * rx, lam * ry, lam], 0).reshape(3, -1)
    XA = R.T @ (XB - t[:, None])
    ua = XA[0] / XA[2] * fc[0] + cc[0]
    va = XA[1] / XA[2] * fc[1] + cc[1]
    img_b = tex(ua, va).reshape(h, w)
    if quantize:
        img_a, img_b = np.round(img_a), np.round(img_b)

    # 3-D points: pixels of A lifted to the plane, expressed in world coordinates
    if grid_step is not None:
        px = grid_points(w, h, grid_step, margin, rng, jitter)
    else:
        px = np.stack([rng.uniform(margin, w - margin, n_points), rng.uniform(margin, h - margin, n_points)], 1)
    XA_pts = np.stack([(px[:, 0] - cc[0]) / fc[0] * depth, (px[:, 1] - cc[1]) / fc[1] * depth,
                       np.full(len(px), depth)], 0)
    Xw = Ra.T @ (XA_pts - ta[:, None])
    return dict(img_a=img_a.astype(dtype), img_b=img_b.astype(dtype), pts3d=np.ascontiguousarray(Xw),
                px_a=px, p_a=p_a, p_b=p_b, fc=fc.astype(np.float32), cc=cc.astype(np.float32),
                wh=np.array([w, h], np.int32))


def _render(tex, w, h, fc, cc, G_ref, G_k, depth):
    """Frame seen through pose G_k of the textured plane Z = depth of the REFERENCE camera G_ref (exact ray/plane
    intersection, no resampling)."""
    Ra, ta = G_ref[:, :3], G_ref[:, 3]
    Rb, tb = G_k[:, :3], G_k[:, 3]
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    R = Rb @ Ra.T
    t = tb - R @ ta
    rx, ry = (xx - cc[0]) / fc[0], (yy - cc[1]) / fc[1]
    n_rt = R[:, 2]
    lam = (depth + n_rt @ t) / (n_rt[0] * rx + n_rt[1] * ry + n_rt[2])
    XB = np.stack([lam * rx, lam * ry, lam], 0).reshape(3, -1)
    XA = R.T @ (XB - t[:, None])
    return tex(XA[0] / XA[2] * fc[0] + cc[0], XA[1] / XA[2] * fc[1] + cc[1]).reshape(h, w)


def make_sequence(w, h, poses, ref_index, n_points, *, seed=42, depth=10.0, fc=None, cc=None, tex_seed=1234,
                  quantize=True, margin=24.0):
    """A multi-frame version of make_scene for the run_track_nposes workload: frames[k] is the plane seen through
    poses[k]; the 3-D points are pixels of frame ref_index lifted to the plane (world coordinates) together with their
    2-D positions in that frame. quantize=True rounds to 8-bit grey levels like an image file would."""
    rng = np.random.default_rng(seed)
    fc = np.array([1000.0, 1200.0]) * (w / 1280.0) if fc is None else np.asarray(fc, np.float64)
    cc = np.array([20.0, 30.0]) * (w / 1280.0) + np.array([w, h]) / 2.0 if cc is None else np.asarray(cc, np.float64)
    tex = _Texture(tex_seed)
    Gs = [se3_exp(p) for p in poses]
    frames = []
    for G in Gs:
        f = _render(tex, w, h, fc, cc, Gs[ref_index], G, depth)
        frames.append((np.round(f) if quantize else f).astype(np.float32))
    px = np.stack([rng.uniform(margin, w - margin, n_points), rng.uniform(margin, h - margin, n_points)], 1)
    Ra, ta = Gs[ref_index][:, :3], Gs[ref_index][:, 3]
    XA = np.stack([(px[:, 0] - cc[0]) / fc[0] * depth, (px[:, 1] - cc[1]) / fc[1] * depth, np.full(n_points, depth)], 0)
    Xw = Ra.T @ (XA - ta[:, None])
    return dict(frames=frames, pts3d=np.ascontiguousarray(Xw.T), px_ref=px, fc=fc.astype(np.float32),
                cc=cc.astype(np.float32), wh=np.array([w, h], np.int32))
