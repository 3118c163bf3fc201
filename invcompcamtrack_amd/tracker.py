"""Host-side mirror of the reference's tracker interface (namespace CTR) on top of the HIP C-ABI.

Same class and method names, argument meaning and call order as the reference:
  CamClass       camera.h:19-31        PoseClass      pose.h:18-40
  OdometerClass  odometer.h:21-30      optparam       utilities.h:46-61
  util_constructpyramide / util_getPatch / util_getPatch_grad / util_SE3_*   utilities.h:63-241
The driver sequence is the reference's (run_io_reprojection_test.cpp:189-223):
    cam = CamClass(lv_f+1, fc, cc, wh, psz); pose = PoseClass(cam, op); odo = OdometerClass(pose, op)
    odo.Set3Dpoints(pt3d, n); odo.SetPose(p, pyr_a, pyr_b); p_out = odo.TrackPose()
All pixel/point arithmetic runs in the HIP kernels; this module only marshals buffers.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import OptParam, TraceRec, check, dp, f32c, f64c, fp

__all__ = ["optparam", "CamClass", "PoseClass", "OdometerClass", "Pyramid", "TrackBatch", "locality_order",
           "util_constructpyramide", "util_getPatch", "util_getPatch_grad", "ncc_score", "util_SE3_coeff_to_group",
           "util_SE3_group_to_coeff", "solve6", "device_count", "timebase_mark"]


def device_count():
    return _lib.load().ictr_device_count()


def timebase_mark():
    """Set the process-wide time base of TrackBatch.kernel_intervals (synchronises the device)."""
    check(_lib.load().ictr_timebase_mark())


def optparam(lv_f, lv_l, psz, maxiter, normdp_ratio, donorm, dopatchnorm, maxpttrack, verbosity=0):
    """Fill an optparam the way the drivers do (run_io_reprojection_test.cpp:112-126), incl. the
    derived fields and the round-up of maxpttrack to a multiple of 4."""
    op = OptParam()
    check(_lib.load().ictr_optparam_init(C.byref(op), lv_f, lv_l, psz, maxiter, normdp_ratio, int(donorm),
                                         int(dopatchnorm), maxpttrack, verbosity))
    return op


class CamClass:
    """camera.h:19-31 -- per level intrinsics and (padded) sizes."""

    def __init__(self, noscales, fc, cc, wh, padding):
        L = _lib.load()
        self._h = C.c_void_p()
        self._fc, self._cc = f32c(fc), f32c(cc)
        self._wh = np.ascontiguousarray(wh, np.int32)
        self.noscales, self.padding = noscales, padding
        check(L.ictr_cam_create(C.byref(self._h), noscales, fp(self._fc), fp(self._cc),
                                self._wh.ctypes.data_as(_lib.IP), padding))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None and getattr(_lib, "load", None):
            _lib.load().ictr_cam_destroy(self._h)
            self._h = None

    def getfx(self, sc): return _lib.load().ictr_cam_getfx(self._h, sc)
    def getfy(self, sc): return _lib.load().ictr_cam_getfy(self._h, sc)
    def getcx(self, sc): return _lib.load().ictr_cam_getcx(self._h, sc)
    def getcy(self, sc): return _lib.load().ictr_cam_getcy(self._h, sc)
    def getswo(self, sc): return _lib.load().ictr_cam_getswo(self._h, sc)
    def getsho(self, sc): return _lib.load().ictr_cam_getsho(self._h, sc)
    def getsw(self, sc): return _lib.load().ictr_cam_getsw(self._h, sc)
    def getsh(self, sc): return _lib.load().ictr_cam_getsh(self._h, sc)


class Pyramid:
    """Device-resident image/gradient pyramid (util_constructpyramide, utilities.cpp:14-52)."""

    def __init__(self, img=None, lv_f=0, imgpadding=0, getgrad=True, *, device_ptr=None, wh=None, stream=None,
                 host_planes=None):
        L = _lib.load()
        self._h = C.c_void_p()
        self.lv_f, self.pad = lv_f, imgpadding
        if host_planes is not None:  # (img_pyr, dx_pyr, dy_pyr) lists of padded host planes, like the reference's arrays
            ip, xp, yp = host_planes
            n = lv_f + 1
            self._keep = [f32c(a) for a in ip], [f32c(a) for a in (xp or [])], [f32c(a) for a in (yp or [])]
            pi = (_lib.FP * n)(*[fp(a) for a in self._keep[0]])
            px = (_lib.FP * n)(*[fp(a) for a in self._keep[1]]) if xp else None
            py = (_lib.FP * n)(*[fp(a) for a in self._keep[2]]) if yp else None
            w, h = wh
            check(L.ictr_pyramid_create_from_host_planes(C.byref(self._h), pi, px, py, w, h, lv_f, imgpadding))
        elif device_ptr is not None:
            w, h = wh
            check(L.ictr_pyramid_create_device(C.byref(self._h), C.c_void_p(device_ptr), w, h, lv_f, int(getgrad),
                                               imgpadding, C.c_void_p(stream or 0)))
        else:
            img = f32c(img)
            h, w = img.shape
            check(L.ictr_pyramid_create(C.byref(self._h), fp(img), w, h, lv_f, int(getgrad), imgpadding))
        self.w, self.h = w, h

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None and getattr(_lib, "load", None):
            _lib.load().ictr_pyramid_destroy(self._h)
            self._h = None

    def rebuild(self, img=None, *, device_ptr=None, stream=None):
        """Refill this pyramid from a new frame of the same size (no allocation, the planes keep their addresses):
        ``img`` a host array, or ``device_ptr`` a frame already in device memory; enqueued on ``stream``."""
        L = _lib.load()
        if device_ptr is not None:
            check(L.ictr_pyramid_rebuild_device(self._h, C.c_void_p(device_ptr), C.c_void_p(stream or 0)))
        else:
            img = f32c(img)
            if img.shape != (self.h, self.w):
                raise ValueError(f"rebuild needs a {self.w}x{self.h} frame, got {img.shape[1]}x{img.shape[0]}")
            check(L.ictr_pyramid_rebuild(self._h, fp(img), C.c_void_p(stream or 0)))
        return self

    def level_dims(self, level):
        sw, sh = C.c_int(), C.c_int()
        check(_lib.load().ictr_pyramid_level_dims(self._h, level, C.byref(sw), C.byref(sh)))
        return sw.value, sh.value

    def download(self, level, which=0):
        """Padded plane of one level as a host array; which: 0 image, 1 dx, 2 dy."""
        sw, sh = self.level_dims(level)
        out = np.empty((sh, sw), np.float32)
        check(_lib.load().ictr_pyramid_download(self._h, level, which, fp(out)))
        return out

    def device_plane(self, level, which=0):
        return _lib.load().ictr_pyramid_device_plane(self._h, level, which)


def util_constructpyramide(img, lv_f, getgrad, imgpadding):
    """utilities.h:63-64 -- returns the device pyramid instead of filling cv::Mat arrays."""
    return Pyramid(img, lv_f, imgpadding, bool(getgrad))


def util_getPatch(pyr, level, mids, op):
    """utilities.cpp:55-113, batched: mids (K,2) -> (K, psz*psz) float32."""
    mids = np.atleast_2d(np.asarray(mids, np.float32))
    K = mids.shape[0]
    soa = np.ascontiguousarray(mids.T)
    out = np.empty((K, op.novals), np.float32)
    check(_lib.load().ictr_get_patch(pyr._h, level, fp(soa), K, op.psz, int(op.dopatchnorm), fp(out)))
    return out


def util_getPatch_grad(pyr, level, mids, op):
    """utilities.cpp:115-189, batched: returns (T, Gx, Gy), each (K, psz*psz)."""
    mids = np.atleast_2d(np.asarray(mids, np.float32))
    K = mids.shape[0]
    soa = np.ascontiguousarray(mids.T)
    o = [np.empty((K, op.novals), np.float32) for _ in range(3)]
    check(_lib.load().ictr_get_patch_grad(pyr._h, level, fp(soa), K, op.psz, int(op.dopatchnorm), fp(o[0]), fp(o[1]),
                                          fp(o[2])))
    return tuple(o)


def ncc_score(pyr_back, pyr_ref, pyr_fwd, level, mids_back, mids_ref, mids_fwd, psz, w_back, w_fwd):
    """Per-point patch correlation of run_track_nposes.cpp:271-355, on the device (ictr_ncc_score).
    mids_*: (K,2) positions at `level` in the backward-most / reference / forward-most frame -> (K,) float32."""
    mb, mr, mf = (np.atleast_2d(np.asarray(m, np.float32)) for m in (mids_back, mids_ref, mids_fwd))
    K = mr.shape[0]
    soa = np.ascontiguousarray(np.concatenate([mb.T, mr.T, mf.T], 0), np.float32)  # xb yb xr yr xf yf, K each
    out = np.empty(K, np.float32)
    check(_lib.load().ictr_ncc_score(pyr_back._h, pyr_ref._h, pyr_fwd._h, level, fp(soa), K, psz, float(w_back),
                                     float(w_fwd), fp(out)))
    return out


def util_SE3_coeff_to_group(p):
    """utilities.h:84-145; float32 in -> float32 arithmetic, float64 in -> float64 (the two instantiations)."""
    L = _lib.load()
    p = np.ascontiguousarray(p)
    if p.dtype == np.float32:
        G = np.empty(12, np.float32)
        L.ictr_se3_coeff_to_group_f(fp(G), fp(p))
    else:
        p = f64c(p)
        G = np.empty(12, np.float64)
        L.ictr_se3_coeff_to_group_d(dp(G), dp(p))
    return G


def util_SE3_group_to_coeff(G):
    """utilities.h:149-241."""
    L = _lib.load()
    G = np.ascontiguousarray(G)
    if G.dtype == np.float32:
        p = np.empty(6, np.float32)
        L.ictr_se3_group_to_coeff_f(fp(p), fp(G))
    else:
        G = f64c(G)
        p = np.empty(6, np.float64)
        L.ictr_se3_group_to_coeff_d(dp(p), dp(G))
    return p


def locality_order(pts3d, p, fc, cc):
    """A permutation that lists the points in the order of the Z-curve (Morton code) of their projection under pose p
    (6 se(3) coefficients) and the level-0 camera (fc, cc): neighbours in the list are neighbours in the image.

    The tracker keeps the caller's point order (Get2DPoints, the patch buffers and the reference's semantics are in
    that order), and the dense kernels hand 32-64 CONSECUTIVE points to a wave -- so consecutive points whose windows
    share cache lines are cheap and scattered ones are not: 32 frame pairs of 32 400 patches at 1080p take 1.00 ms per
    level-0 launch in grid order and 1.55-1.70 ms in random order (profiles/r03_notes.md). A caller whose points come
    unordered (feature detectors) can pass ``pts3d[:, order]`` to Set3Dpoints and un-permute what it reads back with
    ``inverse = np.argsort(order)``. Pure NumPy, no GPU; sums change by summation order only.
    pts3d: (3, n) float64; returns (n,) int64."""
    pts3d = np.asarray(pts3d, np.float64)
    G = util_SE3_coeff_to_group(np.asarray(p, np.float64)).reshape(3, 4)
    Xc = G[:, :3] @ pts3d + G[:, 3:4]
    z = np.where(np.abs(Xc[2]) > 1e-12, Xc[2], 1e-12)
    x = Xc[0] / z * float(fc[0]) + float(cc[0])
    y = Xc[1] / z * float(fc[1]) + float(cc[1])
    ok = np.isfinite(x) & np.isfinite(y)
    xi = np.clip(np.where(ok, x, -1.0), -1.0, 65534.0).astype(np.int64) + 1   # out-of-frame points go to the edges
    yi = np.clip(np.where(ok, y, -1.0), -1.0, 65534.0).astype(np.int64) + 1

    def spread(v):   # 16 bits -> every second bit of 32
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        return (v | (v << 1)) & 0x55555555
    return np.argsort(spread(xi) | (spread(yi) << 1), kind="stable")


def solve6(H, b):
    """Hes.fullPivLu().solve(sumsd), odometer.cpp:509-515 (host copy of the device routine)."""
    H = f32c(H).reshape(36)
    b = f32c(b)
    x = np.empty(6, np.float32)
    _lib.load().ictr_solve6(fp(H), fp(b), fp(x))
    return x


class PoseClass:
    """pose.h:18-40."""

    def __init__(self, camobj, op):
        self.camobj, self.op = camobj, op  # held by reference like pose.cpp:14-18
        self._h = C.c_void_p()
        check(_lib.load().ictr_pose_create(C.byref(self._h), camobj._h, C.byref(op)))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None and getattr(_lib, "load", None):
            _lib.load().ictr_pose_destroy(self._h)
            self._h = None

    def setpose_se3(self, p_in, meanshift_in=(0.0, 0.0, 0.0), varval_in=0.0):
        p_in, ms = f64c(p_in), f64c(meanshift_in)
        check(_lib.load().ictr_pose_setpose_se3(self._h, dp(p_in), dp(ms), float(varval_in)))

    def addpose_se3(self, p_in):
        p_in = f32c(p_in)
        check(_lib.load().ictr_pose_addpose_se3(self._h, fp(p_in)))

    def subpose_se3(self, p_in):
        p_in = f32c(p_in)
        check(_lib.load().ictr_pose_subpose_se3(self._h, fp(p_in)))

    def getPose_se3(self):
        out = np.empty(6, np.float64)
        check(_lib.load().ictr_pose_getpose_se3(self._h, dp(out)))
        return out

    def project_pt(self, pt3d, nopoints, sc, pt2d=None):
        """pt3d: float32 SoA [3*maxpttrack]; returns pt2d float32 SoA [2*maxpttrack] (pose.cpp:307-397)."""
        M = self.op.maxpttrack
        pt3d = f32c(pt3d).reshape(3 * M)
        pt2d = np.zeros(2 * M, np.float32) if pt2d is None else pt2d
        check(_lib.load().ictr_pose_project_pt(self._h, fp(pt3d), fp(pt2d), nopoints, sc))
        return pt2d

    def project_pt_save_rotated(self, pt3d, nopoints, sc):
        M = self.op.maxpttrack
        pt3d = f32c(pt3d).reshape(3 * M)
        rot = np.zeros(3 * M, np.float32)
        pt2d = np.zeros(2 * M, np.float32)
        check(_lib.load().ictr_pose_project_pt_save_rotated(self._h, fp(pt3d), fp(rot), fp(pt2d), nopoints, sc))
        return rot, pt2d

    def state(self):
        p, G = np.empty(6, np.float32), np.empty(12, np.float32)
        check(_lib.load().ictr_pose_get_state(self._h, fp(p), fp(G)))
        return p, G


class OdometerClass:
    """odometer.h:21-30 -- the Gauss-Newton tracker; every step of TrackPose runs on the GPU."""

    def __init__(self, pose_in, op_in):
        self.pose, self.op = pose_in, op_in
        self._h = C.c_void_p()
        self._keep = None
        check(_lib.load().ictr_odometer_create(C.byref(self._h), pose_in._h, C.byref(op_in)))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None and getattr(_lib, "load", None):
            _lib.load().ictr_odometer_destroy(self._h)
            self._h = None

    def set_stream(self, stream_ptr):
        check(_lib.load().ictr_odometer_set_stream(self._h, C.c_void_p(stream_ptr or 0)))

    def set_variant(self, variant):
        check(_lib.load().ictr_odometer_set_variant(self._h, int(variant)))

    def set_team(self, target_points, min_points=0, max_points=1 << 30):
        """One-launch tracker, team form (ictr_batch_set_team): workgroups per problem = ceil(n / target_points)."""
        check(_lib.load().ictr_odometer_set_team(self._h, int(target_points), int(min_points), int(max_points)))

    def set_robust(self, clean_invisible=False, compositional=False, huber_k=0.0):
        """Behaviour-changing options, off by default (see ictr_batch_set_robust)."""
        flags = (1 if clean_invisible else 0) | (2 if compositional else 0) | (4 if huber_k > 0 else 0)
        check(_lib.load().ictr_odometer_set_robust(self._h, flags, float(huber_k)))

    def Set3Dpoints(self, pt_in, nopoints_in=None):
        """pt_in: float64, C-contiguous, SoA X..Y..Z.. (shape (3,n) or flat). Mutated in place when
        op.donorm, exactly like the reference (odometer.cpp:207-212)."""
        if not (isinstance(pt_in, np.ndarray) and pt_in.dtype == np.float64 and pt_in.flags.c_contiguous):
            raise TypeError("Set3Dpoints needs a C-contiguous float64 array (it is normalised in place when donorm)")
        n = pt_in.size // 3 if nopoints_in is None else nopoints_in
        check(_lib.load().ictr_odometer_set3dpoints(self._h, dp(pt_in), n))

    def SetPose(self, p_in, img_ref, img_new):
        """p_in: 6 float64 se(3) coefficients. img_ref / img_new: Pyramid objects (device resident)."""
        p_in = f64c(p_in)
        self._keep = (img_ref, img_new)  # borrowed until the next SetPose (odometer.cpp:243-246)
        check(_lib.load().ictr_odometer_setpose(self._h, dp(p_in), img_ref._h, img_new._h))

    def SetPose_host(self, p_in, img_ref, img_ref_dx, img_ref_dy, img_new):
        """The reference's literal signature: four lists of padded host planes (one per level)."""
        p_in = f64c(p_in)
        n = self.op.lv_f + 1
        keep = [[f32c(a) for a in lst] for lst in (img_ref, img_ref_dx, img_ref_dy, img_new)]
        arrs = [(_lib.FP * n)(*[fp(a) for a in lst]) for lst in keep]
        check(_lib.load().ictr_odometer_setpose_host(self._h, dp(p_in), *arrs))

    def TrackPose(self):
        out = np.empty(6, np.float64)
        check(_lib.load().ictr_odometer_trackpose(self._h, dp(out)))
        return out

    def Get2DPoints(self):
        """SoA x[M] y[M] at level lv_l (odometer.h:30); a copy of the library-owned host mirror."""
        ptr = _lib.load().ictr_odometer_get2dpoints(self._h)
        if not ptr:
            check(1)
        return np.ctypeslib.as_array(ptr, shape=(2 * self.op.maxpttrack,)).copy()

    # ---- inspection helpers for the parity tests
    def enable_trace(self, on=True):
        check(_lib.load().ictr_odometer_enable_trace(self._h, int(on)))

    def trace(self):
        cap = (self.op.lv_f + 1) * max(1, self.op.maxiter)
        recs = (TraceRec * cap)()
        cnt = C.c_int64()
        check(_lib.load().ictr_odometer_trace(self._h, recs, cap, C.byref(cnt)))
        out = []
        for i in range(min(cnt.value, cap)):
            r = recs[i]
            out.append(dict(level=r.level, iter=r.iter, H=np.array(r.H[:], np.float32).reshape(6, 6),
                            b=np.array(r.b[:], np.float32), dp=np.array(r.dp[:], np.float32),
                            p=np.array(r.p[:], np.float32)))
        return out

    def read_buffer(self, which, count):
        out = np.empty(count, np.float32)
        check(_lib.load().ictr_odometer_read_buffer(self._h, which, fp(out), count))
        return out

    def norm(self):
        ms = np.zeros(3)
        vv = C.c_double()
        check(_lib.load().ictr_odometer_get_norm(self._h, dp(ms), C.byref(vv)))
        return ms, vv.value


class TrackBatch:
    """B independent tracking problems per launch (the pose-sample axis of run_track_nposes.cpp:193).
    All problems share the camera and optparam; each has its own points, pose and frame pair."""

    def __init__(self, camobj, op, nproblems):
        self.camobj, self.op, self.B = camobj, op, nproblems
        self._h = C.c_void_p()
        self._keep = {}
        self._keep_all = None
        check(_lib.load().ictr_batch_create(C.byref(self._h), camobj._h, C.byref(op), nproblems))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None and getattr(_lib, "load", None):
            _lib.load().ictr_batch_destroy(self._h)
            self._h = None

    def set_stream(self, stream_ptr):
        check(_lib.load().ictr_batch_set_stream(self._h, C.c_void_p(stream_ptr or 0)))

    def set_variant(self, variant):
        check(_lib.load().ictr_batch_set_variant(self._h, int(variant)))

    def set_team(self, target_points, min_points=0, max_points=1 << 30):
        """One-launch tracker, team form (ictr_batch_set_team): workgroups per problem = ceil(n / target_points)."""
        check(_lib.load().ictr_batch_set_team(self._h, int(target_points), int(min_points), int(max_points)))

    def set_peer_exchange(self, p2p_handle):
        """Sharded resident form (ictr_batch_set_peer_exchange): this batch holds one rank's shard of every problem's
        points; its resident-iteration launches sum H and b over the ranks inside the launch through the mailboxes of a
        connected ictr_p2p object (None: off)."""
        check(_lib.load().ictr_batch_set_peer_exchange(self._h, p2p_handle))

    def set_robust(self, clean_invisible=False, compositional=False, huber_k=0.0):
        """Behaviour-changing options, off by default (see ictr_batch_set_robust)."""
        flags = (1 if clean_invisible else 0) | (2 if compositional else 0) | (4 if huber_k > 0 else 0)
        check(_lib.load().ictr_batch_set_robust(self._h, flags, float(huber_k)))

    def Set3Dpoints(self, problem, pt_in, nopoints_in=None):
        if not (isinstance(pt_in, np.ndarray) and pt_in.dtype == np.float64 and pt_in.flags.c_contiguous):
            raise TypeError("Set3Dpoints needs a C-contiguous float64 array")
        n = pt_in.size // 3 if nopoints_in is None else nopoints_in
        check(_lib.load().ictr_batch_set3dpoints(self._h, problem, dp(pt_in), n))

    def Set3Dpoints_norm(self, problem, pt_in, meanshift, varval, nopoints_in=None):
        """Set3Dpoints with a caller-supplied (global) normalisation; see dist.sharded_set3dpoints."""
        if not (isinstance(pt_in, np.ndarray) and pt_in.dtype == np.float64 and pt_in.flags.c_contiguous):
            raise TypeError("Set3Dpoints needs a C-contiguous float64 array")
        n = pt_in.size // 3 if nopoints_in is None else nopoints_in
        ms = f64c(meanshift)
        check(_lib.load().ictr_batch_set3dpoints_norm(self._h, problem, dp(pt_in), n, dp(ms), float(varval)))

    def norm(self, problem):
        ms = np.zeros(3)
        vv = C.c_double()
        check(_lib.load().ictr_batch_get_norm(self._h, problem, dp(ms), C.byref(vv)))
        return ms, vv.value

    def SetPose(self, problem, p_in, img_ref, img_new):
        p_in = f64c(p_in)
        self._keep[problem] = (img_ref, img_new)
        check(_lib.load().ictr_batch_setpose(self._h, problem, dp(p_in), img_ref._h, img_new._h))

    def SetPoseAll(self, poses, img_ref, img_new):
        """SetPose for every problem in one call: ``poses`` (B, 6), all on the same frame pair (pose samples of
        run_track_nposes.cpp:232-258) -- one library call instead of B."""
        poses = f64c(poses)
        if poses.shape != (self.B, 6):
            raise ValueError(f"SetPoseAll needs a ({self.B}, 6) array of poses, got {poses.shape}")
        self._keep.clear()
        self._keep_all = (img_ref, img_new)  # borrowed until the next SetPose of every problem (no loop over 500 problems here)
        check(_lib.load().ictr_batch_setpose_all(self._h, dp(poses), img_ref._h, img_new._h))

    def track_async(self):
        check(_lib.load().ictr_batch_track_async(self._h))

    def poses(self):
        out = np.empty((self.B, 6), np.float64)
        check(_lib.load().ictr_batch_get_poses(self._h, dp(out)))
        return out

    def iterations(self):
        out = np.zeros(self.B, np.int32)
        check(_lib.load().ictr_batch_get_iterations(self._h, out.ctypes.data_as(_lib.IP)))
        return out

    def Get2DPoints(self, problem):
        out = np.empty(2 * self.op.maxpttrack, np.float32)
        check(_lib.load().ictr_batch_get2dpoints(self._h, problem, fp(out)))
        return out

    def read_buffer(self, problem, which, count):
        out = np.empty(count, np.float32)
        check(_lib.load().ictr_batch_read_buffer(self._h, problem, which, fp(out), count))
        return out

    def set_timing(self, on=True):
        check(_lib.load().ictr_batch_set_timing(self._h, int(on)))

    def level_times(self):
        """(ms_setup[l], ms_iters[l]) of the last completed track, measured with HIP events on the batch's stream."""
        n = self.op.lv_f + 1
        a, b = np.zeros(n, np.float32), np.zeros(n, np.float32)
        check(_lib.load().ictr_batch_get_level_times(self._h, fp(a), fp(b)))
        return a, b

    def kernel_times(self):
        """ms per level spent in the accumulate-kernel launches alone (comparable with rocprofv3 --stats)."""
        a = np.zeros(self.op.lv_f + 1, np.float32)
        check(_lib.load().ictr_batch_get_kernel_times(self._h, fp(a)))
        return a

    def first_iter_times(self):
        """ms per level of the first accumulate launch alone (it also sums H on the 8x8 fast path)."""
        a = np.zeros(self.op.lv_f + 1, np.float32)
        check(_lib.load().ictr_batch_get_first_iter_times(self._h, fp(a)))
        return a

    def kernel_intervals(self):
        """(start_ms, end_ms), each [levels, max(1, maxiter)], of the accumulate launches of the last tracking, in ms
        since ``timebase_mark()``; zeros for launches that did not run."""
        n, m = self.op.lv_f + 1, max(1, self.op.maxiter)
        a, b = np.zeros(n * m, np.float32), np.zeros(n * m, np.float32)
        check(_lib.load().ictr_batch_get_kernel_intervals(self._h, fp(a), fp(b)))
        return a.reshape(n, m), b.reshape(n, m)

    def setup_intervals(self):
        """(start_ms, end_ms) per level of the setup launches of the last tracking, same time base."""
        n = self.op.lv_f + 1
        a, b = np.zeros(n, np.float32), np.zeros(n, np.float32)
        check(_lib.load().ictr_batch_get_setup_intervals(self._h, fp(a), fp(b)))
        return a, b

    def path_name(self):
        """Launch form of the last tracking: per-iteration launches (plain, or replayed as one hipGraph) or the
        one-launch small-problem tracker."""
        name = {0: "k_iter* (per-iteration launches)", 1: "k_track1 (one launch per tracking)",
                2: "k_iter* (per-iteration launches replayed as one hipGraph)",
                3: "k_track1 (one launch per tracking, begin phase and read-back included)",
                4: "k_level_resident (one launch per level: all iterations, templates resident in registers)"}.get(
            _lib.load().ictr_batch_last_path(self._h), "?")
        team = _lib.load().ictr_batch_last_team(self._h)
        return name + (f" x {team} workgroups per problem" if team > 1 else "")

    def last_team(self):
        """Workgroups per problem of the last tracking when it ran as one launch (1 otherwise)."""
        return int(_lib.load().ictr_batch_last_team(self._h))

    def set_reduction_buffer(self, dev_ptr):
        check(_lib.load().ictr_batch_set_reduction_buffer(self._h, C.c_void_p(dev_ptr or 0)))

    # phase-by-phase driving (used by the sharded multi-GPU path, invcompcamtrack_amd/dist.py)
    def enable_sharding(self, on=True):
        check(_lib.load().ictr_batch_enable_sharding(self._h, int(on)))

    def reduction_buffer_ptr(self):
        return _lib.load().ictr_batch_reduction_buffer(self._h)

    @property
    def needs_level_allreduce(self):
        """False on the 8x8 fast path: H is accumulated by the level's first iteration launch and travels with
        that iteration's b, so the level phase needs no collective of its own."""
        return bool(_lib.load().ictr_batch_level_allreduce_needed(self._h))

    def begin(self): check(_lib.load().ictr_batch_begin(self._h))
    def level_accumulate(self, level): check(_lib.load().ictr_batch_level_accumulate(self._h, level))
    def level_finish(self, level): check(_lib.load().ictr_batch_level_finish(self._h, level))
    def iter_accumulate(self, level): check(_lib.load().ictr_batch_iter_accumulate(self._h, level))
    def iter_finish(self, level): check(_lib.load().ictr_batch_iter_finish(self._h, level))
