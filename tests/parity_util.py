"""Shared helpers for the GPU parity tests: run the same tracking through the HIP path (via the C-ABI) and
through the CPU oracle on identical inputs."""
import numpy as np

import invcompcamtrack_amd as ic
from invcompcamtrack_amd import synth


# OR-ed into every Pair's kernel-selection bits (tests/test_gpu_parity.py runs its cases once with the automatic
# choice -- small problems take the one-launch tracker k_track1 -- and once with bit 13 = per-iteration launches)
FORCE_VARIANT = 0
# (target, min, max) points for the one-launch tracker's team form (ictr_batch_set_team), or None = library defaults
FORCE_TEAM = None


class Pair:
    """One scene, one parameter set, both implementations wired like run_io_reprojection_test.cpp:189-193."""

    def __init__(self, O, sc, lv_f, lv_l, psz, maxiter, ratio, donorm, dpn, maxpt=None, variant=0):
        n = sc["pts3d"].shape[1]
        maxpt = n if maxpt is None else maxpt
        self.O, self.sc = O, sc
        self.oop = O.make_op(lv_f, lv_l, psz, maxiter, ratio, donorm, dpn, maxpt)
        self.opa, self.opb = O.Pyramid(sc["img_a"], lv_f, psz), O.Pyramid(sc["img_b"], lv_f, psz)
        self.otr = O.Tracker(self.oop, sc["fc"], sc["cc"], sc["wh"])
        self.op = ic.optparam(lv_f, lv_l, psz, maxiter, ratio, donorm, dpn, maxpt)
        self.cam = ic.CamClass(lv_f + 1, sc["fc"], sc["cc"], sc["wh"], psz)
        self.pose = ic.PoseClass(self.cam, self.op)
        self.odo = ic.OdometerClass(self.pose, self.op)
        self.odo.set_variant(variant | FORCE_VARIANT)
        if FORCE_TEAM is not None:
            self.odo.set_team(*FORCE_TEAM)
        self.odo.enable_trace()
        self.gpa, self.gpb = ic.Pyramid(sc["img_a"], lv_f, psz), ic.Pyramid(sc["img_b"], lv_f, psz)
        self.M = self.op.maxpttrack
        self.n = min(n, self.M)

    def set_points(self, pts=None):
        pts = self.sc["pts3d"] if pts is None else pts
        a, b = np.ascontiguousarray(pts.copy()), np.ascontiguousarray(pts.copy())
        self.otr.set3dpoints(a)
        self.odo.Set3Dpoints(b)
        return a, b

    def set_pose(self, p=None, swap=False):
        p = self.sc["p_a"] if p is None else p
        if swap:
            self.otr.setpose(p, self.opb, self.opa)
            self.odo.SetPose(p, self.gpb, self.gpa)
        else:
            self.otr.setpose(p, self.opa, self.opb)
            self.odo.SetPose(p, self.gpa, self.gpb)

    def track(self):
        return self.otr.trackpose(), self.odo.TrackPose()


def rel(a, b):
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max()
    s = max(np.abs(a).max(), np.abs(b).max(), 1e-30)
    return d / s


def scene(w, h, n, seed, **kw):
    return synth.make_scene(w, h, n_points=n, seed=seed, **kw)
