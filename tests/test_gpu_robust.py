"""The behaviour-changing robustness options (clean handling of invisible points,
compositional SE(3) update, Huber weights). They are OFF by default -- the default reproduces the reference with
its quirks -- and have no counterpart in the reference, so the oracle is the NumPy restatement with the same options
(oracle/np_oracle.py): parity unpinned by the reference."""
import numpy as np
import pytest

import invcompcamtrack_amd as ic

from parity_util import Pair, scene

POSE_TOL = 1e-4

pytestmark = pytest.mark.gpu


def _compose(O):
    def f(p, dp):  # p_new = log(exp(dp) * exp(p)), float32 like the device code
        D = O.se3_exp(np.asarray(dp, np.float32)).reshape(3, 4).astype(np.float32)
        G = O.se3_exp(np.asarray(p, np.float32)).reshape(3, 4).astype(np.float32)
        D4, G4 = np.eye(4, dtype=np.float32), np.eye(4, dtype=np.float32)
        D4[:3], G4[:3] = D, G
        return O.se3_log(np.ascontiguousarray((D4 @ G4)[:3].reshape(12), np.float32))
    return f


@pytest.mark.parametrize("compositional,huber_k", [(True, 0.0), (False, 6.0), (True, 4.0)])
def test_options_match_numpy_oracle(oracle, compositional, huber_k):
    from oracle import np_oracle as N
    sc = scene(256, 224, 150, seed=21, margin=40.0)
    pr = Pair(oracle, sc, 2, 0, 8, 5, 0.0, 0, 0)
    pr.odo.set_robust(compositional=compositional, huber_k=huber_k)
    pr.set_points()
    pr.set_pose()
    pg = pr.odo.TrackPose()
    p_n, trace_n = N.track(sc["pts3d"], sc["p_a"], pr.opa, pr.opb, pr.otr.cam_get, 2, 0, 8, 5, oracle.solve6,
                           compose=_compose(oracle) if compositional else None, huber_k=huber_k)
    tg = pr.odo.trace()
    assert len(tg) == len(trace_n) == 15
    assert np.abs(tg[0]["dp"] - trace_n[0]["dp"]).max() <= 2e-3 * np.abs(trace_n[0]["dp"]).max() + 1e-7
    assert np.abs(pg - p_n).max() <= POSE_TOL
    assert np.abs(pg - sc["p_b"]).max() < 2e-3                     # and it still tracks the true motion
    # the option really changes the arithmetic: the default run differs from it
    pr2 = Pair(oracle, sc, 2, 0, 8, 5, 0.0, 0, 0)
    pr2.set_points()
    pr2.set_pose()
    p_def = pr2.odo.TrackPose()
    assert np.abs(p_def - pg).max() > 0


def test_all_off_is_the_default_path_bit_for_bit(oracle):
    sc = scene(256, 224, 120, seed=22, margin=30.0)
    outs = []
    for call in (False, True):
        pr = Pair(oracle, sc, 2, 0, 8, 6, 0.0, 0, 0)
        if call:
            pr.odo.set_robust()
        pr.set_points()
        pr.set_pose()
        outs.append(pr.odo.TrackPose())
    assert np.array_equal(outs[0], outs[1])


def test_clean_invisible_forgets_stale_patches(oracle):
    """Chain of two frame pairs without Set3Dpoints (run_track_nposes.cpp:232-258). In the second pair about half of
    the points are outside the reference view. Default: their stale patches / coefficients of the first pair stay in
    H and b (odometer.cpp:304). With clean_invisible the chained engine equals a fresh engine that never saw pair 1."""
    sc = scene(640, 368, 200, seed=13)
    res = {}
    for mode in ("quirk", "clean_chained", "clean_fresh"):
        pr = Pair(oracle, sc, 3, 0, 8, 5, 0.0, 0, 0)
        if mode != "quirk":
            pr.odo.set_robust(clean_invisible=True)
        pr.set_points()
        if mode != "clean_fresh":
            pr.set_pose()
            p1 = pr.odo.TrackPose()
            res.setdefault("p1", p1)
        p2 = res["p1"].copy()
        p2[0] += 2.4
        pr.odo.SetPose(p2, pr.gpb, pr.gpa)
        res[mode] = (pr.odo.TrackPose(), pr.odo.trace()[-20]["H"].copy())   # H of the coarsest level of pair 2
    assert np.abs(res["clean_chained"][0] - res["clean_fresh"][0]).max() <= 1e-6
    assert np.allclose(res["clean_chained"][1], res["clean_fresh"][1], rtol=1e-5)
    assert np.abs(res["quirk"][1] - res["clean_fresh"][1]).max() > 1e-3 * np.abs(res["clean_fresh"][1]).max()


def test_bad_flags_are_rejected():
    from invcompcamtrack_amd import _lib
    sc = scene(128, 96, 20, seed=3)
    op = ic.optparam(1, 0, 8, 3, 0.0, 0, 0, 20)
    cam = ic.CamClass(2, sc["fc"], sc["cc"], sc["wh"], 8)
    b = ic.TrackBatch(cam, op, 1)
    L = _lib.load()
    assert L.ictr_batch_set_robust(b._h, 64, 0.0) == 1            # unknown bit
    assert L.ictr_batch_set_robust(b._h, 4, 0.0) == 1             # Huber without a threshold
    assert L.ictr_batch_set_robust(b._h, 0, 0.0) == 0
