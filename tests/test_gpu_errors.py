"""Error behaviour of the C-ABI on a machine WITH a GPU: call-order violations and mismatched inputs come back as
status codes (Python: IctrError) with a message, never as crashes. (The reference ignores all errors)"""
import numpy as np
import pytest

import invcompcamtrack_amd as ic
from invcompcamtrack_amd import _lib

from parity_util import scene

pytestmark = pytest.mark.gpu


def _setup(psz=8, lv_f=2, n=40):
    sc = scene(160, 128, n, seed=2, margin=20.0)
    op = ic.optparam(lv_f, 0, psz, 3, 0.0, 0, 0, n)
    cam = ic.CamClass(lv_f + 1, sc["fc"], sc["cc"], sc["wh"], psz)
    return sc, op, cam


def test_call_order_and_mismatches():
    sc, op, cam = _setup()
    pose = ic.PoseClass(cam, op)
    odo = ic.OdometerClass(pose, op)
    with pytest.raises(ic.IctrError, match="SetPose"):
        odo.TrackPose()                                            # TrackPose before SetPose
    odo.Set3Dpoints(sc["pts3d"].copy())
    good = ic.Pyramid(sc["img_a"], 2, 8)
    with pytest.raises(ic.IctrError, match="pad"):
        odo.SetPose(sc["p_a"], ic.Pyramid(sc["img_a"], 2, 4), good)   # padding != camera padding
    with pytest.raises(ic.IctrError, match="levels"):
        odo.SetPose(sc["p_a"], ic.Pyramid(sc["img_a"], 1, 8), good)   # too few levels
    with pytest.raises(ic.IctrError, match="gradients"):
        odo.SetPose(sc["p_a"], ic.Pyramid(sc["img_a"], 2, 8, getgrad=False), good)
    with pytest.raises(ic.IctrError):
        odo.SetPose(sc["p_a"], ic.Pyramid(np.zeros((64, 96), np.float32), 2, 8), good)   # other frame size
    with pytest.raises(TypeError):
        odo.Set3Dpoints(sc["pts3d"].astype(np.float32))           # the reference takes double* and so do we
    odo.SetPose(sc["p_a"], good, ic.Pyramid(sc["img_b"], 2, 8))
    p = odo.TrackPose()
    assert np.all(np.isfinite(p))


def test_batch_index_and_patch_size_checks():
    sc, op, cam = _setup()
    b = ic.TrackBatch(cam, op, 2)
    pa = ic.Pyramid(sc["img_a"], 2, 8)
    with pytest.raises(ic.IctrError):
        b.SetPose(2, sc["p_a"], pa, pa)                            # problem index out of range
    with pytest.raises(ic.IctrError):
        b.Set3Dpoints(-1, sc["pts3d"].copy())
    b.Set3Dpoints(0, sc["pts3d"].copy())
    b.SetPose(0, sc["p_a"], pa, pa)
    with pytest.raises(ic.IctrError, match="SetPose"):
        b.track_async()                                            # problem 1 has no pose yet
    mids = np.array([[20.0, 20.0]], np.float32)
    with pytest.raises(ic.IctrError, match="psz"):
        ic.util_getPatch(pa, 0, mids, ic.optparam(2, 0, 16, 3, 0.0, 0, 0, 4))   # patch larger than the pyramid padding
    with pytest.raises(ic.IctrError):
        ic.CamClass(0, sc["fc"], sc["cc"], sc["wh"], 8)            # noscales must be 1..16
    with pytest.raises(ic.IctrError):
        ic.Pyramid(np.zeros((4, 4), np.float32), 5, 2)             # a level would be empty


def _dense_batch(n, B, w=640, h=384, seed=3):
    from invcompcamtrack_amd import synth
    sc = synth.make_scene(w, h, n_points=n, seed=seed)
    op = ic.optparam(2, 0, 8, 4, 0.0, 0, 0, n)
    cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
    pa, pb = ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8)
    e = ic.TrackBatch(cam, op, B)
    for k in range(B):
        e.Set3Dpoints(k, sc["pts3d"].copy())
    return sc, e, pa, pb


@pytest.mark.parametrize("form", ["team", "resident"])
def test_in_launch_exchange_timeout_is_reported_and_the_engine_recovers(form, monkeypatch):
    """The bounded polling of the two in-launch exchange forms (ictr_track1.hip team_allsum, ictr_resident.hip res_poll):
    variant bit 25 makes workgroup 0 of every problem skip its gather store, so its peers' polls run into
    ICTR_TEAM_TIMEOUT_S. The launch must END, the wait must return an error that names the form (no hang, no garbage
    poses handed out), and the next tracking on the SAME engine must work and equal a fresh engine's."""
    import time
    if form == "team":
        sc, e, pa, pb = _dense_batch(600, 2)
        e.set_team(64)                                # ten workgroups per problem
    else:
        sc, e, pa, pb = _dense_batch(9000, 2)         # >= 8193 points: the resident-iteration form
    P = np.tile(sc["p_a"], (2, 1))
    e.SetPoseAll(P, pa, pb)
    e.track_async()
    good = e.poses().copy()
    assert ("k_level_resident" in e.path_name()) == (form == "resident"), e.path_name()
    if form == "team":
        assert e.last_team() > 1

    monkeypatch.setenv("ICTR_TEAM_TIMEOUT_S", "0.05")
    e.set_variant(1 << 25)
    e.SetPoseAll(P, pa, pb)
    t0 = time.perf_counter()
    e.track_async()
    with pytest.raises(ic.IctrError, match="timed out") as ei:
        e.poses()
    dt = time.perf_counter() - t0
    assert dt < 5.0, f"the time-out path took {dt:.2f} s"
    assert ("resident" if form == "resident" else "team form") in str(ei.value)
    with pytest.raises(ic.IctrError, match="timed out"):   # the failed tracking stays failed until the next one starts
        e.poses()

    e.set_variant(0)
    e.SetPoseAll(P, pa, pb)
    e.track_async()
    again = e.poses()
    assert np.array_equal(again, good), np.abs(again - good).max()
