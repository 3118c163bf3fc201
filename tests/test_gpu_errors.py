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
