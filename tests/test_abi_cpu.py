"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/ictr.h declares, its host-only
entry points agree bit for bit with the oracle, and every GPU entry point fails loudly without a device
(this container has none) instead of falling back to a CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from invcompcamtrack_amd import _lib
    return _lib.load()


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ictr.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ictr_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(lib):
    from invcompcamtrack_amd import _lib
    names = _declared_symbols()
    assert len(names) > 60
    raw = C.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, missing
    assert sorted(_lib.SIGNATURES) == names  # the ctypes table and the header list the same functions


def test_no_oracle_or_torch_in_the_product_path():
    """The product may not route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "invcompcamtrack_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "ictr_oracle" not in txt and "libictr_oracle" not in txt, f


def test_optparam_layout_matches_reference_struct(lib):
    from invcompcamtrack_amd._lib import OptParam
    assert C.sizeof(OptParam) == 44  # 7 int, 2 bool (+2 pad), int, float, int
    assert OptParam.donorm.offset == 28 and OptParam.dopatchnorm.offset == 29 and OptParam.maxiter.offset == 32
    import invcompcamtrack_amd as ic
    op = ic.optparam(4, 0, 8, 10, 0.01, 1, 0, 50)
    assert (op.pszd2, op.pszd2m3, op.novals, op.maxpttrack, op.donorm, op.dopatchnorm) == (4, 11, 64, 52, True, False)


def test_host_math_is_bit_identical_to_oracle(lib, oracle):
    import invcompcamtrack_amd as ic
    rng = np.random.default_rng(0)
    for _ in range(200):
        scale = 10.0 ** rng.integers(-6, 1)
        p = (rng.normal(0, 1, 6) * np.array([2, 2, 2, scale, scale, scale]))
        for dt in (np.float32, np.float64):
            G = ic.util_SE3_coeff_to_group(p.astype(dt))
            assert np.array_equal(G, oracle.se3_exp(p.astype(dt)))
            assert np.array_equal(ic.util_SE3_group_to_coeff(G), oracle.se3_log(G), equal_nan=True)
        J = rng.normal(size=(12, 6))
        H = (J.T @ J).astype(np.float32)
        b = rng.normal(size=6).astype(np.float32)
        assert np.array_equal(ic.solve6(H, b), oracle.solve6(H, b))
    Hs = np.zeros((6, 6), np.float32)
    Hs[:3, :3] = np.eye(3)
    assert np.array_equal(ic.solve6(Hs, np.arange(6, dtype=np.float32)), oracle.solve6(Hs, np.arange(6, dtype=np.float32)))
    Hk = np.array([[9, 100, 78, 81, 14, 63], [23, 8, 82, 44, 87, 36], [92, 45, 87, 92, 58, 52],
                   [16, 11, 9, 19, 55, 41], [83, 97, 40, 27, 15, 8], [54, 1, 26, 15, 86, 24]], np.float32)
    bk = np.array([12.15, 11.12, 14.13, 6.62, 6.28, 7.68], np.float32)  # odometer.cpp:474-493
    assert np.array_equal(ic.solve6(Hk, bk), oracle.solve6(Hk, bk))


def test_cam_and_pose_host_side(lib, oracle):
    import invcompcamtrack_amd as ic
    op = ic.optparam(4, 0, 8, 10, 0.01, 1, 0, 16)
    cam = ic.CamClass(5, [1000, 1200], [660, 390], [1280, 720], 8)
    oop = oracle.make_op(4, 0, 8, 10, 0.01, 1, 0, 16)
    otr = oracle.Tracker(oop, [1000, 1200], [660, 390], [1280, 720])
    for l in range(5):
        got = [cam.getfx(l), cam.getfy(l), cam.getcx(l), cam.getcy(l), cam.getswo(l), cam.getsho(l), cam.getsw(l),
               cam.getsh(l)]
        assert got == [otr.cam_get(k, l) for k in range(8)]
    pose = ic.PoseClass(cam, op)
    p = np.array([0.4, -0.3, 1.2, 0.05, -0.02, 0.08])
    ms, var = np.array([1.0, -2.0, 15.0]), 7.5
    pose.setpose_se3(p, ms, var)
    L = oracle.lib()
    L.orc_pose_setpose_se3(otr.pose, p.ctypes.data_as(C.POINTER(C.c_double)), ms.ctypes.data_as(C.POINTER(C.c_double)),
                           var)
    pp, G = pose.state()
    assert np.array_equal(pp, otr.pose_p()) and np.array_equal(G, otr.pose_G())
    dp = np.array([1e-3, -2e-3, 5e-4, 1e-4, 2e-4, -3e-4], np.float32)
    pose.addpose_se3(dp)
    L.orc_pose_addpose_se3(otr.pose, dp.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.array_equal(pose.state()[1], otr.pose_G())
    out_o = np.zeros(6)
    L.orc_pose_getpose_se3(otr.pose, out_o.ctypes.data_as(C.POINTER(C.c_double)))
    assert np.array_equal(pose.getPose_se3(), out_o)
    pose.subpose_se3(dp)
    L.orc_pose_subpose_se3(otr.pose, dp.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.array_equal(pose.state()[0], otr.pose_p())


def test_gpu_entry_points_fail_loudly_without_a_device(lib):
    import invcompcamtrack_amd as ic
    from invcompcamtrack_amd._lib import IctrError
    if ic.device_count() > 0:
        pytest.skip("a HIP device is present")
    img = np.zeros((64, 64), np.float32)
    with pytest.raises(IctrError, match="no usable HIP device"):
        ic.Pyramid(img, 1, 8)
    op = ic.optparam(1, 0, 8, 5, 0.01, 0, 0, 16)
    cam = ic.CamClass(2, [100, 100], [32, 32], [64, 64], 8)
    pose = ic.PoseClass(cam, op)
    with pytest.raises(IctrError, match="no usable HIP device"):
        ic.OdometerClass(pose, op)
    with pytest.raises(IctrError, match="no usable HIP device"):
        ic.TrackBatch(cam, op, 4)
    with pytest.raises(IctrError, match="no usable HIP device"):
        pose.project_pt(np.zeros(48, np.float32), 4, 0)


def test_argument_validation(lib):
    import invcompcamtrack_amd as ic
    from invcompcamtrack_amd._lib import IctrError
    with pytest.raises(IctrError):
        ic.CamClass(0, [1, 1], [0, 0], [8, 8], 4)
    cam = ic.CamClass(2, [100, 100], [32, 32], [64, 64], 4)
    op = ic.optparam(3, 0, 8, 5, 0.01, 0, 0, 16)  # lv_f beyond the camera's levels, psz > padding
    pose = ic.PoseClass(cam, op)
    with pytest.raises(IctrError):
        ic.OdometerClass(pose, op)


def test_missing_library_raises(monkeypatch, tmp_path):
    from invcompcamtrack_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.IctrError, match="no CPU fallback"):
        _lib.load()


def test_locality_order_is_a_permutation_that_makes_neighbours_close():
    """tracker.locality_order (host NumPy + the library's host exp map, no GPU): a permutation of the points along the
    Z-curve of their projections, so that the 32-64 consecutive points a wave owns share cache lines."""
    import numpy as np
    from invcompcamtrack_amd import locality_order
    from invcompcamtrack_amd import synth
    sc = synth.make_scene(640, 480, n_points=4000, seed=2)
    order = locality_order(sc["pts3d"], sc["p_a"], sc["fc"], sc["cc"])
    assert sorted(order.tolist()) == list(range(4000))
    G = np.eye(4)
    from invcompcamtrack_amd import util_SE3_coeff_to_group
    G[:3] = util_SE3_coeff_to_group(np.asarray(sc["p_a"], np.float64)).reshape(3, 4)
    Xc = G[:3, :3] @ sc["pts3d"] + G[:3, 3:4]
    px = np.stack([Xc[0] / Xc[2] * sc["fc"][0] + sc["cc"][0], Xc[1] / Xc[2] * sc["fc"][1] + sc["cc"][1]], 1)
    step = lambda q: np.median(np.abs(np.diff(q, axis=0)).max(1))
    assert step(px[order]) < 0.15 * step(px)      # consecutive points: a few pixels apart instead of half a frame
