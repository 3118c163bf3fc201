"""The Python call surface the misc_src/run_*OF* drivers use: classoftrack (pinned by goldens produced by the
reference module itself, tests/golden/make_classoftrack_golden.py) and func_OF_util (no reference outputs exist
for it -- the module is Python-2 only and ships no expected values -- so it is pinned by formulas and by the
tracker's own bilinear convention: "parity unpinned by reference outputs")."""
import os

import numpy as np
import pytest

from invcompcamtrack_amd import classoftrack as ct
from invcompcamtrack_amd import func_OF_util as fu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "classoftrack_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD, allow_pickle=False)


def test_func_get_transf_position_matches_reference(gold):
    res = ct.func_get_transf_position(gold["gtp_xy"], gold["gtp_du"], gold["gtp_dv"])
    assert np.array_equal(res, gold["gtp_res_uv"], equal_nan=True)
    res_u = ct.func_get_transf_position(gold["gtp_xy"], gold["gtp_du"])
    assert np.array_equal(res_u, gold["gtp_res_u"], equal_nan=True)
    # the golden set covers every class of point
    nan_rows = np.isnan(gold["gtp_res_uv"]).any(1)
    assert 10 < nan_rows.sum() < len(nan_rows) - 100


def test_func_get_transf_position_edge_cases():
    H, W = 6, 8
    u, v = np.full((H, W), 0.5), np.full((H, W), -0.25)
    out = ct.func_get_transf_position(np.array([[1.0, 1.0], [W - 1.0, 2.0], [W - 1.001, H - 1.001], [-1e-9, 3.0]]), u, v)
    assert np.allclose(out[0], [1.5, 0.75])
    assert np.isnan(out[1]).all()          # ceil tap = W is outside, even with zero weight
    assert np.allclose(out[2], [W - 0.501, H - 1.251])
    assert np.isnan(out[3]).all()
    assert ct.func_get_transf_position(np.zeros((0, 2)), u, v).shape == (0, 2)  # empty input


def test_oftrack_matches_reference_frame_by_frame(gold):
    bsize = int(gold["of_bsize"])
    tr = ct.oftrack(bsize, 64, 48, th_flowvalid_ratio=.2, th_flowvalid_abs=1)
    nfr = gold["of_forw"].shape[0]
    for k in range(nfr):
        corners = gold[f"corners_{k}"] if bool(gold[f"has_corners_{k}"]) else None
        tr.addframe(gold["of_forw"][k], gold["of_back"][k], corners)
        assert np.array_equal(tr.getpttransfer(), gold[f"pttransfer_{k}"], equal_nan=True)
        assert np.array_equal(tr.getpttransfer(th_min_movement=0.5), gold[f"pttransfer_min_{k}"], equal_nan=True)
    assert tr.frcounter == int(gold["frcounter"]) and len(tr.tracks) == int(gold["ntracks"])
    some_invalid = False
    for i in range(len(tr.tracks)):
        if bool(gold[f"tracks_none_{i}"]):
            assert tr.tracks[i] is None and tr.tracks_valid[i] is None
            continue
        assert tr.tracks[i].dtype == np.float32
        assert np.array_equal(tr.tracks[i], gold[f"tracks_{i}"], equal_nan=True)
        assert np.array_equal(tr.tracks_valid[i], gold[f"tracks_valid_{i}"])
        assert np.array_equal(tr.tracks_absmovement[i], gold[f"tracks_absmovement_{i}"], equal_nan=True)
        some_invalid |= bool((~tr.tracks_valid[i]).any()) or tr.tracks[i].shape[0] < 40
    assert some_invalid  # the forward-backward check did reject tracks in this fixture


def test_oftrack_none_block_leaving_window_does_not_raise():
    """The reference raises TypeError once a corners=None frame leaves the window (classoftrack.py:97-98);
    this restatement skips such blocks. Documented deviation."""
    rng = np.random.default_rng(0)
    tr = ct.oftrack(2, 16, 16)
    z = np.zeros((16, 16, 2), np.float32)
    tr.addframe(z + 0.25, z - 0.25, None)
    tr.addframe(z + 0.25, z - 0.25, rng.uniform(2, 12, (5, 2)))
    tr.addframe(z + 0.25, z - 0.25, None)
    tr.addframe(z + 0.25, z - 0.25, None)
    assert tr.frcounter == 4 and tr.tracks[0] is None


def test_oftrack_savetofile_roundtrip(tmp_path):
    tr = ct.oftrack(3, 16, 16)
    z = np.zeros((16, 16, 2), np.float32)
    tr.addframe(z + 0.5, z - 0.5, np.array([[4.0, 4.0], [8.0, 9.0]]))
    tr.addframe(z + 0.5, z - 0.5, None)
    fn = str(tmp_path / "tracks.npz")
    tr.savetofile(fn)
    x = np.load(fn, allow_pickle=True)["x"]  # our own file
    assert len(x) == 2 and x[1] is None and np.allclose(x[0][:, :, 1], [[4.5, 4.5], [8.5, 9.5]])


def test_flo_roundtrip_and_bad_header(tmp_path, capsys):
    rng = np.random.default_rng(1)
    flow = rng.normal(size=(7, 11, 2)).astype(np.float32)
    fn = str(tmp_path / "a.flo")
    fu.func_write_flo_file(fn, flow)
    raw = open(fn, "rb").read()
    assert raw[:4] == np.array([202021.25], "<f4").tobytes() and np.frombuffer(raw[4:12], "<i4").tolist() == [11, 7]
    assert np.array_equal(fu.func_read_flo_file(fn), flow)
    open(fn, "wb").write(b"\x00" * 32)
    assert fu.func_read_flo_file(fn) == []
    assert "Invalid .flo" in capsys.readouterr().out


def test_pfm_reader(tmp_path, capsys):
    data = np.arange(12, dtype=np.float32).reshape(3, 4)
    fn = str(tmp_path / "a.pfm")
    with open(fn, "wb") as f:
        f.write(b"Pf\n4 3\n-1.0\n")
        data[::-1].tofile(f)  # bottom-up rows
    assert np.array_equal(fu.func_read_pfm_file(fn), data)
    open(fn, "wb").write(b"PF\n")
    assert fu.func_read_pfm_file(fn) == []
    capsys.readouterr()


def test_eval_flowgt_bins():
    gt = np.zeros((2, 3, 2))
    gt[0, 0] = [3, 4]      # mag 5   -> <10
    gt[0, 1] = [12, 16]    # mag 20  -> [10,40)
    gt[0, 2] = [30, 40]    # mag 50  -> >=40
    est = gt.copy()
    est[0, 0, 0] += 1.0
    est[0, 1, 1] += 2.0
    est[0, 2, 0] -= 3.0
    r = fu.func_eval_flowgt(gt, est)
    assert np.allclose(r, [6.0 / 6, 1.0 / 4, 2.0, 3.0])


def test_extract_bil_patch_agrees_with_tracker_convention(oracle):
    """func_OF_util.py:87-129 and util_getPatch (utilities.cpp:55-113) share one bilinear convention for even pz."""
    rng = np.random.default_rng(2)
    P = 8
    img = rng.uniform(0, 255, (40, 48)).astype(np.float32)
    op = oracle.make_op(0, 0, P, 1, 0, 0, 0, 4)
    plane = np.pad(img, P, mode="edge")
    for pt in ([20.3, 12.7], [17.0, 21.5], [25.999, 9.001]):
        a = fu.func_extract_bil_patch(np.array(pt), img[:, :, None].astype(np.float64), P)
        b = oracle.getpatch(plane, pt, op)
        assert a.shape == (P * P, 1)
        assert np.allclose(a[:, 0], b, rtol=1e-5, atol=1e-3)
    # options
    a = fu.func_extract_bil_patch(np.array([20.3, 12.7]), img[:, :, None].astype(np.float64), P, do_zeromean=1,
                                  do_unitnorm=1, do_flatten=0)
    assert a.shape == (P, P, 1) and abs(a.mean()) < 1e-12 and np.isclose(np.linalg.norm(a), 1.0)
    m = fu.func_get_pat_cosmask(P)
    a = fu.func_extract_bil_patch(np.array([20.3, 12.7]), np.ones((40, 48, 2)), P, use_mask=m, do_log=1)
    assert np.allclose(a, 0)  # log(1) = 0


def test_extract_nn_patch_and_masks():
    img = np.arange(20 * 30 * 1, dtype=float).reshape(20, 30, 1)
    a = fu.func_extract_NN_patch([10, 8], img, 4, do_flatten=0)
    assert np.array_equal(a[:, :, 0], img[6:10, 8:12, 0])
    before = img.copy()
    fu.func_extract_NN_patch([10, 8], img, 4, do_zeromean=1)
    assert np.array_equal(img, before)  # no write-through into the caller's image
    m = fu.func_get_pat_cosmask(8)
    assert m.shape == (8, 8) and np.allclose(m, m.T) and np.allclose(m, m[::-1, ::-1])
    assert np.isclose(m[3, 3], np.cos(np.sqrt(0.5) / 4 * np.pi / 2)) and np.isclose(m[0, 0], 0.0, atol=1e-12)
    g = fu.gauss2Dfilter((5, 5), 1.0)
    assert np.isclose(g.sum(), 1.0) and g[2, 2] == g.max() and np.allclose(g, g.T)
    g3 = fu.gauss2Dfilter()
    assert g3.shape == (3, 3) and np.isclose(g3[1, 1] / g3[0, 1], np.exp(1 / (2 * 0.25)))


def test_extract_patch_python2_semantics_and_options():
    """func_OF_util.py:87-165 restated for Python 3. Pinned by properties of the reference's text (the module is
    Python-2 only and ships no expected values: parity unpinned by reference outputs):
    integer `/` (odd pz -> (pz-1)^2 samples), integer points reproduce the image, linearity in the image, the option
    order log -> zero-mean -> mask -> unit-norm, the 1e-15 norm floor, flatten = column per channel, NN = plain slice."""
    import itertools
    rng = np.random.default_rng(0)
    img = rng.uniform(1, 200, (40, 48, 3))
    for pz in (4, 5, 8, 9):   # odd sizes: Python-2 `pz/2` floors, the patch is (pz-1) x (pz-1)
        side = 2 * (pz // 2)
        a = fu.func_extract_bil_patch(np.array([20.25, 15.5]), img, pz, do_flatten=0)
        assert a.shape == (side, side, 3)
        f = fu.func_extract_bil_patch(np.array([20.25, 15.5]), img, pz)
        assert f.shape == (side * side, 3) and np.array_equal(f[:, 1], a[:, :, 1].ravel())
        # integer point: weights (0,0,0,1) -> exactly the window starting pz//2 below-left of the point
        b = fu.func_extract_bil_patch(np.array([20.0, 15.0]), img, pz, do_flatten=0)
        assert np.array_equal(b, img[15 - pz // 2:15 + pz // 2, 20 - pz // 2:20 + pz // 2, :])
        n = fu.func_extract_NN_patch([20, 15], img, pz, do_flatten=0)
        assert np.array_equal(n, b)
    # bilinear: exact on an affine image, at any sub-pixel point
    yy, xx = np.mgrid[0:40, 0:48].astype(float)
    ramp = (3.0 * xx - 2.0 * yy + 7.0)[:, :, None]
    for pt in ([20.3, 12.7], [10.999, 30.001], [25.5, 25.5]):
        a = fu.func_extract_bil_patch(np.array(pt), ramp, 6, do_flatten=0)[:, :, 0]
        ys, xs = np.mgrid[0:6, 0:6].astype(float)
        want = 3.0 * (pt[0] - 3 + xs) - 2.0 * (pt[1] - 3 + ys) + 7.0
        assert np.allclose(a, want, atol=1e-10)
    # linearity in the image
    i2 = rng.uniform(1, 200, (40, 48, 3))
    pa, pb = (fu.func_extract_bil_patch(np.array([17.2, 9.9]), im, 8) for im in (img, i2))
    pab = fu.func_extract_bil_patch(np.array([17.2, 9.9]), 2.0 * img - 0.5 * i2, 8)
    assert np.allclose(pab, 2.0 * pa - 0.5 * pb, atol=1e-9)
    # option order: log first (clamped to [0.1, 255]), then zero-mean, mask, unit norm
    m = fu.func_get_pat_cosmask(8)
    big = np.full((40, 48, 1), 1000.0)
    big[10:14] = 0.0
    for zm, um, lg, un in itertools.product((0, 1), (None, m), (0, 1), (0, 1)):
        got = fu.func_extract_bil_patch(np.array([20.4, 12.6]), big, 8, do_zeromean=zm, use_mask=um, do_log=lg,
                                        do_unitnorm=un, do_flatten=0)[:, :, 0]
        want = fu.func_extract_bil_patch(np.array([20.4, 12.6]), big, 8, do_flatten=0)[:, :, 0].copy()
        if lg:
            want = np.log(np.minimum(255, np.maximum(0.1, want)))
        if zm:
            want = want - want.mean()
        if um is not None:
            want = want * um
        if un:
            want = want / max(np.linalg.norm(want), 1e-15)
        assert np.allclose(got, want, atol=1e-12), (zm, um is not None, lg, un)
    # the norm floor: an all-zero patch stays finite (0 / 1e-15)
    z = fu.func_extract_bil_patch(np.array([20.4, 12.6]), np.zeros((40, 48, 2)), 8, do_unitnorm=1)
    assert np.all(z == 0) and np.all(np.isfinite(z))
    # NN patch at the image border: numpy slicing clips, like the reference's plain slice
    edge = fu.func_extract_NN_patch([46, 38], img, 8, do_flatten=0)
    assert edge.shape == (6, 6, 3) and np.array_equal(edge, img[34:42, 42:50, :])
    # the NN patch without options is a VIEW of the image (the reference slices and returns)
    v = fu.func_extract_NN_patch([20, 15], img, 4, do_flatten=0)
    assert np.shares_memory(v, img)


def test_cosmask_and_gauss_windows_properties():
    """func_OF_util.py:169-187: radial cosine window (1 at the centre ring, 0 from radius psize/2 on, 8-fold
    symmetric) and fspecial-style Gaussian (unit sum, symmetric, centre at ceil((n-1)/2), separable ratios)."""
    for ps in (4, 7, 8, 15):
        m = fu.func_get_pat_cosmask(ps)
        assert m.shape == (ps, ps) and np.allclose(m, m.T) and m.min() >= -1e-16 and m.max() <= 1.0
        c = ps // 2
        want = np.cos(min(1.0, np.sqrt((0.5 ** 2 + 0.5 ** 2) / float(c ** 2))) * np.pi / 2)
        assert np.isclose(m[c, c], want)                       # the sample next to the centre
        assert np.isclose(m[0, 0], 0.0, atol=1e-12) or ps % 2   # corners lie beyond radius psize/2
        if ps % 2 == 0:
            assert np.allclose(m, m[::-1, ::-1])
    for shape, sig in (((3, 3), 0.5), ((5, 5), 1.0), ((5, 9), 2.0), ((4, 6), 1.5), ((31, 31), 0.2)):
        g = fu.gauss2Dfilter(shape, sig)
        assert g.shape == shape and np.isclose(g.sum(), 1.0) and g.min() >= 0
        cy, cx = int(np.ceil((shape[0] - 1) / 2.0)), int(np.ceil((shape[1] - 1) / 2.0))
        assert g[cy, cx] == g.max()
        if shape[0] > cy + 1 and cx + 1 < shape[1] and g[cy, cx + 1] > 0:
            assert np.isclose(g[cy, cx + 1] / g[cy, cx], np.exp(-1.0 / (2 * sig * sig)))
            assert np.isclose(g[cy + 1, cx + 1] / g[cy, cx], np.exp(-2.0 / (2 * sig * sig)))
    g = fu.gauss2Dfilter((31, 31), 0.2)
    assert np.count_nonzero(g) < 31 * 31   # taps below eps * peak are dropped, as fspecial does
    assert np.allclose(fu.gauss2Dfilter((3, 3), 0.5), fu.gauss2Dfilter())


def test_check_pmc_rejects_over_budget_lines(tmp_path):
    """tools/check_pmc.py: the static guard against the rocprofv3 abort of round 1 (a `pmc:` line over the gfx950
    per-block slot budget: signal 6, then a 7-minute silent hang)."""
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "check_pmc.py")
    good = tmp_path / "good.txt"
    good.write_text("pmc: FETCH_SIZE TCC_HIT_sum\npmc: WRITE_SIZE TCC_MISS_sum TCC_REQ_sum\n"
                    "pmc: SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
                    "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM\n")
    bad = tmp_path / "bad.txt"
    bad.write_text("pmc: TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum FETCH_SIZE\n")   # the line that aborted the profiler
    nine = tmp_path / "nine.txt"
    nine.write_text("pmc: " + " ".join(f"SQ_C{k}" for k in range(9)) + "\n")
    assert subprocess.run([_sys.executable, tool, str(good)]).returncode == 0
    r = subprocess.run([_sys.executable, tool, str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "TCC: 6 slots needed, 4 available" in r.stdout
    assert subprocess.run([_sys.executable, tool, str(nine)], capture_output=True).returncode == 1
    # every counter file committed under profiles/ passes
    import glob
    files = sorted(glob.glob(os.path.join(root, "profiles", "pmc_*.txt")))
    assert files and subprocess.run([_sys.executable, tool] + files, capture_output=True).returncode == 0
