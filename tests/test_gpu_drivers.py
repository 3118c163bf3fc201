"""The reference's two CLI drivers, re-created on the HIP engine, against serial oracle restatements of the same
drivers (run_io_reprojection_test.cpp:99-334, run_track_nposes.cpp:133-454), through the real file formats."""
import os
import subprocess

import numpy as np
import pytest

import invcompcamtrack_amd as ic
from invcompcamtrack_amd import io_formats as iof
from invcompcamtrack_amd import run_io_reprojection_test as drv_io
from invcompcamtrack_amd import run_track_nposes as drv_np
from invcompcamtrack_amd import synth
from parity_util import scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cxx_driver(name):
    """tests/cxx/<name> (built by tests/test_drivers_cpu.py on the CPU run); built here when it is missing."""
    exe = os.path.join(ROOT, "tests", "cxx", name)
    if not os.path.exists(exe):
        r = subprocess.run(["g++", "-std=c++11", "-O2", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                            os.path.join(ROOT, "tests", "cxx", name + ".cpp"), "-L" + os.path.join(ROOT, "invcompcamtrack_amd"),
                            "-l:libictr_hip.so", "-Wl,-rpath,$ORIGIN/../../invcompcamtrack_amd"], capture_output=True, text=True)
        if r.returncode != 0:
            return None
    return exe


def _pgm(fn, img):
    with open(fn, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]) + img.astype(np.uint8).tobytes())


@pytest.mark.parametrize("args", ["4 0 4 5 0.01 0 0", "4 0 8 10 0.01 1 1"])  # run_odometer_test.m:140,232
def test_run_io_reprojection_test_cli_and_cxx_facade(oracle, tmp_path, args):
    sc = scene(256, 224, 120, seed=77, margin=12.0)
    ia, ib = np.round(sc["img_a"]), np.round(sc["img_b"])  # 8-bit images, like the files the reference reads
    fa, fb, fin, fout = (str(tmp_path / n) for n in ("a.pgm", "b.pgm", "myFile.txt", "outfile.txt"))
    _pgm(fa, ia)
    _pgm(fb, ib)
    iof.write_pointcam_file(fin, sc["p_a"], sc["fc"], sc["cc"], sc["wh"], sc["pts3d"])
    argv = [fa, fb, fin, fout] + args.split() + ["120", "0"]
    assert drv_io.main(argv) == 0
    got = iof.read_pose_result(fout)
    lv_f, lv_l, psz, maxiter, ratio, donorm, dpn = (float(x) if "." in x else int(x) for x in args.split())
    op = oracle.make_op(lv_f, lv_l, psz, maxiter, ratio, donorm, dpn, 120)
    tr = oracle.Tracker(op, sc["fc"], sc["cc"], sc["wh"])
    tr.set3dpoints(sc["pts3d"].copy())
    tr.setpose(sc["p_a"], oracle.Pyramid(ia, lv_f, psz), oracle.Pyramid(ib, lv_f, psz))
    want = tr.trackpose()
    assert np.abs(got - want).max() <= 1e-4 and np.abs(got - sc["p_b"]).max() < 1e-2
    # the same job through include/ctr_shim.hpp (C++ caller, system HIP runtime, no Python in the process)
    exe = _cxx_driver("shim_driver")
    if exe is None:
        pytest.skip("tests/cxx/shim_driver could not be built (no g++?)")
    ra, rb, fout2 = str(tmp_path / "a.f32"), str(tmp_path / "b.f32"), str(tmp_path / "out2.txt")
    ia.astype(np.float32).tofile(ra)
    ib.astype(np.float32).tofile(rb)
    r = subprocess.run([exe, ra, rb, "256", "224", fin, fout2] + args.split() + ["120"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(iof.read_pose_result(fout2), got)  # same library, same bits
    # verbosity == 1: the reference's 1000-run timing line
    assert drv_io.main(argv[:-1] + ["1"]) == 0


def _oracle_track_nposes(O, inp, images):
    """Serial restatement of run_track_nposes.cpp:185-361 on the oracle (one odometer reused for every sample)."""
    o = inp["op"]
    op = O.make_op(o["lv_f"], o["lv_l"], o["psz"], o["maxiter"], o["normdp_ratio"], o["donorm"], o["dopatchnorm"],
                   o["maxpttrack"])
    pyr = [O.Pyramid(im, op.lv_f, op.psz) for im in images]
    tr = O.Tracker(op, inp["fc"], inp["cc"], inp["wh"])
    nback, nfwd = inp["fbframes"]
    M, lv = op.maxpttrack, op.lv_l
    swo, sho = tr.cam_get(4, lv), tr.cam_get(5, lv)
    out_corr, out_pose = [], []
    for sid in range(len(inp["poses"])):
        ids = inp["inlids"][sid] - 1
        n = len(ids)
        poses = np.zeros((len(images), 6))
        tr.set3dpoints(np.ascontiguousarray(inp["pt3d"][ids].T))

        def reproj(p):
            tr.setpose(p, pyr[0], pyr[0])
            t = tr.pt2d(lv)
            return np.stack([t[:n], t[M:M + n]], 1)

        refe = reproj(inp["poses"][sid])
        cpos = inp["poses"][sid].copy()
        poses[nback] = cpos
        for fr in range(nfwd):
            tr.setpose(cpos, pyr[fr + nback], pyr[fr + nback + 1])
            cpos = tr.trackpose()
            poses[fr + nback + 1] = cpos
        forw = reproj(cpos)
        cpos = inp["poses"][sid].copy()
        for fr in range(nback):
            tr.setpose(cpos, pyr[nback - fr], pyr[nback - fr - 1])
            cpos = tr.trackpose()
            poses[nback - fr - 1] = cpos
        back = reproj(cpos)
        op.dopatchnorm = True  # run_track_nposes.cpp:281
        corr = np.full(n, -1.0)
        pats = [np.zeros(op.novals, np.float32)] * 3
        for i in range(n):
            val = []
            for k, (mids, frame) in enumerate(((back, 0), (refe, nback), (forw, len(images) - 1))):
                m = mids[i].astype(np.float32)
                ok = (m[0] > 0) and (m[1] > 0) and (m[0] < swo) and (m[1] < sho)
                if ok:
                    pats[k] = O.getpatch(pyr[frame].img[lv], m, op)
                val.append(ok)
            if val[1]:
                with np.errstate(invalid="ignore", divide="ignore"):
                    pats = [p / np.sqrt(np.sum(p * p, dtype=np.float32)) for p in pats]
                w0 = float(nback * nback) if val[0] else 0.0
                w1 = float(nfwd * nfwd) if val[2] else 0.0
                cbr = max(0.0, float(np.sum(pats[0] * pats[1], dtype=np.float32))) if val[0] else -1.0
                crf = max(0.0, float(np.sum(pats[1] * pats[2], dtype=np.float32))) if val[2] else -1.0
                c = (cbr * w0 + crf * w1) / (w0 + w1) if (w0 + w1) > 0 else float("nan")
                corr[i] = 0.0 if np.isnan(c) else max(0.0, c)
        out_corr.append(corr)
        out_pose.append(poses)
    return out_corr, out_pose


@pytest.mark.parametrize("dopatchnorm", [0, 1])
def test_run_track_nposes_matches_serial_oracle(oracle, tmp_path, dopatchnorm):
    step = np.array([0.012, -0.008, 0.015, 0.0015, -0.001, 0.002])
    base = np.array([0.3, -0.2, 0.5, 0.02, -0.03, 0.01])
    poses = [base + (k - 2) * step for k in range(5)]  # frames 0..4, reference frame = 2 (nBack = nFwd = 2)
    seq = synth.make_sequence(256, 224, poses, 2, 60, seed=5, margin=16.0)
    files = []
    for k, f in enumerate(seq["frames"]):
        fn = str(tmp_path / ("frame-%05d.pgm" % k))
        _pgm(fn, f)
        files.append(fn)
    rng = np.random.default_rng(3)
    samples = [poses[2] + rng.normal(0, 1.0, 6) * np.array([4e-3, 4e-3, 4e-3, 4e-4, 4e-4, 4e-4]) for _ in range(4)]
    inl = [np.sort(rng.choice(60, size=k, replace=False)) + 1 for k in (40, 25, 60, 33)]
    op = dict(lv_f=3, lv_l=0, psz=8, maxiter=10, normdp_ratio=0.01, donorm=1, dopatchnorm=dopatchnorm, maxpttrack=60,
              verbosity=0)
    fin, fout = str(tmp_path / "myFileRANSAC.txt"), str(tmp_path / "outfileRANSAC.txt")
    iof.write_nposes_input(fin, op, seq["fc"], seq["cc"], seq["wh"], (2, 2), files, seq["px_ref"], seq["pts3d"], samples, inl)
    assert drv_np.main([fin, fout]) == 0
    corr_g, pose_g = iof.read_nposes_result(fout, 5)
    inp = iof.read_nposes_input(fin)
    corr_o, pose_o = _oracle_track_nposes(oracle, inp, [iof.read_image_gray(f) for f in files])
    assert len(pose_g) == 4
    for sid in range(4):
        assert np.abs(pose_g[sid] - pose_o[sid]).max() <= 1e-4, sid
        assert np.abs(pose_g[sid][[0, 4]] - np.array([poses[0], poses[4]])).max() < 2e-2  # tracked to the end frames
        assert np.array_equal(pose_g[sid][2], np.array([float("%.8g" % v) for v in samples[sid]]))
        assert corr_g[sid].shape == (len(inl[sid]),)
        assert np.abs(corr_g[sid] - corr_o[sid]).max() <= 2e-3  # printed with 3 significant digits
        assert corr_g[sid].min() >= 0.0 and np.median(corr_g[sid]) > 0.9
    # the NATIVE caller (tests/cxx/nposes_driver.cpp: run_track_nposes.cpp:133-361 restated on include/ctr_shim.hpp with
    # the reference's control flow -- one OdometerClass for all samples, util_getPatch / NCC through the facade): the
    # same output file byte for byte, and its literal restatement of the NCC lines on util_getPatch agrees with the
    # device score
    exe = _cxx_driver("nposes_driver")
    if exe is not None:
        fout3 = str(tmp_path / "outfileRANSAC_native.txt")
        r = subprocess.run([exe, fin, fout3, "--check"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "max |NCC restated" in r.stderr
        assert open(fout3).read() == open(fout).read()
    else:
        pytest.skip("tests/cxx/nposes_driver could not be built (no g++?)")
    # --gpus 2: the samples split over two ranks (here both on the one GPU of the box), merged by rank 0: same file
    fout2 = str(tmp_path / "outfileRANSAC_2ranks.txt")
    assert drv_np.main([fin, fout2, "--gpus", "2"]) == 0
    assert open(fout2).read() == open(fout).read()


@pytest.mark.parametrize("psz", [8, 4, 12])
def test_ncc_score_on_device_matches_numpy_restatement(oracle, psz):
    """ictr_ncc_score (run_track_nposes.cpp:271-355 in one kernel): valid points against a NumPy restatement on the
    oracle's patches, and every edge of the reference's logic: reference position outside -> -1; one neighbour frame
    outside -> that term dropped; both outside -> 0 (0/0 -> NaN -> std::max(0,NaN) = 0); flat patch -> 0."""
    sc = scene(256, 224, 80, seed=9, margin=14.0)
    lv, pad = 0, psz   # the reference pads by psz (run_track_nposes.cpp:180); the tap offsets assume it
    flat = np.full_like(sc["img_a"], 77.0)
    imgs = [sc["img_a"], sc["img_b"], (0.5 * (sc["img_a"] + sc["img_b"])).astype(np.float32), flat]
    gp = [ic.Pyramid(im, 1, pad) for im in imgs]
    op = oracle.make_op(1, 0, psz, 1, 0.0, 0, 1, 80)   # dopatchnorm on, like run_track_nposes.cpp:281
    opl = [oracle.Pyramid(im, 1, pad) for im in imgs]
    rng = np.random.default_rng(4)
    K = 64
    mr = np.stack([rng.uniform(20, 236, K), rng.uniform(20, 204, K)], 1).astype(np.float32)
    mb = (mr + rng.normal(0, 0.4, (K, 2))).astype(np.float32)
    mf = (mr + rng.normal(0, 0.4, (K, 2))).astype(np.float32)
    mr[0] = (-3.0, 50.0)                    # reference outside -> -1
    mb[1] = (300.0, 50.0)                   # back outside: only the ref-forward term
    mf[2] = (50.0, 0.0)                     # forward on the border (strict test) -> dropped
    mb[3], mf[3] = (-1.0, 5.0), (5.0, 500.0)   # both neighbours outside -> 0
    got = ic.ncc_score(gp[0], gp[1], gp[2], lv, mb, mr, mf, psz, 4.0, 9.0)
    swo, sho = 256.0, 224.0
    want = np.full(K, -1.0)
    for i in range(K):
        ok = [bool((m[i, 0] > 0) and (m[i, 1] > 0) and (m[i, 0] < swo) and (m[i, 1] < sho)) for m in (mb, mr, mf)]
        if not ok[1]:
            continue
        pats = []
        for k, m in enumerate((mb, mr, mf)):
            if ok[k]:
                pt = oracle.getpatch(opl[k].img[lv], m[i], op).astype(np.float64)
                pats.append(pt / np.sqrt(np.sum(pt * pt)))
            else:
                pats.append(None)
        w0, w1 = (4.0 if ok[0] else 0.0), (9.0 if ok[2] else 0.0)
        cbr = max(0.0, float(np.sum(pats[0] * pats[1]))) if ok[0] else -1.0
        crf = max(0.0, float(np.sum(pats[1] * pats[2]))) if ok[2] else -1.0
        with np.errstate(invalid="ignore", divide="ignore"):
            c = np.float64(cbr * w0 + crf * w1) / np.float64(w0 + w1)
        want[i] = 0.0 if np.isnan(c) else max(0.0, c)
    assert want[0] == -1.0 and want[3] == 0.0 and np.sum(want > 0.5) > 40
    assert np.abs(got - want).max() <= 2e-5, np.abs(got - want).argmax()
    # frames without texture at integer positions (bilinear weights 0,0,0,1: the patch is exactly constant): zero
    # norm -> NaN -> std::max(0, NaN) = 0
    mi = np.round(mr[4:8])
    got_flat = ic.ncc_score(gp[3], gp[1], gp[3], lv, mi, mi, mi, psz, 4.0, 9.0)
    assert np.array_equal(got_flat, np.zeros(4, np.float32))


def test_verbosity_2_prints_the_reference_iteration_log(oracle, capfd):
    """odometer.cpp:416-417: printf("Sc%02i,It%02i: %g\\n", sl, it, normdp) after every iteration when verbosity == 2."""
    sc = scene(256, 224, 150, seed=12, margin=12.0)
    for variant in (0, 8192):   # one-launch tracker and per-iteration launches
        op = ic.optparam(2, 0, 8, 4, 0.0, 0, 0, 150, 2)
        cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
        pose = ic.PoseClass(cam, op)
        odo = ic.OdometerClass(pose, op)
        odo.set_variant(variant)
        odo.enable_trace()
        odo.Set3Dpoints(sc["pts3d"].copy())
        odo.SetPose(sc["p_a"], ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8))
        capfd.readouterr()
        odo.TrackPose()
        out = capfd.readouterr().out.strip().splitlines()
        tr = odo.trace()
        assert len(out) == 12 == len(tr)
        for line, r in zip(out, tr):
            nd = np.float32(0)
            d = np.abs(r["dp"]).astype(np.float32)
            nd = (d[0] + (d[1] + d[2])) + (d[3] + (d[4] + d[5]))
            assert line == "Sc%02i,It%02i: %g" % (r["level"], r["iter"], nd), (line, r["level"], r["iter"])
        assert out[0].startswith("Sc02,It00: ") and out[-1].startswith("Sc00,It03: ")
    # verbosity 0 prints nothing
    op0 = ic.optparam(2, 0, 8, 2, 0.0, 0, 0, 150, 0)
    odo0 = ic.OdometerClass(ic.PoseClass(cam, op0), op0)
    odo0.Set3Dpoints(sc["pts3d"].copy())
    odo0.SetPose(sc["p_a"], ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8))
    capfd.readouterr()
    odo0.TrackPose()
    assert capfd.readouterr().out == ""
