"""CPU-side tests of the process boundary: file formats of the two CLI drivers and the C++ facade header."""
import os
import subprocess

import numpy as np
import pytest

from invcompcamtrack_amd import io_formats as iof

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pointcam_file_layout_and_roundtrip(tmp_path):
    """run_io_reprojection_test.cpp:54-79 / run_odometer_test.m:131-138: f64 pose, f32 fc cc, u32 wh, u64 N, f64 XYZ, f32 xy."""
    rng = np.random.default_rng(0)
    n = 37
    pose, fc, cc, wh = rng.normal(size=6), [1000.0, 1200.0], [660.0, 390.0], [1280, 720]
    X, x = rng.normal(size=(3, n)), rng.normal(size=(2, n)).astype(np.float32)
    fn = str(tmp_path / "myFile.txt")
    iof.write_pointcam_file(fn, pose, fc, cc, wh, X, x)
    raw = open(fn, "rb").read()
    assert len(raw) == 48 + 8 + 8 + 8 + 8 + 24 * n + 8 * n
    assert np.array_equal(np.frombuffer(raw[:48], "<f8"), pose)
    assert np.frombuffer(raw[64:72], "<u4").tolist() == wh and int(np.frombuffer(raw[72:80], "<u8")[0]) == n
    assert np.array_equal(np.frombuffer(raw[80:80 + 8 * n], "<f8"), X[0])  # all X, then all Y, then all Z
    d = iof.read_pointcam_file(fn)
    assert np.array_equal(d["pose"], pose) and np.array_equal(d["pts3d"], X) and np.array_equal(d["pts2d"], x)
    assert d["fc"].dtype == np.float32 and d["wh"].tolist() == wh
    out = str(tmp_path / "out.bin")
    iof.write_pose_result(out, pose)
    assert os.path.getsize(out) == 48 and np.array_equal(iof.read_pose_result(out), pose)


def test_nposes_text_formats(tmp_path):
    """run_track_nposes.cpp:39-131."""
    rng = np.random.default_rng(1)
    op = dict(lv_f=4, lv_l=0, psz=8, maxiter=10, normdp_ratio=0.01, donorm=1, dopatchnorm=0, maxpttrack=100, verbosity=0)
    pt2d, pt3d = rng.uniform(0, 500, (12, 2)), rng.normal(size=(12, 3))
    poses = rng.normal(size=(3, 6))
    inl = [np.array([1, 2, 3, 4]), np.array([5, 6, 7, 8, 9]), np.array([12, 1])]
    files = ["f%02d.pgm" % i for i in range(5)]
    fn = str(tmp_path / "in.txt")
    iof.write_nposes_input(fn, op, [1000, 1200], [660, 390], [1280, 720], (2, 2), files, pt2d, pt3d, poses, inl)
    d = iof.read_nposes_input(fn)
    assert d["op"] == op and d["fbframes"] == (2, 2) and d["filenames"] == files
    assert np.array_equal(d["pt3d"], pt3d) and np.array_equal(d["poses"], poses)
    assert all(np.array_equal(a, b) for a, b in zip(d["inlids"], inl))
    first = open(fn).readline().split()
    assert first == ["4", "0", "8", "10", "0.01", "1", "0", "100", "0"]  # the reference's first-line order
    out = str(tmp_path / "out.txt")
    corr = [np.array([0.98765, -1.0, 0.0]), np.array([0.5])]
    pose_out = [rng.normal(size=(5, 6)), rng.normal(size=(5, 6))]
    iof.write_nposes_result(out, corr, pose_out)
    lines = open(out).read().split("\n")
    assert len(lines) == 2 * 6 + 1 and lines[5] == "0.988 -1 0 "  # setprecision(3), trailing blank
    assert lines[0].split()[0] == "%.8g" % pose_out[0][0, 0]
    c2, p2 = iof.read_nposes_result(out, 5)
    assert np.allclose(p2[1], pose_out[1], rtol=1e-7) and np.allclose(c2[0], corr[0], atol=1e-3)


def test_image_readers(tmp_path):
    img = (np.arange(12 * 16).reshape(12, 16) % 251).astype(np.uint8)
    fn = str(tmp_path / "a.pgm")
    with open(fn, "wb") as f:
        f.write(b"P5\n# comment\n16 12\n255\n" + img.tobytes())
    assert np.array_equal(iof.read_image_gray(fn), img.astype(np.float32))
    np.save(str(tmp_path / "a.npy"), img.astype(np.float32) + 0.25)
    assert np.array_equal(iof.read_image_gray(str(tmp_path / "a.npy")), img.astype(np.float32) + 0.25)
    try:
        from PIL import Image
    except ImportError:
        return
    Image.fromarray(img).save(str(tmp_path / "a.png"))
    assert np.array_equal(iof.read_image_gray(str(tmp_path / "a.png")), img.astype(np.float32))


def test_cxx_facade_compiles_against_the_c_abi(tmp_path):
    """include/ctr_shim.hpp (namespace CTR on top of include/ictr.h) + the test driver build with plain g++ -std=c++11
    and link against libictr_hip.so."""
    import __graft_entry__ as g
    g.build()
    exe = os.path.join(ROOT, "tests", "cxx", "shim_driver")
    cmd = ["g++", "-std=c++11", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), "-o", exe,
           os.path.join(ROOT, "tests", "cxx", "shim_driver.cpp"), "-L" + os.path.join(ROOT, "invcompcamtrack_amd"),
           "-l:libictr_hip.so", "-Wl,-rpath,$ORIGIN/../../invcompcamtrack_amd"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the native run_track_nposes caller (run_track_nposes.cpp:133-361 on the facade, util_getPatch / util_patchNCC included)
    exe2 = os.path.join(ROOT, "tests", "cxx", "nposes_driver")
    cmd2 = ["g++", "-std=c++11", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-o", exe2,
            os.path.join(ROOT, "tests", "cxx", "nposes_driver.cpp"), "-L" + os.path.join(ROOT, "invcompcamtrack_amd"),
            "-l:libictr_hip.so", "-Wl,-rpath,$ORIGIN/../../invcompcamtrack_amd"]
    r = subprocess.run(cmd2, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # a plain C translation unit can include the ABI header too
    c = tmp_path / "abi.c"
    c.write_text('#include "ictr.h"\nint main(void){ ictr_optparam op; return ictr_optparam_init(&op,1,0,8,5,0.1f,0,0,10,0) + (int)sizeof(op) - 44; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-I" + os.path.join(ROOT, "include"), "-o", str(tmp_path / "abi"), str(c),
                        "-L" + os.path.join(ROOT, "invcompcamtrack_amd"), "-l:libictr_hip.so",
                        "-Wl,-rpath," + os.path.join(ROOT, "invcompcamtrack_amd")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert subprocess.run([str(tmp_path / "abi")]).returncode == 0


def _worker_nposes(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch.distributed as dist
    from invcompcamtrack_amd import run_track_nposes as drv
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    seen = []

    def fake_local(inp, images, sample_ids):     # stands in for the GPU work: results that identify sample and rank
        ids = list(sample_ids)
        seen.extend(ids)
        return {sid: (np.full(3, sid + 0.5), np.full((2, 6), 10.0 * sid + rank)) for sid in ids}

    drv._run_local = fake_local
    inp = {"poses": [np.zeros(6)] * 7}
    corr, pose = drv.run(inp, dist=dist)
    q.put((rank, seen, corr, pose))
    dist.barrier()
    dist.destroy_process_group()


def test_run_track_nposes_sample_sharding_two_ranks_gloo():
    """`run_track_nposes --gpus N`: contiguous sample ranges per rank, no collective in the data path, rank 0 gathers
    and merges in sample order (run_track_nposes.cpp:193: the samples are independent)."""
    import socket
    import torch.multiprocessing as mp
    from invcompcamtrack_amd import run_track_nposes as drv
    assert drv.partition_samples(7, 2) == [(0, 4), (4, 7)] and drv.partition_samples(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    with pytest.raises(ValueError):
        drv.merge_results([{0: (1, 2)}, {0: (1, 2)}], 1)
    with pytest.raises(ValueError):
        drv.merge_results([{0: (1, 2)}], 2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_nposes, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 1, 2, 3] and res[1][1] == [4, 5, 6]
    assert res[1][2] is None and res[1][3] is None                      # only rank 0 holds the merged result
    corr, pose = res[0][2], res[0][3]
    assert [c[0] for c in corr] == [sid + 0.5 for sid in range(7)]      # sample order
    assert [p_[0, 0] for p_ in pose] == [0, 10, 20, 30, 41, 51, 61]     # samples 4-6 came from rank 1
