"""File / wire formats of the reference's two CLI drivers (the process boundary its MATLAB callers depend on).

  point/cam file  run_io_reprojection_test.cpp:54-79  (writer: run_odometer_test.m:131-138), little endian:
                  6 x f64 pose (se(3) coefficients), 2 x f32 fc, 2 x f32 cc, 2 x u32 wh, 1 x u64 N,
                  N x f64 X, N x f64 Y, N x f64 Z, N x f32 x, N x f32 y (2-D ground truth, unused by the tracker)
  pose result     run_io_reprojection_test.cpp:83-97 : 6 x f64
  nposes input    run_track_nposes.cpp:39-103 (text), nposes output run_track_nposes.cpp:106-131 (text)
  images          the reference decodes with cv::imread(GRAYSCALE) (not available here): .npy / .pgm are read natively,
                  everything else through PIL when installed. Colour -> grey conversion of a decoder is not pinned
                  by the reference; feed grey images for reproducible results.
"""
from __future__ import annotations

import io
import os
import struct

import numpy as np

__all__ = ["read_pointcam_file", "write_pointcam_file", "read_pose_result", "write_pose_result", "read_image_gray",
           "read_nposes_input", "write_nposes_input", "write_nposes_result", "read_nposes_result"]

MAXPTREAD = 10000  # run_io_reprojection_test.cpp:37


def read_pointcam_file(filename):
    with open(filename, "rb") as f:
        pose = np.frombuffer(f.read(48), "<f8").copy()
        fc = np.frombuffer(f.read(8), "<f4").copy()
        cc = np.frombuffer(f.read(8), "<f4").copy()
        wh = np.frombuffer(f.read(8), "<u4").astype(np.int32)
        (n,) = struct.unpack("<Q", f.read(8))
        if n > MAXPTREAD:
            raise ValueError(f"{filename}: {n} points, the reference reads at most {MAXPTREAD}")
        xyz = np.frombuffer(f.read(24 * n), "<f8").reshape(3, n).copy()
        rest = f.read(8 * n)
        xy = np.frombuffer(rest, "<f4").reshape(2, n).copy() if len(rest) == 8 * n else np.zeros((2, n), np.float32)
    return dict(pose=pose, fc=fc, cc=cc, wh=wh, pts3d=np.ascontiguousarray(xyz), pts2d=xy)


def write_pointcam_file(filename, pose, fc, cc, wh, pts3d, pts2d=None):
    pts3d = np.asarray(pts3d, np.float64)
    n = pts3d.shape[1]
    pts2d = np.zeros((2, n), np.float32) if pts2d is None else np.asarray(pts2d, np.float32)
    with open(filename, "wb") as f:
        f.write(np.asarray(pose, "<f8").tobytes())
        f.write(np.asarray(fc, "<f4").tobytes())
        f.write(np.asarray(cc, "<f4").tobytes())
        f.write(np.asarray(wh, "<u4").tobytes())
        f.write(struct.pack("<Q", n))
        f.write(np.ascontiguousarray(pts3d, "<f8").tobytes())
        f.write(np.ascontiguousarray(pts2d, "<f4").tobytes())


def read_pose_result(filename):
    return np.fromfile(filename, "<f8", count=6)


def write_pose_result(filename, pose):
    np.asarray(pose, "<f8").tofile(filename)


def _read_pgm(data):
    f = io.BytesIO(data)
    tokens = []
    while len(tokens) < 4:
        line = f.readline()
        if not line:
            raise ValueError("truncated PGM header")
        tokens += line.split(b"#")[0].split()
    magic, w, h, maxv = tokens[0], int(tokens[1]), int(tokens[2]), int(tokens[3])
    if magic != b"P5":
        raise ValueError("only binary PGM (P5) is supported")
    dt = np.uint8 if maxv < 256 else ">u2"
    return np.frombuffer(f.read(), dt, count=w * h).reshape(h, w).astype(np.float32)


def read_image_gray(filename):
    """Grey float32 image, values as stored (0..255 for 8-bit sources), like imread(GRAYSCALE) + convertTo(CV_32F)."""
    ext = os.path.splitext(filename)[1].lower()
    if ext == ".npy":
        a = np.load(filename, allow_pickle=False)
        return np.ascontiguousarray(a if a.ndim == 2 else a[..., :3].mean(-1), np.float32)
    if ext == ".pgm":
        return _read_pgm(open(filename, "rb").read())
    try:
        from PIL import Image
    except ImportError as exc:
        raise RuntimeError(f"cannot decode {filename}: only .npy/.pgm are built in and PIL is not installed") from exc
    return np.asarray(Image.open(filename).convert("L"), np.float32)


def read_nposes_input(filename):
    """run_track_nposes.cpp:39-103. Returns a dict; poses (S,6); inlids: list of 1-based id arrays."""
    with open(filename) as f:
        lines = [ln.strip() for ln in f if ln.strip() != ""]
    it = iter(lines)
    v = next(it).split()
    op = dict(lv_f=int(v[0]), lv_l=int(v[1]), psz=int(v[2]), maxiter=int(v[3]), normdp_ratio=float(v[4]),
              donorm=int(v[5]), dopatchnorm=int(v[6]), maxpttrack=int(v[7]), verbosity=int(v[8]))
    v = next(it).split()
    fc, cc, wh = np.array(v[0:2], np.float32), np.array(v[2:4], np.float32), np.array(v[4:6], np.int32)
    nback, nfwd = (int(x) for x in next(it).split()[:2])
    files = [next(it).split()[0] for _ in range(nback + nfwd + 1)]
    ncorr = int(next(it).split()[0])
    pt2d, pt3d = np.zeros((ncorr, 2)), np.zeros((ncorr, 3))
    for i in range(ncorr):
        v = [float(x) for x in next(it).split()[:5]]
        pt2d[i], pt3d[i] = v[0:2], v[2:5]
    nsamp = int(next(it).split()[0])
    poses, inl = np.zeros((nsamp, 6)), []
    for i in range(nsamp):
        v = next(it).split()
        poses[i] = [float(x) for x in v[:6]]
        k = int(v[6])
        inl.append(np.array([int(x) for x in v[7:7 + k]], np.int64))
    return dict(op=op, fc=fc, cc=cc, wh=wh, fbframes=(nback, nfwd), filenames=files, pt2d=pt2d, pt3d=pt3d, poses=poses,
                inlids=inl)


def write_nposes_input(filename, op, fc, cc, wh, fbframes, filenames, pt2d, pt3d, poses, inlids):
    with open(filename, "w") as f:
        f.write("%d %d %d %d %g %d %d %d %d\n" % (op["lv_f"], op["lv_l"], op["psz"], op["maxiter"], op["normdp_ratio"],
                                                 op["donorm"], op["dopatchnorm"], op["maxpttrack"], op["verbosity"]))
        f.write("%.9g %.9g %.9g %.9g %d %d\n" % (fc[0], fc[1], cc[0], cc[1], wh[0], wh[1]))
        f.write("%d %d\n" % tuple(fbframes))
        for fn in filenames:
            f.write(fn + "\n")
        f.write("%d\n" % len(pt3d))
        for a, b in zip(pt2d, pt3d):
            f.write("%.17g %.17g %.17g %.17g %.17g\n" % (a[0], a[1], b[0], b[1], b[2]))
        f.write("%d\n" % len(poses))
        for p, ids in zip(poses, inlids):
            f.write(" ".join("%.17g" % x for x in p) + " %d " % len(ids) + " ".join(str(int(i)) for i in ids) + "\n")


def _fmt(x, prec):
    """C++ ostream << double with setprecision(prec) in the default float format == printf %.{prec}g"""
    return "%.*g" % (prec, x)


def write_nposes_result(filename, out_corr, out_pose):
    """run_track_nposes.cpp:106-131: per sample noimages lines of 6 pose coefficients (precision 8, trailing
    space), then one line of per-point correlations (precision 3)."""
    with open(filename, "w") as f:
        for corr, poses in zip(out_corr, out_pose):
            for p in poses:
                f.write("".join(_fmt(v, 8) + " " for v in p) + "\n")
            f.write("".join(_fmt(v, 3) + " " for v in corr) + "\n")


def read_nposes_result(filename, noimages):
    out_corr, out_pose = [], []
    with open(filename) as f:
        lines = f.read().split("\n")
    i = 0
    while i + noimages < len(lines) and lines[i].strip() != "":
        out_pose.append(np.array([[float(x) for x in lines[i + k].split()] for k in range(noimages)]))
        out_corr.append(np.array([float(x) for x in lines[i + noimages].split()]))
        i += noimages + 1
    return out_corr, out_pose
