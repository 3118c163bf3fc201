"""Generates tests/golden/classoftrack_golden.npz by running the REFERENCE's misc_src/classoftrack.py.

Run in the build container only (needs /root/reference): python tests/golden/make_classoftrack_golden.py
The reference module is Python-2 era: it needs two environment aliases under Python 3 / NumPy 2
(np.NaN -> np.nan, xrange -> range); nothing in the module is edited or copied. The .npz holds data only:
seeded inputs (flows, corners, query points) and the reference's outputs.
"""
import builtins
import importlib.util
import os
import sys

import numpy as np

REF = "/root/reference/misc_src/classoftrack.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "classoftrack_golden.npz")


def load_reference():
    if not hasattr(np, "NaN"):
        np.NaN = np.nan
    if not hasattr(builtins, "xrange"):
        builtins.xrange = range
    spec = importlib.util.spec_from_file_location("ref_classoftrack", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def smooth_flow(rng, h, w, amp, noise):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    u = amp * np.sin(2 * np.pi * (xx / w + rng.uniform())) * np.cos(2 * np.pi * (yy / h + rng.uniform()))
    v = amp * np.cos(2 * np.pi * (xx / w + rng.uniform())) * np.sin(2 * np.pi * (yy / h + rng.uniform()))
    f = np.stack([u, v], 2) + rng.normal(0, noise, (h, w, 2))
    return f.astype(np.float32)


def main():
    ref = load_reference()
    rng = np.random.default_rng(0)
    H, W = 48, 64
    out = {}

    # --- func_get_transf_position: interior, border, outside, exact-integer and NaN points
    du = smooth_flow(rng, H, W, 2.0, 0.05)[:, :, 0].astype(np.float64)
    dv = smooth_flow(rng, H, W, 2.0, 0.05)[:, :, 1].astype(np.float64)
    pts = np.concatenate([
        rng.uniform([0, 0], [W - 1, H - 1], (160, 2)),
        np.array([[0.0, 0.0], [W - 1.0, H - 1.0], [W - 2.0, H - 2.0], [W - 1.5, 3.0], [3.0, H - 1.5],
                  [-0.5, 4.0], [4.0, -0.25], [W + 3.0, 5.0], [5.0, H + 2.0], [10.0, 10.0], [10.5, 20.0],
                  [np.nan, 3.0], [7.25, np.nan]]),
        rng.uniform([-3, -3], [W + 3, H + 3], (27, 2))])
    with np.errstate(invalid="ignore"):
        out["gtp_xy"] = pts
        out["gtp_du"] = du
        out["gtp_dv"] = dv
        out["gtp_res_uv"] = ref.func_get_transf_position(pts, du, dv)
        out["gtp_res_u"] = ref.func_get_transf_position(pts, du)

    # --- oftrack: bsize 4, 6 frames; corners on every frame except frame 5 (index 4), which stays inside the
    # window: the reference raises once a corners=None block leaves the window (classoftrack.py:97-98)
    nfr, bsize = 6, 4
    flows_f, flows_b, corners = [], [], []
    for k in range(nfr):
        ff = smooth_flow(rng, H, W, 1.5, 0.02)
        # backward flow ~ -forward (consistent) plus a band where it is inconsistent -> FB check rejects
        fb = -ff + rng.normal(0, 0.01, ff.shape).astype(np.float32)
        fb[10:16, 20:30, :] += 3.0
        flows_f.append(ff)
        flows_b.append(fb)
        corners.append(rng.uniform([2, 2], [W - 3, H - 3], (40, 2)).astype(np.float32) if k != 4 else None)
    tr = ref.oftrack(bsize, W, H, th_flowvalid_ratio=.2, th_flowvalid_abs=1)
    snaps = {}
    with np.errstate(invalid="ignore", divide="ignore"):
        for k in range(nfr):
            tr.addframe(flows_f[k], flows_b[k], corners[k])
            snaps[k] = dict(pt=tr.getpttransfer(), ptm=tr.getpttransfer(th_min_movement=0.5))
    out["of_forw"] = np.stack(flows_f)
    out["of_back"] = np.stack(flows_b)
    out["of_bsize"] = np.array(bsize)
    for k in range(nfr):
        out[f"corners_{k}"] = corners[k] if corners[k] is not None else np.zeros((0, 2), np.float32)
        out[f"has_corners_{k}"] = np.array(corners[k] is not None)
        out[f"pttransfer_{k}"] = snaps[k]["pt"]
        out[f"pttransfer_min_{k}"] = snaps[k]["ptm"]
    out["frcounter"] = np.array(tr.frcounter)
    for i in range(len(tr.tracks)):
        out[f"tracks_{i}"] = tr.tracks[i] if tr.tracks[i] is not None else np.zeros((0, 2, bsize), np.float32)
        out[f"tracks_none_{i}"] = np.array(tr.tracks[i] is None)
        if tr.tracks[i] is not None:
            out[f"tracks_valid_{i}"] = tr.tracks_valid[i]
            out[f"tracks_absmovement_{i}"] = tr.tracks_absmovement[i]
    out["ntracks"] = np.array(len(tr.tracks))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    sys.exit(main())
