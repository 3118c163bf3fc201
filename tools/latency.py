"""Latency at the reference's own problem sizes (run_odometer_test.m: a few hundred points, 8x8 or 4x4 patches,
5 levels): SetPose + TrackPose + poses on the host, GPU (the default selection -- one launch per tracking, one workgroup
per problem up to 192 points and a team of workgroups above --, the one-launch tracker forced to ONE workgroup per
problem, the per-iteration launches replayed as a hipGraph, the plain per-iteration launches) vs the CPU oracle on the same inputs. VERDICT r01 item 3 bars: 100-point pair <= 0.15 ms, 64 x 300-point batch <= 0.3 ms."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import invcompcamtrack_amd as ic  # noqa: E402
from invcompcamtrack_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def one(w, h, n, psz, lv_f, maxiter, B=1, reps=40, cpu=True):
    sc = synth.make_scene(w, h, n_points=n, seed=5)
    op = ic.optparam(lv_f, 0, psz, maxiter, 0.0, 0, 0, n)
    cam = ic.CamClass(lv_f + 1, sc["fc"], sc["cc"], sc["wh"], psz)
    pa, pb = ic.Pyramid(sc["img_a"], lv_f, psz), ic.Pyramid(sc["img_b"], lv_f, psz)
    out = dict(frame=f"{w}x{h}", points=n, problems=B, psz=psz, levels=lv_f + 1, maxiter=maxiter)
    poses = {}
    forms = (("default", 0), ("one_workgroup", 16384 | (1 << 19)), ("graph", 8192),
             ("per_iteration_launches", 8192 | 32768))
    for name, variant in forms:
        if name == "one_workgroup" and n * psz * psz > 2048 * 64:
            continue
        eng = ic.TrackBatch(cam, op, B)
        eng.set_variant(variant)
        for k in range(B):
            eng.Set3Dpoints(k, sc["pts3d"].copy())
        ts = []
        for r in range(reps + 5):
            t0 = time.perf_counter()
            if B > 1:  # one call for all problems (the run_track_nposes driver does the same)
                eng.SetPoseAll(np.tile(sc["p_a"], (B, 1)), pa, pb)
            else:
                eng.SetPose(0, sc["p_a"], pa, pb)
            eng.track_async()
            p = eng.poses()
            ts.append(time.perf_counter() - t0)
        out[name + "_ms"] = round(float(np.median(ts[5:]) * 1e3), 4)
        out[name + "_path"] = eng.path_name()
        poses[name] = p
    out["pose_diff_between_forms"] = float(max(np.abs(poses[k] - poses["per_iteration_launches"]).max() for k in poses))
    if cpu:
        oop = O.make_op(lv_f, 0, psz, maxiter, 0.0, 0, 0, n)
        tr = O.Tracker(oop, sc["fc"], sc["cc"], sc["wh"])
        opa, opb = O.Pyramid(sc["img_a"], lv_f, psz), O.Pyramid(sc["img_b"], lv_f, psz)
        tc = []
        for r in range(5):
            t0 = time.perf_counter()
            tr.set3dpoints(sc["pts3d"].copy())
            tr.setpose(sc["p_a"], opa, opb)
            pc = tr.trackpose()
            tc.append(time.perf_counter() - t0)
        out["cpu_oracle_ms_per_problem"] = round(float(np.median(tc) * 1e3), 3)
        out["pose_diff_vs_cpu"] = float(np.abs(poses["default"][0] - pc).max())
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    for n in (60, 100, 300, 1000, 3000):
        one(640, 480, n, 8, 4, 10)
    one(640, 480, 300, 4, 4, 5)
    one(640, 480, 300, 8, 4, 10, B=64, reps=20)
    one(1280, 720, 60, 8, 4, 10, B=500, reps=5, cpu=False)
    one(640, 480, 5000, 8, 4, 10, reps=10)
