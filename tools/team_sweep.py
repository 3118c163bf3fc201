"""Sweep of the one-launch tracker's team form (ictr_batch_set_team): points per workgroup x problem size x batch
size, against one workgroup per problem and the per-iteration launches. 640x480, 5 levels x 10 iterations, 8x8."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import invcompcamtrack_amd as ic  # noqa: E402
from invcompcamtrack_amd import synth  # noqa: E402


def run(sc, cam, op, pa, pb, B, variant, team, reps):
    eng = ic.TrackBatch(cam, op, B)
    eng.set_variant(variant)
    if team is not None:
        eng.set_team(*team)
    for k in range(B):
        eng.Set3Dpoints(k, sc["pts3d"].copy())
    ts = []
    for r in range(reps + 3):
        t0 = time.perf_counter()
        if B > 1:
            eng.SetPoseAll(np.tile(sc["p_a"], (B, 1)), pa, pb)
        else:
            eng.SetPose(0, sc["p_a"], pa, pb)
        eng.track_async()
        p = eng.poses()
        ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts[3:]) * 1e3), 4), eng.last_team(), p


def main():
    targets = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [32, 48, 64, 96, 128, 160]
    cases = [(1, n) for n in (130, 200, 300, 500, 1000, 2000, 4000, 6000)] + [(64, 300), (64, 1000), (256, 300), (16, 300)]
    for B, n in cases:
        sc = synth.make_scene(640, 480, n_points=n, seed=5)
        op = ic.optparam(4, 0, 8, 10, 0.0, 0, 0, n)
        cam = ic.CamClass(5, sc["fc"], sc["cc"], sc["wh"], 8)
        pa, pb = ic.Pyramid(sc["img_a"], 4, 8), ic.Pyramid(sc["img_b"], 4, 8)
        reps = 20 if B * n < 20000 else 8
        out = dict(problems=B, points=n)
        out["launches_ms"], _, pref = run(sc, cam, op, pa, pb, B, 8192, None, reps)
        if n <= 1000:
            out["one_workgroup_ms"], _, _ = run(sc, cam, op, pa, pb, B, 16384 | (1 << 19), None, reps)
        for t in targets:
            if (n + t - 1) // t < 2 or B * ((n + t - 1) // t) > 2048:
                continue
            ms, team, p = run(sc, cam, op, pa, pb, B, 0, (t, 0, 1 << 30), reps)
            out[f"team_target{t}_ms"] = ms
            out[f"team_target{t}_wgs"] = team
            out[f"team_target{t}_posediff"] = float(np.abs(p - pref).max())
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
