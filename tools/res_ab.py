"""A/B of the resident-iteration form on the headline workload: ms per step of B dense 1080p pairs (one engine, one
stream), for the environment the caller set (ICTR_RESIDENT_* variables are read once per process).
    python tools/res_ab.py [B] [steps]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import invcompcamtrack_amd as ic
from invcompcamtrack_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sc = synth.make_scene(1920, 1080, n_points=32400, seed=11)
pa, pb = ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8)
cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
op = ic.optparam(2, 0, 8, 10, 0.0, 0, 0, 32400)
e = ic.TrackBatch(cam, op, B)
for k in range(B):
    e.Set3Dpoints(k, sc["pts3d"].copy())
P = np.tile(sc["p_a"], (B, 1))
ts = []
for r in range(steps + 3):
    t0 = time.perf_counter()
    e.SetPoseAll(P, pa, pb)
    e.track_async()
    p = e.poses()
    ts.append(time.perf_counter() - t0)
import os
env = {k: v for k, v in os.environ.items() if k.startswith("ICTR_")}
print(f"{e.path_name()[:40]} B={B} env={env}: median {np.median(ts[3:])*1e3:.3f} ms, min {min(ts[3:])*1e3:.3f} ms; pose[0] {p[0][:3]}")
