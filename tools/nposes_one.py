import sys, time
import numpy as np
sys.path.insert(0, ".")
import invcompcamtrack_amd as ic
from invcompcamtrack_amd import synth
sc = synth.make_scene(1280, 720, n_points=60, seed=5)
B = 500
op = ic.optparam(4, 0, 8, 10, 0.01, 0, 0, 60)
cam = ic.CamClass(5, sc["fc"], sc["cc"], sc["wh"], 8)
pa, pb = ic.Pyramid(sc["img_a"], 4, 8), ic.Pyramid(sc["img_b"], 4, 8)
e = ic.TrackBatch(cam, op, B)
for k in range(B):
    e.Set3Dpoints(k, sc["pts3d"].copy())
rng = np.random.default_rng(1)
P = np.tile(sc["p_a"], (B, 1)) + rng.normal(0, 1e-3, (B, 6))
ts = []
for r in range(40):
    t0 = time.perf_counter()
    e.SetPoseAll(P, pa, pb)
    e.track_async()
    p = e.poses()
    ts.append(time.perf_counter() - t0)
print(e.path_name(), "median ms", round(float(np.median(ts[10:])) * 1e3, 4))
