"""Host-side split of one link of the pose-sample workload (500 x 60 points): SetPoseAll / track_async / poses."""
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
P = np.tile(sc["p_a"], (B, 1)) + np.random.default_rng(1).normal(0, 1e-3, (B, 6))
t = np.zeros((60, 4))
for r in range(60):
    t0 = time.perf_counter()
    e.SetPoseAll(P, pa, pb)
    t1 = time.perf_counter()
    e.track_async()
    t2 = time.perf_counter()
    p = e.poses()
    t3 = time.perf_counter()
    t[r] = (t1 - t0, t2 - t1, t3 - t2, t3 - t0)
m = np.median(t[10:], 0) * 1e6
print("us: SetPoseAll %.1f  track_async %.1f  poses (wait + copy out) %.1f  total %.1f" % tuple(m))
