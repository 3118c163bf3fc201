"""One dense 1080p frame pair per tracking, repeated (for rocprofv3 --kernel-trace --stats: what a single tracking's
0.62 ms consists of)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import invcompcamtrack_amd as ic
from invcompcamtrack_amd import synth
sc = synth.make_scene(1920, 1080, grid_step=8, margin=4.0, jitter=0.35, seed=100)
n = sc["pts3d"].shape[1]
op = ic.optparam(2, 0, 8, 10, 0.0, 0, 0, n)
cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
pa, pb = ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8)
e = ic.TrackBatch(cam, op, 1)
e.Set3Dpoints(0, sc["pts3d"].copy())
ts = []
for r in range(60):
    t0 = time.perf_counter()
    e.SetPose(0, sc["p_a"], pa, pb)
    e.track_async()
    p = e.poses()
    ts.append(time.perf_counter() - t0)
print(e.path_name(), "median ms", round(float(np.median(ts[10:])) * 1e3, 4))
