"""One small frame pair tracked 30 times, with the host-side split (SetPose / enqueue / wait) printed -- the command to
put under `rocprofv3 --kernel-trace --memory-copy-trace --output-format csv` when the question is what the GPU does
between two trackings (r02: state upload 2 copies + 1 fill + k_project_ref = ~35 us in front of a 270 us k_track1).
Usage: python tools/trace_one.py [points]"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import invcompcamtrack_amd as ic
from invcompcamtrack_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sc = synth.make_scene(640, 480, n_points=n, seed=5)
op = ic.optparam(4, 0, 8, 10, 0.0, 0, 0, n)
cam = ic.CamClass(5, sc["fc"], sc["cc"], sc["wh"], 8)
pa, pb = ic.Pyramid(sc["img_a"], 4, 8), ic.Pyramid(sc["img_b"], 4, 8)
eng = ic.TrackBatch(cam, op, 1)
eng.Set3Dpoints(0, sc["pts3d"].copy())
ts = []
for r in range(30):
    t0 = time.perf_counter()
    eng.SetPose(0, sc["p_a"], pa, pb)
    t1 = time.perf_counter()
    eng.track_async()
    t2 = time.perf_counter()
    p = eng.poses()
    t3 = time.perf_counter()
    ts.append((t1 - t0, t2 - t1, t3 - t2))
a = np.median(np.array(ts[5:]), 0) * 1e6
print("host us: setpose %.1f enqueue %.1f wait %.1f total %.1f" % (a[0], a[1], a[2], a.sum()))
