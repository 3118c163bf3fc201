"""One latency case for profiling: python tools/lat_one.py POINTS PROBLEMS [REPS] (640x480, 5 levels x 10 iterations)."""
import sys
sys.path.insert(0, ".")
from tools.latency import one
n, B = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
import tools.latency as L
# only the default form
import invcompcamtrack_amd as ic, numpy as np, time, json
from invcompcamtrack_amd import synth
sc = synth.make_scene(640, 480, n_points=n, seed=5)
op = ic.optparam(4, 0, 8, 10, 0.0, 0, 0, n)
cam = ic.CamClass(5, sc["fc"], sc["cc"], sc["wh"], 8)
pa, pb = ic.Pyramid(sc["img_a"], 4, 8), ic.Pyramid(sc["img_b"], 4, 8)
eng = ic.TrackBatch(cam, op, B)
for k in range(B):
    eng.Set3Dpoints(k, sc["pts3d"].copy())
ts = []
for r in range(reps + 5):
    t0 = time.perf_counter()
    if B > 1:
        eng.SetPoseAll(np.tile(sc["p_a"], (B, 1)), pa, pb)
    else:
        eng.SetPose(0, sc["p_a"], pa, pb)
    eng.track_async()
    p = eng.poses()
    ts.append(time.perf_counter() - t0)
print(json.dumps({"points": n, "problems": B, "ms": round(float(np.median(ts[5:])) * 1e3, 4), "path": eng.path_name()}))
