"""k_ref8 under the bench's conditions (32 DISTINCT 1080p frame pairs, resident form): per-level setup time.
    python tools/ref8_bench.py"""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
import invcompcamtrack_amd as ic
from invcompcamtrack_amd import synth
mode = sys.argv[1] if len(sys.argv) > 1 else "distinct-grid"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32   # distinct-grid (the bench) | shared-grid | shared-random | distinct-random
if "grid" in mode:
    scs = [synth.make_scene(1920, 1080, grid_step=8, margin=4.0, jitter=0.35, seed=100 + s, tex_seed=1234 + s) for s in range(2)]
else:
    scs = [synth.make_scene(1920, 1080, n_points=32400, seed=11 + s) for s in range(2)]
n = scs[0]["pts3d"].shape[1]
cam = ic.CamClass(3, scs[0]["fc"], scs[0]["cc"], scs[0]["wh"], 8)
op = ic.optparam(2, 0, 8, 10, 0.0, 0, 0, n)
pyr = [(ic.Pyramid(scs[b % 2]["img_a"], 2, 8), ic.Pyramid(scs[b % 2]["img_b"], 2, 8)) for b in range(B if "distinct" in mode else 2)]
pyr = [pyr[b % len(pyr)] for b in range(B)]
e = ic.TrackBatch(cam, op, B)
e.set_timing(True)
for k in range(B):
    e.Set3Dpoints(k, scs[k % 2]["pts3d"].copy())
ts, setup, iters = [], [], []
for r in range(12):
    t0 = time.perf_counter()
    for k in range(B):
        e.SetPose(k, scs[k % 2]["p_a"], *pyr[k])
    e.track_async()
    p = e.poses()
    ts.append(time.perf_counter() - t0)
    a, b = e.level_times()
    setup.append(a.copy()); iters.append(b.copy())
print(json.dumps({"mode": mode, "B": B, "ms": round(float(np.median(ts[3:])) * 1e3, 3),
                  "setup_us_per_level": [round(float(x) * 1e3, 1) for x in np.median(np.array(setup[3:]), 0)],
                  "resident_us_per_level": [round(float(x) * 1e3, 1) for x in np.median(np.array(iters[3:]), 0)]}))
