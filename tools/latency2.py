"""Where does the small-problem latency go: host enqueue vs GPU dependency chain (TrackBatch B=1)."""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
import invcompcamtrack_amd as ic
from invcompcamtrack_amd import synth
for n, B in ((300, 1), (300, 8), (300, 64), (1000, 1)):
    sc = synth.make_scene(640, 480, n_points=n, seed=5)
    lv_f, psz, maxiter = 4, 8, 10
    op = ic.optparam(lv_f, 0, psz, maxiter, 0.0, 0, 0, n)
    cam = ic.CamClass(lv_f + 1, sc["fc"], sc["cc"], sc["wh"], psz)
    b = ic.TrackBatch(cam, op, B)
    pa, pb = ic.Pyramid(sc["img_a"], lv_f, psz), ic.Pyramid(sc["img_b"], lv_f, psz)
    for k in range(B):
        b.Set3Dpoints(k, sc["pts3d"].copy())
    enq, tot = [], []
    for r in range(40):
        for k in range(B):
            b.SetPose(k, sc["p_a"], pa, pb)
        t0 = time.perf_counter(); b.track_async(); t1 = time.perf_counter(); b.poses(); t2 = time.perf_counter()
        enq.append(t1 - t0); tot.append(t2 - t0)
    print(json.dumps(dict(points=n, B=B, enqueue_ms=round(float(np.median(enq[5:])) * 1e3, 3),
                          total_ms=round(float(np.median(tot[5:])) * 1e3, 3),
                          per_problem_ms=round(float(np.median(tot[5:])) * 1e3 / B, 4))), flush=True)
