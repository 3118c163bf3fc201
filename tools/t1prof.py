"""Phase cycle counters of the one-launch tracker (needs a build with ICTR_EXTRA_HIPCC_FLAGS=-DICTR_T1_PROF)."""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
import invcompcamtrack_amd as ic
from invcompcamtrack_amd import synth
names = ["setup(w0)", "Hred+bar", "Hsum+LU", "stage1", "stage2(w0)", "red+bar", "sum+solve", "bar"]
for n in (60, 100, 300):
    sc = synth.make_scene(640, 480, n_points=n, seed=5)
    lv_f, psz = 4, 8
    op = ic.optparam(lv_f, 0, psz, 10, 0.0, 0, 0, n)
    cam = ic.CamClass(lv_f + 1, sc["fc"], sc["cc"], sc["wh"], psz)
    pa, pb = ic.Pyramid(sc["img_a"], lv_f, psz), ic.Pyramid(sc["img_b"], lv_f, psz)
    eng = ic.TrackBatch(cam, op, 1)
    eng.Set3Dpoints(0, sc["pts3d"].copy())
    ts = []
    for r in range(10):
        eng.SetPose(0, sc["p_a"], pa, pb)
        t0 = time.perf_counter(); eng.track_async(); eng.poses(); ts.append(time.perf_counter() - t0)
    c = eng.read_buffer(0, 9, 8)
    tot = c.sum()
    print(n, "ms", round(np.median(ts) * 1e3, 3), {k: int(v) for k, v in zip(names, c)}, "total cycles", int(tot), flush=True)
