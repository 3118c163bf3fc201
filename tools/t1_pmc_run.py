"""Workload for PMC passes on the one-launch tracker: 256 problems x N points (one workgroup per CU), a few launches."""
import sys
import numpy as np
sys.path.insert(0, ".")
import invcompcamtrack_amd as ic
from invcompcamtrack_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
sc = synth.make_scene(640, 480, n_points=n, seed=5)
lv_f, psz = 4, 8
op = ic.optparam(lv_f, 0, psz, 10, 0.0, 0, 0, n)
cam = ic.CamClass(lv_f + 1, sc["fc"], sc["cc"], sc["wh"], psz)
pa, pb = ic.Pyramid(sc["img_a"], lv_f, psz), ic.Pyramid(sc["img_b"], lv_f, psz)
eng = ic.TrackBatch(cam, op, B)
for k in range(B):
    eng.Set3Dpoints(k, sc["pts3d"].copy())
for r in range(5):
    for k in range(B):
        eng.SetPose(k, sc["p_a"], pa, pb)
    eng.track_async()
    p = eng.poses()
print(eng.path_name(), float(np.abs(p - sc["p_b"]).max()))
