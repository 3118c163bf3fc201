"""4x4 fast path (k_ref4 / k_iter4) against the any-size kernels for a forced chunk size: python tools/probes/cpw4_probe.py
(set ICTR_CPW in the environment)."""
import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import invcompcamtrack_amd as ic
from invcompcamtrack_amd import synth
sc = synth.make_scene(256, 224, n_points=300, seed=9, margin=12.0)
res = []
for variant in (8192, 8192 | 2):
    op = ic.optparam(2, 0, 4, 6, 0.0, 0, 0, 300)
    cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 4)
    odo = ic.OdometerClass(ic.PoseClass(cam, op), op)
    odo.set_variant(variant)
    odo.enable_trace()
    odo.Set3Dpoints(sc["pts3d"].copy())
    pa, pb = ic.Pyramid(sc["img_a"], 2, 4), ic.Pyramid(sc["img_b"], 2, 4)
    odo.SetPose(sc["p_a"], pa, pb)
    p = odo.TrackPose()
    tr = odo.trace()
    res.append((p, odo.read_buffer(0, 16 * 300), odo.read_buffer(1, 16 * 300), odo.read_buffer(2, 16 * 300), odo.read_buffer(7, 16 * 300), tr))
f, g = res
print("ICTR_CPW", os.environ.get("ICTR_CPW"), "pose diff", np.abs(f[0] - g[0]).max(), "buffers equal", [bool(np.array_equal(f[i], g[i])) for i in (1, 2, 3, 4)])
for k in range(min(4, len(f[5]))):
    Hf, Hg, bf, bg = f[5][k]["H"], g[5][k]["H"], f[5][k]["b"], g[5][k]["b"]
    print("  trace", k, "level", f[5][k]["level"], "iter", f[5][k]["iter"], "H rel", np.abs(Hf - Hg).max() / np.abs(Hg).max(), "b rel", np.abs(bf - bg).max() / np.abs(bg).max())
