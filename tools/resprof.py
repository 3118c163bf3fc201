"""Phase cycle counters of the resident-iteration kernel (needs a build with ICTR_EXTRA_HIPCC_FLAGS=-DICTR_RES_PROF):
wave 0 of worker workgroup 0 and of slot 0's solver workgroup, summed over one tracking's launches of the LAST level."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import invcompcamtrack_amd as ic
from invcompcamtrack_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sc = synth.make_scene(1920, 1080, n_points=32400, seed=11)
pa, pb = ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8)
cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
op = ic.optparam(2, 0, 8, 10, 0.0, 0, 0, 32400)
e = ic.TrackBatch(cam, op, B)
for k in range(B):
    e.Set3Dpoints(k, sc["pts3d"].copy())
for r in range(3):
    e.SetPoseAll(np.tile(sc["p_a"], (B, 1)), pa, pb)
    e.track_async()
    e.poses()
c = e.read_buffer(0, 9, 16)
print(e.path_name())
names_w = ["barrier", "stage1", "stage2", "reduce+barrier", "gather+wait broadcast", "pair prologue", "-", "-"]
names_s = ["loop/barrier", "wait+sum granules", "barrier", "solve+broadcast", "-", "-", "-", "-"]
print("worker (level 0 launch, wave 0 of workgroup 0):", {n: int(v) for n, v in zip(names_w, c[:8]) if n != "-"}, "total", int(c[:8].sum()))
print("solver (slot 0):", {n: int(v) for n, v in zip(names_s, c[8:]) if n != "-"}, "total", int(c[8:].sum()))
