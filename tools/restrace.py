"""Time line of the resident-iteration kernel on one CU (needs a build with ICTR_EXTRA_HIPCC_FLAGS=-DICTR_RES_PROF):
wall-clock stamps (100 MHz) of wave 0 of the four worker workgroups that share a CU (one per slot) and of the slots'
solver workgroups, for the LAST level launch of a tracking.   python tools/restrace.py [B] [iterations to print]"""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, ".")
import invcompcamtrack_amd as ic
from invcompcamtrack_amd import synth, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nprint = int(sys.argv[2]) if len(sys.argv) > 2 else 24
sc = synth.make_scene(1920, 1080, n_points=32400, seed=11)
pa, pb = ic.Pyramid(sc["img_a"], 2, 8), ic.Pyramid(sc["img_b"], 2, 8)
cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
op = ic.optparam(2, 0, 8, 10, 0.0, 0, 0, 32400)
e = ic.TrackBatch(cam, op, B)
for k in range(B):
    e.Set3Dpoints(k, sc["pts3d"].copy())
L = _lib.load()
for r in range(3):
    e.SetPoseAll(np.tile(sc["p_a"], (B, 1)), pa, pb)
    e.track_async()
    e.poses()
tr = np.zeros((16, 128, 4), np.uint64)
assert L.ictr_prof_res_trace(tr.ctypes.data_as(C.c_void_p), 0) == 0
tr = tr.astype(np.int64)
t0 = tr[tr > 0].min()
rel = np.where(tr > 0, (tr - t0) / 100.0, np.nan)  # microseconds
print(e.path_name())
print("worker rows: [loop top, stage 2 done, sums in LDS (barrier), broadcast received]; solver rows: [loop top, all granules in, barrier, broadcast stored]  (us)")
for it in range(nprint):
    print(f"it {it:3d} " + " | ".join("W%d " % s + " ".join(f"{rel[s, it, k]:7.2f}" for k in range(4)) for s in range(4)))
    print("       " + " | ".join("S%d " % s + " ".join(f"{rel[8 + s, it, k]:7.2f}" for k in range(4)) for s in range(4)))
w = rel[:4]
n = int(np.isfinite(w[0, :, 0]).sum())
for s in range(4):
    d = np.diff(w[s, :n, 0])
    print(f"slot {s}: iterations {n}, median period {np.median(d):.2f} us; stage1+2 {np.nanmedian(w[s,:n,1]-w[s,:n,0]):.2f}, reduce+barrier {np.nanmedian(w[s,:n,2]-w[s,:n,1]):.2f}, "
          f"wait {np.nanmedian(w[s,:n,3]-w[s,:n,2]):.2f}; solver: wait {np.nanmedian(rel[8+s,:n,1]-rel[8+s,:n,0]):.2f} barrier {np.nanmedian(rel[8+s,:n,2]-rel[8+s,:n,1]):.2f} solve {np.nanmedian(rel[8+s,:n,3]-rel[8+s,:n,2]):.2f}; "
          f"post->solver sees {np.nanmedian(rel[8+s,:n,1]-w[s,:n,2]):.2f}, broadcast->worker sees {np.nanmedian(w[s,:n,3]-rel[8+s,:n,3]):.2f}")
