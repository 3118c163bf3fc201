import os, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import invcompcamtrack_amd as ic
from invcompcamtrack_amd import synth
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
w, h, P, lv_f = 1920, 1080, 8, 2
sc = synth.make_scene(w, h, grid_step=P, margin=P / 2.0, jitter=0.35, seed=100)
n = sc["pts3d"].shape[1]
op = ic.optparam(lv_f, 0, P, 10, 0.0, 0, 0, n)
cam = ic.CamClass(lv_f + 1, sc["fc"], sc["cc"], sc["wh"], P)
eng = ic.TrackBatch(cam, op, B)
pa, pb = ic.Pyramid(sc["img_a"], lv_f, P), ic.Pyramid(sc["img_b"], lv_f, P)
for b in range(B):
    eng.Set3Dpoints(b, sc["pts3d"].copy())
def setposes():
    t = time.perf_counter()
    for b in range(B):
        eng.SetPose(b, sc["p_a"], pa, pb)
    return (time.perf_counter() - t) * 1e3
def pyloop():
    t = time.perf_counter(); s = 0
    for i in range(20000): s += i * i
    return (time.perf_counter() - t) * 1e3
print("idle: setpose ms", [round(setposes(), 3) for _ in range(3)], "pyloop ms", round(pyloop(), 3))
for timing in (False, True):
    eng.set_timing(timing)
    for rep in range(3):
        sp = setposes(); eng.track_async()
        busy_sp = setposes(); busy_py = pyloop()
        t = time.perf_counter(); eng.poses(); w_ = (time.perf_counter() - t) * 1e3
        after_sp = setposes(); after_py = pyloop()
        if timing: eng.level_times(); eng.kernel_times()
        ev_sp = setposes()
        print(f"timing={timing} rep{rep}: before {sp:.3f} | GPU busy: setpose {busy_sp:.3f} pyloop {busy_py:.3f} | wait {w_:.2f} | after sync: setpose {after_sp:.3f} pyloop {after_py:.3f} | after event reads {ev_sp:.3f}")
