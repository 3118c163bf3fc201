"""Single-problem latency at the reference's own problem sizes (run_odometer_test.m: a few hundred points, 8x8 or 4x4
patches, 5 levels) -- GPU OdometerClass.TrackPose vs the CPU oracle, same inputs."""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
import invcompcamtrack_amd as ic
from invcompcamtrack_amd import synth
from oracle import oracle as O

def one(w, h, n, psz, lv_f, maxiter, reps=30):
    sc = synth.make_scene(w, h, n_points=n, seed=5)
    op = ic.optparam(lv_f, 0, psz, maxiter, 0.0, 0, 0, n)
    cam = ic.CamClass(lv_f + 1, sc["fc"], sc["cc"], sc["wh"], psz)
    pose = ic.PoseClass(cam, op)
    odo = ic.OdometerClass(pose, op)
    pa, pb = ic.Pyramid(sc["img_a"], lv_f, psz), ic.Pyramid(sc["img_b"], lv_f, psz)
    odo.Set3Dpoints(sc["pts3d"].copy())
    ts = []
    for r in range(reps + 5):
        t0 = time.perf_counter()
        odo.SetPose(sc["p_a"], pa, pb)
        p = odo.TrackPose()
        ts.append(time.perf_counter() - t0)
    gpu_ms = float(np.median(ts[5:]) * 1e3)
    oop = O.make_op(lv_f, 0, psz, maxiter, 0.0, 0, 0, n)
    tr = O.Tracker(oop, sc["fc"], sc["cc"], sc["wh"])
    opa, opb = O.Pyramid(sc["img_a"], lv_f, psz), O.Pyramid(sc["img_b"], lv_f, psz)
    tc = []
    for r in range(5):
        t0 = time.perf_counter()
        tr.set3dpoints(sc["pts3d"].copy()); tr.setpose(sc["p_a"], opa, opb); pc = tr.trackpose()
        tc.append(time.perf_counter() - t0)
    cpu_ms = float(np.median(tc) * 1e3)
    print(json.dumps(dict(frame=f"{w}x{h}", points=n, psz=psz, levels=lv_f + 1, maxiter=maxiter, gpu_ms=round(gpu_ms, 3),
                          cpu_oracle_ms=round(cpu_ms, 3), launches=(lv_f + 1) * (2 + 2 * maxiter) + 1,
                          pose_diff=float(np.abs(p - pc).max()))), flush=True)

for n in (100, 300, 1000, 5000):
    one(640, 480, n, 8, 4, 10)
one(640, 480, 300, 4, 4, 5)
one(1920, 1080, 32400, 8, 2, 10, reps=10)
