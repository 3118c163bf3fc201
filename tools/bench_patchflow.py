"""BASELINE config 4 shape: 4096 independent 31x31 patches on a 1920x1080 frame pair, 3 levels (flow producer)."""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
import invcompcamtrack_amd as ic
from invcompcamtrack_amd import patchflow as pf, synth

w, h, lv_f, psz = 1920, 1080, 2, 31
sc = synth.make_scene(w, h, n_points=10, seed=3, dp_gt=np.array([0.02, -0.015, 0.03, 0.003, -0.002, 0.004]))
pa, pb = ic.Pyramid(sc["img_a"], lv_f, 32), ic.Pyramid(sc["img_b"], lv_f, 32)
rng = np.random.default_rng(7)
gx, gy = np.meshgrid(np.linspace(60, w - 60, 64), np.linspace(60, h - 60, 64))
pts = (np.stack([gx.ravel(), gy.ravel()], 1) + rng.uniform(-1.5, 1.5, (4096, 2))).astype(np.float32)
for K in (4096, 65536):
    p = np.tile(pts, (K // 4096, 1)) + rng.uniform(-3, 3, (K, 2)).astype(np.float32)
    new, ok, it = pf.track_points(pa, pb, p, psz=psz, lv_f=lv_f, maxiter=10, eps=0.0)
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        new, ok, it = pf.track_points(pa, pb, p, psz=psz, lv_f=lv_f, maxiter=10, eps=0.0)
    dt = (time.perf_counter() - t0) / reps
    pix_iters = float(it[ok].sum()) * psz * psz
    print(json.dumps(dict(patches=K, psz=psz, levels=lv_f + 1, ms=round(dt * 1e3, 3), tracked=int(ok.sum()),
                          mean_iters=float(it[ok].mean()), gpix_iter_per_s=round(pix_iters / dt / 1e9, 2),
                          note="includes host<->device copies of points/results")), flush=True)
