"""Experiment: does running the B frame pairs as S sub-batches on S HIP streams fill the tail/launch gaps?"""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
import torch
import invcompcamtrack_amd as ic
from invcompcamtrack_amd import synth

def run(S, B=16, steps=10, w=1920, h=1080, P=8, lv_f=2, maxiter=10):
    scenes = [synth.make_scene(w, h, grid_step=P, margin=P / 2.0, jitter=0.35, seed=100 + s, tex_seed=1234 + s,
                               dp_gt=np.array([0.02, -0.015, 0.03, 0.003, -0.002, 0.004]) * (1 + 0.5 * s)) for s in range(2)]
    n = scenes[0]["pts3d"].shape[1]
    op = ic.optparam(lv_f, 0, P, maxiter, 0.0, 0, 0, n)
    cam = ic.CamClass(lv_f + 1, scenes[0]["fc"], scenes[0]["cc"], scenes[0]["wh"], P)
    engines, streams, pyrs = [], [], []
    per = B // S
    for s in range(S):
        e = ic.TrackBatch(cam, op, per)
        st = torch.cuda.Stream()
        e.set_stream(st.cuda_stream)
        engines.append(e); streams.append(st)
        for b in range(per):
            sc = scenes[(s * per + b) % 2]
            pa, pb = ic.Pyramid(sc["img_a"], lv_f, P), ic.Pyramid(sc["img_b"], lv_f, P)
            pyrs.append((pa, pb))
            e.Set3Dpoints(b, sc["pts3d"].copy())
    def step():
        for s, e in enumerate(engines):
            for b in range(per):
                pa, pb = pyrs[s * per + b]
                e.SetPose(b, scenes[(s * per + b) % 2]["p_a"], pa, pb)
        for e in engines:
            e.track_async()
        return [e.poses() for e in engines]
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        poses = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    pix = (lv_f + 1) * maxiter * B * n * P * P
    err = max(np.abs(p[b] - scenes[(s * per + b) % 2]["p_b"]).max() for s, p in enumerate(poses) for b in range(per))
    print(json.dumps(dict(S=S, B=B, ms_per_step=dt * 1e3, gpix_s=pix / dt / 1e9, err=float(err))), flush=True)

B_ = int(__import__("os").environ.get("EXP_B", "16"))
for S in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    run(S, B=B_)
