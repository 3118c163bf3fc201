"""Mid-size batches (a few hundred to a few thousand 8x8 patches per problem): the one-launch tracker's team form
against the resident-iteration form (forced: teams off, ICTR_RESIDENT_MINPTS=100 in the environment).
    ICTR_RESIDENT_MINPTS=100 python tools/mid_ab.py"""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
import invcompcamtrack_amd as ic
from invcompcamtrack_amd import synth
cases = [tuple(int(x) for x in a.split('x')) for a in sys.argv[1:]] or [(1000, 64), (3000, 16), (2000, 32), (5000, 8), (1000, 256), (600, 128)]
for n, B in cases:
    sc = synth.make_scene(640, 480, n_points=n, seed=5)
    op = ic.optparam(4, 0, 8, 10, 0.0, 0, 0, n)
    cam = ic.CamClass(5, sc["fc"], sc["cc"], sc["wh"], 8)
    pa, pb = ic.Pyramid(sc["img_a"], 4, 8), ic.Pyramid(sc["img_b"], 4, 8)
    out = {}
    for name, team in (("team", None), ("resident", -1)):
        e = ic.TrackBatch(cam, op, B)
        if team is not None:
            e.set_team(team)
        for k in range(B):
            e.Set3Dpoints(k, sc["pts3d"].copy())
        P = np.tile(sc["p_a"], (B, 1))
        ts = []
        for r in range(25):
            t0 = time.perf_counter()
            e.SetPoseAll(P, pa, pb)
            e.track_async()
            p = e.poses()
            ts.append(time.perf_counter() - t0)
        out[name] = (round(float(np.median(ts[5:])) * 1e3, 4), e.path_name()[:28], p[0][:2].tolist())
    print(json.dumps({"points": n, "problems": B, **{k: v for k, v in out.items()}}), flush=True)
