"""A/B of the gradients-on-the-fly path: (1) pyramid refill of a 1080p frame with gradient planes
(getgrad 1) against image levels only (getgrad 2); (2) the headline step with the 8x8 setup kernel reading the packed
planes (variant bit 27) against forming the gradients from the image plane (the default), resident and streaming form.
    python tools/otf_ab.py"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch
import invcompcamtrack_amd as ic
from invcompcamtrack_amd import synth

w, h, lv_f, pad, K = 1920, 1080, 2, 8, 16
frame = (torch.rand(h, w, device="cuda") * 255).contiguous()
st = torch.cuda.current_stream().cuda_stream
for gg in (1, 2):
    pyr = ic.Pyramid(lv_f=lv_f, imgpadding=pad, device_ptr=frame.data_ptr(), wh=(w, h), stream=st, getgrad=gg)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for rep in range(12):
        ev0.record()
        for _ in range(K):
            pyr.rebuild(device_ptr=frame.data_ptr(), stream=st)
        ev1.record()
        ev1.synchronize()
        ts.append(ev0.elapsed_time(ev1) / K)
    print(json.dumps({"pyramid_refill_1080p_3_levels": {"getgrad": gg, "us_per_frame": round(float(np.median(ts[2:])) * 1e3, 2)}}), flush=True)

sc = synth.make_scene(1920, 1080, n_points=32400, seed=11)
cam = ic.CamClass(3, sc["fc"], sc["cc"], sc["wh"], 8)
op = ic.optparam(2, 0, 8, 10, 0.0, 0, 0, 32400)
pb = ic.Pyramid(sc["img_b"], 2, 8, getgrad=0)
B = 32
for name, variant, gg in (("resident, packed planes (bit 27)", 1 << 27, 1), ("resident, on the fly", 0, 1),
                          ("resident, image-only pyramid", 0, 2), ("streaming, packed planes", (1 << 21) | (1 << 27), 1),
                          ("streaming, on the fly", 1 << 21, 1)):
    pa = ic.Pyramid(sc["img_a"], 2, 8, getgrad=gg)
    e = ic.TrackBatch(cam, op, B)
    e.set_variant(variant)
    e.set_timing(True)
    for k in range(B):
        e.Set3Dpoints(k, sc["pts3d"].copy())
    P = np.tile(sc["p_a"], (B, 1))
    ts, setup = [], []
    for r in range(13):
        t0 = time.perf_counter()
        e.SetPoseAll(P, pa, pb)
        e.track_async()
        p = e.poses()
        ts.append(time.perf_counter() - t0)
        setup.append(e.level_times()[0].copy())
    print(json.dumps({"step_32_pairs": name, "ms": round(float(np.median(ts[3:])) * 1e3, 3),
                      "setup_us_per_level": [round(float(x) * 1e3, 1) for x in np.median(np.array(setup[3:]), 0)],
                      "pose0": [round(float(x), 8) for x in p[0][:3]]}), flush=True)
    del e, pa
