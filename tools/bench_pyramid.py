"""Pyramid builder (util_constructpyramide on the GPU) from a device-resident frame: ms per frame."""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
import torch
import invcompcamtrack_amd as ic

for (w, h, lv_f) in ((640, 480, 4), (1920, 1080, 2), (3840, 2160, 3)):
    img = (torch.rand(h, w, device="cuda") * 255).contiguous()
    torch.cuda.synchronize()
    keep = []
    for getgrad in (True, False):
        for _ in range(3):
            keep.append(ic.Pyramid(device_ptr=img.data_ptr(), wh=(w, h), lv_f=lv_f, imgpadding=8, getgrad=getgrad))
        keep.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            keep.append(ic.Pyramid(device_ptr=img.data_ptr(), wh=(w, h), lv_f=lv_f, imgpadding=8, getgrad=getgrad))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        keep.clear()
        px = sum((w >> l) * (h >> l) for l in range(lv_f + 1))
        print(json.dumps(dict(frame=f"{w}x{h}", levels=lv_f + 1, gradients=getgrad, ms_per_frame=round(dt * 1e3, 3),
                              gpix_per_s=round(px / dt / 1e9, 2),
                              note="includes hipMalloc of the arena and the packed {img,dx,dy,0} planes when gradients")),
              flush=True)
