"""Per-model timing of the full-frame alignment engine (extension; BASELINE configs 2, 3, 5 shapes).
Distinct frame pairs per problem so that HBM traffic is real. Prints one JSON object per config."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import invcompcamtrack_amd as ic  # noqa: E402
from invcompcamtrack_amd import icgn  # noqa: E402

CONFIGS = {
    "C2-vga-se2": dict(w=640, h=480, model="se2", p=[0.01, 2.3, -1.4], lv_f=2, B=16),
    "C3-1080p-affine": dict(w=1920, h=1080, model="affine", p=[0.003, -0.002, 0.004, -0.003, 3.1, -2.2], lv_f=2, B=8),
    "C5-4k-homography": dict(w=3840, h=2160, model="homography",
                             p=[0.002, -0.001, 2e-6, 0.002, -0.002, -3e-6, 3.1, -2.2], lv_f=3, B=4),
    "X-4k-translation": dict(w=3840, h=2160, model="translation", p=[3.1, -2.2], lv_f=3, B=4),
    "X-4k-affine": dict(w=3840, h=2160, model="affine", p=[0.002, -0.001, 0.002, -0.002, 3.1, -2.2], lv_f=3, B=4),
}


def main(names, steps=5, maxiter=10, pad=int(__import__('os').environ.get('ICGN_PAD', '16'))):
    for name in names:
        c = CONFIGS[name]
        w, h, lv_f, B = c["w"], c["h"], c["lv_f"], c["B"]
        C = np.array([[1, 0, w / 2], [0, 1, h / 2], [0, 0, 1.0]])
        Mgt = C @ icgn.warp_matrix(c["model"], c["p"]) @ np.linalg.inv(C)
        eng = icgn.AlignBatch(c["model"], w, h, lv_f, 0, maxiter, 0.0, None, B)
        keep = []
        for k in range(B):
            a, b = icgn.make_warped_pair(w, h, Mgt, seed=100 + k)
            pa, pb = ic.Pyramid(a, lv_f, pad), ic.Pyramid(b, lv_f, pad, getgrad=False)
            eng.set_frames(k, pa, pb)
            keep.append((pa, pb))
        eng.set_timing(True)
        eng.run_async()
        M, it, dp = eng.results()
        c4 = np.array([[0, 0, 1], [w, 0, 1], [0, h, 1], [w, h, 1.0]]).T
        errs = []
        for k in range(B):
            x, y = M[k] @ c4, Mgt @ c4
            errs.append(float(np.abs(x[:2] / x[2] - y[:2] / y[2]).max()))
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.run_async()
        eng.results()
        dt = (time.perf_counter() - t0) / steps
        kt = eng.kernel_times()
        npx = [((w - 4) >> l) * ((h - 4) >> l) for l in range(lv_f + 1)]
        px_iters = B * sum(npx) * maxiter
        out = dict(config=name, pad=pad, B=B, maxiter=maxiter, ms_per_step=dt * 1e3, gpix_iter_per_s=px_iters / dt / 1e9,
                   corner_err_px_max=max(errs), iters=int(it[0]),
                   level_kernel_us=[float(kt[l]) * 1e3 / maxiter for l in range(lv_f + 1)],
                   level_gbps=[16.0 * B * npx[l] / (float(kt[l]) * 1e-3 / maxiter) / 1e9 if kt[l] > 0 else None
                               for l in range(lv_f + 1)])
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or list(CONFIGS))
