"""Short timed records for the workloads beside bench.py's headline, one dict each:

  psz4        the reference's other parameter set ("4 0 4 5 0.01 0 0", run_odometer_test.m:140) on the 1080p workload
  C3          BASELINE config 3 read literally: full-frame 1920x1080 6-parameter affine, 3 levels (extension engine)
  C5          BASELINE config 5: 3840x2160 8-parameter homography, 4 levels (extension engine)
  C4          BASELINE config 4: 4096 independent 31x31 patches on a 1080p pair, 3 levels (flow producer)
  nposes      run_track_nposes' shape (run_ransac_test.m:67,88): 500 pose samples x 60 points, 10 frame pairs
  small       one 100-point frame pair at the reference's own size (run_odometer_test.m), latency
  dense       ONE dense 1080p frame pair per tracking (the headline's pair, unbatched): resident-iteration form
  pyramid     util_constructpyramide (utilities.cpp:14-52) of a 1080p frame, 3 levels, refilled in place per frame

Every record carries its own algorithmic bytes, the measured kernel time (HIP events on the launching stream) and
achieved / 8 TB/s. Extension engines (C3, C4, C5) are build-defined: "parity unpinned by the reference".
Imported by bench.py only after the headline measurement is complete.
"""
from __future__ import annotations

import os
import time

import numpy as np

PEAK = 8000.0  # GB/s


def _budget_loop(fn, seconds, min_reps=2, max_reps=50):
    fn()  # warm-up
    reps, t0 = 0, time.perf_counter()
    while reps < min_reps or (time.perf_counter() - t0 < seconds and reps < max_reps):
        fn()
        reps += 1
    return (time.perf_counter() - t0) / reps, reps


def rec_psz4(seconds):
    import invcompcamtrack_amd as ic
    from invcompcamtrack_amd import synth
    w, h, P, lv_f, B, maxiter = 1920, 1080, 4, 2, 16, 5
    scs = [synth.make_scene(w, h, grid_step=P, margin=P / 2.0, jitter=0.35, seed=100 + s, tex_seed=1234 + s,
                            dp_gt=np.array([0.02, -0.015, 0.03, 0.003, -0.002, 0.004]) * (1 + 0.5 * s))
           for s in range(2)]
    n = scs[0]["pts3d"].shape[1]
    op = ic.optparam(lv_f, 0, P, maxiter, 0.0, 0, 0, n)
    cam = ic.CamClass(lv_f + 1, scs[0]["fc"], scs[0]["cc"], scs[0]["wh"], P)
    eng = ic.TrackBatch(cam, op, B)
    pyrs = []
    for b in range(B):
        sc = scs[b % 2]
        pyrs.append((ic.Pyramid(sc["img_a"], lv_f, P), ic.Pyramid(sc["img_b"], lv_f, P)))
        eng.Set3Dpoints(b, sc["pts3d"].copy())
    eng.set_timing(True)

    def step():
        for b in range(B):
            eng.SetPose(b, scs[b % 2]["p_a"], *pyrs[b])
        eng.track_async()
        return eng.poses()

    dt, reps = _budget_loop(step, seconds)
    poses = step()
    kt = eng.kernel_times()
    nl = (lv_f + 1) * maxiter
    t_k = float(kt.sum()) / nl * 1e-3
    alg = 16.0 * n * P * P * B
    pix = (lv_f + 1) * maxiter * n * P * P * B
    return {"name": "psz4", "workload": f"R1080p-dense-se3 with 4x4 patches: {B} pairs x {n} points, 3 levels x {maxiter} "
            "iterations (run_odometer_test.m:140 parameter set)", "value": pix / dt / 1e6, "unit": "Mpix/s",
            "ms_per_step": dt * 1e3, "reps": reps, "kernel": "k_iter4", "kernel_us": t_k * 1e6,
            "algorithmic_bytes_per_launch": alg, "achieved_GBps": alg / t_k / 1e9, "frac": alg / t_k / 1e9 / PEAK,
            "pose_err_vs_ground_truth": float(max(np.abs(poses[b] - scs[b % 2]["p_b"]).max() for b in range(B)))}


def _rec_icgn(name, w, h, model, p, lv_f, B, seconds, what):
    import invcompcamtrack_amd as ic
    from invcompcamtrack_amd import icgn
    maxiter, pad = 10, 16
    Cm = np.array([[1, 0, w / 2], [0, 1, h / 2], [0, 0, 1.0]])
    Mgt = Cm @ icgn.warp_matrix(model, p) @ np.linalg.inv(Cm)
    eng = icgn.AlignBatch(model, w, h, lv_f, 0, maxiter, 0.0, None, B)
    keep = []
    for k in range(B):
        a, b = icgn.make_warped_pair(w, h, Mgt, seed=100 + k)
        pa, pb = ic.Pyramid(a, lv_f, pad), ic.Pyramid(b, lv_f, pad, getgrad=False)
        eng.set_frames(k, pa, pb)
        keep.append((pa, pb))
    eng.set_timing(True)

    def step():
        eng.run_async()
        return eng.results()

    dt, reps = _budget_loop(step, seconds)
    M, it, _ = step()
    kt = eng.kernel_times()
    c4 = np.array([[0, 0, 1], [w, 0, 1], [0, h, 1], [w, h, 1.0]]).T
    err = 0.0
    for k in range(B):
        x, y = M[k] @ c4, Mgt @ c4
        err = max(err, float(np.abs(x[:2] / x[2] - y[:2] / y[2]).max()))
    npx = [((w - 4) >> l) * ((h - 4) >> l) for l in range(lv_f + 1)]
    t0 = float(kt[0]) / maxiter * 1e-3           # level-0 iteration launch (the dominant one)
    alg0 = 16.0 * B * npx[0]
    return {"name": name, "workload": what + f"; {B} pairs per step, {lv_f + 1} levels x {maxiter} iterations; "
            "extension engine, parity unpinned by the reference", "value": B * sum(npx) * maxiter / dt / 1e6,
            "unit": "Mpix/s", "ms_per_step": dt * 1e3, "reps": reps, "kernel": "k_icgn_iter (level 0)",
            "kernel_us": t0 * 1e6, "algorithmic_bytes_per_launch": alg0, "achieved_GBps": alg0 / t0 / 1e9,
            "frac": alg0 / t0 / 1e9 / PEAK, "corner_err_px_vs_ground_truth": err, "iterations": int(it[0])}


def rec_c3(seconds):
    return _rec_icgn("C3", 1920, 1080, "affine", [0.003, -0.002, 0.004, -0.003, 3.1, -2.2], 2, 8, seconds,
                     "BASELINE config 3: full-frame 1920x1080 6-DoF affine IC-LK")


def rec_c5(seconds):
    return _rec_icgn("C5", 3840, 2160, "homography", [0.002, -0.001, 2e-6, 0.002, -0.002, -3e-6, 3.1, -2.2], 3, 4,
                     seconds, "BASELINE config 5: 3840x2160 8-parameter homography")


def rec_c4(seconds):
    import invcompcamtrack_amd as ic
    from invcompcamtrack_amd import patchflow as pf, synth
    w, h, lv_f, psz, K, maxiter = 1920, 1080, 2, 31, 4096, 10
    sc = synth.make_scene(w, h, n_points=10, seed=3, dp_gt=np.array([0.02, -0.015, 0.03, 0.003, -0.002, 0.004]))
    pa, pb = ic.Pyramid(sc["img_a"], lv_f, 32), ic.Pyramid(sc["img_b"], lv_f, 32)
    rng = np.random.default_rng(7)
    gx, gy = np.meshgrid(np.linspace(60, w - 60, 64), np.linspace(60, h - 60, 64))
    pts = (np.stack([gx.ravel(), gy.ravel()], 1) + rng.uniform(-1.5, 1.5, (K, 2))).astype(np.float32)
    res = {}

    def step():
        res["r"] = pf.track_points(pa, pb, pts, psz=psz, lv_f=lv_f, maxiter=maxiter, eps=0.0)

    dt, reps = _budget_loop(step, seconds)
    new, ok, it = res["r"]
    k_ms = pf.last_kernel_ms()
    n = psz * psz
    # the templates stay in registers: per level 12 B/px gathered once (T, Gx, Gy windows), per iteration one
    # current-frame texel per pixel
    alg = float(K) * n * ((lv_f + 1) * 12.0 + float(np.mean(it)) * 4.0)
    out = {"name": "C4", "workload": f"BASELINE config 4: {K} independent {psz}x{psz} patches on a 1920x1080 pair, "
           f"{lv_f + 1} levels x {maxiter} iterations, one launch; build-defined flow producer, parity unpinned",
           "value": float(it[ok].sum()) * n / dt / 1e6, "unit": "Mpix/s (pixel-iterations, host copies included)",
           "ms_per_step": dt * 1e3, "reps": reps, "kernel": "k_patchflow", "tracked": int(ok.sum()),
           "algorithmic_bytes_per_launch": alg}
    if k_ms is not None and k_ms > 0:
        out.update({"kernel_us": k_ms * 1e3, "achieved_GBps": alg / (k_ms * 1e-3) / 1e9,
                    "frac": alg / (k_ms * 1e-3) / 1e9 / PEAK,
                    "note": "latency-bound by design: 4096 waves, each a serial chain of 30 dependent iterations"})
    return out


def _small_scene(w, h, n, seed=5):
    from invcompcamtrack_amd import synth
    return synth.make_scene(w, h, n_points=n, seed=seed)


def rec_nposes(seconds, cpu_track=None):
    """500 pose samples x 60 points, 5 frame pairs forward + 5 backward (run_ransac_test.m:67,88,
    func_ransac_fitcameras_odom.m: psz 8, 5 levels, maxiter 10, normdp_ratio 0.01): every chain link is ONE batch
    tracking of all samples (samples are independent problems, run_track_nposes.cpp:193)."""
    import invcompcamtrack_amd as ic
    w, h, n, S, lv_f, P, links = 1280, 720, 60, 500, 4, 8, 10
    sc = _small_scene(w, h, n)
    op = ic.optparam(lv_f, 0, P, 10, 0.01, 0, 0, n)
    cam = ic.CamClass(lv_f + 1, sc["fc"], sc["cc"], sc["wh"], P)
    pa, pb = ic.Pyramid(sc["img_a"], lv_f, P), ic.Pyramid(sc["img_b"], lv_f, P)
    eng = ic.TrackBatch(cam, op, S)
    rng = np.random.default_rng(3)
    starts = sc["p_a"][None, :] + rng.normal(0, 2e-3, (S, 6))
    for k in range(S):
        eng.Set3Dpoints(k, sc["pts3d"].copy())

    def chain():
        # a forward / backward chain like run_track_nposes.cpp:229-258: link k tracks A -> B from the sample's pose,
        # link k+1 tracks B -> A from the pose just found; every link is one batch tracking of all samples
        p = starts
        for k in range(links):
            ref, new = (pa, pb) if k % 2 == 0 else (pb, pa)
            eng.SetPoseAll(p, ref, new)
            eng.track_async()
            p = eng.poses()
        return p

    dt, reps = _budget_loop(chain, seconds, min_reps=1, max_reps=5)
    p = chain()
    its = eng.iterations()
    pix = float(its.sum()) * n * P * P * links  # executed iterations of the last link x links (same inputs each link)
    cpu = {}
    if cpu_track is not None:
        # the CPU path on the FIRST chain link of a few samples (same start poses, frames, points, options): ms per
        # tracking on one host core, and the GPU batch's poses of that link against it
        eng.SetPoseAll(starts, pa, pb)
        eng.track_async()
        p1 = eng.poses()
        ks = list(range(0, S, S // 8))[:8]
        ms, diff = [], 0.0
        for k in ks:
            pc, m, _ = cpu_track(sc, lv_f, P, 10, 0.01, n, seconds=0.0, p_start=starts[k])
            ms.append(m)
            diff = max(diff, float(np.abs(p1[k] - pc).max()))
        cores = os.cpu_count() or 1
        cpu = {"cpu_ms_per_tracking": float(np.mean(ms)), "cpu_samples": len(ks), "pose_diff_vs_cpu": diff,
               "cpu_trackings_per_s_one_core": 1e3 / float(np.mean(ms)),
               "cpu_trackings_per_s_all_host_cores_extrapolated": cores * 1e3 / float(np.mean(ms)), "host_cores": cores}
    return {**cpu, "name": "nposes", "workload": f"run_track_nposes shape: {S} pose samples x {n} points x {links} frame "
            f"pairs (chained), 8x8 patches, 5 levels, maxiter 10, normdp_ratio 0.01, {w}x{h}",
            "value": S * links / dt, "unit": "trackings/s", "ms_per_step": dt * 1e3, "ms_per_chain_link": dt * 1e3 / links,
            "reps": reps, "kernel": eng.path_name() if hasattr(eng, "path_name") else "k_iter8 (per-iteration launches)",
            "mean_iterations": float(its.mean()), "aligned_Mpix_per_s": pix / dt / 1e6,
            "algorithmic_bytes_per_launch": None,
            # floor: every sample is one workgroup's dependency chain; 256 CUs x 2 workgroups run side by side
            "floor_ms_per_chain_link": issue_floor_ms(n, 5, float(its.mean()) / 5.0) * max(1, -(-S // 512)),
            "frac": issue_floor_ms(n, 5, float(its.mean()) / 5.0) * max(1, -(-S // 512)) / (dt * 1e3 / links),
            "note": "latency-bound (60-point problems): reported as trackings/s; frac = issue-rate floor of a chain link "
                    "(issue_floor_ms with the executed iteration count, samples beyond 2 workgroups per CU queue up) / "
                    "measured, not a bandwidth fraction; pose_diff_vs_cpu "
                    "= first chain link of 8 samples against the oracle (early exit on: iteration counts may differ)",
            "pose_err_vs_ground_truth_median": float(np.median(np.abs(p - sc["p_a"][None, :]).max(1)))}


def rec_dense(seconds, cpu_track=None):
    """The headline's frame pair as the reference would run it: ONE dense 1080p pair (and a batch of 4) per tracking,
    SetPose + TrackPose + poses on the host. Default = k_level_resident (all iterations of a level in one launch,
    templates resident on the chip); the streaming per-iteration kernels beside it (variant bit 21)."""
    import invcompcamtrack_amd as ic
    from invcompcamtrack_amd import synth
    w, h, lv_f, P = 1920, 1080, 2, 8
    sc = synth.make_scene(w, h, grid_step=P, margin=P / 2.0, jitter=0.35, seed=100)  # the headline's scene 0
    n = sc["pts3d"].shape[1]
    op = ic.optparam(lv_f, 0, P, 10, 0.0, 0, 0, n)
    cam = ic.CamClass(lv_f + 1, sc["fc"], sc["cc"], sc["wh"], P)
    pa, pb = ic.Pyramid(sc["img_a"], lv_f, P), ic.Pyramid(sc["img_b"], lv_f, P)
    cases = []
    poses_1 = {}
    for B in (1, 4):
        row = {"pairs": B, "points_per_pair": n}
        for name, variant in (("default", 0), ("streaming", 1 << 21)):
            eng = ic.TrackBatch(cam, op, B)
            eng.set_variant(variant)
            for k in range(B):
                eng.Set3Dpoints(k, sc["pts3d"].copy())
            p_all = np.tile(sc["p_a"], (B, 1))

            def step():
                eng.SetPoseAll(p_all, pa, pb)
                eng.track_async()
                return eng.poses()

            step()
            ts = []
            t_end = time.perf_counter() + seconds / 4
            while len(ts) < 5 or (time.perf_counter() < t_end and len(ts) < 100):
                t0 = time.perf_counter()
                p = step()
                ts.append(time.perf_counter() - t0)
            row[name + "_ms"] = float(np.median(ts)) * 1e3
            row[name + "_kernel"] = eng.path_name()
            row[name + "_pose_err_vs_ground_truth"] = float(np.abs(p - sc["p_b"][None, :]).max())
            if B == 1:
                poses_1[name] = p[0].copy()
        cases.append(row)
    pix = 30.0 * n * P * P
    cpu = {}
    if cpu_track is not None:
        pc, m, runs = cpu_track(sc, lv_f, P, 10, 0.0, n, seconds=0.0)
        cpu = {"cpu_ms": m, "cpu_runs": runs,
               "pose_diff_vs_cpu": {k_: float(np.abs(v_ - pc).max()) for k_, v_ in poses_1.items()}}
    return {**cpu, "name": "dense", "workload": f"ONE dense {w}x{h} frame pair per tracking ({n} 8x8 patches, 3 levels x 10 "
            "iterations), host calls included; and a batch of 4", "value": cases[0]["default_ms"], "unit": "ms per tracking "
            "(1 pair)", "aligned_Mpix_per_s": pix / (cases[0]["default_ms"] * 1e-3) / 1e6, "cases": cases,
            "kernel": cases[0]["default_kernel"], "algorithmic_bytes_per_launch": None,
            # floor: the pair's serial chain -- per level the setup's 12 B read + 12 B written per pixel at 8 TB/s and
            # maxiter iterations of the 8.2 us chain measured for a pair alone on the chip (profiles/r03_notes.md)
            "floor_ms": 3 * (24.0 * n * P * P / 8e12 + 10 * 8.2e-6) * 1e3,
            "frac": 3 * (24.0 * n * P * P / 8e12 + 10 * 8.2e-6) * 1e3 / cases[0]["default_ms"],
            "note": "latency regime: an iteration is a ~8-10 us chain (patches from registers, mailbox gather, solve, "
                    "broadcast), not a stream of T/Gx/Gy from HBM; frac = chain floor / measured (host calls included)"}


def issue_floor_ms(n_points, levels, maxiter, workgroups=1, waves=8, clock_ghz=2.4):
    """Floor model of a one-launch tracking (k_track1_p8), the figure `frac` of the latency records is taken against:
    a tracking is ONE dependency chain on one CU per workgroup, and a wave64 issues at most one instruction per 4 cycles
    (16-lane SIMD). Chain of a Gauss-Newton iteration = one wave's patches (ceil(points / waves) x ~40 instructions: two
    window loads, seven LDS reads, the blend, the residual, six multiply-add pairs) + the wave reduction (6 x 11) + the
    solver's turn (~330 dependent instructions: substitution with six correctly rounded divisions, pose update, exp
    map); a level's setup = ceil(points / waves) x ~75 instructions (three planes' windows, three blends, stores, 21
    multiply-adds) + 21 wave sums + the full-pivot LU (~420). No memory latency, no barrier, no exchange in the model:
    what the measured time exceeds it by is those (profiles/r03_notes.md has the measured phases)."""
    ppw = -(-max(1, -(-n_points // max(1, workgroups))) // waves)
    it = (ppw * 40 + 66 + 330) * 4
    setup = (ppw * 75 + 21 * 11 + 420) * 4
    return levels * (setup + maxiter * it) / (clock_ghz * 1e9) * 1e3


def rec_small(seconds, cpu_track=None):
    import invcompcamtrack_amd as ic
    out = []
    cases = ((640, 480, 100, 1), (640, 480, 300, 1), (640, 480, 1000, 1), (640, 480, 5000, 1), (640, 480, 300, 64))
    for (w, h, n, B) in cases:
        lv_f, P = 4, 8
        sc = _small_scene(w, h, n)
        op = ic.optparam(lv_f, 0, P, 10, 0.0, 0, 0, n)
        cam = ic.CamClass(lv_f + 1, sc["fc"], sc["cc"], sc["wh"], P)
        pa, pb = ic.Pyramid(sc["img_a"], lv_f, P), ic.Pyramid(sc["img_b"], lv_f, P)
        eng = ic.TrackBatch(cam, op, B)
        for k in range(B):
            eng.Set3Dpoints(k, sc["pts3d"].copy())

        p_all = np.tile(sc["p_a"], (B, 1))

        def step():
            if B > 1:  # one call for all problems of the frame pair, as the run_track_nposes driver does
                eng.SetPoseAll(p_all, pa, pb)
            else:
                eng.SetPose(0, sc["p_a"], pa, pb)
            eng.track_async()
            return eng.poses()

        ts = []
        step()
        t_end = time.perf_counter() + seconds / len(cases)
        while len(ts) < 5 or (time.perf_counter() < t_end and len(ts) < 200):
            t0 = time.perf_counter()
            p = step()
            ts.append(time.perf_counter() - t0)
        cpu = {}
        if cpu_track is not None:
            pc, m, runs = cpu_track(sc, lv_f, P, 10, 0.0, n, seconds=0.15)
            cpu = {"cpu_ms": m * B, "cpu_runs": runs, "pose_diff_vs_cpu": float(np.abs(p - pc[None, :]).max())}
        kname = eng.path_name() if hasattr(eng, "path_name") else ""
        wgs = int(kname.split(" x ")[1].split()[0]) if " workgroups per problem" in kname else 1
        floor = issue_floor_ms(n, lv_f + 1, 10, wgs) * max(1, -(-B * wgs // 256))  # problems beyond the CU count queue up
        out.append({**cpu, "points": n, "problems": B, "frame": f"{w}x{h}", "levels": lv_f + 1, "maxiter": 10,
                    "ms": float(np.median(ts)) * 1e3, "floor_ms": floor, "frac": floor / (float(np.median(ts)) * 1e3),
                    "kernel": eng.path_name() if hasattr(eng, "path_name") else "per-iteration launches",
                    "pose_err_vs_ground_truth": float(np.abs(p - sc["p_b"][None, :]).max())})
    return {"name": "small", "workload": "latency at the reference's own problem sizes (run_odometer_test.m): SetPose + "
            "TrackPose + poses on the host, 640x480, 5 levels x 10 iterations (normdp_ratio 0); ONE launch per tracking: "
            "one workgroup per problem up to 128 points, a team of workgroups with an in-launch all-gather above",
            "value": out[0]["ms"], "unit": "ms (100-point pair)", "cases": out,
            "algorithmic_bytes_per_launch": None, "frac": out[0]["frac"], "floor_ms": out[0]["floor_ms"],
            "cpu_ms": out[0].get("cpu_ms"),
            "note": "latency-bound: frac = issue-rate floor / measured (one instruction per wave per 4 cycles along the "
                    "tracking's dependency chain, no memory latency, barriers or exchanges; tools/secondary.py "
                    "issue_floor_ms) -- not a bandwidth fraction"}


def rec_pyramid(seconds):
    import torch
    import invcompcamtrack_amd as ic
    w, h, lv_f, pad, K = 1920, 1080, 2, 8, 16
    rng = np.random.default_rng(3)
    frame = torch.from_numpy(rng.integers(0, 256, (h, w)).astype(np.float32)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for gg in (1, 2):  # 1: image + dx + dy + packed texels (the reference's getgrad); 2: image levels only (r03)
        pyr = ic.Pyramid(lv_f=lv_f, imgpadding=pad, device_ptr=frame.data_ptr(), wh=(w, h), stream=st, getgrad=gg)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        times = []

        def step():  # K frames back to back (torch's current stream is the stream the kernels are launched on)
            ev0.record()
            for _ in range(K):
                pyr.rebuild(device_ptr=frame.data_ptr(), stream=st)
            ev1.record()
            ev1.synchronize()
            times.append(ev0.elapsed_time(ev1) * 1e-3 / K)

        _budget_loop(step, seconds / 2)
        px_out = sum(pyr.level_dims(l)[0] * pyr.level_dims(l)[1] for l in range(lv_f + 1))
        res[gg] = (float(np.median(times[1:] or times)), len(times), px_out)
        del pyr
    t, reps, px_out = res[1]
    t2 = res[2][0]
    px_in = sum((w >> l) * (h >> l) for l in range(lv_f + 1))
    alg = 4.0 * px_in + 28.0 * px_out  # read every level's source once; write image, dx, dy and the 16-B packed texel
    alg2 = 4.0 * px_in + 4.0 * px_out
    return {"name": "pyramid", "workload": f"util_constructpyramide of one {w}x{h} frame (device-resident f32), {lv_f + 1} "
            f"levels, padding {pad}, gradients + packed texels, refilled in place ({K} frames back to back per sample)",
            "value": 1.0 / t, "unit": "frames/s", "ms_per_step": t * 1e3, "reps": reps, "kernel": "k_pyr_level "
            "(one launch per level)", "kernel_us": t * 1e6, "algorithmic_bytes_per_launch": alg,
            "achieved_GBps": alg / t / 1e9, "frac": alg / t / 1e9 / PEAK,
            "image_only": {"what": "getgrad = 2: image levels only, the tracker's 8x8 setup kernel forms the gradient patches "
                                   "on the fly (bit-identical patches, tests/test_gpu_pyramid_patch.py)",
                           "kernel_us": t2 * 1e6, "speedup": t / t2, "algorithmic_bytes_per_launch": alg2,
                           "frac": alg2 / t2 / 1e9 / PEAK, "note": "three dependent ~4 us launches: launch-bound"},
            "note": "three dependent launches per frame; bit-exact against the oracle (tests/test_gpu_pyramid_patch.py)"}


def run_all(seconds=1.0, cpu_track=None):
    """cpu_track: bench.py's cpu_baseline leg (the oracle on one tracking -> pose, ms, runs); tools/ never imports the
    oracle itself. None: no CPU figures."""
    recs = []
    for fn in (rec_psz4, rec_c3, rec_c5, rec_c4, rec_nposes, rec_small, rec_dense, rec_pyramid):
        t0 = time.perf_counter()
        try:
            r = fn(seconds, cpu_track) if fn in (rec_nposes, rec_small, rec_dense) else fn(seconds)
        except Exception as exc:  # a secondary record must never take the headline down
            r = {"name": fn.__name__[4:], "error": repr(exc)}
        r["wall_s"] = round(time.perf_counter() - t0, 2)
        recs.append(r)
        import gc
        gc.collect()
    return recs


if __name__ == "__main__":
    import json
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for r in run_all(float(sys.argv[1]) if len(sys.argv) > 1 else 1.0):
        print(json.dumps(r), flush=True)
