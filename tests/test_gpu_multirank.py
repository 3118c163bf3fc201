"""Multi-rank entry points rehearsed on ONE GPU (two ranks, both on cuda:0, gloo for the host-side group): the code paths
a multi-GPU node runs, minus the link. What cannot be measured here -- any transport between different GPUs -- is said
so in DESIGN.md section 6; these tests pin behaviour, not speed."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "2", "--width", "640", "--height", "384",
                        "--steps", "2", "--warmup", "1", "--no-secondary", "--cpu-seconds", "0"] + extra,
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.timeout(900)
def test_bench_strong_scaling_mode_splits_one_point_set_over_the_ranks():
    """bench.py --gpus 2 --strong: ONE set of points per pair, split in contiguous blocks (the north star's partition);
    the sharded result must be the single-process tracker's (same points, sums in another order)."""
    one = _bench(["--gpus", "1", "--variant", str(1 << 21)])
    two = _bench(["--gpus", "2", "--strong", "--rehearse-gloo"])
    assert two["scaling"] == "strong" and two["n_gpus"] == 2 and one["scaling"] == "weak"
    n1 = one["config"]["points_per_pair_per_gpu"]
    assert two["config"]["points_per_pair_per_gpu"] * 2 == n1 == two["config"]["points_per_pair_whole_job"]
    # same frame pairs, same points: the poses agree far inside the 1e-4 bar (summation order differs)
    assert abs(two["pose_err_vs_ground_truth"] - one["pose_err_vs_ground_truth"]) < 2e-5
    assert two["value"] > 0 and two["unit"] == one["unit"]


@pytest.mark.timeout(900)
def test_row_band_alignment_launcher_two_ranks():
    """python -m invcompcamtrack_amd.run_align_sharded --gpus 2 (BASELINE config 5's form on a small frame): both ranks
    end with bit-identical warps, equal to the unsharded engine's within 1e-3 px at the frame corners."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "invcompcamtrack_amd.run_align_sharded", "--gpus", "2", "--config", "small",
                        "--steps", "2", "--rehearse-gloo"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["ranks_agree_bitwise"] is True
    assert out["corner_err_px_vs_unsharded"] < 1e-3 and out["corner_err_px_vs_ground_truth"] < 0.05
    assert out["row_band_of_rank0"] == [2, 2 + (256 - 4) // 2]


def _worker_patchflow(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import invcompcamtrack_amd as ic
    from invcompcamtrack_amd import patchflow as pf, synth
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    sc = synth.make_scene(320, 256, n_points=50, seed=4)
    pa, pb = ic.Pyramid(sc["img_a"], 2, 16), ic.Pyramid(sc["img_b"], 2, 16)
    rng = np.random.default_rng(3)
    pts = np.stack([rng.uniform(30, 290, 203), rng.uniform(30, 226, 203)], 1).astype(np.float32)
    res = pf.track_points(pa, pb, pts, psz=15, dist=dist)
    if rank == 0:
        solo = pf.track_points(pa, pb, pts, psz=15)
        q.put([bool(np.array_equal(a, b, equal_nan=True)) for a, b in zip(res, solo)] + [int(solo[1].sum())])
    else:
        q.put(res is None)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_patch_flow_ranges_over_two_ranks_equal_the_single_process_result():
    """BASELINE config 4's multi-GPU form with the real kernel: 203 patches as two contiguous ranges on two ranks (one
    GPU), gathered by rank 0 = the single-process launch bit for bit."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_patchflow, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    full = [r for r in res if isinstance(r, list)][0]
    assert full[:3] == [True, True, True] and full[3] > 150
    assert True in [r for r in res if not isinstance(r, list)]


@pytest.mark.timeout(900)
def test_bench_sharded_resident_form_two_ranks():
    """bench.py --gpus 2 --strong --resident-p2p (rehearsed on one GPU): every rank runs the headline's kernels --
    k_ref8 + ONE k_level_resident launch per level -- on its half of the points and the ranks' sums meet inside the
    launches. Same poses as the single-process tracker (2e-5), and the line says which path ran."""
    one = _bench(["--gpus", "1"])
    two = _bench(["--gpus", "2", "--strong", "--rehearse-p2p", "--resident-p2p"])
    assert two["scaling"] == "strong" and two["n_gpus"] == 2
    assert "RESIDENT" in two["config"]["collective"], two["config"]["collective"]
    assert abs(two["pose_err_vs_ground_truth"] - one["pose_err_vs_ground_truth"]) < 2e-5
