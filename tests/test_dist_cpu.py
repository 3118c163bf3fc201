"""World-size-2 (gloo, CPU) test of the sharded Gauss-Newton loop: points split over ranks, H all-reduced once per
level and b once per iteration (invcompcamtrack_amd/dist.py). No GPU here, so the per-rank accumulate / finish steps
are played by a NumPy stand-in built on the oracle's element-wise pieces (test infrastructure); what is under test is
the partition, the collective sequence and that every rank ends with the same pose as the unsharded oracle."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class NumpyShardEngine:
    """begin / level_accumulate / level_finish / iter_accumulate / iter_finish on one shard of the points, with the
    reduction buffer laid out like the HIP engine's (27 floats: 21 H upper triangle + 6 b)."""

    def __init__(self, O, N, sc, lo, hi, lv_f, psz, maxiter):
        self.O, self.N = O, N
        self.pts = sc["pts3d"][:, lo:hi].astype(np.float32)
        self.pa, self.pb = O.Pyramid(sc["img_a"], lv_f, psz), O.Pyramid(sc["img_b"], lv_f, psz)
        op = O.make_op(lv_f, 0, psz, maxiter, 0.0, 0, 0, max(hi - lo, 4))
        self.cam = O.Tracker(op, sc["fc"], sc["cc"], sc["wh"])
        self.psz = psz
        self.p = sc["p_a"].astype(np.float32)
        self.red = np.zeros(27, np.float32)

    def begin(self):
        G0 = self.N.exp_se3(self.p)
        X, Y, Z = self.pts
        _, _, self.Xc, self.Yc, self.Zc = self.N.project(G0, X, Y, Z, np.float32(1), np.float32(1), np.float32(0), np.float32(0))
        self.G0 = G0

    def _cam(self, sl):
        return [np.float32(self.cam.cam_get(k, sl)) for k in range(4)]

    def level_accumulate(self, sl):
        N, P = self.N, self.psz
        fx, fy, cx, cy = self._cam(sl)
        X, Y, Z = self.pts
        mx, my, _, _, _ = N.project(self.G0, X, Y, Z, fx, fy, cx, cy)
        self.T = N.patches(self.pa.img[sl], mx, my, P)
        Gx, Gy = N.patches(self.pa.dx[sl], mx, my, P), N.patches(self.pa.dy[sl], mx, my, P)
        cxk, cyk = N.sd_coefs(self.Xc, self.Yc, self.Zc, fx, fy)
        sd = Gx[:, None] * cxk[:, :, None, None] + Gy[:, None] * cyk[:, :, None, None]
        sd[:, 0], sd[:, 1] = Gx * cxk[:, 0, None, None], Gy * cyk[:, 1, None, None]
        self.sd = sd
        sdf = sd.reshape(len(X), 6, -1).astype(np.float64)
        H = np.einsum("kip,kjp->ij", sdf, sdf) if len(X) else np.zeros((6, 6))
        self.red[:21] = H[np.triu_indices(6)].astype(np.float32)

    def level_finish(self, sl):
        H = np.zeros((6, 6), np.float32)
        H[np.triu_indices(6)] = self.red[:21]
        self.H = H + np.triu(H, 1).T
        self.red[:] = 0

    def iter_accumulate(self, sl):
        N = self.N
        fx, fy, cx, cy = self._cam(sl)
        X, Y, Z = self.pts
        nx, ny, _, _, _ = N.project(N.exp_se3(self.p), X, Y, Z, fx, fy, cx, cy)
        r = self.T - N.patches(self.pb.img[sl], nx, ny, self.psz)
        b = (self.sd * r[:, None]).reshape(len(X), 6, -1).astype(np.float64).sum((0, 2)) if len(X) else np.zeros(6)
        self.red[21:] = b.astype(np.float32)

    def iter_finish(self, sl):
        self.p = self.p + self.O.solve6(self.H, self.red[21:].copy())
        self.red[21:] = 0


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from invcompcamtrack_amd import synth
    from invcompcamtrack_amd.dist import run_sharded_levels, shard_slices
    from oracle import np_oracle as N
    from oracle import oracle as O
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    sc = synth.make_scene(320, 240, n_points=61, seed=12, margin=40.0)  # odd count: unbalanced shards
    lo, hi = shard_slices(61, world)[rank]
    lv_f, psz, maxiter = 2, 8, 4
    eng = NumpyShardEngine(O, N, sc, lo, hi, lv_f, psz, maxiter)
    calls = []

    def allreduce():
        t = torch.from_numpy(eng.red)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)  # in place on the engine's buffer, like the RCCL path
        calls.append(1)

    op = O.make_op(lv_f, 0, psz, maxiter, 0.0, 0, 0, 64)
    run_sharded_levels(eng, op, allreduce)
    q.put((rank, eng.p.copy(), len(calls), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def _worker_interleaved(rank, world, port, q):
    """Two groups (two different scenes) software-pipelined against each other's ASYNCHRONOUS all-reduces."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from invcompcamtrack_amd import synth
    from invcompcamtrack_amd.dist import run_interleaved, shard_slices, sharded_program
    from oracle import np_oracle as N
    from oracle import oracle as O
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lv_f, psz, maxiter = 1, 8, 3
    engs, order = [], []
    for g, seed in enumerate((12, 13)):
        sc = synth.make_scene(256, 192, n_points=45, seed=seed, margin=40.0)
        lo, hi = shard_slices(45, world)[rank]
        engs.append(NumpyShardEngine(O, N, sc, lo, hi, lv_f, psz, maxiter))
    op = O.make_op(lv_f, 0, psz, maxiter, 0.0, 0, 0, 64)

    def ar(g):
        order.append(("start", g))
        t = torch.from_numpy(engs[g].red)
        w = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)

        class H:
            def wait(self_inner):
                order.append(("wait", g))
                w.wait()
        return H()

    run_interleaved([sharded_program(e, op, (lambda g=g: ar(g))) for g, e in enumerate(engs)])
    q.put((rank, [e.p.copy() for e in engs], order))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_interleaved_groups_overlap_collectives_gloo(oracle):
    import torch.multiprocessing as mp
    from invcompcamtrack_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_interleaved, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, pa, order0), (_, pb, order1) = res
    assert order0 == order1
    # the schedule: group 1's collective is started before group 0's is waited for, every time
    assert order0[:4] == [("start", 0), ("start", 1), ("wait", 0), ("start", 0)]
    assert len(order0) == 2 * 2 * 2 * (1 + 3)          # start+wait x groups x levels x (1 + maxiter)
    for g, seed in enumerate((12, 13)):
        assert np.array_equal(pa[g], pb[g])
        sc = synth.make_scene(256, 192, n_points=45, seed=seed, margin=40.0)
        op = oracle.make_op(1, 0, 8, 3, 0.0, 0, 0, 45)
        tr = oracle.Tracker(op, sc["fc"], sc["cc"], sc["wh"])
        tr.set3dpoints(sc["pts3d"].copy())
        tr.setpose(sc["p_a"], oracle.Pyramid(sc["img_a"], 1, 8), oracle.Pyramid(sc["img_b"], 1, 8))
        tr.trackpose()
        assert np.allclose(pa[g], tr.pose_p(), atol=2e-5)


@pytest.mark.timeout(300)
def test_sharded_loop_two_ranks_gloo(oracle):
    import torch.multiprocessing as mp
    from invcompcamtrack_amd import synth
    from invcompcamtrack_amd.dist import shard_slices
    assert shard_slices(61, 2) == [(0, 31), (31, 61)] and shard_slices(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, p0, n0, s0), (r1, p1, n1, s1) = res
    assert np.array_equal(p0, p1)                      # identical bits on both ranks: no broadcast needed
    assert n0 == n1 == 3 * (1 + 4)                     # one all-reduce per level + one per iteration
    assert s0 == (0, 31) and s1 == (31, 61)
    # unsharded oracle on the same inputs
    sc = synth.make_scene(320, 240, n_points=61, seed=12, margin=40.0)
    op = oracle.make_op(2, 0, 8, 4, 0.0, 0, 0, 61)
    tr = oracle.Tracker(op, sc["fc"], sc["cc"], sc["wh"])
    tr.set3dpoints(sc["pts3d"].copy())
    tr.setpose(sc["p_a"], oracle.Pyramid(sc["img_a"], 2, 8), oracle.Pyramid(sc["img_b"], 2, 8))
    tr.trackpose()
    assert np.allclose(p0, tr.pose_p(), atol=2e-5)


def test_global_norm_matches_single_rank_formula():
    """dist.global_norm == the reference's mean / mean squared radius (odometer.cpp:193-214) over all ranks' points."""
    from invcompcamtrack_amd.dist import global_norm

    class FakeDist:  # two "ranks" summed by hand
        def __init__(self, other):
            self.other = other

        def all_reduce(self, t, group=None):
            t += self.other.pop(0)

        def get_backend(self, group=None):
            return "gloo"

    import torch
    rng = np.random.default_rng(0)
    a, b = rng.normal(size=(3, 10)) + 5, rng.normal(size=(3, 7)) - 2
    allp = np.concatenate([a, b], 1)
    mean = allp.mean(1)
    other = [torch.tensor([b[0].sum(), b[1].sum(), b[2].sum(), 7.0], dtype=torch.float64),
             torch.tensor([float(((b - mean[:, None]) ** 2).sum())], dtype=torch.float64)]
    m, v = global_norm(a, dist=FakeDist(other))
    assert np.allclose(m, mean) and np.isclose(v, ((allp - mean[:, None]) ** 2).sum(0).mean())


def _worker_norm(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from invcompcamtrack_amd.dist import global_norm, shard_slices, sharded_set3dpoints
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    allp = rng.normal(size=(3, 37)) * np.array([[3.0], [1.0], [0.5]]) + np.array([[4.0], [-2.0], [9.0]])
    lo, hi = shard_slices(37, world)[rank]
    mine = np.ascontiguousarray(allp[:, lo:hi])
    mean, var = global_norm(mine)                      # through the real process group (CPU tensors for gloo)

    class Op:
        donorm = True

    class FakeBatch:                                   # records what the engine would be handed
        op = Op()

        def Set3Dpoints_norm(self, problem, pts, ms, vv):
            self.call = (problem, pts.copy(), np.array(ms), vv)
            pts -= np.asarray(ms)[:, None]             # the engine normalises in place (odometer.cpp:207-212)
            pts /= vv

        def Set3Dpoints(self, problem, pts):
            self.call = ("plain", problem)

    fb = FakeBatch()
    m2, v2 = sharded_set3dpoints(fb, 3, mine)
    q.put((rank, mean, var, m2, v2, fb.call[0], mine.copy(), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_global_norm_and_sharded_set3dpoints_two_ranks_gloo():
    """donorm with sharded points: mean / mean squared radius must be those of the UNION of the shards
    (odometer.cpp:193-214), obtained through torch.distributed itself (CPU tensors under gloo, GPU tensors
    under nccl), and handed to the engine by dist.sharded_set3dpoints."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_norm, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rng = np.random.default_rng(5)
    allp = rng.normal(size=(3, 37)) * np.array([[3.0], [1.0], [0.5]]) + np.array([[4.0], [-2.0], [9.0]])
    mean = allp.mean(1)
    var = ((allp - mean[:, None]) ** 2).sum(0).mean()
    for rank, m, v, m2, v2, prob, pts_after, (lo, hi) in res:
        assert np.allclose(m, mean, rtol=1e-14) and np.isclose(v, var, rtol=1e-14)
        assert np.array_equal(m, m2) and v == v2 and prob == 3
        assert np.allclose(pts_after, (allp[:, lo:hi] - mean[:, None]) / var)   # normalised in place, global statistics
    assert np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]     # identical on both ranks


def _worker_patchflow(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch.distributed as dist
    from invcompcamtrack_amd import patchflow as pf
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    pts = np.stack([np.arange(11, dtype=np.float32), 100 + np.arange(11, dtype=np.float32)], 1)
    seen = []

    def fake_local(qpts):   # stands in for the GPU call: results that identify patch and rank
        seen.append(qpts.copy())
        return (qpts + np.float32(0.5), qpts[:, 0] % 2 == 0, np.full(len(qpts), 10 + rank, np.int32))
    res = pf.track_points(None, None, pts, dist=dist, _local=fake_local)
    q.put((rank, None if res is None else [np.asarray(x).tolist() for x in res], [s.tolist() for s in seen]))
    dist.barrier()
    dist.destroy_process_group()


def test_patchflow_patch_ranges_over_two_ranks_gathered_by_rank0():
    """BASELINE config 4's multi-GPU form (patchflow.track_points(..., dist=...)): contiguous patch ranges per rank,
    no data-path collective, rank 0 assembles the full result in patch order; world 2 over gloo, GPU call replaced."""
    import torch.multiprocessing as mp
    from invcompcamtrack_amd import patchflow as pf
    assert pf.partition_patches(11, 2) == [(0, 6), (6, 11)] and pf.partition_patches(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_patchflow, args=(r, 2, port, q)) for r in range(2)]
    for p_ in ps:
        p_.start()
    got = dict()
    for _ in range(2):
        r, res, seen = q.get(timeout=120)
        got[r] = (res, seen)
    for p_ in ps:
        p_.join(60)
        assert p_.exitcode == 0
    assert got[1][0] is None
    new, st, it = (np.asarray(x) for x in got[0][0])
    assert np.array_equal(new[:, 0], np.arange(11) + 0.5) and np.array_equal(new[:, 1], 100.5 + np.arange(11))
    assert np.array_equal(st, np.arange(11) % 2 == 0)
    assert np.array_equal(it, [10] * 6 + [11] * 5)
    assert np.asarray(got[0][1][0]).shape == (6, 2) and np.asarray(got[1][1][0]).shape == (5, 2)
