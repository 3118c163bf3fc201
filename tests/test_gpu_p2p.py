"""One-shot peer-to-peer all-reduce (csrc/ictr_p2p.hip, dist.P2PDirect) with several processes on ONE GPU: every rank
maps the others' mailboxes with hipIpc, stores its record into all of them and sums its own mailbox in rank order.
What one GPU can show: the IPC set-up, the granule protocol (tags, double buffering over hundreds of back-to-back
exchanges, ragged counts), bit-identical sums on every rank, the lock-step fallback verdicts, the time-out flag.
What it cannot: xGMI transport between different GPUs (no multi-GPU box in this pool)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, rounds, count):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from invcompcamtrack_amd.dist import P2PDirect
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    box = P2PDirect(torch, dist, None, count)
    if not box.ok:
        q.put((rank, "setup failed: " + box.why, None, None))
        dist.barrier()
        dist.destroy_process_group()
        return
    st = torch.cuda.current_stream().cuda_stream
    gen = np.random.default_rng(100 + rank)
    mism, last = 0, None
    for k in range(rounds):
        n = count if k % 3 else max(1, count // 2 + k % 7)   # ragged counts re-use the same mailboxes
        mine = gen.normal(size=n).astype(np.float32)
        d = torch.from_numpy(mine).cuda()
        box.all_reduce_sum_f32(d.data_ptr(), n, st)
        got = d.cpu().numpy()
        # expected: every rank's vector of this round, added in rank order in float32
        allv = [torch.zeros(n, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(allv, torch.from_numpy(mine))
        want = np.zeros(n, np.float32)
        for v in allv:
            want = (want + v.numpy()).astype(np.float32)
        mism += int(not np.array_equal(got, want))
        last = got
    err = box.error()
    box.close()
    q.put((rank, "ok", mism, (err, last.tobytes())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_p2p_allreduce_between_processes_on_one_gpu(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, 200, 27 * 32)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]
    assert all(r[2] == 0 for r in res), "some exchange did not equal the rank-ordered float32 sum"
    assert all(r[3][0] is False for r in res), "time-out flag raised"
    assert all(r[3][1] == res[0][3][1] for r in res), "ranks ended with different bits"


def test_p2p_world_of_one_and_timeout_flag():
    """A world of one needs no peer; a rank whose peer never arrives must give up (sticky error flag), not hang."""
    import ctypes as C
    import torch
    from invcompcamtrack_amd import _lib
    L = _lib.load()
    h = C.c_void_p()
    _lib.check(L.ictr_p2p_create(C.byref(h), 0, 1, 64))
    d = torch.arange(64, dtype=torch.float32, device="cuda")
    for _ in range(5):
        _lib.check(L.ictr_p2p_allreduce(h, C.c_void_p(d.data_ptr()), 64, C.c_void_p(0)))
    assert L.ictr_p2p_error(h) == 0 and torch.equal(d.cpu(), torch.arange(64, dtype=torch.float32))
    L.ictr_p2p_destroy(h)
    # world of two, the second mailbox is this process's own memory standing in for a peer that never writes
    os.environ["ICTR_P2P_TIMEOUT_S"] = "0.05"
    try:
        h2 = C.c_void_p()
        _lib.check(L.ictr_p2p_create(C.byref(h2), 0, 2, 64))
    finally:
        del os.environ["ICTR_P2P_TIMEOUT_S"]
    assert L.ictr_p2p_allreduce(h2, C.c_void_p(d.data_ptr()), 64, C.c_void_p(0)) != 0   # not connected yet: refused
    L.ictr_p2p_destroy(h2)


def _worker_tracker(rank, world, port, q):
    """The sharded Gauss-Newton loop (dist.ShardedTracker) with the P2P exchange as its collective: real kernels,
    real phase sequence, every rank on cuda:0."""
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import invcompcamtrack_amd as ic
    from invcompcamtrack_amd import synth
    from invcompcamtrack_amd.dist import ShardedTracker, shard_slices
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    lv_f, psz, B = 2, 8, 3
    scs = [synth.make_scene(256, 224, n_points=301, seed=40 + k, margin=12.0) for k in range(B)]
    cam = ic.CamClass(lv_f + 1, scs[0]["fc"], scs[0]["cc"], scs[0]["wh"], psz)
    lo, hi = shard_slices(301, world)[rank]
    op = ic.optparam(lv_f, 0, psz, 6, 0.01, 0, 0, max(hi - lo, 4))
    eng = ic.TrackBatch(cam, op, B)
    pyr = [(ic.Pyramid(sc["img_a"], lv_f, psz), ic.Pyramid(sc["img_b"], lv_f, psz)) for sc in scs]
    for k, sc in enumerate(scs):
        eng.Set3Dpoints(k, np.ascontiguousarray(sc["pts3d"][:, lo:hi]))
        eng.SetPose(k, sc["p_a"], *pyr[k])
    tr = ShardedTracker(eng, p2p=True)
    ok = tr.p2p is not None
    poses = None
    if ok:
        for _ in range(2):   # twice: the mailboxes' sequence numbers carry on across trackings
            for k, sc in enumerate(scs):
                eng.SetPose(k, sc["p_a"], *pyr[k])
            tr.track()
            poses = tr.poses()
    iters = eng.iterations() if ok else None
    tr.close()
    q.put((rank, ok, poses, iters))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_tracker_over_p2p_two_processes_one_gpu():
    import torch.multiprocessing as mp
    import invcompcamtrack_amd as ic
    from invcompcamtrack_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_tracker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1], "P2P path was not available on both ranks"
    assert np.array_equal(res[0][2], res[1][2])        # identical bits on both ranks: no broadcast needed
    assert np.array_equal(res[0][3], res[1][3])        # the early exit (normdp_ratio 0.01) stayed in lockstep
    # the unsharded engine on the same problems
    lv_f, psz, B = 2, 8, 3
    scs = [synth.make_scene(256, 224, n_points=301, seed=40 + k, margin=12.0) for k in range(B)]
    cam = ic.CamClass(lv_f + 1, scs[0]["fc"], scs[0]["cc"], scs[0]["wh"], psz)
    op = ic.optparam(lv_f, 0, psz, 6, 0.01, 0, 0, 301)
    eng = ic.TrackBatch(cam, op, B)
    keep = []
    for k, sc in enumerate(scs):
        pa, pb = ic.Pyramid(sc["img_a"], lv_f, psz), ic.Pyramid(sc["img_b"], lv_f, psz)
        keep.append((pa, pb))
        eng.Set3Dpoints(k, sc["pts3d"].copy())
        eng.SetPose(k, sc["p_a"], pa, pb)
    eng.track_async()
    ref = eng.poses()
    assert np.abs(res[0][2] - ref).max() <= 2e-5
    assert np.abs(res[0][2] - np.array([sc["p_b"] for sc in scs])).max() < 5e-3


def _worker_tracker_late_peer(rank, world, port, q):
    """Rank 1 enqueues its tracking half a second late under a 0.05 s exchange limit: rank 0's kernel waits in vain
    (time-out, local flag), rank 1 finds rank 0's records waiting and sees nothing wrong."""
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["ICTR_P2P_TIMEOUT_S"] = "0.05"
    import time
    import torch
    import torch.distributed as dist
    import invcompcamtrack_amd as ic
    from invcompcamtrack_amd import synth
    from invcompcamtrack_amd.dist import ShardedTracker, shard_slices
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    lv_f, psz = 1, 8
    sc = synth.make_scene(256, 224, n_points=200, seed=41, margin=12.0)
    cam = ic.CamClass(lv_f + 1, sc["fc"], sc["cc"], sc["wh"], psz)
    lo, hi = shard_slices(200, world)[rank]
    op = ic.optparam(lv_f, 0, psz, 3, 0.0, 0, 0, hi - lo)
    eng = ic.TrackBatch(cam, op, 1)
    pa, pb = ic.Pyramid(sc["img_a"], lv_f, psz), ic.Pyramid(sc["img_b"], lv_f, psz)
    eng.Set3Dpoints(0, np.ascontiguousarray(sc["pts3d"][:, lo:hi]))
    eng.SetPose(0, sc["p_a"], pa, pb)
    tr = ShardedTracker(eng, p2p=True)
    had_p2p = tr.p2p is not None
    local_flag, raised, msg = None, False, ""
    if had_p2p:
        dist.barrier()
        if rank == 1:
            time.sleep(0.5)
        tr.track()
        torch.cuda.synchronize()
        local_flag = any(c.error() for c in tr.p2p)
        try:
            tr.poses()
        except RuntimeError as exc:
            raised, msg = True, str(exc)
    q.put((rank, had_p2p, local_flag, raised, msg, tr.p2p is None))
    tr.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_p2p_timeout_on_one_rank_fails_the_tracking_on_every_rank():
    """a P2P time-out is local knowledge. The ranks agree on the verdict in poses(): BOTH raise, both drop
    the mailboxes -- also the rank that saw nothing wrong."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_tracker_late_peer, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1], "P2P path was not available on both ranks"
    assert res[0][2] is True                  # rank 0 waited in vain
    assert res[0][3] and res[1][3], res       # ... and BOTH ranks raise
    assert "timed out" in res[0][4] and "timed out" in res[1][4]
    assert res[0][5] and res[1][5]            # mailboxes closed on both


def _worker_resident_sharded(rank, world, port, q):
    """The sharded RESIDENT form (dist.ResidentShardedTracker): every rank's resident-iteration launches exchange H and
    b with the peer ranks inside the launch; both ranks on cuda:0."""
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["ICTR_RESIDENT_SLOTS"] = "2"   # two frame pairs in flight: the five pairs take three rounds per launch
    import torch
    import torch.distributed as dist
    import invcompcamtrack_amd as ic
    from invcompcamtrack_amd import synth
    from invcompcamtrack_amd.dist import ResidentShardedTracker, shard_slices
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    lv_f, psz, B, n = 2, 8, 5, 3001
    scs = [synth.make_scene(320, 256, n_points=n, seed=50 + k, margin=12.0) for k in range(B)]
    cam = ic.CamClass(lv_f + 1, scs[0]["fc"], scs[0]["cc"], scs[0]["wh"], psz)
    lo, hi = shard_slices(n, world)[rank]
    op = ic.optparam(lv_f, 0, psz, 6, 0.01, 0, 0, hi - lo)
    eng = ic.TrackBatch(cam, op, B)
    pyr = [(ic.Pyramid(sc["img_a"], lv_f, psz), ic.Pyramid(sc["img_b"], lv_f, psz)) for sc in scs]
    for k, sc in enumerate(scs):
        eng.Set3Dpoints(k, np.ascontiguousarray(sc["pts3d"][:, lo:hi]))
    tr = ResidentShardedTracker(eng)
    ok, poses, path = tr.ok, None, ""
    if ok:
        for _ in range(3):   # three trackings: the pairs' exchange counters carry on across launches
            for k, sc in enumerate(scs):
                eng.SetPose(k, sc["p_a"], *pyr[k])
            tr.track()
            poses = tr.poses()
        path = eng.path_name()
    iters = eng.iterations() if ok else None
    tr.close()
    q.put((rank, ok, poses, iters, path, tr.why))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_resident_form_two_processes_one_gpu():
    """Points of every frame pair split over two ranks; each rank's k_level_resident launches sum H and b over the ranks
    inside the launch (no collective, no kernel boundary between iterations). Both ranks must end with identical bits,
    identical iteration counts (early exit on: normdp_ratio 0.01), and the unsharded engine's poses to 2e-5. Five frame
    pairs with two in flight per launch (three rounds: a pair's exchange counter outlives its slot), three trackings."""
    import torch.multiprocessing as mp
    import invcompcamtrack_amd as ic
    from invcompcamtrack_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_resident_sharded, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1], ("exchange not available on both ranks", res[0][5], res[1][5])
    assert "k_level_resident" in res[0][4] and "k_level_resident" in res[1][4]
    assert np.array_equal(res[0][2], res[1][2])        # identical bits on both ranks
    assert np.array_equal(res[0][3], res[1][3])        # the early exit stayed in lockstep
    lv_f, psz, B, n = 2, 8, 5, 3001
    scs = [synth.make_scene(320, 256, n_points=n, seed=50 + k, margin=12.0) for k in range(B)]
    cam = ic.CamClass(lv_f + 1, scs[0]["fc"], scs[0]["cc"], scs[0]["wh"], psz)
    op = ic.optparam(lv_f, 0, psz, 6, 0.01, 0, 0, n)
    eng = ic.TrackBatch(cam, op, B)
    keep = []
    for k, sc in enumerate(scs):
        pa, pb = ic.Pyramid(sc["img_a"], lv_f, psz), ic.Pyramid(sc["img_b"], lv_f, psz)
        keep.append((pa, pb))
        eng.Set3Dpoints(k, sc["pts3d"].copy())
        eng.SetPose(k, sc["p_a"], pa, pb)
    eng.track_async()
    ref = eng.poses()
    assert np.abs(res[0][2] - ref).max() <= 2e-5, np.abs(res[0][2] - ref).max()
    assert np.array_equal(res[0][3], eng.iterations())
