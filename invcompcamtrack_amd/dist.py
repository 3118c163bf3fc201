"""Multi-GPU tracking: one process per GPU, points sharded over ranks, RCCL all-reduce of the normal equations.

The reference is single-threaded (no threads, no collectives), so this layer is new. The only
coupling between points is the sum into one 6x6 / 6x1 system -- ComputeHessian (odometer.cpp:428-455) and
step 9a (odometer.cpp:399-404) -- so every rank owns a contiguous block of points (SoA slices), frames are
replicated, and per problem the ranks exchange
    21 floats of H once per pyramid level,   6 floats of b once per Gauss-Newton iteration
(on the 8x8 fast path the H of a level is accumulated by its first iteration launch and travels in the same
27-float message as that iteration's b: one collective per iteration, none per level)
with ``torch.distributed.all_reduce(SUM)`` (backend "nccl" = RCCL over xGMI). Every rank then runs the same
solve / pose update on identical bits, so no broadcast is needed and the early-exit test stays in lockstep.
Nothing is read back to the host inside the loop: the accumulate kernels, the collective and the finish
kernels are all enqueued on the current torch stream.

For tests without several GPUs, ``staged=True`` moves the 27*B floats through host memory and reduces with
any backend (gloo); the kernels in between are the real HIP kernels.
"""
from __future__ import annotations

import numpy as np

RED_STRIDE = 27  # per problem: 21 (upper triangle of H, row-major) + 6 (b); kRedStride in csrc/ictr_dev.h


def shard_slices(n_points: int, world: int):
    """Contiguous, balanced point blocks: rank r owns [lo, hi). Empty shards are allowed."""
    base, rem = divmod(n_points, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def _coll_device(dist, group=None):
    """Device a tensor must live on to go through this group's collectives: RCCL ("nccl") only takes GPU tensors."""
    try:
        return "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    except Exception:
        return "cpu"


def global_norm(pts_local, group=None, dist=None):
    """Mean and mean squared radius (odometer.cpp:193-214) of the union of all ranks' points.
    pts_local: (3, n_local) float64. Two tiny all-reduces (f64), on the backend's device."""
    import torch
    if dist is None:
        import torch.distributed as dist
    dev = _coll_device(dist, group)
    acc = torch.tensor([pts_local[0].sum(), pts_local[1].sum(), pts_local[2].sum(), float(pts_local.shape[1])],
                       dtype=torch.float64, device=dev)
    dist.all_reduce(acc, group=group)
    acc = acc.cpu()
    mean = (acc[:3] / acc[3]).numpy().copy()
    var = torch.tensor([float(((pts_local - mean[:, None]) ** 2).sum())], dtype=torch.float64, device=dev)
    dist.all_reduce(var, group=group)
    return mean, float(var.cpu()[0] / acc[3])


def sharded_set3dpoints(batch, problem, pts_local, group=None, dist=None):
    """Set3Dpoints for a rank's shard of a problem's points when ``donorm`` is on: the cloud normalisation of
    odometer.cpp:193-214 must use the mean / mean squared radius of ALL ranks' points, so they are all-reduced first
    (global_norm) and handed to the engine (TrackBatch.Set3Dpoints_norm). pts_local: (3, n_local) float64,
    C-contiguous, normalised in place like the reference does with its caller's array. Returns (mean, varval)."""
    if not batch.op.donorm:
        batch.Set3Dpoints(problem, pts_local)
        return np.zeros(3), 0.0
    mean, var = global_norm(pts_local, group=group, dist=dist)
    batch.Set3Dpoints_norm(problem, pts_local, mean, var)
    return mean, var


def run_sharded_levels(engine, op, allreduce):
    """The coarse-to-fine loop of odometer.cpp:261-420 in its sharded form. ``engine`` exposes
    begin / level_accumulate / level_finish / iter_accumulate / iter_finish (a TrackBatch, or a test double);
    ``allreduce()`` sums the engine's reduction buffer over all ranks."""
    engine.begin()
    level_ar = getattr(engine, "needs_level_allreduce", True)
    for sl in range(op.lv_f, op.lv_l - 1, -1):
        engine.level_accumulate(sl)   # steps 4-6 on the local points -> local H (8x8 fast path: H comes with the
        if level_ar:                  # first iteration's b instead, 27 floats in one message, no level collective)
            allreduce()
        engine.level_finish(sl)       # adopt the global H (if it is there yet), reset the iteration state
        for _ in range(op.maxiter):   # converged problems skip their work on the device (same decision on every rank)
            engine.iter_accumulate(sl)  # steps 7-9a on the local points -> local b
            allreduce()
            engine.iter_finish(sl)      # steps 9b-10, identical on every rank


def sharded_program(engine, op, allreduce_async):
    """The same loop as a generator that yields right after every collective has been STARTED and waits for it
    when it is resumed. Several such programs (one per group of problems) run round-robin
    (``run_interleaved``): while group A's 27 floats per problem travel, group B's accumulate kernel runs, so the
    collective latency is hidden behind compute instead of adding to every Gauss-Newton iteration.
    ``allreduce_async()`` starts the reduction of the engine's buffer and returns an object with ``wait()``."""
    engine.begin()
    level_ar = getattr(engine, "needs_level_allreduce", True)
    for sl in range(op.lv_f, op.lv_l - 1, -1):
        engine.level_accumulate(sl)
        if level_ar:
            h = allreduce_async()
            yield
            h.wait()
        engine.level_finish(sl)
        for _ in range(op.maxiter):
            engine.iter_accumulate(sl)
            h = allreduce_async()
            yield
            h.wait()
            engine.iter_finish(sl)


def run_interleaved(programs):
    """Round-robin over generator programs until all are exhausted."""
    active = list(programs)
    while active:
        for p in list(active):
            try:
                next(p)
            except StopIteration:
                active.remove(p)


class _Done:
    def wait(self):
        return None


def _agree(torch, dist, group, ok):
    """MIN all-reduce of a success flag: every rank learns whether EVERY rank succeeded. All ranks must call it at
    the same point whatever happened to them locally, so that the group's collectives stay matched."""
    flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32, device=_coll_device(dist, group))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return float(flag.cpu()[0]) >= 1.0


class RcclDirect:
    """RCCL called directly (ctypes on the librccl.so PyTorch ships) with its own communicator, so that the all-reduce
    is enqueued ON THE COMPUTE STREAM between the tail and the finish kernels. torch.distributed runs collectives on
    its own stream and synchronises with events in both directions; measured on MI355X that costs ~24 us of idle
    compute stream per collective even when there is nothing to wait for (profiles/r01_notes.md) -- per Gauss-Newton
    iteration. Opt-in (ShardedTracker(direct=True) / ICTR_RCCL_DIRECT=1): it has only ever run with a world of one.

    Setting up is a fixed sequence of collectives on the torch group that EVERY rank executes whatever fails locally
    (a rank that raised early would leave the others blocked in a broadcast):
      1. rank 0 broadcasts 129 bytes: a status byte + the unique id (status 0: it could not create one); all ranks
         agree (MIN all-reduce) that each of them holds the id and has the library;
      2. every rank calls ncclCommInitRank, then all ranks agree;
      3. every rank runs the self-test (sum of ones == world), then all ranks agree again.
    After any disagreement every rank destroys what it built and ``ok`` is False on every rank."""

    def __init__(self, torch, dist, group=None):
        import ctypes as C
        import os
        self._C, self._lib, self._comm, self.ok, self.why = C, None, None, False, ""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        self.world = world
        dev = _coll_device(dist, group)
        UniqueId = None
        local_ok = True
        try:
            path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            L = self._lib = C.CDLL(path)

            class UniqueId(C.Structure):
                _fields_ = [("internal", C.c_char * 128)]

            L.ncclGetUniqueId.argtypes, L.ncclGetUniqueId.restype = [C.POINTER(UniqueId)], C.c_int
            L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
            L.ncclCommInitRank.restype = C.c_int
            L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
            L.ncclAllReduce.restype = C.c_int
            L.ncclCommDestroy.argtypes, L.ncclCommDestroy.restype = [C.c_void_p], C.c_int
        except Exception as exc:  # the library is missing on this rank: still walk through the collectives below
            local_ok, self.why = False, f"librccl.so: {exc!r}"
        # 1. status byte + unique id, always broadcast by rank 0
        msg = bytearray(129)
        uid = UniqueId() if UniqueId is not None else None
        if rank == 0 and local_ok:
            if self._lib.ncclGetUniqueId(C.byref(uid)) == 0:
                msg[0] = 1
                msg[1:] = C.string_at(C.addressof(uid), 128)  # raw bytes (a c_char array read as bytes stops at a NUL)
            else:
                self.why = "ncclGetUniqueId failed"
        t = torch.tensor(list(msg), dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = bytes(t.cpu().tolist())
        have_uid = raw[0] == 1
        # ncclCommInitRank is itself a rendezvous of all ranks: only enter it when every rank can
        if not _agree(torch, dist, group, local_ok and have_uid):
            self.why = self.why or ("rank 0 has no unique id" if not have_uid else "another rank cannot load RCCL")
            return
        # 2. communicator
        if local_ok and have_uid:
            try:
                C.memmove(C.addressof(uid), raw[1:], 128)
                comm = C.c_void_p()
                rc = self._lib.ncclCommInitRank(C.byref(comm), world, uid, rank)
                if rc != 0 or not comm:
                    raise RuntimeError(f"ncclCommInitRank failed ({rc})")
                self._comm = comm
            except Exception as exc:
                local_ok, self.why = False, repr(exc)
        if not _agree(torch, dist, group, local_ok and have_uid):
            self.why = self.why or "another rank could not set up its communicator"
            self.close()
            return
        # 3. self-test before anything relies on it: the sum of ones over the ranks must be the world size
        try:
            probe = torch.ones(4, dtype=torch.float32, device="cuda")
            self.all_reduce_sum_f32(probe.data_ptr(), 4, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            if not bool((probe == float(world)).all().item()):
                raise RuntimeError(f"self-test: {probe.tolist()} != {world}")
        except Exception as exc:
            local_ok, self.why = False, repr(exc)
        if not _agree(torch, dist, group, local_ok):
            self.why = self.why or "another rank failed the self-test"
            self.close()
            return
        self.ok = True

    def all_reduce_sum_f32(self, dev_ptr, count, stream):
        rc = self._lib.ncclAllReduce(dev_ptr, dev_ptr, count, 7, 0, self._comm, stream)  # ncclFloat32, ncclSum
        if rc != 0:
            raise RuntimeError(f"ncclAllReduce failed ({rc})")

    def close(self):
        if getattr(self, "_comm", None):
            try:
                self._lib.ncclCommDestroy(self._comm)
            finally:
                self._comm = None
        self.ok = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class P2PDirect:
    """One-shot peer-to-peer all-reduce (csrc/ictr_p2p.hip): every rank writes its records into a mailbox slot in every
    peer's memory (hipIpc-mapped) and sums its own mailbox in rank order -- one hop over the point-to-point xGMI links
    instead of a ring's 2(N-1), one small kernel on the compute stream, identical bits on every rank. The mailbox
    handles travel through the existing torch.distributed group. Like RcclDirect the set-up is a fixed sequence of
    collectives that every rank executes whatever fails locally, and ends with the same verdict (``ok``) everywhere:
    create + all_gather of the handles -> agree -> connect (hipIpcOpenMemHandle) -> agree -> self-test -> agree."""

    def __init__(self, torch, dist, group, count):
        import ctypes as C
        from . import _lib
        self._C, self._h, self.ok, self.why = C, None, False, ""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        self.world, self.count = world, int(count)
        dev = _coll_device(dist, group)
        local_ok = True
        hb = 64
        try:
            self._L = _lib.load()
            hb = self._L.ictr_p2p_handle_bytes()
            h = C.c_void_p()
            _lib.check(self._L.ictr_p2p_create(C.byref(h), rank, world, self.count))
            self._h = h
            mine = (C.c_char * hb)()
            _lib.check(self._L.ictr_p2p_local_handle(self._h, C.cast(mine, C.c_void_p)))
            mine = bytes(mine)
        except Exception as exc:
            local_ok, self.why, mine = False, repr(exc), bytes(hb)
        t = torch.tensor(list(mine), dtype=torch.uint8, device=dev)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t, group=group)
        if not _agree(torch, dist, group, local_ok):
            self.why = self.why or "another rank could not create its mailbox"
            self.close()
            return
        try:
            allh = b"".join(bytes(g.cpu().tolist()) for g in gathered)
            buf = C.create_string_buffer(allh, len(allh))
            _lib.check(self._L.ictr_p2p_connect(self._h, C.cast(buf, C.c_void_p)))
        except Exception as exc:
            local_ok, self.why = False, repr(exc)
        if not _agree(torch, dist, group, local_ok):
            self.why = self.why or "another rank could not map the peers' mailboxes"
            self.close()
            return
        try:  # self-test: the sum of (rank + 1) over the ranks
            probe = torch.full((4,), float(rank + 1), dtype=torch.float32, device="cuda")
            self.all_reduce_sum_f32(probe.data_ptr(), 4, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            want = float(world * (world + 1) // 2)
            if self._L.ictr_p2p_error(self._h) or not bool((probe == want).all().item()):
                raise RuntimeError(f"self-test: {probe.tolist()} != {want}")
        except Exception as exc:
            local_ok, self.why = False, repr(exc)
        if not _agree(torch, dist, group, local_ok):
            self.why = self.why or "another rank failed the self-test"
            self.close()
            return
        self.ok = True

    def all_reduce_sum_f32(self, dev_ptr, count, stream):
        from . import _lib
        _lib.check(self._L.ictr_p2p_allreduce(self._h, self._C.c_void_p(dev_ptr), int(count), self._C.c_void_p(stream)))

    def error(self):
        return bool(self._L.ictr_p2p_error(self._h)) if self._h else True

    def close(self):
        if getattr(self, "_h", None):
            self._L.ictr_p2p_destroy(self._h)
            self._h = None
        self.ok = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _OnStream:
    """A torch.distributed work handle whose wait() holds up the given stream instead of the current one."""

    def __init__(self, torch, stream, work):
        self._torch, self._stream, self._work = torch, stream, work

    def wait(self):
        with self._torch.cuda.stream(self._stream):
            self._work.wait()


class ResidentShardedTracker:
    """Sharded RESIDENT form: every rank holds a contiguous block of each problem's points (frames replicated) and runs
    the resident-iteration launches of its own batch -- one setup launch + ONE launch per level with all iterations
    inside -- whose solver workgroups add H (once per level) and b (once per iteration) over the ranks THEMSELVES: each
    writes its sums into every rank's mailbox (hipIpc-mapped, one hop over the point-to-point xGMI links) and polls its
    own, inside the launch (csrc/ictr_resident.hip res_xchg_sum; csrc/ictr_p2p.hip's protocol). Between two iterations
    there is no kernel boundary, no host call and no communicator -- the multi-GPU form of the single-GPU headline
    kernel, where ShardedTracker drives per-iteration phase kernels and a collective from the host.

    ``batch``: a TrackBatch with this rank's shard set (Set3Dpoints) -- NOT in the phase-driven sharded mode. The
    mailboxes are a P2PDirect object (lock-step set-up with self-test over the torch group; ``ok`` is the same on every
    rank). Like P2PDirect it has never crossed a real link (one GPU per box in the build pool; two processes on one GPU
    in tests/test_gpu_p2p.py), hence opt-in: ``bench.py --resident-p2p``."""

    def __init__(self, batch, group=None, dist=None):
        import torch
        if dist is None:
            import torch.distributed as dist
        self._torch, self._dist, self.group, self.batch = torch, dist, group, batch
        self.box = P2PDirect(torch, dist, group, 64 * batch.B)
        self.ok, self.why = self.box.ok, self.box.why
        local_ok = True
        if self.ok:
            try:
                batch.set_peer_exchange(self.box._h)
            except Exception as exc:
                local_ok, self.why = False, repr(exc)
            if not _agree(torch, dist, group, local_ok):
                self.close()
                self.why = self.why or "another rank could not attach the exchange to its batch"

    def track(self):
        """Enqueue the whole tracking (every level's setup + resident launch); returns without synchronising."""
        if not self.ok:
            raise RuntimeError("sharded resident form: the exchange is not available (" + (self.why or "closed") + "); a "
                               "shard tracked on its own would give a silently wrong pose -- build a new tracker")
        self.batch.track_async()

    def poses(self):
        """All poses (identical on every rank). A rank whose launch ran into an exchange time-out raises -- and so does
        every other rank (one MIN all-reduce on the torch group, called by all ranks whatever they saw)."""
        out, ok = None, True
        try:
            out = self.batch.poses()
        except Exception as exc:
            ok, self.why = False, repr(exc)
        if not _agree(self._torch, self._dist, self.group, ok):
            # the ranks' exchange counters may have parted: the mailboxes are closed on EVERY rank (track() refuses from
            # now on; a new tracker starts from fresh mailboxes and counters)
            why = self.why
            self.close()
            self.why = why or "a peer rank's exchange timed out"
            raise RuntimeError("sharded resident form: an in-launch exchange timed out on " +
                               ("this rank" if not ok else "a peer rank") + "; the results of this tracking are invalid "
                               "on EVERY rank" + (f" ({why})" if not ok else ""))
        return out

    def close(self):
        if getattr(self, "box", None) is not None:
            try:
                self._torch.cuda.synchronize()
                self.batch.set_peer_exchange(None)
            except Exception:
                pass
            self.box.close()
            self.box = None
        self.ok = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedTracker:
    """Drives one TrackBatch, or several (groups of problems that are software-pipelined against each other's
    collectives), in sharded mode. Collective path: RcclDirect (in-stream, when asked for and every rank can set it up),
    else torch.distributed (asynchronous on the process group's stream; ``wait()`` makes the compute stream wait,
    not the host), else -- ``staged`` -- through host memory with any backend (tests). Default: torch.distributed;
    RcclDirect is opt-in (``direct=True`` or ICTR_RCCL_DIRECT=1)."""

    def __init__(self, batch, group=None, staged=False, direct=None, p2p=False):
        import os
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.batches = list(batch) if isinstance(batch, (list, tuple)) else [batch]
        self.batch = self.batches[0]
        self.group, self.staged = group, staged
        self.reds, self._hosts = [], []
        if direct is None:  # opt-in: the in-stream RCCL path has never run with more than one rank
            direct = os.environ.get("ICTR_RCCL_DIRECT") == "1"
        want_direct = (bool(direct) and not staged and dist.get_backend(group) == "nccl"
                       and not os.environ.get("ICTR_NO_RCCL_DIRECT"))
        # several groups: every group gets its own HIP stream, so that one group's collective (pure link latency), its
        # latency-bound setup kernel and its launch gaps overlap the other group's HBM-bound iteration kernel -- the
        # same effect as the two concurrent engines of the single-GPU mode (+8 %). The in-stream paths (P2P, direct
        # RCCL) additionally get their own mailbox / communicator per group.
        self._torch_streams = [torch.cuda.Stream() if (len(self.batches) > 1 and not staged) else None
                               for _ in self.batches]
        self._streams = []
        for b, ts in zip(self.batches, self._torch_streams):
            b.enable_sharding(True)
            red = torch.zeros(b.B * RED_STRIDE, dtype=torch.float32, device="cuda")
            b.set_reduction_buffer(red.data_ptr())
            handle = (ts or torch.cuda.current_stream()).cuda_stream
            b.set_stream(handle)
            self._streams.append(handle)
            self.reds.append(red)
            self._hosts.append(torch.zeros(b.B * RED_STRIDE, dtype=torch.float32) if staged else None)
        self.red = self.reds[0]
        torch.cuda.synchronize()  # the zero-fills above ran on the current stream
        self.direct, self.p2p = None, None
        if p2p and not staged:
            boxes = [P2PDirect(torch, dist, group, b.B * RED_STRIDE) for b in self.batches]
            if all(c.ok for c in boxes):
                self.p2p = boxes
            else:
                import sys
                print("[ictr.dist] P2P exchange unavailable (" + "; ".join(c.why for c in boxes if c.why) +
                      "); using torch.distributed", file=sys.stderr)
                for c in boxes:
                    c.close()
        if want_direct and self.p2p is None:
            # every rank builds the same number of communicators in the same order; each construction is itself a
            # lock-step sequence of collectives that ends with the same verdict on every rank (RcclDirect)
            comms = [RcclDirect(torch, dist, group) for _ in self.batches]
            if all(c.ok for c in comms):
                self.direct = comms
            else:
                import sys
                print("[ictr.dist] direct RCCL unavailable (" + "; ".join(c.why for c in comms if c.why) +
                      "); using torch.distributed", file=sys.stderr)
                for c in comms:
                    c.close()

    def close(self):
        """Destroy the direct communicators (if any); the tracker falls back to torch.distributed afterwards."""
        if self.direct is not None:
            self._torch.cuda.synchronize()
            for c in self.direct:
                c.close()
            self.direct = None
        if self.p2p is not None:
            self._torch.cuda.synchronize()
            for c in self.p2p:
                c.close()
            self.p2p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _allreduce_async(self, g):
        dist = self._dist
        if self.p2p is not None:  # in-stream, one hop
            self.p2p[g].all_reduce_sum_f32(self.reds[g].data_ptr(), self.reds[g].numel(), self._streams[g])
            return _Done()
        if self.direct is not None:  # in-stream: nothing to wait for afterwards
            self.direct[g].all_reduce_sum_f32(self.reds[g].data_ptr(), self.reds[g].numel(), self._streams[g])
            return _Done()
        if self.staged:
            self._hosts[g].copy_(self.reds[g])  # synchronises with the current stream
            dist.all_reduce(self._hosts[g], op=dist.ReduceOp.SUM, group=self.group)
            self.reds[g].copy_(self._hosts[g])
            return _Done()
        ts = self._torch_streams[g]
        if ts is None:
            return dist.all_reduce(self.reds[g], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        # torch.distributed orders a collective after the CURRENT torch stream and wait() makes the CURRENT stream wait
        # for it: issue and wait under the group's own stream, so that only this group's kernels are held up
        with self._torch.cuda.stream(ts):
            work = dist.all_reduce(self.reds[g], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return _OnStream(self._torch, ts, work)

    def _allreduce(self):  # the synchronous single-group form (kept for callers of run_sharded_levels)
        self._allreduce_async(0).wait()

    def track(self):
        """Enqueue the whole tracking of all problems; returns without synchronising (unless staged)."""
        if len(self.batches) == 1:
            run_sharded_levels(self.batch, self.batch.op, self._allreduce)
        else:
            run_interleaved([sharded_program(b, b.op, (lambda g=g: self._allreduce_async(g)))
                             for g, b in enumerate(self.batches)])

    def poses(self):
        out = np.concatenate([b.poses() for b in self.batches], 0)
        if self.p2p is not None:
            # A time-out is local knowledge: the rank whose mailbox filled in time saw nothing wrong, while the peer that
            # waited in vain went on with a wrong sum -- the lock-step solves have diverged and every later exchange
            # would carry the damage on. All ranks therefore agree on the verdict (one MIN all-reduce on the torch group,
            # every rank calls it at this point whatever it saw), and on a failure ALL of them drop the mailboxes
            # (torch.distributed serves the next tracking) and raise.
            ok = not any(c.error() for c in self.p2p)
            if not _agree(self._torch, self._dist, self.group, ok):
                self._torch.cuda.synchronize()
                for c in self.p2p:
                    c.close()
                self.p2p = None
                raise RuntimeError("P2P exchange timed out on " + ("this rank" if not ok else "a peer rank") +
                                   ": a rank did not receive its peers' records in time; the results of this tracking "
                                   "are invalid on EVERY rank (the mailboxes are closed, torch.distributed takes over)")
        return out
