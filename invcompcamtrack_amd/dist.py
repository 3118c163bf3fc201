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


def global_norm(pts_local, group=None, dist=None):
    """Mean and mean squared radius (odometer.cpp:193-214) of the union of all ranks' points.
    pts_local: (3, n_local) float64. Two tiny all-reduces (f64)."""
    import torch
    if dist is None:
        import torch.distributed as dist
    acc = torch.tensor([pts_local[0].sum(), pts_local[1].sum(), pts_local[2].sum(), float(pts_local.shape[1])],
                       dtype=torch.float64)
    dist.all_reduce(acc, group=group)
    mean = (acc[:3] / acc[3]).numpy().copy()
    var = torch.tensor([float(((pts_local - mean[:, None]) ** 2).sum())], dtype=torch.float64)
    dist.all_reduce(var, group=group)
    return mean, float(var[0] / acc[3])


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


class RcclDirect:
    """RCCL called directly (ctypes on the librccl.so PyTorch ships) with its own communicator, so that the all-reduce
    is enqueued ON THE COMPUTE STREAM between the tail and the finish kernels. torch.distributed runs collectives on
    its own stream and synchronises with events in both directions; measured on MI355X that costs ~24 us of idle
    compute stream per collective even when there is nothing to wait for (profiles/r01_notes.md) -- per Gauss-Newton
    iteration. The unique id is created on rank 0 and broadcast through the existing torch.distributed group.
    Any failure while setting up raises, and ShardedTracker falls back to torch.distributed."""

    def __init__(self, torch, dist, group=None):
        import ctypes as C
        import os
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self._C, self._lib = C, C.CDLL(path)

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]

        L = self._lib
        L.ncclGetUniqueId.argtypes, L.ncclGetUniqueId.restype = [C.POINTER(UniqueId)], C.c_int
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        L.ncclCommInitRank.restype = C.c_int
        L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclAllReduce.restype = C.c_int
        L.ncclCommDestroy.argtypes, L.ncclCommDestroy.restype = [C.c_void_p], C.c_int
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        uid = UniqueId()
        if rank == 0 and L.ncclGetUniqueId(C.byref(uid)) != 0:
            raise RuntimeError("ncclGetUniqueId failed")
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        raw = C.string_at(C.addressof(uid), 128)  # the raw 128 bytes (a c_char array read as bytes stops at a NUL)
        t = torch.tensor(list(raw), dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        C.memmove(C.addressof(uid), bytes(t.cpu().tolist()), 128)
        self._comm = C.c_void_p()
        rc = L.ncclCommInitRank(C.byref(self._comm), world, uid, rank)
        if rc != 0 or not self._comm:
            raise RuntimeError(f"ncclCommInitRank failed ({rc})")
        self.world = world
        # self-test before anything relies on it: the sum of ones over the ranks must be the world size
        probe = torch.ones(4, dtype=torch.float32, device="cuda")
        self.all_reduce_sum_f32(probe.data_ptr(), 4, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        if not bool((probe == float(world)).all().item()):
            raise RuntimeError(f"direct all-reduce self-test failed: {probe.tolist()} != {world}")

    def all_reduce_sum_f32(self, dev_ptr, count, stream):
        rc = self._lib.ncclAllReduce(dev_ptr, dev_ptr, count, 7, 0, self._comm, stream)  # ncclFloat32, ncclSum
        if rc != 0:
            raise RuntimeError(f"ncclAllReduce failed ({rc})")

    def close(self):
        if getattr(self, "_comm", None):
            self._lib.ncclCommDestroy(self._comm)
            self._comm = None


class ShardedTracker:
    """Drives one TrackBatch, or several (groups of problems that are software-pipelined against each other's
    collectives), in sharded mode. Collective path: RcclDirect (in-stream, default when every rank can set it up),
    else torch.distributed (asynchronous on the process group's stream; ``wait()`` makes the compute stream wait,
    not the host), else -- ``staged`` -- through host memory with any backend (tests)."""

    def __init__(self, batch, group=None, staged=False, direct=True):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.batches = list(batch) if isinstance(batch, (list, tuple)) else [batch]
        self.batch = self.batches[0]
        self.group, self.staged = group, staged
        self.reds, self._hosts = [], []
        import os
        want_direct = (direct and not staged and dist.get_backend(group) == "nccl"
                       and not os.environ.get("ICTR_NO_RCCL_DIRECT"))
        # direct path with several groups: every group gets its own stream AND its own communicator, so that one
        # group's in-stream all-reduce (pure link latency) runs while the other group's kernels use the GPU
        self._torch_streams = [torch.cuda.Stream() if (want_direct and len(self.batches) > 1) else None
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
        self.direct = None
        if want_direct:
            try:
                self.direct = [RcclDirect(torch, dist, group) for _ in self.batches]
            except Exception as exc:  # keep working through torch.distributed
                import sys
                print(f"[ictr.dist] direct RCCL unavailable ({exc!r}); using torch.distributed", file=sys.stderr)
                self.direct = None
            # every rank must take the same path: agree on it through the torch group
            flag = torch.tensor([1.0 if self.direct is not None else 0.0], device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if float(flag.item()) < 1.0:
                self.direct = None
            if self.direct is None:
                for b in self.batches:
                    b.set_stream(torch.cuda.current_stream().cuda_stream)
                self._streams = [torch.cuda.current_stream().cuda_stream for _ in self.batches]

    def _allreduce_async(self, g):
        dist = self._dist
        if self.direct is not None:  # in-stream: nothing to wait for afterwards
            self.direct[g].all_reduce_sum_f32(self.reds[g].data_ptr(), self.reds[g].numel(), self._streams[g])
            return _Done()
        if self.staged:
            self._hosts[g].copy_(self.reds[g])  # synchronises with the current stream
            dist.all_reduce(self._hosts[g], op=dist.ReduceOp.SUM, group=self.group)
            self.reds[g].copy_(self._hosts[g])
            return _Done()
        return dist.all_reduce(self.reds[g], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

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
        return np.concatenate([b.poses() for b in self.batches], 0)
