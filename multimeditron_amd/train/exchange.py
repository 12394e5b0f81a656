"""Bucketed gradient all-reduce over a flat gradient buffer, launched from inside backward.

Transport: torch.distributed (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests) by default, or the
C-ABI communicator (`comm=RcclComm`, train/comm.py -> mm_comm_allreduce_bucket) when the trainer is started with
MM_COMM=abi / abi-rsag; no kernels here.
The flat trainable ranges are cut into buckets (default 256 MiB of bf16: large enough to run RCCL at link rate, small
enough that the first bucket leaves while the decoder is still back-propagating).  A bucket is launched the moment the
LAST expected gradient write of every parameter that overlaps it has been enqueued; `async_op=True` puts the collective
on the process group's own stream, ordered after the work enqueued so far, so it overlaps the rest of backward.
How many writes a parameter receives per step (tied embedding / lm_head = 2) is learned during the first step, which
therefore exchanges after backward.

Collectives must be issued in the SAME order on every rank, but readiness order is data dependent (a text-only
micro-batch never touches the vision tower).  So the launch sequence is fixed once: the order in which buckets
completed during rank 0's first step, broadcast to all ranks; afterwards a bucket leaves only when it is ready AND every
bucket before it in that sequence has left; whatever is still pending is flushed in sequence order after backward."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch


class _Bucket:
    __slots__ = ("start", "end", "params", "pending", "work", "q")

    def __init__(self, start, end):
        self.start, self.end, self.params, self.pending, self.work = start, end, [], 0, None
        self.q = 0           # sharded exchange: elements per rank of the reduce-scattered part [start, start + world * q)

    def shard(self, rank, world):
        """-> (s, e) of rank's 1/world share of the reduce-scattered part, and (ts, te) of the tail every rank keeps whole."""
        s = self.start + rank * self.q
        return (s, s + self.q), (self.start + world * self.q, self.end)


class _Works:
    """several collectives of one bucket behind the Work interface (wait() orders the current stream after all of them)"""
    __slots__ = ("works",)

    def __init__(self, works):
        self.works = [w for w in works if w is not None]

    def wait(self):
        for w in self.works:
            w.wait()


class GradExchanger:
    def __init__(self, flat_grad: torch.Tensor, ranges: Sequence[Tuple[int, int]], segments: Sequence[Tuple[int, int, int]],
                 bucket_elems: int, dist=None, group=None, force: bool = False, comm=None, sharded: bool = False):
        """ranges: trainable [start,end) of flat_grad; segments: (param_key, start, end) for every trainable param.
        comm: an `RcclComm` to carry the buckets instead of dist.all_reduce (dist stays the control channel).
        sharded: REDUCE-SCATTER instead of all-reduce (the sharded optimiser step, reference config/deepspeed.json:5-19): after the
        exchange rank r holds the summed gradient of its 1/world share of every bucket only (`_Bucket.shard`), plus the bucket's
        tail (fewer than 8 * world elements, all-reduced) which every rank keeps; `all_gather_params` is the other half."""
        self.grad = flat_grad
        self.dist, self.group = dist, group
        self.comm = comm
        self.world = dist.get_world_size(group) if dist is not None else 1
        self.rank = dist.get_rank(group) if dist is not None else 0
        self.force = force
        self.sharded = bool(sharded) and dist is not None
        self.buckets: List[_Bucket] = []
        for s, e in ranges:
            a = s
            while a < e:
                b = min(e, a + bucket_elems)
                self.buckets.append(_Bucket(a, b))
                a = b
        if self.sharded:
            for bk in self.buckets:
                bk.q = (bk.end - bk.start) // (8 * self.world) * 8           # shards start on 16-byte boundaries
        self.param_buckets: Dict[int, List[_Bucket]] = {}
        for key, s, e in segments:
            for bk in self.buckets:
                if s < bk.end and e > bk.start:
                    bk.params.append(key)
                    self.param_buckets.setdefault(key, []).append(bk)
        self.expected: Optional[Dict[int, int]] = None
        self._count: Dict[int, int] = {}
        self.launched_early = 0
        self.active = False
        self.order: List[int] = list(range(len(self.buckets)))     # agreed launch sequence (bucket indices)
        self._ready = [False] * len(self.buckets)
        self._next = 0
        self._tick = 0
        self._last_touch = [0] * len(self.buckets)
        self._bucket_index = {id(b): i for i, b in enumerate(self.buckets)}

    def begin_step(self, exchange_this_step: bool):
        self._count = {}
        self.active = exchange_this_step and (self.world > 1 or self.force)
        self.launched_early = 0
        self._ready = [False] * len(self.buckets)
        self._next = 0
        for bk in self.buckets:
            bk.pending, bk.work = len(bk.params), None

    def on_ready(self, key: int):
        c = self._count.get(key, 0) + 1
        self._count[key] = c
        if self.expected is None:                    # learning step: remember when each bucket was last written
            self._tick += 1
            for bk in self.param_buckets.get(key, ()):
                self._last_touch[self._bucket_index[id(bk)]] = self._tick
            return
        if not self.active or c != self.expected.get(key, 1):
            return
        for bk in self.param_buckets.get(key, ()):
            bk.pending -= 1
            if bk.pending == 0:
                self._ready[self._bucket_index[id(bk)]] = True
        self._drain(early=True)

    def _drain(self, early: bool):
        while self._next < len(self.order) and self._ready[self.order[self._next]]:
            self._launch(self.buckets[self.order[self._next]])
            self._next += 1
            if early:
                self.launched_early += 1

    def _launch(self, bk: _Bucket):
        if self.sharded:
            bk.work = self._reduce_scatter(bk)
            return
        if self.comm is not None:
            bk.work = self.comm.all_reduce(self.grad[bk.start:bk.end])
            return
        bk.work = self.dist.all_reduce(self.grad[bk.start:bk.end], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish_step(self):
        """Launch whatever has not left yet, then make the current stream wait for every bucket."""
        if self.expected is None:
            self.expected = dict(self._count)
            # launch sequence = completion order of this first step on rank 0 (never-written buckets last)
            order = sorted(range(len(self.buckets)), key=lambda i: (self._last_touch[i] == 0, self._last_touch[i], i))
            if self.dist is not None and (self.world > 1 or self.force):
                box = [order]
                self.dist.broadcast_object_list(box, src=0, group=self.group)
                order = list(box[0])
            self.order = order
        if not self.active:
            return
        self._ready = [True] * len(self.buckets)
        self._drain(early=False)
        for bk in self.buckets:
            bk.work.wait()
            bk.work = None

    def written(self, key: int) -> bool:
        return self._count.get(key, 0) > 0

    # ------------------------------------------------------------------ sharded exchange (reduce-scatter / all-gather)
    def _reduce_scatter(self, bk: _Bucket):
        """rank r's share of the bucket's sum lands IN PLACE in its share of the gradient buffer (the other shares are left with
        partial data nobody reads); the tail is all-reduced."""
        d, g, W = self.dist, self.grad, self.world
        (s, e), (ts, te) = bk.shard(self.rank, W)
        works = []
        if bk.q > 0:
            main = g[bk.start:ts]
            if self.comm is not None:
                works.append(self.comm.reduce_scatter(main))
            else:
                works.append(d.reduce_scatter_tensor(g[s:e], main, op=d.ReduceOp.SUM, group=self.group, async_op=True))
        if te > ts:
            works.append(self.comm.all_reduce(g[ts:te]) if self.comm is not None else
                         d.all_reduce(g[ts:te], op=d.ReduceOp.SUM, group=self.group, async_op=True))
        return _Works(works)

    def all_gather_params(self, flat_data: torch.Tensor, bk: _Bucket):
        """The other half: every rank has updated its share of the bucket's parameters (and the whole tail); collect the shares in
        place.  Enqueued behind the CURRENT stream's work (the optimiser's side stream); returns a Work."""
        if bk.q == 0:
            return _Works([])
        (s, e), (ts, _) = bk.shard(self.rank, self.world)
        main = flat_data[bk.start:ts]
        if self.comm is not None:
            return _Works([self.comm.all_gather(main)])
        return _Works([self.dist.all_gather_into_tensor(main, flat_data[s:e], group=self.group, async_op=True)])
