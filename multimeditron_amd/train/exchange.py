"""Bucketed gradient all-reduce over a flat gradient buffer, launched from inside backward.

Pure torch.distributed (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests); no kernels here.
The flat trainable ranges are cut into buckets (default 256 MiB of bf16: large enough to run RCCL at link rate, small
enough that the first bucket leaves while the decoder is still back-propagating).  A bucket is launched the moment the
LAST expected gradient write of every parameter that overlaps it has been enqueued; `async_op=True` puts the collective
on the process group's own stream, ordered after the work enqueued so far, so it overlaps the rest of backward.
How many writes a parameter receives per step (tied embedding / lm_head = 2) is learned during the first step, which
therefore exchanges after backward."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch


class _Bucket:
    __slots__ = ("start", "end", "params", "pending", "work")

    def __init__(self, start, end):
        self.start, self.end, self.params, self.pending, self.work = start, end, [], 0, None


class GradExchanger:
    def __init__(self, flat_grad: torch.Tensor, ranges: Sequence[Tuple[int, int]], segments: Sequence[Tuple[int, int, int]],
                 bucket_elems: int, dist=None, group=None):
        """ranges: trainable [start,end) of flat_grad; segments: (param_key, start, end) for every trainable param."""
        self.grad = flat_grad
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group) if dist is not None else 1
        self.buckets: List[_Bucket] = []
        for s, e in ranges:
            a = s
            while a < e:
                b = min(e, a + bucket_elems)
                self.buckets.append(_Bucket(a, b))
                a = b
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

    def begin_step(self, exchange_this_step: bool):
        self._count = {}
        self.active = exchange_this_step and self.world > 1
        self.launched_early = 0
        for bk in self.buckets:
            bk.pending, bk.work = len(bk.params), None

    def on_ready(self, key: int):
        c = self._count.get(key, 0) + 1
        self._count[key] = c
        if not self.active or self.expected is None or c != self.expected.get(key, 1):
            return
        for bk in self.param_buckets.get(key, ()):
            bk.pending -= 1
            if bk.pending == 0:
                self._launch(bk)
                self.launched_early += 1

    def _launch(self, bk: _Bucket):
        bk.work = self.dist.all_reduce(self.grad[bk.start:bk.end], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish_step(self):
        """Launch whatever has not left yet, then make the current stream wait for every bucket."""
        if self.expected is None:
            self.expected = dict(self._count)
        if not self.active:
            return
        for bk in self.buckets:
            if bk.work is None:
                self._launch(bk)
        for bk in self.buckets:
            bk.work.wait()
            bk.work = None

    def written(self, key: int) -> bool:
        return self._count.get(key, 0) > 0
