"""The C-ABI communicator (`mm_comm_*`, csrc/mm_comm.hip) as the gradient exchange's transport.

`GradExchanger` launches a bucket either through torch.distributed (`dist.all_reduce(async_op=True)`: the default, and the
only transport the CPU/gloo tests can run) or through `RcclComm`, which drives RCCL directly from libmmhip.so:
  * one communicator per process, created from a 128-byte id that rank 0 makes and torch.distributed's object broadcast
    (any initialised backend: it is only the side channel) carries to the other ranks;
  * collectives run on ONE dedicated high-priority HIP stream owned here; a bucket's launch = record an event on the compute
    stream (everything that wrote the bucket), make the comm stream wait for it, enqueue the collective, record the bucket's
    done-event; `wait()` makes the compute stream wait for that event.  No host synchronisation anywhere;
  * `algo` 0 = ncclAllReduce, 1 = reduce-scatter + all-gather in place (the two halves a sharded optimiser step slots
    between; also exported on their own as `reduce_scatter` / `all_gather`).
Selected with MM_COMM=abi (all-reduce) or MM_COMM=abi-rsag.  Replaces DeepSpeed's gradient reduction
(reference config/deepspeed.json:5-19)."""
from __future__ import annotations

import ctypes

import torch

from .. import _lib
from ..kernels import dt, _p


class _EventWork:
    """What GradExchanger.finish_step waits on (the torch.distributed Work interface, reduced to wait())."""
    __slots__ = ("event",)

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class RcclComm:
    def __init__(self, dist=None, group=None, algo: int = 0, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("RcclComm drives RCCL on this process's GPU: it needs one (the CPU tests use torch.distributed/gloo)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.rank = dist.get_rank(group) if dist is not None else 0
        self.world = dist.get_world_size(group) if dist is not None else 1
        self.algo = int(algo)
        ident = ctypes.create_string_buffer(128)
        if self.rank == 0:
            _lib.call("mm_comm_unique_id", ident)
        box = [bytes(ident.raw)]
        if dist is not None and self.world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self._comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.call("mm_comm_init", box[0], self.rank, self.world, ctypes.byref(self._comm))
            self.stream = torch.cuda.Stream(device=self.device, priority=-1)     # high priority: collective kernels take freed CUs first

    def _on_comm_stream(self, fn):
        ev = torch.cuda.Event()
        ev.record()                                          # everything enqueued on the compute stream so far
        self.stream.wait_event(ev)
        with torch.cuda.stream(self.stream):
            fn(self.stream.cuda_stream)
            done = torch.cuda.Event()
            done.record(self.stream)
        return _EventWork(done)

    def all_reduce(self, t: torch.Tensor) -> _EventWork:
        """in-place sum of a contiguous bf16 / fp32 slice across the ranks; returns a handle whose wait() orders the compute
        stream after it."""
        assert t.is_contiguous() and t.is_cuda
        return self._on_comm_stream(lambda s: _lib.call("mm_comm_allreduce_bucket", self._comm, dt(t), _p(t), t.numel(), self.algo, s))

    def reduce_scatter(self, t: torch.Tensor) -> _EventWork:
        assert t.is_contiguous() and t.is_cuda and t.numel() % self.world == 0
        return self._on_comm_stream(lambda s: _lib.call("mm_comm_reduce_scatter", self._comm, dt(t), _p(t), t.numel(), s))

    def all_gather(self, t: torch.Tensor) -> _EventWork:
        assert t.is_contiguous() and t.is_cuda and t.numel() % self.world == 0
        return self._on_comm_stream(lambda s: _lib.call("mm_comm_all_gather", self._comm, dt(t), _p(t), t.numel(), s))

    def close(self):
        if self._comm:
            self.stream.synchronize()
            _lib.call("mm_comm_finalize", self._comm)
            self._comm = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
