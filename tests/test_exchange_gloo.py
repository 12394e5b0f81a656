"""N>1 data-parallel exchange on CPU: 2 ranks, gloo.  Checks that (a) the bucketed all-reduce equals the sum over
ranks, (b) buckets are launched from inside 'backward' as soon as their last parameter is ready (after the learning
step), (c) a parameter written twice per step (tied weights) only releases its bucket on the second write."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multimeditron_amd.train.exchange import GradExchanger


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1000
        # params: (key, start, end); key 3 is "tied" (two writes per step)
        segs = [(1, 0, 256), (2, 256, 600), (3, 600, 1000)]
        grad = torch.zeros(n)
        ex = GradExchanger(grad, [(0, n)], segs, bucket_elems=300, dist=dist)
        assert len(ex.buckets) == 4
        order = [3, 2, 3, 1]          # backward order: last layers first, key 3 written twice
        results = []
        for step in range(3):
            grad.copy_(torch.arange(n, dtype=torch.float32) * (rank + 1) + step)
            ex.begin_step(True)
            early_before_finish = []
            for k in order:
                ex.on_ready(k)
                early_before_finish.append(ex.launched_early)
            ex.finish_step()
            expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world)) + step * world
            assert torch.equal(grad, expect), (rank, step)
            results.append(early_before_finish)
        # step 0 learns the write counts and the launch sequence -> nothing early; later steps launch from inside "backward"
        assert results[0] == [0, 0, 0, 0]
        # learned completion order: [300,600) (key 2), then [600,900),[900,1000) (2nd write of key 3), then [0,300) (key 1)
        assert ex.order == [1, 2, 3, 0], ex.order
        assert results[1] == [0, 1, 3, 4], results[1]
        assert results[1] == results[2]
        # a rank whose data skips a parameter (never ready) must still issue every collective, in the agreed order
        grad.copy_(torch.arange(n, dtype=torch.float32) * (rank + 1))
        ex.begin_step(True)
        for k in ([3, 3, 1] if rank == 0 else order):      # rank 0 never writes key 2 this step
            ex.on_ready(k)
        ex.finish_step()
        assert torch.equal(grad, torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))), rank
        # accumulation micro-step: no exchange
        grad.fill_(rank + 1.0)
        ex.begin_step(False)
        for k in order:
            ex.on_ready(k)
        ex.finish_step()
        assert torch.all(grad == rank + 1.0)
        q.put((rank, "ok", results[1]))
    except Exception as e:  # pragma: no cover
        q.put((rank, f"fail: {type(e).__name__}: {e}", None))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(o[1] == "ok" for o in out), out


def test_single_process_is_a_noop():
    grad = torch.ones(100)
    ex = GradExchanger(grad, [(0, 100)], [(1, 0, 100)], 64, dist=None)
    ex.begin_step(True)
    ex.on_ready(1)
    ex.finish_step()
    assert torch.all(grad == 1) and ex.written(1) and not ex.written(2)
