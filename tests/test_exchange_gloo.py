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


def _worker_sharded(rank, world, port, q):
    """Sharded exchange (reduce-scatter + all-gather) on CPU tensors: the halves the trainer's sharded optimiser step slots between."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1000
        ranges = [(0, 601), (608, 1000)]                         # two trainable ranges (decay / no decay), ragged ends
        segs = [(1, 0, 256), (2, 256, 601), (3, 608, 1000)]
        grad = torch.zeros(n)
        ex = GradExchanger(grad, ranges, segs, bucket_elems=300, dist=dist, sharded=True)
        assert [(b.start, b.end) for b in ex.buckets] == [(0, 300), (300, 600), (600, 601), (608, 908), (908, 1000)]
        assert [b.q for b in ex.buckets] == [144, 144, 0, 144, 40]   # multiples of 8; bucket (600, 601) is all tail
        params = torch.arange(n, dtype=torch.float32).clone()    # replicated "parameters"
        ref = params.clone()
        lr = 0.5
        for step in range(3):
            full = [torch.sin(torch.arange(n, dtype=torch.float32) * (r + 1) + step) for r in range(world)]
            grad.copy_(full[rank])
            ex.begin_step(True)
            for k in (3, 2, 1):
                ex.on_ready(k)
            ex.finish_step()
            tot = sum(full)
            own = torch.zeros(n, dtype=torch.bool)
            for bk in ex.buckets:
                (s, e), (ts, te) = bk.shard(rank, world)
                assert torch.allclose(grad[s:e], tot[s:e]) and torch.allclose(grad[ts:te], tot[ts:te]), (rank, step, bk.start)
                own[s:e] = True
                own[ts:te] = True
                params[s:e] -= lr * grad[s:e]                    # "optimiser": my share + the tail every rank keeps
                params[ts:te] -= lr * grad[ts:te]
            works = [ex.all_gather_params(params, bk) for bk in ex.buckets]
            for w in works:
                w.wait()
            for a, b in ranges:
                ref[a:b] -= lr * tot[a:b]
            assert torch.allclose(params, ref, atol=1e-6), (rank, step, float((params - ref).abs().max()))
            assert torch.equal(params[601:608], torch.arange(601, 608, dtype=torch.float32))          # outside the ranges: untouched
        # every element of a range is owned by exactly one rank, or by all of them (tails)
        cnt = own.float()
        dist.all_reduce(cnt)
        for a, b in ranges:
            assert set(cnt[a:b].tolist()) <= {1.0, float(world)}
        q.put((rank, "ok", params.tolist()))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, f"fail: {traceback.format_exc()[-800:]}", None))
    finally:
        dist.destroy_process_group()


def test_sharded_exchange_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sharded, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(o[1] == "ok" for o in out), out
    assert out[0][2] == out[1][2]                                # the two ranks end with bit-identical parameters


def test_single_process_is_a_noop():
    grad = torch.ones(100)
    ex = GradExchanger(grad, [(0, 100)], [(1, 0, 100)], 64, dist=None)
    ex.begin_step(True)
    ex.on_ready(1)
    ex.finish_step()
    assert torch.all(grad == 1) and ex.written(1) and not ex.written(2)
