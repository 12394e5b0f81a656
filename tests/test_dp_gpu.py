"""Data-parallel path with real kernels (GPU): 2 ranks sharing the one GPU of the box, gloo backend for the collective.
Invariant: DP over 2 ranks with micro-batches (b0 | b1) == 1 process with gradient_accumulation_steps=2 over (b0, b1)
(both average the two micro-batch gradients before clip + AdamW).  Also exercises bucket launches from inside backward."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(tmp, dtype="float32"):
    import sys
    sys.path.insert(0, ROOT)
    from oracle import ref_cpu as R
    from tests.model_utils import build_from_golden
    meta, w, v = R.load_golden("tiny_clip_llama", os.path.join(ROOT, "tests", "golden"))
    return build_from_golden(meta, w, tmp, dtype), v, R


def _rank(rank, world, port, tmp, q, shard="1"):
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MM_SHARD_OPTIM=shard)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from multimeditron_amd.train.trainer import MultimodalTrainer, TrainingMode
        from tests.model_utils import to_device
        model, v, R = _build(os.path.join(tmp, f"r{rank}"))
        tr = MultimodalTrainer(model, training_mode=TrainingMode.FULL, learning_rate=1e-3, betas=(0.9, 0.95), max_grad_norm=1.0,
                               bucket_mb=1)   # tiny buckets -> several buckets, launched from inside backward
        assert tr.shard_optim == (shard == "1")
        n_state = int(tr.master.numel())
        cases = [("right", "interleaved4"), ("left", "textonly"), ("interleaved4", "right")]
        early = []
        for step in cases:
            tr.training_step(to_device(R.golden_batch(v, step[rank])))
            early.append(tr.exchanger.launched_early)
        tr.synchronize()
        torch.cuda.synchronize()
        sd = {k: p.detach().float().cpu().numpy() for k, p in model.named_parameters()}   # by value through the queue
        q.put((rank, "ok", sd, early, len(tr.exchanger.buckets), n_state))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "fail: " + traceback.format_exc()[-1500:], None, None, None, None))


def _run_dp2(tmp, shard):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, os.path.join(str(tmp), "shard" + shard), q, shard)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(o[1] == "ok" for o in out), [o[1] for o in out]
    return sorted(out, key=lambda o: o[0])


def test_dp2_equals_grad_accumulation(tmp_path):
    """DP2 (sharded optimiser step: reduce-scatter, AdamW on each rank's half of the state, all-gather of the parameters -- the
    default for world > 1; and the replicated step, MM_SHARD_OPTIM=0) == one process with gradient accumulation over the same
    micro-batches; the two ranks of a run end with bit-identical parameters; a rank of the sharded run holds half the state."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sharded = _run_dp2(tmp_path, "1")
    repl = _run_dp2(tmp_path, "0")
    for run in (sharded, repl):
        for k in run[0][2]:
            assert (run[0][2][k] == run[1][2][k]).all(), k                      # rank 0 == rank 1, bit for bit
    assert sharded[0][5] + sharded[1][5] <= repl[0][5] + 16 * sharded[0][4] and sharded[0][5] < 0.6 * repl[0][5]
    for k in repl[0][2]:
        a, b = torch.from_numpy(sharded[0][2][k]), torch.from_numpy(repl[0][2][k])
        assert float((a - b).norm() / (b.norm() + 1e-12)) < 1e-4, k             # summation order of the norm; the fp32 attention backward sums dK/dV with float atomics
    r0 = sharded[0]
    dp_params, early, nb = r0[2], r0[3], r0[4]
    assert nb > 1
    assert early[0] == 0 and early[1] > 0 and early[2] > 0, early     # step 1 learns write counts; later steps overlap

    # single process, gradient accumulation over the same micro-batches
    from multimeditron_amd.train.trainer import MultimodalTrainer, TrainingMode
    from tests.model_utils import to_device
    model, v, R = _build(str(tmp_path / "single"))
    tr = MultimodalTrainer(model, training_mode=TrainingMode.FULL, learning_rate=1e-3, betas=(0.9, 0.95), max_grad_norm=1.0,
                           gradient_accumulation_steps=2)
    for a, b in [("right", "interleaved4"), ("left", "textonly"), ("interleaved4", "right")]:
        tr.training_step(to_device(R.golden_batch(v, a)))
        tr.training_step(to_device(R.golden_batch(v, b)))
    tr.synchronize()
    torch.cuda.synchronize()
    for k, p in model.named_parameters():
        ref = p.detach().float().cpu()
        err = float((torch.from_numpy(dp_params[k]) - ref).norm() / (ref.norm() + 1e-12))
        assert err < 1e-4, (k, err)


def _abi_rank(port, tmp, q):
    """One rank driving RCCL through the C-ABI communicator (mm_comm_*): a 1-rank sum is the identity, so every result is
    known exactly; what is exercised is the binding, the stream/event ordering and the bucket plumbing."""
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MM_FORCE_EXCHANGE="1")
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=0, world_size=1)
        from multimeditron_amd.train.comm import RcclComm
        from multimeditron_amd.train.trainer import MultimodalTrainer, TrainingMode
        from tests.model_utils import to_device
        res = {}
        for algo in (0, 1):
            c = RcclComm(dist, None, algo=algo)
            for dtype in (torch.bfloat16, torch.float32):
                x = torch.randn(1_000_003, device="cuda").to(dtype)            # odd count: the tail path of algo 1
                y = x.clone() * 3                                              # a producer on the compute stream the comm stream must wait for
                ref = y.clone()
                c.all_reduce(y).wait()
                z = ref.clone()[:1_000_000]
                c.reduce_scatter(z).wait()
                c.all_gather(z).wait()
                torch.cuda.synchronize()
                res[(algo, str(dtype))] = bool(torch.equal(y, ref)) and bool(torch.equal(z, ref[:1_000_000]))
            c.close()
        params = {}
        for mode in ("torch-none", "abi", "abi-rsag", "abi-sharded"):
            if mode == "torch-none":
                os.environ.pop("MM_COMM", None)
                os.environ.pop("MM_FORCE_EXCHANGE", None)
            else:      # abi-sharded: the sharded optimiser step's reduce-scatter / all-gather through the communicator
                os.environ.update(MM_COMM="abi" if mode == "abi-sharded" else mode, MM_FORCE_EXCHANGE="1",
                                  MM_SHARD_OPTIM="1" if mode == "abi-sharded" else "0")
            model, v, R = _build(os.path.join(tmp, mode), "bfloat16")
            tr = MultimodalTrainer(model, training_mode=TrainingMode.FULL, learning_rate=1e-3, betas=(0.9, 0.95), max_grad_norm=1.0,
                                   bucket_mb=1)
            early = []
            for name in ("right", "interleaved4", "left"):
                tr.training_step(to_device(R.golden_batch(v, name)))
                early.append(tr.exchanger.launched_early)
            tr.synchronize()
            torch.cuda.synchronize()
            params[mode] = ({k: p.detach().float().cpu().numpy() for k, p in model.named_parameters()}, early,
                            type(tr.exchanger.comm).__name__)
            tr.close()
        q.put(("ok", res, params))
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put(("fail: " + traceback.format_exc()[-2500:], None, None))


def test_c_abi_communicator_single_rank(tmp_path):
    """mm_comm_* (RCCL behind the C-ABI): collectives on a 1-rank communicator leave the data bit-identical, and a trainer whose
    buckets travel through it (launched from inside backward, on the communicator's stream) ends with the same parameters,
    bit for bit, as one that exchanges nothing."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_abi_rank, args=(_free_port(), str(tmp_path), q))
    p.start()
    status, res, params = q.get(timeout=240)
    p.join(timeout=60)
    assert status == "ok", status
    assert all(res.values()), res
    base = params["torch-none"][0]
    for mode in ("abi", "abi-rsag"):
        got, early, kind = params[mode]
        assert kind == "RcclComm"
        assert early[0] == 0 and early[1] > 0 and early[2] > 0, early
        for k in base:
            assert (got[k] == base[k]).all(), (mode, k)
    got, early, kind = params["abi-sharded"]        # same update up to the summation order of the gradient norm (bf16 parameters)
    assert kind == "RcclComm" and early[1] > 0
    for k in base:
        a, b = torch.from_numpy(got[k]), torch.from_numpy(base[k])
        assert float((a - b).norm() / (b.norm() + 1e-12)) < 2e-3, k


def _rank_nccl(port, tmp, q, shard, force):
    """ONE rank on the DEFAULT transport (torch.distributed backend "nccl" = RCCL, high-priority process-group stream), set up exactly
    as bench.py does for the driver's multi-GPU launch."""
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MM_SHARD_OPTIM=shard)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if force:
            os.environ["MM_FORCE_EXCHANGE"] = "1"
        else:
            os.environ.pop("MM_FORCE_EXCHANGE", None)
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        if force:
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, pg_options=opts)
        from multimeditron_amd.train.trainer import MultimodalTrainer, TrainingMode
        from tests.model_utils import to_device
        model, v, R = _build(os.path.join(tmp, f"nccl{shard}{int(force)}"), dtype="bfloat16")
        tr = MultimodalTrainer(model, training_mode=TrainingMode.FULL, learning_rate=1e-3, betas=(0.9, 0.95), max_grad_norm=1.0, bucket_mb=1)
        early = []
        for name in ("right", "interleaved4", "left"):
            tr.training_step(to_device(R.golden_batch(v, name)))
            early.append(tr.exchanger.launched_early if force else 0)
        tr.synchronize()
        torch.cuda.synchronize()
        sd = {k: p.detach().float().cpu().numpy() for k, p in model.named_parameters()}
        q.put(("ok", sd, early, bool(tr.shard_optim)))
        if force:
            dist.barrier()
            dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put(("fail: " + traceback.format_exc()[-2000:], None, None, None))


def _run_one(tmp, shard, force):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rank_nccl, args=(_free_port(), str(tmp), q, shard, force))
    p.start()
    out = q.get(timeout=120)                 # a hang in a collective fails here, not at the end of the round
    p.join(timeout=60)
    assert out[0] == "ok", out[0]
    return out


@pytest.mark.parametrize("shard", ["1", "0"])
def test_default_rccl_transport_with_one_rank(tmp_path, shard):
    """The transport the driver's 8-GPU run takes (reference config/deepspeed.json:5-19, cli/train.py:200-201 -> here train/exchange.py over
    torch.distributed "nccl"): in-place reduce_scatter_tensor + all_gather_into_tensor (MM_SHARD_OPTIM=1) and bucketed all_reduce (=0), on a
    high-priority process-group stream, launched from inside backward -- run through RCCL on the GPU with ONE rank (all one GPU allows).
    Three bf16 trainer steps must leave the parameters BIT-identical to a run that exchanges nothing, and buckets must launch early."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import numpy as np
    ref = _run_one(tmp_path, shard, False)
    got = _run_one(tmp_path, shard, True)
    assert got[3] == (shard == "1")
    assert all(e > 0 for e in got[2][1:]), got[2]                     # from step 2 on buckets leave from inside backward
    for k in ref[1]:
        assert np.array_equal(ref[1][k], got[1][k]), k
