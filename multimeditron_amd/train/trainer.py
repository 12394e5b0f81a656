"""Data-parallel training step for the multimodal hot path: the MI355X replacement for the reference's
`MultimodalTrainer(transformers.Trainer)` + DeepSpeed ZeRO-3 (train/trainer.py:16-198, config/deepspeed.json).

What is kept from the reference: `TrainingMode` and its freeze policies (trainer.py:16-23,132-144), `compute_loss`
(trainer.py:91-118: loss = mean CE over the non-ignored tokens of the micro-batch), the optimiser recipe of
config/config_alignment.yaml:38-59 (AdamW lr 1e-4, wd 0.01, clip 1.0, cosine_with_min_lr, gradient accumulation).

What is redesigned for MI355X: parameters / AdamW state are REPLICATED (8.35 B x 14 B = 117 GB fits 288 GB HBM3E, so
no ZeRO sharding, no CPU offload); one process per GPU; gradients live in one flat bf16 buffer cut into buckets;
each bucket's all-reduce (RCCL, `torch.distributed` backend "nccl", which runs it on its own HIP stream) is launched
from the backward pass as soon as the last wgrad of that bucket has been enqueued, so the exchange overlaps the rest
of backward (decoder layers first, vision tower last); the global grad-norm, clipping and AdamW are three fused
HBM-bound kernels over the flat buffers with no host synchronisation."""
from __future__ import annotations

import math
import os
from enum import IntEnum
from typing import Any, Dict, Iterable, List, Optional

import torch

from ..utils import trace_range

from .. import functional as Fm
from .. import kernels as K
from .exchange import GradExchanger


class TrainingMode(IntEnum):
    ALIGNMENT = 0
    END2END = 1
    LM_ONLY = 2
    FULL = 3


TRAINING_MAPPING = {i.name: i for i in TrainingMode}


def cosine_with_min_lr(step: int, total: int, base_lr: float, min_lr: float, warmup: int = 0) -> float:
    """Learning rate of optimiser step `step` (0 = the first) under HF's `lr_scheduler_type: cosine_with_min_lr` with
    `lr_scheduler_kwargs: {min_lr}` (reference config/config_alignment.yaml:54-56 -> HF:optimization.py
    _get_cosine_schedule_with_warmup_lr_lambda, num_cycles = 0.5): linear warm-up from 0, then
    lr = base * (f * (1 - r) + r), f = (1 + cos(pi * progress)) / 2, r = min_lr / base.  tests/test_schedule_cpu.py holds it to
    HF's scheduler step by step."""
    if total <= 0:
        return base_lr
    if step < warmup:
        return base_lr * float(step) / float(max(1, warmup))
    progress = float(step - warmup) / float(max(1, total - warmup))
    factor = 0.5 * (1.0 + math.cos(math.pi * progress))
    rate = min_lr / base_lr if base_lr else 0.0
    return base_lr * max(0.0, factor * (1.0 - rate) + rate)


LR_SCHEDULES = ("cosine_with_min_lr", "cosine", "linear", "constant", "constant_with_warmup")


def scheduled_lr(kind: str, step: int, total: int, base_lr: float, min_lr: float, warmup: int = 0) -> float:
    """HF `lr_scheduler_type` -> learning rate of optimiser step `step` (HF:optimization.py get_scheduler; the lambdas of
    get_linear_schedule_with_warmup, get_cosine_schedule_with_warmup, get_constant_schedule(_with_warmup) and
    get_cosine_with_min_lr_schedule_with_warmup): what the reference's recipes select through `training_args.lr_scheduler_type`."""
    if kind == "cosine_with_min_lr":
        return cosine_with_min_lr(step, total, base_lr, min_lr, warmup)
    if kind == "cosine":
        return cosine_with_min_lr(step, total, base_lr, 0.0, warmup)
    if kind == "constant":
        return base_lr
    if step < warmup:
        return base_lr * float(step) / float(max(1, warmup))
    if kind == "constant_with_warmup" or total <= 0:
        return base_lr
    if kind == "linear":
        return base_lr * max(0.0, float(total - step) / float(max(1, total - warmup)))
    raise ValueError(f"lr_scheduler_type {kind!r}: expected one of {LR_SCHEDULES}")


_SIDE_CUS_DEFAULT = {"MM_ADAMW_CUS": 0, "MM_DEFER_CUS": 0}      # CUs a side burst may use (0 = all): see MultimodalTrainer._side_stream


class MultimodalTrainer:
    def __init__(self, model, training_mode: TrainingMode = TrainingMode.ALIGNMENT, learning_rate: float = 1e-4,
                 weight_decay: float = 0.01, betas=(0.9, 0.999), eps: float = 1e-8, max_grad_norm: float = 1.0,
                 gradient_accumulation_steps: int = 1, max_steps: int = 0, min_lr: Optional[float] = None, warmup_steps: int = 0,
                 bucket_mb: int = 256, process_group=None, data_collator=None, train_dataset=None,
                 overlap_optimizer: bool = True, lr_scheduler_type: str = "cosine_with_min_lr", per_device_train_batch_size: int = 4,
                 shard_optimizer: Optional[bool] = None, loss_rows_only: Optional[bool] = None):
        self.model = model
        # compute_loss: final norm, lm_head and the loss on the labelled rows only (same loss, same gradients; MM_LOSS_ROWS=0 or
        # loss_rows_only=False computes every position's logits as HF does)
        self.loss_rows_only = (os.environ.get("MM_LOSS_ROWS", "1") != "0") if loss_rows_only is None else bool(loss_rows_only)
        self.training_mode = TrainingMode(training_mode)
        self.lr, self.wd, self.betas, self.eps = learning_rate, weight_decay, betas, eps
        self.max_grad_norm = max_grad_norm
        self.accum = max(1, gradient_accumulation_steps)
        self.max_steps, self.min_lr, self.warmup = max_steps, (min_lr if min_lr is not None else learning_rate), warmup_steps
        if lr_scheduler_type not in LR_SCHEDULES:
            raise ValueError(f"lr_scheduler_type {lr_scheduler_type!r}: expected one of {LR_SCHEDULES}")
        self.lr_scheduler_type = lr_scheduler_type
        self.per_device_train_batch_size = per_device_train_batch_size
        self.data_collator, self.train_dataset = data_collator, train_dataset
        self.step_count = 0
        self._micro = 0
        import torch.distributed as dist
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.pg = process_group
        self.world = self.dist.get_world_size(self.pg) if self.dist else 1
        self.bucket_elems = bucket_mb * 1024 * 1024 // 2
        self.overlap_optimizer = overlap_optimizer and torch.cuda.is_available()      # side HIP stream: device only
        # Sharded optimiser step under data parallelism (the reference partitions optimiser state across ranks: ZeRO stage 3,
        # config/deepspeed.json:5-19): reduce-scatter the gradients, AdamW on this rank's 1/world share of master / m / v (only that
        # share is allocated), all-gather the bf16 parameters under the next forward.  Parameters and gradients stay replicated
        # (they fit HBM; no parameter gathering inside forward / backward as ZeRO-3 needs).  Default on for world > 1.
        if shard_optimizer is None:
            shard_optimizer = os.environ.get("MM_SHARD_OPTIM", "1") != "0"
        self.shard_optim = bool(shard_optimizer) and self.dist is not None and (self.world > 1 or bool(os.environ.get("MM_FORCE_EXCHANGE")))
        self._set_mode()
        self._setup_state()
        self._setup_optimizer_pipeline()
        self._setup_wgrad_deferral()
        self._setup_early_gradnorm()

    def _side_stream(self, env: str, priority: int = 0):
        """The HIP stream of a burst that runs beside a chain of small modality-tower kernels (AdamW beside the next ViT forward,
        the deferred wgrad GEMMs beside the ViT backward).  `env` names the number of CUs the burst may use (kernels.masked_stream:
        hipExtStreamCreateWithCUMask, the CUs left out spread evenly over the XCDs): the chain then always finds free CUs instead
        of waiting for a CU to drain.  0 / unset = every CU (a plain stream)."""
        n = int(os.environ.get(env, str(_SIDE_CUS_DEFAULT.get(env, 0))))
        if n > 0 and torch.cuda.is_available():
            return K.masked_stream(n, tag=env)
        # MM_ADAMW_PRIO / MM_DEFER_PRIO: a HIP stream priority (experiment).  Round 4, with the update as short workgroups: the LOWEST priority for
        # AdamW costs 30 ms per step (377 vs 348: the decoder's GEMMs overtake it and then wait for it); the image tower on a HIGHEST-priority
        # stream of its own (tried as MM_VIT_PRIO, removed) costs 43 ms (386 vs 343: its backward then preempts the weight-gradient GEMMs)
        lowprio = os.environ.get(env.replace("_CUS", "_PRIO"))
        if lowprio is not None and torch.cuda.is_available():
            return K.priority_stream(int(lowprio), tag=env)
        return torch.cuda.Stream(priority=priority)

    def _setup_wgrad_deferral(self):
        """Hold back the weight-gradient GEMMs of the first decoder layers (the last to run in backward) and launch them on a
        side stream beside the modality backward (functional.set_wgrad_deferral).  Only when a modality embedder is
        trainable (FULL mode): otherwise there is nothing latency-bound at the tail to hide them under.  Measured on the 8B
        workload at 1 GPU (round 2, tools/step_ab.py): 0 / 8 / 12 / 16 / 20 layers = 391.7 / 391.4 / 388.9 / 386.1 / 386.8 ms/step (the modality chain also slows down when it shares the chip, so most
        of its 14 ms stays exposed); 2-8 layers: no change.  Off under data parallelism: a held-back gradient cannot
        enter its all-reduce bucket before the end of backward.  MM_DEFER_WGRAD_LAYERS overrides the layer count (0 = off)."""
        n = int(os.environ.get("MM_DEFER_WGRAD_LAYERS", "16" if self.world == 1 else "0"))
        # a trainable modality tower (`feature_extractor` of the single-tower modalities, `experts` of the MoE ones): anything of a
        # modality that is not its projector(s)
        tail = any(p.requires_grad for mod in getattr(self.model, "modalities_with_projection", ())
                   for n, p in mod.named_parameters() if not n.startswith(("projector.", "projectors.")))
        layers = getattr(getattr(self.model.model, "model", None), "layers", None)
        side = os.environ.get("MM_WGRAD_SIDE", "0") == "1" and self.world == 1
        if ((n <= 0 or not tail) and not side) or layers is None or not torch.cuda.is_available():
            Fm.set_wgrad_deferral(None, ())
            return
        def firsts(ls):
            out = []
            for layer in ls:
                a, m = layer.self_attn, layer.mlp
                for group in (a._wqkv, Fm.as_group(a.o_proj.weight), m._wgu, Fm.as_group(m.down_proj.weight)):
                    if group.requires_grad:
                        out.append(group.params[0])
            return out

        if not tail:
            n = 0
        ids = firsts(list(layers)[:n])
        # MM_WGRAD_SIDE=1 (experiment): the wgrad GEMMs of the OTHER decoder layers (and an untied lm_head) are not held back but
        # launched at once on the side stream, beside the main stream's input-gradient chain
        now = []
        if os.environ.get("MM_WGRAD_SIDE", "0") == "1":
            now = firsts(list(layers)[n:])
            head = getattr(self.model.model, "lm_head", None)
            if head is not None and head.weight.requires_grad and head.weight is not self.model.model.get_input_embeddings().weight:
                now.append(head.weight)
        prio = int(os.environ.get("MM_WGRAD_SIDE_PRIO", "0"))
        self._wgrad_stream = self._side_stream("MM_DEFER_CUS", priority=prio)
        Fm.set_wgrad_deferral(self._wgrad_stream, ids, immediate=now)

    def _setup_early_gradnorm(self):
        """Global gradient norm (clip_grad_norm_): how the sum of squares of 16.7 GB of gradients is taken.

        DEFAULT: one sweep (`sumsq_kernel`, 5.4 TB/s = 3.1 ms) over the trainable ranges after backward, partials summed in a
        fixed order by the finish kernel.  Two ways to hide those 3 ms were built, measured on the 8B step with
        tools/step_ab.py (same box, same process, interleaved) and found to COST time; they stay behind switches:
          * MM_FUSED_NORM=1 -- the decoder's weight-gradient GEMMs (97 % of the parameters) and an untied lm_head report the sum of
            squares of what they store (`mm_gemm_sumsq`, per-workgroup slots: no atomics, bit-reproducible) and only the rest is
            swept: 399.6 vs 395.5 ms/step when the sum was taken inside the GEMM epilogue (an extra convert + FMA per stored
            element lengthened every wgrad GEMM by more than the sweep it replaced, and the inlined path cost every other GEMM
            1.6 %: csrc/mm_gemm.hip).  `mm_gemm_sumsq` is now the GEMM followed by a reduction pass over its output.
          * MM_EARLY_NORM=1 -- each decoder layer swept on a side stream as soon as its wgrads are enqueued, under the rest
            of backward: 409.2 vs 404.8 ms/step: the sweep's HBM reads slow the GEMMs they run beside.
        One GPU only in both cases: under data parallelism a gradient is final only after its bucket's all-reduce."""
        self._norm_chunks = [(s, e, None) for s, e, _ in self.ranges]       # (start, end, trigger param id or None)
        if self.shard_optim:       # after the reduce-scatter a rank holds final gradients for its own pieces only
            self._norm_chunks = [(a, b, None) for a, b, _, _, _, _ in self.pieces]
        self._norm_triggers: Dict[int, int] = {}
        self._norm_stream = None
        self._ss = None
        for seg in self.flat.segments:                  # slots of an earlier trainer on this model are void
            if hasattr(seg.param, "_mm_ss"):
                del seg.param._mm_ss
        layers = getattr(getattr(self.model.model, "model", None), "layers", None)
        # (one GPU only: with a sharded optimiser step, also at world == 1 under MM_FORCE_EXCHANGE, the norm is assembled over the
        # rank's PIECES, whose positions the sharded step indexes -- the two experiments would replace that list)
        fused = os.environ.get("MM_FUSED_NORM", "0") == "1" and self.flat.dtype == torch.bfloat16 and not self.shard_optim
        early = os.environ.get("MM_EARLY_NORM", "0") == "1" and not self.shard_optim
        if self.world > 1 or layers is None or not torch.cuda.is_available() or not (fused or early):
            self._alloc_norm_partials()
            return
        seg_of = {id(sg.param): sg for sg in self._trainable}
        covered = []                                     # (start, end, trigger id) of the ranges NOT swept at the end
        groups = []                                      # (first param, [params]) whose wgrad GEMM produces its own sum of squares
        for layer in layers:
            a, m = layer.self_attn, layer.mlp
            mats = [a.q_proj.weight, a.k_proj.weight, a.v_proj.weight, a.o_proj.weight, m.gate_proj.weight, m.up_proj.weight,
                    m.down_proj.weight]
            if not all(id(p) in seg_of for p in mats):
                continue
            sgs = [seg_of[id(p)] for p in mats]
            if any((x.end + 7) // 8 * 8 != y.start for x, y in zip(sgs[:-1], sgs[1:])):
                continue                                                   # not contiguous in the flat buffer: swept at the end
            if fused:
                groups += [mats[0:3], mats[3:4], mats[4:6], mats[6:7]]
            elif Fm._is_deferred(a.q_proj.weight):
                continue
            covered.append((sgs[0].start, (sgs[-1].end + 7) // 8 * 8, id(a.v_proj.weight)))     # q/k/v wgrad = the layer's last write
        head = getattr(self.model.model, "lm_head", None)
        emb = self.model.model.get_input_embeddings().weight
        if head is not None and head.weight is not emb and id(head.weight) in seg_of:
            sg = seg_of[id(head.weight)]
            covered.append((sg.start, (sg.end + 7) // 8 * 8, id(head.weight)))
            if fused:
                groups.append([head.weight])
        late = []
        for s0, e0, _ in self.ranges:
            cur = s0
            for a0, b0, _t in sorted(x for x in covered if s0 <= x[0] and x[1] <= e0):
                if a0 > cur:
                    late.append((cur, a0, None))
                cur = b0
            if cur < e0:
                late.append((cur, e0, None))
        if fused and groups:
            from .._lib import GEMM_TN
            sizes = [K.gemm_sumsq_slots(GEMM_TN, sum(p.shape[0] for p in g), g[0].shape[1], 4096) for g in groups]
            self._ss = torch.zeros(sum(sizes), dtype=torch.float32, device=self.flat.device)
            off = 0
            for g, n in zip(groups, sizes):
                g[0]._mm_ss = self._ss[off:off + n]        # functional.ParamGroup.sumsq_slots() hands it to the wgrad GEMM
                off += n
            self._norm_chunks = late
        elif early and covered:
            self._norm_chunks = covered + late
            self._norm_triggers = {t: i for i, (_, _, t) in enumerate(covered)}
            self._norm_stream = torch.cuda.Stream()
        self._alloc_norm_partials()

    def _alloc_norm_partials(self):
        self._norm_slots = []
        off = 0
        for s, e, _ in self._norm_chunks:
            nb = int(min(1024, max(1, (e - s) // 65536)))
            self._norm_slots.append((off, nb))
            off += nb
        # one buffer for the finish kernel: [partials of the swept chunks | the wgrad GEMMs' slots]
        n_ss = self._ss.numel() if self._ss is not None else 0
        buf = torch.zeros(off + n_ss, dtype=torch.float32, device=self.flat.device)
        if n_ss:
            old = self._ss
            self._ss = buf[off:]
            for seg in self._trainable:                         # re-point the views handed out above into the joint buffer
                v = getattr(seg.param, "_mm_ss", None)
                if v is not None and v.untyped_storage().data_ptr() == old.untyped_storage().data_ptr():
                    o0 = v.storage_offset() - old.storage_offset()
                    seg.param._mm_ss = self._ss[o0:o0 + v.numel()]
        self._norm_partial = buf
        self._norm_done = set()
        self._norm_armed = False

    def _norm_chunk(self, i):
        s, e, _ = self._norm_chunks[i]
        off, nb = self._norm_slots[i]
        K.gradnorm_partial(self.flat.grad[s:e], self._norm_partial[off:off + nb])
        self._norm_done.add(i)

    def _norm_on_ready(self, key: int):
        i = self._norm_triggers.get(key)
        if i is None or not self._norm_armed or i in self._norm_done:
            return
        ev = torch.cuda.Event()
        ev.record()                                   # everything enqueued so far (this layer's wgrads included)
        self._norm_stream.wait_event(ev)
        with torch.cuda.stream(self._norm_stream):
            self._norm_chunk(i)

    # ------------------------------------------------------------------ setup
    def _set_mode(self):
        m = self.model
        m.train()
        {TrainingMode.ALIGNMENT: m.freeze_for_alignment, TrainingMode.LM_ONLY: m.freeze_for_lm,
         TrainingMode.END2END: m.freeze_for_end2end, TrainingMode.FULL: m.unfreeze}[self.training_mode]()

    def _setup_state(self):
        flat = self.model.flat_params()
        flat.ensure_grad()
        self.flat = flat
        self.ranges = flat.trainable_ranges()            # [(start, end, decay)]
        if not self.ranges:
            raise ValueError("no trainable parameters in this training mode")
        dev = flat.device
        segs = [(id(seg.param), seg.start, seg.end) for seg in flat.segments if seg.param.requires_grad]
        self._trainable = [seg for seg in flat.segments if seg.param.requires_grad]
        import os
        force = bool(os.environ.get("MM_FORCE_EXCHANGE")) and self.dist is not None    # rehearse RCCL calls with 1 rank
        # transport of the buckets: torch.distributed's all_reduce (default) or the C-ABI communicator (train/comm.py)
        self.comm = None
        mode = os.environ.get("MM_COMM", "")
        if mode in ("abi", "abi-rsag") and (self.world > 1 or force) and flat.grad.is_cuda:
            from .comm import RcclComm
            self.comm = RcclComm(self.dist, self.pg, algo=1 if mode == "abi-rsag" else 0)
        elif mode not in ("", "torch"):
            raise ValueError(f"MM_COMM={mode!r}: expected 'torch', 'abi' or 'abi-rsag'")
        self.exchanger = GradExchanger(flat.grad, [(s, e) for s, e, _ in self.ranges], segs, self.bucket_elems,
                                       dist=self.dist if (self.world > 1 or force) else None, group=self.pg, force=force,
                                       comm=self.comm, sharded=self.shard_optim)
        # optimizer state: for the trainable ranges, packed back to back -- all of them (replicated), or this rank's pieces only
        if self.shard_optim:
            rank = self.exchanger.rank
            decay_of = lambda a: next(d for s0, e0, d in self.ranges if s0 <= a < e0)
            self.pieces = []                              # (start, end, decay, state offset, bucket index, replicated tail?)
            off = 0
            for bi, bk in enumerate(self.exchanger.buckets):
                (s0, e0), (ts, te) = bk.shard(rank, self.world)
                for a, b, tail in ((s0, e0, False), (ts, te, True)):
                    if b > a:
                        self.pieces.append((a, b, decay_of(bk.start), off, bi, tail))
                        off += b - a
            n_state = off
        else:
            self.pieces = None
            n_state = sum(e - s for s, e, _ in self.ranges)
        # bf16 models keep the fp32 master weight as (the bf16 parameter itself, an int16 remainder): `self.master` then holds the
        # remainders (mm_adamw_step_split: 26 B instead of 28 B of traffic per parameter, no second copy of the weights).  fp32
        # models (the parity path) and MM_ADAMW_SPLIT=0 keep a separate fp32 master.
        self.split_master = flat.dtype == torch.bfloat16 and flat.data.is_cuda and os.environ.get("MM_ADAMW_SPLIT", "1") != "0"
        self.master = (torch.zeros(n_state, dtype=torch.int16, device=dev) if self.split_master      # master == parameter: remainder 0
                       else torch.empty(n_state, dtype=torch.float32, device=dev))
        self.m = torch.zeros(n_state, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n_state, dtype=torch.float32, device=dev)
        self.state_off = []
        if self.shard_optim:
            if not self.split_master:
                for a, b, _, o, _, _ in self.pieces:
                    self.master[o:o + b - a].copy_(flat.data[a:b])
        else:
            off = 0
            for s, e, _ in self.ranges:
                if not self.split_master:
                    self.master[off:off + e - s].copy_(flat.data[s:e])          # device copy of the bf16 weights (plumbing)
                self.state_off.append(off)
                off += e - s
        if self.world > 1 and not os.environ.get("MM_GEMM_PERSIST"):
            # The GEMM's persistent grid is one resident workgroup per CU, each walking 1/256 of the tiles.  While an RCCL
            # kernel holds some CUs (its waves and an 8-wave GEMM workgroup do not fit one CU together), the workgroups that
            # find no CU start only when another one exits -- at the very end -- and the launch takes up to twice as long.
            # One tile per workgroup lets the dispatcher hand tiles to whichever CUs are free (measured neutral at 1 GPU).
            from .._lib import lib
            lib().mm_set_option(b"gemm_persist", 0)

    # ------------------------------------------------------------------ reference surface
    def compute_loss(self, model, inputs, return_outputs=False, **kwargs):
        """reference trainer.py:91-118"""
        labels = inputs["labels"]
        rows = None
        if self.loss_rows_only and model.training:
            # the step needs the loss, not the logits: final norm, lm_head and cross-entropy on the labelled rows only (exactly
            # the rows HF's loss keeps; functional.LossRows).  Known without a device sync for host labels / prefetched batches.
            rows = getattr(labels, "_mm_loss_rows", None)
            if rows is None and torch.is_tensor(labels) and not labels.is_cuda:
                from ..functional import LossRows
                rows = LossRows.from_host_labels(labels, self.flat.device)
        outputs = model(input_ids=inputs["input_ids"], attention_mask=inputs.get("attention_mask"), labels=labels,
                        position_ids=inputs["position_ids"], processed_multimodal_inputs=inputs["processed_multimodal_inputs"],
                        loss_rows=rows)
        return (outputs.loss, outputs) if return_outputs else outputs.loss

    # ------------------------------------------------------------------ one optimisation step
    def training_step(self, inputs: Dict[str, Any]) -> torch.Tensor:
        """forward + backward of one micro-batch; on the accumulation boundary also exchange, clip and AdamW.
        Returns the (detached, device) micro-batch loss."""
        first = self._micro == 0
        last = self._micro == self.accum - 1
        if first:
            self.flat.attach_grads(fresh=True)       # no memset: the first wgrad of the step overwrites
            if self._ss is not None:
                self._ss.zero_()                     # the wgrad GEMMs overwrite their own slots; the rest must read 0
        ex = self.exchanger
        ex.begin_step(exchange_this_step=last)
        self._norm_done = set()
        self._norm_armed = last and bool(self._norm_triggers)      # a gradient is final only in the last micro-batch

        def on_ready(p, _ex=ex.on_ready, _nr=self._norm_on_ready):
            _ex(id(p))
            _nr(id(p))

        Fm.set_grad_ready_hook(on_ready)
        try:
            with trace_range("forward"):
                loss = self.compute_loss(self.model, inputs)
            # mean over micro-batches and ranks (HF Trainer with model_accepts_loss_kwargs=False, trainer.py:80)
            self._wait_optimizer()      # the previous update still reads the flat gradient buffer on the side stream
            if loss.requires_grad:      # e.g. ALIGNMENT mode on a text-only micro-batch: nothing trainable is on the path
                with trace_range("backward(+gradient exchange)"):
                    loss.backward(gradient=torch.full_like(loss, 1.0 / (self.accum * self.world)))
                ev = Fm.flush_deferred_wgrads()      # also covers a backward that never reached the embedding splice
                if ev is not None:
                    torch.cuda.current_stream().wait_event(ev)
        finally:
            Fm.set_grad_ready_hook(None)
        self._micro += 1
        if last:
            # a parameter nothing wrote to this step (e.g. the vision tower on a text-only batch) still holds the previous
            # step's values because nothing memsets the flat buffer: zero exactly those slices
            for seg in self._trainable:
                if getattr(seg.param, "_mm_fresh", False):
                    self.flat.grad[seg.start:seg.end].zero_()
                    seg.param._mm_fresh = False
            with trace_range("exchange tail + clip + AdamW launch"):
                ex.finish_step()
                self._optimizer_step()
            self._micro = 0
        else:
            ex.finish_step()
        return loss.detach()

    # ------------------------------------------------------------------ optimiser, overlapped with the next forward
    def _setup_optimizer_pipeline(self):
        """AdamW is HBM-bound (30 B/param), the forward pass is MFMA-bound: run the update of step n on a side HIP stream
        while step n+1's forward runs, block by block in FORWARD order (vision tower, projector, embedding, decoder layers,
        final norm, lm_head).  A forward pre-hook on each block makes the compute stream wait for that block's update
        event, and the compute stream waits for the last event before the next backward writes gradients."""
        import re
        self._blocks = []          # [(module or None, [(start, end, decay, state_off)])] in forward order
        self._defer_from, self._deferred = None, None
        self._hooks = []
        self._pending = {}
        self._all_done = None
        if self.shard_optim:
            self._setup_sharded_pipeline()
            return
        if not self.overlap_optimizer:
            return
        state_off = {}
        for (s0, e0, _), off in zip(self.ranges, self.state_off):
            state_off[(s0, e0)] = off

        def st_off(start):
            for (s0, e0), off in state_off.items():
                if s0 <= start < e0:
                    return off + (start - s0)
            raise KeyError(start)

        def block_key(name):
            m = re.match(r"(.*?layers\.\d+)\.", name)
            return m.group(1) if m else name.rsplit(".", 1)[0]

        groups: Dict[str, List] = {}
        for seg in self._trainable:
            groups.setdefault(block_key(seg.name), []).append(seg)
        # MM_ADAMW_DEFER=1 (default off, measured slower): the decoder's blocks are not launched with the rest at the end of
        # the step but when the NEXT forward reaches the first decoder layer -- see _launch_deferred
        defer = os.environ.get("MM_ADAMW_DEFER", "0") == "1"
        self._defer_from: Optional[int] = None
        self._deferred = None

        def order(key):    # forward order: modality towers (embeddings, pre-norm, layers, projector), then the LLM
            llm = key.startswith("model.")
            m = re.search(r"layers\.(\d+)$", key)
            li = int(m.group(1)) if m else -1
            if not llm:
                rank = 3 if ".projector" in key else (2 if li >= 0 else (0 if "embeddings" in key else 1))
            else:
                rank = 10 if "embed_tokens" in key else (11 if li >= 0 else (12 if key.endswith(".norm") else 13))
            return (rank, li, key)

        for key in sorted(groups, key=order):
            segs = sorted(groups[key], key=lambda sg: sg.start)
            runs = []
            for sg in segs:                                   # merge adjacent slices with equal decay flag
                end = min((sg.end + 7) // 8 * 8, self.flat.numel)
                if runs and runs[-1][1] == sg.start and runs[-1][2] == sg.decay:
                    runs[-1][1] = end
                else:
                    runs.append([sg.start, end, sg.decay])
            self._blocks.append((self._hook_module(key), [(a, b, d, st_off(a)) for a, b, d in runs]))
            if self._defer_from is None and defer and order(key)[0] >= 11 and self._blocks[-1][0] is not None:
                self._defer_from = len(self._blocks) - 1            # first decoder layer: see _launch_deferred
        covered = sum(b - a for _, rs in self._blocks for a, b, _, _ in rs)
        assert covered == sum(e - s0 for s0, e, _ in self.ranges), "optimizer pipeline must cover every trainable range"
        self._opt_stream = self._side_stream("MM_ADAMW_CUS")
        self._pending: Dict[int, torch.cuda.Event] = {}
        self._all_done: Optional[torch.cuda.Event] = None
        self._hooks = []
        self._unfired: List[int] = []
        self._deferred_mods = set() if self._defer_from is None else {id(m) for m, _ in self._blocks[self._defer_from:] if m is not None}
        hooked = set()
        for mod, _ in self._blocks:
            if mod is not None and id(mod) not in hooked:      # several blocks may share one call site (ViT embeddings.*)
                hooked.add(id(mod))
                self._hooks.append(mod.register_forward_pre_hook(self._wait_block))

    def _setup_sharded_pipeline(self):
        """Sharded step: the unit of work is a BUCKET (update my share, then all-gather the bucket's parameters), in forward order
        of what the buckets hold (modality towers and projector, the decoder's norm weights, then its matrices layer by layer); a
        forward pre-hook on every block waits for the all-gathers of the buckets its parameters lie in."""
        import re
        ex = self.exchanger
        llm_lo = min((sg.start for sg in self._trainable if sg.name.startswith("model.")), default=None)
        llm_hi = max((sg.end for sg in self._trainable if sg.name.startswith("model.")), default=None)

        def order(i):
            bk = ex.buckets[i]
            llm = llm_lo is not None and llm_lo <= bk.start < llm_hi
            decay = next(d for s0, e0, d in self.ranges if s0 <= bk.start < e0)
            return (1 if llm else 0, 1 if (llm and decay) else 0, bk.start)

        self._bucket_order = sorted(range(len(ex.buckets)), key=order)
        self._opt_stream = self._side_stream("MM_ADAMW_CUS") if torch.cuda.is_available() and self.flat.device.type == "cuda" else None

        def block_key(name):
            m = re.match(r"(.*?layers\.\d+)\.", name)
            return m.group(1) if m else name.rsplit(".", 1)[0]

        groups: Dict[str, List] = {}
        for seg in self._trainable:
            groups.setdefault(block_key(seg.name), []).append(seg)
        self._block_buckets: Dict[int, List[int]] = {}
        self._eager_buckets = set()
        for key, segs in groups.items():
            mod = self._hook_module(key)
            idx = sorted({bi for sg in segs for bi, bk in enumerate(ex.buckets) if sg.start < bk.end and sg.end > bk.start})
            if mod is None:
                self._eager_buckets.update(idx)
            else:
                self._block_buckets.setdefault(id(mod), [])
                self._block_buckets[id(mod)] = sorted(set(self._block_buckets[id(mod)]) | set(idx))
                if not any(h is mod for h in getattr(self, "_hooked_mods", [])):
                    self._hooked_mods = getattr(self, "_hooked_mods", []) + [mod]
                    self._hooks.append(mod.register_forward_pre_hook(self._wait_block_sharded))
        self._gather_works: Dict[int, Any] = {}

    def _wait_block_sharded(self, mod, _inputs):
        for bi in self._block_buckets.get(id(mod), ()):
            w = self._gather_works.pop(bi, None)
            if w is not None:
                w.wait()

    def _adamw(self, a, b, off, lr, decay, total):
        """AdamW on flat[a:b] with the state slice starting at `off`."""
        n = b - a
        args = (self.m[off:off + n], self.v[off:off + n], lr, self.betas[0], self.betas[1], self.eps, self.wd if decay else 0.0, self.step_count)
        if self.split_master:
            K.adamw_step_split(self.flat.data[a:b], self.flat.grad[a:b], self.master[off:off + n], *args, clip=total)
        else:
            K.adamw_step(self.flat.data[a:b], self.flat.grad[a:b], self.master[off:off + n], *args, clip=total)

    def _state_pieces(self):
        """(start, end, state offset) of every slice the optimiser state covers, in state order."""
        if self.shard_optim:
            return [(a, b, off) for a, b, _, off, _, _ in self.pieces]
        return [(s0, e0, off) for (s0, e0, _), off in zip(self.ranges, self.state_off)]

    def master_fp32(self):
        """The fp32 master weights of the state slices, packed (what optimizer checkpoints hold)."""
        if not self.split_master:
            return self.master
        out = torch.empty(self.master.numel(), dtype=torch.float32, device=self.master.device)
        for a, b, off in self._state_pieces():
            out[off:off + b - a].copy_(K.master_join(self.flat.data[a:b], self.master[off:off + b - a]))
        return out

    def _optimizer_step_sharded(self, lr):
        """clip + AdamW on this rank's pieces, then the all-gathers (see _setup_sharded_pipeline)."""
        ex, g = self.exchanger, self.flat.grad
        self._norm_partial.zero_()                         # slots this rank does not count must read 0 in the sum over ranks
        for i, (a, b, _, _, _, tail) in enumerate(self.pieces):
            if tail and ex.rank != 0:
                continue                                   # every rank holds the tails: count them once
            self._norm_chunk(i)
        if self.world > 1:
            self.dist.all_reduce(self._norm_partial, op=self.dist.ReduceOp.SUM, group=self.pg)       # a few KB of fp32 partial sums
        total = K.gradnorm_finish(self._norm_partial, self.max_grad_norm if self.max_grad_norm else 0.0)
        self.last_grad_norm = total
        by_bucket: Dict[int, List] = {}
        for pc in self.pieces:
            by_bucket.setdefault(pc[4], []).append(pc)
        main = torch.cuda.current_stream() if self._opt_stream is not None else None
        if main is not None:
            self._opt_stream.wait_stream(main)

        def run():
            for bi in self._bucket_order:
                for a, b, decay, off, _, _ in by_bucket.get(bi, ()):
                    self._adamw(a, b, off, lr, decay, total)
                self._gather_works[bi] = ex.all_gather_params(self.flat.data, ex.buckets[bi])

        if self._opt_stream is not None:
            with torch.cuda.stream(self._opt_stream):
                run()
                self._all_done = torch.cuda.Event()
                self._all_done.record(self._opt_stream)
        else:
            run()
        for bi in self._eager_buckets:                     # parameters no forward hook guards: wait right away
            w = self._gather_works.pop(bi, None)
            if w is not None:
                w.wait()

    def _hook_module(self, key: str):
        """The module whose __call__ precedes every read of block `key`'s parameters in forward: the block's own module,
        or -- for bare parameter holders that forward never calls (`_mm_param_holder`: the ViT's patch_embedding /
        position_embedding, read by VisionEmbeddings.forward) -- the nearest ancestor that is called.  None = no such
        call site: the update is waited for eagerly."""
        while key:
            try:
                mod = self.model.get_submodule(key)
            except AttributeError:
                mod = None
            if mod is not None and not getattr(mod, "_mm_param_holder", False):
                return mod
            key = key.rsplit(".", 1)[0] if "." in key else ""
        return None

    def _launch_deferred(self):
        """MM_ADAMW_DEFER=1 (an experiment that LOST; off by default): the decoder's share of the update (97 % of the
        parameters, 43 ms of HBM streaming) launched from the NEXT forward at the moment its host code reaches the first decoder
        layer, ordered after everything the compute stream has been given so far, instead of at the end of the step.
        Motivation: a kernel trace (tools/stream_time.py) shows that AdamW's 2048 grid-striding workgroups, launched at the end
        of the step, fill every wave slot while the next step's ViT forward runs: each of the ViT's ~350 short kernels waits
        for slots (attention 402 us instead of 11, GELU 277 instead of 9, LayerNorm 134 instead of 12): 33 ms on the compute
        stream for 5 ms of work.  Measured (tools/step_ab.py, same process): 401.6 ms/step deferred vs 394.2 as is.  Beside the
        decoder's GEMMs the update costs MORE: the 256x256 GEMM's LDS-DMA stream shares HBM with 5 TB/s of optimizer traffic and
        loses about 0.8 ms per ms of update, wherever the update is placed; the ViT forward, being launch-latency-bound, is the
        cheapest thing to run it beside.  (A smaller update grid, `mm_set_option("adamw_blocks", 512)`: 388.6-393.8 vs 394.2,
        inside the noise; round 4: a LARGER grid -- 262 144, one vector per thread, now the library's default --
        is 8-18 % faster stand-alone and 2.7 ms per step: csrc/mm_optim.hip.)  The update therefore costs the step about 35 of its 44 ms on this chip; fewer bytes per parameter
        (28 now) is what would lower it."""
        upd = self._deferred
        if upd is None:
            return
        self._deferred = None
        main, side = torch.cuda.current_stream(), self._opt_stream
        ev0 = torch.cuda.Event()
        ev0.record(main)
        side.wait_event(ev0)
        with torch.cuda.stream(side):
            for mod, runs in self._blocks[self._defer_from:]:
                for s, e, decay, off in runs:
                    upd(s, e, decay, off)
                ev = torch.cuda.Event()
                ev.record(side)
                if mod is not None:
                    self._pending[id(mod)] = ev
            self._all_done = ev
        if any(mod is None for mod, _ in self._blocks[self._defer_from:]):
            main.wait_event(self._all_done)

    def _wait_block(self, mod, _inputs):
        if self._deferred is not None and id(mod) in self._deferred_mods:
            self._launch_deferred()
        ev = self._pending.pop(id(mod), None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def _wait_optimizer(self):
        """Everything the side stream still owes (before gradients are overwritten or parameters read ad hoc)."""
        if self.shard_optim:
            for bi in list(getattr(self, "_gather_works", {})):
                self._gather_works.pop(bi).wait()
            if self._all_done is not None:
                torch.cuda.current_stream().wait_event(self._all_done)
                self._all_done = None
            return
        if getattr(self, "_deferred", None) is not None:
            self._launch_deferred()                           # no forward reached the decoder since the last step
        if getattr(self, "_all_done", None) is not None:
            # blocks whose hook did not fire in the forward that just ran (legitimate for a tower the batch never entered;
            # a bug if forward read them: tests/test_trainer_gpu.py checks this list is empty for an image batch)
            self._unfired = list(self._pending)
            torch.cuda.current_stream().wait_event(self._all_done)
            self._all_done = None
            self._pending.clear()

    def _optimizer_step(self):
        self.step_count += 1
        lr = scheduled_lr(self.lr_scheduler_type, self.step_count - 1, self.max_steps, self.lr, self.min_lr, self.warmup)
        g = self.flat.grad
        self._norm_armed = False
        if self.shard_optim:
            self._optimizer_step_sharded(lr)
            return
        if self._norm_stream is not None and self._norm_done:
            torch.cuda.current_stream().wait_stream(self._norm_stream)     # the chunks summed under backward
        for i in range(len(self._norm_chunks)):
            if i not in self._norm_done:
                self._norm_chunk(i)
        total = K.gradnorm_finish(self._norm_partial, self.max_grad_norm if self.max_grad_norm else 0.0)
        self.last_grad_norm = total

        def upd(s, e, decay, off):
            self._adamw(s, e, off, lr, decay, total)

        if not self._blocks:
            for (s, e, decay), off in zip(self.ranges, self.state_off):
                upd(s, e, decay, off)
            return
        main = torch.cuda.current_stream()
        side = self._opt_stream
        side.wait_stream(main)                               # gradients (and their all-reduce) are complete
        now = self._blocks if self._defer_from is None else self._blocks[:self._defer_from]
        with torch.cuda.stream(side):
            ev = None
            for mod, runs in now:
                for s, e, decay, off in runs:
                    upd(s, e, decay, off)
                ev = torch.cuda.Event()
                ev.record(side)
                if mod is not None:
                    self._pending[id(mod)] = ev
            self._all_done = ev
        # blocks without a hookable module: wait for them right away
        if ev is not None and any(mod is None for mod, _ in now):
            main.wait_event(self._all_done)
        if self._defer_from is not None:
            self._deferred = upd                              # the decoder's blocks: launched by the next forward (_launch_deferred)

    def synchronize(self):
        """Make the compute stream wait for the in-flight optimiser update (call before reading parameters)."""
        self._wait_optimizer()
        if torch.cuda.is_available():
            K.embed_check_pending()          # and surface an out-of-range token id of the steps so far (IndexError, as torch)

    def close(self):
        """Detach from the model (forward hooks, deferred-wgrad registration, gradient-norm slots) and release the optimiser
        state, so that another trainer can be built on the same model (the hooks otherwise keep this one alive)."""
        self._wait_optimizer()
        for h in getattr(self, "_hooks", []):
            h.remove()
        self._hooks = []
        Fm.set_wgrad_deferral(None, ())
        Fm.set_grad_ready_hook(None)
        for seg in self.flat.segments:
            if hasattr(seg.param, "_mm_ss"):
                del seg.param._mm_ss
        self.master = self.m = self.v = self._norm_partial = self._ss = None
        self._blocks = []
        if getattr(self, "comm", None) is not None:
            self.comm.close()
            self.comm = self.exchanger.comm = None

    # ------------------------------------------------------------------ checkpoints (reference cli/train.py:186-195)
    def save_model(self, path: str, **kw):
        """Weights only, safe against the side-stream optimizer: waits for the in-flight AdamW before the copy."""
        self._wait_optimizer()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.model.save_pretrained(path, **kw)

    def _state_signature(self):
        sig = {"ranges": [[int(s), int(e), bool(d)] for s, e, d in self.ranges], "training_mode": self.training_mode.name,
               "numel": int(self.master.numel())}
        if self.shard_optim:      # a sharded state file belongs to one rank of one world size
            sig.update(world=int(self.world), rank=int(self.exchanger.rank), bucket_elems=int(self.bucket_elems))
        return sig

    def _state_file(self):
        if self.shard_optim:
            return f"optimizer_state.rank{self.exchanger.rank:05d}-of-{self.world:05d}.safetensors"
        return "optimizer_state.safetensors"

    def save_state(self, path: str):
        """HF-Trainer-style `checkpoint-N` directory: the model (reference layout) + what a bit-exact resume needs beyond it:
        the fp32 master weights and AdamW moments of the trainable ranges, the step counter and the schedule parameters
        (the reference gets the same from DeepSpeed's optimizer shards, config/deepspeed.json + trainer.train(
        resume_from_checkpoint=...)).  No RNG state: the path has no dropout."""
        import json
        from safetensors.torch import save_file
        rank = self.exchanger.rank if self.shard_optim else (self.dist.get_rank(self.pg) if self.dist else 0)
        if rank == 0:
            self.save_model(path)                 # the parameters are replicated: one copy
        else:
            self._wait_optimizer()
            os.makedirs(path, exist_ok=True)
        if rank != 0 and not self.shard_optim:
            return                                # replicated optimiser state: rank 0's files are everyone's
        save_file({"master": self.master_fp32().detach().cpu(), "exp_avg": self.m.detach().cpu(), "exp_avg_sq": self.v.detach().cpu()},
                  os.path.join(path, self._state_file()), metadata={"format": "pt"})
        with open(os.path.join(path, "trainer_state.json" if rank == 0 else f"trainer_state.rank{rank:05d}.json"), "w") as f:
            json.dump({"global_step": self.step_count, "micro_step": self._micro, "learning_rate": self.lr, "min_lr": self.min_lr,
                       "max_steps": self.max_steps, "warmup_steps": self.warmup, "weight_decay": self.wd, "betas": list(self.betas),
                       "eps": self.eps, "max_grad_norm": self.max_grad_norm, "gradient_accumulation_steps": self.accum,
                       "signature": self._state_signature()}, f, indent=2)

    def load_state(self, path: str, load_model: bool = True):
        """Resume from `save_state`: weights (streamed shard by shard), fp32 master / moments, step counter.  The master
        weights are authoritative: the bf16 parameters are re-derived from them, so the continued run is bit-identical
        to the uninterrupted one."""
        import json
        from safetensors import safe_open
        rank = self.exchanger.rank if self.shard_optim else 0
        st = json.load(open(os.path.join(path, "trainer_state.json" if rank == 0 else f"trainer_state.rank{rank:05d}.json")))
        if st["signature"] != self._state_signature():
            raise ValueError(f"{path}: optimizer state was saved for a different trainable set / training mode "
                             f"({st['signature']['training_mode']}, {st['signature']['numel']} elements)")
        self._wait_optimizer()
        if load_model:
            self.model.load_checkpoint_weights(path, strict=True)
        with safe_open(os.path.join(path, self._state_file()), framework="pt", device="cpu") as f:
            master = f.get_tensor("master").to(self.flat.device)
            self.m.copy_(f.get_tensor("exp_avg"))
            self.v.copy_(f.get_tensor("exp_avg_sq"))
        with torch.no_grad():      # the master weights are authoritative: bf16 parameters = round(master) (my pieces; the others' came with the model)
            for a, b, off in self._state_pieces():
                if self.split_master:
                    K.master_split(master[off:off + b - a].contiguous(), self.flat.data[a:b], self.master[off:off + b - a])
                else:
                    self.master[off:off + b - a].copy_(master[off:off + b - a])
                    self.flat.data[a:b].copy_(master[off:off + b - a])
        self.step_count = int(st["global_step"])
        self._micro = 0
        return st

    def train(self, batches: Optional[Iterable[Dict[str, Any]]] = None, max_steps: Optional[int] = None,
              resume_from_checkpoint: Optional[str] = None, per_device_train_batch_size: Optional[int] = None):
        """Minimal loop: iterate collated batches, or (batches=None) collate `train_dataset` with `data_collator` in chunks
        of `per_device_train_batch_size`, staged to the device by DevicePrefetcher.  `resume_from_checkpoint` = a
        `save_state` directory (reference cli/train.py:188-195)."""
        if resume_from_checkpoint:
            self.load_state(resume_from_checkpoint)
        if batches is None:
            if self.train_dataset is None or self.data_collator is None:
                raise ValueError("train(): give `batches`, or construct the trainer with train_dataset and data_collator")
            ds, bs, coll = self.train_dataset, per_device_train_batch_size or self.per_device_train_batch_size, self.data_collator
            rank = self.dist.get_rank(self.pg) if self.dist else 0

            def gen():      # contiguous chunks, strided over ranks (each rank sees a disjoint share)
                for i in range(rank * bs, len(ds) - bs + 1, bs * self.world):
                    yield coll([ds[j] for j in range(i, i + bs)])

            batches = gen()
            if torch.cuda.is_available():
                from .prefetch import DevicePrefetcher
                batches = DevicePrefetcher(batches, device=self.flat.device)
        losses = []
        steps = max_steps or self.max_steps or 0
        for i, batch in enumerate(batches):
            losses.append(self.training_step(batch))
            if steps and (i + 1) >= steps * self.accum:
                break
        if self.dist is not None:
            self.dist.barrier(group=self.pg)       # reference cli/train.py:200-201
        return losses
