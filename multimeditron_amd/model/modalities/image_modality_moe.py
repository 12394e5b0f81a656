"""`moe_meditron_clip` and `moe_meditron_clip_pep`: the mixture-of-experts image modalities (reference
model/modalities/image_modality_moe.py:10-246, image_modality_moe_pep.py:11-288 and model/attention.py:5-101) on libmmhip kernels.

E CLIP vision towers ("experts") run on every image; a gating network scores the image; one of three fusions combines the
experts' patch tokens; the MLP projector maps them into the LLM's embedding space:

    sequence_append   [n, E*P, C]  all experts' tokens one after the other                          (:165-168)
    weighted_average  [n, P, C]    sum_e w[n,e] * tokens_e                                           (:169-176)
    cross_attn        [n, P, C]    generalist tokens attend over the specialists' tokens, each scaled by the softmax of the
                                   specialists' gate weights (CrossAttention, attention.py:48-101)  (:177-203)

What runs where: the towers are this package's `VisionTransformer` (model/vision.py), the fusion is `mm_expert_fuse`, the
cross-attention is three biased GEMMs + the non-causal flash-attention kernel (Nq = P queries over (E-1)*P keys) + a GEMM.
Parameter names equal the reference's (`experts.{e}.*`, `cross_attn.{q,k,v}_proj / proj`, `projector.projection.{0,2,4}`).

`moe_meditron_clip_pep` (per-expert projection) gives every expert its own MLP projector (`projectors.{e}.projection.{0,2,4}`)
and fuses in the LLM's embedding space; its cross-attention therefore has `hidden_size / cross_attn_heads`-wide heads: 512 in the
shipped recipe (cookbook/sft/moe/*/attn/pep: 4096 / 8), 96 in the shared-projector one (768 / 8).  Neither is a flash-kernel
width; both run on the one-pass cross-attention kernel `mm_xattn_*` (csrc/mm_xattn.hip: up to 512 keys, head widths up to 512).

NOT built: the reference's `GatingNetwork` is a torchvision ResNet-50 (moe/gating.py:37-89); torchvision is
absent and a ResNet is outside the hot path.  `gating_network` is therefore a plug: any callable with the reference's output
contract `pixels [n,3,H,W] -> (logits [n,E], topk_indices, weights [n,E])`.  The experts run side by side, each on its own HIP
stream (`_run_experts`); a grouped multi-expert GEMM launch would be the step after that.  `CrossAttention`'s two dropouts
(p = 0.1 on the attention probabilities and on the output projection, active in the reference whenever the module trains) are
Philox-based: eval mode equals the reference, train mode equals it in distribution (torch's generator cannot be reproduced)."""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional

import torch
import torch.nn as nn

from ... import functional as Fm
from ...nn import Linear, grad_dummy
from ..constants import MODALITY_VALUE_KEY, NUM_EMBEDDINGS_KEY
from ..presets import resolve_preprocessor_config, resolve_vision_config
from ..projectors.mlp import MLPProjector
from ..vision import VisionConfig, VisionTransformer
from .base import AutoModality, BaseModality, BaseModalityConfig, BaseModalityProcessor
from .image_modality import ClipImagePreprocessor


class MOEImageConfig(BaseModalityConfig):
    """reference image_modality_moe.py:10-52 (same arguments)."""

    def __init__(self, hidden_size: int = 1024, use_bias_proj: bool = True, expert_clip_names: Optional[List[str]] = None,
                 image_processor: str = "openai/clip-vit-large-patch14", gating_path: str = "", top_k_experts: int = 1,
                 projection_type: str = "mlp", generalist_idx: int = -1, fusion_method: str = "weighted_average",
                 cross_attn_heads: int = 8, **kwargs):
        super().__init__(modality_type="image", hidden_size=hidden_size)
        self.use_bias_proj = use_bias_proj
        self.expert_clip_names = list(expert_clip_names or [])
        self.top_k_experts = top_k_experts
        self.gating_path = gating_path
        self.projection_type = projection_type
        self.image_processor = image_processor
        self.generalist_idx = generalist_idx
        self.fusion_method = fusion_method
        self.cross_attn_heads = cross_attn_heads


class MOEImageProcessor(BaseModalityProcessor):
    """reference image_modality_moe.py:55-88: CLIP preprocessing; the number of placeholder tokens depends on the fusion."""

    def __init__(self, config: MOEImageConfig):
        super().__init__(config)
        vis = VisionConfig.from_dict(resolve_vision_config(config.image_processor))
        self.image_processor = ClipImagePreprocessor(resolve_preprocessor_config(config.image_processor, vis.image_size))
        self._num_patches_per_entry = vis.num_patches
        self.top_k_experts = config.top_k_experts
        self.fusion_method = config.fusion_method

    def process(self, modality: Dict[str, Any]) -> Dict[str, Any]:
        out = modality.copy()
        out[MODALITY_VALUE_KEY] = self.image_processor(modality[MODALITY_VALUE_KEY])
        if self.fusion_method == "sequence_append":
            out[NUM_EMBEDDINGS_KEY] = self._num_patches_per_entry * self.top_k_experts
        elif self.fusion_method in ("weighted_average", "cross_attn"):
            out[NUM_EMBEDDINGS_KEY] = self._num_patches_per_entry
        else:
            raise ValueError(f"Unknown fusion_method: {self.fusion_method}")
        return out


class CrossAttention(nn.Module):
    """reference model/attention.py:5-101.  `attn_drop` acts on the attention probabilities and `proj_drop` on the output of
    `proj`, both only while the module trains (nn.Dropout semantics); the masks come from Philox streams keyed by torch's seed
    (functional.next_dropout_stream), so they differ from torch's own generator draw for draw: eval mode equals the reference,
    train mode equals it in distribution (tests/test_xattn_gpu.py holds the kernels to a torch restatement on the SAME mask)."""

    def __init__(self, dim: int, num_heads: int = 8, qkv_bias: bool = False, attn_drop: float = 0.1, proj_drop: float = 0.1,
                 dtype=None, device=None):
        super().__init__()
        assert dim > 0 and dim % num_heads == 0, "dim must be divisible by num_heads"
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.q_proj = Linear(dim, dim, bias=qkv_bias, dtype=dtype, device=device)
        self.k_proj = Linear(dim, dim, bias=qkv_bias, dtype=dtype, device=device)
        self.v_proj = Linear(dim, dim, bias=qkv_bias, dtype=dtype, device=device)
        self.proj = Linear(dim, dim, dtype=dtype, device=device)
        self.attn_drop_p, self.proj_drop_p = attn_drop, proj_drop
        self._wkv = Fm.ParamGroup([self.k_proj.weight, self.v_proj.weight])       # one GEMM for K and V
        self._bkv = Fm.ParamGroup([self.k_proj.bias, self.v_proj.bias]) if qkv_bias else None

    def forward(self, x: torch.Tensor, context: torch.Tensor) -> torch.Tensor:
        """x [n, Nq, C] queries, context [n, Nkv, C] (the specialists' tokens, already concatenated along the sequence)."""
        from ... import kernels as K
        n, Nq, C = x.shape
        Nkv = context.shape[1]
        h, d = self.num_heads, self.head_dim
        drop = self.attn_drop_p if self.training else 0.0
        q = self.q_proj(x.reshape(n * Nq, C))
        kv = Fm.linear(context.reshape(n * Nkv, C), self._wkv, self._bkv, dummy=grad_dummy(self.k_proj.weight))
        if K.xattn_supported(x.dtype, Nkv, d):
            # the one-pass kernel: any head width up to 512 (the shipped recipes: 768 / 8 = 96 and 4096 / 8 = 512), up to 1024 keys, dropout inside
            o = Fm.cross_attention(q, kv, n, Nq, Nkv, h, d, self.scale, drop)
        else:
            # more than 1024 keys: the flash kernels, narrower heads zero-padded to 64 / 128 as the SigLIP tower's are
            if drop > 0.0 and not getattr(CrossAttention, "_warned_no_attn_drop", False):
                import warnings
                CrossAttention._warned_no_attn_drop = True      # far outside the reference's recipes (P = 49, E = 5: 196 keys; four ViT-L/14
                                                                # experts: 1024, which mm_xattn_* takes with its dropout); say so, once
                warnings.warn(f"CrossAttention over {Nkv} keys runs on the flash kernels, which have no attention-probability dropout "
                              f"(mm_xattn_* holds <= 1024 keys per query): attn_drop = {drop} is NOT applied; proj_drop still is.")
            hw = Fm.attention_head_width(d, x.dtype)
            if hw != d:
                q, kv = Fm.head_pad(q, h, d, hw), Fm.head_pad(kv, 2 * h, d, hw)
            o = Fm.cross_attention(q, kv, n, Nq, Nkv, h, hw, self.scale, 0.0)
            if hw != d:
                o = Fm.head_strip(o, h, d, hw)
        o = self.proj(o)
        return Fm.dropout(o, self.proj_drop_p, self.training).view(n, Nq, C)


def _run_experts(experts, pixels, post=None):
    """Every expert tower on the same pixels (reference image_modality_moe.py:156-160), off the host-bound path two ways:
      * FROZEN towers (the shipped alignment / end2end recipes freeze them; also any no-grad call) are one captured hipGraph per
        image count: the ~210 launches per ViT-B/32 tower cost the host nothing at replay, and the towers sit on parallel branches
        of the graph (`_frozen_towers`; MM_MOE_GRAPH=0 disables).  The per-expert projectors (`post`, trainable) stay outside.
      * trainable towers run side by side, each on ITS OWN HIP stream: at the modality's size the towers are launch-latency-bound
        chains of short kernels, so E of them side by side take little more than one.  Autograd replays each tower's backward on
        the stream its forward ran on.  `post(e, tokens)` is applied on the expert's stream.  MM_MOE_STREAMS=0: one after the
        other on the current stream.
    Same kernels either way, same results bit for bit."""
    import os
    n = pixels.shape[0]

    def tower(e, expert, px=None):
        hs = expert(pixels if px is None else px).last_hidden_state                             # [n, 1+P, C]
        T = hs.shape[1]
        return Fm.drop_cls(hs.reshape(n * T, -1), n, T)                                          # [n, P, C]

    frozen = not torch.is_grad_enabled() or not any(p.requires_grad for ex in experts for p in ex.parameters())
    if frozen and pixels.is_cuda and os.environ.get("MM_MOE_GRAPH", "1") != "0":
        outs = _frozen_towers(experts, pixels, tower)
        return [post(e, o) if post is not None else o for e, o in enumerate(outs)]
    if (not frozen and pixels.is_cuda and not pixels.requires_grad and os.environ.get("MM_MOE_GRAPH", "1") != "0"
            and os.environ.get("MM_MOE_TRAIN_GRAPH", "1") != "0" and _graphable(experts)):
        outs = _trainable_towers(experts, pixels, tower)
        if outs is not None:
            return [post(e, o) if post is not None else o for e, o in enumerate(outs)]
        # None: a forward of this pixel shape still waits for its backward (two micro-batch losses summed before .backward(), a
        # grad-enabled evaluation in between): the graphs' saved activations are shared buffers, so THIS call takes the eager launches
    if len(experts) == 1 or not pixels.is_cuda or os.environ.get("MM_MOE_STREAMS", "1") == "0":
        return [post(e, tower(e, ex)) if post is not None else tower(e, ex) for e, ex in enumerate(experts)]
    main = torch.cuda.current_stream()
    outs = []
    for e, ex in enumerate(experts):
        st = _expert_stream(pixels.device, e)
        st.wait_stream(main)                                              # pixels (and the previous step's work) are ready
        with torch.cuda.stream(st):
            o = tower(e, ex)
            if post is not None:
                o = post(e, o)
        o.record_stream(main)                                             # produced on `st`, consumed on the compute stream
        outs.append((o, st))
    for _, st in outs:
        main.wait_stream(st)
    return [o for o, _ in outs]


def _frozen_towers(experts, pixels, tower):
    """Captured-graph forward of frozen expert towers: -> list of [n, P, C] token tensors (no autograd history).  One graph per
    (image count, image size, parameter storage); the pixels are copied into the graph's input buffer, the outputs out of its pool."""
    key = (tuple(pixels.shape), pixels.dtype, experts[0].embeddings.position_embedding.weight.data_ptr())
    cache = getattr(experts, "_mm_graphs", None)
    if cache is None:
        cache = experts._mm_graphs = {}
    ent = cache.get(key)
    if ent is None:
        with torch.no_grad():
            static_px = pixels.clone()
            cur = torch.cuda.current_stream()
            warm = torch.cuda.Stream()
            warm.wait_stream(cur)
            with torch.cuda.stream(warm):                                 # lazy initialisation (kernel attributes, id checks) outside capture
                for _ in range(2):
                    for e, ex in enumerate(experts):
                        tower(e, ex, static_px)
            cur.wait_stream(warm)
            graph = torch.cuda.CUDAGraph()
            with _no_gc(), torch.cuda.graph(graph):
                main = torch.cuda.current_stream()
                res = []
                for e, ex in enumerate(experts):                          # fork: every tower on its own branch of the graph
                    st = _expert_stream(pixels.device, e)
                    st.wait_stream(main)
                    with torch.cuda.stream(st):
                        res.append(tower(e, ex, static_px))
                for e in range(len(experts)):
                    main.wait_stream(_expert_stream(pixels.device, e))    # join
        ent = cache[key] = (graph, static_px, res)
        if len(cache) > 8:                                                # a handful of image counts at most: drop the oldest
            cache.pop(next(iter(cache)))
    graph, static_px, res = ent
    static_px.copy_(pixels)
    graph.replay()
    return [o.clone() for o in res]


class _no_gc:
    """No automatic garbage collection while a hipGraph is being captured: a collection that happens to run mid-capture can destroy
    an OLDER graph (an earlier model's towers) -- releasing a graph's memory pool is not a capturable operation and aborts the
    process.  (torch.cuda.graph collects once on entry; this keeps the interpreter from collecting again before the capture ends.)"""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()


def _graphable(experts):
    """Trainable towers can be replayed from captured graphs when every parameter's gradient lives in a flat buffer (the wgrad
    kernels write `param.grad` in place: a gradient tensor allocated during capture would belong to the graph's pool)."""
    return all(getattr(p, "_mm_flat", None) is not None for ex in experts for p in ex.parameters() if p.requires_grad)


class _TowerGraphs:
    """Forward and backward hipGraphs of E TRAINABLE expert towers for one pixel shape.  The forward graph is captured with
    autograd recording, on E parallel branches; the autograd graph of that capture is kept, and the backward graphs are captured
    by running it (per expert, on the expert's branch) -- one graph for `first gradient of the step` (the wgrad kernels overwrite)
    and one for `accumulate`.  A replay costs the host two calls instead of ~1,200 launches per ViT-L/14 tower, which is what
    bounds E trainable towers otherwise (E host-bound chains)."""

    def __init__(self, experts, pixels, tower):
        from ... import functional as F_
        import gc
        torch.cuda.synchronize()            # garbage (older graphs, their pools) goes now, at a quiet point, not in the middle of the
        gc.collect()                        # warm-up / capture / first replay below
        self.experts = experts
        self.params = [p for ex in experts for p in ex.parameters() if p.requires_grad]
        dev = pixels.device
        self.streams = [_expert_stream(dev, e) for e in range(len(experts))]
        self.static_px = pixels.detach().clone()
        cur = torch.cuda.current_stream()
        warm = torch.cuda.Stream()
        warm.wait_stream(cur)
        with torch.cuda.stream(warm), torch.no_grad():                      # lazy initialisation outside capture
            for e, ex in enumerate(experts):
                tower(e, ex, self.static_px)
        cur.wait_stream(warm)
        self.fwd = torch.cuda.CUDAGraph()
        with _no_gc(), torch.enable_grad(), torch.cuda.graph(self.fwd):
            main = torch.cuda.current_stream()
            self.res = []
            for e, ex in enumerate(experts):
                st = self.streams[e]
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    self.res.append(tower(e, ex, self.static_px))
            for st in self.streams:
                main.wait_stream(st)
        self.pool = self.fwd.pool()
        self.static_g = [torch.zeros_like(o) for o in self.res]
        self.bwd = {}
        self.warmed = False
        self._F = F_
        # The saved activations of the capture are SHARED buffers: a second forward before the first one's backward would overwrite
        # what that backward reads, and a second backward of one forward would re-read buffers a later forward has refilled.
        # `generation` counts forwards; `pending` = the generation whose backward has not run yet (None: the buffers are free).
        self.generation = 0
        self.pending = None

    def _flags(self, fresh):
        for p in self.params:
            p._mm_flat.ensure_grad()
            p.grad = p._mm_grad_view
            p._mm_fresh = fresh

    def _run_autograd(self):
        main = torch.cuda.current_stream()
        for e, st in enumerate(self.streams):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                torch.autograd.backward([self.res[e]], [self.static_g[e]], retain_graph=True)
        for st in self.streams:
            main.wait_stream(st)

    def forward(self, pixels):
        mods = [m for ex in self.experts for m in ex.modules() if m._forward_pre_hooks]      # looked up per call: a Trainer may come later
        for m in mods:                                                       # parameter-read hooks (the Trainer's per-block wait for
            kw = getattr(m, "_forward_pre_hooks_with_kwargs", {})            # the overlapped optimiser): the replay calls no module
            for hid, hook in list(m._forward_pre_hooks.items()):
                if hid in kw:
                    hook(m, (), {})
                else:
                    hook(m, ())
        self.static_px.copy_(pixels)
        self.fwd.replay()
        self.generation += 1
        self.pending = self.generation
        return [o.detach().clone() for o in self.res]

    def backward(self, grads, generation=None):
        if generation is not None and generation != self.pending:
            raise RuntimeError("MoE expert towers (graph replay): backward of a forward whose saved activations are gone -- it ran "
                               "twice, or another forward of the same pixel shape ran in between (MM_MOE_TRAIN_GRAPH=0 = eager towers)")
        self.pending = None
        F_ = self._F
        p0 = self.params[0]
        fresh = p0.grad is None or bool(getattr(p0, "_mm_fresh", False))
        for sg, g in zip(self.static_g, grads):
            if g is None:
                sg.zero_()
            else:
                sg.copy_(g)
        if not self.warmed:                 # first backward: eager, through the autograd graph of the capture (its saved tensors are
            self.warmed = True              # the graph's buffers, which the forward replay has just filled) -- a real backward
            self._flags(fresh)
            hook, order = F_._grad_ready_hook, []

            def spy(p):                     # which parameters report a finished gradient, and in which order: replays repeat it
                order.append(p)
                if hook is not None:
                    hook(p)

            F_.set_grad_ready_hook(spy)
            try:
                self._run_autograd()
            finally:
                F_.set_grad_ready_hook(hook)
            self.ready_order = order
            return
        g = self.bwd.get(fresh)
        if g is None:
            hook = F_._grad_ready_hook
            F_.set_grad_ready_hook(None)    # python side effects do not replay: the hooks are called after every replay instead
            try:
                self._flags(fresh)
                g = torch.cuda.CUDAGraph()
                with _no_gc(), torch.cuda.graph(g, pool=self.pool):
                    self._run_autograd()
            finally:
                F_.set_grad_ready_hook(hook)
            self.bwd[fresh] = g
        self._flags(False)                  # what grad_target leaves behind: .grad attached, the next write accumulates
        g.replay()
        for p in self.ready_order:
            F_._ready(p)


class _TrainableTowersFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dummy, pixels, graphs):
        ctx.graphs = graphs
        out = tuple(graphs.forward(pixels))
        ctx.generation = graphs.generation
        return out

    @staticmethod
    def backward(ctx, *grads):
        ctx.graphs.backward(grads, ctx.generation)
        return None, None, None


def _trainable_towers(experts, pixels, tower):
    """Graph-replayed forward + backward of trainable expert towers (see _TowerGraphs); same kernels as the eager path."""
    key = ("train", tuple(pixels.shape), pixels.dtype, experts[0].embeddings.position_embedding.weight.data_ptr(),
           tuple(p.requires_grad for ex in experts for p in ex.parameters()))
    cache = getattr(experts, "_mm_graphs", None)
    if cache is None:
        cache = experts._mm_graphs = {}
    ent = cache.get(key)
    if ent is None:
        ent = cache[key] = _TowerGraphs(experts, pixels, tower)
        if len(cache) > 8:
            cache.pop(next(iter(cache)))
    if ent.pending is not None:
        return None                          # its buffers hold a forward that still waits for its backward: this call runs eagerly
    p0 = ent.params[0]
    return list(_TrainableTowersFn.apply(grad_dummy(p0), pixels, ent))


_EXPERT_STREAMS = {}


def _expert_stream(device, e):
    key = (device.index or 0, e)
    if key not in _EXPERT_STREAMS:
        _EXPERT_STREAMS[key] = torch.cuda.Stream(device=device)
    return _EXPERT_STREAMS[key]


@AutoModality.register("moe_meditron_clip")             # applied last: the config class keeps the reference's model_type
@AutoModality.register("moe_meditron_clip_shared")      # alias: the name the shipped recipes use (cookbook/sft/moe/*/*/shared/config.yaml),
class MOEImageModality(BaseModality):                   # which the reference itself does not register (image_modality_moe.py:89)
    config_class = MOEImageConfig
    preprocessor_class = MOEImageProcessor

    def __init__(self, config: MOEImageConfig, dtype: torch.dtype = torch.bfloat16, device=None,
                 gating_network: Optional[Callable] = None):
        super().__init__(config, dtype=dtype)
        self.expert_names: List[str] = list(config.expert_clip_names)
        assert len(self.expert_names) > 0, "config.expert_clip_names must be non-empty"
        self.experts = nn.ModuleList()
        vis0 = None
        for name in self.expert_names:
            vis = VisionConfig.from_dict(resolve_vision_config(name))
            if vis0 is None:
                vis0 = vis
            elif (vis.hidden_size, vis.num_patches) != (vis0.hidden_size, vis0.num_patches):
                raise ValueError("every expert must produce the same [patches, width] token grid")      # reference :166 comment
            self.experts.append(VisionTransformer(vis, dtype, device))       # = the reference's `expert_model.vision_model`
        self.embedding_size = vis0.hidden_size
        self._num_patches_per_entry = vis0.num_patches
        self.generalist_idx = config.generalist_idx
        self.fusion_method = config.fusion_method.replace("-", "_")
        self.gating_network = gating_network
        names = list(getattr(getattr(gating_network, "config", None), "class_names", []) or [])
        if names:                                                        # reference :118-131: gate class order -> expert order
            lookup = {nm: i for i, nm in enumerate(self.expert_names)}
            try:
                perm = [lookup[nm] for nm in names]
            except KeyError as e:
                raise ValueError(f"Gating class name {e} not found in expert_clip_names: {self.expert_names}")
        else:
            perm = list(range(len(self.experts)))
        self.register_buffer("_gating_to_expert_perm", torch.tensor(perm, dtype=torch.long), persistent=False)
        self.projector = MLPProjector(self.embedding_size, config.hidden_size, dtype=dtype, device=device)
        if self.fusion_method == "cross_attn":
            self.cross_attn = CrossAttention(self.embedding_size, num_heads=config.cross_attn_heads, qkv_bias=True, dtype=dtype,
                                             device=device)
        self.modality_frozen = not self.training

    @property
    def device(self):
        return self.experts[0].embeddings.patch_embedding.weight.device

    def gate_weights(self, pixels: torch.Tensor) -> torch.Tensor:
        if self.gating_network is None:
            raise NotImplementedError(
                "MOEImageModality needs a gating network: the reference's GatingNetwork is a torchvision ResNet-50, which is not "
                "part of this build.  Assign `modality.gating_network = fn` with fn(pixels [n,3,H,W]) -> (logits, topk_indices, "
                "weights [n, E]) (reference moe/gating.py:73-89).")
        _logits, _topk, weights = self.gating_network(pixels)
        w = weights.to(device=self.device, dtype=torch.float32)
        return w.index_select(-1, self._gating_to_expert_perm.to(w.device)).contiguous()      # reference :171-173, :185-186

    def forward(self, inputs, stages=None) -> torch.Tensor:
        pixels = torch.stack(list(inputs), dim=0) if not torch.is_tensor(inputs) else inputs
        pixels = pixels.to(self.device, non_blocking=True)
        n, E = pixels.shape[0], len(self.experts)
        w = self.gate_weights(pixels)                                     # [n, E] fp32, expert order
        feats = _run_experts(self.experts, pixels)                        # every expert on every image (reference :156-160)
        stacked = torch.stack(feats, dim=0)                               # [E, n, P, C] (device copy; autograd unbinds it)
        P, C = stacked.shape[2], stacked.shape[3]
        if self.fusion_method == "sequence_append":
            fused = stacked.permute(1, 0, 2, 3).reshape(n, E * P, C)
        elif self.fusion_method == "weighted_average":
            fused = Fm.expert_fuse(stacked, w, list(range(E)), 0)
        elif self.fusion_method == "cross_attn":
            gi = self.generalist_idx % E
            spec = [i for i in range(E) if i != gi]
            ctx = Fm.expert_fuse(stacked, w, spec, 1)                     # [n, (E-1)*P, C], each specialist scaled by its weight
            fused = self.cross_attn(stacked[gi], ctx)
        else:
            raise ValueError(f"Unsupported fusion_method: {self.fusion_method}")
        out = self.projector(fused.contiguous())
        if stages is not None:
            stages["moe_fused"] = fused
            stages["projector_out"] = out
        return out

    def freeze_modality_embedder(self):
        for e in self.experts:
            for p in e.parameters():
                p.requires_grad = False
        self.modality_frozen = True

    def unfreeze_modality_embedder(self):
        for e in self.experts:
            for p in e.parameters():
                p.requires_grad = True
        self.modality_frozen = False

    def unfreeze_projection(self):
        for p in self.projector.parameters():
            p.requires_grad = True


class MOEImageConfigPEP(MOEImageConfig):
    """reference image_modality_moe_pep.py:11-52 (same arguments; its defaults)."""

    def __init__(self, hidden_size: int = 4096, use_bias_proj: bool = True, expert_clip_names: Optional[List[str]] = None,
                 image_processor: str = "openai/clip-vit-base-patch32", gating_path: str = "", top_k_experts: int = 5,
                 projection_type: str = "mlp", generalist_idx: int = -1, fusion_method: str = "weighted_average",
                 cross_attn_heads: int = 8, **kwargs):
        super().__init__(hidden_size=hidden_size, use_bias_proj=use_bias_proj, expert_clip_names=expert_clip_names,
                         image_processor=image_processor, gating_path=gating_path, top_k_experts=top_k_experts,
                         projection_type=projection_type, generalist_idx=generalist_idx, fusion_method=fusion_method,
                         cross_attn_heads=cross_attn_heads, **kwargs)


class MOEImageProcessorPEP(MOEImageProcessor):
    """reference image_modality_moe_pep.py:55-88 (identical to the shared-projector processor)."""


@AutoModality.register("moe_meditron_clip_pep")
class MOEImageModalityPEP(BaseModality):
    """reference image_modality_moe_pep.py:91-288: experts -> one projector PER expert -> fusion in the projected space."""
    config_class = MOEImageConfigPEP
    preprocessor_class = MOEImageProcessorPEP

    def __init__(self, config: MOEImageConfigPEP, dtype: torch.dtype = torch.bfloat16, device=None,
                 gating_network: Optional[Callable] = None):
        super().__init__(config, dtype=dtype)
        self.expert_names: List[str] = list(config.expert_clip_names)
        assert len(self.expert_names) > 0, "No experts provided in config.expert_clip_names."
        self.experts = nn.ModuleList()
        self.projectors = nn.ModuleList()
        vis0 = None
        for name in self.expert_names:
            vis = VisionConfig.from_dict(resolve_vision_config(name))
            if vis0 is None:
                vis0 = vis
            elif (vis.image_size, vis.patch_size) != (vis0.image_size, vis0.patch_size):      # reference :131-137
                raise ValueError("sequence_append requires identical (image_size, patch_size) across experts.")
            self.experts.append(VisionTransformer(vis, dtype, device))
            if config.projection_type != "mlp":
                raise ValueError(f"Unsupported projection_type: {config.projection_type}")
            self.projectors.append(MLPProjector(vis.hidden_size, config.hidden_size, dtype=dtype, device=device))
        self.embedding_size = config.hidden_size                         # post-projection width seen by the LLM (:252-254)
        self._num_patches_per_entry = vis0.num_patches
        self.generalist_idx = config.generalist_idx
        self.fusion_method = config.fusion_method.replace("-", "_")
        self.gating_network = gating_network
        names = list(getattr(getattr(gating_network, "config", None), "class_names", []) or [])
        if names:
            lookup = {nm: i for i, nm in enumerate(self.expert_names)}
            try:
                perm = [lookup[nm] for nm in names]
            except KeyError as e:
                raise ValueError(f"Gating class name {e} not found in expert_clip_names: {self.expert_names}")
        else:
            perm = list(range(len(self.experts)))
        self.register_buffer("_gating_to_expert_perm", torch.tensor(perm, dtype=torch.long), persistent=False)
        if self.fusion_method == "cross_attn":
            self.cross_attn = CrossAttention(config.hidden_size, num_heads=config.cross_attn_heads, qkv_bias=True, dtype=dtype,
                                             device=device)
        self.modality_frozen = not self.training

    @property
    def device(self):
        return self.experts[0].embeddings.patch_embedding.weight.device

    def forward(self, inputs, stages=None) -> torch.Tensor:
        pixels = torch.stack(list(inputs), dim=0) if not torch.is_tensor(inputs) else inputs
        pixels = pixels.to(self.device, non_blocking=True)
        n, E = pixels.shape[0], len(self.experts)
        if self.gating_network is None:
            raise NotImplementedError(
                "MOEImageModalityPEP needs a gating network: assign `modality.gating_network = fn` with fn(pixels [n,3,H,W]) -> "
                "(logits, topk_indices, weights [n, E]) (reference moe/gating.py:73-89; its ResNet-50 is not part of this build).")
        _logits, _topk, weights = self.gating_network(pixels)
        w_raw = weights.to(device=self.device, dtype=torch.float32).contiguous()          # gate order, as weighted_average uses it (:214)
        outs = _run_experts(self.experts, pixels, post=lambda e, tok: self.projectors[e](tok))      # [n, P, H] each
        stacked = torch.stack(outs, dim=0)                                # [E, n, P, H]
        P, H = stacked.shape[2], stacked.shape[3]
        if self.fusion_method == "sequence_append":
            fused = stacked.permute(1, 0, 2, 3).reshape(n, E * P, H)
        elif self.fusion_method == "weighted_average":
            fused = Fm.expert_fuse(stacked, w_raw, list(range(E)), 0)
        elif self.fusion_method == "cross_attn":
            gi = self.generalist_idx % E
            spec = [i for i in range(E) if i != gi]
            w_exp = w_raw.index_select(-1, self._gating_to_expert_perm.to(w_raw.device)).contiguous()     # expert order (:229-230)
            ctx = Fm.expert_fuse(stacked, w_exp, spec, 1)
            fused = self.cross_attn(stacked[gi], ctx)
        else:
            raise ValueError(f"Unsupported fusion_method: {self.fusion_method}")
        fused = fused.contiguous()
        if stages is not None:
            stages["moe_fused"] = fused
            stages["projector_out"] = fused
        return fused

    def freeze_modality_embedder(self):
        for e in self.experts:
            for p in e.parameters():
                p.requires_grad = False
        self.modality_frozen = True

    def unfreeze_modality_embedder(self):
        for e in self.experts:
            for p in e.parameters():
                p.requires_grad = True
        self.modality_frozen = False

    def unfreeze_projection(self):
        for p in self.projectors.parameters():
            p.requires_grad = True
