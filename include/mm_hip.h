/*
 * mm_hip.h -- C ABI of libmmhip.so: the MI355X (gfx950) kernels under the MultiMeditron
 * multimodal hot path  (modality encoder -> projector -> embed-splice -> LLM decoder, fwd+bwd).
 *
 * The reference (leagrieder/MultiMeditron) has no FFI: its operator boundary for this path is a
 * set of Python/torch calls.  Each entry point below replaces the torch/HF call(s) cited next to
 * it (paths under /root/reference/src/multimeditron/, `HF:` = transformers 5.15.0).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to a row-major buffer; no torch types cross the ABI
 *   - `stream` is a hipStream_t passed as void*; work is enqueued, never synchronised
 *   - nothing is allocated inside; the caller provides outputs and workspaces
 *   - return value: 0 = MM_OK, negative = error (mm_error_string); no exceptions, no aborts
 *   - dtype: MM_BF16 (storage bf16, fp32 accumulate) or MM_F32 (exact fp32; parity path)
 */
#ifndef MM_HIP_H
#define MM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { MM_BF16 = 0, MM_F32 = 1 } mm_dtype;

enum { MM_OK = 0, MM_ERR_ARG = -1, MM_ERR_ALIGN = -2, MM_ERR_UNSUPPORTED = -3, MM_ERR_LAUNCH = -4 };

/* GEMM operand layouts.  C is always [M,N] row-major (ldc). */
enum {
  MM_GEMM_NT = 0, /* A[M,K] (lda), B[N,K] (ldb): y = x W^T      -- nn.Linear forward            */
  MM_GEMM_NN = 1, /* A[M,K] (lda), B[K,N] (ldb): dx = dy W      -- nn.Linear input gradient     */
  MM_GEMM_TN = 2  /* A[K,M] (lda), B[K,N] (ldb): dW = dy^T x    -- nn.Linear weight gradient    */
};

/* GEMM epilogue flags (OR-ed) */
enum {
  MM_EPI_BIAS = 1,       /* + bias[N]                                      */
  MM_EPI_GELU_ERF = 2,   /* exact GELU (projectors/mlp.py:35,37)           */
  MM_EPI_QUICK_GELU = 4, /* x*sigmoid(1.702x) (HF:activations.py:117-123)  */
  MM_EPI_RESIDUAL = 8,   /* + residual[M,N] (ldr)                          */
  MM_EPI_ACCUMULATE = 16,/* C += result (gradient accumulation)            */
  MM_EPI_GELU_TANH = 32  /* tanh GELU (HF:activations.py gelu_pytorch_tanh; SigLIP MLP) */
};

int mm_version(void);
const char* mm_error_string(int code);
/* tuning / A-B switches (benchmarks and the trainer): "gemm_persist" 0|1 (persistent one-workgroup-per-CU grid; the
 * trainer sets 0 under data parallelism), "gemm_kernel" 0 auto | 1 register-staged 128x128 | 2..6 LDS-DMA tiles 256x128,
 * 256x256, 128x128, 64x128, 64x64, "gemm_small" -1 auto | 0..5, "gemm_tail" 0|1 (half-tile last round), "gemm_skinny"
 * 0|1 (M <= 16 weight-streaming kernel), "gemm_issue_waves" / "attn_issue_waves" 4|8 (how many of a workgroup's 8 waves
 * issue the LDS-DMA), "attn_fwd_waves" 8|4, "attn_dkv_pair" 0|1, "attn_dkv_res" 0|1|2 (round 4: the dK/dV kernel with resident K / V fragments on four waves, 2 = items pipelined inside the wave; slower, off); round 4: "gemm_w4" 0 | 1 | n (256x256 tiles on the 4-wave
 * hand-scheduled kernel gemm_bf16_w4_kernel: 0 = the 8-wave kernel, 1 = the shipped schedule, n = another schedule of
 * csrc/gen_gemm_w4.py: 4 / 5 split barriers, 6 / 7 = 1 / 4 with an L2 prefetch), "gemm_w4_big" 4|1 (schedule 4 where N or K >= 14336: default), "gemm_w4_rowmajor" 0|1 (its row-major, LDS-transposed epilogues), "gemm_w4_stream" 0|1 (wait-free plain
 * epilogue in the accumulator layout), "gemm_w4_shuffle" (0; 1 = the plain epilogue by register lane exchange instead of the LDS round trip: bit-identical, measured equal),
 * "gemm_w4_group_m", "gemm_w4_stagger" / "gemm_w4_stagger_slots" (experiments).  Unknown names return MM_ERR_ARG.   */
int mm_set_option(const char* name, int value);
/* current value of a "gemm_*" switch (so that a caller that flips one temporarily can restore what it found) */
int mm_get_option(const char* name, int* value);
int mm_attn_set_issue_waves(int v);

/* ---- GEMM: every nn.Linear / Conv2d(k=s) on the path -------------------------------------
 * replaces F.linear in mlp.py:33-39, HF:clip:280-384 (q/k/v/out/fc1/fc2), HF:clip:152-158 (patch conv as
 * GEMM), HF:llama:163-176,232-244,480 (gate/up/down, q/k/v/o, lm_head) and their autograd backward.
 * Requirements: lda, ldb, ldc, ldr multiples of 8 elements; pointers 16-byte aligned; for a K-contiguous
 * operand whose K is not a multiple of 8 the row padding up to the next multiple of 8 must hold zeros.   */
int mm_gemm(int dtype, int layout, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
            void* C, int ldc, const void* bias, const void* residual, int ldr, int epilogue, void* stream);

/* Fused q|k|v projection + RoPE: qkv[M,N] = x[M,K] . W[N,K]^T (+ bias[N]) with the first rope_cols columns -- heads of width
 * head_dim = 128 -- rotated in the GEMM's epilogue by the per-token tables cos_t / sin_t [M, head_dim/2] f32 of mm_rope_table
 * (HF:llama:232-244 q/k/v_proj followed by apply_rotary_pos_emb :113-160); the remaining columns (v) are stored unrotated.
 * Bit-identical to mm_gemm + mm_rope_apply, one pass over q|k less.  MM_BF16; MM_ERR_UNSUPPORTED for other head widths, N or
 * rope_cols not multiples of 128, or M < 256 (use the two launches).                                                    */
int mm_gemm_rope_fwd(int dtype, int M, int N, int K, const void* X, int ldx, const void* W, int ldw, const void* bias, void* QKV,
                     int ldqkv, int rope_cols, int head_dim, const float* cos_t, const float* sin_t, void* stream);
/* SwiGLU MLP front half in ONE GEMM: replaces `act_fn(gate_proj(x)) * up_proj(x)` (HF:llama:163-176) = two F.linear + silu +
 * mul.  Wgu = the fused [2I, K] gate|up weight (gate rows first); GU [M, 2I] receives the bf16 pre-activations (what the two
 * linears would store; kept for backward), ACT [M, I] = bf16(bf16(silu(gate)) * up): bit-identical to mm_gemm + mm_swiglu_fwd.
 * bf16 only; MM_ERR_UNSUPPORTED when I % 128, K % 64 or M < 256 (use the two-launch form then).                              */
int mm_gemm_swiglu_fwd(int dtype, int M, int I, int K, const void* X, int ldx, const void* Wgu, int ldw, void* GU, int ldgu,
                       void* ACT, int ldact, void* stream);
/* Linear + GELU forward for TRAINING in one launch: ACT = act(X W^T + bias) (+ residual), PRE = the bf16 pre-activation
 * X W^T + bias that the activation's backward needs (CLIP MLP fc1, HF:clip:368-372; projector, mlp.py:33-39).  epilogue = exactly one
 * of MM_EPI_GELU_ERF / QUICK_GELU / GELU_TANH, optionally | MM_EPI_BIAS | MM_EPI_RESIDUAL.  Rounding points as in mm_gemm +
 * mm_gelu_fwd (+ mm_add): bit-identical.  bf16 only, M > 16 (MM_ERR_UNSUPPORTED otherwise: use the separate launches).          */
int mm_gemm_act_fwd(int dtype, int M, int N, int K, const void* X, int ldx, const void* W, int ldw, const void* bias,
                    void* PRE, int ldpre, void* ACT, int ldact, const void* residual, int ldr, int epilogue, void* stream);
/* backward of down_proj's input and of the SwiGLU in one launch: dGU [M, 2I] = swiglu'(GU) * (dY [M, H] . Wd [H, I]), the
 * product d(act) staying in registers (autograd of HF:llama:163-176; bit-identical to mm_gemm NN + mm_swiglu_bwd).          */
int mm_gemm_swiglu_bwd(int dtype, int M, int I, int H, const void* dY, int lddy, const void* Wd, int ldw, const void* GU,
                       int ldgu, void* dGU, int lddgu, void* stream);

/* a GEMM that also returns the sum of squares of what it stores: the call overwrites the first n <= capacity partials and leaves
 * the others alone; on a buffer the caller zeroed beforehand they sum to sum(C^2) over the bf16 values written (after
 * MM_EPI_ACCUMULATE, the only epilogue allowed).  For the weight gradients, so that the global gradient norm of `max_grad_norm`
 * clipping (config_alignment.yaml:49 -> HF Trainer -> torch.nn.utils.clip_grad_norm_) can be assembled per GEMM.  Deterministic:
 * fixed slot per workgroup, no atomics.  Implemented as the GEMM followed by a reduction pass over C (the in-epilogue form cost
 * every GEMM more than it saved: csrc/mm_gemm.hip).  mm_gemm_sumsq_slots -> the capacity to provide for a problem.               */
int mm_gemm_sumsq_slots(int dtype, int layout, int M, int N, int K, int64_t* slots);
int mm_gemm_sumsq(int dtype, int layout, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                  int epilogue, float* partials, int64_t capacity, void* stream);

/* column sums: out[N] (+)= sum_m X[m,n]   (bias gradients)                                           */
int mm_colsum(int dtype, const void* X, int M, int N, int ldx, void* out, int accumulate, void* stream);

/* ids outside [0, vocab): *flag (device int, sticky) = 1.  nn.Embedding raises for them (model.py:433 embeds every id of the
 * batch before the splice); mm_embed_splice_fwd reads row 0 instead of out of bounds, so the caller checks this flag (the
 * Python layer reads it one call later, without a stall, and raises IndexError).                                           */
int mm_embed_check_ids(const int64_t* ids, int T, int64_t vocab, int* flag, void* stream);

/* ---- embed + modality splice: model.py:433-444 ---------------------------------------------------
 * out[t,:] = proj[src[t],:] if src[t] >= 0 else emb[ids[t],:]; src is built from (batch_idx, token_range)
 * by mm_splice_build_map (last writer wins, like index_put).                                         */
int mm_splice_build_map(const int64_t* batch_idx, const int64_t* token_range, int n_mod, int S, int T,
                        int32_t* src_map, void* stream);
int mm_embed_splice_fwd(int dtype, const void* emb, int64_t vocab, int H, const int64_t* ids, const void* proj,
                        const int32_t* src_map, int T, void* out, void* stream);
/* token order for the embedding gradient (depends on ids / src_map only, so it is built at forward time): a stable
 * sort of the T tokens by id; tokens overwritten by a modality row or with an id outside [0, vocab) sort last and get no
 * gradient.  order/skey: int32 [order_elems]; key_ws: int32 [T] workspace; sizes from mm_embed_sort_sizes.            */
int mm_embed_sort_sizes(int T, int H, int64_t* order_elems, int64_t* scratch_floats);
int mm_embed_sort(const int64_t* ids, const int32_t* src_map, int T, int64_t vocab, int32_t* key_ws, int32_t* order,
                  int32_t* skey, void* stream);
/* backward (autograd of model.py:433-444): dproj[i,:] = dE[pos(i),:]; demb[id,:] (+)= sum of dE[t,:] over the tokens t
 * with ids[t] == id that were NOT overwritten -- summed in fp32 in ascending token order, rounded once, one write per
 * touched row: bitwise reproducible, no atomics.  accumulate = 0 overwrites the touched rows (the caller has zeroed
 * demb), 1 adds to what demb holds (tied lm_head gradient, gradient accumulation).  scratch: fp32 [scratch_floats].   */
int mm_embed_splice_bwd(int dtype, const void* dE, int H, const int64_t* ids, const int32_t* src_map, int T,
                        const int64_t* batch_idx, const int64_t* token_range, int n_mod, int S, void* dproj,
                        void* demb, int64_t vocab, const int32_t* order, const int32_t* skey, float* scratch,
                        int accumulate, void* stream);

/* ---- ViT patch embedding glue: HF:clip:138-218 -----------------------------------------------------
 * patchify: pixels f32 [n,3,Himg,Wimg] -> patches [n*P, Kpad] (k = c*ps*ps + py*ps + px, zero padded)     */
int mm_patchify(int dtype, const float* pixels, int n, int himg, int wimg, int ps, int kpad, void* patches, void* stream);
/* x[n,0,:] = cls + pos[0]; x[n,1+p,:] = patch_out[n*P+p,:] + pos[1+p]                                    */
int mm_vit_embed_fwd(int dtype, const void* patch_out, const void* cls, const void* pos, int n, int P, int D,
                     void* x, void* stream);
/* dpatch_out = dx[:,1:,:]; dcls (+)= sum_n dx[n,0]; dpos (+)= sum_n dx[n]                                  */
int mm_vit_embed_bwd(int dtype, const void* dx, int n, int P, int D, void* dpatch_out, void* dcls, void* dpos,
                     int accumulate, void* stream);
/* dst[n,P,D] = src[n,1+P,D][:,1:,:] (image_modality.py:133) and its adjoint (zero CLS row)                */
/* ---- plug-in towers without a CLS token and with head_dim outside {64,128} (SigLIP-so400m: 16 heads x 72) ----------
 * mm_bcast_add: y[n,L] = x[n,L] + b[L]  (learned positions added to every image; HF:siglip SiglipVisionEmbeddings)
 * mm_head_pad:  inverse = 0: dst[rows, nheads*dpad] = src[rows, nheads*d] with each head zero-padded to dpad;
 *               inverse = 1: dst[rows, nheads*d] = the first d columns of every head of src[rows, nheads*dpad].
 *               Zero columns change neither q.k nor p.v, so attention on the padded heads is exact.                  */
int mm_bcast_add(int dtype, const void* x, const void* b, int n, int64_t L, void* y, void* stream);
int mm_head_pad(int dtype, const void* src, int64_t rows, int nheads, int d, int dpad, void* dst, int inverse, void* stream);
int mm_drop_cls_fwd(int dtype, const void* src, int n, int P, int D, void* dst, void* stream);
int mm_drop_cls_bwd(int dtype, const void* ddst, int n, int P, int D, void* dsrc, void* stream);
/* Loss rows (HF:loss/loss_utils.py:36-71 ignores labels == -100; llama modeling's lm_head + loss_function call): the training
 * step computes the final norm, lm_head and the loss only on the rows that carry a label.  dst[r, :D] = src[map[r], :D], or
 * zeros where map[r] < 0 (or >= n_src).  Forward: map = indices of the labelled rows; backward: map = the inverse map.       */
int mm_rows_select(int dtype, const void* src, int64_t ld_src, const int* map, int n_src, int n_dst, int D, void* dst,
                   int64_t ld_dst, void* stream);

/* ---- norms ------------------------------------------------------------------------------------------
 * RMSNorm: HF:llama:53-70.  rstd[M] f32 is saved for backward.                                            */
int mm_rmsnorm_fwd(int dtype, const void* x, const void* w, int M, int H, float eps, void* y, float* rstd, void* stream);
/* dx = rstd*(g - xhat*mean(g*xhat)) (+ dres), g = dy*w;  dw_partial[nblk,H] f32 (nblk = mm_norm_bwd_blocks(M)).
 * dres (optional, [M,H]) is the gradient arriving through the residual branch that shares x: fusing the add here
 * replaces autograd's separate accumulation pass.                                                              */
int mm_rmsnorm_bwd(int dtype, const void* dy, const void* x, const void* w, const float* rstd, int M, int H,
                   void* dx, float* dw_partial, const void* dres, void* stream);
/* LayerNorm: HF:clip:338-339,608 (nn.LayerNorm).  mean/rstd [M] f32 saved.                                 */
int mm_layernorm_fwd(int dtype, const void* x, const void* w, const void* b, int M, int H, float eps, void* y,
                     float* mean, float* rstd, void* stream);
int mm_layernorm_bwd(int dtype, const void* dy, const void* x, const void* w, const float* mean, const float* rstd,
                     int M, int H, void* dx, float* dw_partial, float* db_partial, const void* dres, void* stream);
int mm_norm_bwd_blocks(int M);
/* out[H] (+)= sum_b partial[b,H]  (f32 partials -> param-dtype gradient)                                    */
int mm_reduce_partials(int dtype, const float* partial, int nblk, int H, void* out, int accumulate, void* stream);
/* the same for two (partials, output) pairs in one launch: LayerNorm's dw and db */
int mm_reduce_partials2(int dtype, const float* partial0, const float* partial1, int nblk, int H, void* out0, void* out1, int accumulate0,
                        int accumulate1, void* stream);

/* ---- RoPE: HF:llama:113-160 (rotate_half form) --------------------------------------------------------
 * cos/sin tables [T, D/2] f32 from position_ids and inv_freq (HF:llama:113-127; llama3 scaling is applied by
 * the caller to inv_freq, HF:modeling_rope_utils.py:641-662).  round_bf16: round cos/sin to bf16 (HF casts them
 * to the activation dtype).                                                                               */
int mm_rope_table(const int64_t* position_ids, const float* inv_freq, int T, int half, int round_bf16, float* cos_t,
                  float* sin_t, void* stream);
/* in place on x viewed as [T, nheads, D] with row stride ld (elements); inverse=1 applies the adjoint          */
int mm_rope_apply(int dtype, void* x, int T, int nheads, int D, int ld, const float* cos_t, const float* sin_t,
                  int inverse, void* stream);

/* ---- attention: HF:llama:191-213 (eager softmax attention, GQA via repeat_kv), HF:clip:280-334 ------------
 * q [B,Sq,Hq,D], k/v [B,Skv,Hkv,D] with element strides (batch, seq, head); D contiguous; D in {64,128} for
 * MM_BF16 (MFMA path), any D<=256 for MM_F32.  key_mask [B,Skv] int64 (1 = attend) or NULL.  causal aligns the
 * LAST query with the LAST key (q position = i + Skv - Sq).  out [B,Sq,Hq,D] contiguous; lse [B,Hq,Sq] f32.     */
int mm_attn_fwd(int dtype, const void* q, const void* k, const void* v, int B, int Sq, int Skv, int Hq, int Hkv, int D,
                int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t k_sb, int64_t k_ss, int64_t k_sh, int64_t v_sb,
                int64_t v_ss, int64_t v_sh, const int64_t* key_mask, int causal, float scale, void* out, float* lse,
                void* stream);
/* dq/dk/dv use the SAME strides as q/k/v.  delta [B,Hq,Sq] f32 workspace.  dk/dv are overwritten.
 * For MM_F32 dk/dv must be zero-filled by the caller (atomic accumulation).                                   */
int mm_attn_bwd(int dtype, const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                int B, int Sq, int Skv, int Hq, int Hkv, int D, int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t k_sb,
                int64_t k_ss, int64_t k_sh, int64_t v_sb, int64_t v_ss, int64_t v_sh, const int64_t* key_mask, int causal,
                float scale, void* dq, void* dk, void* dv, float* delta, void* stream);

/* Decode step: mm_rope_apply on the q and k heads of x [T, (Hq+2Hkv)*D] (row stride ld) plus, in the same pass, the append
 * of the roped k heads and the v heads to the KV cache (HF cache update inside LlamaAttention.forward, HF:llama:247-252):
 * kdst / vdst point at the cache row of this step for sequence 0 ([Hkv, D] contiguous), dstride = elements between
 * consecutive sequences' rows (one token per sequence: T = batch).                                                   */
int mm_rope_append(int dtype, void* x, int T, int Hq, int Hkv, int D, int ld, const float* cos_t, const float* sin_t,
                   void* kdst, void* vdst, int64_t dstride, void* stream);

/* ---- decode-step fusions on the weight-streaming GEMM (KV-cache decode of `generate`, reference model.py:595-602; M = batch <= 16;
 * MM_BF16).  Each replaces a GEMM + the tiny launches around it in HF's LlamaDecoderLayer (HF:llama:284-325) with the arithmetic
 * and rounding points of the separate kernels (same bits).  `in_norm_w` (may be NULL) / `eps`: the RMSNorm IN FRONT of the projection
 * (input_layernorm, post_attention_layernorm, the model's final norm), applied to x while it is staged: y = x' W^T with
 * x' = RMSNorm(x) * in_norm_w.  M * K * 2 bytes of x must fit 144 KB of LDS (MM_ERR_UNSUPPORTED otherwise: use the separate kernels).
 * mm_decode_gateup_swiglu:   ACT[M,I] = silu(x' Wg^T) * (x' Wu^T), Wgu = fused [2I,K] gate|up weight    (= [mm_rmsnorm_fwd +] mm_gemm + mm_swiglu_fwd)
 * mm_decode_qkv_rope_append: QKV[M,(Hq+2Hkv)*128] = x' W^T (+ bias), q and k heads rotated with cos_t / sin_t [M,64], roped k and v
 *                            appended to the KV cache rows kdst / vdst (+ m * dstride)                  (= [mm_rmsnorm_fwd +] mm_gemm + mm_rope_append)
 * mm_decode_linear:          C[M,N] = x' W^T (+ bias) (+ residual)                                      (= [mm_rmsnorm_fwd +] mm_gemm)             */
int mm_decode_gateup_swiglu(int dtype, int M, int I, int K, const void* X, int ldx, const void* Wgu, int ldw, void* ACT, int ldact,
                            const void* in_norm_w, float eps, void* stream);
int mm_decode_qkv_rope_append(int dtype, int M, int Hq, int Hkv, int D, int K, const void* X, int ldx, const void* W, int ldw,
                              const void* bias, void* QKV, int ldqkv, const float* cos_t, const float* sin_t, void* kdst, void* vdst,
                              int64_t dstride, const void* in_norm_w, float eps, void* stream);
int mm_decode_linear(int dtype, int M, int N, int K, const void* X, int ldx, const void* W, int ldw, const void* bias,
                     const void* residual, int ldr, void* C, int ldc, const void* in_norm_w, float eps, void* stream);

/* KV-cache decode step: ONE query token per sequence over the cached keys (reference model.py:595-602 calling the LLM with
 * past_key_values; HF:llama:217-281 with q_len = 1).  q [B,Hq,D] (strides q_sb,q_sh; D contiguous), k/v [B,Skv,Hkv,D] as
 * in mm_attn_fwd, key_mask [B,Skv] or NULL, out [B,Hq,D] contiguous.  MM_BF16, D in {64,128}, Hq/Hkv in {1,2,4,7,8}.
 * workspace: f32 [B*Hq*nsplit*(D+2)] with nsplit = mm_attn_decode_splits(B,Hkv,Skv) (slices of the keys, merged in a
 * fixed order).  sync: int [B*Hkv], zero on entry and left zero -- the slice that arrives last does the merge inside
 * the same launch; NULL = merge in a second launch.  Rows with no visible key give 0.                              */
int mm_attn_decode_splits(int B, int Hkv, int Skv);
int mm_attn_decode(int dtype, const void* q, const void* k, const void* v, int B, int Skv, int Hq, int Hkv, int D,
                   int64_t q_sb, int64_t q_sh, int64_t k_sb, int64_t k_ss, int64_t k_sh, int64_t v_sb, int64_t v_ss,
                   int64_t v_sh, const int64_t* key_mask, float scale, void* out, float* workspace, int nsplit, int* sync,
                   void* stream);

/* ---- activations -----------------------------------------------------------------------------------------
 * SwiGLU: HF:llama:163-176.  gu [M, 2I] = [gate | up] from the fused gate/up GEMM; out [M,I] = silu(gate)*up     */
int mm_swiglu_fwd(int dtype, const void* gu, int M, int I, void* out, void* stream);
int mm_swiglu_bwd(int dtype, const void* gu, const void* dout, int M, int I, void* dgu, void* stream);
/* kind: 0 = erf GELU (mlp.py:35,37), 1 = quick GELU (HF:clip fc1).  x is the pre-activation.                       */
/* kind: 0 = erf GELU, 1 = quick GELU, 2 = tanh GELU */
int mm_gelu_fwd(int dtype, int kind, const void* x, int64_t n, void* y, void* stream);
int mm_gelu_bwd(int dtype, int kind, const void* x, const void* dy, int64_t n, void* dx, void* stream);
/* y = a + b (residual adds that are not fused into a GEMM epilogue)                                             */
int mm_add(int dtype, const void* a, const void* b, int64_t n, void* y, void* stream);

/* ---- loss: HF:loss/loss_utils.py:36-71 --------------------------------------------------------------------------
 * logits [T, ld] (V valid columns); labels already shifted by the caller; ignore_index = -100.
 * fwd: lse[T] f32, loss_row[T] f32 (0 for ignored rows).  loss = sum(loss_row)/count is reduced by mm_ce_reduce.    */
int mm_ce_fwd(int dtype, const void* logits, int T, int V, int ld, const int64_t* labels, float* lse, float* loss_row, void* stream);
/* out[0] = sum(loss_row)/max(count,1), out[1] = count (number of labels != -100)                                   */
int mm_ce_reduce(const float* loss_row, const int64_t* labels, int T, float* out, void* stream);
/* dlogits[t,v] = (exp(logit - lse[t]) - [v==label]) * gscale[0] / count, 0 for ignored rows and for v in [V, ld)    */
int mm_ce_bwd(int dtype, const void* logits, int T, int V, int ld, const int64_t* labels, const float* lse,
              const float* loss_and_count, const float* gscale, void* dlogits, void* stream);
/* next-token selection of model.py:607-621: argmax(softmax(logits/T)) over the LAST dim, first max wins          */
int mm_argmax_softmax(int dtype, const void* logits, int rows, int V, int ld, float temperature, int64_t* out, void* stream);
/* the same selection for long rows, the vocabulary cut into chunks over many workgroups (3 short launches instead of one block per
 * row sweeping 128 258 logits three times); ws: mm_argmax_softmax_ws_bytes(rows, V) bytes, 8-byte aligned                     */
int mm_argmax_softmax_ws_bytes(int rows, int V);
int mm_argmax_softmax_split(int dtype, const void* logits, int rows, int V, int ld, float temperature, int64_t* out, void* ws,
                            void* stream);
/* generate()'s per-token bookkeeping ON the device (the reference syncs per token: model.py:618-625,637-638): id = finished[b]
 * ? eos : tok[b]; finished[b] |= id == eos; out[b, col] = id; next_ids[b] = id (the next step's embedding lookup).       */
int mm_decode_select(const int64_t* tok, unsigned char* finished, int64_t eos, int B, int64_t* out, int ld_out, int col,
                     int64_t* next_ids, void* stream);

/* ---- MoE image modality: gating-weighted fusion of the experts' token features (modalities/image_modality_moe.py:163-205)
 * X [E, n, L] (L = P*C, expert-major), gate [n, E] fp32 (the gating network's softmax weights, expert order), idx[J] = the
 * experts taking part (host array).  mode 0: out[n, L] = sum_j gate[n, idx[j]] * X[idx[j], n]  (`weighted_average`, :170-176);
 * mode 1: out[n, J, L] = softmax_j(gate[n, idx[.]])[j] * X[idx[j], n]  (the specialists' scaled contexts of `cross_attn`,
 * :186-199).  backward = 1: X is d(out), out is dX [E, n, L] (only the listed experts' slices are written).                */
int mm_expert_fuse(int dtype, int backward, int mode, const void* X, const float* gate, const int* idx, int J, int E, int n,
                   int64_t L, void* out, void* stream);

/* ---- MoE image modality: the core of `CrossAttention` (model/attention.py:79-96: softmax(q k^T * scale) -> attn_drop -> @ v,
 * called with the generalist's P tokens as queries over the specialists' (E-1)*P tokens, image_modality_moe.py:177-203 and
 * image_modality_moe_pep.py:216-244) for head widths the flash kernels do not tile: any D that is a multiple of 8 up to 512
 * (MM_BF16; the shipped recipes need 96 = 768/8 and 512 = 4096/8), any D for MM_F32; Nkv <= 1024 (the whole score row of a
 * query stays in registers: no online softmax).  q [n,Nq,H,D], k/v [n,Nkv,H,D] with element strides (image, token, head);
 * out [n,Nq,H,D] contiguous; lse [n,H,Nq] f32.  drop_p = probability of zeroing an attention weight (`attn_drop`, 0.1 in the
 * reference whenever the module trains; 0 = eval): Philox4x32-10 with key `seed` and counter (call, offset); weight (row, key)
 * with row = (image * H + head) * Nq + query takes component key & 3 of call row * KP/4 + key/4, KP = Nkv rounded up to 32
 * (mm_dropout_mask(seed, offset, rows * KP, p) lists the same keep flags).  Backward regenerates the mask, is deterministic
 * (no atomics) and needs a workspace of mm_xattn_ws_bytes; dq/dk/dv take the strides of q/k/v.                            */
int mm_xattn_ws_bytes(int dtype, int n, int Nq, int Nkv, int H, int64_t* bytes);
int mm_xattn_fwd(int dtype, const void* q, const void* k, const void* v, int n, int Nq, int Nkv, int H, int D, int64_t q_sb,
                 int64_t q_ss, int64_t q_sh, int64_t k_sb, int64_t k_ss, int64_t k_sh, int64_t v_sb, int64_t v_ss, int64_t v_sh,
                 float scale, float drop_p, int64_t seed, int64_t offset, void* out, float* lse, void* stream);
int mm_xattn_bwd(int dtype, const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                 int n, int Nq, int Nkv, int H, int D, int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t k_sb, int64_t k_ss,
                 int64_t k_sh, int64_t v_sb, int64_t v_ss, int64_t v_sh, float scale, float drop_p, int64_t seed, int64_t offset,
                 void* dq, void* dk, void* dv, void* ws, int64_t ws_bytes, void* stream);
/* y[i] = keep(i) ? x[i] / (1 - p) : 0 -- nn.Dropout of CrossAttention's output projection (attention.py:41,98, `proj_drop`);
 * element i takes component i & 3 of Philox call i / 4.  Its backward is the same call on dy with the same (seed, offset).
 * mm_dropout_mask: the keep flags (uint8 0/1) of elements 0 .. n-1 of that stream (tests).                                */
int mm_dropout(int dtype, const void* x, int64_t n, float p, int64_t seed, int64_t offset, void* y, void* stream);
int mm_dropout_mask(int64_t seed, int64_t offset, int64_t n, float p, void* mask_u8, void* stream);

/* ---- optimizer: AdamW (config_alignment.yaml:38-59 -> torch.optim.AdamW semantics) + grad-norm clip ----------------
 * sumsq partial: out[blk] = sum g^2 over a slice; mm_gradnorm_finish: total[0] = sqrt(sum) ; clip coef in total[1]  */
int mm_gradnorm_partial(int dtype, const void* g, int64_t n, float* partial, int nblk, void* stream);
int mm_gradnorm_finish(const float* partial, int nblk, float max_norm, float* total, void* stream);
/* p (param dtype), g (param dtype), master/m/v f32.  clip = device scalar (total+1) or NULL.                          */
int mm_adamw_step(int dtype, void* p, const void* g, float* master, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, const float* clip, void* stream);

/* ---- gradient exchange over RCCL (one communicator per process = per GPU) -------------------------------------------
 * Replaces the reference's DeepSpeed gradient reduce / parameter gather (config/deepspeed.json:5-19) and the NCCL process
 * group torch.distributed builds for it (cli/train.py:200-201).  RCCL is bound at run time (dlopen); without it every call
 * returns MM_ERR_UNSUPPORTED.  mm_comm_unique_id: rank 0 makes the 128-byte id, the caller carries it to the other ranks
 * (any side channel: the torchrun store, a file, MPI) and every rank calls mm_comm_init on ITS device (hipSetDevice first).
 * All collectives are in place, sum, enqueued on `stream` (asynchronous; the caller orders them with events).
 * mm_comm_allreduce_bucket algo: 0 = ncclAllReduce, 1 = reduce-scatter + all-gather in one group (shard r = rank r's; the
 * tail count % world through a small all-reduce).  mm_comm_reduce_scatter / mm_comm_all_gather: the halves on their own for
 * a sharded optimiser step (count % world == 0; shard = count / world elements at buf + rank * shard).                     */
int mm_comm_unique_id(void* id128);
int mm_comm_init(const void* id128, int rank, int world, void** comm_out);
int mm_comm_rank(void* comm, int* rank, int* world);
int mm_comm_allreduce_bucket(void* comm, int dtype, void* buf, int64_t count, int algo, void* stream);
int mm_comm_reduce_scatter(void* comm, int dtype, void* buf, int64_t count, void* stream);
int mm_comm_all_gather(void* comm, int dtype, void* buf, int64_t count, void* stream);
int mm_comm_finalize(void* comm);

/* AdamW with the fp32 master weight held as (bf16 parameter, int16 remainder): master_bits = (p_bits << 16) + lo, p = RNE(master).
 * Same update as mm_adamw_step (MM_BF16), 26 B instead of 28 B of HBM traffic per parameter and no separate fp32 copy; one
 * remainder value in 2^17 (+0x8000, a round-to-even tie) is stored as 0x7FFF, i.e. the master moves by one fp32 ulp there.
 * mm_master_split / mm_master_join convert between that form and an fp32 master (optimiser checkpoints keep fp32).              */
int mm_adamw_step_split(void* p_bf16, const void* g_bf16, void* lo_i16, float* m, float* v, int64_t n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int step, const float* clip, void* stream);
int mm_master_split(const float* master, int64_t n, void* p_bf16, void* lo_i16, void* stream);
int mm_master_join(const void* p_bf16, const void* lo_i16, int64_t n, float* master, void* stream);

/* ---- image preprocessing on the device (optional row) ---------------------------------------------------
 * Replaces the CPU image processor the reference runs in its collator (image_modality.py:77,88-93 -> HF CLIPImageProcessor:
 * PIL resize BICUBIC, center crop, rescale 1/255, normalize) for an already decoded uint8 RGB image [src_h, src_w, 3] in HBM.
 * Pillow's two-pass fixed-point resampling, bit for bit; the int32 weight tables (bounds [n][2] = first tap, tap count; coef
 * [n][k]) are made on the host exactly as Pillow's precompute_coeffs / normalize_coeffs_8bpc do (dataset/gpu_image.py).
 * mm_image_resample_h: tmp [nrows, cw, 3] uint8 = horizontal pass of source rows r0 .. r0+nrows-1, resized columns left ..
 * left+cw-1.  mm_image_resample_v_norm: out [3, ch, cw] fp32 = vertical pass of resized rows top .. top+ch-1 over tmp, then
 * x * rescale, (x - mean) / std as separate float32 operations (the CPU path's); mean / std are HOST arrays of 3 floats.   */
int mm_image_resample_h(const void* src_u8, int src_h, int src_w, int src_row_stride, int r0, int nrows, const int* xbounds,
                        const int* xcoef, int kx, int left, int cw, void* tmp_u8, void* stream);
int mm_image_resample_v_norm(const void* tmp_u8, int r0, int nrows, int cw, const int* ybounds, const int* ycoef, int ky, int top,
                             int ch, float rescale, int do_rescale, const float* mean3_host, const float* std3_host, int do_norm,
                             float* out_chw, void* stream);

/* ---- utilities ---------------------------------------------------------------------------------------------------- */
int mm_cast(int src_dtype, int dst_dtype, const void* src, void* dst, int64_t n, void* stream);
int mm_fill_zero(void* p, int64_t bytes, void* stream);

/* ---- diagnostics (tests only): raw lane maps of ds_read_b64_tr_b16 and the bf16 MFMAs ------------------------------
 * tr_read: img = 4096 bf16 copied to LDS; lane l reads at byte address addr[l]; out[l*4+j] = its j-th element.
 * mfma: shape 32 -> v_mfma_f32_32x32x16_bf16 (out 64x16 f32), 16 -> v_mfma_f32_16x16x32_bf16 (out 64x4 f32);
 *       a/b = 64 lanes x 8 bf16 fragments.                                                                         */
/* Streams restricted to a subset of the CUs (no reference counterpart: the reference leaves kernel placement to PyTorch).  `mask`:
 * bit i of word i / 32 enables CU i (hipExtStreamCreateWithCUMask).  The Trainer runs AdamW and the deferred weight-gradient GEMMs
 * on such streams so that the small kernels of the modality tower they overlap always find free CUs.  mm_debug_cu_probe: where
 * the workgroups of a launch on `stream` ran (XCC_ID and HW_ID registers), for tools/cumask_probe.py.                            */
int mm_stream_create_cu_mask(const unsigned* mask, int nwords, void** stream);
int mm_stream_priority_range(int* least, int* greatest);
int mm_stream_create_priority(int priority, void** stream);
int mm_stream_destroy(void* stream);
int mm_device_cu_count(void);
int mm_debug_cu_probe(void* out_u32, int n_wg, int threads, int64_t spin_ticks, void* stream);
int mm_debug_tr_read(const void* img_bf16_4096, const void* lane_byte_addr_i32_64, void* out_bf16_256, void* stream);
int mm_debug_mfma(int shape, const void* a_frag_bf16_512, const void* b_frag_bf16_512, void* out_f32, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MM_HIP_H */
