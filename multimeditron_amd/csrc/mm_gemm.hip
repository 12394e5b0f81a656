// GEMM for every linear layer on the path (forward, input-gradient, weight-gradient).
//
// bf16, the kernel the step runs (gemm_bf16_dma_kernel): 256x256x64 block tile, 8 waves (2x4), each wave 128x64 = 8x4 tiles of
// v_mfma_f32_16x16x32_bf16, fp32 accumulate.  Operands stream HBM -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, inline asm)
// into a 2-deep ring with ONE raw s_barrier per K-tile; per-lane DMA offsets are loop-invariant (the K-step enters through the
// instruction's SGPR soffset); workgroups are persistent (one per CU, tiles in an XCD-aware order, GROUP_M = 8) and the last,
// half-empty round of tiles is cut into 256x128 halves.  Smaller instantiations (128x128, 64x128) serve problems that do not
// fill 256 CUs with 256-wide tiles (ViT-L on 4 images, the projector); gemm_bf16_kernel (128x128, register-staged, 2
// workgroups per CU) remains for mid-sized problems and for operands whose byte offsets exceed 32 bits; gemm_skinny_kernel
// (M <= 16) streams the weights once for the decode step.
// Two LDS images, chosen per operand by how it lies in memory:
//   KC (K contiguous, global [X][K])  image [x][k] swizzled  fragments by ds_read_b128 (conflict free)
//   KS (K strided,    global [K][X])  image [k][x] rotated   fragments by 2x ds_read_b64_tr_b16
// so NT / NN / TN need no transposed copy of any tensor.  The MFMA is issued with swapped operands
// (D^T = B^T A^T) so each lane owns 4 consecutive output columns -> 8-byte stores, and bias / activation / residual /
// accumulate / SwiGLU (forward: gate|up tile pairs; backward: on the down_proj dgrad) are applied in registers.
// Out-of-range rows are handled by buffer loads (hardware bounds check returns 0), so M, N need no
// padding and K only has to be a multiple of 8 for K-contiguous operands.
//
// f32 path (parity only): 64x64x16 tiles on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain).
#include <stdlib.h>
#include <type_traits>
#include <string.h>

#include "mm_common.h"

// default DMA variant when both fit: 1 = 256x128 (3-stage ring), 2 = 256x256 (2 stages)
#ifndef MM_DEFAULT_DMA_VARIANT
#define MM_DEFAULT_DMA_VARIANT(tiles256) ((tiles256) >= 192 ? 2 : 1)
#endif

namespace {

struct GemmArgs {
  int M, N, K;
  const void* A; int lda;
  const void* B; int ldb;
  void* C; int ldc;
  const void* bias;
  const void* residual; int ldr;
  int epi;
  int nbm, nbn;
  int tail;   // persistent 256x256 grid only: the last `tail` tiles are each cut into two 256x128 halves (see the kernel)
  // fused SwiGLU (mm_gemm_swiglu_fwd): swi_I = intermediate size I (0 = plain GEMM).  B is the fused [2I, K] gate|up weight;
  // a 256-column tile holds 128 gate columns and the 128 matching up columns, interleaved per wave so that a lane owns both
  // values of a feature; C = pre-activations [M, 2I] (kept for backward), C2 = silu(gate) * up [M, I].
  int swi_I;
  void* C2; int ldc2;
  const void* aux; int ldaux;   // MM_EPI_SWIGLU_BWD: the saved pre-activations [M, 2I]
  int pipe;                     // pipelined epilogue (mm_set_option "gemm_epi_pipe", default 1)
  // fused RoPE (mm_gemm_rope_fwd): the first rope_cols output columns are heads of width 128 that the epilogue rotates with the
  // per-token tables rope_cos / rope_sin [M, 64] f32 (0 = no rotation).  The B tile's rows are gathered so that a lane owns
  // columns d and d + 64 of a head (rope_row).
  const float* rope_cos; const float* rope_sin; int rope_cols;
  // sum of squares of what the epilogue stores (mm_gemm_sumsq, EK = 5): per output tile 16 slots = [column half][wave]; a full
  // 256x256 tile fills half 0 and zeroes half 1, the half tiles of the last round fill their own half: the slot of a value does
  // not depend on how the tiles were scheduled (persistent / one tile per workgroup, tail split or not).
  float* ss;
  int group_m;                  // 4-wave kernel: GROUP_M of the tile order (8)
  int stream_epi;               // 4-wave kernel: the plain epilogue without waits between its stores (gemm_epilogue_plain_stream)
  int shuffle;                  // 4-wave kernel: plain stores by register lane exchange instead of the LDS round trip (w4_shuffle_half)
  int diag_epi;                 // MM_W4_DIAG builds only: bits that drop parts of the row-major epilogue (timing-only; tools/w4_stamps.py)
  int stagger, stagger_slots;   // 4-wave kernel: start delay in cycles per slot ((blockIdx.x >> 3) % slots: workgroups that share an XCD, whose L2 write
                                // path is what a round's epilogue burst queues at) -- spreads the bursts inside every XCD
  int rowmajor;                 // 4-wave kernel: the plain epilogue in its row-major form (16-byte accesses; set by the host when alignment allows)
};
constexpr int MM_EPI_SWIGLU_BWD = 1 << 20;   // internal epilogue flag (mm_gemm_swiglu_bwd), not part of the ABI enum

__device__ __forceinline__ float act_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float act_quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float act_gelu_tanh(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);   // tanh(u) = 1 - 2/(1+exp(2u))
  return 0.5f * x * (2.0f - 2.0f / (1.0f + __expf(2.0f * u)));
}

__device__ __forceinline__ void block_to_tile(int bid, int nbm, int nbn, int& pm, int& pn, int group_m = 8) {
  const int nwg = nbm * nbn;
  // bijective XCD remap: blocks b and b+8 share an XCD; give each XCD a contiguous chunk of tile ids
  const int xcd = bid & 7, idx = bid >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int GROUP_M = group_m;       // (8 everywhere; the 4-wave kernel takes it from GemmArgs for the experiment of tools/w4_check.py --group-m)
  const int group = GROUP_M * nbn;
  const int gid = swz / group;
  const int first_m = gid * GROUP_M;
  const int gsz = min(nbm - first_m, GROUP_M);
  const int in_g = swz - gid * group;
  pm = first_m + in_g % gsz;
  pn = in_g / gsz;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int64_t bytes) {
  if (bytes < 0) bytes = 0;
  const unsigned nb = bytes > 0xFFFFFFFFll ? 0xFFFFFFFFu : (unsigned)bytes;
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, nb, 0x00020000);
}

__device__ __forceinline__ int ks_swz(int k) { return (k & 3) | ((k >> 1) & 4); }
// KC image = row-major [x][64 k] (128-B rows), 16-B chunk index XOR (row>>1)&7: ds_write_b128 of one row (8 lanes) and
// ds_read_b128 of 16 rows at one/two chunk columns are both bank-conflict free
__device__ __forceinline__ int kc_swz(int row) { return (row >> 1) & 7; }

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;  // 16 KiB per operand tile

// ---- staging: global -> registers ------------------------------------------------------------------
// KC operand: rows x0.., tile [128 x][64 k]; lane: r_in = l&7, kc = l>>3; wave w, i: row = (w*4+i)*8 + r_in
template <bool KC>
__device__ __forceinline__ void stage_load(u32x4 (&r)[4], const bf16* base, int ld, int x0, int Xtot, int k0, int Ktot) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  if constexpr (KC) {
    const bf16* b = base + (int64_t)x0 * ld + k0;
    auto rs = make_rsrc(b, ((int64_t)(Xtot - x0) * ld - k0) * 2);
    const int kc = l & 7, r_in = l >> 3;   // 8 adjacent lanes = one full 128-B line of a row
    const bool kok = (k0 + kc * 8) < Ktot;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (w * 4 + i) * 8 + r_in;
      unsigned off = (unsigned)(row * ld + kc * 8) * 2u;
      if (!kok) off = 0xFFFFFFFFu;
      r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    }
  } else {
    const bf16* b = base + (int64_t)k0 * ld + x0;
    auto rs = make_rsrc(b, ((int64_t)(Ktot - k0) * ld - x0) * 2);
    const int xc = l & 15, kr = l >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = i * 16 + w * 4 + kr;
      unsigned off = (unsigned)(k * ld + xc * 8) * 2u;
      r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    }
  }
}

template <bool KC>
__device__ __forceinline__ void stage_store(const u32x4 (&r)[4], char* tile) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  if constexpr (KC) {
    const int kc = l & 7, r_in = l >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (w * 4 + i) * 8 + r_in;
      *(u32x4*)(tile + row * 128 + ((kc ^ kc_swz(row)) * 16)) = r[i];
    }
  } else {
    const int xc = l & 15, kr = l >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = i * 16 + w * 4 + kr;
      *(u32x4*)(tile + k * 256 + (((xc >> 1) ^ ks_swz(k)) * 32) + (xc & 1) * 16) = r[i];
    }
  }
}

// ---- fragments: LDS -> registers ---------------------------------------------------------------------
// returns the 16x32 MFMA operand fragment (lane: idx = l&15 along x, k = 8*(l>>4)+j) for x-block xb (16 wide)
template <bool KC>
__device__ __forceinline__ bf16x8 frag_load(const char* tile, int xb, int ks) {
  const int l = threadIdx.x & 63;
  if constexpr (KC) {
    const int row = xb * 16 + (l & 15), kc = ks * 4 + (l >> 4);
    return *(const bf16x8*)(tile + row * 128 + ((kc ^ kc_swz(row)) * 16));
  } else {
    const int g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
    const int k = ks * 32 + 8 * g + q;
    const int sw = (xb ^ ks_swz(k)) * 32 + p * 8;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, tile + k * 256 + sw));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, tile + (k + 4) * 256 + sw));
    bf16x8 o;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
    o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
    return o;
  }
}

// epilogue shared by the bf16 kernels: acc[i][j][r] = C[mw + i*16 + (l&15)][nw + j*16 + 4*(l>>4) + r]
// ALLOW_PRE: compile the pre-activation store of mm_gemm_act_fwd into this instantiation.  Only the small-tile DMA kernels get it:
// in the 256x256 kernel (128 accumulator registers per lane) the extra path cost 528 bytes of scratch per lane and 10 % of the
// GEMM's speed for EVERY launch -- found by the step going from 401 to 440 ms.
template <int MREP, int NREP, bool ALLOW_PRE, bool ACT = true>
__device__ __forceinline__ void gemm_epilogue_plain(const GemmArgs& g, f32x4 (&acc)[MREP][NREP], int mw, int nw, int lane = -1) {
  const int l = lane >= 0 ? lane : (int)(threadIdx.x & 63);
  bf16* C = (bf16*)g.C;
  const bf16* bias = (const bf16*)g.bias;
  const bf16* R = (const bf16*)g.residual;
  const int epi = g.epi;
#pragma unroll
  for (int i = 0; i < MREP; ++i) {
    const int m = mw + i * 16 + (l & 15);
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
      const int n = nw + j * 16 + 4 * (l >> 4);
      if (n >= g.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      const bool full = (n + 3) < g.N;
      bf16* cp = C + (int64_t)m * g.ldc + n;
      if (epi & MM_EPI_BIAS) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (full || n + r < g.N) v[r] += (float)bias[n + r];
      }
      const bool keep_pre = ALLOW_PRE && g.C2 != nullptr;   // mm_gemm_act_fwd: the pre-activation goes to C2 (backward needs it) and
      if (keep_pre) {                                   // the arithmetic takes the roundings of the two-launch form (bit-identical)
        bf16* pp = (bf16*)g.C2 + (int64_t)m * g.ldc2 + n;
        if (full) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) { o[r] = (bf16)v[r]; v[r] = (float)o[r]; }
          *(bf16x4*)pp = o;
        } else {
          for (int r = 0; r < 4; ++r)
            if (n + r < g.N) { const bf16 o = (bf16)v[r]; pp[r] = o; v[r] = (float)o; }
        }
      }
      if (ACT) {                                        // EK == 1 instantiations only (NT: a forward Linear with an activation)
        if (epi & MM_EPI_GELU_ERF) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = act_gelu_erf(v[r]);
        } else if (epi & MM_EPI_QUICK_GELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = act_quick_gelu(v[r]);
        } else if (epi & MM_EPI_GELU_TANH) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = act_gelu_tanh(v[r]);
        }
      }
      if (keep_pre && (epi & MM_EPI_RESIDUAL)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (float)(bf16)v[r];       // the activation kernel's store, before the add kernel
      }
      if (epi & MM_EPI_RESIDUAL) {
        const bf16* rp = R + (int64_t)m * g.ldr + n;
        if (full) {
          bf16x4 rv = *(const bf16x4*)rp;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
        } else {
          for (int r = 0; r < 4; ++r)
            if (n + r < g.N) v[r] += (float)rp[r];
        }
      }
      if (epi & MM_EPI_ACCUMULATE) {
        if (full) {
          bf16x4 cv = *(const bf16x4*)cp;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)cv[r];
        } else {
          for (int r = 0; r < 4; ++r)
            if (n + r < g.N) v[r] += (float)cp[r];
        }
      }
      if (full) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16)v[r];
        *(bf16x4*)cp = o;      // default cache policy: non-temporal stores here measured -4 % (tools/build_diag.sh)
      } else {
        for (int r = 0; r < 4; ++r)
          if (n + r < g.N) cp[r] = (bf16)v[r];
      }
    }
  }
}

// MM_EPI_SWIGLU_BWD (mm_gemm_swiglu_bwd) as its own epilogue: kept out of the plain one, whose instruction count and register
// pressure every GEMM launch pays for.
template <int MREP, int NREP>
__device__ __forceinline__ void gemm_epilogue_swiglu_bwd(const GemmArgs& g, f32x4 (&acc)[MREP][NREP], int mw, int nw, int lane = -1) {
  const int l = lane >= 0 ? lane : (int)(threadIdx.x & 63);
  bf16* C = (bf16*)g.C;
#pragma unroll
  for (int i = 0; i < MREP; ++i) {
    const int m = mw + i * 16 + (l & 15);
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
      const int n = nw + j * 16 + 4 * (l >> 4);
      if (n >= g.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      // C = d(gate|up) [M, 2N]; acc = d(act) [M, N] (down_proj's input gradient); aux = saved gate|up.  Same arithmetic and
      // rounding points as down_proj dgrad (bf16 store) followed by swiglu_bwd_kernel: N % 4 == 0 is checked by the host.
      const bf16* gp = (const bf16*)g.aux + (int64_t)m * g.ldaux + n;
      const bf16x4 gv = *(const bf16x4*)gp, uv = *(const bf16x4*)(gp + g.N);
      bf16x4 dg, du;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gf = (float)gv[r], sig = 1.0f / (1.0f + __expf(-gf));
        const float sg = gf * sig;
        const float dd = (float)(bf16)v[r];
        du[r] = (bf16)(dd * sg);
        dg[r] = (bf16)(dd * (float)uv[r] * (sig * (1.0f + gf * (1.0f - sig))));
      }
      bf16* dp = C + (int64_t)m * g.ldc + n;
      *(bf16x4*)dp = dg;
      *(bf16x4*)(dp + g.N) = du;
    }
  }
}

// ---- the same two epilogues, PIPELINED (round 3) -------------------------------------------------------------------------
// The loops above compile to 32 serial round trips per wave and tile: every (i, j) step is `branch on the bounds -> load
// residual / C / aux -> s_waitcnt vmcnt(0) (which also waits for the PREVIOUS step's stores) -> arithmetic -> store`, because the
// bounds `continue`s are divergent branches the loads cannot be hoisted across.  With K = 4096 (64 K-steps of ~1.1 us) that tail
// was a fifth of the tile: the SwiGLU-backward dgrad ran at 1017 TFLOP/s inside the step where the plain dgrad GEMMs reach 1346
// (profiles/r03_*).  Here the bounds go into the buffer instructions' hardware range check (descriptor anchored at the wave's
// first row, so a row >= M lies beyond num_records; a column group >= N gets an out-of-range offset), nothing branches per
// element, and the operands of row block i + 1 are requested before row block i is computed and stored.  Same arithmetic, same
// rounding points: bit-identical output.  A wave whose columns reach a ragged N (N % 4 != 0: the padded-stride logits)
// keeps the scalar path.
constexpr unsigned EPI_OOB = 0xFFFFFFFFu;

template <int MREP, int NREP, bool SUMSQ = false>
__device__ __forceinline__ void gemm_epilogue_plain_pipe(const GemmArgs& g, f32x4 (&acc)[MREP][NREP], int mw, int nw, int ss_slot = 0, int lane = -1) {
  const int l = lane >= 0 ? lane : (int)(threadIdx.x & 63);      // lane: the caller's (opaque) copy of the lane index, see gemm_bf16_w4_kernel
  float ssq = 0.f;
  mw = __builtin_amdgcn_readfirstlane(mw);
  nw = __builtin_amdgcn_readfirstlane(nw);
  const int epi = g.epi;
  const bool has_res = (epi & MM_EPI_RESIDUAL) != 0, has_acc = (epi & MM_EPI_ACCUMULATE) != 0, has_bias = (epi & MM_EPI_BIAS) != 0;
  const int rows = g.M - mw;                                   // <= 0: the whole wave tile is below the matrix (everything out of range)
  auto rc = make_rsrc((const bf16*)g.C + (int64_t)mw * g.ldc, (int64_t)rows * g.ldc * 2);
  auto rr = make_rsrc(has_res ? (const bf16*)g.residual + (int64_t)mw * g.ldr : (const bf16*)g.C, has_res ? (int64_t)rows * g.ldr * 2 : 0);
  unsigned colb[NREP];                                         // byte offset of the lane's 4 columns inside a row, or out of range
  float bv[NREP][4];
#pragma unroll
  for (int j = 0; j < NREP; ++j) {
    const int n = nw + j * 16 + 4 * (l >> 4);
    colb[j] = n + 3 < g.N ? (unsigned)n * 2u : EPI_OOB;          // a group that straddles a ragged N: scalar tail below
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[j][r] = 0.f;
    if (has_bias && n + 3 < g.N) {
      const bf16x4 b4 = *(const bf16x4*)((const bf16*)g.bias + n);
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[j][r] = (float)b4[r];
    }
  }
  u32x2 rbuf[2][NREP], cbuf[2][NREP];
  auto request = [&](int i, int s) {
    const unsigned rowc = (unsigned)((i * 16 + (l & 15)) * g.ldc) * 2u, rowr = (unsigned)((i * 16 + (l & 15)) * g.ldr) * 2u;
    if (has_res) {
#pragma unroll
      for (int j = 0; j < NREP; ++j) rbuf[s][j] = __builtin_amdgcn_raw_buffer_load_b64(rr, colb[j] == EPI_OOB ? EPI_OOB : rowr + colb[j], 0, 0);
    }
    if (has_acc) {
#pragma unroll
      for (int j = 0; j < NREP; ++j) cbuf[s][j] = __builtin_amdgcn_raw_buffer_load_b64(rc, colb[j] == EPI_OOB ? EPI_OOB : rowc + colb[j], 0, 0);
    }
  };
  request(0, 0);
  const unsigned row_lim = rows > 0 ? (unsigned)((rows < MREP * 16 ? rows : MREP * 16) * g.ldc) * 2u : 0u;      // wave-uniform: byte offset of the first row >= M
#pragma unroll
  for (int i = 0; i < MREP; ++i) {
    if (i + 1 < MREP) request(i + 1, (i + 1) & 1);
    const unsigned rowc = (unsigned)((i * 16 + (l & 15)) * g.ldc) * 2u;
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (has_bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bv[j][r];
      }
      if (has_res) {
        const bf16x4 rv = __builtin_bit_cast(bf16x4, rbuf[i & 1][j]);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
      }
      if (has_acc) {
        const bf16x4 cv = __builtin_bit_cast(bf16x4, cbuf[i & 1][j]);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)cv[r];
      }
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (bf16)v[r];
      if constexpr (SUMSQ) {      // of the bf16 values as stored; a row >= M or a column >= N holds whatever the operand tiles'
                                  // neighbours hold (only the STORE is range-checked), so both are masked here
        if (colb[j] != EPI_OOB && rowc < row_lim) {
#pragma unroll
          for (int r = 0; r < 4; ++r) ssq = __builtin_fmaf((float)o[r], (float)o[r], ssq);
        }
      }
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rc, colb[j] == EPI_OOB ? EPI_OOB : rowc + colb[j], 0, 0);
    }
  }
  // ragged N (N % 4 != 0: the padded-stride logits): the one column group that straddles N, element by element
  if ((g.N & 3) != 0 && nw + NREP * 16 > (g.N & ~3)) {
    const bf16* bias = (const bf16*)g.bias;
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
      const int m = mw + i * 16 + (l & 15);
#pragma unroll
      for (int j = 0; j < NREP; ++j) {
        const int n = nw + j * 16 + 4 * (l >> 4);
        if (m < g.M && n < g.N && n + 3 >= g.N) {
          bf16* cp = (bf16*)g.C + (int64_t)m * g.ldc + n;
          for (int r = 0; r < 4; ++r)
            if (n + r < g.N) {
              float v = acc[i][j][r];
              if (has_bias) v += (float)bias[n + r];
              if (has_res) v += (float)((const bf16*)g.residual)[(int64_t)m * g.ldr + n + r];
              if (has_acc) v += (float)cp[r];
              const bf16 ob = (bf16)v;
              cp[r] = ob;
              if constexpr (SUMSQ) ssq = __builtin_fmaf((float)ob, (float)ob, ssq);
            }
        }
      }
    }
  }
  if constexpr (SUMSQ) {
    ssq = wave_sum(ssq);
    if (l == 0) {
      g.ss[ss_slot] = ssq;
      if (ss_slot >= 0 && NREP == 4) g.ss[ss_slot + 8] = 0.f;      // a full tile (NREP == 4) also zeroes its second-half slot
    }
  }
}

// ---- the plain epilogue with NO wait between its stores (round 4) -----------------------------------------------------------------
// gemm_epilogue_plain_pipe decides residual / accumulate at run time, so its counted waits (`s_waitcnt vmcnt(7)` in front of every
// store: "the loads of the next row block may be in flight") are there in the plain case too, where the only vector-memory
// operations are the stores themselves: at most 8 stores of a wave are ever outstanding and a 256x256 tile's epilogue takes ~21 000
// cycles, 13 % of a K = 4096 tile (tools/w4_stamps.py; vmcnt retires in issue order, a store ~2 us after its issue).  MODE is a
// compile-time parameter here: 0 = no loads at all -> no wait, the stores stream; 1 (residual) / 2 (accumulate) = every load of the
// wave tile is requested first, ONE wait, then the stores stream.  Same arithmetic, same rounding points: bit-identical output.
template <int MREP, int NREP, int MODE>
__device__ __forceinline__ void gemm_epilogue_plain_stream(const GemmArgs& g, f32x4 (&acc)[MREP][NREP], int mw, int nw, int lane) {
  const int l = lane;
  mw = __builtin_amdgcn_readfirstlane(mw);
  nw = __builtin_amdgcn_readfirstlane(nw);
  const bool has_bias = (g.epi & MM_EPI_BIAS) != 0;
  const int rows = g.M - mw;
  auto rc = make_rsrc((const bf16*)g.C + (int64_t)mw * g.ldc, (int64_t)rows * g.ldc * 2);
  auto rl = MODE == 1 ? make_rsrc((const bf16*)g.residual + (int64_t)mw * g.ldr, (int64_t)rows * g.ldr * 2) : rc;
  const int ldl = MODE == 1 ? g.ldr : g.ldc;
  unsigned colb[NREP];
  float bv[NREP][4];
#pragma unroll
  for (int j = 0; j < NREP; ++j) {
    const int n = nw + j * 16 + 4 * (l >> 4);
    colb[j] = n + 3 < g.N ? (unsigned)n * 2u : EPI_OOB;
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[j][r] = 0.f;
    if (has_bias && n + 3 < g.N) {
      const bf16x4 b4 = *(const bf16x4*)((const bf16*)g.bias + n);
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[j][r] = (float)b4[r];
    }
  }
  u32x2 lbuf[MODE ? MREP : 1][NREP];
  if constexpr (MODE != 0) {
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
      const unsigned rowl = (unsigned)((i * 16 + (l & 15)) * ldl) * 2u;
#pragma unroll
      for (int j = 0; j < NREP; ++j) lbuf[i][j] = __builtin_amdgcn_raw_buffer_load_b64(rl, colb[j] == EPI_OOB ? EPI_OOB : rowl + colb[j], 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < MREP; ++i) {
    const unsigned rowc = (unsigned)((i * 16 + (l & 15)) * g.ldc) * 2u;
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (has_bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bv[j][r];
      }
      if constexpr (MODE != 0) {
        const bf16x4 lv = __builtin_bit_cast(bf16x4, lbuf[i][j]);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)lv[r];
      }
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (bf16)v[r];
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rc, colb[j] == EPI_OOB ? EPI_OOB : rowc + colb[j], 0, 0);
    }
  }
}

template <int MREP, int NREP>
__device__ __forceinline__ void gemm_epilogue_swiglu_bwd_pipe(const GemmArgs& g, f32x4 (&acc)[MREP][NREP], int mw, int nw, int lane = -1) {
  const int l = lane >= 0 ? lane : (int)(threadIdx.x & 63);
  mw = __builtin_amdgcn_readfirstlane(mw);
  nw = __builtin_amdgcn_readfirstlane(nw);
  const int rows = g.M - mw;
  auto rc = make_rsrc((const bf16*)g.C + (int64_t)mw * g.ldc, (int64_t)rows * g.ldc * 2);
  auto ra = make_rsrc((const bf16*)g.aux + (int64_t)mw * g.ldaux, (int64_t)rows * g.ldaux * 2);
  unsigned colb[NREP];
#pragma unroll
  for (int j = 0; j < NREP; ++j) {
    const int n = nw + j * 16 + 4 * (l >> 4);
    colb[j] = n < g.N ? (unsigned)n * 2u : EPI_OOB;         // N % 4 == 0 (checked by the host): a column group is whole or absent
  }
  const unsigned upo = (unsigned)g.N * 2u;                   // the up half sits N columns to the right of the gate half
  u32x2 gbuf[2][NREP], ubuf[2][NREP];
  auto request = [&](int i, int s) {
    const unsigned rowa = (unsigned)((i * 16 + (l & 15)) * g.ldaux) * 2u;
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
      const unsigned o = colb[j] == EPI_OOB ? EPI_OOB : rowa + colb[j];
      gbuf[s][j] = __builtin_amdgcn_raw_buffer_load_b64(ra, o, 0, 0);
      ubuf[s][j] = __builtin_amdgcn_raw_buffer_load_b64(ra, o == EPI_OOB ? EPI_OOB : o + upo, 0, 0);
    }
  };
  request(0, 0);
#pragma unroll
  for (int i = 0; i < MREP; ++i) {
    if (i + 1 < MREP) request(i + 1, (i + 1) & 1);
    const unsigned rowc = (unsigned)((i * 16 + (l & 15)) * g.ldc) * 2u;
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
      const bf16x4 gv = __builtin_bit_cast(bf16x4, gbuf[i & 1][j]), uv = __builtin_bit_cast(bf16x4, ubuf[i & 1][j]);
      bf16x4 dg, du;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gf = (float)gv[r], sig = 1.0f / (1.0f + __expf(-gf));
        const float sg = gf * sig;
        const float dd = (float)(bf16)acc[i][j][r];
        du[r] = (bf16)(dd * sg);
        dg[r] = (bf16)(dd * (float)uv[r] * (sig * (1.0f + gf * (1.0f - sig))));
      }
      const unsigned o = colb[j] == EPI_OOB ? EPI_OOB : rowc + colb[j];
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, dg), rc, o, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, du), rc, o == EPI_OOB ? EPI_OOB : o + upo, 0, 0);
    }
  }
}

#ifndef MM_GEMM_EPI_PIPE
#define MM_GEMM_EPI_PIPE 1
#endif

// EK == 0 / EK == 2 epilogue of the LDS-DMA kernels.  The pipelined forms address a wave tile with 32-bit byte offsets: the host
// (gemm_launch) sends a problem whose leading dimensions do not allow that to the register-staged kernel.
template <int MREP, int NREP>
__device__ __forceinline__ void gemm_epilogue_ek0(const GemmArgs& g, f32x4 (&acc)[MREP][NREP], int mw, int nw, int lane = -1) {
#if MM_GEMM_EPI_PIPE == 2                 // A/B build: both epilogues compiled in, chosen by mm_set_option("gemm_epi_pipe")
  if (!g.pipe) { gemm_epilogue_plain<MREP, NREP, false, false>(g, acc, mw, nw, lane); return; }
#endif
#if MM_GEMM_EPI_PIPE
  gemm_epilogue_plain_pipe<MREP, NREP>(g, acc, mw, nw, 0, lane);
#else
  gemm_epilogue_plain<MREP, NREP, false, false>(g, acc, mw, nw, lane);
#endif
}

template <int MREP, int NREP>
__device__ __forceinline__ void gemm_epilogue_ek2(const GemmArgs& g, f32x4 (&acc)[MREP][NREP], int mw, int nw, int lane = -1) {
#if MM_GEMM_EPI_PIPE == 2
  if (!g.pipe) { gemm_epilogue_swiglu_bwd<MREP, NREP>(g, acc, mw, nw, lane); return; }
#endif
#if MM_GEMM_EPI_PIPE
  gemm_epilogue_swiglu_bwd_pipe<MREP, NREP>(g, acc, mw, nw, lane);
#else
  gemm_epilogue_swiglu_bwd<MREP, NREP>(g, acc, mw, nw, lane);
#endif
}

// runtime dispatch between the two (the register-staged 128x128 kernel, which the step does not run; the LDS-DMA kernels pick
// their epilogue at compile time, see EK)
template <int MREP, int NREP, bool ALLOW_PRE = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x4 (&acc)[MREP][NREP], int mw, int nw) {
  if (g.epi & MM_EPI_SWIGLU_BWD) gemm_epilogue_swiglu_bwd<MREP, NREP>(g, acc, mw, nw);
  else gemm_epilogue_plain<MREP, NREP, ALLOW_PRE>(g, acc, mw, nw);
}

// sum of squares of a bf16 [M, N] matrix with leading dimension ld -> partial[blockIdx.x] (mm_gemm_sumsq's second pass)
__global__ __launch_bounds__(256) void sumsq2d_kernel(const bf16* C, int M, int N, int ld, float* partial) {
  __shared__ float red[8];
  float s = 0.f;
  const int64_t nv = (int64_t)M * (N / 4);                   // N % 4 == 0 is guaranteed by ldc % 4 == 0 ... checked by the host
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / (N / 4), c = (i % (N / 4)) * 4;
    const bf16x4 v = *(const bf16x4*)(C + r * ld + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) s = __builtin_fmaf((float)v[k], (float)v[k], s);
  }
  if ((N & 3) && blockIdx.x == 0)
    for (int64_t i = threadIdx.x; i < (int64_t)M * (N & 3); i += 256) {
      const float q = (float)C[(i / (N & 3)) * ld + (N & ~3) + i % (N & 3)];
      s = __builtin_fmaf(q, q, s);
    }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][A tile | B tile]
  int pm, pn;
  block_to_tile(blockIdx.x, g.nbm, g.nbn, pm, pn);
  const int m0 = pm * BM, n0 = pn * BN;
  const bf16* A = (const bf16*)g.A;
  const bf16* B = (const bf16*)g.B;
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = w >> 1, wn = w & 1;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ra[4], rb[4];
  const int nk = (g.K + BK - 1) / BK;
  stage_load<A_KC>(ra, A, g.lda, m0, g.M, 0, g.K);
  stage_load<B_KC>(rb, B, g.ldb, n0, g.N, 0, g.K);
  stage_store<A_KC>(ra, smem);
  stage_store<B_KC>(rb, smem + TILE_BYTES);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    char* cur = smem + (kt & 1) * (2 * TILE_BYTES);
    char* nxt = smem + ((kt + 1) & 1) * (2 * TILE_BYTES);
    const bool more = (kt + 1) < nk;
    if (more) {
      stage_load<A_KC>(ra, A, g.lda, m0, g.M, (kt + 1) * BK, g.K);
      stage_load<B_KC>(rb, B, g.ldb, n0, g.N, (kt + 1) * BK, g.K);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = frag_load<A_KC>(cur, wm * 4 + i, ks);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = frag_load<B_KC>(cur + TILE_BYTES, wn * 4 + j, ks);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    if (more) {
      stage_store<A_KC>(ra, nxt);
      stage_store<B_KC>(rb, nxt + TILE_BYTES);
    }
    __syncthreads();
  }

  gemm_epilogue<4, 4>(g, acc, m0 + wm * 64, n0 + wn * 64);
}

// ------------------------------------------------------------------------------------------------------
// v2 main kernel: 256x128x64 tile, 8 waves (4 along M x 2 along N, 64x64 each), operands streamed HBM -> LDS by
// LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers, no ds_write, hardware bounds check) into a 3-deep
// ring; ONE raw s_barrier per K-tile and a counted s_waitcnt vmcnt(6) that leaves the next tile's six DMAs in
// flight across the barrier (the DMA of tile t+2 is issued right after the barrier that retires tile t-1's reads).
// LDS images are identical to v1 (swizzle applied on the per-lane SOURCE address, destination linear in lane
// order as LDS-DMA requires), so fragment reads stay bank-conflict free.
// ------------------------------------------------------------------------------------------------------
constexpr int G_BK = 64;

// epilogue of the fused gate|up GEMM: acc[i][j] (j < NREP/2) = gate of features fw + j*16 + 4*(l>>4) + r, acc[i][j + NREP/2] = up
// of the SAME features (the B tile's rows were gathered that way).  Writes the bf16 pre-activations (what the unfused GEMM
// stores) and act = silu(gate) * up computed from those rounded values exactly as swiglu_fwd_kernel does: bit-identical
// to GEMM + mm_swiglu_fwd, one pass less over [M, 2I] and one launch less per layer.
template <int MREP, int NREP>
__device__ __forceinline__ void gemm_epilogue_swiglu(const GemmArgs& g, f32x4 (&acc)[MREP][NREP], int mw, int fw, int lane = -1) {
  const int l = lane >= 0 ? lane : (int)(threadIdx.x & 63);
  bf16* GU = (bf16*)g.C;
  bf16* ACT = (bf16*)g.C2;
  const int I = g.swi_I;
#pragma unroll
  for (int i = 0; i < MREP; ++i) {
    const int m = mw + i * 16 + (l & 15);
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < NREP / 2; ++j) {
      const int f = fw + j * 16 + 4 * (l >> 4);
      if (f >= I) continue;                       // I % 4 == 0
      bf16x4 gb, ub, o;
#pragma unroll
      for (int r = 0; r < 4; ++r) { gb[r] = (bf16)acc[i][j][r]; ub[r] = (bf16)acc[i][j + NREP / 2][r]; }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gf = (float)gb[r];
        const float sg = (float)(bf16)(gf / (1.0f + __expf(-gf)));     // HF: act_fn(gate) in the storage dtype, then multiply
        o[r] = (bf16)(sg * (float)ub[r]);
      }
      bf16* gp = GU + (int64_t)m * g.ldc + f;
      *(bf16x4*)gp = gb;
      *(bf16x4*)(gp + I) = ub;
      *(bf16x4*)(ACT + (int64_t)m * g.ldc2 + f) = o;
    }
  }
}

// epilogue of the fused q|k|v projection + RoPE (mm_gemm_rope_fwd): acc[i][j] (j < 2) = columns d .. d+3 of a 128-wide head,
// acc[i][j + 2] = columns d + 64 .. (the B tile's rows were gathered that way, rope_row).  The projection output is rounded to
// bf16 first, exactly what the unfused GEMM stores, then rotated with the arithmetic of rope_apply_kernel (rope_lo / rope_hi):
// bit-identical to mm_gemm + mm_rope_apply, one read-modify-write pass over q|k less.  Tiles at or beyond rope_cols (the v heads)
// are stored unrotated.  Table rows and the rows of C go through the buffer range check (descriptors anchored at the wave's
// first row); the loads of row block i + 1 are requested before block i is rotated and stored.
template <int MREP, int NREP>
__device__ __forceinline__ void gemm_epilogue_rope(const GemmArgs& g, f32x4 (&acc)[MREP][NREP], int mw, int n0, int wn, int lane = -1) {
  static_assert(NREP == 4, "a wave owns 64 columns: two 16-column tiles of d and their partners d + 64");
  const int l = lane >= 0 ? lane : (int)(threadIdx.x & 63);
  mw = __builtin_amdgcn_readfirstlane(mw);
  const int hb = __builtin_amdgcn_readfirstlane(n0 + (wn >> 1) * 128);          // first column of the wave's head
  const int dlo = __builtin_amdgcn_readfirstlane((wn & 1) * 32);
  const bool rot = hb < g.rope_cols;                                             // per HEAD: a tile may hold the last k head and the first v head
  const int rows = g.M - mw;
  auto rc = make_rsrc((const bf16*)g.C + (int64_t)mw * g.ldc, (int64_t)rows * g.ldc * 2);
  auto rcs = make_rsrc(g.rope_cos + (int64_t)mw * 64, rot ? (int64_t)rows * 64 * 4 : 0);
  auto rsn = make_rsrc(g.rope_sin + (int64_t)mw * 64, rot ? (int64_t)rows * 64 * 4 : 0);
  const bool has_bias = (g.epi & MM_EPI_BIAS) != 0;
  const bool head_ok = hb < g.N;                                                 // N is a multiple of 128: a head is whole or absent
  float b1[2][4], b2[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int d = dlo + j * 16 + 4 * (l >> 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) { b1[j][r] = 0.f; b2[j][r] = 0.f; }
    if (has_bias && head_ok) {
      const bf16x4 v1 = *(const bf16x4*)((const bf16*)g.bias + hb + d), v2 = *(const bf16x4*)((const bf16*)g.bias + hb + d + 64);
#pragma unroll
      for (int r = 0; r < 4; ++r) { b1[j][r] = (float)v1[r]; b2[j][r] = (float)v2[r]; }
    }
  }
  u32x4 cb[2][2], sb[2][2];
  auto request = [&](int i, int s) {
    const unsigned row = (unsigned)(i * 16 + (l & 15)) * 256u;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned o = row + (unsigned)(dlo + j * 16 + 4 * (l >> 4)) * 4u;
      cb[s][j] = __builtin_amdgcn_raw_buffer_load_b128(rcs, o, 0, 0);
      sb[s][j] = __builtin_amdgcn_raw_buffer_load_b128(rsn, o, 0, 0);
    }
  };
  if (rot) request(0, 0);
#pragma unroll
  for (int i = 0; i < MREP; ++i) {
    if (rot && i + 1 < MREP) request(i + 1, (i + 1) & 1);
    const unsigned rowc = (unsigned)((i * 16 + (l & 15)) * g.ldc) * 2u;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int d = dlo + j * 16 + 4 * (l >> 4);
      bf16x4 o1, o2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o1[r] = (bf16)(acc[i][j][r] + b1[j][r]);
        o2[r] = (bf16)(acc[i][j + 2][r] + b2[j][r]);
      }
      if (rot) {
        const f32x4 c4 = __builtin_bit_cast(f32x4, cb[i & 1][j]), s4 = __builtin_bit_cast(f32x4, sb[i & 1][j]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = (float)o1[r], b = (float)o2[r];
          o1[r] = (bf16)rope_lo(a, b, c4[r], s4[r]);
          o2[r] = (bf16)rope_hi(a, b, c4[r], s4[r]);
        }
      }
      const unsigned o = head_ok ? rowc + (unsigned)(hb + d) * 2u : EPI_OOB;
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o1), rc, o, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o2), rc, o == EPI_OOB ? EPI_OOB : o + 128u, 0, 0);
    }
  }
}

// N LDS-DMA pieces of one operand tile in ONE asm statement: M0 saved/restored once, one hazard pad for the
// freshly written descriptor SGPRs (the compiler pads nothing inside an asm string).
template <int N, int STEP>
__device__ __forceinline__ void lds_dma16xN(const SRsrc& r, const unsigned (&voff)[N], unsigned lds_addr0) {
  static_assert(N == 2 || N == 4, "pieces per call");
  u32x4 d = {r.w0, r.w1, r.w2, r.w3};
  unsigned keep;
  if constexpr (N == 4) {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 4\n\t"
        "buffer_load_dwordx4 %1, %6, 0 offen lds\n\t"
        "s_add_u32 m0, m0, %7\n\t"
        "buffer_load_dwordx4 %2, %6, 0 offen lds\n\t"
        "s_add_u32 m0, m0, %7\n\t"
        "buffer_load_dwordx4 %3, %6, 0 offen lds\n\t"
        "s_add_u32 m0, m0, %7\n\t"
        "buffer_load_dwordx4 %4, %6, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(lds_addr0), "s"(d), "i"(STEP)
        : "memory", "scc");
  } else {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 4\n\t"
        "buffer_load_dwordx4 %1, %4, 0 offen lds\n\t"
        "s_add_u32 m0, m0, %5\n\t"
        "buffer_load_dwordx4 %2, %4, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff[0]), "v"(voff[1]), "s"(lds_addr0), "s"(d), "i"(STEP)
        : "memory", "scc");
  }
}

// descriptor of one operand anchored at the tile origin (k = 0); built once per workgroup
template <bool KC>
__device__ __forceinline__ SRsrc tile_rsrc(const bf16* base, int ld, int x0, int Xtot, int Ktot) {
#ifdef MM_GEMM_DIAG_OOB
  return make_srsrc(base, 0);   // timing experiment: every DMA lane out of range (zeros land in LDS, no memory traffic)
#endif
  if constexpr (KC) return make_srsrc(base + (int64_t)x0 * ld, (int64_t)(Xtot - x0) * ld * 2);
  else return make_srsrc(base + x0, ((int64_t)Ktot * ld - x0) * 2);
}

// fused SwiGLU: local row r of the 256-row B tile (wave wn = r >> 6 owns 64 of them as 4 n-tiles of 16: two gate, two up)
// -> row of the fused [2I, K] weight relative to the tile's first gate row
__device__ __forceinline__ int swiglu_row(int r, int I) {
  return (r >> 6) * 32 + ((r >> 4) & 1) * 16 + (r & 15) + ((r >> 5) & 1) * I;
}

// fused RoPE (head width 128): local row r of the 256-row B tile (2 heads; wave wn = r >> 6 owns 64 of them as 4 n-tiles of 16)
// -> row of the [N, K] weight relative to the tile's first row: n-tiles 0,1 of a wave hold d = (wn & 1) * 32 + 0..31 of head
// wn >> 1, n-tiles 2,3 the partners d + 64, so acc[i][j] and acc[i][j + 2] of a lane are the two halves rotate_half pairs up
__device__ __forceinline__ int rope_row(int r) {
  return (r >> 7) * 128 + ((r >> 6) & 1) * 32 + ((r >> 4) & 1) * 16 + (r & 15) + ((r >> 5) & 1) * 64;
}
// B-row gather of the fused epilogues: mode > 0 = SwiGLU with I = mode, mode < 0 = RoPE, 0 = none
__device__ __forceinline__ int gather_row(int r, int mode) { return mode > 0 ? swiglu_row(r, mode) : (mode < 0 ? rope_row(r) : r); }

// NW = number of waves that issue the tile's DMA (8 = all; 4 = waves 0-3 only, which staggers the two waves of a SIMD)
template <bool KC, int XR, int NW>
__device__ __forceinline__ void dma_tile(unsigned tile, const SRsrc& rs, int ld, int k0, int Ktot, int swi_I = 0) {
  const int l = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int PPW = XR / (8 * NW);      // 1-KiB pieces per issuing wave
  constexpr int CALLS = PPW > 4 ? PPW / 4 : 1, PER = PPW > 4 ? 4 : PPW;
  if (w >= NW) return;
#pragma unroll
  for (int c = 0; c < CALLS; ++c) {
    unsigned offs[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int pc = w + NW * (c * PER + i);
      if constexpr (KC) {                 // piece = 8 rows x 128 B
        const int row = pc * 8 + (l >> 3);
        const int kc = (l & 7) ^ kc_swz(row);           // source chunk whose home is slot (l&7) of this row
        const int srow = gather_row(row, swi_I);
        unsigned off = (unsigned)(srow * ld + k0 + kc * 8) * 2u;
        if ((k0 + kc * 8) >= Ktot) off = 0xFFFFFFFFu;
        offs[i] = off;
      } else {
        constexpr int SPR = XR / 8;       // 16-B slots per k-row (32 or 16)
        constexpr int RPP = 64 / SPR;     // k-rows per piece (2 or 4)
        const int k = pc * RPP + l / SPR;
        const int sl = l % SPR;
        const int c32 = ((sl >> 1) - ks_swz(k)) & (XR / 16 - 1);   // rotation: the source stays two ascending runs per row
        offs[i] = ((unsigned)(k0 + k) * (unsigned)ld + (unsigned)(c32 * 16 + (sl & 1) * 8)) * 2u;   // < 4 GiB by mm_gemm's check
      }
    }
    lds_dma16xN<PER, NW * 1024>(rs, offs, tile + (w + NW * c * PER) * 1024);
  }
}

// ---- loop-invariant form of dma_tile: the per-lane offsets of a wave's pieces do not depend on the K-step (the
// descriptor is anchored at the tile origin), so they are computed ONCE per workgroup and the K-step enters through the
// instruction's SGPR soffset (k0*2 bytes for a K-contiguous operand, k0*ld*2 for a K-strided one).  soffset is not part
// of the hardware range check, so this form is only used for K-steps that lie wholly inside K (the ragged last step
// goes through dma_tile, whose offsets carry the K bound).
template <bool KC, int XR, int NW>
__device__ __forceinline__ void dma_offsets(unsigned (&offs)[XR / (8 * NW)], int ld, int swi_I = 0, int lane = -1) {
  const int l = lane >= 0 ? lane : (int)(threadIdx.x & 63);
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int PPW = XR / (8 * NW);
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pc = w + NW * i;
    if constexpr (KC) {
      const int row = pc * 8 + (l >> 3);
      const int kc = (l & 7) ^ kc_swz(row);
      offs[i] = (unsigned)(gather_row(row, swi_I) * ld + kc * 8) * 2u;
    } else {
      constexpr int SPR = XR / 8, RPP = 64 / SPR;
      const int k = pc * RPP + l / SPR;
      const int sl = l % SPR;
      const int c32 = ((sl >> 1) - ks_swz(k)) & (XR / 16 - 1);
      offs[i] = ((unsigned)k * (unsigned)ld + (unsigned)(c32 * 16 + (sl & 1) * 8)) * 2u;
    }
  }
}

template <int STEP>
__device__ __forceinline__ void lds_dma16x4s(const SRsrc& r, unsigned v0, unsigned v1, unsigned v2, unsigned v3,
                                             unsigned lds_addr0, unsigned soff) {
  u32x4 d = {r.w0, r.w1, r.w2, r.w3};
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %5\n\t"
      "s_nop 4\n\t"
      "buffer_load_dwordx4 %1, %6, %8 offen lds\n\t"
      "s_add_u32 m0, m0, %7\n\t"
      "buffer_load_dwordx4 %2, %6, %8 offen lds\n\t"
      "s_add_u32 m0, m0, %7\n\t"
      "buffer_load_dwordx4 %3, %6, %8 offen lds\n\t"
      "s_add_u32 m0, m0, %7\n\t"
      "buffer_load_dwordx4 %4, %6, %8 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(lds_addr0), "s"(d), "i"(STEP), "s"(soff)
      : "memory", "scc");
}

template <int XR, int NW>
__device__ __forceinline__ void dma_tile_inv(unsigned tile, const SRsrc& rs, const unsigned (&offs)[XR / (8 * NW)],
                                             unsigned soff) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int PPW = XR / (8 * NW);
  static_assert(PPW % 4 == 0, "pieces per wave");
  if (w >= NW) return;
#pragma unroll
  for (int c = 0; c < PPW / 4; ++c)
    lds_dma16x4s<NW * 1024>(rs, offs[4 * c], offs[4 * c + 1], offs[4 * c + 2], offs[4 * c + 3],
                            tile + (w + NW * c * 4) * 1024, soff);
}

template <bool KC, int XR>
__device__ __forceinline__ bf16x8 frag_load2(const char* tile, int xb, int ks) {
  const int l = threadIdx.x & 63;
  if constexpr (KC) {
    const int row = xb * 16 + (l & 15), kc = ks * 4 + (l >> 4);
    return *(const bf16x8*)(tile + row * 128 + ((kc ^ kc_swz(row)) * 16));
  } else {
    constexpr int ROWB = XR * 2;
    const int g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
    const int k = ks * 32 + 8 * g + q;
    const int sw = ((xb + ks_swz(k)) & (XR / 16 - 1)) * 32 + p * 8;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, tile + k * ROWB + sw));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, tile + (k + 4) * ROWB + sw));
    bf16x8 o;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
    o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
    return o;
  }
}

// BMxBN block tile, 8 waves as WGM x WGN, STAGES-deep LDS ring (3: counted vmcnt keeps one tile in flight across the
// barrier; 2: the next tile's DMA is issued right after the barrier and has one whole compute phase to land).
template <bool A_KC, bool B_KC, int BM_, int BN_, int WGM, int STAGES, int ISSUE_WAVES, int EK = 0>
__global__ __launch_bounds__(512, 2) void gemm_bf16_dma_kernel(GemmArgs g) {
  constexpr int WGN = 8 / WGM;
  constexpr int MREP = BM_ / WGM / 16, NREP = BN_ / WGN / 16;
  constexpr int A_BYTES = BM_ * G_BK * 2, B_BYTES = BN_ * G_BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bf16* A = (const bf16*)g.A;
  const bf16* B = (const bf16*)g.B;
  const int w = threadIdx.x >> 6;
  const int wm = w / WGN, wn = w % WGN;
  const int nk = (g.K + G_BK - 1) / G_BK;
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(char, smem);     // LDS byte address of the ring
  const int total = g.nbm * g.nbn;

  // PERSISTENT workgroups: each walks tiles blockIdx.x, +gridDim.x, ... and treats their K-steps as ONE stream through
  // the LDS ring, so the first DMA of the next tile is already in flight while this tile's epilogue stores run, and
  // there is no per-tile dispatch / prologue bubble.  (tile id) & 7 == blockIdx.x & 7, so the XCD grouping of
  // block_to_tile is preserved.
  // WAVE QUANTISATION.  T tiles on a 256-workgroup persistent grid take ceil(T/256) rounds: 384 tiles (the 8B model's
  // q/k/v wgrad) run 2 rounds at 75 % occupancy, 896 (down_proj wgrad) 4 rounds at 87.5 %.  When the remainder R = T mod
  // grid is at most half the grid, the last R tiles are each cut into two 256x128 halves (g.tail = R), handed to
  // workgroups 0 .. 2R-1 after their full tiles: R half-rounds instead of one more full round, no cross-workgroup sum.
  const int total_full = total - g.tail;
  int tile = blockIdx.x;
  int pm = 0, pn = 0;
  if (tile < total_full) block_to_tile(tile, g.nbm, g.nbn, pm, pn);
  // EK (epilogue kind) is a TEMPLATE parameter: 0 plain without activation code, 1 plain with the GELU kinds (NT: a forward
  // Linear with an activation), 2 SwiGLU backward (mm_gemm_swiglu_bwd, NN), 3 fused gate|up
  // (mm_gemm_swiglu_fwd, NT 256x256).  With the rare epilogues inlined behind runtime branches every GEMM of the step ran
  // 1.6 % slower than the round-1 library on the same box (tools/gemm_bench.py, 15 shapes); with them in their own
  // instantiations the plain kernel is the round-1 kernel again.
  const int swi = (EK == 3) ? g.swi_I : (EK == 4 ? -1 : 0);              // B-row gather: fused SwiGLU (a tile = 128 features, gate + up) / RoPE
  const int nstep = swi > 0 ? BN_ / 2 : BN_;
  int m0 = pm * BM_, n0 = pn * nstep;
  SRsrc ra = tile_rsrc<A_KC>(A, g.lda, m0, g.M, g.K);
  SRsrc rb = tile_rsrc<B_KC>(B, g.ldb, n0, g.N, g.K);
  constexpr bool INV = (BM_ / (8 * ISSUE_WAVES)) % 4 == 0 && (BN_ / (8 * ISSUE_WAVES)) % 4 == 0;
  unsigned offa[INV ? BM_ / (8 * ISSUE_WAVES) : 1], offb[INV ? BN_ / (8 * ISSUE_WAVES) : 1];
  if constexpr (INV) {
    dma_offsets<A_KC, BM_, ISSUE_WAVES>(offa, g.lda);
    dma_offsets<B_KC, BN_, ISSUE_WAVES>(offb, g.ldb, swi);
  }
  const int nk_full = g.K / G_BK;                                     // K-steps wholly inside K
  const unsigned sa = A_KC ? (unsigned)(G_BK * 2) : (unsigned)(G_BK * 2) * (unsigned)g.lda;   // soffset per K-step
  const unsigned sb = B_KC ? (unsigned)(G_BK * 2) : (unsigned)(G_BK * 2) * (unsigned)g.ldb;
  auto issue = [&](const SRsrc& da, const SRsrc& db, int t, int stage) {
    const unsigned st = lds0 + (unsigned)(stage * STAGE_BYTES);
    if constexpr (INV) {
      if (t < nk_full) {
        dma_tile_inv<BM_, ISSUE_WAVES>(st, da, offa, (unsigned)t * sa);
        dma_tile_inv<BN_, ISSUE_WAVES>(st + A_BYTES, db, offb, (unsigned)t * sb);
        return;
      }
    }
    dma_tile<A_KC, BM_, ISSUE_WAVES>(st, da, g.lda, t * G_BK, g.K);
    dma_tile<B_KC, BN_, ISSUE_WAVES>(st + A_BYTES, db, g.ldb, t * G_BK, g.K, swi);
  };
  static_assert(STAGES == 2, "the persistent stream below is written for the 2-stage ring");
  int sidx = 0;                                                       // global K-step counter (ring position)
  if (tile < total_full) issue(ra, rb, 0, 0);
  while (tile < total_full) {
    f32x4 acc[MREP][NREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
      for (int j = 0; j < NREP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int next = tile + gridDim.x;
    int nm0 = 0, nn0 = 0;
    SRsrc nra = ra, nrb = rb;
    if (next < total_full) {
      int qm, qn;
      block_to_tile(next, g.nbm, g.nbn, qm, qn);
      nm0 = qm * BM_;
      nn0 = qn * nstep;
      nra = tile_rsrc<A_KC>(A, g.lda, nm0, g.M, g.K);
      nrb = tile_rsrc<B_KC>(B, g.ldb, nn0, g.N, g.K);
    }
    for (int t = 0; t < nk; ++t, ++sidx) {
#ifndef MM_GEMM_DIAG_NOWAIT
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      __builtin_amdgcn_s_barrier();
      const char* cur = smem + (sidx & 1) * STAGE_BYTES;
#if defined(MM_GEMM_DIAG_NODMA)
#elif defined(MM_GEMM_DIAG_HOTK)
      if (t + 1 < nk) issue(ra, rb, 0, (sidx + 1) & 1);
      else if (next < total_full) issue(ra, rb, 0, (sidx + 1) & 1);
#else
      if (t + 1 < nk) issue(ra, rb, t + 1, (sidx + 1) & 1);
      else if (next < total_full) issue(nra, nrb, 0, (sidx + 1) & 1);
#endif
      // 4 phases of (MREP/2 x NREP) MFMAs; the fragments of phase p+1 are requested BEFORE phase p's MFMAs are issued, so
      // inside a wave the LDS latency of all but the first phase hides under 16 MFMAs (+1.3 % over read-then-multiply).
      {
        constexpr int HM = MREP / 2;
        static_assert(MREP % 2 == 0, "two A halves per k-step");
        bf16x8 fbq[2][NREP], faq[2][HM];
#pragma unroll
        for (int j = 0; j < NREP; ++j) fbq[0][j] = frag_load2<B_KC, BN_>(cur + A_BYTES, wn * NREP + j, 0);
#pragma unroll
        for (int i = 0; i < HM; ++i) faq[0][i] = frag_load2<A_KC, BM_>(cur, wm * MREP + i, 0);
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
          const int ks = ph >> 1, hf = ph & 1;
          if (ph < 3) {
            const int nks = (ph + 1) >> 1, nhf = (ph + 1) & 1;
            if (nhf == 0) {
#pragma unroll
              for (int j = 0; j < NREP; ++j) fbq[nks & 1][j] = frag_load2<B_KC, BN_>(cur + A_BYTES, wn * NREP + j, nks);
            }
#pragma unroll
            for (int i = 0; i < HM; ++i) faq[(ph + 1) & 1][i] = frag_load2<A_KC, BM_>(cur, wm * MREP + nhf * HM + i, nks);
          }
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int i = 0; i < HM; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j)
              acc[hf * HM + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbq[ks & 1][j], faq[ph & 1][i], acc[hf * HM + i][j], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if constexpr (EK == 4) gemm_epilogue_rope<MREP, NREP>(g, acc, m0 + wm * (BM_ / WGM), n0, wn);
    else if constexpr (EK == 3) gemm_epilogue_swiglu<MREP, NREP>(g, acc, m0 + wm * (BM_ / WGM), n0 + wn * (BN_ / WGN / 2));
    else if constexpr (EK == 2) gemm_epilogue_ek2<MREP, NREP>(g, acc, m0 + wm * (BM_ / WGM), n0 + wn * (BN_ / WGN));
    else if constexpr (EK == 0) gemm_epilogue_ek0<MREP, NREP>(g, acc, m0 + wm * (BM_ / WGM), n0 + wn * (BN_ / WGN));
    else if constexpr (EK == 5)
      gemm_epilogue_plain_pipe<MREP, NREP, true>(g, acc, m0 + wm * (BM_ / WGM), n0 + wn * (BN_ / WGN),
                                                 __builtin_amdgcn_readfirstlane(((m0 / BM_) * g.nbn + n0 / BN_) * 16 + w));
    else gemm_epilogue_plain<MREP, NREP, (BM_ * BN_ <= 128 * 128) && EK == 1, EK == 1>(g, acc, m0 + wm * (BM_ / WGM), n0 + wn * (BN_ / WGN));
    tile = next;
    m0 = nm0;
    n0 = nn0;
    ra = nra;
    rb = nrb;
  }
  // ---- the half-tile round (256 x BN/2): same A tile and image, B tile of half the rows; not chained to the stream above
  if constexpr (BM_ == 256 && BN_ == 256 && (NREP % 2) == 0 && EK != 4) {
    if (g.tail > 0 && (int)blockIdx.x < 2 * g.tail) {
      constexpr int BNH = BN_ / 2, NREPH = NREP / 2, HM = MREP / 2;
      constexpr bool INVH = INV && (BNH / (8 * ISSUE_WAVES)) % 4 == 0;
      int hm, hn;
      block_to_tile(total_full + ((int)blockIdx.x >> 1), g.nbm, g.nbn, hm, hn);
      const int hm0 = hm * BM_, hn0 = hn * BN_ + ((int)blockIdx.x & 1) * BNH;
      const SRsrc ha = tile_rsrc<A_KC>(A, g.lda, hm0, g.M, g.K);
      const SRsrc hb = tile_rsrc<B_KC>(B, g.ldb, hn0, g.N, g.K);
      unsigned offbh[INVH ? BNH / (8 * ISSUE_WAVES) : 1];
      if constexpr (INVH) dma_offsets<B_KC, BNH, ISSUE_WAVES>(offbh, g.ldb);
      auto issue_h = [&](int t, int stage) {
        const unsigned st = lds0 + (unsigned)(stage * STAGE_BYTES);
        if constexpr (INVH) {
          if (t < nk_full) {
            dma_tile_inv<BM_, ISSUE_WAVES>(st, ha, offa, (unsigned)t * sa);
            dma_tile_inv<BNH, ISSUE_WAVES>(st + A_BYTES, hb, offbh, (unsigned)t * sb);
            return;
          }
        }
        dma_tile<A_KC, BM_, ISSUE_WAVES>(st, ha, g.lda, t * G_BK, g.K);
        dma_tile<B_KC, BNH, ISSUE_WAVES>(st + A_BYTES, hb, g.ldb, t * G_BK, g.K);
      };
      f32x4 acch[MREP][NREPH];
#pragma unroll
      for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < NREPH; ++j) acch[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      __builtin_amdgcn_s_barrier();            // every wave has left the ring of the last full tile
      issue_h(0, 0);
      for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const char* cur = smem + (t & 1) * STAGE_BYTES;
        if (t + 1 < nk) issue_h(t + 1, (t + 1) & 1);
        bf16x8 fbq[2][NREPH], faq[2][HM];
#pragma unroll
        for (int j = 0; j < NREPH; ++j) fbq[0][j] = frag_load2<B_KC, BNH>(cur + A_BYTES, wn * NREPH + j, 0);
#pragma unroll
        for (int i = 0; i < HM; ++i) faq[0][i] = frag_load2<A_KC, BM_>(cur, wm * MREP + i, 0);
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
          const int ks = ph >> 1, hf = ph & 1;
          if (ph < 3) {
            const int nks = (ph + 1) >> 1, nhf = (ph + 1) & 1;
            if (nhf == 0) {
#pragma unroll
              for (int j = 0; j < NREPH; ++j) fbq[nks & 1][j] = frag_load2<B_KC, BNH>(cur + A_BYTES, wn * NREPH + j, nks);
            }
#pragma unroll
            for (int i = 0; i < HM; ++i) faq[(ph + 1) & 1][i] = frag_load2<A_KC, BM_>(cur, wm * MREP + nhf * HM + i, nks);
          }
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int i = 0; i < HM; ++i)
#pragma unroll
            for (int j = 0; j < NREPH; ++j)
              acch[hf * HM + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbq[ks & 1][j], faq[ph & 1][i], acch[hf * HM + i][j], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (EK == 2) gemm_epilogue_ek2<MREP, NREPH>(g, acch, hm0 + wm * (BM_ / WGM), hn0 + wn * (BNH / WGN));
      else if constexpr (EK == 0) gemm_epilogue_ek0<MREP, NREPH>(g, acch, hm0 + wm * (BM_ / WGM), hn0 + wn * (BNH / WGN));
      else if constexpr (EK == 5)
        gemm_epilogue_plain_pipe<MREP, NREPH, true>(g, acch, hm0 + wm * (BM_ / WGM), hn0 + wn * (BNH / WGN),
                                                    __builtin_amdgcn_readfirstlane((hm * g.nbn + hn) * 16 + ((int)blockIdx.x & 1) * 8 + w));
      else gemm_epilogue_plain<MREP, NREPH, false, EK == 1>(g, acch, hm0 + wm * (BM_ / WGM), hn0 + wn * (BNH / WGN));
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// The same 256x256x64 tile on FOUR waves (2 x 2), one wave per SIMD, 128 x 128 of the tile per wave (round 4).
// The 8-wave form above moves 8 x 24 KB of fragment reads + 64 KB of LDS-DMA writes through the LDS port per K-step: as many port
// clocks as the K-step's MFMAs take; four waves of 128 x 128 read 4 x 32 KB.  That shape needs the whole
// 512-entry register file of a SIMD (256 accumulators + two sets of 16 fragments) and an instruction order hipcc does not
// produce (round 3: ~170 spilt registers), so the K loop of one output tile is ONE asm statement with literal registers,
// generated by gen_gemm_w4.py (stream structure, register map and hazards: that file's header).  Everything around it is the code
// above: the LDS images and DMA source swizzles (dma_offsets), the persistent XCD-aware tile map, the buffer descriptors with the
// hardware range check, and the epilogues, which see a wave as two of the 8-wave form's 128 x 64 wave tiles (columns 64 c ..).
// Same products in the same K order as the 8-wave kernel: bit-identical output.  A is K-contiguous (NT, NN); K % 64 == 0 and
// K >= 192 (the host sends anything else to the 8-wave kernel).
// ------------------------------------------------------------------------------------------------------
#include "mm_gemm_w4.inc"
#ifndef MM_W4_DIAG
#define W4_STAMP(st, i) do { } while (0)
#endif
#ifdef MM_W4_DIAG
__device__ unsigned g_w4_diag2[256 * 4 * 8];       // stamps inside the register-exchange epilogue (W4_STAMP): cycles since the loop's end, summed over tiles
#define W4_STAMP(st, i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory"); (st)[i] = (unsigned)t_; } while (0)
__device__ unsigned g_w4_diag[256 * 4 * 7];        // schedule 121: per (workgroup, wave) cycles at the barrier (sum, max), loop cycles, tiles
#endif

// ---- row-major epilogue of the 4-wave kernel (EK == 0: bias / residual / accumulate) -------------------------------------------
// In the accumulator layout a lane owns 4 consecutive columns of a row and a wave-wide store covers 16 rows x 32 bytes: 64 such
// stores per wave and tile, each split into 16 partial-line writes.  Stamps (tools/w4_stamps.py) put the 256x256 tile's epilogue at
// ~21 000 cycles -- 13 % of a K = 4096 tile -- almost all of it store ISSUE.  Here 32 rows x 64 columns of fp32 accumulators at a
// time go through the wave's 8 KB of spare LDS (XOR-swizzled 16-byte chunks: conflict-free both ways) and come back with a lane
// owning 8 consecutive columns: residual / C are read and C is written 16 bytes per lane, a wave-wide access = 8 rows x one full
// 128-byte line.  Arithmetic and rounding points are those of gemm_epilogue_plain_pipe (acc + bias + residual + C, one rounding):
// bit-identical output.  Rows >= M and column groups >= N fall to the buffer range check (the host requires N % 8 == 0).
template <int IP, int C, int MODE>
__device__ __forceinline__ void w4_rm_block(const GemmArgs& g, char* xp, __amdgpu_buffer_rsrc_t rc, __amdgpu_buffer_rsrc_t rr, unsigned colb,
                                            const float (&bv)[8], bool has_bias, int l) {
  constexpr bool has_res = MODE == 1, has_acc = MODE == 2;
  const int wr = l & 15, wq = l >> 4, r8 = l >> 3, c8 = l & 7;
  {   // the block leaves the accumulator registers for LDS without passing through VGPRs (w4_store_acc_blk)
    const unsigned xb = (unsigned)(uintptr_t)LDS_PTR(char, xp) + (unsigned)(wr * 256);
    __builtin_amdgcn_wave_barrier();
#ifdef MM_W4_DIAG
    if (!(g.diag_epi & 4))
#endif
    w4_store_acc_blk<IP, C>(xb + (unsigned)(((0 + wq) ^ wr) * 16), xb + (unsigned)(((4 + wq) ^ wr) * 16), xb + (unsigned)(((8 + wq) ^ wr) * 16),
                            xb + (unsigned)(((12 + wq) ^ wr) * 16));
    __builtin_amdgcn_wave_barrier();
  }
  u32x4 rbuf[4], cbuf[4];
  unsigned off[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = IP * 32 + it * 8 + r8;
    off[it] = colb == EPI_OOB ? EPI_OOB : (unsigned)(row * g.ldc) * 2u + colb;
    if (has_res) rbuf[it] = __builtin_amdgcn_raw_buffer_load_b128(rr, colb == EPI_OOB ? EPI_OOB : (unsigned)(row * g.ldr) * 2u + colb, 0, 0);
    if (has_acc) cbuf[it] = __builtin_amdgcn_raw_buffer_load_b128(rc, off[it], 0, 0);
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + r8;
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
#ifdef MM_W4_DIAG
    if (!(g.diag_epi & 2))
#endif
    {
      lo = *(const f32x4*)(xp + row * 256 + (((2 * c8) ^ (row & 15)) * 16));
      hi = *(const f32x4*)(xp + row * 256 + (((2 * c8 + 1) ^ (row & 15)) * 16));
    }
    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    if (has_bias) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] += bv[r];
    }
    if (has_res) {
      const bf16x8 rv = __builtin_bit_cast(bf16x8, rbuf[it]);
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] += (float)rv[r];
    }
    if (has_acc) {
      const bf16x8 cv = __builtin_bit_cast(bf16x8, cbuf[it]);
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] += (float)cv[r];
    }
    bf16x8 o;
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = (bf16)v[r];
#ifdef MM_W4_DIAG
    if (g.diag_epi & 1) { asm volatile("" ::"v"(o)); continue; }
#endif
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rc, off[it], 0, 0);
  }
  __builtin_amdgcn_wave_barrier();
}

template <int C, int MODE>
__device__ __forceinline__ void w4_rm_half(const GemmArgs& g, char* xp, __amdgpu_buffer_rsrc_t rc, __amdgpu_buffer_rsrc_t rr, int nw, bool has_bias, int l) {
  const int n = nw + C * 64 + (l & 7) * 8;                  // the lane's 8 columns (N % 8 == 0: whole or absent)
  const unsigned colb = n < g.N ? (unsigned)n * 2u : EPI_OOB;
  float bv[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) bv[r] = 0.f;
  if (has_bias && n < g.N) {
    const bf16x8 b8 = *(const bf16x8*)((const bf16*)g.bias + n);
#pragma unroll
    for (int r = 0; r < 8; ++r) bv[r] = (float)b8[r];
  }
  w4_rm_block<0, C, MODE>(g, xp, rc, rr, colb, bv, has_bias, l);
  w4_rm_block<1, C, MODE>(g, xp, rc, rr, colb, bv, has_bias, l);
  w4_rm_block<2, C, MODE>(g, xp, rc, rr, colb, bv, has_bias, l);
  w4_rm_block<3, C, MODE>(g, xp, rc, rr, colb, bv, has_bias, l);
}

// 32 rows x 64 columns of the wave's fp32 accumulators through its 8 KB of spare LDS: out v[it][0..7] = row 8 it + (l >> 3), columns
// 8 (l & 7) .. + 7 of the block (rows 32 IP.., columns 64 C.. of the wave tile)
template <int IP, int C>
__device__ __forceinline__ void w4_xpose_blk(char* xp, int l, float (&v)[4][8]) {
  const int wr = l & 15, wq = l >> 4, r8 = l >> 3, c8 = l & 7;
  {   // the block leaves the accumulator registers for LDS without passing through VGPRs (w4_store_acc_blk)
    const unsigned xb = (unsigned)(uintptr_t)LDS_PTR(char, xp) + (unsigned)(wr * 256);
    __builtin_amdgcn_wave_barrier();
    w4_store_acc_blk<IP, C>(xb + (unsigned)(((0 + wq) ^ wr) * 16), xb + (unsigned)(((4 + wq) ^ wr) * 16), xb + (unsigned)(((8 + wq) ^ wr) * 16),
                            xb + (unsigned)(((12 + wq) ^ wr) * 16));
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + r8;
    const f32x4 lo = *(const f32x4*)(xp + row * 256 + (((2 * c8) ^ (row & 15)) * 16));
    const f32x4 hi = *(const f32x4*)(xp + row * 256 + (((2 * c8 + 1) ^ (row & 15)) * 16));
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[it][r] = lo[r]; v[it][4 + r] = hi[r]; }
  }
  __builtin_amdgcn_wave_barrier();
}

// SwiGLU-backward epilogue (mm_gemm_swiglu_bwd, EK == 2) of the 4-wave kernel in row-major form: per 32 x 64 block a lane reads the saved
// gate / up pre-activations of its 8 columns (2 x 16 bytes per row) and writes d(gate), d(up) (2 x 16 bytes): a wave-wide access = 8 rows
// x one 128-byte line, where the accumulator layout gave 16 rows x 32 bytes and four times as many instructions.  The loads of block
// b + 1 are requested before block b's stores, so a wait for them never waits for a store.  Arithmetic of gemm_epilogue_swiglu_bwd_pipe
// element for element: bit-identical.  Host: N (= I) % 8 == 0, 16-byte aligned C / aux rows.
struct W4Swb { u32x4 g[4], u[4]; };
template <int B8>
__device__ __forceinline__ void w4_swb_request(const GemmArgs& g, __amdgpu_buffer_rsrc_t ra, int nw, int l, W4Swb& q) {
  constexpr int IP = B8 & 3, C = B8 >> 2;
  const int n = nw + C * 64 + (l & 7) * 8;
  const unsigned upo = (unsigned)g.N * 2u;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = IP * 32 + it * 8 + (l >> 3);
    const unsigned o = n < g.N ? (unsigned)(row * g.ldaux) * 2u + (unsigned)n * 2u : EPI_OOB;
    q.g[it] = __builtin_amdgcn_raw_buffer_load_b128(ra, o, 0, 0);
    q.u[it] = __builtin_amdgcn_raw_buffer_load_b128(ra, o == EPI_OOB ? EPI_OOB : o + upo, 0, 0);
  }
}
template <int B8>
__device__ __forceinline__ void w4_swb_block(const GemmArgs& g, char* xp, __amdgpu_buffer_rsrc_t rc, int nw, int l, const W4Swb& q) {
  constexpr int IP = B8 & 3, C = B8 >> 2;
  float v[4][8];
  w4_xpose_blk<IP, C>(xp, l, v);
  const int n = nw + C * 64 + (l & 7) * 8;
  const unsigned upo = (unsigned)g.N * 2u;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = IP * 32 + it * 8 + (l >> 3);
    const bf16x8 gv = __builtin_bit_cast(bf16x8, q.g[it]), uv = __builtin_bit_cast(bf16x8, q.u[it]);
    bf16x8 dg, du;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float gf = (float)gv[r], sig = 1.0f / (1.0f + __expf(-gf));
      const float sg = gf * sig;
      const float dd = (float)(bf16)v[it][r];
      du[r] = (bf16)(dd * sg);
      dg[r] = (bf16)(dd * (float)uv[r] * (sig * (1.0f + gf * (1.0f - sig))));
    }
    const unsigned o = n < g.N ? (unsigned)(row * g.ldc) * 2u + (unsigned)n * 2u : EPI_OOB;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dg), rc, o, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, du), rc, o == EPI_OOB ? EPI_OOB : o + upo, 0, 0);
  }
}
__device__ __forceinline__ void w4_epilogue_swiglu_bwd_rowmajor(const GemmArgs& g, char* xp, int mw, int nw, int l) {
  mw = __builtin_amdgcn_readfirstlane(mw);
  nw = __builtin_amdgcn_readfirstlane(nw);
  const int rows = g.M - mw;
  auto rc = make_rsrc((const bf16*)g.C + (int64_t)mw * g.ldc, (int64_t)rows * g.ldc * 2);
  auto ra = make_rsrc((const bf16*)g.aux + (int64_t)mw * g.ldaux, (int64_t)rows * g.ldaux * 2);
  W4Swb q0, q1;
  w4_swb_request<0>(g, ra, nw, l, q0);
  w4_swb_request<1>(g, ra, nw, l, q1); w4_swb_block<0>(g, xp, rc, nw, l, q0);
  w4_swb_request<2>(g, ra, nw, l, q0); w4_swb_block<1>(g, xp, rc, nw, l, q1);
  w4_swb_request<3>(g, ra, nw, l, q1); w4_swb_block<2>(g, xp, rc, nw, l, q0);
  w4_swb_request<4>(g, ra, nw, l, q0); w4_swb_block<3>(g, xp, rc, nw, l, q1);
  w4_swb_request<5>(g, ra, nw, l, q1); w4_swb_block<4>(g, xp, rc, nw, l, q0);
  w4_swb_request<6>(g, ra, nw, l, q0); w4_swb_block<5>(g, xp, rc, nw, l, q1);
  w4_swb_request<7>(g, ra, nw, l, q1); w4_swb_block<6>(g, xp, rc, nw, l, q0);
  w4_swb_block<7>(g, xp, rc, nw, l, q1);
}

// Fused gate|up epilogue (mm_gemm_swiglu_fwd, EK == 3) of the 4-wave kernel through the same transposition: of a block's 64 columns the
// first 32 are gate pre-activations of 32 features and the last 32 the up pre-activations of the SAME features (swiglu_row); a lane
// takes 4 features of a row -- the gate chunk and its up chunk -- and writes gate, up and silu(gate) * up as 8-byte pieces, 8 lanes = 64
// contiguous bytes per row and output (the accumulator layout: 32 bytes, behind bounds branches).  gemm_epilogue_swiglu's arithmetic.
template <int IP, int C>
__device__ __forceinline__ void w4_swf_block(const GemmArgs& g, char* xp, __amdgpu_buffer_rsrc_t rgu, __amdgpu_buffer_rsrc_t ract, int fw, int l) {
  const int wr = l & 15, wq = l >> 4, r8 = l >> 3, c8 = l & 7;
  {   // the block leaves the accumulator registers for LDS without passing through VGPRs (w4_store_acc_blk)
    const unsigned xb = (unsigned)(uintptr_t)LDS_PTR(char, xp) + (unsigned)(wr * 256);
    __builtin_amdgcn_wave_barrier();
    w4_store_acc_blk<IP, C>(xb + (unsigned)(((0 + wq) ^ wr) * 16), xb + (unsigned)(((4 + wq) ^ wr) * 16), xb + (unsigned)(((8 + wq) ^ wr) * 16),
                            xb + (unsigned)(((12 + wq) ^ wr) * 16));
    __builtin_amdgcn_wave_barrier();
  }
  const int f = fw + 4 * c8;
  const unsigned I2 = (unsigned)g.swi_I * 2u;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + r8;
    const f32x4 ga = *(const f32x4*)(xp + row * 256 + ((c8 ^ (row & 15)) * 16));
    const f32x4 ua = *(const f32x4*)(xp + row * 256 + (((8 + c8) ^ (row & 15)) * 16));
    bf16x4 gb, ub, o;
#pragma unroll
    for (int r = 0; r < 4; ++r) { gb[r] = (bf16)ga[r]; ub[r] = (bf16)ua[r]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gf = (float)gb[r];
      const float sg = (float)(bf16)(gf / (1.0f + __expf(-gf)));     // HF: act_fn(gate) in the storage dtype, then multiply
      o[r] = (bf16)(sg * (float)ub[r]);
    }
    const int grow = IP * 32 + row;
    const unsigned og = f < g.swi_I ? (unsigned)(grow * g.ldc) * 2u + (unsigned)f * 2u : EPI_OOB;
    const unsigned oa = f < g.swi_I ? (unsigned)(grow * g.ldc2) * 2u + (unsigned)f * 2u : EPI_OOB;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, gb), rgu, og, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ub), rgu, og == EPI_OOB ? EPI_OOB : og + I2, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), ract, oa, 0, 0);
  }
  __builtin_amdgcn_wave_barrier();
}
template <int C>
__device__ __forceinline__ void w4_swf_half(const GemmArgs& g, char* xp, __amdgpu_buffer_rsrc_t rgu, __amdgpu_buffer_rsrc_t ract, int fw, int l) {
  w4_swf_block<0, C>(g, xp, rgu, ract, fw, l);
  w4_swf_block<1, C>(g, xp, rgu, ract, fw, l);
  w4_swf_block<2, C>(g, xp, rgu, ract, fw, l);
  w4_swf_block<3, C>(g, xp, rgu, ract, fw, l);
}
__device__ __forceinline__ void w4_epilogue_swiglu_fwd_rowmajor(const GemmArgs& g, char* xp, int mw, int n0, int wn, int l) {
  mw = __builtin_amdgcn_readfirstlane(mw);
  const int rows = g.M - mw;
  auto rgu = make_rsrc((const bf16*)g.C + (int64_t)mw * g.ldc, (int64_t)rows * g.ldc * 2);
  auto ract = make_rsrc((const bf16*)g.C2 + (int64_t)mw * g.ldc2, (int64_t)rows * g.ldc2 * 2);
  w4_swf_half<0>(g, xp, rgu, ract, __builtin_amdgcn_readfirstlane(n0 + (2 * wn) * 32), l);
  w4_swf_half<1>(g, xp, rgu, ract, __builtin_amdgcn_readfirstlane(n0 + (2 * wn + 1) * 32), l);
}

template <int MODE>
__device__ __forceinline__ void w4_epilogue_rowmajor(const GemmArgs& g, char* xp, int mw, int nw, int l) {
  mw = __builtin_amdgcn_readfirstlane(mw);
  nw = __builtin_amdgcn_readfirstlane(nw);
  const bool has_bias = (g.epi & MM_EPI_BIAS) != 0;
  const int rows = g.M - mw;
  auto rc = make_rsrc((const bf16*)g.C + (int64_t)mw * g.ldc, (int64_t)rows * g.ldc * 2);
  auto rr = MODE == 1 ? make_rsrc((const bf16*)g.residual + (int64_t)mw * g.ldr, (int64_t)rows * g.ldr * 2) : rc;
  w4_rm_half<0, MODE>(g, xp, rc, rr, nw, has_bias, l);
  w4_rm_half<1, MODE>(g, xp, rc, rr, nw, has_bias, l);
}

// ---- plain stores without the LDS round trip (MODE 0, no bias) -----------------------------------------------------------------
// Measured (tools/probes/gen_epi_probe.py, tools/w4_stamps.py epi=...): a wave alone on its SIMD issues one vector instruction per
// 4.3 cycles; 64 `ds_write_b128` of accumulator registers cost the four waves 3 400 cycles of the LDS port, the 64 reads 1 000; a
// wave-wide 16-byte store costs the CU's store path 16 cycles when it covers 8 rows x one full 128-byte line, and 64 (whatever the
// other CUs do) when it covers 16 rows x 64 bytes.  So: full lines, and no LDS.  A lane owns columns 4q .. 4q+3 (q = l >> 4) of row
// r = l & 15 in each 16 x 16 tile; the accumulators are rounded first and the lanes exchange packed bf16 pairs in registers:
//   level 1, tiles (X, Y) side by side: `v_permlane16_swap` (odd 16-lane rows of X <-> even rows of Y) leaves lane rows q = 0, 2
//            with columns 0-7, 8-15 of X and q = 1, 3 with those of Y: 16 bytes = 8 consecutive columns per lane (P = 32 columns);
//   level 2, blocks (P0, P1) side by side: lanes r >= 8 of P0 <-> lanes r < 8 of P1 inside every 16-lane row (two DPP `row_ror:8`
//            moves with bank masks): P0' = rows 0-7 x all 128 bytes, P1' = rows 8-15 x all 128 bytes.
// Per row block and 64 columns: 16 accumulator reads, 8 conversions, 4 swaps, ~12 DPP moves, 2 stores -- ~700 instructions per wave
// and tile, no LDS traffic, no waits.  The value stored is bf16(acc) either way: bit-identical.
template <int C>
__device__ __forceinline__ void w4_shuffle_half(const GemmArgs& g, __amdgpu_buffer_rsrc_t rc, int nw, int l, unsigned* st = nullptr) {
  f32x4 acc[8][4];
  if constexpr (C == 0) w4_read_acc_c0(acc);
  else w4_read_acc_c1(acc);
  if (st) W4_STAMP(st, 1 + 2 * C);
  const int r = l & 15, q = l >> 4;
  const int chunk = (r >= 8 ? 4 : 0) + (((q & 1) << 1) | (q >> 1));      // the lane's 16-byte chunk of the 128-byte line
  const int n = nw + C * 64 + chunk * 8;
  const unsigned colb = n < g.N ? (unsigned)n * 2u : EPI_OOB;
  const unsigned ld2 = (unsigned)g.ldc * 2u;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    u32x4 P[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const f32x4 x = acc[i][2 * p], y = acc[i][2 * p + 1];
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = (bf16)x[e]; o[4 + e] = (bf16)y[e]; }
      const u32x4 pk = __builtin_bit_cast(u32x4, o);
      const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0], pk[2], false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(pk[1], pk[3], false, false);
      P[p] = u32x4{s0[0], s1[0], s0[1], s1[1]};
    }
    u32x4 lo, hi;                                   // lo: rows 0-7 of the row block, hi: rows 8-15, 128 bytes per row each
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lo[e] = (unsigned)__builtin_amdgcn_update_dpp((int)P[0][e], (int)P[1][e], 0x128, 0xF, 0xC, false);      // row_ror:8 into lanes r >= 8
      hi[e] = (unsigned)__builtin_amdgcn_update_dpp((int)P[1][e], (int)P[0][e], 0x128, 0xF, 0x3, false);      // row_ror:8 into lanes r < 8
    }
    const unsigned off = colb == EPI_OOB ? EPI_OOB : (unsigned)(i * 16 + (r & 7)) * ld2 + colb;
    const unsigned off8 = colb == EPI_OOB ? EPI_OOB : off + 8u * ld2;
#ifdef MM_W4_DIAG
    if (g.diag_epi & 1) { asm volatile("" ::"v"(lo), "v"(hi)); continue; }
#endif
    __builtin_amdgcn_raw_buffer_store_b128(lo, rc, off, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(hi, rc, off8, 0, 0);
  }
  if (st) W4_STAMP(st, 2 + 2 * C);
}

__device__ __forceinline__ void w4_epilogue_shuffle(const GemmArgs& g, int mw, int nw, int l, unsigned* st = nullptr) {
  mw = __builtin_amdgcn_readfirstlane(mw);
  nw = __builtin_amdgcn_readfirstlane(nw);
  auto rc = make_rsrc((const bf16*)g.C + (int64_t)mw * g.ldc, (int64_t)(g.M - mw) * g.ldc * 2);
  if (st) W4_STAMP(st, 0);
  w4_shuffle_half<0>(g, rc, nw, l, st);
  w4_shuffle_half<1>(g, rc, nw, l, st);
}

__device__ __forceinline__ unsigned w4_sgpr(unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); }

template <bool A_KC, bool B_KC, int EK, int SCHED>
__global__ __launch_bounds__(256) void gemm_bf16_w4_kernel(GemmArgs g) {
  static_assert(A_KC || !B_KC, "NT, NN, TN");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bf16* A = (const bf16*)g.A;
  const bf16* B = (const bf16*)g.B;
  const int l = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int nk = (g.K + G_BK - 1) / G_BK;          // a ragged last K-step: rows >= K of a K-strided operand lie beyond the running descriptor's
                                                    // num_records; a K-contiguous operand's lanes beyond K get bit 31 in their offsets (rag_mask)
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(char, smem);
  if (lds0 != 0) __builtin_trap();                  // the ring's stage bit (0x10000) is flipped by XOR on absolute LDS addresses
  const int total_all = g.nbm * g.nbn;
  const int total = total_all - g.tail;             // full tiles; the last g.tail tiles are cut into 256 x 128 halves below (EK == 0)
  int tile = blockIdx.x;
  if (tile >= total && !(g.tail > 0 && (int)blockIdx.x < 2 * g.tail)) return;
  const int swi = (EK == 3) ? g.swi_I : (EK == 4 ? -1 : 0);       // B-row gather of the fused gate|up / RoPE tiles (gather_row)
  const int nstep = swi > 0 ? 128 : 256;
  if (g.stagger > 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), wait = (unsigned long long)g.stagger * ((blockIdx.x >> 3) % (unsigned)g.stagger_slots);
    while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
  }
  int pm, pn;
  block_to_tile(tile, g.nbm, g.nbn, pm, pn, g.group_m);
  int m0 = pm * 256, n0 = pn * nstep;
  SRsrc ra = tile_rsrc<A_KC>(A, g.lda, m0, g.M, g.K);
  SRsrc rb = tile_rsrc<B_KC>(B, g.ldb, n0, g.N, g.K);
  unsigned offa[8], offb[8];
  dma_offsets<A_KC, 256, 4>(offa, g.lda);
  dma_offsets<B_KC, 256, 4>(offb, g.ldb, swi);
  const unsigned lda2 = (unsigned)g.lda * 2u, ldb2 = (unsigned)g.ldb * 2u;
  // K-steps 0 and 1 of the first tile; from here on every tile's asm statement issues K-steps 2.. and the next tile's 0 and 1
  if (tile < total) {
    dma_tile_inv<256, 4>(lds0, ra, offa, 0u);
    dma_tile_inv<256, 4>(lds0 + 32768u, rb, offb, 0u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    dma_tile_inv<256, 4>(lds0 + 65536u, ra, offa, A_KC ? 128u : 64u * lda2);
    dma_tile_inv<256, 4>(lds0 + 65536u + 32768u, rb, offb, B_KC ? 128u : 64u * ldb2);
  }
  // lane parts of the fragment addresses (frag_load2's formulas with the wave's first row block; + 2048 per 16 rows in the asm).
  // Recomputed per tile from an opaque copy of the lane index: as loop invariants they would have to live through the asm statement,
  // where the compiler has v0-v91 only, and came back from scratch behind an s_waitcnt vmcnt(0) that also waits out the epilogue's stores.
  auto frag_addr = [&](int lane_, unsigned& rda_lo, unsigned& rda_hi, unsigned& rdb_lo, unsigned& rdb_hi) {
    const int lr = lane_ & 15, lg = lane_ >> 4;
    if constexpr (A_KC) {
      rda_lo = (unsigned)((wm * 128 + lr) * 128 + (((0 + lg) ^ kc_swz(lr)) * 16));
      rda_hi = (unsigned)((wm * 128 + lr) * 128 + (((4 + lg) ^ kc_swz(lr)) * 16));
    } else {
      const int q = lr >> 2, p = lr & 3, k = 8 * lg + q;
      rda_lo = (unsigned)(k * 512 + p * 8);
      rda_hi = (unsigned)(((wm * 8 + ks_swz(k)) & 15) * 32);
    }
    if constexpr (B_KC) {
      rdb_lo = 32768u + (unsigned)((wn * 128 + lr) * 128 + (((0 + lg) ^ kc_swz(lr)) * 16));
      rdb_hi = 32768u + (unsigned)((wn * 128 + lr) * 128 + (((4 + lg) ^ kc_swz(lr)) * 16));
    } else {                                        // K-strided image: row part, and the rotated 32-byte column slot of fragment 0
      const int q = lr >> 2, p = lr & 3, k = 8 * lg + q;
      rdb_lo = 32768u + (unsigned)(k * 512 + p * 8);
      rdb_hi = (unsigned)(((wn * 8 + ks_swz(k)) & 15) * 32);
    }
  };
  // wave-uniform strides between a wave's DMA pieces (piece i = 8 rows at local row 8 w + 32 i; bits 0, 1, 2 of i)
  const unsigned ta = w4_sgpr(A_KC ? 32u * lda2 : 16u * lda2);       // A is never gathered: 32 rows per piece step (K-strided: 16 k-rows)
  unsigned tb0, tb1 = 0, tb2 = 0;
  if constexpr (B_KC) {
    if (swi > 0) { tb0 = (unsigned)swi * ldb2; tb1 = 32u * ldb2; tb2 = 64u * ldb2; }          // swiglu_row: bit 0 -> + I rows
    else if (swi < 0) { tb0 = 64u * ldb2; tb1 = 32u * ldb2; tb2 = 128u * ldb2; }               // rope_row
    else { tb0 = 32u * ldb2; tb1 = 64u * ldb2; tb2 = 128u * ldb2; }
  } else {
    tb0 = 16u * ldb2;                               // 16 k-rows between pieces of equal parity
  }
  tb0 = w4_sgpr(tb0); tb1 = w4_sgpr(tb1); tb2 = w4_sgpr(tb2);
  const unsigned nk_s = w4_sgpr((unsigned)nk), wv_s = w4_sgpr((unsigned)w);
  const unsigned nkm1_s = w4_sgpr((g.K & 63) ? (unsigned)(nk - 1) : 0xFFFFFFFFu);      // the ragged K-step (K-contiguous operands: gen_gemm_w4.py)
  auto rag_mask = [&](int lane_) -> unsigned {      // bit 31 for the lanes whose source chunk of a K-contiguous row starts at or beyond K
    const int kc = (lane_ & 7) ^ ((((w & 1) << 2) | (lane_ >> 4)) & 7);      // dma_offsets: (l & 7) ^ kc_swz(row), the same for all of a wave's pieces
    return ((g.K & 63) && kc * 8 >= (g.K & 63)) ? 0x80000000u : 0u;
  };
  unsigned sidx = 0;                                // K-steps streamed so far (ring position)
#ifdef MM_W4_DIAG
  unsigned w4_diag_end = 0;
#endif
  while (tile < total) {
    const int next = tile + gridDim.x;
    int nm0 = 0, nn0 = 0;
    unsigned na0 = 0, na1 = 0, na2 = 0, nb0 = 0, nb1 = 0, nb2 = 0;     // no next tile: empty descriptors (every DMA lane out of range)
    if (next < total) {
      int qm, qn;
      block_to_tile(next, g.nbm, g.nbn, qm, qn, g.group_m);
      nm0 = qm * 256;
      nn0 = qn * nstep;
      const SRsrc nra = tile_rsrc<A_KC>(A, g.lda, nm0, g.M, g.K), nrb = tile_rsrc<B_KC>(B, g.ldb, nn0, g.N, g.K);
      na0 = nra.w0; na1 = nra.w1; na2 = nra.w2; nb0 = nrb.w0; nb1 = nrb.w1; nb2 = nrb.w2;
    }
    na0 = w4_sgpr(na0); na1 = w4_sgpr(na1); na2 = w4_sgpr(na2); nb0 = w4_sgpr(nb0); nb1 = w4_sgpr(nb1); nb2 = w4_sgpr(nb2);
    const unsigned st = (sidx & 1u) * 65536u;
    int lane_in = l;
    asm volatile("" : "+v"(lane_in));
    unsigned rda_lo, rda_hi, rdb_lo, rdb_hi;
    frag_addr(lane_in, rda_lo, rda_hi, rdb_lo, rdb_hi);
    unsigned toa[8], tob[8];                        // (the asm rebuilds pieces 1.. / 2.. from the first one or two and the strides)
    dma_offsets<A_KC, 256, 4>(toa, g.lda, 0, lane_in);
    dma_offsets<B_KC, 256, 4>(tob, g.ldb, swi, lane_in);
    const unsigned voffa0 = toa[0], voffa1 = toa[1], voffb0 = tob[0], voffb1 = tob[1];
    const unsigned vrag = rag_mask(lane_in);
    // L2 prefetch (schedules 6, 7; gen_gemm_w4.py `pf`): the lane's line of the wave's share of a K-step.  K-contiguous operand: row
    // (w + 4 (l >> 3)) * 8 + (l & 7) of the tile (the rows of the wave's DMA pieces), first bytes of the K-step; K-strided: k-row l,
    // line w of the 512-byte row
    // One workgroup per shared panel does it: in an XCD's 8 x 4 block of concurrent tiles (block_to_tile) a row panel of A is read by 4
    // workgroups and a column panel of B by 8, all at the same K-step; every one of them prefetching doubled the L2 request count and
    // cost 5-10 % (tools/w4_check.py --scheds=1,4,6,7).  The others pass an out-of-range offset (bit 31: adding the instruction offset cannot wrap):
    // the instruction issues, nothing moves.
    const int tpm = m0 >> 8, tpn = n0 / nstep;
    const bool pf_a = (tpn & 3) == (tpm & 3), pf_b = (tpm & 7) == (tpn & 7);
    const unsigned pfa = !pf_a ? 0x80000000u : A_KC ? (unsigned)(((w + 4 * (lane_in >> 3)) * 8 + (lane_in & 7)) * g.lda) * 2u : (unsigned)(lane_in * g.lda) * 2u + (unsigned)w * 128u;
    const unsigned pfb = !pf_b ? 0x80000000u : B_KC ? (unsigned)(gather_row((w + 4 * (lane_in >> 3)) * 8 + (lane_in & 7), swi) * g.ldb) * 2u : (unsigned)(lane_in * g.ldb) * 2u + (unsigned)w * 128u;
    const unsigned a0 = ra.w0, a1 = ra.w1, a2 = ra.w2, b0 = rb.w0, b1 = rb.w1, b2 = rb.w2;
    const unsigned rda0 = rda_lo + st, rda1 = A_KC ? rda_hi + st : rda_hi, rdb0 = rdb_lo + st;
    const unsigned dst = w4_sgpr(st + (unsigned)w * 1024u);
    if constexpr (B_KC) {
      const unsigned rdb1 = rdb_hi + st;
      MM_W4_RUN_NT(SCHED, voffa0, rda0, rda1, voffb0, rdb0, rdb1, a0, a1, a2, b0, b1, b2, na0, na1, na2, nb0, nb1, nb2, ta, tb0, tb1, tb2, nk_s, dst, wv_s, pfa, pfb, nkm1_s, vrag, vrag)
#ifdef MM_W4_DIAG
      MM_W4_RUN_NT_DIAG(SCHED, voffa0, rda0, rda1, voffb0, rdb0, rdb1, a0, a1, a2, b0, b1, b2, na0, na1, na2, nb0, nb1, nb2, ta, tb0, tb1, tb2, nk_s, dst, wv_s, pfa, pfb, nkm1_s, vrag, vrag)
      if constexpr (SCHED == 121) {
        unsigned o0, o1, o2, o3, o4;
        asm volatile(MM_W4_ASM_NT_S121
                     : [o0] "=&s"(o0), [o1] "=&s"(o1), [o2] "=&s"(o2), [o3] "=&s"(o3), [o4] "=&s"(o4)
                     : MM_W4_INPUTS_NT(voffa0, rda0, rda1, voffb0, rdb0, rdb1, a0, a1, a2, b0, b1, b2, na0, na1, na2, nb0, nb1, nb2, ta, tb0, tb1, tb2, nk_s, dst, wv_s, pfa, pfb, nkm1_s, vrag, vrag)
                     : MM_W4_CLOBBERS_DIAG);
        if (l == 0 && blockIdx.x < 256) {
          unsigned* d = g_w4_diag + (blockIdx.x * 4 + w) * 4;
          d[0] += o0; d[1] += o1; d[2] += o2; d[3] += 1;
          unsigned* e2 = g_w4_diag + 256 * 4 * 4 + (blockIdx.x * 4 + w) * 2;       // gap between two tiles' loops (epilogue + set-up)
          if (sidx != 0) e2[0] += o3 - e2[1];
          e2[1] = o4;
        }
        w4_diag_end = o4;
      }
#endif
    } else if constexpr (A_KC) {
      const unsigned rdb1 = rdb_hi;
      MM_W4_RUN_NN(SCHED, voffa0, rda0, rda1, voffb0, voffb1, rdb0, rdb1, a0, a1, a2, b0, b1, b2, na0, na1, na2, nb0, nb1, nb2, ta, tb0, nk_s, dst, wv_s, pfa, pfb, nkm1_s, vrag)
#ifdef MM_W4_DIAG
      MM_W4_RUN_NN_DIAG(SCHED, voffa0, rda0, rda1, voffb0, voffb1, rdb0, rdb1, a0, a1, a2, b0, b1, b2, na0, na1, na2, nb0, nb1, nb2, ta, tb0, nk_s, dst, wv_s, pfa, pfb, nkm1_s, vrag)
#endif
    } else {
      const unsigned rdb1 = rdb_hi;
      MM_W4_RUN_TN(SCHED, voffa0, voffa1, rda0, rda1, voffb0, voffb1, rdb0, rdb1, a0, a1, a2, b0, b1, b2, na0, na1, na2, nb0, nb1, nb2, ta, tb0, nk_s, dst, wv_s, pfa, pfb)
#ifdef MM_W4_DIAG
      MM_W4_RUN_TN_DIAG(SCHED, voffa0, voffa1, rda0, rda1, voffb0, voffb1, rdb0, rdb1, a0, a1, a2, b0, b1, b2, na0, na1, na2, nb0, nb1, nb2, ta, tb0, nk_s, dst, wv_s, pfa, pfb)
#endif
    }
    sidx += (unsigned)nk;
    // epilogue: the accumulators leave a[0:255] in two halves of 64 columns = one wave tile of the 8-wave form each
    const int mw = m0 + wm * 128;
    // Everything the epilogue derives from the lane index (row / column offsets, bounds, bias values) would be hoisted out of the
    // tile loop and carried THROUGH the asm statement, where only v0-v91 exist: it came back from scratch, and every scratch reload's
    // s_waitcnt vmcnt(0) also waited out the stores issued before it -- ~21 000 cycles per tile (tools/w4_stamps.py).  An opaque copy
    // of the lane index made here keeps those values on this side of the K loop.
    int lane = l;
    asm volatile("" : "+v"(lane));
    const int emode = (g.epi & MM_EPI_RESIDUAL) ? ((g.epi & MM_EPI_ACCUMULATE) ? 3 : 1) : ((g.epi & MM_EPI_ACCUMULATE) ? 2 : 0);   // wave-uniform
    if (EK == 3 && g.rowmajor) {
      w4_epilogue_swiglu_fwd_rowmajor(g, smem + 131072 + w * 8192, mw, n0, wn, lane);
    } else if (EK == 2 && g.rowmajor) {
      w4_epilogue_swiglu_bwd_rowmajor(g, smem + 131072 + w * 8192, mw, n0 + wn * 128, lane);
    } else if (EK == 0 && g.rowmajor && emode != 3) {
      char* xp = smem + 131072 + w * 8192;
#ifdef MM_W4_DIAG
      if (g.diag_epi & 8) {} else
#endif
#ifdef MM_W4_DIAG
      if (emode == 0 && g.shuffle && !(g.epi & MM_EPI_BIAS)) {
        unsigned st[5] = {0, 0, 0, 0, 0};
        w4_epilogue_shuffle(g, mw, n0 + wn * 128, lane, st);
        if (l == 0 && blockIdx.x < 256)
          for (int q = 0; q < 5; ++q) g_w4_diag2[(blockIdx.x * 4 + w) * 8 + q] += st[q] - w4_diag_end;
      } else
#endif
      if (emode == 0 && g.shuffle && !(g.epi & MM_EPI_BIAS)) w4_epilogue_shuffle(g, mw, n0 + wn * 128, lane);
      else if (emode == 0) w4_epilogue_rowmajor<0>(g, xp, mw, n0 + wn * 128, lane);
      else if (emode == 1) w4_epilogue_rowmajor<1>(g, xp, mw, n0 + wn * 128, lane);
      else w4_epilogue_rowmajor<2>(g, xp, mw, n0 + wn * 128, lane);
    } else {
      auto chunk = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        f32x4 acc[8][4];
        if constexpr (c == 0) w4_read_acc_c0(acc);
        else w4_read_acc_c1(acc);
        const int vw = 2 * wn + c;                  // the 8-wave form's wave column
        if constexpr (EK == 4) gemm_epilogue_rope<8, 4>(g, acc, mw, n0, vw, lane);
        else if constexpr (EK == 3) gemm_epilogue_swiglu<8, 4>(g, acc, mw, n0 + vw * 32, lane);
        else if constexpr (EK == 2) gemm_epilogue_ek2<8, 4>(g, acc, mw, n0 + vw * 64, lane);
        else if constexpr (EK == 0) {
          if (!g.stream_epi || emode == 3 || (g.N & 3)) gemm_epilogue_ek0<8, 4>(g, acc, mw, n0 + vw * 64, lane);      // (ragged N: its scalar tail)
          else if (emode == 0) gemm_epilogue_plain_stream<8, 4, 0>(g, acc, mw, n0 + vw * 64, lane);
          else if (emode == 1) gemm_epilogue_plain_stream<8, 4, 1>(g, acc, mw, n0 + vw * 64, lane);
          else gemm_epilogue_plain_stream<8, 4, 2>(g, acc, mw, n0 + vw * 64, lane);
        } else gemm_epilogue_plain<8, 4, false, true>(g, acc, mw, n0 + vw * 64, lane);
      };
      chunk(std::integral_constant<int, 0>{});
      chunk(std::integral_constant<int, 1>{});
    }
#ifdef MM_W4_DIAG
    if constexpr (SCHED == 121) {        // cycles from the loop's end to the last epilogue instruction issued (the rest of the gap = waiting)
      unsigned long long tnow;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow)::"memory");
      if (l == 0 && blockIdx.x < 256) g_w4_diag[256 * 4 * 4 + 256 * 4 * 2 + blockIdx.x * 4 + w] += (unsigned)tnow - w4_diag_end;
    }
#endif
    tile = next;
    m0 = nm0;
    n0 = nn0;
    ra.w0 = na0; ra.w1 = na1; ra.w2 = na2;          // the next tile's descriptors were built for the prefetch: reuse them (a wave alone
    rb.w0 = nb0; rb.w1 = nb1; rb.w2 = nb2;          // on its SIMD hides none of the ~1 500 cycles of scalar arithmetic they take)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the (empty) DMAs issued for a tile that does not exist
  // ---- the half-tile round (wave quantisation, as in the 8-wave kernel): when the last round of the persistent grid is at most half
  // full its tiles are cut into two 256 x 128 halves handed to workgroups 0 .. 2 tail - 1.  Same A tile, a B tile of 128 rows / columns
  // (a 16 KB image, 4 DMA pieces per wave), a wave owns 128 x 64: the MM_W4_ASM_*_H loop; same products in the same K order.
  if constexpr (EK == 0) {
    if (g.tail > 0 && (int)blockIdx.x < 2 * g.tail) {
      int hm, hn;
      block_to_tile(total + ((int)blockIdx.x >> 1), g.nbm, g.nbn, hm, hn, g.group_m);
      const int hm0 = hm * 256, hn0 = hn * 256 + ((int)blockIdx.x & 1) * 128;
      const SRsrc ha = tile_rsrc<A_KC>(A, g.lda, hm0, g.M, g.K);
      const SRsrc hb = tile_rsrc<B_KC>(B, g.ldb, hn0, g.N, g.K);
      int lane_in = l;                                // (opaque: nothing of this section lives through the main loop's asm statements)
      asm volatile("" : "+v"(lane_in));
      unsigned toa[8], tob[4];
      dma_offsets<A_KC, 256, 4>(toa, g.lda, 0, lane_in);
      dma_offsets<B_KC, 128, 4>(tob, g.ldb, 0, lane_in);
      __builtin_amdgcn_s_barrier();                   // every wave has left the ring of the last full tile
      dma_tile_inv<256, 4>(lds0, ha, toa, 0u);
      dma_tile_inv<128, 4>(lds0 + 32768u, hb, tob, 0u);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      dma_tile_inv<256, 4>(lds0 + 65536u, ha, toa, A_KC ? 128u : 64u * lda2);
      dma_tile_inv<128, 4>(lds0 + 65536u + 32768u, hb, tob, B_KC ? 128u : 64u * ldb2);
      unsigned rda_lo, rda_hi, rdb_lo, rdb_hi;
      frag_addr(lane_in, rda_lo, rda_hi, rdb_lo, rdb_hi);
      const int lr = lane_in & 15, lg = lane_in >> 4;
      if constexpr (B_KC) {                           // the wave's 64 rows of the 128-row B image
        rdb_lo = 32768u + (unsigned)((wn * 64 + lr) * 128 + (((0 + lg) ^ kc_swz(lr)) * 16));
        rdb_hi = 32768u + (unsigned)((wn * 64 + lr) * 128 + (((4 + lg) ^ kc_swz(lr)) * 16));
      } else {                                        // [k][128 x] image: 256 bytes per k-row, 8 column slots of 32 bytes
        const int q = lr >> 2, p = lr & 3, k = 8 * lg + q;
        rdb_lo = 32768u + (unsigned)(k * 256 + p * 8);
        rdb_hi = (unsigned)(((wn * 4 + ks_swz(k)) & 7) * 32);
      }
      const unsigned voffa0 = toa[0], voffa1 = toa[1], voffb0 = tob[0];
      const unsigned vrag = rag_mask(lane_in);
      const unsigned pfa = 0u, pfb = 0u;            // (no L2 prefetch in the half-tile loop)
      const unsigned a0 = ha.w0, a1 = ha.w1, a2 = ha.w2, b0 = hb.w0, b1 = hb.w1, b2 = hb.w2;
      const unsigned z = w4_sgpr(0u);
      const unsigned rda0 = rda_lo, rda1 = rda_hi, rdb0 = rdb_lo, rdb1 = rdb_hi;      // stage 0
      const unsigned dst = w4_sgpr((unsigned)w * 1024u);
      const unsigned tbh0 = B_KC ? w4_sgpr(32u * ldb2) : w4_sgpr(16u * ldb2), tbh1 = w4_sgpr(64u * ldb2);
      if constexpr (B_KC)
        asm volatile(MM_W4_ASM_NT_H : : MM_W4_INPUTS_NT_H(voffa0, rda0, rda1, voffb0, rdb0, rdb1, a0, a1, a2, b0, b1, b2, z, z, z, z, z, z, ta, tbh0, tbh1, tbh1, nk_s, dst, wv_s, pfa, pfb, nkm1_s, vrag, vrag) : MM_W4_CLOBBERS);
      else if constexpr (A_KC)
        asm volatile(MM_W4_ASM_NN_H : : MM_W4_INPUTS_NN_H(voffa0, rda0, rda1, voffb0, rdb0, rdb1, a0, a1, a2, b0, b1, b2, z, z, z, z, z, z, ta, tbh0, nk_s, dst, wv_s, pfa, pfb, nkm1_s, vrag) : MM_W4_CLOBBERS);
      else
        asm volatile(MM_W4_ASM_TN_H : : MM_W4_INPUTS_TN_H(voffa0, voffa1, rda0, rda1, voffb0, rdb0, rdb1, a0, a1, a2, b0, b1, b2, z, z, z, z, z, z, ta, tbh0, nk_s, dst, wv_s, pfa, pfb) : MM_W4_CLOBBERS);
      int lane = l;
      asm volatile("" : "+v"(lane));
      const int mw = hm0 + wm * 128, nw = hn0 + wn * 64;
      const int emode = (g.epi & MM_EPI_RESIDUAL) ? ((g.epi & MM_EPI_ACCUMULATE) ? 3 : 1) : ((g.epi & MM_EPI_ACCUMULATE) ? 2 : 0);
      if (g.rowmajor && emode != 3) {
        char* xp = smem + 131072 + w * 8192;
        const int mwu = __builtin_amdgcn_readfirstlane(mw), nwu = __builtin_amdgcn_readfirstlane(nw);
        const bool has_bias = (g.epi & MM_EPI_BIAS) != 0;
        const int rows = g.M - mwu;
        auto rc = make_rsrc((const bf16*)g.C + (int64_t)mwu * g.ldc, (int64_t)rows * g.ldc * 2);
        auto rr = emode == 1 ? make_rsrc((const bf16*)g.residual + (int64_t)mwu * g.ldr, (int64_t)rows * g.ldr * 2) : rc;
        if (emode == 0 && g.shuffle && !has_bias) w4_shuffle_half<0>(g, rc, nwu, lane);
        else if (emode == 0) w4_rm_half<0, 0>(g, xp, rc, rr, nwu, has_bias, lane);
        else if (emode == 1) w4_rm_half<0, 1>(g, xp, rc, rr, nwu, has_bias, lane);
        else w4_rm_half<0, 2>(g, xp, rc, rr, nwu, has_bias, lane);
      } else {
        f32x4 acc[8][4];
        w4_read_acc_c0(acc);
        gemm_epilogue_ek0<8, 4>(g, acc, mw, nw, lane);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Skinny NT GEMM for KV-cache decode (reference model.py:595-602: one new token per sequence, M = batch <= 16).
// C[M,N] = A[M,K] . W[N,K]^T is a stream over W (16 GB of bf16 weights per token for the 8B decoder): HBM-bound, so the
// kernel is laid out for bytes in flight, not for MFMA occupancy.  One workgroup = 16 rows of W, its 8 waves split K;
// a wave's lane reads 16 contiguous bytes of its W row straight into the MFMA operand layout (no LDS), 8 K-steps
// (8 KiB per wave, 64 KiB per CU) in flight; x^T is the other operand, read through L2 (it is M*K*2 bytes, shared by
// every workgroup); rows of x beyond M and rows of W beyond N are zeros from the buffer range check.  The 8 partial
// 16x16 tiles are summed through LDS in a fixed order (deterministic) and wave 0 applies the epilogue.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void gemm_skinny_kernel(GemmArgs g) {
  __shared__ float red[8][4][64];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 16;
  const bf16* A = (const bf16*)g.A;
  const bf16* B = (const bf16*)g.B;
  const int nrows = min(16, g.N - n0);
  auto rw = make_rsrc(B + (int64_t)n0 * g.ldb, (int64_t)nrows * g.ldb * 2);
  auto rx = make_rsrc(A, (int64_t)g.M * g.lda * 2);
  const int row = l & 15, kc = l >> 4;
  const int nks = (g.K + 31) / 32;
  const int per = (nks + 7) / 8;
  const int ks0 = w * per, ks1 = min(nks, ks0 + per);
  const unsigned wrow = (unsigned)(row * g.ldb) * 2u, xrow = (unsigned)(row * g.lda) * 2u;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  constexpr int U = 8;
  for (int ks = ks0; ks < ks1; ks += U) {
    u32x4 fw[U], fx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = (ks + u) * 32 + kc * 8;
      const bool ok = (ks + u) < ks1 && k < g.K;
      // default cache policy on W: the non-temporal one (aux = 2) measured 5.4 vs 5.1 ms/token on the 8B decoder
      fw[u] = __builtin_amdgcn_raw_buffer_load_b128(rw, ok ? wrow + (unsigned)k * 2u : 0xFFFFFFFFu, 0, 0);
      fx[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? xrow + (unsigned)k * 2u : 0xFFFFFFFFu, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw[u]), __builtin_bit_cast(bf16x8, fx[u]), acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[w][r][l] = acc[r];
  __syncthreads();
  if (w != 0) return;
  // lane: m = l & 15, n = n0 + 4*(l>>4) + r
  const int m = l & 15, n = n0 + 4 * (l >> 4);
  if (m >= g.M || n >= g.N) return;
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float t = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) t += red[ww][r][l];
    v[r] = t;
  }
  const int epi = g.epi;
  const bf16* bias = (const bf16*)g.bias;
  const bf16* R = (const bf16*)g.residual;
  bf16* cp = (bf16*)g.C + (int64_t)m * g.ldc + n;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (n + r >= g.N) break;
    float t = v[r];
    if (epi & MM_EPI_BIAS) t += (float)bias[n + r];
    if (epi & MM_EPI_GELU_ERF) t = act_gelu_erf(t);
    else if (epi & MM_EPI_QUICK_GELU) t = act_quick_gelu(t);
    else if (epi & MM_EPI_GELU_TANH) t = act_gelu_tanh(t);
    if (epi & MM_EPI_RESIDUAL) t += (float)R[(int64_t)m * g.ldr + n + r];
    if (epi & MM_EPI_ACCUMULATE) t += (float)cp[r];
    cp[r] = (bf16)t;
  }
}


// ------------------------------------------------------------------------------------------------------
// Decode-step fusions on the weight-streaming kernel (round 3).  A decode step of the 8B decoder was ~320 launches of which
// ~190 were tiny (RMSNorm of 4 rows, SwiGLU, RoPE + cache append, the split merge): 4.5-6 us each on the timeline for a few
// KB of work.  They become prologues / epilogues of the GEMM that consumes / produces their data (same arithmetic, same
// rounding points):
//   NORM    the RMSNorm in FRONT of a projection, applied while the workgroup stages x into LDS
//   MODE 1  gate|up + SwiGLU     a workgroup streams 8 gate rows and the 8 matching up rows of the fused [2I, K] weight; after the
//                                K-split sums meet in LDS a lane has g and u of a feature: act = bf16(bf16(silu(g)) * u)
//   MODE 2  q|k|v + RoPE + KV-cache append   8 rows d .. d+7 and their rotate_half partners d+64 .. of one 128-wide head; the
//                                epilogue rotates q / k heads (rope_lo / rope_hi), writes q|k|v and appends k, v to the cache row
// ------------------------------------------------------------------------------------------------------
struct SkinnyArgs {
  int M, N, K;
  const bf16* A; int lda;
  const bf16* B; int ldb;
  bf16* C; int ldc;
  const bf16* bias;
  const bf16* residual; int ldr;
  int I;                                        // MODE 1: intermediate size (B = [2I, K], C = act [M, I])
  int Hq, Hkv;                                  // MODE 2
  const float* cos_t; const float* sin_t;       // [M, 64]
  bf16* kdst; bf16* vdst; int64_t dstride;      // cache row of this step for sequence 0; elements between sequences
  const bf16* norm_w; float eps;                // NORM: weight / eps of the RMSNorm in front of the projection (nullptr: none)
  int epi;                                      // gemv_stream_kernel MODE 0: MM_EPI_* activation / accumulate flags
};

// ------------------------------------------------------------------------------------------------------
// gemv_stream_kernel: the weight-streaming kernel of the decode step, second form (round 3).  Same work split as
// gemm_skinny_kernel (one workgroup = 16 rows of W, its 8 waves split K, the 8 partial tiles summed through LDS in wave order:
// the same bits), but laid out for BYTES IN FLIGHT:
//   * a wave keeps a ring of U = 16 W loads (16 KiB) in flight and refills a slot right after the MFMA that consumed it; the
//     older form issued 8, waited for all 8, multiplied, issued 8 more: an empty memory queue once per batch (K = 4096 is two
//     batches per wave, so half of every such launch was spent with nothing requested);
//   * x (M <= 16 rows, the MFMA's other operand) is staged ONCE per workgroup into LDS instead of being re-read from L2 at every
//     K-step into registers: the vector-memory queue and the registers belong to W alone;
//   * NORM: the RMSNorm in front of the projection (HF:llama LlamaRMSNorm) is applied while x is staged -- every workgroup
//     normalises the M rows itself (M x K elements: nothing beside 16 x K of weights), with rmsnorm_fwd_kernel's own arithmetic
//     and summation order, so the bits are those of the separate launch.  A decode layer is then 6 launches with NO norm
//     launches and no cross-workgroup hand-off (the round's first form ran the NEXT norm in the GEMM's last-arriving workgroup:
//     a device-scope ticket plus a dependent round trip, ~6 us per launch, twice per layer).
// MODE 0 plain epilogue (bias, activation, residual, accumulate; any N), 1 gate|up + SwiGLU, 2 q|k|v + RoPE + KV-cache append.
// NT: non-temporal policy on the W loads (read once).
// ------------------------------------------------------------------------------------------------------
template <int CH, int R>
__device__ __forceinline__ void rmsnorm_rows_to_lds(const bf16* x, int ldx, const bf16* w, int H, float eps, char* xs, int nrows, float* red) {
  // rmsnorm_fwd_kernel's arithmetic (mm_rowwise.hip) for rows 0 .. nrows-1, R rows' loads in flight at a time; threads >= 256 idle
  const int t = threadIdx.x;
  for (int r0 = 0; r0 < nrows; r0 += R) {
    bf16x8 xv[R][CH];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      auto rx = make_rsrc(x + (int64_t)(r0 + r) * ldx, (r0 + r) < nrows ? (int64_t)H * 2 : 0);
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int e = (c * 256 + t) * 8;
        xv[r][c] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rx, (t < 256 && e < H) ? (unsigned)e * 2u : 0xFFFFFFFFu, 0, 0));
      }
    }
    // the R rows' sums of squares meet in ONE pair of barriers (block_sum_256's order per row: lanes of a wave, then waves 0..3)
    float ssr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float ss = 0.f;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int e = (c * 256 + t) * 8;
        if (t < 256 && e < H && r0 + r < nrows) {
#pragma unroll
          for (int i = 0; i < 8; ++i) { const float f = (float)xv[r][c][i]; ss += f * f; }
        }
      }
      ssr[r] = wave_sum(ss);
    }
    __syncthreads();
    if ((t & 63) == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) red[(t >> 6) * 4 + r] = ssr[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (r0 + r < nrows) {                                // uniform
        const float ss = red[0 * 4 + r] + red[1 * 4 + r] + red[2 * 4 + r] + red[3 * 4 + r];
        const float rs = rsqrtf(ss / (float)H + eps);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int e = (c * 256 + t) * 8;
          if (t < 256 && e < H) {
            const bf16x8 wv = *(const bf16x8*)(w + e);
            bf16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const float nrm = (float)(bf16)((float)xv[r][c][i] * rs);
              o[i] = (bf16)((float)wv[i] * nrm);
            }
            *(bf16x8*)(xs + ((int64_t)(r0 + r) * H + e) * 2) = o;
          }
        }
      }
    }
  }
}

template <int MODE, bool NORM, bool NT>
__global__ __launch_bounds__(512, 4) void gemv_stream_kernel(SkinnyArgs g, int nblocks) {      // 4 waves per SIMD (2 workgroups per CU): <= 128 VGPRs
  extern __shared__ __attribute__((aligned(16))) char xs[];          // x, M rows of K bf16 (normalised when NORM)
  __shared__ float red[2][8][4][64];
  __shared__ float nred[8 * 4];                                       // [wave][row of the norm batch]
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = l & 15, kc = l >> 4;
  // the lane's row of W in row block `blk` (as MFMA A-operand row `row`) as a byte offset, or out of range
  auto wrow_of = [&](int blk) -> unsigned {
    int wr;
    if (blk >= nblocks) return 0xFFFFFFFFu;
    if constexpr (MODE == 1) {
      const int f = blk * 8 + (row & 7);
      wr = f < g.I ? f + (row >> 3) * g.I : -1;
    } else if constexpr (MODE == 2) {
      wr = (blk >> 3) * 128 + (blk & 7) * 8 + (row & 7) + (row >> 3) * 64;
    } else {
      wr = blk * 16 + row;
      if (wr >= g.N) wr = -1;
    }
    return wr < 0 ? 0xFFFFFFFFu : (unsigned)wr * (unsigned)g.ldb * 2u;
  };
  auto rw = make_rsrc(g.B, (int64_t)(MODE == 1 ? 2 * g.I : g.N) * g.ldb * 2);
  constexpr int U = 16;
  constexpr int WPOL = NT ? 2 : 0;
  const int nks = (g.K + 31) / 32;
  const int per = (nks + 7) / 8;                            // K-steps per wave and row block (the wave's share of K) ...
  const int ipb = (per + U - 1) / U;                        // ... walked in ipb rounds of the U-slot ring (steps beyond `per` load nothing)
  const int ks0 = w * per, ks1 = min(nks, ks0 + per);
  u32x4 fw[U];
  // PERSISTENT: the workgroup walks row blocks blockIdx.x, + gridDim.x, ...; x is staged (and normalised) once, the ring runs across
  // block boundaries (the slot a step frees is refilled with the same step of the NEXT block when this block has no further round)
  auto issue = [&](int u, unsigned wrow, int i) {            // i = step inside the block
    const int ks = ks0 + i;
    const int k = ks * 32 + kc * 8;
    const bool ok = i < per && ks < ks1 && k < g.K && wrow != 0xFFFFFFFFu;
    fw[u] = __builtin_amdgcn_raw_buffer_load_b128(rw, ok ? wrow + (unsigned)k * 2u : 0xFFFFFFFFu, 0, WPOL);
  };
  int blk = blockIdx.x;
  unsigned wrow = wrow_of(blk);
  // ---- x -> LDS.  Plain: requested first (L2 hits), stored once the W ring has been requested behind it.  NORM: the rows pass
  // through registers (sum of squares); half of the ring is requested before, the other half after (the norm's own registers and
  // the whole ring together would not leave room for two workgroups per CU).
  const int xchunks = g.M * (g.K / 8);                      // 16-byte chunks of x (K % 8 == 0: host)
  if constexpr (NORM) {
#pragma unroll
    for (int u = 0; u < U / 2; ++u) issue(u, wrow, u);
    if (g.K <= 2048) rmsnorm_rows_to_lds<1, 4>(g.A, g.lda, g.norm_w, g.K, g.eps, xs, g.M, nred);
    else if (g.K <= 4096) rmsnorm_rows_to_lds<2, 4>(g.A, g.lda, g.norm_w, g.K, g.eps, xs, g.M, nred);
    else rmsnorm_rows_to_lds<4, 2>(g.A, g.lda, g.norm_w, g.K, g.eps, xs, g.M, nred);
#pragma unroll
    for (int u = U / 2; u < U; ++u) issue(u, wrow, u);
  } else {
    constexpr int XU = 8;                                   // chunks per thread and pass
    const int cpr = g.K / 8;
    bool first = true;
    for (int base = 0; base < xchunks; base += 512 * XU) {
      u32x4 xv[XU];
#pragma unroll
      for (int j = 0; j < XU; ++j) {
        const int i = base + j * 512 + (int)threadIdx.x;
        const int r = i / cpr, c = i - r * cpr;
        xv[j] = i < xchunks ? *(const u32x4*)(g.A + (int64_t)r * g.lda + c * 8) : u32x4{0u, 0u, 0u, 0u};
      }
      if (first) {
#pragma unroll
        for (int u = 0; u < U; ++u) issue(u, wrow, u);
        first = false;
      }
#pragma unroll
      for (int j = 0; j < XU; ++j) {
        const int i = base + j * 512 + (int)threadIdx.x;
        if (i < xchunks) *(u32x4*)(xs + (int64_t)i * 16) = xv[j];
      }
    }
  }
  __syncthreads();
  const char* xrow = xs + (int64_t)row * g.K * 2 + kc * 16;
  const bool xlane = row < g.M;
  const int m = l & 15, gq = l >> 4;                        // epilogue (wave 0): m = activation row, local W rows 4 * gq + r
  for (int j = 0; blk < nblocks; ++j, blk += gridDim.x) {
    const unsigned wnext = wrow_of(blk + gridDim.x);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < ipb; ++it) {
      const bool more = it + 1 < ipb;                       // another round of this block, or on to the next block
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = it * U + u, kk = ks0 + i;
        u32x4 fx = {0u, 0u, 0u, 0u};
        if (xlane && i < per && kk < ks1 && kk * 32 + kc * 8 < g.K) fx = *(const u32x4*)(xrow + kk * 64);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fw[u]), __builtin_bit_cast(bf16x8, fx), acc, 0, 0, 0);
        issue(u, more ? wrow : wnext, more ? i + U : u);
      }
    }
    wrow = wnext;
    float (*rb)[4][64] = red[j & 1];
#pragma unroll
    for (int r = 0; r < 4; ++r) rb[w][r][l] = acc[r];
    __syncthreads();                 // red[j & 1] is rewritten two blocks later: wave 0 has passed the barrier in between
    if (w != 0) continue;
    if constexpr (MODE == 0) {
      if (m >= g.M) continue;
      const int n = blk * 16 + 4 * gq;
      if (n >= g.N) continue;
      const int epi = g.epi;
      bf16* cp = g.C + (int64_t)m * g.ldc + n;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (n + r >= g.N) break;
        float t = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) t += rb[ww][r][l];
        if (g.bias) t += (float)g.bias[n + r];
        if (epi & MM_EPI_GELU_ERF) t = act_gelu_erf(t);
        else if (epi & MM_EPI_QUICK_GELU) t = act_quick_gelu(t);
        else if (epi & MM_EPI_GELU_TANH) t = act_gelu_tanh(t);
        if (g.residual) t += (float)g.residual[(int64_t)m * g.ldr + n + r];
        if (epi & MM_EPI_ACCUMULATE) t += (float)cp[r];
        cp[r] = (bf16)t;
      }
    } else {                                                 // MODE 1 / 2: lanes 0-31 own the pairs (own rows, lane + 32's rows)
      float v1[4], v2[4];                                    // every lane sums its own rows (wave order), then takes lane + 32's sums
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) a += rb[ww][r][l];
        v1[r] = a;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v2[r] = __shfl(v1[r], (l + 32) & 63, 64);
      if (gq >= 2 || m >= g.M) continue;
      if constexpr (MODE == 1) {
        const int f = blk * 8 + 4 * gq;
        if (f < g.I) {                                       // I % 4 == 0 (host)
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float gf = (float)(bf16)v1[r], uf = (float)(bf16)v2[r];
            const float sg = (float)(bf16)(gf / (1.0f + __expf(-gf)));
            o[r] = (bf16)(sg * uf);
          }
          *(bf16x4*)(g.C + (int64_t)m * g.ldc + f) = o;
        }
      } else {
        const int h = blk >> 3, d = (blk & 7) * 8 + 4 * gq;
        bf16x4 o1, o2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a = v1[r], b = v2[r];
          if (g.bias) { a += (float)g.bias[h * 128 + d + r]; b += (float)g.bias[h * 128 + d + 64 + r]; }
          o1[r] = (bf16)a;
          o2[r] = (bf16)b;
        }
        if (h < g.Hq + g.Hkv) {
          const f32x4 c4 = *(const f32x4*)(g.cos_t + (int64_t)m * 64 + d), s4 = *(const f32x4*)(g.sin_t + (int64_t)m * 64 + d);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float a = (float)o1[r], b = (float)o2[r];
            o1[r] = (bf16)rope_lo(a, b, c4[r], s4[r]);
            o2[r] = (bf16)rope_hi(a, b, c4[r], s4[r]);
          }
        }
        bf16* cp = g.C + (int64_t)m * g.ldc + h * 128 + d;
        *(bf16x4*)cp = o1;
        *(bf16x4*)(cp + 64) = o2;
        if (h >= g.Hq) {
          bf16* dp = (h < g.Hq + g.Hkv ? g.kdst + (h - g.Hq) * 128 : g.vdst + (h - g.Hq - g.Hkv) * 128) + (int64_t)m * g.dstride + d;
          *(bf16x4*)dp = o1;
          *(bf16x4*)(dp + 64) = o2;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// exact-fp32 GEMM (parity path).  64x64 tile, BK = 16, 4 waves each 32x32 (2x2 of 16x16x4 f32 MFMA).
// LDS images are [k][x] for both operands (any global layout is re-tiled by scalar loads).
// ------------------------------------------------------------------------------------------------------
template <int LAYOUT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  constexpr int TM = 64, TN = 64, TK = 16, LD = 80;  // LD = 80: rows k, k+1 land on disjoint bank halves
  __shared__ float As[TK * LD];
  __shared__ float Bs[TK * LD];
  const int pm = blockIdx.x % g.nbm, pn = blockIdx.x / g.nbm;
  const int m0 = pm * TM, n0 = pn * TN;
  const float* A = (const float*)g.A;
  const float* B = (const float*)g.B;
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = w >> 1, wn = w & 1;
  f32x4 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < g.K; k0 += TK) {
    // each thread loads 4 elements of A-tile and 4 of B-tile
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = threadIdx.x + e * 256;  // 0..1023 over [TK][64]
      int k, x;
      // A: NT/NN are [M][K] (k fastest), TN is [K][M] (m fastest)
      if (LAYOUT == MM_GEMM_TN) { x = idx & 63; k = idx >> 6; } else { k = idx & 15; x = idx >> 4; }
      float va = 0.f;
      if (m0 + x < g.M && k0 + k < g.K)
        va = (LAYOUT == MM_GEMM_TN) ? A[(int64_t)(k0 + k) * g.lda + m0 + x] : A[(int64_t)(m0 + x) * g.lda + k0 + k];
      As[k * LD + x] = va;
      // B: NT is [N][K] (k fastest), NN/TN are [K][N] (n fastest)
      if (LAYOUT == MM_GEMM_NT) { k = idx & 15; x = idx >> 4; } else { x = idx & 63; k = idx >> 6; }
      float vb = 0.f;
      if (n0 + x < g.N && k0 + k < g.K)
        vb = (LAYOUT == MM_GEMM_NT) ? B[(int64_t)(n0 + x) * g.ldb + k0 + k] : B[(int64_t)(k0 + k) * g.ldb + n0 + x];
      Bs[k * LD + x] = vb;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; kk += 4) {
      float fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = As[(kk + (l >> 4)) * LD + wm * 32 + i * 16 + (l & 15)];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = Bs[(kk + (l >> 4)) * LD + wn * 32 + j * 16 + (l & 15)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j], fa[i], acc[i][j], 0, 0, 0);  // swapped: D^T
    }
    __syncthreads();
  }
  float* C = (float*)g.C;
  const float* bias = (const float*)g.bias;
  const float* R = (const float*)g.residual;
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 32 + i * 16 + (l & 15);
    if (m >= g.M) continue;
    for (int j = 0; j < 2; ++j) {
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 32 + j * 16 + 4 * (l >> 4) + r;
        if (n >= g.N) continue;
        float v = acc[i][j][r];
        if (g.epi & MM_EPI_BIAS) v += bias[n];
        if (g.epi & MM_EPI_GELU_ERF) v = act_gelu_erf(v);
        else if (g.epi & MM_EPI_QUICK_GELU) v = act_quick_gelu(v);
        else if (g.epi & MM_EPI_GELU_TANH) v = act_gelu_tanh(v);
        if (g.epi & MM_EPI_RESIDUAL) v += R[(int64_t)m * g.ldr + n];
        float* cp = C + (int64_t)m * g.ldc + n;
        if (g.epi & MM_EPI_ACCUMULATE) v += *cp;
        *cp = v;
      }
    }
  }
}

// column sums (bias gradients): block = 256 threads = 32 column groups (16 bytes each) x 8 row lanes; the grid also
// splits the rows (gridDim.y) and finishes with float atomics into an f32 scratch only when needed -- here rows are
// few enough (<= 8192) that one block per 32 column groups sweeping all rows with 8 row lanes is HBM-efficient.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* X, int M, int N, int ldx, T* out, int accumulate) {
  // one block = 8 column chunks of 16 B (one 128-B line per row) x 32 row lanes: the bias-gradient inputs are short
  // (ViT: 1028 rows) and narrow, so parallelism comes from many small blocks and 4 independent loads per lane in flight
  constexpr int VN = Vec16<T>::N, CG = 8, RL = 32;
  __shared__ float red[RL][CG * VN + 1];
  const int cg = threadIdx.x & (CG - 1), rl = threadIdx.x / CG;
  const int n0 = (blockIdx.x * CG + cg) * VN;
  float acc[VN];
#pragma unroll
  for (int k = 0; k < VN; ++k) acc[k] = 0.f;
  if (n0 + VN <= N) {
    int m = rl;
    for (; m + 3 * RL < M; m += 4 * RL) {
      Vec16<T> v0 = *(const Vec16<T>*)(X + (int64_t)m * ldx + n0);
      Vec16<T> v1 = *(const Vec16<T>*)(X + (int64_t)(m + RL) * ldx + n0);
      Vec16<T> v2 = *(const Vec16<T>*)(X + (int64_t)(m + 2 * RL) * ldx + n0);
      Vec16<T> v3 = *(const Vec16<T>*)(X + (int64_t)(m + 3 * RL) * ldx + n0);
#pragma unroll
      for (int k = 0; k < VN; ++k) acc[k] += (v0.get(k) + v1.get(k)) + (v2.get(k) + v3.get(k));
    }
    for (; m < M; m += RL) {
      Vec16<T> v = *(const Vec16<T>*)(X + (int64_t)m * ldx + n0);
#pragma unroll
      for (int k = 0; k < VN; ++k) acc[k] += v.get(k);
    }
  } else if (n0 < N) {
    for (int m = rl; m < M; m += RL)
      for (int k = 0; k < VN && n0 + k < N; ++k) acc[k] += to_f32(X[(int64_t)m * ldx + n0 + k]);
  }
#pragma unroll
  for (int k = 0; k < VN; ++k) red[rl][cg * VN + k] = acc[k];
  __syncthreads();
  for (int c = threadIdx.x; c < CG * VN; c += 256) {
    const int n = blockIdx.x * CG * VN + c;
    if (n < N) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < RL; ++r) t += red[r][c];
      if (accumulate) t += to_f32(out[n]);
      out[n] = from_f32<T>(t);
    }
  }
}

}  // namespace

static int g_opt_persist = 1;   // walk tiles with resident workgroups
static int g_opt_kernel = 0;    // 0 auto, 1 v1 (128x128 register staged), 2..6 LDS-DMA tiles 256x128, 256x256, 128x128, 64x128, 64x64
static int g_opt_tail = 1;      // cut the tiles of a less-than-half-full last round into 256x128 halves (persistent 256x256 grid)
static int g_opt_skinny = 1;    // M <= 16 NT problems (decode) on the weight-streaming kernel
static int g_opt_gemv_stream = 1;   // ... in its ring-buffered form (gemv_stream_kernel); 0 = gemm_skinny_kernel (A/B)
static int g_opt_gemv_nt = 0;       // non-temporal policy on the decode kernels' weight loads (A/B: tools/gemv_bench.py)
static int g_opt_gemv_wgs = 0;      // persistent workgroups per CU of gemv_stream_kernel (0 = what fits: 2, or 1 with a large x)
static int g_opt_small = -1;    // experiments: force the DMA variant for problems that do not fill the chip (-1 = heuristic)
// problems too small for the 256-wide tiles (ViT-L/14 on 4 images = 1028 rows, projector): these are latency-bound, so
// the tile is chosen by how many workgroups it yields (tools/gemm_bench.py --small: 64x128 wins up to ~96 tiles of
// 128x128, the 128x128 LDS-DMA kernel up to ~320, beyond that the register-staged 128x128 kernel at 2 workgroups/CU)
static int small_variant(int M, int N, int K) {
  (void)K;
  if (g_opt_small >= 0) return g_opt_small;
  const int64_t t = (int64_t)((M + 127) / 128) * ((N + 127) / 128);
  return t <= 96 ? 4 : (t <= 320 ? 3 : 0);
}
static int g_opt_epi_pipe = 1;      // pipelined, branch-free epilogue of the plain / SwiGLU-backward LDS-DMA kernels
static int g_opt_issue_waves = 4;   // waves that issue the 256x256 kernel's DMA (4 staggers the two waves of each SIMD)
static int g_opt_w4_big = 4;        // 4 = the split-barrier schedule 4 for operands that stream from HBM (N or K >= 14336), 1 = schedule 1 everywhere.  Step A/B, three series on three boxes: 354.2 / 348.6 / 353.9 vs 355.6 / 352.2 / 355.4 ms (medians)
                                    // SwiGLU GEMMs alone, 348.6 vs 352.2 ms at step level, tools/step_ab.py: off)
static int g_opt_w4_group_m = 8;    // experiment: GROUP_M of the 4-wave kernel's tile order
static int g_opt_w4_stream = 1;     // 4-wave kernel: wait-free plain epilogue (0 = gemm_epilogue_plain_pipe, A/B)
static int g_opt_w4_shuffle = 0;    // GemmArgs::shuffle (measured equal to the LDS form within +-0.5 %: DESIGN.md section 4, the round-4 list, item 8)
static int g_opt_w4_diag_epi = 0;   // MM_W4_DIAG builds: GemmArgs::diag_epi
static int g_last_kernel = -1;      // which kernel the last bf16 mm_gemm* call launched: 10 = the 4-wave 256x256 kernel, 0..5 = v1 / the 8-wave DMA tiles, 20 = skinny
static int g_opt_w4_stagger_slots = 4;
static int g_opt_w4_stagger = 0;    // experiment: see GemmArgs::stagger
static int g_opt_w4_rowmajor = 1;   // 4-wave kernel: row-major (LDS-transposed, 16-byte) plain epilogue; 0 = the accumulator-layout epilogue (A/B)
static int g_opt_w4 = [] { const char* e = getenv("MM_GEMM_W4"); return e ? atoi(e) : 1; }();            // (MM_GEMM_W4=0: A/B at step level) NT / NN 256x256 tiles on the 4-wave hand-scheduled kernel (gemm_bf16_w4_kernel); 0 = the 8-wave kernel (A/B)

extern "C" int mm_attn_set_issue_waves(int v);
int mm_attn_option(const char* name, int value);

extern int g_adamw_blocks;    // mm_optim.hip

#ifdef MM_W4_DIAG
extern "C" int mm_w4_diag2_read(unsigned* out, int reset) {     // 256 x 4 x 8 words
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_w4_diag2), sizeof(unsigned) * 256 * 4 * 8) != hipSuccess) return MM_ERR_LAUNCH;
  if (reset) {
    static unsigned zeros[256 * 4 * 8];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_w4_diag2), zeros, sizeof(zeros)) != hipSuccess) return MM_ERR_LAUNCH;
  }
  return MM_OK;
}
extern "C" int mm_w4_diag_read(unsigned* out, int reset) {      // diag builds only (not in the ABI header): 256 x 4 x 4 words
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_w4_diag), sizeof(unsigned) * 256 * 4 * 7) != hipSuccess) return MM_ERR_LAUNCH;
  if (reset) {
    static unsigned zeros[256 * 4 * 7];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_w4_diag), zeros, sizeof(zeros)) != hipSuccess) return MM_ERR_LAUNCH;
  }
  return MM_OK;
}
#endif

extern "C" int mm_set_option(const char* name, int value) {
  if (!name) return MM_ERR_ARG;
  if (!strcmp(name, "attn_issue_waves")) return mm_attn_set_issue_waves(value);
  if (!strncmp(name, "attn_", 5)) return mm_attn_option(name, value);
  if (!strcmp(name, "adamw_blocks")) { if (value < 0) return MM_ERR_ARG; g_adamw_blocks = value; return MM_OK; }
  if (!strcmp(name, "gemm_tail")) { g_opt_tail = value != 0; return MM_OK; }
  if (!strcmp(name, "gemm_skinny")) { g_opt_skinny = value != 0; return MM_OK; }
  if (!strcmp(name, "gemv_stream")) { g_opt_gemv_stream = value != 0; return MM_OK; }
  if (!strcmp(name, "gemv_nt")) { g_opt_gemv_nt = value != 0; return MM_OK; }
  if (!strcmp(name, "gemv_wgs")) { g_opt_gemv_wgs = value; return MM_OK; }
  if (!strcmp(name, "gemm_small")) { if (value < -1 || value > 5) return MM_ERR_ARG; g_opt_small = value; return MM_OK; }
  if (!strcmp(name, "gemm_persist")) { g_opt_persist = value != 0; return MM_OK; }
  if (!strcmp(name, "gemm_epi_pipe")) { g_opt_epi_pipe = value != 0; return MM_OK; }
  if (!strcmp(name, "gemm_issue_waves")) { if (value != 4 && value != 8) return MM_ERR_ARG; g_opt_issue_waves = value; return MM_OK; }
  if (!strcmp(name, "gemm_w4_rowmajor")) { g_opt_w4_rowmajor = value != 0; return MM_OK; }
  if (!strcmp(name, "gemm_w4_big")) { if (value != 1 && value != 4) return MM_ERR_ARG; g_opt_w4_big = value; return MM_OK; }
  if (!strcmp(name, "gemm_w4_group_m")) { if (value < 1) return MM_ERR_ARG; g_opt_w4_group_m = value; return MM_OK; }
  if (!strcmp(name, "gemm_w4_stream")) { g_opt_w4_stream = value != 0; return MM_OK; }
  if (!strcmp(name, "gemm_w4_shuffle")) { g_opt_w4_shuffle = value != 0; return MM_OK; }
  if (!strcmp(name, "gemm_w4_diag_epi")) { g_opt_w4_diag_epi = value; return MM_OK; }
  if (!strcmp(name, "gemm_w4_stagger_slots")) { if (value < 1 || value > 32) return MM_ERR_ARG; g_opt_w4_stagger_slots = value; return MM_OK; }
  if (!strcmp(name, "gemm_w4_stagger")) { if (value < 0) return MM_ERR_ARG; g_opt_w4_stagger = value; return MM_OK; }
  if (!strcmp(name, "gemm_w4")) { if (value < 0) return MM_ERR_ARG; g_opt_w4 = value; return MM_OK; }      // 0 off, 1 the shipped schedule, n > 1: gen_gemm_w4.py SCHEDS
  if (!strcmp(name, "gemm_kernel")) { if (value < 0 || value > 6) return MM_ERR_ARG; g_opt_kernel = value; return MM_OK; }
  return MM_ERR_ARG;
}

extern "C" int mm_get_option(const char* name, int* value) {
  if (!name || !value) return MM_ERR_ARG;
  if (!strcmp(name, "gemm_tail")) { *value = g_opt_tail; return MM_OK; }
  if (!strcmp(name, "gemm_skinny")) { *value = g_opt_skinny; return MM_OK; }
  if (!strcmp(name, "gemv_stream")) { *value = g_opt_gemv_stream; return MM_OK; }
  if (!strcmp(name, "gemv_nt")) { *value = g_opt_gemv_nt; return MM_OK; }
  if (!strcmp(name, "gemv_wgs")) { *value = g_opt_gemv_wgs; return MM_OK; }
  if (!strcmp(name, "gemm_small")) { *value = g_opt_small; return MM_OK; }
  if (!strcmp(name, "gemm_persist")) { *value = g_opt_persist; return MM_OK; }
  if (!strcmp(name, "gemm_epi_pipe")) { *value = g_opt_epi_pipe; return MM_OK; }
  if (!strcmp(name, "gemm_issue_waves")) { *value = g_opt_issue_waves; return MM_OK; }
  if (!strcmp(name, "gemm_w4")) { *value = g_opt_w4; return MM_OK; }
  if (!strcmp(name, "gemm_w4_rowmajor")) { *value = g_opt_w4_rowmajor; return MM_OK; }
  if (!strcmp(name, "gemm_w4_shuffle")) { *value = g_opt_w4_shuffle; return MM_OK; }
  if (!strcmp(name, "gemm_last_kernel")) { *value = g_last_kernel; return MM_OK; }       // (bench.py: which launches the roofline's kernel took)
  if (!strcmp(name, "gemm_kernel")) { *value = g_opt_kernel; return MM_OK; }
  return MM_ERR_ARG;
}

static int gemm_launch(GemmArgs g, int dtype, int layout, hipStream_t s);

extern "C" int mm_gemm(int dtype, int layout, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
                       int ldc, const void* bias, const void* residual, int ldr, int epilogue, void* stream) {
  if (M < 0 || N < 0 || K < 0 || layout < 0 || layout > 2) return MM_ERR_ARG;
  if (M == 0 || N == 0) return MM_OK;
  if (!A || !B || !C) return MM_ERR_ARG;
  if ((epilogue & MM_EPI_BIAS) && !bias) return MM_ERR_ARG;
  if ((epilogue & MM_EPI_RESIDUAL) && !residual) return MM_ERR_ARG;
  if (epilogue & ~63) return MM_ERR_ARG;
  GemmArgs g{M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, epilogue, 0, 0, 0, 0, nullptr, 0, nullptr, 0};
  return gemm_launch(g, dtype, layout, (hipStream_t)stream);
}

// y = silu(x Wg^T) * (x Wu^T) with ONE GEMM over the fused [2I, K] gate|up weight (HF:models/llama/modeling_llama.py:163-176
// `act_fn(gate_proj(x)) * up_proj(x)`): GU [M, 2I] receives the pre-activations (saved for backward), ACT [M, I] the product.
// MM_ERR_UNSUPPORTED when the shape does not take the 256x256 LDS-DMA tile (the caller then uses mm_gemm + mm_swiglu_fwd).
extern "C" int mm_gemm_swiglu_fwd(int dtype, int M, int I, int K, const void* X, int ldx, const void* Wgu, int ldw, void* GU, int ldgu,
                                  void* ACT, int ldact, void* stream) {
  if (M < 0 || I <= 0 || K <= 0) return MM_ERR_ARG;
  if (M == 0) return MM_OK;
  if (!X || !Wgu || !GU || !ACT) return MM_ERR_ARG;
  if (dtype != MM_BF16 || (I & 127) || (K & 63) || M < 256) return MM_ERR_UNSUPPORTED;
  if ((ldgu & 3) || (ldact & 3) || (((uintptr_t)ACT) & 7) || (int64_t)2 * I * ldw * 2 >= 0xFFFFFFFFll) return MM_ERR_ALIGN;
  GemmArgs g{M, 2 * I, K, X, ldx, Wgu, ldw, GU, ldgu, nullptr, nullptr, 0, 0, 0, 0, 0, I, ACT, ldact, nullptr, 0};
  return gemm_launch(g, dtype, MM_GEMM_NT, (hipStream_t)stream);
}

// qkv = x W^T (+ b) for the fused [q | k | v] projection with RoPE applied to the q and k heads in the GEMM's epilogue
// (HF:models/llama/modeling_llama.py:232-244 q/k/v_proj, then apply_rotary_pos_emb :113-160): rope_cols = (Hq + Hkv) * 128 leading
// columns are 128-wide heads rotated with the per-token tables cos_t / sin_t [M, 64] f32 (mm_rope_table); the rest (v) is stored
// as is.  Bit-identical to mm_gemm + mm_rope_apply.  MM_ERR_UNSUPPORTED when the shape does not take the 256x256 LDS-DMA tile or the
// head width is not 128 (the caller then uses the two launches).
extern "C" int mm_gemm_rope_fwd(int dtype, int M, int N, int K, const void* X, int ldx, const void* W, int ldw, const void* bias, void* QKV,
                                int ldqkv, int rope_cols, int head_dim, const float* cos_t, const float* sin_t, void* stream) {
  if (M < 0 || N <= 0 || K <= 0 || rope_cols < 0 || rope_cols > N) return MM_ERR_ARG;
  if (M == 0) return MM_OK;
  if (!X || !W || !QKV || !cos_t || !sin_t) return MM_ERR_ARG;
  if (dtype != MM_BF16 || head_dim != 128 || (N & 127) || (rope_cols & 127) || M < 256) return MM_ERR_UNSUPPORTED;
  if ((ldqkv & 3) || (((uintptr_t)QKV) & 7) || (((uintptr_t)cos_t) & 15) || (((uintptr_t)sin_t) & 15) || (bias && (((uintptr_t)bias) & 7)))
    return MM_ERR_ALIGN;
  if ((int64_t)N * ldw * 2 >= 0xFFFFFFFFll) return MM_ERR_UNSUPPORTED;
  GemmArgs g{M, N, K, X, ldx, W, ldw, QKV, ldqkv, bias, nullptr, 0, bias ? MM_EPI_BIAS : 0, 0, 0, 0, 0, nullptr, 0, nullptr, 0};
  g.rope_cos = cos_t;
  g.rope_sin = sin_t;
  g.rope_cols = rope_cols ? rope_cols : -1;          // -1: the fused kernel with nothing to rotate (keeps the dispatch below simple)
  return gemm_launch(g, dtype, MM_GEMM_NT, (hipStream_t)stream);
}

// y = act(x W^T + b) (+ residual) with the bf16 pre-activation kept in PRE for backward: the forward of a Linear + GELU in ONE
// launch where training used three (GEMM, activation, add).  Roundings as in that form: bit-identical results.
extern "C" int mm_gemm_act_fwd(int dtype, int M, int N, int K, const void* X, int ldx, const void* W, int ldw, const void* bias,
                               void* PRE, int ldpre, void* ACT, int ldact, const void* residual, int ldr, int epilogue, void* stream) {
  if (M < 0 || N <= 0 || K <= 0) return MM_ERR_ARG;
  if (M == 0) return MM_OK;
  if (!X || !W || !PRE || !ACT) return MM_ERR_ARG;
  if ((epilogue & MM_EPI_BIAS) && !bias) return MM_ERR_ARG;
  if ((epilogue & MM_EPI_RESIDUAL) && !residual) return MM_ERR_ARG;
  const int acts = epilogue & (MM_EPI_GELU_ERF | MM_EPI_QUICK_GELU | MM_EPI_GELU_TANH);
  if (!acts || (acts & (acts - 1)) || (epilogue & ~(acts | MM_EPI_BIAS | MM_EPI_RESIDUAL))) return MM_ERR_ARG;
  if (dtype != MM_BF16 || M <= 16) return MM_ERR_UNSUPPORTED;       // the fp32 path and the decode GEMM keep the separate passes
  if ((ldpre & 3) || (((uintptr_t)PRE) & 7)) return MM_ERR_ALIGN;
  GemmArgs g{M, N, K, X, ldx, W, ldw, ACT, ldact, bias, residual, ldr, epilogue, 0, 0, 0, 0, PRE, ldpre, nullptr, 0};
  return gemm_launch(g, dtype, MM_GEMM_NT, (hipStream_t)stream);
}

// d(gate|up) [M, 2I] = swiglu'(GU) applied to d(act) = dY [M, H] . Wd [H, I], in the dgrad GEMM's epilogue: the backward of
// down_proj's input and of the SwiGLU in one launch (d(act) never goes to HBM).  Any GEMM variant (epilogue only).
extern "C" int mm_gemm_swiglu_bwd(int dtype, int M, int I, int H, const void* dY, int lddy, const void* Wd, int ldw, const void* GU,
                                  int ldgu, void* dGU, int lddgu, void* stream) {
  if (M < 0 || I <= 0 || H <= 0) return MM_ERR_ARG;
  if (M == 0) return MM_OK;
  if (!dY || !Wd || !GU || !dGU) return MM_ERR_ARG;
  if (dtype != MM_BF16 || (I & 3)) return MM_ERR_UNSUPPORTED;
  if ((ldgu & 3) || (lddgu & 3) || (((uintptr_t)GU) & 7) || (((uintptr_t)dGU) & 7)) return MM_ERR_ALIGN;
  GemmArgs g{M, I, H, dY, lddy, Wd, ldw, dGU, lddgu, nullptr, nullptr, 0, MM_EPI_SWIGLU_BWD, 0, 0, 0, 0, nullptr, 0, GU, ldgu};
  return gemm_launch(g, dtype, MM_GEMM_NN, (hipStream_t)stream);
}

// C = A.B (any layout, MM_EPI_ACCUMULATE allowed) followed by the sum of squares of the stored bf16 values: partials[0 .. n)
// (n = mm_gemm_sumsq_slots <= capacity) are OVERWRITTEN, the rest is left alone; the caller zeroes the buffer once so that
// sum(partials[0 .. capacity)) = sum(C^2) (clip_grad_norm_'s global norm assembled per weight-gradient GEMM, reference
// config_alignment.yaml:49).  Round 2 first computed the sum INSIDE the GEMM epilogue (one FMA per stored element, a slot per
// wave): that made the gradient-norm sweep unnecessary but lengthened every wgrad GEMM by more than the sweep costs
// (399.6 vs 395.5 ms/step), and merely compiling the path into the kernel cost every OTHER GEMM 1.6 %.  It is a second,
// bandwidth-bound pass over C now (C is L2/MALL-hot right after the GEMM); same ABI, same results up to summation order.
static int sumsq_blocks(int64_t M, int64_t N) {
  const int64_t v = (M * N / 4 + 255) / 256;
  return (int)(v < 1 ? 1 : (v > 1024 ? 1024 : v));
}
// slots of the in-epilogue form: 16 per 256x256 output tile
static int64_t sumsq_tile_slots(int64_t M, int64_t N) { return ((M + 255) / 256) * ((N + 255) / 256) * 16; }

extern "C" int mm_gemm_sumsq_slots(int dtype, int layout, int M, int N, int K, int64_t* slots) {
  if (!slots || M <= 0 || N <= 0 || K < 0 || layout < 0 || layout > 2) return MM_ERR_ARG;
  if (dtype != MM_BF16) return MM_ERR_UNSUPPORTED;
  const int64_t a = sumsq_blocks(M, N), b = sumsq_tile_slots(M, N);
  *slots = a > b ? a : b;                      // whichever form a call takes fits
  return MM_OK;
}

// Round 3: the sum is taken INSIDE the epilogue again, but in an instantiation of its own (EK = 5, the TN 256x256 kernel the
// decoder's weight gradients run): round 2's first cut had the extra FMA inlined into the shared epilogue, where it cost every
// other GEMM 1.6 %; the second cut (GEMM + a reduction pass over C) was neutral against the gradient-norm sweep it replaces.
// Other shapes / layouts keep the two-pass form.  Either way partials[0 .. n) are OVERWRITTEN (n <= mm_gemm_sumsq_slots).
extern "C" int mm_gemm_sumsq(int dtype, int layout, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
                             int ldc, int epilogue, float* partials, int64_t capacity, void* stream) {
  if (M < 0 || N < 0 || K < 0 || layout < 0 || layout > 2) return MM_ERR_ARG;
  if (dtype != MM_BF16) return MM_ERR_UNSUPPORTED;
  if (!A || !B || !C || !partials || capacity <= 0 || (epilogue & ~MM_EPI_ACCUMULATE)) return MM_ERR_ARG;
  if (M == 0 || N == 0) return hipMemsetAsync(partials, 0, (size_t)capacity * sizeof(float), (hipStream_t)stream) == hipSuccess ? MM_OK : MM_ERR_LAUNCH;
  int64_t need = 0;
  mm_gemm_sumsq_slots(dtype, layout, M, N, K, &need);
  if (capacity < need) return MM_ERR_ARG;
  GemmArgs g{M, N, K, A, lda, B, ldb, C, ldc, nullptr, nullptr, 0, epilogue};
  static const bool in_epi = [] { const char* e = getenv("MM_SUMSQ_EPILOGUE"); return !e || e[0] != '0'; }();
  if (in_epi && layout == MM_GEMM_TN) {
    g.ss = partials;
    const int rc = gemm_launch(g, dtype, layout, (hipStream_t)stream);
    if (rc == MM_OK) {                         // slots beyond the tiles' (the two-pass count may be larger) must read zero
      const int64_t used = sumsq_tile_slots(M, N);
      if (need > used && hipMemsetAsync(partials + used, 0, (size_t)(need - used) * sizeof(float), (hipStream_t)stream) != hipSuccess) return MM_ERR_LAUNCH;
      return MM_OK;
    }
    if (rc != MM_ERR_UNSUPPORTED) return rc;
    g.ss = nullptr;
  }
  const int nb = sumsq_blocks(M, N);
  const int rc = gemm_launch(g, dtype, layout, (hipStream_t)stream);
  if (rc != MM_OK) return rc;
  hipLaunchKernelGGL(sumsq2d_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const bf16*)C, M, N, ldc, partials);
  if (need > nb && hipMemsetAsync(partials + nb, 0, (size_t)(need - nb) * sizeof(float), (hipStream_t)stream) != hipSuccess) return MM_ERR_LAUNCH;
  MM_CHECK_LAUNCH();
  return MM_OK;
}

// ---- decode-step entry points (KV-cache decode of generate, reference model.py:595-602: M = batch <= 16 rows) ------------------
static int skinny_common(int dtype, int M, int K, const void* X, int ldx, const void* W, int ldw, int64_t wrows) {
  if (dtype != MM_BF16) return MM_ERR_UNSUPPORTED;
  if (M <= 0 || M > 16 || K <= 0 || !X || !W) return MM_ERR_ARG;
  if ((ldx & 7) || (ldw & 7) || !mm_aligned16(X) || !mm_aligned16(W)) return MM_ERR_ALIGN;
  if (wrows * ldw * 2 >= 0xFFFFFFFFll || (int64_t)16 * ldx * 2 >= 0xFFFFFFFFll) return MM_ERR_UNSUPPORTED;
  return MM_OK;
}

// x rows staged in LDS beside gemv_stream_kernel's STATIC arrays (red: 2 x 8 x 4 x 64 floats = 16 384 B, nred: 128 B) in the 160 KB of a
// CU: 143 KB leaves 896 B of slack.  kernels.decode_fits mirrors the number; a launch that still cannot get its LDS reports
// MM_ERR_UNSUPPORTED (the caller then takes the tiled decode step).
constexpr int64_t GEMV_X_LDS_MAX = 143 * 1024;
static_assert(GEMV_X_LDS_MAX + 2 * 8 * 4 * 64 * 4 + 8 * 4 * 4 <= 160 * 1024, "x + static reduction scratch must fit the CU's LDS");

static bool gemv_stream_fits(int M, int K) { return (K & 7) == 0 && (int64_t)M * K * 2 <= GEMV_X_LDS_MAX; }

template <int MODE>
static int gemv_stream_launch(const SkinnyArgs& g, unsigned nblk, hipStream_t s) {
  if (!gemv_stream_fits(g.M, g.K)) return MM_ERR_UNSUPPORTED;
  if (g.norm_w && (g.K > 8192 || !mm_aligned16(g.norm_w))) return MM_ERR_UNSUPPORTED;
  const size_t lds = (size_t)g.M * g.K * 2;
  // persistent grid: as many workgroups as the chip holds at once (2 per CU by registers, fewer when x takes most of the LDS)
  static const int ncu = [] { int d = 0, n = 256; hipDeviceProp_t p; if (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) n = p.multiProcessorCount; return n; }();
  const int per_cu = g_opt_gemv_wgs > 0 ? g_opt_gemv_wgs : ((lds + 17 * 1024) * 2 <= 160 * 1024 ? 2 : 1);
  const unsigned cap = (unsigned)(ncu * per_cu);
  const unsigned grid = nblk < cap ? nblk : cap;
#define MM_GEMV_LAUNCH(NORM, NT)                                                                              \
  do {                                                                                                        \
    auto kfn = gemv_stream_kernel<MODE, NORM, NT>;                                                            \
    if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
      (void)hipGetLastError();                                                                                \
      return MM_ERR_UNSUPPORTED;                                                                              \
    }                                                                                                         \
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), lds, s, g, (int)nblk);                                     \
  } while (0)
  if (g.norm_w) { if (g_opt_gemv_nt) MM_GEMV_LAUNCH(true, true); else MM_GEMV_LAUNCH(true, false); }
  else { if (g_opt_gemv_nt) MM_GEMV_LAUNCH(false, true); else MM_GEMV_LAUNCH(false, false); }
#undef MM_GEMV_LAUNCH
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_decode_gateup_swiglu(int dtype, int M, int I, int K, const void* X, int ldx, const void* Wgu, int ldw, void* ACT, int ldact,
                                       const void* in_norm_w, float eps, void* stream) {
  int rc = skinny_common(dtype, M, K, X, ldx, Wgu, ldw, (int64_t)2 * I);
  if (rc != MM_OK) return rc;
  if (I <= 0 || !ACT) return MM_ERR_ARG;
  if ((I & 3) || (ldact & 3) || (((uintptr_t)ACT) & 7)) return MM_ERR_ALIGN;
  SkinnyArgs g{};
  g.M = M; g.N = 2 * I; g.K = K; g.A = (const bf16*)X; g.lda = ldx; g.B = (const bf16*)Wgu; g.ldb = ldw; g.C = (bf16*)ACT; g.ldc = ldact; g.I = I;
  g.norm_w = (const bf16*)in_norm_w; g.eps = eps;
  return gemv_stream_launch<1>(g, (unsigned)((I + 7) / 8), (hipStream_t)stream);
}

extern "C" int mm_decode_qkv_rope_append(int dtype, int M, int Hq, int Hkv, int D, int K, const void* X, int ldx, const void* W, int ldw,
                                         const void* bias, void* QKV, int ldqkv, const float* cos_t, const float* sin_t, void* kdst, void* vdst,
                                         int64_t dstride, const void* in_norm_w, float eps, void* stream) {
  if (Hq <= 0 || Hkv <= 0) return MM_ERR_ARG;
  const int N = (Hq + 2 * Hkv) * D;
  int rc = skinny_common(dtype, M, K, X, ldx, W, ldw, N);
  if (rc != MM_OK) return rc;
  if (D != 128) return MM_ERR_UNSUPPORTED;
  if (!QKV || !cos_t || !sin_t || !kdst || !vdst) return MM_ERR_ARG;
  if ((ldqkv & 3) || (dstride & 3) || (((uintptr_t)QKV) & 7) || (((uintptr_t)kdst) & 7) || (((uintptr_t)vdst) & 7) || (((uintptr_t)cos_t) & 15) ||
      (((uintptr_t)sin_t) & 15))
    return MM_ERR_ALIGN;
  SkinnyArgs g{};
  g.M = M; g.N = N; g.K = K; g.A = (const bf16*)X; g.lda = ldx; g.B = (const bf16*)W; g.ldb = ldw; g.C = (bf16*)QKV; g.ldc = ldqkv;
  g.bias = (const bf16*)bias; g.Hq = Hq; g.Hkv = Hkv; g.cos_t = cos_t; g.sin_t = sin_t; g.kdst = (bf16*)kdst; g.vdst = (bf16*)vdst; g.dstride = dstride;
  g.norm_w = (const bf16*)in_norm_w; g.eps = eps;
  return gemv_stream_launch<2>(g, (unsigned)(N / 16), (hipStream_t)stream);
}

extern "C" int mm_decode_linear(int dtype, int M, int N, int K, const void* X, int ldx, const void* W, int ldw, const void* bias,
                                const void* residual, int ldr, void* C, int ldc, const void* in_norm_w, float eps, void* stream) {
  int rc = skinny_common(dtype, M, K, X, ldx, W, ldw, N);
  if (rc != MM_OK) return rc;
  if (N <= 0 || !C) return MM_ERR_ARG;
  if ((((uintptr_t)C) & 1) || ldc < N || (residual && ldr < N)) return MM_ERR_ALIGN;
  SkinnyArgs g{};
  g.M = M; g.N = N; g.K = K; g.A = (const bf16*)X; g.lda = ldx; g.B = (const bf16*)W; g.ldb = ldw; g.C = (bf16*)C; g.ldc = ldc;
  g.bias = (const bf16*)bias; g.residual = (const bf16*)residual; g.ldr = ldr; g.norm_w = (const bf16*)in_norm_w; g.eps = eps;
  return gemv_stream_launch<0>(g, (unsigned)((N + 15) / 16), (hipStream_t)stream);
}

static int gemm_launch(GemmArgs g, int dtype, int layout, hipStream_t s) {
  g.pipe = g_opt_epi_pipe;
  const int M = g.M, N = g.N, K = g.K, lda = g.lda, ldb = g.ldb, ldc = g.ldc, ldr = g.ldr, epilogue = g.epi;
  const void *A = g.A, *B = g.B;
  void* C = g.C;
  if (dtype == MM_BF16) {
    g_last_kernel = 0;
    if ((lda & 7) || (ldb & 7) || (ldc & 3) || ((epilogue & MM_EPI_RESIDUAL) && (ldr & 3))) return MM_ERR_ALIGN;
    if (!mm_aligned16(A) || !mm_aligned16(B) || (((uintptr_t)C) & 7)) return MM_ERR_ALIGN;
    // kernel choice: the LDS-DMA 256x128 kernel when its grid fills the chip, else the 128x128 register-staged one.
    // MM_GEMM_KERNEL=v1|dma forces one (A/B benchmarking).
    static const int forced_env = [] {
      const char* e = getenv("MM_GEMM_KERNEL");      // v1 | dma (256x128) | big (256x256)
      return !e ? 0 : (e[0] == 'v' ? 1 : (e[0] == 'b' ? 3 : 2));
    }();
    const int forced = g_opt_kernel ? g_opt_kernel : forced_env;
    if (forced == 0 && g_opt_skinny && layout == MM_GEMM_NT && M <= 16 && !g.swi_I && !g.rope_cols &&
        (int64_t)16 * lda * 2 < 0xFFFFFFFFll && (int64_t)16 * ldb * 2 < 0xFFFFFFFFll) {   // decode: stream W once
      if (g_opt_gemv_stream && gemv_stream_fits(M, K) && (int64_t)N * ldb * 2 < 0xFFFFFFFFll) {   // the ring-buffered form (same bits)
        SkinnyArgs q{};
        q.M = M; q.N = N; q.K = K; q.A = (const bf16*)A; q.lda = lda; q.B = (const bf16*)B; q.ldb = ldb; q.C = (bf16*)C; q.ldc = ldc;
        q.bias = (epilogue & MM_EPI_BIAS) ? (const bf16*)g.bias : nullptr;
        q.residual = (epilogue & MM_EPI_RESIDUAL) ? (const bf16*)g.residual : nullptr;
        q.ldr = ldr; q.epi = epilogue;
        return gemv_stream_launch<0>(q, (unsigned)((N + 15) / 16), s);
      }
      g_last_kernel = 20;
      dim3 grid((unsigned)((N + 15) / 16)), block(512);
      hipLaunchKernelGGL(gemm_skinny_kernel, grid, block, 0, s, g);
      MM_CHECK_LAUNCH();
      return MM_OK;
    }
    // the DMA kernels address a K-strided operand with 32-bit byte offsets over the whole matrix
    // ... and (pipelined epilogues) one wave tile of C / residual / aux with 32-bit byte offsets: 128 rows of the largest leading dimension
    const int64_t ldmax = (int64_t)ldc > ldr ? (ldc > g.ldaux ? ldc : g.ldaux) : (ldr > g.ldaux ? ldr : g.ldaux);
    const bool fits32 = ldmax * 128 * 2 + (int64_t)N * 2 < 0x7FFFFFFFll &&
                        ((layout == MM_GEMM_NT) || ((int64_t)K * ldb * 2 < 0xFFFFFFFFll && (layout != MM_GEMM_TN || (int64_t)K * lda * 2 < 0xFFFFFFFFll)));
    const int64_t tiles_128 = (int64_t)((M + 255) / 256) * ((N + 127) / 128);
    const int64_t tiles_256 = (int64_t)((M + 255) / 256) * ((N + 255) / 256);
    int variant = 0;   // 0 = v1 (128x128 register staged); DMA tiles: 1 = 256x128, 2 = 256x256, 3 = 128x128, 4 = 64x128, 5 = 64x64
    const bool keep_pre = g.C2 != nullptr && !g.swi_I;      // mm_gemm_act_fwd: compiled into the small-tile DMA kernels only
    if (g.swi_I || g.rope_cols) {             // the fused gate|up / RoPE tiles are defined on the 256x256 kernel only
      if (!fits32) return MM_ERR_UNSUPPORTED;
      variant = 2;
    } else if (keep_pre) {
      if (!fits32 || forced == 1) return MM_ERR_UNSUPPORTED;
      variant = (forced >= 4 && forced <= 6) ? forced - 1 : small_variant(M, N, K);
      if (variant < 3) variant = 3;           // a problem that would take a 256-wide tile: 128x128 (the caller may prefer the separate launches)
    } else if (fits32) {
      if (forced >= 2 && forced <= 6) variant = forced - 1;
      else if (forced == 0 && tiles_128 >= 192) variant = MM_DEFAULT_DMA_VARIANT(tiles_256);
      else if (forced == 0) variant = small_variant(M, N, K);
    }
    if (variant) {
      static const int TBM[6] = {0, 256, 256, 128, 64, 64}, TBN[6] = {0, 128, 256, 128, 128, 64};
      const int bm = TBM[variant], bn = TBN[variant];
      g.nbm = (M + bm - 1) / bm;
      g.nbn = g.swi_I ? g.swi_I / 128 : (N + bn - 1) / bn;
      const int64_t nwg = (int64_t)g.nbm * g.nbn;
      if (nwg > 0x7FFFFFFF) return MM_ERR_ARG;
      const size_t lds = 2 * (bm + bn) * G_BK * 2;
      static const int ncu = [] { int d = 0, n = 256; hipDeviceProp_t p; if (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) n = p.multiProcessorCount; return n; }();
      // persistent: one resident workgroup per CU walks the tiles; otherwise one tile each
      int64_t nblk = g_opt_persist ? (nwg < (int64_t)ncu ? nwg : (int64_t)ncu) : nwg;
      // NT / NN 256x256 tiles: the 4-wave kernel (A K-contiguous, whole K-steps, the epilogue kinds it instantiates)
      const bool acts = (epilogue & (MM_EPI_GELU_ERF | MM_EPI_QUICK_GELU | MM_EPI_GELU_TANH)) != 0;
      // (a last round at most half full: its tiles are cut into 256 x 128 halves, g.tail -- plain epilogue kinds only)
      const int64_t rem4 = g_opt_persist ? nwg % ncu : 0;
      const bool tail4 = g_opt_tail && !g.swi_I && !g.rope_cols && !(epilogue & MM_EPI_SWIGLU_BWD) && rem4 > 0 && 2 * rem4 <= ncu;
      g_last_kernel = variant;
      if (g_opt_w4 && variant == 2 && K >= 192 && !g.ss && !acts) {
        g_last_kernel = 10;
        int64_t nb4 = g_opt_persist ? (nwg < (int64_t)ncu ? nwg : (int64_t)ncu) : nwg;
        if (tail4) {
          g.tail = (int)rem4;
          nb4 = nwg >= ncu ? ncu : 2 * rem4;
        }
        dim3 grid4((unsigned)nb4), block4(256);
        const size_t lds = 160 * 1024;                  // the ring (128 KB) + 8 KB per wave for the row-major epilogue's transposition
        g.stagger = g_opt_w4_stagger;
        g.stagger_slots = g_opt_w4_stagger_slots;
        g.diag_epi = g_opt_w4_diag_epi;
        g.shuffle = g_opt_w4_shuffle;
        g.group_m = g_opt_w4_group_m;
        g.stream_epi = g_opt_w4_stream;
        g.rowmajor = g_opt_w4_rowmajor && (N & 7) == 0 && (ldc & 7) == 0 && mm_aligned16(C) &&
                     (!(epilogue & MM_EPI_RESIDUAL) || ((ldr & 7) == 0 && mm_aligned16(g.residual))) &&
                     (!(epilogue & MM_EPI_BIAS) || mm_aligned16(g.bias)) &&
                     (!(epilogue & MM_EPI_SWIGLU_BWD) || ((g.ldaux & 7) == 0 && mm_aligned16(g.aux)));
        if (g.swi_I) g.rowmajor = g_opt_w4_rowmajor;       // mm_gemm_swiglu_fwd: 8-byte accesses, alignment checked at the entry point
#define MM_LAUNCH_W4(AKC, BKC, EK, SCHED)                                                                                    \
  do {                                                                                                                   \
    auto kfn = gemm_bf16_w4_kernel<AKC, BKC, EK, SCHED>;                                                                     \
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
    hipLaunchKernelGGL(kfn, grid4, block4, lds, s, g);                                                                   \
  } while (0)
#define MM_W4_CASE(AKC, BKC, SCHED) case SCHED: MM_LAUNCH_W4(AKC, BKC, 0, SCHED); break;
#ifdef MM_W4_DIAG
#define MM_W4_CASES(AKC, BKC) MM_W4_CASE(AKC, BKC, 2) MM_W4_CASE(AKC, BKC, 3) MM_W4_CASE(AKC, BKC, 4) MM_W4_CASE(AKC, BKC, 5) MM_W4_CASE(AKC, BKC, 6) MM_W4_CASE(AKC, BKC, 7) MM_W4_CASE(AKC, BKC, 104) MM_W4_CASE(AKC, BKC, 105) MM_W4_CASE(AKC, BKC, 111) MM_W4_CASE(AKC, BKC, 112) MM_W4_CASE(AKC, BKC, 113) MM_W4_CASE(AKC, BKC, 114) MM_W4_CASE(AKC, BKC, 115) MM_W4_CASE(AKC, BKC, 117) MM_W4_CASE(AKC, BKC, 121)
#else
#define MM_W4_CASES(AKC, BKC) MM_W4_CASE(AKC, BKC, 4) MM_W4_CASE(AKC, BKC, 6) MM_W4_CASE(AKC, BKC, 7)
#endif
        // schedule: 1 everywhere (one barrier per K-step), except -- "gemm_w4_big" = 4 -- on operands that stream from beyond the
        // Infinity Cache (a wide N or a long K): there the split-barrier schedule 4 (a DMA piece gets 105-168 MFMAs to land instead of
        // 68-128) measured +3...+9 % (tools/w4_check.py --scheds), while on MALL-resident 4096-sized operands it costs 3 %
        const bool big = g_opt_w4_big == 4 && g_opt_w4 == 1 && ((int64_t)N >= 14336 || (int64_t)K >= 14336);
        if (g.rope_cols) { if (layout != MM_GEMM_NT) return MM_ERR_ARG; MM_LAUNCH_W4(true, true, 4, 1); }
        else if (g.swi_I) { if (layout != MM_GEMM_NT) return MM_ERR_ARG; if (big) MM_LAUNCH_W4(true, true, 3, 4); else MM_LAUNCH_W4(true, true, 3, 1); }
        else if (epilogue & MM_EPI_SWIGLU_BWD) { if (layout != MM_GEMM_NN) return MM_ERR_ARG; if (big) MM_LAUNCH_W4(true, false, 2, 4); else MM_LAUNCH_W4(true, false, 2, 1); }
        else if (layout == MM_GEMM_NT) { switch (big ? 4 : g_opt_w4) { MM_W4_CASES(true, true) default: MM_LAUNCH_W4(true, true, 0, 1); } }
        else if (layout == MM_GEMM_NN) { switch (big ? 4 : g_opt_w4) { MM_W4_CASES(true, false) default: MM_LAUNCH_W4(true, false, 0, 1); } }
        else { switch (g_opt_w4) { MM_W4_CASES(false, false) default: MM_LAUNCH_W4(false, false, 0, 1); } }
#undef MM_W4_CASES
#undef MM_W4_CASE
#undef MM_LAUNCH_W4
        MM_CHECK_LAUNCH();
        return MM_OK;
      }
      if (g_opt_persist && g_opt_tail && variant == 2 && !g.swi_I && !g.rope_cols) {       // wave quantisation: see the kernel
        const int64_t rem = nwg % ncu;
        if (rem > 0 && 2 * rem <= ncu) {
          g.tail = (int)rem;
          nblk = nwg >= ncu ? ncu : 2 * rem;
        }
      }
      dim3 grid((unsigned)nblk), block(512);
#define MM_LAUNCH_ONE(...)                                                                                               \
  do {                                                                                                                   \
    auto kfn = gemm_bf16_dma_kernel<__VA_ARGS__>;                                                                        \
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
    hipLaunchKernelGGL(kfn, grid, block, lds, s, g);                                                                     \
  } while (0)
#define MM_LAUNCH_DMA(AKC, BKC, EK)                                                                                      \
  do {                                                                                                                   \
    if (variant == 1) MM_LAUNCH_ONE(AKC, BKC, 256, 128, 4, 2, 8, EK);                                                    \
    else if (variant == 3) MM_LAUNCH_ONE(AKC, BKC, 128, 128, 2, 2, 4, EK);                                               \
    else if (variant == 4) MM_LAUNCH_ONE(AKC, BKC, 64, 128, 2, 2, 4, EK);                                                \
    else if (variant == 5) MM_LAUNCH_ONE(AKC, BKC, 64, 64, 2, 2, 4, EK);                                                 \
    else if (g_opt_issue_waves == 4) MM_LAUNCH_ONE(AKC, BKC, 256, 256, 2, 2, 4, EK);                                     \
    else MM_LAUNCH_ONE(AKC, BKC, 256, 256, 2, 2, 8, EK);                                                                 \
  } while (0)
      // epilogue kind = kernel instantiation: the SwiGLU ones exist for the layout their entry point uses only
      if (g.ss) {                                                         // mm_gemm_sumsq in the epilogue: TN, 256x256
        if (layout != MM_GEMM_TN || variant != 2 || (epilogue & ~MM_EPI_ACCUMULATE)) return MM_ERR_UNSUPPORTED;
        if (g_opt_issue_waves == 4) MM_LAUNCH_ONE(false, false, 256, 256, 2, 2, 4, 5);
        else MM_LAUNCH_ONE(false, false, 256, 256, 2, 2, 8, 5);
      } else if (g.rope_cols) {                                           // mm_gemm_rope_fwd: NT, 256x256
        if (layout != MM_GEMM_NT) return MM_ERR_ARG;
        if (g_opt_issue_waves == 4) MM_LAUNCH_ONE(true, true, 256, 256, 2, 2, 4, 4);
        else MM_LAUNCH_ONE(true, true, 256, 256, 2, 2, 8, 4);
      } else if (g.swi_I) {                                               // mm_gemm_swiglu_fwd: NT, 256x256 (variant 2 above)
        if (layout != MM_GEMM_NT) return MM_ERR_ARG;
        if (g_opt_issue_waves == 4) MM_LAUNCH_ONE(true, true, 256, 256, 2, 2, 4, 3);
        else MM_LAUNCH_ONE(true, true, 256, 256, 2, 2, 8, 3);
      } else if (epilogue & MM_EPI_SWIGLU_BWD) {                          // mm_gemm_swiglu_bwd: NN
        if (layout != MM_GEMM_NN) return MM_ERR_ARG;
        MM_LAUNCH_DMA(true, false, 2);
      } else if (epilogue & (MM_EPI_GELU_ERF | MM_EPI_QUICK_GELU | MM_EPI_GELU_TANH)) {      // activation: NT (mm_gemm / mm_gemm_act_fwd)
        if (layout != MM_GEMM_NT) return MM_ERR_UNSUPPORTED;
        MM_LAUNCH_DMA(true, true, 1);
      } else {
        switch (layout) {
          case MM_GEMM_NT: MM_LAUNCH_DMA(true, true, 0); break;
          case MM_GEMM_NN: MM_LAUNCH_DMA(true, false, 0); break;
          default: MM_LAUNCH_DMA(false, false, 0); break;
        }
      }
#undef MM_LAUNCH_ONE
#undef MM_LAUNCH_DMA
    } else {
      if (g.ss) return MM_ERR_UNSUPPORTED;
      g.nbm = (M + BM - 1) / BM;
      g.nbn = (N + BN - 1) / BN;
      const int64_t nwg = (int64_t)g.nbm * g.nbn;
      if (nwg > 0x7FFFFFFF) return MM_ERR_ARG;
      const size_t lds = 4 * TILE_BYTES;
      dim3 grid((unsigned)nwg), block(256);
      switch (layout) {
        case MM_GEMM_NT: hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, block, lds, s, g); break;
        case MM_GEMM_NN: hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), grid, block, lds, s, g); break;
        default: hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, block, lds, s, g); break;
      }
    }
  } else if (dtype == MM_F32) {
    g.nbm = (M + 63) / 64;
    g.nbn = (N + 63) / 64;
    dim3 grid((unsigned)(g.nbm * g.nbn)), block(256);
    switch (layout) {
      case MM_GEMM_NT: hipLaunchKernelGGL((gemm_f32_kernel<MM_GEMM_NT>), grid, block, 0, s, g); break;
      case MM_GEMM_NN: hipLaunchKernelGGL((gemm_f32_kernel<MM_GEMM_NN>), grid, block, 0, s, g); break;
      default: hipLaunchKernelGGL((gemm_f32_kernel<MM_GEMM_TN>), grid, block, 0, s, g); break;
    }
  } else {
    return MM_ERR_UNSUPPORTED;
  }
  MM_CHECK_LAUNCH();
  return MM_OK;
}

extern "C" int mm_colsum(int dtype, const void* X, int M, int N, int ldx, void* out, int accumulate, void* stream) {
  if (!X || !out || M < 0 || N <= 0) return MM_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int vn = dtype == MM_BF16 ? 8 : 4;
  if ((ldx % vn) || !mm_aligned16(X)) return MM_ERR_ALIGN;
  dim3 grid((N + 8 * vn - 1) / (8 * vn)), block(256);
  if (dtype == MM_BF16)
    hipLaunchKernelGGL(colsum_kernel<bf16>, grid, block, 0, s, (const bf16*)X, M, N, ldx, (bf16*)out, accumulate);
  else
    hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, s, (const float*)X, M, N, ldx, (float*)out, accumulate);
  MM_CHECK_LAUNCH();
  return MM_OK;
}
