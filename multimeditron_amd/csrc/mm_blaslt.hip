// Plain bf16 GEMMs through the vendor library (hipBLASLt), bound at RUN time.
//
// libmmhip's own 256x256 LDS-DMA kernel (mm_gemm.hip) carries every fused epilogue of the step and wins on the weight-gradient
// (TN) shapes; on the PLAIN forward (NT) and input-gradient (NN) shapes of the 8B step the vendor's hand-scheduled assembly
// kernels (256x256x64 macro tile on four 128x128 waves, stream-K) are 5-34 % faster (tools/gemm_vs_library.py,
// profiles/r03_gemm_vs_library.md).  mm_gemm_lib hands exactly those plain products -- C = A B^T or A B, optionally + residual,
// fp32 accumulation, bf16 in and out -- to the library; everything else stays on the hand-written kernels.  Same results bit for
// bit on the step's shapes (tests/test_kernels_gpu.py::test_gemm_lib_matches_mm_gemm).
//
// The library is the copy the process already holds (torch ships one; RTLD_NOLOAD first), found with dlopen / dlsym: libmmhip has
// no link-time dependency on it, and mm_gemm_lib returns MM_ERR_UNSUPPORTED when it is absent (the caller keeps mm_gemm).
#include <dlfcn.h>
#include <hipblaslt/hipblaslt.h>

#include <map>
#include <mutex>
#include <tuple>

#include "mm_common.h"

namespace {
struct Lt {
  void* h = nullptr;
  decltype(&hipblasLtCreate) create = nullptr;
  decltype(&hipblasLtMatmulDescCreate) desc_create = nullptr;
  decltype(&hipblasLtMatmulDescSetAttribute) desc_set = nullptr;
  decltype(&hipblasLtMatrixLayoutCreate) layout_create = nullptr;
  decltype(&hipblasLtMatmulPreferenceCreate) pref_create = nullptr;
  decltype(&hipblasLtMatmulPreferenceSetAttribute) pref_set = nullptr;
  decltype(&hipblasLtMatmulAlgoGetHeuristic) heuristic = nullptr;
  decltype(&hipblasLtMatmul) matmul = nullptr;
  hipblasLtHandle_t handle = nullptr;
  bool ok = false;
};

Lt& lt() {
  static Lt r = [] {
    Lt x;
    const char* names[] = {"libhipblaslt.so", "libhipblaslt.so.1", "libhipblaslt.so.0", "/opt/rocm/lib/libhipblaslt.so"};
    for (const char* n : names) {
      x.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);      // the copy torch.matmul already runs on, if any
      if (x.h) break;
    }
    if (!x.h)
      for (const char* n : names) {
        x.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (x.h) break;
      }
    if (!x.h) return x;
#define MM_LT_SYM(field, sym) x.field = (decltype(x.field))dlsym(x.h, #sym)
    MM_LT_SYM(create, hipblasLtCreate);
    MM_LT_SYM(desc_create, hipblasLtMatmulDescCreate);
    MM_LT_SYM(desc_set, hipblasLtMatmulDescSetAttribute);
    MM_LT_SYM(layout_create, hipblasLtMatrixLayoutCreate);
    MM_LT_SYM(pref_create, hipblasLtMatmulPreferenceCreate);
    MM_LT_SYM(pref_set, hipblasLtMatmulPreferenceSetAttribute);
    MM_LT_SYM(heuristic, hipblasLtMatmulAlgoGetHeuristic);
    MM_LT_SYM(matmul, hipblasLtMatmul);
#undef MM_LT_SYM
    if (!(x.create && x.desc_create && x.desc_set && x.layout_create && x.pref_create && x.pref_set && x.heuristic && x.matmul)) return x;
    if (x.create(&x.handle) != HIPBLAS_STATUS_SUCCESS) return x;
    x.ok = true;
    return x;
  }();
  return r;
}

struct Plan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr, d = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t ws = 0;
  bool ok = false;
};
using Key = std::tuple<int, int, int, int, int, int, int, int, long long>;
std::map<Key, Plan> g_plans;
std::mutex g_mu;
}  // namespace

extern "C" int mm_gemm_lib_available(void) { return lt().ok ? 1 : 0; }

// C[M,N] (+= residual) = A . B^T (MM_GEMM_NT: A [M,K], B [N,K]) or A . B (MM_GEMM_NN: A [M,K], B [K,N]); row-major, bf16, fp32
// accumulation.  residual may be NULL.  ws / ws_bytes: caller-provided workspace (the library's stream-K kernels use it).
extern "C" int mm_gemm_lib(int dtype, int layout, int M, int N, int K, const void* A, int lda, const void* B, int ldb, const void* residual,
                           int ldr, void* C, int ldc, void* ws, int64_t ws_bytes, void* stream) {
  if (dtype != MM_BF16 || (layout != MM_GEMM_NT && layout != MM_GEMM_NN)) return MM_ERR_UNSUPPORTED;
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || lda < K || ldc < N || (residual && ldr < N) || ldb < (layout == MM_GEMM_NT ? K : N))
    return MM_ERR_ARG;
  Lt& L = lt();
  if (!L.ok) return MM_ERR_UNSUPPORTED;
  const Key key{layout, M, N, K, lda, ldb, ldc, residual ? ldr : -1, (long long)ws_bytes};
  Plan* p = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_plans.find(key);
    if (it == g_plans.end()) {
      Plan q;
      // the library is column-major: row-major C [M,N] is its N x M matrix D = op(B) . A^T-as-stored
      //   NT: D (N x M) = T(Bcm [K x N, ld ldb]) . Acm [K x M, ld lda]      NN: D (N x M) = Bcm [N x K, ld ldb] . Acm [K x M, ld lda]
      const hipblasOperation_t ta = layout == MM_GEMM_NT ? HIPBLAS_OP_T : HIPBLAS_OP_N, tb = HIPBLAS_OP_N;
      bool ok = L.desc_create(&q.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) == HIPBLAS_STATUS_SUCCESS;
      ok = ok && L.desc_set(q.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)) == HIPBLAS_STATUS_SUCCESS;
      ok = ok && L.desc_set(q.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)) == HIPBLAS_STATUS_SUCCESS;
      if (layout == MM_GEMM_NT) ok = ok && L.layout_create(&q.a, HIP_R_16BF, (uint64_t)K, (uint64_t)N, (int64_t)ldb) == HIPBLAS_STATUS_SUCCESS;
      else ok = ok && L.layout_create(&q.a, HIP_R_16BF, (uint64_t)N, (uint64_t)K, (int64_t)ldb) == HIPBLAS_STATUS_SUCCESS;
      ok = ok && L.layout_create(&q.b, HIP_R_16BF, (uint64_t)K, (uint64_t)M, (int64_t)lda) == HIPBLAS_STATUS_SUCCESS;
      ok = ok && L.layout_create(&q.c, HIP_R_16BF, (uint64_t)N, (uint64_t)M, (int64_t)(residual ? ldr : ldc)) == HIPBLAS_STATUS_SUCCESS;
      ok = ok && L.layout_create(&q.d, HIP_R_16BF, (uint64_t)N, (uint64_t)M, (int64_t)ldc) == HIPBLAS_STATUS_SUCCESS;
      if (ok) {
        hipblasLtMatmulPreference_t pref = nullptr;
        ok = L.pref_create(&pref) == HIPBLAS_STATUS_SUCCESS;
        const uint64_t wsb = ws ? (uint64_t)ws_bytes : 0;
        ok = ok && L.pref_set(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsb, sizeof(wsb)) == HIPBLAS_STATUS_SUCCESS;
        hipblasLtMatmulHeuristicResult_t res[1];
        int found = 0;
        ok = ok && L.heuristic(L.handle, q.desc, q.a, q.b, q.c, q.d, pref, 1, res, &found) == HIPBLAS_STATUS_SUCCESS && found > 0;
        if (ok) {
          q.algo = res[0].algo;
          q.ws = res[0].workspaceSize;
          ok = q.ws <= wsb;
        }
      }
      q.ok = ok;
      it = g_plans.emplace(key, q).first;
    }
    p = &it->second;
  }
  if (!p->ok) return MM_ERR_UNSUPPORTED;
  const float alpha = 1.f, beta = residual ? 1.f : 0.f;
  const hipblasStatus_t st = L.matmul(L.handle, p->desc, &alpha, B, p->a, A, p->b, &beta, residual ? residual : C, p->c, C, p->d, &p->algo, ws,
                                      p->ws, (hipStream_t)stream);
  return st == HIPBLAS_STATUS_SUCCESS ? MM_OK : MM_ERR_LAUNCH;
}
