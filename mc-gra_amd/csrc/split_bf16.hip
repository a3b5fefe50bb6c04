// OPT-IN (MCGRA_SPLIT_BF16=1): P1 = (H Kf H) Xc evaluated as a 3-plane bf16 split on the bf16 matrix cores.
//
// x = x0 + x1 + x2 with x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1): 24 mantissa bits, the residuals are
// exact in fp32.  a b = sum_{i+j<=2} a_i b_j + O(2^-24 |a b|): six bf16 products, each exact in the fp32 accumulator
// of the MFMA -- the same error class as an fp32 product (measured on 10 240^3: 4.3e-7 of |A||B| vs 3.4e-7 for the
// fp32 MFMA kernel, scripts/split_bf16_probe.py).  The six plane products are ONE plain bf16 GEMM with the planes
// concatenated along K,  A' = [a0 a0 a0 a1 a1 a2] (n x 6k),  B'^T = [b0 b1 b2 b0 b1 b0] (n x 6k),  which is exactly
// the "plain library GEMM" case: it is handed to hipBLASLt (1.25 PFLOP/s bf16 issued = 209 TFLOP/s fp32-equivalent,
// against 120-128 for the hand-written fp32 MFMA SYMM).  The library is dlopen'ed on first use, so the default
// (fp32) build has no dependency on it.  B' is produced row-wise from adj_norm (Xc^T[j][k] = adj_norm[j][k] - mean_j
// by the symmetry of adj_norm at eps == 0), A' once per graph from the lower tile storage of the constant Gram.
#include <dlfcn.h>
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mcgra.h"
#include "common.h"

namespace mcgra {

namespace {
struct LtApi {
  void* lib = nullptr;
  bool tried = false, ok = false;
  hipblasLtHandle_t handle = nullptr;
  void* workspace = nullptr;
  size_t workspace_bytes = 0;
  decltype(&hipblasLtCreate) Create = nullptr;
  decltype(&hipblasLtMatmulDescCreate) DescCreate = nullptr;
  decltype(&hipblasLtMatmulDescDestroy) DescDestroy = nullptr;
  decltype(&hipblasLtMatmulDescSetAttribute) DescSet = nullptr;
  decltype(&hipblasLtMatrixLayoutCreate) LayoutCreate = nullptr;
  decltype(&hipblasLtMatrixLayoutDestroy) LayoutDestroy = nullptr;
  decltype(&hipblasLtMatmulPreferenceCreate) PrefCreate = nullptr;
  decltype(&hipblasLtMatmulPreferenceDestroy) PrefDestroy = nullptr;
  decltype(&hipblasLtMatmulPreferenceSetAttribute) PrefSet = nullptr;
  decltype(&hipblasLtMatmulAlgoGetHeuristic) Heuristic = nullptr;
  decltype(&hipblasLtMatmul) Matmul = nullptr;
};
LtApi g_lt;

template <typename F>
bool sym(void* lib, const char* name, F* out) {
  *out = reinterpret_cast<F>(dlsym(lib, name));
  return *out != nullptr;
}

LtApi* lt() {
  if (g_lt.tried) return g_lt.ok ? &g_lt : nullptr;
  g_lt.tried = true;
  // by SONAME first: a process that already holds a copy (PyTorch bundles one) keeps using that copy
  const char* names[] = {"libhipblaslt.so.1", "libhipblaslt.so", "/opt/rocm/lib/libhipblaslt.so.1"};
  for (const char* nm : names) {
    g_lt.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (g_lt.lib) break;
  }
  if (!g_lt.lib) { set_error("hipBLASLt not found (%s)", dlerror()); return nullptr; }
  void* L = g_lt.lib;
  bool ok = sym(L, "hipblasLtCreate", &g_lt.Create) && sym(L, "hipblasLtMatmulDescCreate", &g_lt.DescCreate) &&
            sym(L, "hipblasLtMatmulDescDestroy", &g_lt.DescDestroy) &&
            sym(L, "hipblasLtMatmulDescSetAttribute", &g_lt.DescSet) &&
            sym(L, "hipblasLtMatrixLayoutCreate", &g_lt.LayoutCreate) &&
            sym(L, "hipblasLtMatrixLayoutDestroy", &g_lt.LayoutDestroy) &&
            sym(L, "hipblasLtMatmulPreferenceCreate", &g_lt.PrefCreate) &&
            sym(L, "hipblasLtMatmulPreferenceDestroy", &g_lt.PrefDestroy) &&
            sym(L, "hipblasLtMatmulPreferenceSetAttribute", &g_lt.PrefSet) &&
            sym(L, "hipblasLtMatmulAlgoGetHeuristic", &g_lt.Heuristic) && sym(L, "hipblasLtMatmul", &g_lt.Matmul);
  if (!ok) { set_error("hipBLASLt: missing symbol"); return nullptr; }
  if (g_lt.Create(&g_lt.handle) != HIPBLAS_STATUS_SUCCESS) { set_error("hipblasLtCreate failed"); return nullptr; }
  g_lt.workspace_bytes = (size_t)128 << 20;
  if (hipMalloc(&g_lt.workspace, g_lt.workspace_bytes) != hipSuccess) { g_lt.workspace = nullptr; g_lt.workspace_bytes = 0; }
  g_lt.ok = true;
  return &g_lt;
}

__device__ __forceinline__ void split3(float x, __hip_bfloat16& p0, __hip_bfloat16& p1, __hip_bfloat16& p2) {
  p0 = __float2bfloat16(x);
  const float r1 = x - __bfloat162float(p0);
  p1 = __float2bfloat16(r1);
  const float r2 = r1 - __bfloat162float(p1);
  p2 = __float2bfloat16(r2);
}

// A' row i from the symmetric S given in lower tile storage (element (i, j) stored iff j < (i / 128 + 1) * 128)
__global__ __launch_bounds__(256) void k_split3_sym(int n, int ld, const float* __restrict__ S, int kp,
                                                    __hip_bfloat16* __restrict__ out) {
  const int i = blockIdx.x;
  __hip_bfloat16* o = out + (size_t)i * 6 * kp;
  const int lim = (i / SYM_TILE + 1) * SYM_TILE;
  for (int j = threadIdx.x; j < kp; j += 256) {
    float x = 0.f;
    if (j < n) x = j < lim ? S[(size_t)i * ld + j] : S[(size_t)j * ld + i];
    __hip_bfloat16 p0, p1, p2;
    split3(x, p0, p1, p2);
    o[j] = p0; o[kp + j] = p0; o[2 * kp + j] = p0; o[3 * kp + j] = p1; o[4 * kp + j] = p1; o[5 * kp + j] = p2;
  }
}
// B'^T row j = planes of (X[j][k] - mean[j]) over k, order [b0 b1 b2 b0 b1 b0]
__global__ __launch_bounds__(256) void k_split3_rows(int n, int ld, const float* __restrict__ X,
                                                     const float* __restrict__ mean, int kp,
                                                     __hip_bfloat16* __restrict__ out) {
  const int j = blockIdx.x;
  __hip_bfloat16* o = out + (size_t)j * 6 * kp;
  const float mu = mean[j];
  for (int k = threadIdx.x; k < kp; k += 256) {
    const float x = k < n ? X[(size_t)j * ld + k] - mu : 0.f;
    __hip_bfloat16 p0, p1, p2;
    split3(x, p0, p1, p2);
    o[k] = p0; o[kp + k] = p1; o[2 * kp + k] = p2; o[3 * kp + k] = p0; o[4 * kp + k] = p1; o[5 * kp + k] = p0;
  }
}
}  // namespace

bool split_bf16_available() { return lt() != nullptr; }
int split_bf16_kpad(int n) { return (n + 7) & ~7; }

void split3_planes_sym(hipStream_t st, int n, int ld, const float* S_lower, void* Acat) {
  hipLaunchKernelGGL(k_split3_sym, dim3(n), dim3(256), 0, st, n, ld, S_lower, split_bf16_kpad(n), (__hip_bfloat16*)Acat);
}
void split3_planes_rows(hipStream_t st, int n, int ld, const float* X, const float* mean, void* Bcat) {
  hipLaunchKernelGGL(k_split3_rows, dim3(n), dim3(256), 0, st, n, ld, X, mean, split_bf16_kpad(n), (__hip_bfloat16*)Bcat);
}

// C[row0 .. row0+nrows)[0..n) (row-major, ldc) = A'[row0 ...] B'^T, fp32 out.  Column-major view for the library:
// D^T (n x nrows) = op_T(B' as (6kp x n)) * (A' rows as (6kp x nrows)).
int split_bf16_gemm(hipStream_t st, int n, int row0, int nrows, const void* Acat, const void* Bcat, float* C, int ldc) {
  LtApi* L = lt();
  if (!L) return MCGRA_ENOSUP;
  if (nrows <= 0) return 0;
  const int64_t K = (int64_t)6 * split_bf16_kpad(n);
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
  hipblasLtMatmulPreference_t pref = nullptr;
  int rc = 0;
  auto fail = [&](const char* what) { set_error("hipBLASLt: %s failed", what); rc = MCGRA_EHIP; };
  const int32_t opT = HIPBLAS_OP_T, opN = HIPBLAS_OP_N;
  if (L->DescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) fail("MatmulDescCreate");
  if (!rc && L->DescSet(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opT, sizeof(opT)) != HIPBLAS_STATUS_SUCCESS) fail("TRANSA");
  if (!rc && L->DescSet(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opN, sizeof(opN)) != HIPBLAS_STATUS_SUCCESS) fail("TRANSB");
  if (!rc && L->LayoutCreate(&la, HIP_R_16BF, K, n, K) != HIPBLAS_STATUS_SUCCESS) fail("layout A");
  if (!rc && L->LayoutCreate(&lb, HIP_R_16BF, K, nrows, K) != HIPBLAS_STATUS_SUCCESS) fail("layout B");
  if (!rc && L->LayoutCreate(&lc, HIP_R_32F, n, nrows, ldc) != HIPBLAS_STATUS_SUCCESS) fail("layout C");
  if (!rc && L->PrefCreate(&pref) != HIPBLAS_STATUS_SUCCESS) fail("PreferenceCreate");
  uint64_t wsb = L->workspace_bytes;
  if (!rc && L->PrefSet(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsb, sizeof(wsb)) != HIPBLAS_STATUS_SUCCESS)
    fail("workspace preference");
  // the heuristic query costs tens of microseconds of host time: keep the last shape's answer
  static int c_n = -1, c_rows = -1, c_ldc = -1;
  static hipblasLtMatmulHeuristicResult_t heur[1];
  if (!rc && !(c_n == n && c_rows == nrows && c_ldc == ldc)) {
    int found = 0;
    c_n = -1;
    if (L->Heuristic(L->handle, desc, la, lb, lc, lc, pref, 1, heur, &found) != HIPBLAS_STATUS_SUCCESS || found < 1)
      fail("AlgoGetHeuristic");
    else { c_n = n; c_rows = nrows; c_ldc = ldc; }
  }
  if (!rc) {
    const float one = 1.f, zero = 0.f;
    const char* a = (const char*)Acat + (size_t)row0 * K * 2;
    float* c = C + (size_t)row0 * ldc;
    if (L->Matmul(L->handle, desc, &one, Bcat, la, a, lb, &zero, c, lc, c, lc, &heur[0].algo, L->workspace,
                  L->workspace_bytes, st) != HIPBLAS_STATUS_SUCCESS)
      fail("Matmul");
  }
  if (pref) L->PrefDestroy(pref);
  if (la) L->LayoutDestroy(la);
  if (lb) L->LayoutDestroy(lb);
  if (lc) L->LayoutDestroy(lc);
  if (desc) L->DescDestroy(desc);
  return rc;
}

}  // namespace mcgra
