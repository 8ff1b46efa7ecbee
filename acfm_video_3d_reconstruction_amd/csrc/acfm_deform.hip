// Template deformation apply and its backward on the f32 matrix cores.
//
// The reference solves (L^T L + A^T A) v = L^T L v_mean + A^T (A v_mean + delta) per frame with a
// dense Cholesky (multiframe/main.py:586-609).  With P = (L^T L + A^T A)^-1 A^T (factorised once
// per optimiser step, deform.py) this is  v_n = v_mean + P delta_n, i.e. three small dense
// contractions per step that share P:
//     verts[n,v,c]      = mean[v,c] + sum_k P[v,k] delta[n,k,c]          (V x K_h) . (K_h x 3N)
//     grad_delta[n,k,c] = sum_v P[v,k] g[n,v,c]                          (K_h x V) . (V x 3N)
//     grad_P[v,k]       = sum_{n,c} g[n,v,c] delta[n,k,c]                (V x 3N) . (3N x K_h)
//     grad_mean[v,c]    = sum_n g[n,v,c]
// Each 16x16 output tile is one wave of v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: exact
// fp32 FMA chain in k order, no reduced precision), operands read straight from HBM/L2 -- the
// whole problem is a few hundred KB, the kernels are launch-latency sized.
#include "acfm_common.h"

namespace acfm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// One wave computes a 16x16 tile of C = A.B with K_total inner steps.  fa(i, k) / fb(k, j)
// return the operand (0 outside the matrix); lane l feeds A[i = l&15][k = 4s + (l>>4)] and
// B[k = 4s + (l>>4)][j = l&15]; on return acc[r] = C[i = 4*(l>>4) + r][j = l&15].
template <class FA, class FB>
__device__ __forceinline__ f32x4 mfma_tile_16x16(int lane, int k_begin, int k_end, FA&& fa, FB&& fb) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int i = lane & 15, kk = lane >> 4;
  for (int k0 = k_begin; k0 < k_end; k0 += 4) {
    const float a = fa(i, k0 + kk);
    const float b = fb(k0 + kk, i);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  return acc;
}

// verts[n,v,c] = mean[v,c] + sum_k P[v,k] delta[n,k,c];  grid (ceil(V/16), ceil(3N/64)), 4 waves
__global__ __launch_bounds__(256) void k_deform_apply(const float* __restrict__ mean,
                                                      const float* __restrict__ P,
                                                      const float* __restrict__ delta, int N, int V,
                                                      int Kh, float* __restrict__ verts) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int v0 = blockIdx.x * 16, j0 = (blockIdx.y * 4 + wv) * 16;
  const int J = 3 * N;
  if (j0 >= J) return;
  const f32x4 acc = mfma_tile_16x16(
      lane, 0, Kh,
      [&](int i, int k) { const int v = v0 + i; return (v < V && k < Kh) ? P[(size_t)v * Kh + k] : 0.f; },
      [&](int k, int jj) {
        const int j = j0 + jj;
        if (j >= J || k >= Kh) return 0.f;
        const int n = j / 3, c = j - 3 * n;
        return delta[((size_t)n * Kh + k) * 3 + c];
      });
  const int j = j0 + (lane & 15);
  if (j >= J) return;
  const int n = j / 3, c = j - 3 * n;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int v = v0 + (lane >> 4) * 4 + r;
    if (v < V) verts[((size_t)n * V + v) * 3 + c] = mean[(size_t)v * 3 + c] + acc[r];
  }
}

// grad_delta[n,k,c] = sum_v P[v,k] g[n,v,c];  grid (ceil(Kh/16), ceil(3N/64), ceil(V/VCHUNK)):
// the long inner dimension (V) is split over workgroups, partial tiles are added with float
// atomics into the zeroed output (the output is tiny: K_h x 3N).
constexpr int VCHUNK = 64;
__global__ __launch_bounds__(256) void k_deform_grad_delta(const float* __restrict__ P,
                                                           const float* __restrict__ g, int N, int V,
                                                           int Kh, float* __restrict__ gdelta) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int k0 = blockIdx.x * 16, j0 = (blockIdx.y * 4 + wv) * 16;
  const int J = 3 * N;
  if (j0 >= J) return;
  const int vb = blockIdx.z * VCHUNK, ve = min(vb + VCHUNK, V);
  const f32x4 acc = mfma_tile_16x16(
      lane, vb, ve,
      [&](int i, int v) { const int k = k0 + i; return (k < Kh && v < ve) ? P[(size_t)v * Kh + k] : 0.f; },
      [&](int v, int jj) {
        const int j = j0 + jj;
        if (j >= J || v >= ve) return 0.f;
        const int n = j / 3, c = j - 3 * n;
        return g[((size_t)n * V + v) * 3 + c];
      });
  const int j = j0 + (lane & 15);
  if (j >= J) return;
  const int n = j / 3, c = j - 3 * n;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = k0 + (lane >> 4) * 4 + r;
    if (k < Kh) atomicAdd(&gdelta[((size_t)n * Kh + k) * 3 + c], acc[r]);
  }
}

// grad_P[v,k] = sum_{n,c} g[n,v,c] delta[n,k,c];  grid (ceil(V/16), ceil(Kh/64))
__global__ __launch_bounds__(256) void k_deform_grad_P(const float* __restrict__ g,
                                                       const float* __restrict__ delta, int N, int V,
                                                       int Kh, float* __restrict__ gP) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int v0 = blockIdx.x * 16, k0 = (blockIdx.y * 4 + wv) * 16;
  if (k0 >= Kh) return;
  const int J = 3 * N;
  const f32x4 acc = mfma_tile_16x16(
      lane, 0, J,
      [&](int i, int j) {
        const int v = v0 + i;
        if (v >= V || j >= J) return 0.f;
        const int n = j / 3, c = j - 3 * n;
        return g[((size_t)n * V + v) * 3 + c];
      },
      [&](int j, int kk) {
        const int k = k0 + kk;
        if (k >= Kh || j >= J) return 0.f;
        const int n = j / 3, c = j - 3 * n;
        return delta[((size_t)n * Kh + k) * 3 + c];
      });
  const int k = k0 + (lane & 15);
  if (k >= Kh) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int v = v0 + (lane >> 4) * 4 + r;
    if (v < V) gP[(size_t)v * Kh + k] = acc[r];
  }
}

// grad_mean[v,c] = sum_n g[n,v,c]: frames split over blockIdx.y (8 per block), partial sums added
// atomically into the zeroed output
constexpr int MEAN_FRAMES = 8;
__global__ void k_deform_grad_mean(const float* __restrict__ g, int N, int V3, float* __restrict__ gmean) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V3) return;
  const int n0 = blockIdx.y * MEAN_FRAMES, n1 = min(n0 + MEAN_FRAMES, N);
  float s = 0.f;
  for (int n = n0; n < n1; ++n) s += g[(size_t)n * V3 + i];
  atomicAdd(&gmean[i], s);
}

// grad_delta and grad_mean in ONE launch, without atomics or zero fills (they were two kernels and
// two zero-fill launches: 21 us of launch latency for a few hundred KB of work):
//   blocks [0, TK*TJ): one 16x16 tile of grad_delta each; the workgroup's four waves take a quarter of
//     the inner dimension (V) each and the partial tiles are summed through LDS;
//   blocks [TK*TJ, ..): grad_mean, one thread per (v, c), four partial sums over the frames in flight.
constexpr int DM_WAVES = 16;   // waves per workgroup of k_deform_bwd_dm
__global__ __launch_bounds__(64 * DM_WAVES) void k_deform_bwd_dm(const float* __restrict__ P,
                                                                const float* __restrict__ g, int N, int V,
                                                                int Kh, int TK, int TJ,
                                                                float* __restrict__ gdelta,
                                                                float* __restrict__ gmean) {
  __shared__ float s_part[DM_WAVES][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int J = 3 * N;
  if ((int)blockIdx.x < TK * TJ) {
    if (!gdelta) return;
    const int k0 = ((int)blockIdx.x % TK) * 16, j0 = ((int)blockIdx.x / TK) * 16;
    const int q = ((V + DM_WAVES - 1) / DM_WAVES + 3) / 4 * 4;   // this wave's share of V, multiple of the MFMA's 4
    const int vb = wv * q, ve = min(vb + q, V);
    // operands of eight MFMA steps are loaded before the first of them is issued
    const int i = lane & 15, kk = lane >> 4;
    const int k = k0 + i, j = j0 + i;
    const int jn = j / 3, jc = j - 3 * jn;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int v0 = vb; v0 < ve; v0 += 32) {
      float av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int v = v0 + 4 * u + kk;
        av[u] = (k < Kh && v < ve) ? P[(size_t)v * Kh + k] : 0.f;
        bv[u] = (j < J && v < ve) ? g[((size_t)jn * V + v) * 3 + jc] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) s_part[wv][lane * 4 + r] = acc[r];
    __syncthreads();
    if (wv != 0) return;
    const int jo = j0 + (lane & 15);
    if (jo >= J) return;
    const int n = jo / 3, c = jo - 3 * n;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ko = k0 + (lane >> 4) * 4 + r;
      const int e = lane * 4 + r;
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < DM_WAVES; ++w) sum += s_part[w][e];
      if (ko < Kh) gdelta[((size_t)n * Kh + ko) * 3 + c] = sum;
    }
    return;
  }
  if (!gmean) return;
  const int V3 = 3 * V;
  const int i = ((int)blockIdx.x - TK * TJ) * 64 * DM_WAVES + (int)threadIdx.x;
  if (i >= V3) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int n = 0;
  for (; n + 3 < N; n += 4) {
    s0 += g[(size_t)n * V3 + i]; s1 += g[(size_t)(n + 1) * V3 + i];
    s2 += g[(size_t)(n + 2) * V3 + i]; s3 += g[(size_t)(n + 3) * V3 + i];
  }
  for (; n < N; ++n) s0 += g[(size_t)n * V3 + i];
  gmean[i] = (s0 + s1) + (s2 + s3);
}

// G = sum_n g_n delta_n^T accumulated in DOUBLE (and sum_n g_n): the pre-solve sums that a sharded step exchanges.
// d lbs = solve_backward(G) amplifies the rounding of G by the conditioning of the deformation system (~1e5), and a
// float32 sum depends on how the frames are split over ranks; a double sum does not at float32's resolution, so the
// exchanged buffer holds doubles and G is rounded to float32 ONCE, after the all-reduce -- the sharded and the
// single-process step then hand the same bits to the solve's backward.  No atomics, fixed order.
// One wave per (vertex, quarter of the handles): lane j takes the frames j, j + 64, ..., holds its g_n[v] in registers,
// and the wave sums the 64 partial products of every handle of its quarter with a butterfly in double.
constexpr int GP64_KQ = 4;   // handles per wave = ceil(K_h / GP64_KQ)
__global__ __launch_bounds__(256) void k_deform_grad_P_f64(const float* __restrict__ g, const float* __restrict__ delta,
                                                          int N, int V, int Kh, double* __restrict__ G64,
                                                          float* __restrict__ G32, double* __restrict__ m64) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int v = wave / GP64_KQ, q = wave % GP64_KQ;
  if (v >= V) return;                                   // (whole wave)
  const int kq = (Kh + GP64_KQ - 1) / GP64_KQ, k0 = q * kq, k1 = min(Kh, k0 + kq);
  if (m64 && q == GP64_KQ - 1) {                        // sum_n g_n[v] as well (the last quarter has the fewest handles)
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int n = lane; n < N; n += 64) {
      const float* gv = g + ((size_t)n * V + v) * 3;
      a0 += (double)gv[0]; a1 += (double)gv[1]; a2 += (double)gv[2];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { a0 += __shfl_xor(a0, m, 64); a1 += __shfl_xor(a1, m, 64); a2 += __shfl_xor(a2, m, 64); }
    if (lane == 0) { m64[(size_t)v * 3] = a0; m64[(size_t)v * 3 + 1] = a1; m64[(size_t)v * 3 + 2] = a2; }
  }
  for (int k = k0; k < k1; ++k) {
    double a = 0.0;
    for (int n = lane; n < N; n += 64) {
      const float* gv = g + ((size_t)n * V + v) * 3;
      const float* dk = delta + ((size_t)n * Kh + k) * 3;
      a = fma((double)gv[0], (double)dk[0], a); a = fma((double)gv[1], (double)dk[1], a); a = fma((double)gv[2], (double)dk[2], a);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
    if (lane == 0) {
      G64[(size_t)v * Kh + k] = a;
      if (G32) G32[(size_t)v * Kh + k] = (float)a;
    }
  }
}
}  // namespace acfm

using namespace acfm;

extern "C" {

int acfm_deform_presolve_sums_f64(const float* delta, const float* grad_verts, int N, int V, int Kh, double* G64,
                                  double* mean64, float* grad_P, void* stream) {
  if (!delta || !grad_verts || !G64 || N <= 0 || V <= 0 || Kh <= 0 || N > 1000000 || (size_t)V * Kh > 0x7fffffffull)
    return ACFM_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(ACFM_PROF_DEFORM_BWD, st);
  hipLaunchKernelGGL(k_deform_grad_P_f64, dim3((unsigned)(((size_t)V * GP64_KQ + 3) / 4)), dim3(256), 0, st, grad_verts,
                     delta, N, V, Kh, G64, grad_P, mean64);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_deform_apply(const float* mean_v, const float* P, const float* delta, int N, int V, int Kh,
                      float* verts, void* stream) {
  if (!mean_v || !P || !delta || !verts || N <= 0 || V <= 0 || Kh <= 0 || N > 1000000) return ACFM_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(ACFM_PROF_DEFORM, st);
  hipLaunchKernelGGL(k_deform_apply, dim3((V + 15) / 16, (3 * N + 63) / 64), dim3(256), 0, st, mean_v, P,
                     delta, N, V, Kh, verts);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_deform_apply_backward(const float* P, const float* delta, const float* grad_verts, int N, int V,
                               int Kh, float* grad_delta, float* grad_mean, float* grad_P, void* stream) {
  if (!P || !delta || !grad_verts || N <= 0 || V <= 0 || Kh <= 0 || N > 1000000) return ACFM_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(ACFM_PROF_DEFORM_BWD, st);
  const int TK = (Kh + 15) / 16, TJ = (3 * N + 15) / 16;
  const int DMT = 64 * DM_WAVES;
  if ((grad_delta || grad_mean) && (size_t)TK * TJ + (size_t)(3 * V + DMT - 1) / DMT <= 0x7fffffffull) {
    hipLaunchKernelGGL(k_deform_bwd_dm, dim3((unsigned)(TK * TJ + (3 * V + DMT - 1) / DMT)), dim3(DMT), 0, st, P,
                       grad_verts, N, V, Kh, TK, TJ, grad_delta, grad_mean);
  } else {
    if (grad_delta) {
      if (zero_async(grad_delta, sizeof(float) * 3 * (size_t)N * Kh, st) != ACFM_OK) return ACFM_E_LAUNCH;
      hipLaunchKernelGGL(k_deform_grad_delta, dim3((Kh + 15) / 16, (3 * N + 63) / 64, (V + VCHUNK - 1) / VCHUNK),
                         dim3(256), 0, st, P, grad_verts, N, V, Kh, grad_delta);
    }
    if (grad_mean) {
      if (zero_async(grad_mean, sizeof(float) * 3 * (size_t)V, st) != ACFM_OK) return ACFM_E_LAUNCH;
      hipLaunchKernelGGL(k_deform_grad_mean, dim3((3 * V + 255) / 256, (N + MEAN_FRAMES - 1) / MEAN_FRAMES),
                         dim3(256), 0, st, grad_verts, N, 3 * V, grad_mean);
    }
  }
  if (grad_P)
    hipLaunchKernelGGL(k_deform_grad_P, dim3((V + 15) / 16, (Kh + 63) / 64), dim3(256), 0, st, grad_verts,
                       delta, N, V, Kh, grad_P);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

}  // extern "C"
