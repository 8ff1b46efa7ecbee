// Template deformation solve: P = (L^T L + A^T A)^-1 A^T and its backward, fp64 on the matrix
// cores.
//
// The reference builds M = L^T L + A^T A (A = softmax(lbs, dim 0)^T, L = cotangent Laplacian of
// the current mean shape, no grad) for every frame and calls torch.cholesky + cholesky_solve
// (multiframe/main.py:586-609).  Both the mean shape and lbs are learned, so M changes every
// optimiser step: one V x V SPD factorisation per step is part of the training hot path.  This
// file does it once per step for all frames (deform.py) as a blocked Cholesky with 32 x 32 tiles,
// fp64 throughout (v_mfma_f64_16x16x4_f64), in three launches:
//
//   k_solve_softmax   A = softmax over vertices of each handle's logits (fp64), one WG per handle; the workgroups
//                     beyond the 32 handles fill the sentinels of the tiles the factorisation publishes
//                     (solve_prepare_block) and reset the status word and the job ticket
//   k_solve_gram_rows W = L^T L + A^T A, one WG per row; only the non-zeros of L's column are
//                     visited (the cotangent Laplacian has ~7 per column), fixed summation order
//   k_chol_tiles      ONE launch for the factorisation and both substitutions: every tile of the
//                     factor is a job (acc = W_ij - sum_k L_ik L_jk^T as the L tiles appear, then
//                     L_ij = acc L_jj^-T; the diagonal job factorises and inverts its tile with the
//                     four waves of the workgroup, potrf32_wg).  Two more groups of tile rows are
//                     jobs like the others: the right-hand sides A (so Y^T = A L^-T is free) and an
//                     identity (so R = L^-T is free); the last jobs form P = R Y.  Jobs are dealt by
//                     a ticket in an order in which a job only waits for lower tickets, and tiles are
//                     handed from job to job as self-validating 8-byte words (see k_chol_tiles).
// backward (dP -> dlbs):  Q = M^-1 dP = R (R^T dP)  (k_apply_R twice: with R explicit the
//   substitutions are tile GEMMs, one WG of 16 waves per 32 vertices, no serial chain),
//   dA = Q^T - (A Q) P^T - (A P) Q^T,   dlbs = softmax backward per handle     (k_solve_bwd_lbs)
//
// -DACFM_CHOL_STEPS=1 (make VARIANT=steps EXTRA=-DACFM_CHOL_STEPS=1) builds the round-1 schedule instead
// -- one launch per tile column (k_chol_first, k_chol_step x nblk, k_apply_R) -- for A/B runs; both give
// bit-identical results (same operations in the same order).
//
// Storage: (2 nblk + 1) x nblk tiles, row-major, ld = n_pad = 32 nblk; tile rows [0, nblk) the
// matrix, tile row nblk the right-hand sides (row h = handle h), tile rows (nblk, 2 nblk] the
// identity / R.  Indices in [n, n_pad) are padded with the identity.
#include "acfm_common.h"

#include <atomic>

#pragma clang fp contract(fast)  // fp64 solve: fused multiply-adds are welcome here

namespace acfm {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int NB = 32;        // tile edge
constexpr int LDT = NB + 2;   // padded LDS row stride (doubles); even: rows stay 16-byte aligned
constexpr int KHP = 32;       // max handles = one tile row

struct SolveWs {
  double* W;     // [(2 nblk+1)*32, n_pad] working matrix (trailing updates in place)
  double* Lf;    // same shape: the factor L; tile row nblk = Y^T = A L^-T; tile rows above = R = L^-T
  double* Linv;  // [nblk, 32, 32] inverses of the diagonal factor tiles
  double* A64;   // [32, n_pad] handle weights
  double* X;     // [n_pad, 32] P in fp64
  double* Z;     // [n_pad, 32] backward: R^T dP
  double* Q;     // [n_pad, 32] backward: M^-1 dP
  int* info;     // 0, or 1 + index of the first non-positive pivot
  int n, n_pad, nblk, ld;
  size_t bytes;
};

static inline SolveWs carve_solve(void* base, int V) {
  SolveWs s;
  s.n = V;
  s.nblk = (V + NB - 1) / NB;
  s.n_pad = s.nblk * NB;
  s.ld = s.n_pad;
  char* p = (char*)base;
  size_t o = 0;
  const size_t mat = sizeof(double) * (size_t)(2 * s.n_pad + NB) * s.ld;
  s.W = (double*)(p + o);    o += align256(mat);
  s.Lf = (double*)(p + o);   o += align256(mat);
  s.Linv = (double*)(p + o); o += align256(sizeof(double) * (size_t)s.nblk * NB * NB);
  s.A64 = (double*)(p + o);  o += align256(sizeof(double) * (size_t)KHP * s.n_pad);
  s.X = (double*)(p + o);    o += align256(sizeof(double) * (size_t)s.n_pad * KHP);
  s.Z = (double*)(p + o);    o += align256(sizeof(double) * (size_t)s.n_pad * KHP);
  s.Q = (double*)(p + o);    o += align256(sizeof(double) * (size_t)s.n_pad * KHP);
  s.info = (int*)(p + o);    o += 256;
  s.bytes = o;
  return s;
}

__device__ __forceinline__ f64x4 mfma64(double a, double b, f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// v_mfma_f64_16x16x4_f64 result element e of lane l sits at row (l>>4) + 4e, column l&15
__device__ __forceinline__ int acc_row(int lane, int e) { return (lane >> 4) + 4 * e; }

__device__ __forceinline__ double bcast_lane(double x, int src) {  // src: compile-time lane
  int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
  return __hiloint2double(hi, lo);
}

template <class OP>
__device__ __forceinline__ double block_reduce(double v, double* scratch /*[256]*/, OP op) {
  const int t = threadIdx.x;
  scratch[t] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) scratch[t] = op(scratch[t], scratch[t + s]);
    __syncthreads();
  }
  const double r = scratch[0];
  __syncthreads();
  return r;
}

// C (32x32, quadrant of this wave) += sum_p Aop(r, p) * Bop(p, c);  fa(r, p), fb(p, c) with
// r, c in [0,16) relative to the wave's quadrant
template <class FA, class FB>
__device__ __forceinline__ f64x4 tile_mma(int lane, f64x4 acc, FA&& fa, FB&& fb) {
  const int x = lane & 15, y = lane >> 4;
#pragma unroll
  for (int k0 = 0; k0 < NB; k0 += 4) acc = mfma64(fa(x, k0 + y), fb(k0 + y, x), acc);
  return acc;
}

// The workgroup (4 waves) factorises the SPD tile sC[NB][LDT] (lower part read) and inverts the factor.
// On return sU[j][r] = L_rj for r >= j (row j of L^T; the entries r < j are unspecified: use potrf_L) and
// sU[j][32..63] = row j of L^-1; sU ([NB][64], 16 KB) may overlay sC and what follows it.
// Lanes 0-31 of every wave: lane r owns row r of L; lanes 32-63: lane 32+c owns column c of L^-1
// (forward substitution).  Both are the recurrence  u_j = (c_j - sum_{p<j} u_p L_jp) / L_jj  with the
// same wave-uniform L_jp, so one instruction stream serves both halves.  A single wave doing all of it
// is bound by instruction issue (about 75 instructions per pivot, most of them the updates
// c_q -= u_j L_qj of far columns), so the COLUMNS are dealt to the waves: wave w keeps the start
// values of columns 8w..8w+7 and runs their eight pivots -- per pivot only the serial chain (pivot
// broadcast, 1/sqrt by v_rsq_f64 and one second-order correction, no IEEE divide / sqrt sequences)
// and the updates inside its panel, with L_qj taken from lane q's register (v_readlane) -- publishes
// each u_j in sU and bumps a counter in LDS.  Before its own panel a wave follows that counter and
// applies the pivots of the earlier panels to its eight columns (one 64-lane read of u_j, L_qj as
// broadcast ds_read_b128), off everybody else's chain.
__device__ __forceinline__ double bcast_lane_dyn(double x, int src) {  // src: wave-uniform
  int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
  return __hiloint2double(hi, lo);
}

constexpr int PW = 8;  // columns per wave
__device__ __forceinline__ double potrf_L(const double* sU, int rr, int cc) { return cc <= rr ? sU[cc * 64 + rr] : 0.0; }
#ifdef ACFM_DIAG
__device__ long long g_potrf_stamps[16];
#define POTRF_STAMP(slot) do { if (lane == 0) g_potrf_stamps[4 * w + (slot)] = (long long)wall_clock64(); } while (0)
#else
#define POTRF_STAMP(slot) do {} while (0)
#endif
__device__ __forceinline__ void potrf32_wg(const double* sC, double* sU, int* sCount, int t, int base_index, int* info) {
  const int lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = lane & 31;
  const bool low = lane < 32;
  double c[PW];
#pragma unroll
  for (int q = 0; q < PW; ++q) {
    const double cj = sC[r * LDT + PW * w + q];
    c[q] = low ? cj : (r == PW * w + q ? 1.0 : 0.0);
  }
  if (t == 0) *sCount = 0;
  __syncthreads();  // start values are in registers: sU may overwrite sC from here on
  POTRF_STAMP(0);
  // volatile LDS accesses: program order is kept and the LDS executes a wave's accesses in order, which is all
  // the publish (u_j, then the counter) and the follow (counter, then u_j) need; the explicit address space keeps
  // them ds_ instructions (a volatile generic pointer becomes a flat system-scope access)
  typedef __attribute__((address_space(3))) volatile int lds_vint;
  typedef __attribute__((address_space(3))) volatile double lds_vdouble;
  lds_vint* count = (lds_vint*)sCount;
  lds_vdouble* U = (lds_vdouble*)sU;
  // ---- the pivots of the earlier panels, as they appear: every poll reads the counter together with the row
  // of the next pivot, so a published pivot costs one LDS round trip, not two.  (Requesting the row after it as
  // well, so that a follower with a backlog catches up faster, measured 4 us SLOWER over the 21 tiles: the
  // followers' extra LDS traffic delays the leader; polling the counter alone and then taking the published pivots
  // two at a time, 2.7 us slower; polling lightly until the wave is next, 12 us slower.)
  for (int done = 0; done < PW * w;) {
    const int avail = *count;
    const double u = U[done * 64 + lane];
    double l[PW];
#pragma unroll
    for (int q = 0; q < PW; ++q) l[q] = U[done * 64 + PW * w + q];
    if (avail <= done) continue;
#pragma unroll
    for (int q = 0; q < PW; ++q) c[q] -= u * l[q];
    ++done;
  }
  // ---- this wave's panel.  Nothing but the chain and the panel's own updates in the loop: a lone wave issues an
  // instruction every 2-4 ns, so rows above the diagonal are stored as they come (the readers mask them) and the
  // pivot check is made afterwards from the stored diagonal.
  POTRF_STAMP(1);
  lds_vdouble* Uw = U + PW * w * 64 + lane;
  lds_vint* cw = count;
#pragma unroll
  for (int jj = 0; jj < PW; ++jj) {
    const int j = PW * w + jj;
    const double sum = c[jj];
    const double piv = bcast_lane_dyn(sum, j);
    // y = piv^-1/2: y0 = rsq, e = 1/2 - piv/2 y0^2, y = y0 (1 + e (1 + 3/2 e)) + O(e^3); the lane's value is
    // sum * y (lane j: piv * y = the diagonal entry)
    const double y0 = __builtin_amdgcn_rsq(piv);
    const double s0 = sum * y0;
    const double e = 0.5 - (0.5 * piv) * y0 * y0;
    const double h = e * (1.0 + 1.5 * e);
    const double v = s0 + s0 * h;
#pragma unroll
    for (int q = jj + 1; q < PW; ++q) c[q] -= v * bcast_lane_dyn(v, PW * w + q);
    Uw[jj * 64] = v;
    *cw = j + 1;  // every lane stores the same word: LDS executes a wave's accesses in order
  }
  POTRF_STAMP(2);
  __syncthreads();
  if (w == 0) {
    // lane j < 32: the diagonal entry piv_j^(1/2) is positive and finite unless pivot j (or one before it) was not
    // positive
    const double d = sU[r * 64 + r];
    const unsigned long long badmask = __ballot(low && !(d > 0.0 && d < 1.0e300));
    if (badmask && lane == 0) atomicMax(info, base_index + __ffsll((long long)badmask));
  }
  POTRF_STAMP(3);
}

// ---- A = softmax(lbs[:, h]) over the vertices, fp64; grid = 32 (rows >= Kh are zero) ----------
// grid = 32, or 32 + solve_prepare_blocks: the workgroups beyond the 32 handles fill the sentinels of the tile launch
// (neither half depends on the other, both precede the rows of W: one launch instead of two)
__device__ __forceinline__ void solve_prepare_block(const SolveWs& s, int b_in);
__global__ __launch_bounds__(256) void k_solve_softmax(const float* __restrict__ lbs, SolveWs s, int Kh) {
  __shared__ double scratch[256];
  if (blockIdx.x >= KHP) {
    solve_prepare_block(s, (int)blockIdx.x - KHP);
    return;
  }
  const int h = blockIdx.x, t = threadIdx.x;
  double* rowW = s.W + (size_t)(s.n_pad + h) * s.ld;
  double* rowA = s.A64 + (size_t)h * s.n_pad;
  if (h == 0 && t == 0) { s.info[0] = 0; s.info[1] = 0; }  // status, tile ticket of k_chol_tiles
#if ACFM_CHOL_STEPS
  // identity under the right-hand sides (the region was zeroed by the host): R = L^-T rides along
  for (int v = h * 256 + t; v < s.n_pad; v += KHP * 256) s.W[(size_t)(s.n_pad + NB + v) * s.ld + v] = 1.0;
#endif
  if (h >= Kh) {
    for (int v = t; v < s.n_pad; v += 256) rowW[v] = rowA[v] = 0.0;
    return;
  }
  double m = -1e300;
  for (int v = t; v < s.n; v += 256) m = fmax(m, (double)lbs[(size_t)v * Kh + h]);
  m = block_reduce(m, scratch, [](double a, double b) { return fmax(a, b); });
  double sum = 0.0;
  for (int v = t; v < s.n; v += 256) sum += exp((double)lbs[(size_t)v * Kh + h] - m);
  sum = block_reduce(sum, scratch, [](double a, double b) { return a + b; });
  for (int v = t; v < s.n_pad; v += 256)
    rowW[v] = rowA[v] = v < s.n ? exp((double)lbs[(size_t)v * Kh + h] - m) / sum : 0.0;
}

// ---- W = L^T L + A^T A, one WG per row i (all columns written: the matrix is symmetric) ---------
// W[i][c] = sum_h A[h][i] A[h][c] + sum_r L[r][i] L[r][c]: the WG scans column i of L, keeps the
// non-zero (r, L[r][i]) in ascending r, and adds those rows of L; each output element is summed
// by one thread in a fixed order (h ascending, then r ascending).
// The work is a few hundred flops per thread behind load latencies, so loads travel in batches: the column scan
// asks for GRAM_RPT rows per thread at once (1024 rows per round), and the rows of L that are added are fetched
// GRAM_EB at a time before their terms are summed in order.
constexpr int GRAM_CPT = 4;   // columns per thread and pass
constexpr int GRAM_RPT = 4;   // rows of the column scan per thread and round
constexpr int GRAM_EB = 8;    // rows of L in flight while summing
constexpr int GRAM_HB = 8;    // handles of A in flight while summing
__global__ __launch_bounds__(256) void k_solve_gram_rows(const float* __restrict__ L, SolveWs s, int Kh) {
  const int i = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int n = s.n;
  double* wrow = s.W + (size_t)i * s.ld;
  if (i >= n) {
    for (int c = t; c < s.n_pad; c += 256) wrow[c] = (c == i) ? 1.0 : 0.0;
    return;
  }
  __shared__ int s_r[256 * GRAM_RPT];
  __shared__ double s_v[256 * GRAM_RPT];
  __shared__ int s_wcnt[GRAM_RPT][4];
  for (int cb = 0; cb < s.n_pad; cb += 256 * GRAM_CPT) {
    double acc[GRAM_CPT];
    int col[GRAM_CPT];
#pragma unroll
    for (int q = 0; q < GRAM_CPT; ++q) {
      col[q] = cb + t + 256 * q;
      acc[q] = 0.0;
    }
    // the first round of the column scan is asked for before the A^T A terms are summed: its latency hides behind them
    float val[GRAM_RPT];
#pragma unroll
    for (int k = 0; k < GRAM_RPT; ++k) {
      const int r = 256 * k + t;
      val[k] = r < n ? L[(size_t)r * n + i] : 0.f;
    }
    for (int h0 = 0; h0 < Kh; h0 += GRAM_HB) {  // GRAM_HB handles' loads in flight, summed in order
      double ai[GRAM_HB], ac[GRAM_HB][GRAM_CPT];
#pragma unroll
      for (int e = 0; e < GRAM_HB; ++e) {
        const int h = h0 + e < Kh ? h0 + e : h0;
        ai[e] = s.A64[(size_t)h * s.n_pad + i];
#pragma unroll
        for (int q = 0; q < GRAM_CPT; ++q) ac[e][q] = col[q] < n ? s.A64[(size_t)h * s.n_pad + col[q]] : 0.0;
      }
#pragma unroll
      for (int e = 0; e < GRAM_HB; ++e) {
        if (h0 + e >= Kh) break;
#pragma unroll
        for (int q = 0; q < GRAM_CPT; ++q)
          if (col[q] < n) acc[q] += ai[e] * ac[e][q];
      }
    }
    for (int r0 = 0; r0 < n; r0 += 256 * GRAM_RPT) {
      if (r0 > 0) {
#pragma unroll
        for (int k = 0; k < GRAM_RPT; ++k) {
          const int r = r0 + 256 * k + t;
          val[k] = r < n ? L[(size_t)r * n + i] : 0.f;
        }
      }
      unsigned long long bal[GRAM_RPT];
#pragma unroll
      for (int k = 0; k < GRAM_RPT; ++k) {
        bal[k] = __ballot(val[k] != 0.f);
        if (lane == 0) s_wcnt[k][w] = __popcll(bal[k]);
      }
      __syncthreads();
      int cnt = 0;
#pragma unroll
      for (int k = 0; k < GRAM_RPT; ++k) {  // ascending r: sub-chunk k, then wave, then lane
        int off = cnt + __popcll(bal[k] & ((1ull << lane) - 1ull));
        for (int ww = 0; ww < w; ++ww) off += s_wcnt[k][ww];
        if (val[k] != 0.f) { s_r[off] = r0 + 256 * k + t; s_v[off] = (double)val[k]; }
        cnt += s_wcnt[k][0] + s_wcnt[k][1] + s_wcnt[k][2] + s_wcnt[k][3];
      }
      __syncthreads();
      for (int e0 = 0; e0 < cnt; e0 += GRAM_EB) {
        float lv[GRAM_EB][GRAM_CPT];
#pragma unroll
        for (int e = 0; e < GRAM_EB; ++e) {
          const float* lrow = L + (size_t)s_r[e0 + e < cnt ? e0 + e : e0] * n;
#pragma unroll
          for (int q = 0; q < GRAM_CPT; ++q) lv[e][q] = col[q] < n ? lrow[col[q]] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < GRAM_EB; ++e) {
          if (e0 + e >= cnt) break;
          const double v = s_v[e0 + e];
#pragma unroll
          for (int q = 0; q < GRAM_CPT; ++q)
            if (col[q] < n) acc[q] += v * (double)lv[e][q];
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < GRAM_CPT; ++q)
      if (col[q] < s.n_pad) wrow[col[q]] = acc[q];
  }
}

// ---- factorise tile (0,0); one WG ----------------------------------------------------------------
__global__ __launch_bounds__(256) void k_chol_first(SolveWs s) {
  __shared__ __attribute__((aligned(16))) double sBuf[2 * NB * LDT];
  __shared__ int sCount;
  const int t = threadIdx.x;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = t + 256 * e, rr = idx >> 5, cc = idx & 31;
    sBuf[rr * LDT + cc] = s.W[(size_t)rr * s.ld + cc];
  }
  __syncthreads();
  potrf32_wg(sBuf, sBuf, &sCount, t, 0, s.info);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = t + 256 * e, rr = idx >> 5, cc = idx & 31;
    s.Lf[(size_t)rr * s.ld + cc] = potrf_L(sBuf, rr, cc);
    s.Linv[rr * NB + cc] = sBuf[rr * 64 + 32 + cc];
  }
}

// ---- one tile column of the factorisation; grid (m+1, nblk+1), m = nblk-1-k --------------------
// blockIdx.y -> tile row i: y < m the matrix rows k+1+y, y == m the right-hand sides (i = nblk),
// y > m the identity rows 0..k (i = nblk+1+(y-m-1); rows below k are still zero in column k).
// blockIdx.x < m -> update of tile (i, j = k+1+x), blockIdx.x == m -> store L_ik.
__global__ __launch_bounds__(256) void k_chol_step(SolveWs s, int k) {
  const int m = s.nblk - 1 - k;
  const int y = blockIdx.y;
  const int i = y < m ? k + 1 + y : s.nblk + (y - m);
  const bool panel = (int)blockIdx.x == m;
  const int j = panel ? i : k + 1 + (int)blockIdx.x;
  if (!panel && j > i) return;
  __shared__ __attribute__((aligned(16))) double sInv[NB][LDT], sWW[2][NB][LDT], sLi[NB][LDT], sLj[NB][LDT];
  __shared__ int sCount;
  double(*sWi)[LDT] = sWW[0];
  double(*sWj)[LDT] = sWW[1];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int qi = w >> 1, qj = w & 1;
  const double* inv = s.Linv + (size_t)k * NB * NB;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = t + 256 * e, rr = idx >> 5, cc = idx & 31;
    sInv[rr][cc] = inv[rr * NB + cc];
    sWi[rr][cc] = s.W[(size_t)(NB * i + rr) * s.ld + NB * k + cc];
    if (!panel && j != i) sWj[rr][cc] = s.W[(size_t)(NB * j + rr) * s.ld + NB * k + cc];
  }
  __syncthreads();
  const f64x4 zero = {0.0, 0.0, 0.0, 0.0};
  const int col = 16 * qj + (lane & 15);
  // L_ik = W_ik L_kk^-T
  f64x4 li = tile_mma(lane, zero, [&](int r, int p) { return sWi[16 * qi + r][p]; },
                      [&](int p, int c) { return sInv[16 * qj + c][p]; });
  if (panel) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      s.Lf[(size_t)(NB * i + 16 * qi + acc_row(lane, e)) * s.ld + NB * k + col] = li[e];
    return;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) sLi[16 * qi + acc_row(lane, e)][col] = li[e];
  if (j != i) {
    f64x4 lj = tile_mma(lane, zero, [&](int r, int p) { return sWj[16 * qi + r][p]; },
                        [&](int p, int c) { return sInv[16 * qj + c][p]; });
#pragma unroll
    for (int e = 0; e < 4; ++e) sLj[16 * qi + acc_row(lane, e)][col] = lj[e];
  }
  __syncthreads();
  double(*sR)[LDT] = (j != i) ? sLj : sLi;
  f64x4 acc;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    acc[e] = s.W[(size_t)(NB * i + 16 * qi + acc_row(lane, e)) * s.ld + NB * j + col];
  acc = tile_mma(lane, acc, [&](int r, int p) { return -sLi[16 * qi + r][p]; },
                 [&](int p, int c) { return sR[16 * qj + c][p]; });
  const bool next_diag = (i == k + 1) && (j == k + 1);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int row = 16 * qi + acc_row(lane, e);
    s.W[(size_t)(NB * i + row) * s.ld + NB * j + col] = acc[e];
    if (next_diag) sWi[row][col] = acc[e];
  }
  if (next_diag) {
    __syncthreads();
    double* sU = &sWW[0][0][0];
    potrf32_wg(sU, sU, &sCount, t, NB * i, s.info);
    double* invn = s.Linv + (size_t)i * NB * NB;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = t + 256 * e, rr = idx >> 5, cc = idx & 31;
      s.Lf[(size_t)(NB * i + rr) * s.ld + NB * i + cc] = potrf_L(sU, rr, cc);
      invn[rr * NB + cc] = sU[rr * 64 + 32 + cc];
    }
  }
}

// ---- the factorisation as ONE launch: tiles as dataflow ------------------------------------------
// Every tile (i, j) of the factor (matrix rows, the right-hand-side row, the identity rows that become
// R = L^-T) is one job:   acc = W_ij - sum_{k<j} L_ik L_jk^T   (k ascending, the order of the trailing
// updates of a right-looking sweep),  then  L_ij = acc L_jj^-T  (diagonal: factorise + invert).
// Jobs are handed out by a ticket in column-major order (column j: diagonal, matrix rows below it,
// right-hand sides, identity rows 0..j), so a job depends only on jobs with a lower ticket: whichever
// workgroup holds the lowest unfinished ticket can always finish, with any number of resident
// workgroups and any dispatch order.
//
// Hand-off without flags or fences: Lf and Linv are pre-filled with a sentinel (all-ones, a NaN no
// arithmetic produces); producers write every word of a finished tile with one agent-scope (sc1,
// write-through) 8-byte store, consumers read operands with agent-scope 8-byte loads straight into
// the MFMA operand registers and repeat until no lane holds the sentinel.  Each word validates itself,
// so no ordering between words is needed.  Off the critical path a wave first polls one word of the
// tile it waits for (one 8-byte request per poll instead of 8 KB).
//
// Critical path: the diagonal job of column c also carries the tile left of it, (c, c-1), so that after
// L_{c-1,c-1}^-1 arrives it needs no second hand-off:  L_{c,c-1} = acc2 Linv^T,  acc -= L_{c,c-1} L_{c,c-1}^T,
// factorise.  (The job of tile (c, c-1) computes the same tile for everybody else.)
// Spins are bounded PER WAIT (2^19 polls of >= 0.1 us: a wave starved by time-slicing or by side-stream kernels on
// its CU gets >= 50 ms for every single hand-off, not for the whole kernel): on expiry the job raises
// ACFM_SOLVE_INFO_HANDOFF in info and carries on with what it has (sentinel NaNs then reach P), so the grid always
// drains; the status word is how the host learns of it (acfm_deform_solve_info, or a non-blocking copy of the word at
// acfm_deform_solve_info_offset).
constexpr unsigned long long CHOL_SENTINEL = ~0ull;
constexpr int CHOL_E_HANDOFF = ACFM_SOLVE_INFO_HANDOFF;
constexpr int CHOL_SPIN_LIMIT = 1 << 19;

__device__ __forceinline__ unsigned long long ld_word(const double* p) {
  return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_word(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the 8 MFMA operand words of this lane from tile rows [row0 + x], columns 4 m + y
__device__ __forceinline__ bool ld_operand(const double* tile_row, double (&v)[8]) {
  bool bad = false;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const unsigned long long u = ld_word(tile_row + 4 * m);
    bad |= u == CHOL_SENTINEL;
    v[m] = __longlong_as_double((long long)u);
  }
  return bad;
}
// wave-uniform wait for one word of a tile; false on expiry (a fresh budget for every wait)
__device__ __forceinline__ bool gate(const double* word, int& budget) {
  budget = CHOL_SPIN_LIMIT;
  while (ld_word(word) == CHOL_SENTINEL) {
    if (--budget < 0) return false;
    __builtin_amdgcn_s_sleep(4);
  }
  return true;
}

__device__ __forceinline__ void solve_prepare_block(const SolveWs& s, int b_in) {
  // sentinel into every tile k_chol_tiles publishes (and the diagonal inverses); grid = (2 nblk + 1) nblk + nblk
  const int nb = s.nblk, tiles = (2 * nb + 1) * nb, b = b_in, t = threadIdx.x;
  ulonglong2 ones = {CHOL_SENTINEL, CHOL_SENTINEL};
  if (b >= tiles) {
    ulonglong2* d = reinterpret_cast<ulonglong2*>(s.Linv + (size_t)(b - tiles) * NB * NB);
    d[t] = ones; d[t + 256] = ones;
    return;
  }
  const int ti = b / nb, tj = b % nb;
  const bool used = ti < nb ? tj <= ti : (ti == nb || tj >= ti - nb - 1);
  if (!used) return;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int idx = t + 256 * e, rr = idx >> 4, cc = (idx & 15) * 2;
    *reinterpret_cast<ulonglong2*>(s.Lf + (size_t)(NB * ti + rr) * s.ld + NB * tj + cc) = ones;
  }
}

#ifdef ACFM_DIAG
#define CHOL_STAMP(slot) do { if (diag && t == 0) reinterpret_cast<long long*>(s.Z)[8 * j + (slot)] = (long long)wall_clock64(); } while (0)
#else
#define CHOL_STAMP(slot) do {} while (0)
#endif

// P = R Y as the last jobs of the same launch: job c = the 32 vertices of tile row c for all handles,
//   P[v][h] = sum_{p >= c} sum_q R[32c + v][32p + q] Y^T[h][32p + q],
// operands read (and awaited) word by word like every other hand-off; the two wave pairs take alternate p and
// meet in LDS.  Writes X (fp64, the backward's copy) and the fp32 result.
__device__ __forceinline__ void apply_job(const SolveWs& s, int c, int Kh, float* __restrict__ P, double* sRed /*[2][2][4][64]*/,
                                          int t, int& budget, bool& expired) {
  const int lane = t & 63, w = t >> 6, qj = w & 1, par = w >> 1, x = lane & 15, y = lane >> 4;
  const int nqi = Kh > 16 ? 2 : 1, nb = s.nblk, ld = s.ld;
  const double* rowR = s.Lf + (size_t)(NB * (nb + 1 + c) + 16 * qj + x) * ld + y;
  const double* rowY0 = s.Lf + (size_t)(NB * nb + x) * ld + y;
  const double* rowY1 = rowY0 + (size_t)16 * ld;
  f64x4 acc[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
  for (int p = c + par; p < nb; p += 2) {
    double a0[8], a1[8], bb[8];
    for (;;) {
      bool bad = ld_operand(rowR + NB * p, bb) | ld_operand(rowY0 + NB * p, a0);
      if (nqi > 1) bad |= ld_operand(rowY1 + NB * p, a1);
      if (!__any(bad) || expired) break;
      const bool ok = gate(s.Lf + (size_t)(NB * (nb + 1 + c)) * ld + NB * p, budget) &&
                      gate(s.Lf + (size_t)(NB * nb) * ld + NB * p, budget);
      if (!ok) expired = true;
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      acc[0] = mfma64(a0[m], bb[m], acc[0]);
      if (nqi > 1) acc[1] = mfma64(a1[m], bb[m], acc[1]);
    }
  }
  if (par == 1) {
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
      for (int e = 0; e < 4; ++e) sRed[((qi * 2 + qj) * 4 + e) * 64 + lane] = acc[qi][e];
  }
  __syncthreads();
  if (par == 0) {
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (qi >= nqi) continue;
        const int h = 16 * qi + acc_row(lane, e), v = NB * c + 16 * qj + x;
        const double r = acc[qi][e] + sRed[((qi * 2 + qj) * 4 + e) * 64 + lane];
        s.X[(size_t)v * KHP + h] = r;
        if (v < s.n && h < Kh) P[(size_t)v * Kh + h] = (float)r;
      }
  }
}

__global__ __launch_bounds__(256) void k_chol_tiles(SolveWs s, int Kh, float* __restrict__ P) {
  __shared__ __attribute__((aligned(16))) double sPQ[2][NB][LDT];
  __shared__ int s_ticket, sCount;
  double(*sP)[LDT] = sPQ[0];
  double(*sQ)[LDT] = sPQ[1];
  const int nb = s.nblk, ntiles = nb * (nb + 2), ld = s.ld;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int qi = w >> 1, qj = w & 1, x = lane & 15, y = lane >> 4;
  const int col = 16 * qj + x;
  const f64x4 zero = {0.0, 0.0, 0.0, 0.0};
  int budget = CHOL_SPIN_LIMIT;  // polls this wave may still spend waiting
  bool expired = false;
  for (;;) {
    if (t == 0) s_ticket = atomicAdd(s.info + 1, 1);
    __syncthreads();
    const int b = s_ticket;
    __syncthreads();
    if (b >= ntiles + nb) break;
    if (b >= ntiles) {
      apply_job(s, b - ntiles, Kh, P, &sPQ[0][0][0], t, budget, expired);
      if (expired && lane == 0) atomicOr(s.info, CHOL_E_HANDOFF);
      continue;
    }
    const int j = b / (nb + 2), yy = b % (nb + 2);
    const int i = yy < nb - j ? j + yy : (yy == nb - j ? nb : nb + 1 + (yy - (nb - j) - 1));
    const bool diag = i == j;
    const int k0 = i > nb ? i - nb - 1 : 0;  // identity row r: zero left of column r
    CHOL_STAMP(0);
    // ---- start value
    f64x4 acc, acc2 = zero;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = 16 * qi + acc_row(lane, e);
      if (i <= nb) {
        acc[e] = s.W[(size_t)(NB * i + row) * ld + NB * j + col];
        if (diag && j > 0) acc2[e] = s.W[(size_t)(NB * i + row) * ld + NB * (j - 1) + col];
      } else {
        acc[e] = (j == k0 && row == col) ? 1.0 : 0.0;
      }
    }
    // ---- acc -= L_ik L_jk^T, k ascending; the diagonal job stops one short and feeds (i, j-1) alongside
    const double* rowA = s.Lf + (size_t)(NB * i + 16 * qi + x) * ld + y;
    const double* rowB = s.Lf + (size_t)(NB * j + 16 * qj + x) * ld + y;
    const double* rowB2 = s.Lf + (size_t)(NB * (j > 0 ? j - 1 : 0) + 16 * qj + x) * ld + y;
    const int kend = diag ? j - 1 : j;
    for (int k = k0; k < kend; ++k) {
      double a[8], bb[8], b2[8];
      for (;;) {
        bool bad = ld_operand(rowA + NB * k, a);
        if (diag) {
          bad |= ld_operand(s.Lf + (size_t)(NB * i + 16 * qj + x) * ld + y + NB * k, bb);
          bad |= ld_operand(rowB2 + NB * k, b2);
        } else {
          bad |= ld_operand(rowB + NB * k, bb);
        }
        if (!__any(bad) || expired) break;
        // wait for a word of each tile, then read again
        const bool ok = gate(s.Lf + (size_t)(NB * i) * ld + NB * k, budget) &&
                        gate(s.Lf + (size_t)(NB * (diag ? j - 1 : j)) * ld + NB * k, budget);
        if (!ok) expired = true;
      }
#pragma unroll
      for (int m = 0; m < 8; ++m) acc = mfma64(-a[m], bb[m], acc);
      if (diag) {
#pragma unroll
        for (int m = 0; m < 8; ++m) acc2 = mfma64(-a[m], b2[m], acc2);
      }
    }
    // ---- finish
    CHOL_STAMP(1);
    const int jd = diag ? j - 1 : j;  // the diagonal inverse this job waits for
    if (jd >= 0) {
      double inv[8];
      const double* rowI = s.Linv + (size_t)jd * NB * NB + (size_t)(16 * qj + x) * NB + y;
      budget = CHOL_SPIN_LIMIT;
      for (;;) {
        const bool bad = ld_operand(rowI, inv);
        if (!__any(bad) || expired) break;
        if (diag) {  // critical path: poll with the operand loads themselves
          if (--budget < 0) expired = true;
          __builtin_amdgcn_s_sleep(1);
        } else if (!gate(s.Linv + (size_t)jd * NB * NB, budget)) {
          expired = true;
        }
      }
      CHOL_STAMP(2);
      const f64x4 src = diag ? acc2 : acc;
#pragma unroll
      for (int e = 0; e < 4; ++e) sP[16 * qi + acc_row(lane, e)][col] = src[e];
      __syncthreads();
      f64x4 li = zero;
#pragma unroll
      for (int m = 0; m < 8; ++m) li = mfma64(sP[16 * qi + x][4 * m + y], inv[m], li);
      if (!diag) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          st_word(s.Lf + (size_t)(NB * i + 16 * qi + acc_row(lane, e)) * ld + NB * j + col, li[e]);
      } else {
        // acc -= L_{j,j-1} L_{j,j-1}^T
#pragma unroll
        for (int e = 0; e < 4; ++e) sQ[16 * qi + acc_row(lane, e)][col] = li[e];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 8; ++m) acc = mfma64(-sQ[16 * qi + x][4 * m + y], sQ[16 * qj + x][4 * m + y], acc);
      }
    }
    if (diag) {
      __syncthreads();  // every wave is done with sP / sQ
#pragma unroll
      for (int e = 0; e < 4; ++e) sP[16 * qi + acc_row(lane, e)][col] = acc[e];
      __syncthreads();
      CHOL_STAMP(3);
      double* sU = &sPQ[0][0][0];
      potrf32_wg(sU, sU, &sCount, t, NB * i, s.info);
      CHOL_STAMP(4);
      double* invn = s.Linv + (size_t)i * NB * NB;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int idx = t + 256 * e, rr = idx >> 5, cc = idx & 31;
        st_word(invn + rr * NB + cc, sU[rr * 64 + 32 + cc]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int idx = t + 256 * e, rr = idx >> 5, cc = idx & 31;
        st_word(s.Lf + (size_t)(NB * i + rr) * ld + NB * i + cc, potrf_L(sU, rr, cc));
      }
      CHOL_STAMP(5);
    }
    if (expired && lane == 0) atomicOr(s.info, CHOL_E_HANDOFF);
  }
}

// ---- out = R rhs (TRANS: R^T rhs), R = L^-T upper block-triangular; grid = nblk ------------------
// WG c produces the 32 vertices of tile row c for all handles.  Right-hand sides and results are
// [n_pad][32] (vertex-major); RHS 0: rhs^T = Y^T in the factor's right-hand-side tile row,
// 1: an [n_pad][32] fp64 buffer, 2: an fp32 [V][Kh] tensor (the incoming gradient).
constexpr int APPLY_PAR = 8;  // waves sharing the tile sum of one (vertex tile, handle half)
template <bool TRANS, int RHS>
__global__ __launch_bounds__(64 * 2 * APPLY_PAR) void k_apply_R(SolveWs s, const double* __restrict__ rhs64,
                                                 const float* __restrict__ rhs32, int Kh,
                                                 double* __restrict__ out, float* __restrict__ out32) {
  __shared__ double sRed[APPLY_PAR - 1][2][2][4][64];
  const int c = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int qj = w & 1, par = w >> 1;
  const int nqi = Kh > 16 ? 2 : 1;
  const int x = lane & 15, y = lane >> 4;
  const int n = s.n, ld = s.ld, nblk = s.nblk;
  const double* R = s.Lf + (size_t)(s.n_pad + NB) * ld;
  f64x4 acc[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
  const int pb = (TRANS ? 0 : c) + par, pe = TRANS ? c + 1 : nblk;
  for (int p = pb; p < pe; p += APPLY_PAR) {
    double a[2][8], b[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int q = 4 * ks + y;
      b[ks] = TRANS ? R[(size_t)(NB * p + q) * ld + NB * c + 16 * qj + x]
                    : R[(size_t)(NB * c + 16 * qj + x) * ld + NB * p + q];
#pragma unroll
      for (int qi = 0; qi < 2; ++qi) {
        if (qi >= nqi) continue;
        const int h = 16 * qi + x, v = NB * p + q;
        if (RHS == 0) a[qi][ks] = s.Lf[(size_t)(s.n_pad + h) * ld + v];
        else if (RHS == 1) a[qi][ks] = rhs64[(size_t)v * KHP + h];
        else a[qi][ks] = (v < n && h < Kh) ? (double)rhs32[(size_t)v * Kh + h] : 0.0;
      }
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int qi = 0; qi < 2; ++qi)
        if (qi < nqi) acc[qi] = mfma64(a[qi][ks], b[ks], acc[qi]);
  }
  if (par > 0) {
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (qi < nqi) sRed[par - 1][qi][qj][e][lane] = acc[qi][e];
  }
  __syncthreads();
  if (par == 0) {
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (qi >= nqi) continue;
        const int h = 16 * qi + acc_row(lane, e), v = NB * c + 16 * qj + x;
        double r = acc[qi][e];
#pragma unroll
        for (int pp = 0; pp < APPLY_PAR - 1; ++pp) r += sRed[pp][qi][qj][e][lane];  // fixed order
        out[(size_t)v * KHP + h] = r;
        if (out32 && v < n && h < Kh) out32[(size_t)v * Kh + h] = (float)r;
      }
  }
}

// ---- dlbs from Q = M^-1 dP; grid = Kh, one handle (one softmax column) per WG of 16 waves --------------------
// A handful of flops per vertex behind long load latencies: 1024 threads keep every loop to a few rounds of
// independent loads (256 threads: 80 dependent rounds in the (A Q), (A P) sums alone, 33 us; now ~10 us).
constexpr int BWD_T = 1024;
__global__ __launch_bounds__(BWD_T) void k_solve_bwd_lbs(SolveWs s, int Kh, float* __restrict__ grad_lbs) {
  __shared__ double scratch[BWD_T], scratch2[BWD_T];
  __shared__ double sAQ[KHP], sAP[KHP];
  const int h = blockIdx.x, t = threadIdx.x;
  const double* a_row = s.A64 + (size_t)h * s.n_pad;
  {  // (A Q)[h][h'], (A P)[h][h']: 32 partial sums per output, two independent chains each
    const int hp = t & 31, part = t >> 5;
    double aq0 = 0.0, aq1 = 0.0, ap0 = 0.0, ap1 = 0.0;
    for (int v = part; v < s.n; v += 64) {
      const int v1 = v + 32;
      const double a0 = a_row[v], a1 = v1 < s.n ? a_row[v1] : 0.0;
      const size_t o0 = (size_t)v * KHP + hp, o1 = (size_t)(v1 < s.n ? v1 : v) * KHP + hp;
      aq0 += a0 * s.Q[o0]; ap0 += a0 * s.X[o0];
      aq1 += a1 * s.Q[o1]; ap1 += a1 * s.X[o1];
    }
    scratch[t] = aq0 + aq1;
    scratch2[t] = ap0 + ap1;
    __syncthreads();
    if (t < 64) {  // fixed order over the 32 parts
      const double* src = t < 32 ? scratch : scratch2;
      double z = 0.0;
      for (int q = 0; q < 32; ++q) z += src[(t & 31) + 32 * q];
      (t < 32 ? sAQ : sAP)[t & 31] = z;
    }
    __syncthreads();
  }
  double* gA = s.W + (size_t)(s.n_pad + h) * s.ld;  // the right-hand-side row of W is free by now
  double dot = 0.0;
  for (int v = t; v < s.n; v += BWD_T) {
    const double* q = s.Q + (size_t)v * KHP;
    const double* p = s.X + (size_t)v * KHP;
    double g = q[h];
    for (int hp = 0; hp < Kh; ++hp) g -= sAQ[hp] * p[hp] + sAP[hp] * q[hp];
    gA[v] = g;
    dot += a_row[v] * g;
  }
  scratch[t] = dot;
  __syncthreads();
  for (int w = BWD_T / 2; w > 0; w >>= 1) {
    if (t < w) scratch[t] += scratch[t + w];
    __syncthreads();
  }
  dot = scratch[0];
  for (int v = t; v < s.n; v += BWD_T) grad_lbs[(size_t)v * Kh + h] = (float)(a_row[v] * (gA[v] - dot));
}

}  // namespace acfm

using namespace acfm;

#ifndef ACFM_CHOL_STEPS
#define ACFM_CHOL_STEPS 0  // 1: the factorisation as one launch per tile column (the round-1 path, kept for A/B runs)
#endif

// One workgroup per CU: a second one on the CU of a diagonal job, even one that only waits, slows that job's
// serial chain (measured on the 642-vertex solve: 186 us with two per CU, 175 with one; 176 / 200 / 296 us with
// 192 / 128 / 64 workgroups in all).
#ifndef CHOL_PER_CU
#define CHOL_PER_CU 1
#endif
// workgroups of k_chol_tiles the device holds at once; only a grid size (the ticket makes any number correct)
static int chol_resident_workgroups() {
  static std::atomic<int> cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int v = cached[dev].load(std::memory_order_relaxed);
  if (v > 0) return v;
  int cus = 0, per = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k_chol_tiles, 256, 0) != hipSuccess || per <= 0) per = 1;
  v = cus * (per > CHOL_PER_CU ? CHOL_PER_CU : per);
  cached[dev].store(v, std::memory_order_relaxed);
  return v;
}

extern "C" {

size_t acfm_deform_solve_workspace_bytes(int V, int Kh) {
  if (V <= 0 || Kh <= 0 || Kh > KHP) return 0;
  return carve_solve(nullptr, V).bytes;
}

int acfm_deform_solve(const float* L, const float* lbs, int V, int Kh, float* P, void* ws, size_t ws_bytes,
                      void* stream) {
  if (!L || !lbs || !P || !ws || V <= 0 || Kh <= 0 || Kh > KHP || V > 16384) return ACFM_E_BADARG;
  SolveWs s = carve_solve(ws, V);
  if (ws_bytes < s.bytes) return ACFM_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(ACFM_PROF_SOLVE, st);
#if ACFM_CHOL_STEPS
  if (zero_async(s.W + (size_t)(s.n_pad + NB) * s.ld, sizeof(double) * (size_t)s.n_pad * s.ld, st) != ACFM_OK)
    return ACFM_E_LAUNCH;
  hipLaunchKernelGGL(k_solve_softmax, dim3(KHP), dim3(256), 0, st, lbs, s, Kh);
#else
  hipLaunchKernelGGL(k_solve_softmax, dim3(KHP + (2 * s.nblk + 2) * s.nblk), dim3(256), 0, st, lbs, s, Kh);
#endif
  hipLaunchKernelGGL(k_solve_gram_rows, dim3(s.n_pad), dim3(256), 0, st, L, s, Kh);
#if ACFM_CHOL_STEPS
  hipLaunchKernelGGL(k_chol_first, dim3(1), dim3(256), 0, st, s);
  for (int k = 0; k < s.nblk; ++k) {
    const int m = s.nblk - 1 - k;
    hipLaunchKernelGGL(k_chol_step, dim3(m + 1, s.nblk + 1), dim3(256), 0, st, s, k);
  }
  hipLaunchKernelGGL((k_apply_R<false, 0>), dim3(s.nblk), dim3(64 * 2 * APPLY_PAR), 0, st, s, (const double*)nullptr,
                     (const float*)nullptr, Kh, s.X, P);
#else
  {  // factorisation, R = L^-T, Y^T and P = R Y in one launch
    const int jobs = s.nblk * (s.nblk + 3), cap = chol_resident_workgroups();
    hipLaunchKernelGGL(k_chol_tiles, dim3(jobs < cap ? jobs : cap), dim3(256), 0, st, s, Kh, P);
  }
#endif
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_deform_solve_backward(const float* grad_P, int V, int Kh, void* ws, size_t ws_bytes, float* grad_lbs,
                               void* stream) {
  if (!grad_P || !grad_lbs || !ws || V <= 0 || Kh <= 0 || Kh > KHP || V > 16384) return ACFM_E_BADARG;
  SolveWs s = carve_solve(ws, V);
  if (ws_bytes < s.bytes) return ACFM_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(ACFM_PROF_SOLVE_BWD, st);
  hipLaunchKernelGGL((k_apply_R<true, 2>), dim3(s.nblk), dim3(64 * 2 * APPLY_PAR), 0, st, s, (const double*)nullptr, grad_P, Kh,
                     s.Z, (float*)nullptr);
  hipLaunchKernelGGL((k_apply_R<false, 1>), dim3(s.nblk), dim3(64 * 2 * APPLY_PAR), 0, st, s, (const double*)s.Z,
                     (const float*)nullptr, Kh, s.Q, (float*)nullptr);
  hipLaunchKernelGGL(k_solve_bwd_lbs, dim3(Kh), dim3(BWD_T), 0, st, s, Kh, grad_lbs);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

#ifdef ACFM_DIAG
int acfm_debug_potrf_stamps(long long* out_host) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(acfm::g_potrf_stamps), sizeof(long long) * 16) == hipSuccess ? ACFM_OK : ACFM_E_LAUNCH;
}
// diagnostic build only: the 8 clock stamps (10 ns units) of each diagonal job of k_chol_tiles
int acfm_debug_solve_stamps(const void* ws, int V, long long* out_host, int n) {
  SolveWs s = carve_solve(const_cast<void*>(ws), V);
  if (n > 8 * s.nblk) n = 8 * s.nblk;
  return hipMemcpy(out_host, s.Z, sizeof(long long) * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess ? ACFM_OK : ACFM_E_LAUNCH;
}
#endif

size_t acfm_deform_solve_info_offset(int V) {
  if (V <= 0) return 0;
  SolveWs s = carve_solve(nullptr, V);
  return (size_t)((char*)s.info - (char*)nullptr);
}

int acfm_deform_solve_info(const void* ws, size_t ws_bytes, int V, int* info_host, void* stream) {
  if (!ws || !info_host || V <= 0) return ACFM_E_BADARG;
  SolveWs s = carve_solve(const_cast<void*>(ws), V);
  if (ws_bytes < s.bytes) return ACFM_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemcpyAsync(info_host, s.info, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) return ACFM_E_LAUNCH;
  if (hipStreamSynchronize(st) != hipSuccess) return ACFM_E_LAUNCH;
  return ACFM_OK;
}

}  // extern "C"
