// Reprojection-loss kernels for gfx950 (HBM-bound streaming reductions):
//   fused silhouette losses  <- loss_utils.l1_loss / iou / edt_loss   (loss_utils.py:18-32, 72-77, 245-253)
//   visible-vertex bitmap    <- loss_utils.bds_loss :214-224 / optical_flow_loss :432-443
//   boundary loss            <- loss_utils.bds_loss :204-237
#include "acfm_common.h"

namespace acfm {

constexpr int LTPB = 256;
constexpr int PIX_PER_BLOCK = 2048;  // 8 px per thread: 2 x float4

// out[n] += (sum|m-gt|/HW, sum m*gt, sum(m+gt-m*gt), sum edt*m/HW); one pass over the mask.
// (prediction n is compared with reference n % RB: the G camera hypotheses of a frame share the
// frame's ground truth, so the trainer's gt.repeat(G, ...) copies are never made)
__global__ __launch_bounds__(LTPB) void k_mask_losses(const float* __restrict__ mask,
                                                      const float* __restrict__ gt,
                                                      const float* __restrict__ edt, int HW, int RB,
                                                      float* __restrict__ out) {
  __shared__ float s_red[4][4];
  const int n = blockIdx.y, tid = threadIdx.x;
  const size_t base = (size_t)n * HW, rbase = (size_t)(n % RB) * HW;
  const int start = blockIdx.x * PIX_PER_BLOCK;
  const int end = min(start + PIX_PER_BLOCK, HW);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const bool vec = ((HW & 3) == 0);
  if (vec) {
    // the block's two 16-byte pieces per thread and array are loaded before any of them is used
    constexpr int U = PIX_PER_BLOCK / (LTPB * 4);
    float4 m[U], g[U], e[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = start + (u * LTPB + tid) * 4;
      const bool in = i < end;
      m[u] = in ? *reinterpret_cast<const float4*>(mask + base + i) : make_float4(0, 0, 0, 0);
      g[u] = (in && gt) ? *reinterpret_cast<const float4*>(gt + rbase + i) : make_float4(0, 0, 0, 0);
      e[u] = (in && edt) ? *reinterpret_cast<const float4*>(edt + rbase + i) : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a0 += fabsf(m[u].x - g[u].x) + fabsf(m[u].y - g[u].y) + fabsf(m[u].z - g[u].z) + fabsf(m[u].w - g[u].w);
      a1 += m[u].x * g[u].x + m[u].y * g[u].y + m[u].z * g[u].z + m[u].w * g[u].w;
      a2 += (m[u].x + g[u].x - m[u].x * g[u].x) + (m[u].y + g[u].y - m[u].y * g[u].y) +
            (m[u].z + g[u].z - m[u].z * g[u].z) + (m[u].w + g[u].w - m[u].w * g[u].w);
      a3 += e[u].x * m[u].x + e[u].y * m[u].y + e[u].z * m[u].z + e[u].w * m[u].w;
    }
  } else {
    for (int i = start + tid; i < end; i += LTPB) {
      const float m = mask[base + i], g = gt ? gt[rbase + i] : 0.f, e = edt ? edt[rbase + i] : 0.f;
      a0 += fabsf(m - g); a1 += m * g; a2 += m + g - m * g; a3 += e * m;
    }
  }
  a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
  const int w = tid >> 6;
  if ((tid & 63) == 0) { s_red[w][0] = a0; s_red[w][1] = a1; s_red[w][2] = a2; s_red[w][3] = a3; }
  __syncthreads();
  if (tid < 4) {
    float v = s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid];
    if (tid == 0 || tid == 3) v = v / (float)HW;
    atomicAdd(&out[4 * (size_t)n + tid], v);
  }
}

// grad_mask = go0*sign(m-gt)/HW + go1*gt + go2*(1-gt) + go3*edt/HW
__global__ __launch_bounds__(LTPB) void k_mask_losses_bwd(const float* __restrict__ mask,
                                                          const float* __restrict__ gt,
                                                          const float* __restrict__ edt,
                                                          const float* __restrict__ go, int HW, int RB,
                                                          float* __restrict__ grad_mask) {
  const int n = blockIdx.y;
  const size_t base = (size_t)n * HW, rbase = (size_t)(n % RB) * HW;
  const float inv = 1.0f / (float)HW;
  const float g0 = go[4 * n] * inv, g1 = go[4 * n + 1], g2 = go[4 * n + 2], g3 = go[4 * n + 3] * inv;
  const int start = blockIdx.x * PIX_PER_BLOCK;
  const int end = min(start + PIX_PER_BLOCK, HW);
  auto one = [&](float m, float g, float e) {
    const float df = m - g;
    const float sgn = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);
    return g0 * sgn + g1 * g + g2 * (1.0f - g) + g3 * e;
  };
  if ((HW & 3) == 0) {   // 16-byte loads and stores
    for (int i = start + threadIdx.x * 4; i < end; i += LTPB * 4) {
      const float4 m = *reinterpret_cast<const float4*>(mask + base + i);
      float4 g = make_float4(0, 0, 0, 0), e = make_float4(0, 0, 0, 0);
      if (gt) g = *reinterpret_cast<const float4*>(gt + rbase + i);
      if (edt) e = *reinterpret_cast<const float4*>(edt + rbase + i);
      *reinterpret_cast<float4*>(grad_mask + base + i) =
          make_float4(one(m.x, g.x, e.x), one(m.y, g.y, e.y), one(m.z, g.z, e.z), one(m.w, g.w, e.w));
    }
    return;
  }
  for (int i = start + threadIdx.x; i < end; i += LTPB) {
    const float m = mask[base + i], g = gt ? gt[rbase + i] : 0.f, e = edt ? edt[rbase + i] : 0.f;
    grad_mask[base + i] = one(m, g, e);
  }
}

// masked texture MSE (multiframe/main.py:655-662): per mesh mean over (3,H,W) of
// (tex*m - img*m)^2; out[n] accumulated with one atomic per block.
__global__ __launch_bounds__(LTPB) void k_tex_mse(const float* __restrict__ tex,
                                                  const float* __restrict__ img,
                                                  const float* __restrict__ m, int HW, int RB,
                                                  float* __restrict__ out) {
  __shared__ float s_red[4];
  const int n = blockIdx.y, tid = threadIdx.x;
  const size_t b3 = (size_t)n * 3 * HW, r3 = (size_t)(n % RB) * 3 * HW, b1 = (size_t)(n % RB) * HW;
  const int start = blockIdx.x * PIX_PER_BLOCK;
  const int end = min(start + PIX_PER_BLOCK, HW);
  float acc = 0.f;
  if ((HW & 3) == 0) {   // 16-byte loads
    for (int i = start + tid * 4; i < end; i += LTPB * 4) {
      const float4 mk = *reinterpret_cast<const float4*>(m + b1 + i);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float4 t = *reinterpret_cast<const float4*>(tex + b3 + (size_t)c * HW + i);
        const float4 g = *reinterpret_cast<const float4*>(img + r3 + (size_t)c * HW + i);
        const float d0 = t.x * mk.x - g.x * mk.x, d1 = t.y * mk.y - g.y * mk.y;
        const float d2 = t.z * mk.z - g.z * mk.z, d3 = t.w * mk.w - g.w * mk.w;
        acc += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
      }
    }
  } else
  for (int i = start + tid; i < end; i += LTPB) {
    const float mk = m[b1 + i];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = tex[b3 + (size_t)c * HW + i] * mk - img[r3 + (size_t)c * HW + i] * mk;
      acc += d * d;
    }
  }
  acc = wave_sum(acc);
  if ((tid & 63) == 0) s_red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) atomicAdd(&out[n], (s_red[0] + s_red[1] + s_red[2] + s_red[3]) / (3.0f * (float)HW));
}

__global__ __launch_bounds__(LTPB) void k_tex_mse_bwd(const float* __restrict__ tex,
                                                      const float* __restrict__ img,
                                                      const float* __restrict__ m,
                                                      const float* __restrict__ go, int HW, int RB,
                                                      float* __restrict__ gtex) {
  const int n = blockIdx.y;
  const size_t b3 = (size_t)n * 3 * HW, r3 = (size_t)(n % RB) * 3 * HW, b1 = (size_t)(n % RB) * HW;
  const float w = go[n] * 2.0f / (3.0f * (float)HW);
  const int start = blockIdx.x * PIX_PER_BLOCK;
  const int end = min(start + PIX_PER_BLOCK, HW);
  if ((HW & 3) == 0) {   // 16-byte loads and stores
    for (int i = start + threadIdx.x * 4; i < end; i += LTPB * 4) {
      const float4 mk = *reinterpret_cast<const float4*>(m + b1 + i);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const size_t o = b3 + (size_t)c * HW + i;
        const float4 t = *reinterpret_cast<const float4*>(tex + o);
        const float4 g = *reinterpret_cast<const float4*>(img + r3 + (size_t)c * HW + i);
        *reinterpret_cast<float4*>(gtex + o) =
            make_float4(w * (t.x * mk.x - g.x * mk.x) * mk.x, w * (t.y * mk.y - g.y * mk.y) * mk.y,
                        w * (t.z * mk.z - g.z * mk.z) * mk.z, w * (t.w * mk.w - g.w * mk.w) * mk.w);
      }
    }
    return;
  }
  for (int i = start + threadIdx.x; i < end; i += LTPB) {
    const float mk = m[b1 + i];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const size_t o = b3 + (size_t)c * HW + i;
      gtex[o] = w * (tex[o] * mk - img[r3 + (size_t)c * HW + i] * mk) * mk;
    }
  }
}

__global__ void k_visible(const int64_t* __restrict__ p2f, const int64_t* __restrict__ faces, int V,
                          int F, int HW, int K, uint8_t* __restrict__ vis) {
  const int n = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const int64_t fp = p2f[((size_t)n * HW + p) * K];
  if (fp < 0) return;
  int f = (int)(fp - (int64_t)n * F);
  f = min(max(f, 0), F - 1);
  const int64_t* fi = faces + ((size_t)n * F + f) * 3;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int v = min(max((int)fi[j], 0), V - 1);
    vis[(size_t)n * V + v] = 1;
  }
}

// One workgroup = 64 boundary points x all vertices: its four waves search a quarter of the vertices
// each (the search is a chain of V dependent compare-selects per point: one wave per SIMD left the
// chip idle for 20 us) and wave 0 merges the four partial minima in vertex order, so the first
// nearest vertex wins exactly as in a single ascending scan.
__global__ __launch_bounds__(LTPB) void k_bds_loss(const float* __restrict__ verts_xy,
                                                   const float* __restrict__ bds,
                                                   const uint8_t* __restrict__ vis, int V, int P, int RB,
                                                   float* __restrict__ loss, int32_t* __restrict__ argmin) {
  extern __shared__ float s_xy[];  // [V][2], x = +inf for invisible verts
  __shared__ float s_best[4][64];
  __shared__ int s_bi[4][64];
  const int n = blockIdx.y, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  // (unconditional loads, four vertices per thread in flight: with the loads behind the visibility
  // test this fill was a chain of ~9 memory latencies and the whole kernel took 21 us)
  for (int v0 = tid; v0 < V; v0 += 4 * LTPB) {
    uint8_t vz[4];
    float2 xy[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int v = min(v0 + u * LTPB, V - 1);
      vz[u] = vis[(size_t)n * V + v];
      xy[u] = *reinterpret_cast<const float2*>(verts_xy + ((size_t)n * V + v) * 2);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int v = v0 + u * LTPB;
      if (v < V) {
        s_xy[2 * v] = vz[u] ? xy[u].x : __builtin_inff();
        s_xy[2 * v + 1] = vz[u] ? xy[u].y : 0.f;
      }
    }
  }
  __syncthreads();
  const int p = blockIdx.x * 64 + lane;
  float bx = 0.f, by = 0.f, bm = 0.f;
  if (p < P) {
    const float* b = bds + ((size_t)(n % RB) * P + p) * 3;
    bx = b[0]; by = b[1]; bm = b[2];
  }
  float best = 1000.0f;  // loss_utils.py:228: invisible vertices sit at distance 1000
  int bi = -1;
  const int chunk = (V + 3) / 4, v0 = wv * chunk, v1 = min(V, v0 + chunk);
  // The wave's quarter of the vertices is packed IN PLACE to its visible ones (half of a closed mesh is never seen):
  // 64 vertices a round, every lane reads its vertex before any lane writes, a kept vertex moves to a position at or
  // before its own, order kept -- so the first nearest vertex still wins; s_idx remembers where it came from.
  int* s_idx = reinterpret_cast<int*>(s_xy + 2 * (size_t)V);
  int nvis = 0;
  {
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int base = v0; base < v1; base += 64) {
      const int v = base + lane;
      float2 q = make_float2(__builtin_inff(), 0.f);
      if (v < v1) q = *reinterpret_cast<const float2*>(s_xy + 2 * v);
      const bool keep = (v < v1) && !(q.x == __builtin_inff());
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const int pos = v0 + nvis + __popcll(m & lt);
        *reinterpret_cast<float2*>(s_xy + 2 * pos) = q;
        s_idx[pos] = v;
      }
      nvis += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
#pragma unroll 8
  for (int v = v0; v < v0 + nvis; ++v) {
    const float2 q = *reinterpret_cast<const float2*>(s_xy + 2 * v);
    const float dx = bx - q.x, dy = by - q.y;
    const float d = dx * dx + dy * dy;
    if (d < best) { best = d; bi = v; }
  }
  if (bi >= 0) bi = s_idx[bi];
  s_best[wv][lane] = best; s_bi[wv][lane] = bi;
  __syncthreads();
  if (wv != 0) return;
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const float d = s_best[w][lane];
    if (d < best) { best = d; bi = s_bi[w][lane]; }
  }
  float contrib = 0.f;
  if (p < P) {
    contrib = best * bm;
    argmin[(size_t)n * P + p] = bi;
  }
  contrib = wave_sum(contrib);
  if (lane == 0) atomicAdd(&loss[n], contrib);
}

// One workgroup per mesh: the points' contributions are summed per vertex in LDS (P <= ~1000 points
// onto V vertices) and the whole [V,2] gradient row is stored, zeros included -- no global atomics
// (64 k scattered memory-side atomics took 13 us) and no zero fill of the output.
__global__ __launch_bounds__(256) void k_bds_loss_bwd(const float* __restrict__ verts_xy,
                                                      const float* __restrict__ bds,
                                                      const int32_t* __restrict__ argmin,
                                                      const float* __restrict__ gl, int V, int P, int RB,
                                                      float* __restrict__ gv) {
  extern __shared__ float s_g[];  // [V][2]
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < 2 * V; i += 256) s_g[i] = 0.f;
  __syncthreads();
  const float go = gl[n];
  for (int p0 = tid; p0 < P; p0 += 4 * 256) {      // four points per thread in flight
    int v[4];
    float bx[4], by[4], bm[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = min(p0 + u * 256, P - 1);
      v[u] = (p0 + u * 256 < P) ? argmin[(size_t)n * P + p] : -1;
      const float* b = bds + ((size_t)(n % RB) * P + p) * 3;
      bx[u] = b[0]; by[u] = b[1]; bm[u] = b[2];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float g = go * bm[u];
      if (v[u] < 0 || g == 0.f) continue;
      const float* x = verts_xy + ((size_t)n * V + v[u]) * 2;
      atomicAdd(&s_g[2 * v[u]], 2.0f * (x[0] - bx[u]) * g);
      atomicAdd(&s_g[2 * v[u] + 1], 2.0f * (x[1] - by[u]) * g);
    }
  }
  __syncthreads();
  float* o = gv + (size_t)n * V * 2;
  for (int i = tid; i < 2 * V; i += 256) o[i] = s_g[i];
}

// ---- optical-flow loss (loss_utils.py:419-474), one workgroup per (clip, frame k >= 1) --------------
// per vertex: GT flow = nearest pixel of the frame's flow image at the projected vertex
// (grid_sample, mode nearest, align_corners False, zero padding), predicted flow = pixel
// displacement of the vertex between frame k-1 and k, both counted where the vertex is visible in
// frame k and the GT flow is non-zero;  loss[c,k-1] = sum |gt - pred|_1 / H / (count + 1).
// Where the GT flow of (clip c, frame k) comes from: flows holds `clips` clips of T frames (clip c reads c % clips:
// the G hypotheses of a clip share its data), frame k reads T-1-k if flip_t (main.py:680: torch.flip(flows, dims=[1])),
// and the value is multiplied by masks[(c % clips) T + k] at the same pixel if masks is given (main.py:680-681).
struct OfSrc {
  const float* flows;
  const float* masks;
  int clips, flip_t;
};
__device__ __forceinline__ bool of_term(const float* __restrict__ proj, const OfSrc& src,
                                        const uint8_t* __restrict__ vis, int c, int k, int v, int T, int V,
                                        int H, int W, float& dx, float& dy) {
  const size_t fk = (size_t)c * T + k;
  const float* pk = proj + (fk * V + v) * 3;
  const float* pp = proj + ((fk - 1) * V + v) * 3;
  const float x = pk[0], y = pk[1];
  const float ixf = rintf(((x + 1.0f) * (float)W - 1.0f) / 2.0f), iyf = rintf(((y + 1.0f) * (float)H - 1.0f) / 2.0f);
  float gx = 0.f, gy = 0.f;
  if (ixf >= 0.f && ixf <= (float)(W - 1) && iyf >= 0.f && iyf <= (float)(H - 1)) {
    const size_t cs = (size_t)(c % src.clips) * T;
    const size_t pixel = (size_t)(int)iyf * W + (int)ixf;
    const float* g = src.flows + ((cs + (src.flip_t ? T - 1 - k : k)) * H * (size_t)W + pixel) * 2;
    gx = g[0]; gy = g[1];
    if (src.masks) {
      const float m = src.masks[(cs + k) * H * (size_t)W + pixel];
      gx *= m; gy *= m;
    }
  }
  const bool keep = (fabsf(gx) + fabsf(gy) != 0.0f) && vis[fk * V + v] != 0;
  const float fw = (float)W;
  const float fpx = fw * (pp[0] + 1.0f) / 2.0f - fw * (x + 1.0f) / 2.0f;
  const float fpy = fw * (pp[1] + 1.0f) / 2.0f - fw * (y + 1.0f) / 2.0f;
  dx = gx - fpx; dy = gy - fpy;
  return keep;
}

__global__ __launch_bounds__(LTPB) void k_of_loss(const float* __restrict__ proj, OfSrc flows,
                                                  const uint8_t* __restrict__ vis, int T, int V, int H, int W,
                                                  float* __restrict__ loss, float* __restrict__ count) {
  __shared__ float s_red[4][2];
  const int k = blockIdx.x + 1, c = blockIdx.y, tid = threadIdx.x;
  float sum = 0.f, cnt = 0.f;
  for (int v = tid; v < V; v += LTPB) {
    float dx, dy;
    if (of_term(proj, flows, vis, c, k, v, T, V, H, W, dx, dy)) { sum += fabsf(dx) + fabsf(dy); cnt += 1.0f; }
  }
  sum = wave_sum(sum); cnt = wave_sum(cnt);
  if ((tid & 63) == 0) { s_red[tid >> 6][0] = sum; s_red[tid >> 6][1] = cnt; }
  __syncthreads();
  if (tid == 0) {
    const float s = s_red[0][0] + s_red[1][0] + s_red[2][0] + s_red[3][0];
    const float n = s_red[0][1] + s_red[1][1] + s_red[2][1] + s_red[3][1];
    loss[(size_t)c * (T - 1) + k - 1] = s / (float)H / (n + 1.0f);
    count[(size_t)c * (T - 1) + k - 1] = n;
  }
}

__global__ __launch_bounds__(LTPB) void k_of_loss_bwd(const float* __restrict__ proj, OfSrc flows,
                                                      const uint8_t* __restrict__ vis, const float* __restrict__ count,
                                                      const float* __restrict__ gl, int T, int V, int H, int W,
                                                      float* __restrict__ gproj) {
  const int k = blockIdx.y + 1, c = blockIdx.z;
  const int v = blockIdx.x * LTPB + threadIdx.x;
  if (v >= V) return;
  float dx, dy;
  if (!of_term(proj, flows, vis, c, k, v, T, V, H, W, dx, dy)) return;
  const size_t o = (size_t)c * (T - 1) + k - 1;
  // d|gt - pred| / d pred = -sign(gt - pred);  pred = pix[k-1] - pix[k],  pix = W (xy + 1) / 2
  const float g = gl[o] / (float)H / (count[o] + 1.0f) * (float)W / 2.0f;
  const float sx = dx > 0.f ? 1.f : (dx < 0.f ? -1.f : 0.f), sy = dy > 0.f ? 1.f : (dy < 0.f ? -1.f : 0.f);
  const size_t fk = (size_t)c * T + k;
  float* gk = gproj + (fk * V + v) * 3;
  float* gp = gproj + ((fk - 1) * V + v) * 3;
  atomicAdd(&gp[0], -sx * g); atomicAdd(&gp[1], -sy * g);
  atomicAdd(&gk[0], sx * g); atomicAdd(&gk[1], sy * g);
}

// total = (1/N) sum_n sum_t sum_c w[t][c] T_t[n,c]: the weighted sum of per-mesh loss terms and its mean
// over the batch (multiframe/main.py:716-746 without the hypothesis weights) as ONE launch each way
// instead of ~13 elementwise / reduction launches on 64-element vectors forward and as many backward.
struct CombArgs {
  const float* t[4];
  float* g[4];
  int cols[4];
  float w[4][4];
  int nterms;
};
__global__ __launch_bounds__(256) void k_combine(CombArgs a, int N, float* __restrict__ total) {
  __shared__ float s_red[4];
  float acc = 0.f;
  for (int n = threadIdx.x; n < N; n += 256)
    for (int t = 0; t < a.nterms; ++t)
      for (int c = 0; c < a.cols[t]; ++c)
        if (a.w[t][c] != 0.f) acc += a.w[t][c] * a.t[t][(size_t)n * a.cols[t] + c];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) total[0] = (s_red[0] + s_red[1] + s_red[2] + s_red[3]) / (float)N;
}
__global__ __launch_bounds__(256) void k_combine_bwd(CombArgs a, int N, const float* __restrict__ go) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float g = go[0] / (float)N;
  for (int t = 0; t < a.nterms; ++t)
    if (a.g[t])
      for (int c = 0; c < a.cols[t]; ++c) a.g[t][(size_t)n * a.cols[t] + c] = a.w[t][c] * g;
}

// ---- total per hypothesis, softmax weighting over the hypotheses, weighted mean: one launch each way ---------
// multiframe/main.py:716-746.  One thread per frame n walks its G hypotheses; everything is summed in a fixed
// order (one workgroup), so the result is reproducible.
constexpr int HYP_MAX_TERMS = 8;
struct HypArgs {
  const float* t[HYP_MAX_TERMS];
  float* g[HYP_MAX_TERMS];
  float w[HYP_MAX_TERMS];
  float aux_w[HYP_MAX_TERMS];
  int aux_group[HYP_MAX_TERMS];   // -1, 0 or 1: which auxiliary sum the term also enters (with aux_w)
  int nterms;
};
__global__ __launch_bounds__(256) void k_hyp_total(HypArgs a, int G, int N, float* __restrict__ total,
                                                   float* __restrict__ probs, float* __restrict__ aux0,
                                                   float* __restrict__ aux1, float* __restrict__ out) {
  __shared__ float s_red[4][4 + HYP_MAX_TERMS];
  float acc_w = 0.f, acc_t = 0.f, acc_a0 = 0.f, acc_a1 = 0.f, acc_term[HYP_MAX_TERMS];
#pragma unroll
  for (int t = 0; t < HYP_MAX_TERMS; ++t) acc_term[t] = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) {
    float lmin = __builtin_inff();
    for (int g = 0; g < G; ++g) {
      const size_t i = (size_t)g * N + n;
      float tot = 0.f, x0 = 0.f, x1 = 0.f;
#pragma unroll
      for (int t = 0; t < HYP_MAX_TERMS; ++t) {
        if (t >= a.nterms) break;
        const float v = a.t[t][i];
        tot += a.w[t] * v;
        acc_term[t] += v;
        if (a.aux_group[t] == 0) x0 += a.aux_w[t] * v;
        if (a.aux_group[t] == 1) x1 += a.aux_w[t] * v;
      }
      total[i] = tot;
      if (aux0) aux0[i] = x0;
      if (aux1) aux1[i] = x1;
      acc_t += tot; acc_a0 += x0; acc_a1 += x1;
      lmin = fminf(lmin, tot);
    }
    float den = 0.f;
    for (int g = 0; g < G; ++g) den += expf(-(total[(size_t)g * N + n] - lmin));   // softmax(-total) over g
    float wsum = 0.f;
    for (int g = 0; g < G; ++g) {
      const size_t i = (size_t)g * N + n;
      const float p = expf(-(total[i] - lmin)) / den;
      probs[i] = p;
      wsum += p * total[i];
    }
    acc_w += wsum;
  }
  // out: [0] weighted loss, [1] mean total, [2] mean aux0, [3] mean aux1, [4 + t] mean of term t
  float vals[4 + HYP_MAX_TERMS] = {acc_w, acc_t, acc_a0, acc_a1};
#pragma unroll
  for (int t = 0; t < HYP_MAX_TERMS; ++t) vals[4 + t] = acc_term[t];
#pragma unroll
  for (int k = 0; k < 4 + HYP_MAX_TERMS; ++k) {
    const float v = wave_sum(vals[k]);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4 + HYP_MAX_TERMS) {
    const int k = threadIdx.x;
    const float v = s_red[0][k] + s_red[1][k] + s_red[2][k] + s_red[3][k];
    out[k] = v / (k == 0 ? (float)N : (float)N * (float)G);
  }
}
__global__ __launch_bounds__(256) void k_hyp_total_bwd(HypArgs a, int GN, int N, const float* __restrict__ go,
                                                       const float* __restrict__ probs) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= GN) return;
  const float g = go[0] * probs[i] / (float)N;   // the probabilities carry no gradient (detached in the reference)
#pragma unroll
  for (int t = 0; t < HYP_MAX_TERMS; ++t) {
    if (t >= a.nterms) break;
    if (a.g[t]) a.g[t][i] = a.w[t] * g;
  }
}

// ---- texture temporal-consistency term (multiframe/main.py:705-711), as written there ---------------------------
// The atlases [B*T,F,R,R,3] are regrouped as [B,F,R,R,T,3], that buffer is RESHAPED to rows [-1,R,R] (the reference's
// own regrouping of the trailing (R,T,3) block; reproduced, not "fixed"), and the loss is the mean L2 norm over the
// last axis of the differences of neighbouring rows i, i+1.  X[m,i,j] below is element m R^2 + i R + j of the regrouped
// buffer; cyc_src maps it back to its place in the atlases.  Forward: one thread per (m, i) writes the norm and a
// partial sum per workgroup, a second small kernel adds the partials in order.  Backward: one thread per element
// (every atlas entry is written exactly once: no atomics, no zero fill).
struct CycDims { int B, T, F, R; };
// (index arithmetic in 32 bits whenever the atlas batch has fewer than 2^31 entries: the decode below is five
// divisions per element and 64-bit ones made these kernels 3x longer than the tensor traffic)
template <class I>
__device__ __forceinline__ I cyc_src(I q, const CycDims& d) {
  const I c = q % 3; q /= 3;
  const I t = q % (I)d.T; q /= (I)d.T;
  const I r2 = q % (I)d.R; q /= (I)d.R;
  const I r1 = q % (I)d.R; q /= (I)d.R;
  const I f = q % (I)d.F; q /= (I)d.F;
  const I b = q;
  return ((((b * (I)d.T + t) * (I)d.F + f) * (I)d.R + r1) * (I)d.R + r2) * 3 + c;
}
template <class I>
__global__ __launch_bounds__(256) void k_tex_cycle(const float* __restrict__ tex, CycDims d, size_t pairs,
                                                   float* __restrict__ norms, float* __restrict__ partial) {
  __shared__ float s_red[4];
  const I p = (I)blockIdx.x * 256 + threadIdx.x;   // pair (m, i), i < R - 1
  float nrm = 0.f;
  if ((size_t)p < pairs) {
    const I m = p / (I)(d.R - 1);
    const I i = p % (I)(d.R - 1);
    const I q0 = (m * (I)d.R + i) * (I)d.R;
    float acc = 0.f;
    for (int j = 0; j < d.R; ++j) {
      const float df = tex[cyc_src<I>(q0 + (I)j, d)] - tex[cyc_src<I>(q0 + (I)d.R + (I)j, d)];
      acc += df * df;
    }
    nrm = sqrtf(acc);
    norms[p] = nrm;
  }
  const float v = wave_sum(nrm);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}
__global__ __launch_bounds__(256) void k_tex_cycle_finish(const float* __restrict__ partial, int nblocks, float scale,
                                                          float* __restrict__ out) {
  __shared__ float s_red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) acc += partial[i];
  const float v = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) * scale;
}
template <class I>
__global__ __launch_bounds__(256) void k_tex_cycle_bwd(const float* __restrict__ tex, const float* __restrict__ norms,
                                                       const float* __restrict__ go, CycDims d, size_t total, float scale,
                                                       float* __restrict__ grad) {
  const I q = (I)blockIdx.x * 256 + threadIdx.x;   // element (m, i, j) of the regrouped buffer
  if ((size_t)q >= total) return;
  const I i = (q / (I)d.R) % (I)d.R;
  const I m = q / ((I)d.R * (I)d.R);
  const I src = cyc_src<I>(q, d);
  const float x = tex[src];
  float g = 0.f;
  if (i + 1 < (I)d.R) {         // pair (m, i): d norm / d x = (x - x_below) / norm
    const float n = norms[m * (I)(d.R - 1) + i];
    if (n > 0.f) g += (x - tex[cyc_src<I>(q + (I)d.R, d)]) / n;
  }
  if (i > 0) {                  // pair (m, i - 1): d norm / d x = -(x_above - x) / norm
    const float n = norms[m * (I)(d.R - 1) + i - 1];
    if (n > 0.f) g -= (tex[cyc_src<I>(q - (I)d.R, d)] - x) / n;
  }
  grad[src] = g * go[0] * scale;
}

}  // namespace acfm

using namespace acfm;

extern "C" {

int acfm_mask_losses(const float* mask, const float* gt, const float* edt, int N, int HW, int ref_batch,
                     float* out, void* stream) {
  if (!mask || !out || N <= 0 || N > 65535 || HW <= 0 || ref_batch <= 0 || N % ref_batch != 0) return ACFM_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (zero_async(out, sizeof(float) * 4 * (size_t)N, st) != ACFM_OK) return ACFM_E_LAUNCH;
  const int chunks = (HW + PIX_PER_BLOCK - 1) / PIX_PER_BLOCK;
  ProfScope ps(ACFM_PROF_MASK_LOSS, st);
  hipLaunchKernelGGL(k_mask_losses, dim3(chunks, N), dim3(LTPB), 0, st, mask, gt, edt, HW, ref_batch, out);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_mask_losses_backward(const float* mask, const float* gt, const float* edt,
                              const float* grad_out, int N, int HW, int ref_batch, float* grad_mask,
                              void* stream) {
  if (!mask || !grad_out || !grad_mask || N <= 0 || N > 65535 || HW <= 0 || ref_batch <= 0 || N % ref_batch != 0)
    return ACFM_E_BADARG;
  const int chunks = (HW + PIX_PER_BLOCK - 1) / PIX_PER_BLOCK;
  ProfScope ps(ACFM_PROF_MASK_LOSS_BWD, (hipStream_t)stream);
  hipLaunchKernelGGL(k_mask_losses_bwd, dim3(chunks, N), dim3(LTPB), 0, (hipStream_t)stream, mask, gt,
                     edt, grad_out, HW, ref_batch, grad_mask);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_tex_mse(const float* tex, const float* img, const float* mask, int N, int HW, int ref_batch,
                 float* out, void* stream) {
  if (!tex || !img || !mask || !out || N <= 0 || N > 65535 || HW <= 0 || ref_batch <= 0 || N % ref_batch != 0)
    return ACFM_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (zero_async(out, sizeof(float) * (size_t)N, st) != ACFM_OK) return ACFM_E_LAUNCH;
  ProfScope ps(ACFM_PROF_TEX_MSE, st);
  hipLaunchKernelGGL(k_tex_mse, dim3((HW + PIX_PER_BLOCK - 1) / PIX_PER_BLOCK, N), dim3(LTPB), 0, st, tex,
                     img, mask, HW, ref_batch, out);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_tex_mse_backward(const float* tex, const float* img, const float* mask, const float* grad_out,
                          int N, int HW, int ref_batch, float* grad_tex, void* stream) {
  if (!tex || !img || !mask || !grad_out || !grad_tex || N <= 0 || N > 65535 || HW <= 0 || ref_batch <= 0 ||
      N % ref_batch != 0)
    return ACFM_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(ACFM_PROF_TEX_MSE_BWD, st);
  hipLaunchKernelGGL(k_tex_mse_bwd, dim3((HW + PIX_PER_BLOCK - 1) / PIX_PER_BLOCK, N), dim3(LTPB), 0, st,
                     tex, img, mask, grad_out, HW, ref_batch, grad_tex);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

static int fill_comb(CombArgs& a, const void* const* terms, void* const* grads, const int* cols,
                     const float* weights, int nterms) {
  if (!cols || !weights || nterms <= 0 || nterms > 4) return ACFM_E_BADARG;
  a.nterms = nterms;
  int o = 0;
  for (int t = 0; t < 4; ++t) {
    a.t[t] = nullptr; a.g[t] = nullptr; a.cols[t] = 0;
    for (int c = 0; c < 4; ++c) a.w[t][c] = 0.f;
    if (t >= nterms) continue;
    if (cols[t] <= 0 || cols[t] > 4) return ACFM_E_BADARG;
    a.cols[t] = cols[t];
    if (terms) { if (!terms[t]) return ACFM_E_BADARG; a.t[t] = (const float*)terms[t]; }
    if (grads) a.g[t] = (float*)grads[t];
    for (int c = 0; c < cols[t]; ++c) a.w[t][c] = weights[o++];
  }
  return ACFM_OK;
}

int acfm_combine_losses(const void* const* terms, const int* cols, const float* weights, int nterms, int N,
                        float* total, void* stream) {
  CombArgs a;
  if (!terms || !total || N <= 0 || fill_comb(a, terms, nullptr, cols, weights, nterms) != ACFM_OK)
    return ACFM_E_BADARG;
  hipLaunchKernelGGL(k_combine, dim3(1), dim3(256), 0, (hipStream_t)stream, a, N, total);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_combine_losses_backward(const float* grad_total, void* const* grads, const int* cols,
                                 const float* weights, int nterms, int N, void* stream) {
  CombArgs a;
  if (!grad_total || !grads || N <= 0 || fill_comb(a, nullptr, grads, cols, weights, nterms) != ACFM_OK)
    return ACFM_E_BADARG;
  hipLaunchKernelGGL(k_combine_bwd, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, N, grad_total);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_hypothesis_total(const void* const* terms, const float* weights, const int* aux_group, const float* aux_weights,
                          int nterms, int G, int N, float* total, float* probs, float* aux0, float* aux1, float* out,
                          void* stream) {
  if (!terms || !weights || !total || !probs || !out || nterms < 1 || nterms > HYP_MAX_TERMS || G <= 0 || G > 4096 ||
      N <= 0 || (size_t)G * N > 0x7fffffffull)
    return ACFM_E_BADARG;
  HypArgs a;
  for (int t = 0; t < HYP_MAX_TERMS; ++t) {
    a.t[t] = t < nterms ? (const float*)terms[t] : nullptr;
    a.g[t] = nullptr;
    a.w[t] = t < nterms ? weights[t] : 0.f;
    a.aux_group[t] = (t < nterms && aux_group) ? aux_group[t] : -1;
    a.aux_w[t] = (t < nterms && aux_weights) ? aux_weights[t] : 0.f;
    if (t < nterms && !a.t[t]) return ACFM_E_BADARG;
    if (a.aux_group[t] < -1 || a.aux_group[t] > 1) return ACFM_E_BADARG;
  }
  a.nterms = nterms;
  hipLaunchKernelGGL(k_hyp_total, dim3(1), dim3(256), 0, (hipStream_t)stream, a, G, N, total, probs, aux0, aux1, out);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_hypothesis_total_backward(const float* grad_weighted, const float* probs, const float* weights, int nterms,
                                   int G, int N, void* const* grads, void* stream) {
  if (!grad_weighted || !probs || !weights || !grads || nterms < 1 || nterms > HYP_MAX_TERMS || G <= 0 || N <= 0 ||
      (size_t)G * N > 0x7fffffffull)
    return ACFM_E_BADARG;
  HypArgs a;
  for (int t = 0; t < HYP_MAX_TERMS; ++t) {
    a.t[t] = nullptr;
    a.g[t] = t < nterms ? (float*)grads[t] : nullptr;
    a.w[t] = t < nterms ? weights[t] : 0.f;
    a.aux_group[t] = -1;
    a.aux_w[t] = 0.f;
  }
  a.nterms = nterms;
  const int GN = G * N;
  hipLaunchKernelGGL(k_hyp_total_bwd, dim3((GN + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, GN, N,
                     grad_weighted, probs);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

size_t acfm_texture_cycle_scratch_floats(int B, int T, int F, int R) {
  if (B <= 0 || T <= 0 || F <= 0 || R < 2) return 0;
  const size_t total = (size_t)B * T * F * R * R * 3, pairs = total / ((size_t)R * R) * (R - 1);
  return pairs + (pairs + 255) / 256;
}

int acfm_texture_cycle(const float* textures, int B, int T, int F, int R, float* scratch, float* loss, void* stream) {
  if (!textures || !scratch || !loss || B <= 0 || T <= 0 || F <= 0 || R < 2) return ACFM_E_BADARG;
  const size_t total = (size_t)B * T * F * R * R * 3, pairs = total / ((size_t)R * R) * (R - 1);
  const size_t nblocks = (pairs + 255) / 256;
  if (nblocks > 0x7fffffffull) return ACFM_E_BADARG;
  const CycDims d = {B, T, F, R};
  hipStream_t st = (hipStream_t)stream;
  if (total < 0x7fffffffull)
    hipLaunchKernelGGL(k_tex_cycle<unsigned>, dim3((unsigned)nblocks), dim3(256), 0, st, textures, d, pairs, scratch, scratch + pairs);
  else
    hipLaunchKernelGGL(k_tex_cycle<size_t>, dim3((unsigned)nblocks), dim3(256), 0, st, textures, d, pairs, scratch, scratch + pairs);
  hipLaunchKernelGGL(k_tex_cycle_finish, dim3(1), dim3(256), 0, st, scratch + pairs, (int)nblocks, 1.0f / (float)pairs, loss);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_texture_cycle_backward(const float* textures, const float* scratch, const float* grad_loss, int B, int T, int F,
                                int R, float* grad_textures, void* stream) {
  if (!textures || !scratch || !grad_loss || !grad_textures || B <= 0 || T <= 0 || F <= 0 || R < 2) return ACFM_E_BADARG;
  const size_t total = (size_t)B * T * F * R * R * 3, pairs = total / ((size_t)R * R) * (R - 1);
  const size_t nblocks = (total + 255) / 256;
  if (nblocks > 0x7fffffffull) return ACFM_E_BADARG;
  const CycDims d = {B, T, F, R};
  if (total < 0x7fffffffull)
    hipLaunchKernelGGL(k_tex_cycle_bwd<unsigned>, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, textures, scratch,
                       grad_loss, d, total, 1.0f / (float)pairs, grad_textures);
  else
    hipLaunchKernelGGL(k_tex_cycle_bwd<size_t>, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, textures, scratch,
                       grad_loss, d, total, 1.0f / (float)pairs, grad_textures);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_visible_vertices(const int64_t* pix_to_face, const int64_t* faces, int N, int V, int F, int HW,
                          int K, uint8_t* vis, void* stream) {
  if (!pix_to_face || !faces || !vis || N <= 0 || N > 65535 || V <= 0 || F <= 0 || HW <= 0 || K <= 0)
    return ACFM_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (zero_async(vis, (size_t)N * V, st) != ACFM_OK) return ACFM_E_LAUNCH;
  ProfScope ps(ACFM_PROF_VISIBLE, st);
  hipLaunchKernelGGL(k_visible, dim3((HW + 255) / 256, N), dim3(256), 0, st, pix_to_face, faces, V, F,
                     HW, K, vis);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_bds_loss(const float* verts_xy, const float* bds, const uint8_t* vis, int N, int V, int P,
                  int ref_batch, float* loss, int32_t* argmin, void* stream) {
  if (!verts_xy || !bds || !vis || !loss || !argmin || N <= 0 || N > 65535 || V <= 0 || P <= 0 || ref_batch <= 0 ||
      N % ref_batch != 0)
    return ACFM_E_BADARG;
  const size_t lds = sizeof(float) * 3 * (size_t)V;   // (x, y) per vertex + the index a packed vertex came from
  if (lds > 150 * 1024) return ACFM_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (zero_async(loss, sizeof(float) * (size_t)N, st) != ACFM_OK) return ACFM_E_LAUNCH;
  ProfScope ps(ACFM_PROF_BDS, st);
  hipLaunchKernelGGL(k_bds_loss, dim3((P + 63) / 64, N), dim3(LTPB), lds, st, verts_xy, bds, vis,
                     V, P, ref_batch, loss, argmin);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_bds_loss_backward(const float* verts_xy, const float* bds, const int32_t* argmin,
                           const float* grad_loss, int N, int V, int P, int ref_batch, float* grad_verts_xy,
                           void* stream) {
  if (!verts_xy || !bds || !argmin || !grad_loss || !grad_verts_xy || N <= 0 || N > 65535 || V <= 0 ||
      P <= 0 || ref_batch <= 0 || N % ref_batch != 0)
    return ACFM_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = sizeof(float) * 2 * (size_t)V;
  if (lds > 150 * 1024) return ACFM_E_BADARG;
  ProfScope ps(ACFM_PROF_BDS_BWD, st);
  hipLaunchKernelGGL(k_bds_loss_bwd, dim3(N), dim3(256), lds, st, verts_xy, bds, argmin,
                     grad_loss, V, P, ref_batch, grad_verts_xy);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_of_loss_shared(const float* proj, const float* flows, const float* masks, const uint8_t* vis, int B, int T,
                        int V, int H, int W, int clips, int flip_t, float* loss, float* count, void* stream) {
  if (!proj || !flows || !vis || !loss || !count || B <= 0 || B > 65535 || T < 2 || T > 65535 || V <= 0 || H <= 0 ||
      W <= 0 || clips <= 0 || B % clips != 0)
    return ACFM_E_BADARG;
  const OfSrc src = {flows, masks, clips, flip_t ? 1 : 0};
  hipLaunchKernelGGL(k_of_loss, dim3(T - 1, B), dim3(LTPB), 0, (hipStream_t)stream, proj, src, vis, T, V, H, W,
                     loss, count);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_of_loss_shared_backward(const float* proj, const float* flows, const float* masks, const uint8_t* vis,
                                 const float* count, const float* grad_loss, int B, int T, int V, int H, int W, int clips,
                                 int flip_t, float* grad_proj, void* stream) {
  if (!proj || !flows || !vis || !count || !grad_loss || !grad_proj || B <= 0 || B > 65535 || T < 2 || T > 65535 ||
      V <= 0 || H <= 0 || W <= 0 || clips <= 0 || B % clips != 0)
    return ACFM_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (zero_async(grad_proj, sizeof(float) * 3 * (size_t)B * T * V, st) != ACFM_OK) return ACFM_E_LAUNCH;
  const OfSrc src = {flows, masks, clips, flip_t ? 1 : 0};
  hipLaunchKernelGGL(k_of_loss_bwd, dim3((V + LTPB - 1) / LTPB, T - 1, B), dim3(LTPB), 0, st, proj, src, vis, count,
                     grad_loss, T, V, H, W, grad_proj);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_of_loss(const float* proj, const float* flows, const uint8_t* vis, int B, int T, int V, int H, int W,
                 float* loss, float* count, void* stream) {
  return acfm_of_loss_shared(proj, flows, nullptr, vis, B, T, V, H, W, B, 0, loss, count, stream);
}

int acfm_of_loss_backward(const float* proj, const float* flows, const uint8_t* vis, const float* count,
                          const float* grad_loss, int B, int T, int V, int H, int W, float* grad_proj,
                          void* stream) {
  return acfm_of_loss_shared_backward(proj, flows, nullptr, vis, count, grad_loss, B, T, V, H, W, B, 0, grad_proj, stream);
}

}  // extern "C"
