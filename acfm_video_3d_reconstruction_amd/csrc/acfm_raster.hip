// Rasterisation kernels for gfx950: face setup, tiled top-K forward (soft silhouette K<=32,
// hard K=1 with optional atlas shading) and the silhouette backward.
//
// Replaces PyTorch3D 0.3.0's rasterize_meshes coarse/fine/backward CUDA kernels and the
// blending shaders as used by multiframe/nnutils/nmr.py:143-200, 224-238 (semantics:
// SURVEY.md App-A; the CPU oracle in oracle/acfm_oracle.c is the bit-level spec).
//
// Design:
//   * k_setup: one workgroup per mesh projects the V vertices into LDS (weak-perspective
//     camera, y flip, view transform) and writes per-face records + blur-expanded boxes.
//   * k_raster_fwd: one 256-thread workgroup per 16x16 pixel tile, each wave owns an 8x8
//     pixel block (wave64-shaped, not 32-wide).  Faces are binned against the tile with
//     wave ballots into an LDS candidate list (deterministic, face-ordered); every lane then
//     walks the list (LDS broadcast reads) and keeps its top-K (depth, face) keys in an LDS
//     column it alone owns (conflict-free [slot][lane] layout); a rank sort at the end
//     gives the ascending-depth order PyTorch3D returns.
//   * k_sil_bwd: per pixel, recompute the K distances from pix_to_face, accumulate the
//     vertex gradients of the tile in LDS (ds_add_f32) and flush non-zeros with one global
//     float atomic per touched vertex coordinate.
#include "acfm_common.h"

namespace acfm {

constexpr int TILE = 16;      // pixels per tile side (PyTorch3D's auto bin size at 128/256)
constexpr int TPB = 256;      // threads per workgroup = TILE*TILE
constexpr int CAP = 256;      // LDS candidate-list capacity (flushed in rounds when exceeded)

// ------------------------------------------------------------------------------- setup
// mode 0: verts are world coordinates -> project with cams, flip y   (nmr.py:145-149)
// mode 1: verts are already projected, no y flip                     (nmr.py:224-238)
__global__ __launch_bounds__(TPB) void k_setup(const float* __restrict__ verts,
                                               const int64_t* __restrict__ faces,
                                               const float* __restrict__ cams, int V, int F,
                                               float offset_z, int mode, float margin, RasterWs ws) {
  extern __shared__ float s_v[];  // [V][3]
  __shared__ float s_red[4][4];
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* cam = cams ? cams + 7 * (size_t)n : nullptr;
  for (int v = tid; v < V; v += TPB) {
    const float* x = verts + ((size_t)n * V + v) * 3;
    float px, py, pz;
    if (mode == 0) {
      project_point(cam, x[0], x[1], x[2], offset_z, px, py, pz);
      py = py * -1.0f;
    } else {
      px = x[0]; py = x[1]; pz = x[2];
    }
    px = -px;               // view R = diag(-1, 1, 1)
    pz = pz + ACFM_EYE_Z;   // view T = (0, 0, 2.732)
    s_v[3 * v + 0] = px; s_v[3 * v + 1] = py; s_v[3 * v + 2] = pz;
    float* o = ws.ndc + ((size_t)n * V + v) * 3;
    o[0] = px; o[1] = py; o[2] = pz;
  }
  __syncthreads();
  const float INF = __builtin_inff();
  float bx0 = INF, bx1 = -INF, by0 = INF, by1 = -INF;
  for (int f = tid; f < F; f += TPB) {
    const int64_t* fi = faces + ((size_t)n * F + f) * 3;
    int i0 = (int)fi[0], i1 = (int)fi[1], i2 = (int)fi[2];
    i0 = min(max(i0, 0), V - 1); i1 = min(max(i1, 0), V - 1); i2 = min(max(i2, 0), V - 1);
    const float x0 = s_v[3 * i0], y0 = s_v[3 * i0 + 1], z0 = s_v[3 * i0 + 2];
    const float x1 = s_v[3 * i1], y1 = s_v[3 * i1 + 1], z1 = s_v[3 * i1 + 2];
    const float x2 = s_v[3 * i2], y2 = s_v[3 * i2 + 1], z2 = s_v[3 * i2 + 2];
    const float area = edge_fn(x2, y2, x0, y0, x1, y1);
    const bool degenerate = (area <= ACFM_K_EPS && area >= -1.0f * ACFM_K_EPS);
    float4 b;
    b.x = min3f(x0, x1, x2) - margin; b.y = max3f(x0, x1, x2) + margin;
    b.z = min3f(y0, y1, y2) - margin; b.w = max3f(y0, y1, y2) + margin;
    if (degenerate) {
      b = make_float4(INF, -INF, INF, -INF);  // fails every "inside box" test
    } else {
      bx0 = fminf(bx0, b.x); bx1 = fmaxf(bx1, b.y); by0 = fminf(by0, b.z); by1 = fmaxf(by1, b.w);
    }
    const size_t o = (size_t)n * F + f;
    ws.recA[o] = make_float4(x0, y0, x1, y1);
    ws.recB[o] = make_float4(x2, y2, z0, z1);
    ws.recC[o] = make_float4(z2, area, 0.f, 0.f);
    ws.box[o] = b;
    ws.vidx[o] = make_int4(i0, i1, i2, 0);
  }
  bx0 = wave_min(bx0); bx1 = wave_max(bx1); by0 = wave_min(by0); by1 = wave_max(by1);
  const int w = tid >> 6;
  if ((tid & 63) == 0) { s_red[w][0] = bx0; s_red[w][1] = bx1; s_red[w][2] = by0; s_red[w][3] = by1; }
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < 4; ++i) {
      bx0 = fminf(bx0, s_red[i][0]); bx1 = fmaxf(bx1, s_red[i][1]);
      by0 = fminf(by0, s_red[i][2]); by1 = fmaxf(by1, s_red[i][3]);
    }
    ws.mbox[n] = make_float4(bx0, bx1, by0, by1);
  }
}

// ------------------------------------------------------------------------------- forward
struct FwdOut {
  float* mask;          // SOFT: [N,H,H]
  int64_t* p2f;         // [N,H,H,K]
  // texture branch (TEX)
  const float* atlas;   // [N,F,R,R,3]
  float* imgs;          // [N,3,H,H]
  float* sil;           // [N,H,H]
  int32_t* tidx;        // [N,H,H]
  int R;
  float gamma;
};

template <int K>
struct PixList {
  // per-thread top-K list in LDS, [slot][thread] so a wave's accesses never conflict
  unsigned long long key[K][TPB];
  float sd[K][TPB];
};

template <int K, bool CLIP, bool TEX>
__global__ __launch_bounds__(TPB) void k_raster_fwd(RasterWs ws, int F, int H, float blur, float sigma,
                                                    FwdOut out) {
  __shared__ float4 s_box[CAP], s_a[CAP], s_b[CAP];
  __shared__ float2 s_c[CAP];
  __shared__ int s_fid[CAP];
  __shared__ int s_wcnt[4];
  constexpr int KL = (K > 1) ? K : 1;
  __shared__ PixList<(K > 1) ? KL : 1> s_list;  // unused (1 slot) when K == 1

  const int n = blockIdx.y, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int tiles = (H + TILE - 1) / TILE;
  const int ty = blockIdx.x / tiles, tx = blockIdx.x % tiles;
  const int yi = ty * TILE + (wv >> 1) * 8 + (lane >> 3);
  const int xi = tx * TILE + (wv & 1) * 8 + (lane & 7);
  const bool valid = (yi < H) && (xi < H);
  const float yf = pix_to_ndc(H - 1 - yi, H);
  const float xf = pix_to_ndc(H - 1 - xi, H);
  const size_t pix = ((size_t)n * H + yi) * H + xi;

  // tile extent in NDC (pixel centres; x/y decrease with the pixel index)
  const float t_xmax = pix_to_ndc(H - 1 - tx * TILE, H), t_xmin = pix_to_ndc(H - 1 - (tx * TILE + TILE - 1), H);
  const float t_ymax = pix_to_ndc(H - 1 - ty * TILE, H), t_ymin = pix_to_ndc(H - 1 - (ty * TILE + TILE - 1), H);

  int cnt = 0;                              // entries in this pixel's list
  unsigned long long maxkey = 0; int maxslot = 0;   // valid when cnt == K (K > 1)
  unsigned long long bestkey = ~0ull; float bestsd = 0.f, bestb0 = 0.f, bestb1 = 0.f;  // K == 1

  const float4 mb = ws.mbox[n];
  const bool tile_hit = !(t_xmin > mb.y || t_xmax < mb.x || t_ymin > mb.w || t_ymax < mb.z);
  if (tile_hit) {
    int list_n = 0;
    for (int base = 0; base < F; base += TPB) {
      const int f = base + tid;
      bool pass = false;
      if (f < F) {
        const float4 b = ws.box[(size_t)n * F + f];
        pass = !(t_xmin > b.y || t_xmax < b.x || t_ymin > b.w || t_ymax < b.z);
      }
      const unsigned long long bal = __ballot(pass);
      if (lane == 0) s_wcnt[wv] = __popcll(bal);
      __syncthreads();
      int off = list_n, tot = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = s_wcnt[i];
        if (i < wv) off += c;
        tot += c;
      }
      if (pass) {
        const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
        const size_t o = (size_t)n * F + f;
        s_box[pos] = ws.box[o];
        s_a[pos] = ws.recA[o];
        s_b[pos] = ws.recB[o];
        const float4 c4 = ws.recC[o];
        s_c[pos] = make_float2(c4.x, c4.y);
        s_fid[pos] = f;
      }
      list_n += tot;
      __syncthreads();
      const bool last = (base + TPB >= F);
      if (list_n > CAP - TPB || last) {
        // ---- every pixel walks the candidate list
        if (valid) {
          for (int c = 0; c < list_n; ++c) {
            const float4 bx = s_box[c];
            if (xf > bx.y || xf < bx.x || yf > bx.w || yf < bx.z) continue;
            const float4 A = s_a[c], B = s_b[c];
            const float2 C = s_c[c];
            const float x0 = A.x, y0 = A.y, x1 = A.z, y1 = A.w, x2 = B.x, y2 = B.y;
            const float z0 = B.z, z1 = B.w, z2 = C.x;
            const float denom = C.y + ACFM_K_EPS;
            const float w0 = edge_fn(xf, yf, x1, y1, x2, y2) / denom;
            const float w1 = edge_fn(xf, yf, x2, y2, x0, y0) / denom;
            const float w2 = edge_fn(xf, yf, x0, y0, x1, y1) / denom;
            float c0 = w0, c1 = w1, c2 = w2;
            if (CLIP) {
              c0 = fmaxf(fminf(w0, 1.0f), 0.0f);
              c1 = fmaxf(fminf(w1, 1.0f), 0.0f);
              c2 = fmaxf(fminf(w2, 1.0f), 0.0f);
              const float s = fmaxf(c0 + c1 + c2, 1e-5f);
              c0 = c0 / s; c1 = c1 / s; c2 = c2 / s;
            }
            const float pz = c0 * z0 + c1 * z1 + c2 * z2;
            if (pz < 0.0f) continue;
            const float d01 = point_line_dist(xf, yf, x0, y0, x1, y1);
            const float d02 = point_line_dist(xf, yf, x0, y0, x2, y2);
            const float d12 = point_line_dist(xf, yf, x1, y1, x2, y2);
            const float d = fminf(fminf(d01, d02), d12);
            const bool inside = (w0 > 0.0f) && (w1 > 0.0f) && (w2 > 0.0f);
            if (!inside && d >= blur) continue;
            const float sd = inside ? -d : d;
            // (depth, face) key: pz >= 0 so its bit pattern orders like the float; +0.0f
            // folds -0.0 into +0.0.  Smaller face id wins a depth tie.
            const unsigned long long key =
                ((unsigned long long)__float_as_uint(pz + 0.0f) << 32) | (unsigned)s_fid[c];
            if (K == 1) {
              if (key < bestkey) { bestkey = key; bestsd = sd; bestb0 = c0; bestb1 = c1; }
            } else {
              bool rescan = false;
              if (cnt < K) {
                s_list.key[cnt][tid] = key; s_list.sd[cnt][tid] = sd;
                cnt++;
                rescan = (cnt == K);
              } else if (key < maxkey) {
                s_list.key[maxslot][tid] = key; s_list.sd[maxslot][tid] = sd;
                rescan = true;
              }
              if (rescan) {
                maxkey = 0; maxslot = 0;
                for (int k = 0; k < KL; ++k) {
                  const unsigned long long kk = s_list.key[k][tid];
                  if (kk >= maxkey) { maxkey = kk; maxslot = k; }
                }
              }
            }
          }
        }
        list_n = 0;
        __syncthreads();
      }
    }
  }

  if (!valid) return;
  const int64_t fbase = (int64_t)n * F;
  if (K == 1) {
    const bool hit = (bestkey != ~0ull);
    const int f = (int)(bestkey & 0xffffffffu);
    out.p2f[pix] = hit ? fbase + f : (int64_t)-1;
    if (TEX) {
      // TexturesAtlas.sample_textures + ambient-only Phong + softmax_rgb_blend, K = 1
      // (oracle_atlas_shade is the line-by-line spec)
      const size_t HW = (size_t)H * H;
      float* img = out.imgs + (size_t)n * 3 * HW + (size_t)yi * H + xi;
      if (!hit) {
        img[0] = 0.f; img[HW] = 0.f; img[2 * HW] = 0.f;
        out.sil[pix] = 0.f;
        out.tidx[pix] = -1;
      } else {
        const int R = out.R;
        const float zb = __uint_as_float((unsigned)(bestkey >> 32));
        int ix = (int)(bestb0 * (float)R), iy = (int)(bestb1 * (float)R);
        const bool below = ((bestb0 + bestb1) * (float)R - ((float)ix + (float)iy)) <= 1.0f;
        if (!below) { ix = R - 1 - ix; iy = R - 1 - iy; }
        ix = min(max(ix, 0), R - 1); iy = min(max(iy, 0), R - 1);
        const size_t ti = (((size_t)(fbase + f) * R + iy) * R + ix);
        const float eps = 1e-10f, znear = 1.0f, zfar = 100.0f;
        const float prob = sigmoid_neg(bestsd, sigma);
        const float z_inv = (zfar - zb) / (zfar - znear);
        const float z_inv_max = fmaxf(z_inv, eps);
        const float wnum = prob * expf((z_inv - z_inv_max) / out.gamma);
        const float delta = fmaxf(expf((eps - z_inv_max) / out.gamma), eps);
        const float den = wnum + delta;
        const float* tx3 = out.atlas + ti * 3;
        img[0] = (wnum * tx3[0] + delta * 0.0f) / den;
        img[HW] = (wnum * tx3[1] + delta * 0.0f) / den;
        img[2 * HW] = (wnum * tx3[2] + delta * 0.0f) / den;
        out.sil[pix] = 1.0f - (1.0f - prob);
        out.tidx[pix] = (int32_t)ti;
      }
    }
  } else {
    // rank sort of the (<= K) kept keys, blend in list order
    int64_t* o = out.p2f + pix * K;
    float alpha = 1.0f;
    for (int i = 0; i < cnt; ++i) {
      const unsigned long long ki = s_list.key[i][tid];
      int rank = 0;
      for (int j = 0; j < cnt; ++j) rank += (s_list.key[j][tid] < ki) ? 1 : 0;
      o[rank] = fbase + (int64_t)(ki & 0xffffffffu);
      const float prob = sigmoid_neg(s_list.sd[i][tid], sigma);
      alpha = alpha * (1.0f - prob);
    }
    for (int k = cnt; k < K; ++k) o[k] = -1;
    out.mask[pix] = 1.0f - alpha;
  }
}

// ------------------------------------------------------------------------------- backward
// PointLineDistanceBackward with the clamped t held constant
__device__ __forceinline__ void point_line_dist_bwd(float px, float py, float ax, float ay, float bx,
                                                    float by, float g, float& gax, float& gay,
                                                    float& gbx, float& gby) {
  const float bax = bx - ax, bay = by - ay;
  const float l2 = bax * bax + bay * bay;
  if (l2 <= ACFM_K_EPS) {
    gax = 0.f; gay = 0.f;
    gbx = -2.0f * (px - bx) * g; gby = -2.0f * (py - by) * g;
    return;
  }
  float t = (bax * (px - ax) + bay * (py - ay)) / l2;
  t = fminf(fmaxf(t, 0.0f), 1.0f);
  const float qx = (1.0f - t) * ax + t * bx, qy = (1.0f - t) * ay + t * by;
  const float ex = 2.0f * (qx - px), ey = 2.0f * (qy - py);
  gax = g * (1.0f - t) * ex; gay = g * (1.0f - t) * ey;
  gbx = g * t * ex; gby = g * t * ey;
}

__global__ __launch_bounds__(TPB) void k_sil_bwd(RasterWs ws, const float* __restrict__ mask,
                                                 const int64_t* __restrict__ p2f,
                                                 const float* __restrict__ grad_mask, int V, int F,
                                                 int H, int K, float sigma) {
  extern __shared__ float s_g[];  // [V][2] tile-local vertex gradient
  const int n = blockIdx.y, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int tiles = (H + TILE - 1) / TILE;
  const int ty = blockIdx.x / tiles, tx = blockIdx.x % tiles;
  const int yi = ty * TILE + (wv >> 1) * 8 + (lane >> 3);
  const int xi = tx * TILE + (wv & 1) * 8 + (lane & 7);
  const bool valid = (yi < H) && (xi < H);
  const size_t pix = ((size_t)n * H + yi) * H + xi;

  // skip tiles with nothing to do before touching LDS
  float coef = 0.f;
  bool work = false;
  if (valid) {
    const float g = grad_mask[pix], m = mask[pix];
    // d mask / d sd_k = -(1 - mask) * p_k / sigma   (robust form)
    coef = -g * (1.0f - m) / sigma;
    work = (coef != 0.0f) && (p2f[pix * K] >= 0);
  }
  if (!__syncthreads_or(work)) return;

  for (int i = tid; i < 2 * V; i += TPB) s_g[i] = 0.f;
  __syncthreads();

  if (work) {
    const float yf = pix_to_ndc(H - 1 - yi, H);
    const float xf = pix_to_ndc(H - 1 - xi, H);
    const int64_t fbase = (int64_t)n * F;
    for (int k = 0; k < K; ++k) {
      const int64_t fp = p2f[pix * K + k];
      if (fp < 0) break;
      int f = (int)(fp - fbase);
      f = min(max(f, 0), F - 1);
      const size_t o = (size_t)n * F + f;
      const float4 A = ws.recA[o], B = ws.recB[o], C = ws.recC[o];
      const int4 vi = ws.vidx[o];
      const float x0 = A.x, y0 = A.y, x1 = A.z, y1 = A.w, x2 = B.x, y2 = B.y;
      const float denom = C.y + ACFM_K_EPS;
      const float w0 = edge_fn(xf, yf, x1, y1, x2, y2) / denom;
      const float w1 = edge_fn(xf, yf, x2, y2, x0, y0) / denom;
      const float w2 = edge_fn(xf, yf, x0, y0, x1, y1) / denom;
      const bool inside = (w0 > 0.0f) && (w1 > 0.0f) && (w2 > 0.0f);
      const float d01 = point_line_dist(xf, yf, x0, y0, x1, y1);
      const float d02 = point_line_dist(xf, yf, x0, y0, x2, y2);
      const float d12 = point_line_dist(xf, yf, x1, y1, x2, y2);
      const float d = fminf(fminf(d01, d02), d12);
      const float sd = inside ? -d : d;
      const float gs = coef * sigmoid_neg(sd, sigma);  // dL / d sd
      const float gd = inside ? -gs : gs;              // dL / d d
      float gax, gay, gbx, gby;
      int ia, ib;
      if (d01 <= d02 && d01 <= d12) {
        point_line_dist_bwd(xf, yf, x0, y0, x1, y1, gd, gax, gay, gbx, gby); ia = vi.x; ib = vi.y;
      } else if (d02 <= d01 && d02 <= d12) {
        point_line_dist_bwd(xf, yf, x0, y0, x2, y2, gd, gax, gay, gbx, gby); ia = vi.x; ib = vi.z;
      } else {
        point_line_dist_bwd(xf, yf, x1, y1, x2, y2, gd, gax, gay, gbx, gby); ia = vi.y; ib = vi.z;
      }
      atomicAdd(&s_g[2 * ia], gax); atomicAdd(&s_g[2 * ia + 1], gay);
      atomicAdd(&s_g[2 * ib], gbx); atomicAdd(&s_g[2 * ib + 1], gby);
    }
  }
  __syncthreads();
  float* gout = ws.grad_ndc + (size_t)n * V * 2;
  for (int i = tid; i < 2 * V; i += TPB) {
    const float v = s_g[i];
    if (v != 0.0f) atomicAdd(&gout[i], v);
  }
}

// ------------------------------------------------------------------------------- projection
__global__ __launch_bounds__(TPB) void k_project(const float* __restrict__ verts,
                                                 const float* __restrict__ cams, int V, float offset_z,
                                                 float* __restrict__ proj) {
  const int n = blockIdx.y;
  const int v = blockIdx.x * TPB + threadIdx.x;
  if (v >= V) return;
  const float* x = verts + ((size_t)n * V + v) * 3;
  float* o = proj + ((size_t)n * V + v) * 3;
  project_point(cams + 7 * (size_t)n, x[0], x[1], x[2], offset_z, o[0], o[1], o[2]);
}

// Backward of proj = s * rot(q, X) + (tx, ty, offset_z), q not normalised here:
//   r      = (q0^2 - u.u) X + 2 (u.X) u + 2 q0 (u x X)
//   dL/ds  = g.r ; dL/dt = g.xy ; with G = s g:
//   dL/dq0 = 2 q0 (G.X) + 2 G.(u x X)
//   dL/du  = -2 (G.X) u + 2 (G.u) X + 2 (u.X) G + 2 q0 (X x G)
//   dL/dX  = (q0^2 - u.u) G + 2 (G.u) u + 2 q0 (G x u)
// NDC2 = true: the upstream gradient is grad_ndc [N,V,2] of the rasteriser
// (x_ndc = -x_p, y_ndc = -y_p, no z gradient on the silhouette path).
template <bool NDC2>
__global__ __launch_bounds__(TPB) void k_project_bwd(const float* __restrict__ verts,
                                                     const float* __restrict__ cams,
                                                     const float* __restrict__ gin, int V,
                                                     float* __restrict__ grad_verts,
                                                     float* __restrict__ grad_cams) {
  __shared__ float s_red[4][7];
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* c = cams + 7 * (size_t)n;
  const float s = c[0], q0 = c[3], ux = c[4], uy = c[5], uz = c[6];
  const float uu = ux * ux + uy * uy + uz * uz;
  const float a = q0 * q0 - uu;
  float acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int v = tid; v < V; v += TPB) {
    const float* x = verts + ((size_t)n * V + v) * 3;
    const float X = x[0], Y = x[1], Z = x[2];
    float gx, gy, gz;
    if (NDC2) {
      const float* g = gin + ((size_t)n * V + v) * 2;
      gx = -g[0]; gy = -g[1]; gz = 0.f;
    } else {
      const float* g = gin + ((size_t)n * V + v) * 3;
      gx = g[0]; gy = g[1]; gz = g[2];
    }
    const float uX = ux * X + uy * Y + uz * Z;
    const float cx = uy * Z - uz * Y, cy = uz * X - ux * Z, cz = ux * Y - uy * X;  // u x X
    const float rx = a * X + 2.f * uX * ux + 2.f * q0 * cx;
    const float ry = a * Y + 2.f * uX * uy + 2.f * q0 * cy;
    const float rz = a * Z + 2.f * uX * uz + 2.f * q0 * cz;
    acc[0] += gx * rx + gy * ry + gz * rz;
    acc[1] += gx;
    acc[2] += gy;
    const float Gx = s * gx, Gy = s * gy, Gz = s * gz;
    const float GX = Gx * X + Gy * Y + Gz * Z;
    const float Gu = Gx * ux + Gy * uy + Gz * uz;
    acc[3] += 2.f * q0 * GX + 2.f * (Gx * cx + Gy * cy + Gz * cz);
    const float xg_x = Y * Gz - Z * Gy, xg_y = Z * Gx - X * Gz, xg_z = X * Gy - Y * Gx;  // X x G
    acc[4] += -2.f * GX * ux + 2.f * Gu * X + 2.f * uX * Gx + 2.f * q0 * xg_x;
    acc[5] += -2.f * GX * uy + 2.f * Gu * Y + 2.f * uX * Gy + 2.f * q0 * xg_y;
    acc[6] += -2.f * GX * uz + 2.f * Gu * Z + 2.f * uX * Gz + 2.f * q0 * xg_z;
    if (grad_verts) {
      const float gu_x = Gy * uz - Gz * uy, gu_y = Gz * ux - Gx * uz, gu_z = Gx * uy - Gy * ux;  // G x u
      float* o = grad_verts + ((size_t)n * V + v) * 3;
      o[0] = a * Gx + 2.f * Gu * ux + 2.f * q0 * gu_x;
      o[1] = a * Gy + 2.f * Gu * uy + 2.f * q0 * gu_y;
      o[2] = a * Gz + 2.f * Gu * uz + 2.f * q0 * gu_z;
    }
  }
  if (!grad_cams) return;
#pragma unroll
  for (int i = 0; i < 7; ++i) acc[i] = wave_sum(acc[i]);
  const int w = tid >> 6;
  if ((tid & 63) == 0)
    for (int i = 0; i < 7; ++i) s_red[w][i] = acc[i];
  __syncthreads();
  if (tid < 7) grad_cams[7 * (size_t)n + tid] = s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid];
}

// ------------------------------------------------------------------------------- texture bwd
__global__ void k_tex_bwd(const float* __restrict__ grad_imgs, const int32_t* __restrict__ tidx,
                          size_t HW, size_t total, float* __restrict__ grad_atlas) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int32_t t = tidx[i];
  if (t < 0) return;
  const size_t n = i / HW, p = i % HW;
  const float* g = grad_imgs + n * 3 * HW + p;
  // d rgb / d texel = wnum / (wnum + delta) = 1 in fp32 (wnum >= 0.5, delta = 1e-10)
  atomicAdd(&grad_atlas[(size_t)t * 3 + 0], g[0]);
  atomicAdd(&grad_atlas[(size_t)t * 3 + 1], g[HW]);
  atomicAdd(&grad_atlas[(size_t)t * 3 + 2], g[2 * HW]);
}

// ------------------------------------------------------------------------------- profiling
static bool g_prof_on = false;
static hipEvent_t g_ev[ACFM_PROF_RING][2];
static int g_ev_id[ACFM_PROF_RING];
static int g_ev_n = 0;
static bool g_ev_made = false;

void prof_begin(int id, hipStream_t st) {
  if (!g_prof_on || g_ev_n >= ACFM_PROF_RING) return;
  g_ev_id[g_ev_n] = id;
  (void)hipEventRecord(g_ev[g_ev_n][0], st);
}
void prof_end(hipStream_t st) {
  if (!g_prof_on || g_ev_n >= ACFM_PROF_RING) return;
  (void)hipEventRecord(g_ev[g_ev_n][1], st);
  g_ev_n++;
}

// ------------------------------------------------------------------------------- host side
static int launch_setup(const float* verts, const int64_t* faces, const float* cams, int N, int V,
                        int F, float offset_z, int mode, float blur, const RasterWs& ws,
                        hipStream_t st) {
  const float margin = sqrtf(blur);
  const size_t lds = sizeof(float) * 3 * (size_t)V;
  if (lds > 150 * 1024) return ACFM_E_BADARG;
  ProfScope ps(ACFM_PROF_SETUP, st);
  hipLaunchKernelGGL(k_setup, dim3(N), dim3(TPB), lds, st, verts, faces, cams, V, F, offset_z, mode,
                     margin, ws);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

static bool bad_dims(int N, int V, int F, int H) {
  return N <= 0 || N > 65535 || V <= 0 || F <= 0 || F > ACFM_MAX_FACES || H <= 0 || H > 4096 ||
         (size_t)N * F > 0x7fffffffull;
}

template <int K>
static int launch_sil_fwd(const RasterWs& ws, int N, int F, int H, float blur, float sigma,
                          const FwdOut& out, hipStream_t st) {
  const int tiles = (H + TILE - 1) / TILE;
  ProfScope ps(ACFM_PROF_SIL_FWD, st);
  hipLaunchKernelGGL((k_raster_fwd<K, false, false>), dim3(tiles * tiles, N), dim3(TPB), 0, st, ws, F,
                     H, blur, sigma, out);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

}  // namespace acfm

using namespace acfm;

extern "C" {

int acfm_version(void) { return 1000; }
const char* acfm_arch(void) { return "gfx950"; }

int acfm_prof_enable(int on) {
  if (on && !g_ev_made) {
    for (int i = 0; i < ACFM_PROF_RING; ++i)
      if (hipEventCreate(&g_ev[i][0]) != hipSuccess || hipEventCreate(&g_ev[i][1]) != hipSuccess)
        return ACFM_E_LAUNCH;
    g_ev_made = true;
  }
  g_ev_n = 0;
  g_prof_on = on != 0;
  return ACFM_OK;
}

int acfm_prof_collect(float* ms_host, int* count_host, int n) {
  if (!ms_host || !count_host || n < ACFM_PROF_NKERNELS) return ACFM_E_BADARG;
  for (int i = 0; i < n; ++i) { ms_host[i] = 0.f; count_host[i] = 0; }
  for (int i = 0; i < g_ev_n; ++i) {
    if (hipEventSynchronize(g_ev[i][1]) != hipSuccess) return ACFM_E_LAUNCH;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_ev[i][0], g_ev[i][1]) != hipSuccess) return ACFM_E_LAUNCH;
    ms_host[g_ev_id[i]] += ms;
    count_host[g_ev_id[i]] += 1;
  }
  g_ev_n = 0;
  return ACFM_OK;
}

const char* acfm_prof_name(int id) {
  static const char* names[ACFM_PROF_NKERNELS] = {
      "k_setup", "k_raster_fwd<K,soft>", "k_sil_bwd", "k_project_bwd", "k_raster_fwd<1,tex>",
      "k_raster_fwd<1,hard>", "k_tex_bwd", "k_mask_losses", "k_mask_losses_bwd", "k_visible",
      "k_bds_loss", "k_bds_loss_bwd", "k_project", "", "", ""};
  return (id >= 0 && id < ACFM_PROF_NKERNELS) ? names[id] : "";
}

size_t acfm_raster_workspace_bytes(int N, int V, int F) {
  if (N <= 0 || V <= 0 || F <= 0) return 0;
  return carve_ws(nullptr, N, V, F).bytes;
}

int acfm_project(const float* verts, const float* cams, int N, int V, float offset_z, float* proj,
                 void* stream) {
  if (!verts || !cams || !proj || N <= 0 || N > 65535 || V <= 0) return ACFM_E_BADARG;
  ProfScope ps(ACFM_PROF_PROJECT, (hipStream_t)stream);
  hipLaunchKernelGGL(k_project, dim3((V + TPB - 1) / TPB, N), dim3(TPB), 0, (hipStream_t)stream, verts,
                     cams, V, offset_z, proj);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_project_backward(const float* verts, const float* cams, const float* grad_proj, int N, int V,
                          float* grad_verts, float* grad_cams, void* stream) {
  if (!verts || !cams || !grad_proj || N <= 0 || V <= 0) return ACFM_E_BADARG;
  ProfScope ps(ACFM_PROF_PROJ_BWD, (hipStream_t)stream);
  hipLaunchKernelGGL((k_project_bwd<false>), dim3(N), dim3(TPB), 0, (hipStream_t)stream, verts, cams,
                     grad_proj, V, grad_verts, grad_cams);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_sil_forward(const float* verts_world, const int64_t* faces, const float* cams, int N, int V,
                     int F, int H, int K, float blur_radius, float sigma, float offset_z, float* mask,
                     int64_t* pix_to_face, void* wsp, size_t ws_bytes, void* stream) {
  if (!verts_world || !faces || !cams || !mask || !pix_to_face || !wsp) return ACFM_E_BADARG;
  if (bad_dims(N, V, F, H) || K < 2 || K > ACFM_MAX_K || !(sigma > 0.f) || blur_radius < 0.f)
    return ACFM_E_BADARG;
  const RasterWs ws = carve_ws(wsp, N, V, F);
  if (ws.bytes > ws_bytes) return ACFM_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_setup(verts_world, faces, cams, N, V, F, offset_z, 0, blur_radius, ws, st);
  if (rc) return rc;
  FwdOut out = {};
  out.mask = mask;
  out.p2f = pix_to_face;
  switch (K) {
    case 20: return launch_sil_fwd<20>(ws, N, F, H, blur_radius, sigma, out, st);
    case 10: return launch_sil_fwd<10>(ws, N, F, H, blur_radius, sigma, out, st);
    case 8: return launch_sil_fwd<8>(ws, N, F, H, blur_radius, sigma, out, st);
    case 4: return launch_sil_fwd<4>(ws, N, F, H, blur_radius, sigma, out, st);
    case 2: return launch_sil_fwd<2>(ws, N, F, H, blur_radius, sigma, out, st);
    case 32: return launch_sil_fwd<32>(ws, N, F, H, blur_radius, sigma, out, st);
    default: return ACFM_E_BADARG;  // supported K: 2, 4, 8, 10, 20, 32
  }
}

int acfm_sil_backward(const float* verts_world, const int64_t* faces, const float* cams,
                      const float* mask, const int64_t* pix_to_face, const float* grad_mask, int N,
                      int V, int F, int H, int K, float sigma, float offset_z, float* grad_verts,
                      float* grad_cams, void* wsp, size_t ws_bytes, void* stream) {
  if (!verts_world || !faces || !cams || !mask || !pix_to_face || !grad_mask || !wsp)
    return ACFM_E_BADARG;
  if (bad_dims(N, V, F, H) || K < 1 || K > ACFM_MAX_K || !(sigma > 0.f)) return ACFM_E_BADARG;
  const RasterWs ws = carve_ws(wsp, N, V, F);
  if (ws.bytes > ws_bytes) return ACFM_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_setup(verts_world, faces, cams, N, V, F, offset_z, 0, 0.f, ws, st);
  if (rc) return rc;
  if (hipMemsetAsync(ws.grad_ndc, 0, sizeof(float) * 2 * (size_t)N * V, st) != hipSuccess)
    return ACFM_E_LAUNCH;
  const size_t lds = sizeof(float) * 2 * (size_t)V;
  if (lds > 150 * 1024) return ACFM_E_BADARG;
  const int tiles = (H + TILE - 1) / TILE;
  {
    ProfScope ps(ACFM_PROF_SIL_BWD, st);
    hipLaunchKernelGGL(k_sil_bwd, dim3(tiles * tiles, N), dim3(TPB), lds, st, ws, mask, pix_to_face,
                       grad_mask, V, F, H, K, sigma);
  }
  ACFM_CHECK_LAUNCH();
  if (grad_verts || grad_cams) {
    ProfScope ps(ACFM_PROF_PROJ_BWD, st);
    hipLaunchKernelGGL((k_project_bwd<true>), dim3(N), dim3(TPB), 0, st, verts_world, cams,
                       (const float*)ws.grad_ndc, V, grad_verts, grad_cams);
    ACFM_CHECK_LAUNCH();
  }
  return ACFM_OK;
}

int acfm_hard_raster(const float* verts_proj, const int64_t* faces, int N, int V, int F, int H,
                     int64_t* pix_to_face, void* wsp, size_t ws_bytes, void* stream) {
  if (!verts_proj || !faces || !pix_to_face || !wsp || bad_dims(N, V, F, H)) return ACFM_E_BADARG;
  const RasterWs ws = carve_ws(wsp, N, V, F);
  if (ws.bytes > ws_bytes) return ACFM_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_setup(verts_proj, faces, nullptr, N, V, F, 0.f, 1, 0.f, ws, st);
  if (rc) return rc;
  FwdOut out = {};
  out.p2f = pix_to_face;
  const int tiles = (H + TILE - 1) / TILE;
  ProfScope ps(ACFM_PROF_HARD_FWD, st);
  hipLaunchKernelGGL((k_raster_fwd<1, false, false>), dim3(tiles * tiles, N), dim3(TPB), 0, st, ws, F, H,
                     0.f, 1e-4f, out);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_tex_forward(const float* verts_world, const int64_t* faces, const float* cams,
                     const float* atlas, int N, int V, int F, int H, int R, float sigma, float gamma,
                     float offset_z, float* imgs, float* sil, int64_t* pix_to_face, int32_t* texel_idx,
                     void* wsp, size_t ws_bytes, void* stream) {
  if (!verts_world || !faces || !cams || !atlas || !imgs || !sil || !pix_to_face || !texel_idx || !wsp)
    return ACFM_E_BADARG;
  if (bad_dims(N, V, F, H) || R <= 0 || R > 256 || !(sigma > 0.f) || !(gamma > 0.f)) return ACFM_E_BADARG;
  if ((size_t)N * F * R * R > 0x7fffffffull) return ACFM_E_BADARG;  // texel_idx is int32
  const RasterWs ws = carve_ws(wsp, N, V, F);
  if (ws.bytes > ws_bytes) return ACFM_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_setup(verts_world, faces, cams, N, V, F, offset_z, 0, 0.f, ws, st);
  if (rc) return rc;
  FwdOut out = {};
  out.p2f = pix_to_face;
  out.atlas = atlas; out.imgs = imgs; out.sil = sil; out.tidx = texel_idx; out.R = R; out.gamma = gamma;
  const int tiles = (H + TILE - 1) / TILE;
  ProfScope ps(ACFM_PROF_TEX_FWD, st);
  hipLaunchKernelGGL((k_raster_fwd<1, true, true>), dim3(tiles * tiles, N), dim3(TPB), 0, st, ws, F, H,
                     0.f, sigma, out);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_tex_backward(const float* grad_imgs, const int32_t* texel_idx, int N, int F, int H, int R,
                      float* grad_atlas, void* stream) {
  if (!grad_imgs || !texel_idx || !grad_atlas || N <= 0 || F <= 0 || H <= 0 || R <= 0) return ACFM_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const size_t total = (size_t)N * H * H;
  if (hipMemsetAsync(grad_atlas, 0, sizeof(float) * 3 * (size_t)N * F * R * R, st) != hipSuccess)
    return ACFM_E_LAUNCH;
  ProfScope ps(ACFM_PROF_TEX_BWD, st);
  hipLaunchKernelGGL(k_tex_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, grad_imgs,
                     texel_idx, (size_t)H * H, total, grad_atlas);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

}  // extern "C"
