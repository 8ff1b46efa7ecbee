// Rasterisation kernels for gfx950: face setup, tiled top-K forward (soft silhouette K<=32,
// hard K=1 with optional atlas shading) and the silhouette backward.
//
// Replaces PyTorch3D 0.3.0's rasterize_meshes coarse/fine/backward CUDA kernels and the
// blending shaders as used by multiframe/nnutils/nmr.py:143-200, 224-238 (semantics:
// SURVEY.md App-A; the CPU oracle in oracle/acfm_oracle.c is the bit-level spec).
//
// Design:
//   * k_setup: four workgroups per mesh project the V vertices into LDS (weak-perspective
//     camera, y flip, view transform) and write one 64-byte record per face (blur-expanded box,
//     vertices, depths), the face bitmask of every 16x16 coarse tile and a cost count per 8x8
//     block; k_order sorts every XCD group's (mesh, block) entries heavy-first (counting sort,
//     no atomics) and puts the blocks no face box comes near at the end.
//   * k_raster_fwd / k_sil_bwd share one skeleton: a ONE-WAVE workgroup owns an 8x8 pixel
//     block (four 4x4 blocks, one per 16-lane group).  Faces are binned against the block with
//     wave ballots into an LDS candidate list (deterministic, face-ordered, no barriers); the
//     groups then walk their own sub-lists.  A launch has entries / div workgroups per group:
//     workgroup j renders the entries j, j + stride, .. that have work and, in the forward,
//     first stores the constant outputs of its share of the empty blocks (struct Sched).
//   * forward, K > 1: every lane keeps its pixel's K nearest (depth|face) keys and blend
//     factors SORTED in registers (bubble-through insertion on static register indices, 4-slot
//     blocks no lane's element can enter are skipped): no per-pixel LDS or memory lists, no final
//     sort, the blend product runs over exactly the kept faces; the block's candidates are walked
//     roughly front to back (8 depth classes); ids leave through LDS as 16-byte stores in image
//     order.  128 VGPRs, 4 waves per SIMD.
//   * forward, K = 1 (blur 0): only the winner's edge distances are evaluated; 6-7 waves per SIMD.
//   * backward: the same walk; membership of a face in a pixel's top-K is `key <= kth[pixel]`
//     (kth saved by the forward), gradients of a candidate are summed over the 16 lanes of its
//     group (pair swap, then DPP row shifts on three values per lane) and TWO lanes add them to the
//     candidate's LDS accumulator, which is flushed with one global float atomic per touched
//     coordinate.  23 waves per CU.
//   * forward, K > 1, ACFM_RECORD_COVER: the walk also keeps the nearest COVERING face per pixel (the
//     hard K = 1 render's answer); k_tex_cover shades the texture render of the same geometry from it.
//   * workgroups are dealt so that all blocks of a mesh run on one XCD (its face records stay
//     in that XCD's L2).
//   * k_tex_bwd_faces: the atlas gradient as a per-face gather over the face's box (no global
//     atomics); k_tex_bwd: the scatter form (one atomic per covered pixel and channel).
#include "acfm_common.h"

#include <atomic>
#include <mutex>
#include <type_traits>

namespace acfm {

// Diagnostic event counters (make VARIANT=count EXTRA="-DACFM_DIAG -DACFM_DIAG_COUNT", tools/count_events.py): what a
// launch of the K-nearest forward actually executes -- walk iterations, how many reach each stage, with how many
// live lanes, how many 4-slot blocks of the sorted insertion run.  Not in the shipping build.
#ifdef ACFM_DIAG_COUNT
__device__ unsigned long long g_diag[16];
#define DIAG_ADD(i, v)                                                                              \
  do {                                                                                              \
    const unsigned long long v_ = (unsigned long long)(v);   /* (evaluated by every active lane: v may hold a ballot) */ \
    const unsigned long long m_ = __ballot(true);                                                   \
    if ((int)(threadIdx.x & 63) == (int)__builtin_ctzll(m_)) atomicAdd(&g_diag[i], v_);             \
  } while (0)
#else
#define DIAG_ADD(i, v) do {} while (0)
#endif

// A raster launch has entries / div workgroups per XCD group (see Sched; div = Tune::div of the call): the
// flagged-empty blocks -- 70 % of a 256^2 frame of the bird -- cost no workgroup dispatch of their own.
constexpr int RBLK = 8;       // pixels per block side: one wave64 per block
constexpr int RT = 64;        // threads per raster workgroup = RBLK*RBLK
constexpr int RCAP = 128;     // LDS candidate-list capacity of a block (walked early when it could overflow)
constexpr int TPB = 256;      // threads per workgroup of the per-mesh kernels (setup, projection)
constexpr unsigned long long KEY_NONE = ~0ull;
constexpr int CNT_TILE = 8;   // cost counters per 8x8 pixels (= per raster block)
constexpr int SETUP_LDS_TILES = 4096;  // counters kept in LDS up to 1024x1024 images
constexpr int ENTRY_EMPTY = 1 << 30;   // order entry flag: no face box comes near this block
constexpr int ENTRY_SPLIT = 1 << 29;   // order entry flag: a heavy block, rendered by four workgroups (one per 4x4 pixels)
constexpr int ENTRY_FLAGS = ENTRY_EMPTY | ENTRY_SPLIT;
constexpr int SPLIT_MAX_CLASS = 4;     // ... if their cost class is at most this (>= 80 face boxes)
constexpr int SETUP_LDS_MASK_BYTES = 64 * 1024;  // coarse masks built in LDS up to this size
typedef unsigned short fl_t;  // face ids of one mesh (F <= ACFM_MAX_FACES = 65535)
constexpr int FLCAP = 512;    // LDS face-id list of one wave (faces of its coarse tile, 4096 faces at a time)

// ------------------------------------------------------------------------------- setup
// mode 0: verts are world coordinates -> project with cams, flip y   (nmr.py:145-149)
// mode 1: verts are already projected, no y flip                     (nmr.py:224-238)
// Grid (N, ws.slices): every workgroup projects the mesh's V vertices into LDS (cheap, and it
// keeps the slices independent) and handles one slice of the faces; slice boundaries are multiples
// of 64 faces, so each slice owns whole words of the coarse masks and builds them in LDS without
// talking to the others.  Tile counters are summed into zeroed memory with global atomics, the
// mesh box is left as one box per slice (the raster kernels take the union of the four).
__host__ __device__ __forceinline__ int setup_slice_faces(int F, int slices) {
  return ((F + slices - 1) / slices + 63) / 64 * 64;
}

template <int SLICES>   // face slices per mesh (grid y): 4, 8 or 16 -- a template so that the slice arithmetic stays compile-time
__global__ __launch_bounds__(TPB) void k_setup(const float* __restrict__ verts,
                                               const int64_t* __restrict__ faces,
                                               const float* __restrict__ cams, int V, int F, int H,
                                               float offset_z, int mode, float margin, RasterWs ws,
                                               uint8_t* __restrict__ vis, float* __restrict__ proj_xy) {
  extern __shared__ float s_v[];  // [V][3], then [tiles^2] int counters and this slice's mask words (if they fit)
  __shared__ float s_red[4][4];
  const int n = blockIdx.x, slice = blockIdx.y, tid = threadIdx.x;
  const float* cam = cams ? cams + 7 * (size_t)n : nullptr;
  const int tiles_ = (H + CNT_TILE - 1) / CNT_TILE, tt_ = tiles_ * tiles_;
  const bool lds_cnt = tt_ <= SETUP_LDS_TILES;
  int* s_cnt = reinterpret_cast<int*>(s_v + 3 * V);
  if (lds_cnt)
    for (int i = tid; i < tt_; i += TPB) s_cnt[i] = 0;
  const int q = setup_slice_faces(F, SLICES);
  const int f_lo = slice * q, f_hi = min(F, f_lo + q);
  // coarse face masks: bit f of row (cty, ctx) <=> the box of face f may touch that CTILE x CTILE tile
  const int ctiles_ = (H + CTILE - 1) / CTILE, mwords = 2 * ((F + 63) / 64);  // u32 words per row
  const int rows = ctiles_ * ctiles_;
  const int w_lo = f_lo >> 5, w_n = max(0, min(mwords, (f_lo + q) >> 5) - w_lo);  // this slice's words of a row
  const bool lds_mask = (size_t)rows * w_n * sizeof(unsigned) <= (size_t)SETUP_LDS_MASK_BYTES;
  unsigned* g_mask = ws.cmask + (size_t)n * rows * mwords;              // zeroed by the host if !lds_mask
  unsigned* s_mask = reinterpret_cast<unsigned*>(s_cnt + (lds_cnt ? tt_ : 0));
  if (lds_mask)
    for (int i = tid; i < rows * w_n; i += TPB) s_mask[i] = 0u;
  for (int v = tid; v < V; v += TPB) {
    const float* x = verts + ((size_t)n * V + v) * 3;
    float px, py, pz;
    if (mode == 0) {
      project_point(cam, x[0], x[1], x[2], offset_z, px, py, pz);
      // NeuralRenderer.project_points of the same vertices and cameras (nmr.py:127-129: proj_fn(...)[:, :, :2]) is
      // this very (px, py): handed out on request (AcfmSilExtras.proj_xy) instead of being projected again
      if (proj_xy && slice == 0) { proj_xy[((size_t)n * V + v) * 2] = px; proj_xy[((size_t)n * V + v) * 2 + 1] = py; }
      py = py * -1.0f;
    } else {
      px = x[0]; py = x[1]; pz = x[2];
    }
    px = -px;               // view R = diag(-1, 1, 1)
    pz = pz + ACFM_EYE_Z;   // view T = (0, 0, 2.732)
    s_v[3 * v + 0] = px; s_v[3 * v + 1] = py; s_v[3 * v + 2] = pz;
    if (slice == 0) {
      float* o = ws.ndc + ((size_t)n * V + v) * 3;
      o[0] = px; o[1] = py; o[2] = pz;
      // zeroed here instead of by launches of their own: the visible-vertex bytes the raster kernel
      // marks, and the NDC-gradient scratch the backward accumulates into (k_project_bwd<1> leaves
      // it zeroed again after reading it)
      if (vis) vis[(size_t)n * V + v] = 0;
      ws.grad_ndc[((size_t)n * V + v) * 2] = 0.f;
      ws.grad_ndc[((size_t)n * V + v) * 2 + 1] = 0.f;
      ws.grad_fix[((size_t)n * V + v) * 2] = 0;
      ws.grad_fix[((size_t)n * V + v) * 2 + 1] = 0;
    }
  }
  __syncthreads();
  const float INF = __builtin_inff();
  float bx0 = INF, bx1 = -INF, by0 = INF, by1 = -INF;
  bool big = false;
  auto count_box = [&](int xa, int ya, int xb, int yb) {   // pixel range (clamped to the image) -> +1 on its 8x8 blocks
    xa /= CNT_TILE; ya /= CNT_TILE; xb /= CNT_TILE; yb /= CNT_TILE;
    if ((xb - xa + 1) * (yb - ya + 1) > 256) {
      big = true;  // too many tiles to count one by one: every tile of the mesh gets +1 below
    } else
      for (int ty = ya; ty <= yb; ++ty)
        for (int tx = xa; tx <= xb; ++tx) {
          if (lds_cnt) atomicAdd(&s_cnt[ty * tiles_ + tx], 1);
          else atomicAdd(&ws.tile_cnt[((size_t)n * tiles_ + ty) * tiles_ + tx], 1);
        }
  };
  for (int f = f_lo + tid; f < f_hi; f += TPB) {
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
    // (x1, x2) and (y1, y2) sit in aligned register pairs after the 16-byte LDS reads: operands of the packed fp32 pipe
    FaceRec& r = ws.rec[o];
    r.box = b;
    r.a = make_float4(x0, y0, x1, x2);
    r.b = make_float4(y1, y2, z0, z1);
    {
      const float denom = area + ACFM_K_EPS;
      r.c = make_float4(z2, area, denom, recip_refined(denom));
#if ACFM_REC_EDGES
      // point_line_dist's own operations on (a, b) = (v0, v1), (v0, v2), (v1, v2): bax = bx - ax, l2 = bax bax + bay bay
      const float e01x = x1 - x0, e01y = y1 - y0, e02x = x2 - x0, e02y = y2 - y0, e12x = x2 - x1, e12y = y2 - y1;
      const float l01 = e01x * e01x + e01y * e01y, l02 = e02x * e02x + e02y * e02y, l12 = e12x * e12x + e12y * e12y;
      const bool deg = (l01 <= ACFM_K_EPS) || (l02 <= ACFM_K_EPS) || (l12 <= ACFM_K_EPS);
      r.e0 = make_float4(l01, l02, recip_refined(l01), recip_refined(l02));
      r.e1 = make_float4(l12, recip_refined(l12), deg ? 1.0f : 0.0f, 0.0f);
#endif
    }
    ws.vidx[o] = make_int4(i0, i1, i2, 0);
    ws.fvis[o] = 0;
    if (!degenerate) {
      // pixel index of an NDC coordinate: i = H-1 - ((c+1)H - 1)/2; one pixel of slack
      const float hf = (float)H;
      int xa = (int)floorf(hf - 1.0f - ((b.y + 1.0f) * hf - 1.0f) * 0.5f) - 1;
      int xb = (int)ceilf(hf - 1.0f - ((b.x + 1.0f) * hf - 1.0f) * 0.5f) + 1;
      int ya = (int)floorf(hf - 1.0f - ((b.w + 1.0f) * hf - 1.0f) * 0.5f) - 1;
      int yb = (int)ceilf(hf - 1.0f - ((b.z + 1.0f) * hf - 1.0f) * 0.5f) + 1;
      if (xb >= 0 && yb >= 0 && xa < H && ya < H) {
        xa = max(xa, 0); ya = max(ya, 0); xb = min(xb, H - 1); yb = min(yb, H - 1);
        const unsigned bit = 1u << (f & 31);
        for (int cy = ya / CTILE; cy <= yb / CTILE; ++cy)
          for (int cx = xa / CTILE; cx <= xb / CTILE; ++cx) {
            const int row = cy * ctiles_ + cx;
            if (lds_mask) atomicOr(&s_mask[row * w_n + ((f >> 5) - w_lo)], bit);
            else atomicOr(&g_mask[(size_t)row * mwords + (f >> 5)], bit);
          }
        // cost estimate for heavy-first scheduling: +1 on every 8x8 block the box may touch.  With the counters in LDS
        // (images up to 512^2) every slice counts its own faces and stores its plane of ws.tile_part (k_order adds the
        // four planes): no zero fill, no global atomics; larger images: every slice adds its faces to the zeroed ws.tile_cnt
        count_box(xa, ya, xb, yb);
      }
    }
  }
  bx0 = wave_min(bx0); bx1 = wave_max(bx1); by0 = wave_min(by0); by1 = wave_max(by1);
  const int w = tid >> 6;
  if ((tid & 63) == 0) { s_red[w][0] = bx0; s_red[w][1] = bx1; s_red[w][2] = by0; s_red[w][3] = by1; }
  const int any_big = __syncthreads_or(big) ? 1 : 0;  // (also the barrier before the copies below)
  if (lds_cnt) {
    int* part = ws.tile_part + ((size_t)slice * gridDim.x + n) * tt_;
    for (int i = tid; i < tt_; i += TPB) part[i] = s_cnt[i] + any_big;   // this slice's faces: plain stores
  } else {
    for (int i = tid; i < tt_; i += TPB)                // ws.tile_cnt was zeroed by the host
      if (any_big) atomicAdd(&ws.tile_cnt[(size_t)n * tt_ + i], any_big);
  }
  if (lds_mask)
    for (int i = tid; i < rows * w_n; i += TPB)
      g_mask[(size_t)(i / w_n) * mwords + w_lo + (i % w_n)] = s_mask[i];
  if (tid == 0) {
    for (int i = 1; i < 4; ++i) {
      bx0 = fminf(bx0, s_red[i][0]); bx1 = fmaxf(bx1, s_red[i][1]);
      by0 = fminf(by0, s_red[i][2]); by1 = fmaxf(by1, s_red[i][3]);
    }
    ws.mbox[(size_t)n * SLICES + slice] = make_float4(bx0, bx1, by0, by1);
  }
}

// Zero-fill by a kernel instead of hipMemsetAsync: a memset node in front of k_setup came out
// wrong when the call was captured into a hipGraph and replayed (tests/test_gpu_render.py::
// test_hip_graph_capture_and_replay); kernel nodes replay reliably, so the library uses no memsets.
__global__ void k_zero_bytes(unsigned char* __restrict__ p, size_t nbytes) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nw = nbytes >> 2;
  if (i < nw) reinterpret_cast<unsigned*>(p)[i] = 0u;
  if (i < (nbytes & 3)) p[(nw << 2) + i] = 0;
}
// large 16-byte-aligned buffers (atlas gradients, the solver's identity rows): 16-byte stores, 4 per thread
__global__ __launch_bounds__(256) void k_zero_vec(uint4* __restrict__ p, size_t n16) {
  const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const size_t i = base + 256 * (size_t)k;
    if (i < n16) p[i] = z;
  }
}
int zero_async(void* p, size_t nbytes, hipStream_t st) {
  if (nbytes == 0) return ACFM_OK;
  if (((uintptr_t)p & 3) != 0) return ACFM_E_BADARG;
  if (nbytes >= (1u << 16) && ((uintptr_t)p & 15) == 0) {
    const size_t n16 = nbytes >> 4;
    hipLaunchKernelGGL(k_zero_vec, dim3((unsigned)((n16 + 1023) / 1024)), dim3(256), 0, st, (uint4*)p, n16);
    p = (unsigned char*)p + (n16 << 4);
    nbytes &= 15;
    if (nbytes == 0) return hipGetLastError() == hipSuccess ? ACFM_OK : ACFM_E_LAUNCH;
  }
  const size_t n = (nbytes >> 2) > 4 ? (nbytes >> 2) : 4;
  hipLaunchKernelGGL(k_zero_bytes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (unsigned char*)p, nbytes);
  return hipGetLastError() == hipSuccess ? ACFM_OK : ACFM_E_LAUNCH;
}

// ------------------------------------------------------------------------------- scheduling
// Heavy-first order.  Per-block work is heavy-tailed (dense clusters of tiny faces: a block can
// take 20x the average), so every XCD group visits its (mesh, block) entries in descending cost
// class; the long blocks start first and the short ones fill in behind them.
// Entry e of group g  <->  mesh (e / tt) * G + g, block e % tt   (G = 8 groups if N % 8 == 0, else 1).
// The cost of a block is the face count of its 16x16 tile (k_setup); count 0 = no face box comes
// near: the entry is flagged and the raster kernels write that block's zeros without looking at
// the mesh at all.
constexpr int NCLASS = 10;
// (the classes above 160 exist to ORDER the heaviest blocks: at 64 frames the blocks of >= 240 face boxes -- 89 of
// 15 474, 3 % of the work -- start first and still run for the whole launch; see the split rule in k_order.  Every
// class costs k_order a ballot per entry and pass: 12 classes measured 18.1 us per launch against 13.3 with 8)
__device__ __forceinline__ int cost_class(int c) {
  return c >= 240 ? 0 : c >= 200 ? 1 : c >= 160 ? 2 : c >= 112 ? 3 : c >= 80 ? 4 : c >= 56 ? 5 : c >= 36 ? 6 : c >= 20 ? 7 : c >= 1 ? 8 : 9;
}
__device__ __forceinline__ int block_cost(const RasterWs& ws, int n, int bl, int H) {
  const int blocks = (H + RBLK - 1) / RBLK, tiles = (H + CNT_TILE - 1) / CNT_TILE;
  const int by = bl / blocks, bx = bl % blocks;
  return ws.tile_cnt[((size_t)n * tiles + by * RBLK / CNT_TILE) * tiles + bx * RBLK / CNT_TILE];
}
template <bool MORE>   // MORE: k_setup ran 8 or 16 face slices per mesh (few meshes); false: the usual four
__global__ __launch_bounds__(1024) void k_order(RasterWs ws, int N, int tt, int H, int g_split_dev) {
  // counting sort by cost class without atomics: every wave counts its entries per class (ballots,
  // wave-uniform counters), the counts are prefix-summed over (class, wave), and every wave then
  // scatters its entries from its own running offsets.  Deterministic order.
  constexpr int NW = 16;   // waves of the workgroup
  __shared__ int s_cnt[NCLASS][NW], s_off[NCLASS][NW], s_hist[NCLASS], s_split;
  const int G = gridDim.x, g = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int per = (N / G) * tt;
  const int iters = (per + (int)blockDim.x - 1) / (int)blockDim.x;
  // entry e = m tt + bl  <->  mesh m G + g, block bl; its cost is tile_cnt[mesh][bl] (CNT_TILE == RBLK).
  // (m, bl) advance with e: no integer divisions in the loops (they were 2000 VALU instructions per wave).
  static_assert(CNT_TILE == RBLK, "cost counters are per raster block");
  const int m_first = (int)threadIdx.x / tt, bl_first = (int)threadIdx.x % tt;
  if constexpr (MORE) {
    // the mesh boxes: k_setup left one box per face slice; the raster kernels test a block against the mesh's box: with
    // more than four slices it is joined here, once, into the first slot
    for (int m = threadIdx.x; m < N / G; m += blockDim.x) {
      float4* mb = ws.mbox + (size_t)(m * G + g) * ws.slices;
      float4 u = mb[0];
      for (int i = 1; i < ws.slices; ++i) {
        const float4 m2 = mb[i];
        u.x = fminf(u.x, m2.x); u.y = fmaxf(u.y, m2.y); u.z = fminf(u.z, m2.z); u.w = fmaxf(u.w, m2.w);
      }
      mb[0] = u;
    }
  }
  constexpr int OCH = 8;   // entries per thread whose cost loads are in flight together
  const bool parts = tt <= SETUP_LDS_TILES;   // k_setup kept its counters in LDS: one plane per face slice
  int cnt[NCLASS];
#pragma unroll
  for (int c = 0; c < NCLASS; ++c) cnt[c] = 0;
  int m1 = m_first, bl1 = bl_first;
  int cst0[OCH];          // the costs of the first OCH entries of this thread: all of them up to 8192 entries per group
                          // (64 frames @256^2), so that the scatter pass below does not load them again
  for (int it0 = 0; it0 < iters; it0 += OCH) {
    int cst[OCH];
#pragma unroll
    for (int u = 0; u < OCH; ++u) {
      const int e = (it0 + u) * blockDim.x + threadIdx.x;
      cst[u] = -1;
      if (it0 + u < iters && e < per) {
        const size_t o = (size_t)(m1 * G + g) * tt + bl1;
        if (parts) {   // the four face slices' planes (k_setup); the sum is kept for the later readers (k_tex_cover)
          const size_t plane = (size_t)N * tt;
          cst[u] = (ws.tile_part[o] + ws.tile_part[plane + o]) + (ws.tile_part[2 * plane + o] + ws.tile_part[3 * plane + o]);
          if constexpr (MORE)
            for (int sl = 4; sl < ws.slices; sl += 4)     // 8 or 16 planes
              cst[u] += (ws.tile_part[sl * plane + o] + ws.tile_part[(sl + 1) * plane + o]) +
                        (ws.tile_part[(sl + 2) * plane + o] + ws.tile_part[(sl + 3) * plane + o]);
          ws.tile_cnt[o] = cst[u];
        } else {
          cst[u] = ws.tile_cnt[o];
        }
      }
      bl1 += blockDim.x;
      while (bl1 >= tt) { bl1 -= tt; ++m1; }
      if (it0 == 0) cst0[u] = cst[u];
    }
#pragma unroll
    for (int u = 0; u < OCH; ++u) {
      const int cl = cst[u] < 0 ? -1 : cost_class(cst[u]);
#pragma unroll
      for (int c = 0; c < NCLASS; ++c) cnt[c] += __popcll(__ballot(cl == c));
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < NCLASS; ++c) s_cnt[c][wv] = cnt[c];
  }
  __syncthreads();
  if (threadIdx.x < NCLASS) {   // per class: total, then (below) the offsets of the waves inside the class
    int t = 0;
    for (int w = 0; w < NW; ++w) t += s_cnt[threadIdx.x][w];
    s_hist[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x < NCLASS) {
    int acc = 0;
    for (int c = 0; c < (int)threadIdx.x; ++c) acc += s_hist[c];
    for (int w = 0; w < NW; ++w) { s_off[threadIdx.x][w] = acc; acc += s_cnt[threadIdx.x][w]; }
  }
  if (threadIdx.x == 0) {
    int nw = 0;
    for (int c = 0; c < NCLASS - 1; ++c) nw += s_hist[c];
    ws.n_work[g] = nw;   // the flagged-empty class sits at the end of the order
    // Split the heaviest blocks over four workgroups each?  It adds ~25 % work to those blocks and shortens them about
    // 3x.  A launch lasts at least as long as its longest block (per-block stamps at 64 frames @256^2: the blocks of
    // ~300 face boxes start at t = 0 and end with the kernel, 225 us, while the work spread over the wave slots comes to
    // 195 us), so a block is split when its cost exceeds `ratio` x the group's mean work per wave slot (512 slots per
    // XCD at 16 one-wave workgroups per CU): a whole small launch, the top few dozen blocks of a large one.
    // split_mode < 0: ratio = -split_mode / 4 (default -5: 1.25).
    const int mid[NCLASS] = {290, 220, 180, 136, 96, 68, 46, 28, 10, 0};
    long total = 0;
    for (int c = 0; c < NCLASS; ++c) total += (long)s_hist[c] * mid[c];
    int max_class = -1;                      // split the classes 0 .. max_class
    if (ws.split_slots > 0) {
      if (g_split_dev > 0) max_class = SPLIT_MAX_CLASS;
      else if (g_split_dev < 0)
        for (int c = 0; c <= SPLIT_MAX_CLASS; ++c)
          if (4L * mid[c] * 512 > (long)(-g_split_dev) * total) max_class = c;
    }
    s_split = max_class;
  }
  __syncthreads();
  const int split_slots = ws.split_slots, split_class = s_split;
  // pass 2: scatter (order inside a class is arbitrary: results never depend on it)
  int off[NCLASS];
#pragma unroll
  for (int c = 0; c < NCLASS; ++c) off[c] = s_off[c][wv];
  const unsigned long long lt = (1ull << lane) - 1ull;
  int* ord = ws.order + (size_t)g * per;
  m1 = m_first; bl1 = bl_first;
  for (int it0 = 0; it0 < iters; it0 += OCH) {
    int cst[OCH];
#pragma unroll
    for (int u = 0; u < OCH; ++u) {
      const int e = (it0 + u) * blockDim.x + threadIdx.x;
      if (it0 == 0) cst[u] = cst0[u];
      else cst[u] = (it0 + u < iters && e < per) ? ws.tile_cnt[(size_t)(m1 * G + g) * tt + bl1] : -1;
      bl1 += blockDim.x;
      while (bl1 >= tt) { bl1 -= tt; ++m1; }
    }
#pragma unroll
    for (int u = 0; u < OCH; ++u) {
      const int e = (it0 + u) * blockDim.x + threadIdx.x;
      const int cls = cst[u] < 0 ? -1 : cost_class(cst[u]);
#pragma unroll
      for (int c = 0; c < NCLASS; ++c) {
        const unsigned long long m = __ballot(cls == c);
        if (cls == c) {
          const int pos = off[c] + __popcll(m & lt);
          ord[pos] = e | (c == NCLASS - 1 ? ENTRY_EMPTY : 0) | ((c <= split_class && pos < split_slots) ? ENTRY_SPLIT : 0);
        }
        off[c] += __popcll(m);
      }
    }
  }
}

// ------------------------------------------------------------------------------- tile skeleton
struct Tile {
  int n, tid, wv, lane, yi, xi;
  bool valid, empty, none;  // none: nothing to do for this workgroup
  int sub;                   // >= 0: split role, this wave renders 4x4 pixels (group `sub` of the block) with its four
                             // 16-lane groups taking every fourth candidate each; -1: the whole 8x8 block
  float xf, yf;
  size_t pix;
  float t_xmin, t_xmax, t_ymin, t_ymax;
};

// Workgroup (= one wave) -> entries of the heavy-first order of its XCD group.
// Workgroups are dealt round-robin over the 8 XCDs, so with N % 8 == 0 group g = b % 8 owns the
// meshes n % 8 == g (one XCD's L2 then holds the records of the meshes it renders).  Pure speed:
// any mapping gives the same result.
// A group of `per` entries gets stride = ceil(per / div) workgroups (+ 4 per split slot in front).
// Workgroup j renders the entries j, j + stride, ... that have work (e < n_work: normally just
// one, the order is heavy-first and most of a frame is empty) and, in the forward kernels, first
// stores the constant outputs of the flagged-empty entries n_work + j, n_work + j + stride, ...:
// those stores drain while the block is rendered, and an empty block costs no workgroup dispatch
// (65 536 one-wave workgroups per 64 frames took ~90 us of the chip's dispatcher by themselves).
struct Sched {
  int G, g, per, j0, stride, e_end, n_work, sub;
};
__device__ __forceinline__ Sched make_sched(const RasterWs& ws, int N, int H, bool with_split) {
  Sched s;
  const int tiles = (H + RBLK - 1) / RBLK;
  const int tt = tiles * tiles;
  s.G = (N & 7) == 0 ? 8 : 1;
  s.per = (N / s.G) * tt;
  const int nsplit = with_split ? s.G * ws.split_slots * 4 : 0;
  int b = (int)blockIdx.x;
  s.sub = -1;
  if (b < nsplit) {          // split role: workgroups 4 slot .. 4 slot + 3 of a group take the four 4x4 groups of entry `slot`
    s.g = s.G == 8 ? (b & 7) : 0;
    const int q = s.G == 8 ? (b >> 3) : b;
    s.j0 = q >> 2;
    s.sub = q & 3;
    s.stride = 1;
    s.n_work = ws.n_work[s.g];
    s.e_end = min(s.j0 + 1, s.n_work);
  } else {
    b -= nsplit;
    s.g = s.G == 8 ? (b & 7) : 0;
    s.j0 = s.G == 8 ? (b >> 3) : b;
    s.stride = ((int)gridDim.x - nsplit) / s.G;
    s.n_work = ws.n_work[s.g];
    s.e_end = s.n_work;
  }
  return s;
}

// order entry -> (mesh, first pixel row, first pixel column) of its 8x8 block
__device__ __forceinline__ void entry_block(int eo, const Sched& s, int H, int& n, int& by, int& bx) {
  const int tiles = (H + RBLK - 1) / RBLK;
  const int tt = tiles * tiles;
  const int e = eo & ~ENTRY_FLAGS;
  n = (e / tt) * s.G + s.g;
  const int tl = e % tt;
  by = (tl / tiles) * RBLK; bx = (tl % tiles) * RBLK;
}

__device__ __forceinline__ Tile make_tile(const RasterWs& ws, const Sched& s, int j, int N, int H, bool with_split) {
  Tile t;
  const int tiles = (H + RBLK - 1) / RBLK;
  const int tt = tiles * tiles;
  const int G = s.G, g = s.g, per = s.per;
  t.none = false;
  t.sub = s.sub;
  const int eo = ws.order[(size_t)g * per + j];
  if (t.sub >= 0 && !(eo & ENTRY_SPLIT)) t.none = true;
  if (t.sub < 0 && with_split && (eo & ENTRY_SPLIT)) t.none = true;   // rendered by its four split workgroups
  const int e = eo & ~ENTRY_FLAGS;
  t.empty = (eo & ENTRY_EMPTY) != 0;
  const int n = (e / tt) * G + g, tl = e % tt;
  t.n = n;
  // (opaque to the optimiser: lane-derived addressing stays inside the per-block code instead of
  // being hoisted out of the workgroup's block loop into long-lived registers)
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  t.tid = tid_; t.wv = 0; t.lane = t.tid & 63;
  const int ty = tl / tiles, tx = tl % tiles;
  // the 8x8 pixels = four 4x4 blocks, one per 16-lane group (= one DPP row); split role: all four
  // 16-lane groups hold the same 4x4 pixels
  const int grp = t.sub >= 0 ? t.sub : (t.lane >> 4), jj = t.lane & 15;
  t.yi = ty * RBLK + (grp >> 1) * 4 + (jj >> 2);
  t.xi = tx * RBLK + (grp & 1) * 4 + (jj & 3);
  t.valid = (t.yi < H) && (t.xi < H);
  t.yf = pix_to_ndc(H - 1 - t.yi, H);
  t.xf = pix_to_ndc(H - 1 - t.xi, H);
  t.pix = ((size_t)n * H + t.yi) * H + t.xi;
  // extent of the block (split role: of the 4x4 group) in NDC (pixel centres; x/y decrease with the pixel index)
  const int ex0 = tx * RBLK + (t.sub >= 0 ? (t.sub & 1) * 4 : 0), ey0 = ty * RBLK + (t.sub >= 0 ? (t.sub >> 1) * 4 : 0);
  const int ext = t.sub >= 0 ? 3 : RBLK - 1;
  t.t_xmax = pix_to_ndc(H - 1 - ex0, H); t.t_xmin = pix_to_ndc(H - 1 - (ex0 + ext), H);
  t.t_ymax = pix_to_ndc(H - 1 - ey0, H); t.t_ymin = pix_to_ndc(H - 1 - (ey0 + ext), H);
  return t;
}

template <int CAP_>
struct CandListT {
  static constexpr int CAP = CAP_;
  float4 box[CAP_], a[CAP_], b[CAP_];
  float4 c[CAP_];    // (z2, denom = area + kEps, refined 1 / denom, face id as bits)
  unsigned char sub[4][CAP_];  // per 16-lane group: candidates meeting its 4x4 pixels (list positions < CAP <= 256)
};


struct Cand {
  float4 box, a, b;
  float2 c;      // (z2, denom = area + kEps)
  float rden;    // refined 1 / denom (k_setup's, the very operations the per-pixel code used to repeat)
  int fid, idx;  // idx: list position (the ACFM_EDGE_CONST walk reads L.e0 / L.e1[idx] when it reaches the distance stage)
};

template <class LT, class = void> struct has_edge_const : std::false_type {};
template <class LT> struct has_edge_const<LT, std::void_t<decltype(std::declval<LT&>().e0)>> : std::true_type {};

template <class LT>
__device__ __forceinline__ Cand load_cand(const LT& L, int i) {
  Cand r;
  r.box = L.box[i]; r.a = L.a[i]; r.b = L.b[i];
  const float4 c = L.c[i];   // one 16-byte read
  r.c = make_float2(c.x, c.y); r.rden = c.z; r.fid = __float_as_int(c.w); r.idx = i;
  return r;
}

// Which of four pixel blocks (centres (cx0|cx1, cy0|cy1), half extents hx, hy) lie entirely
// farther than r outside one edge line of the triangle (a = x0 y0 x1 y1, b = x2 y2 ..): no pixel
// of such a block is inside the face or within r of it, so the pair can be dropped.  Conservative
// (separating axis on the three edge normals only); r carries a 1e-3 slack over sqrt(blur), far
// above the rounding of the per-pixel distance.  bit0 x0y0, bit1 x1y0, bit2 x0y1, bit3 x1y1.
__device__ __forceinline__ unsigned edge_cull4(const float4 a, const float4 b, float cx0, float cx1,
                                               float cy0, float cy1, float hx, float hy, float r) {
  const float px[3] = {a.x, a.z, a.w}, py[3] = {a.y, b.x, b.y};
  unsigned out = 0u;
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const int q = (e + 1) % 3, o = (e + 2) % 3;
    float nx = py[q] - py[e], ny = px[e] - px[q];
    const float side = nx * (px[o] - px[e]) + ny * (py[o] - py[e]);
    if (side > 0.f) { nx = -nx; ny = -ny; }  // outward: away from the third vertex
    const float thr2 = r * r * (nx * nx + ny * ny);
    const float ext = fabsf(nx) * hx + fabsf(ny) * hy;
    const float ax0 = nx * (cx0 - px[e]) - ext, ax1 = nx * (cx1 - px[e]) - ext;
    const float ay0 = ny * (cy0 - py[e]), ay1 = ny * (cy1 - py[e]);
    float m;
    m = ax0 + ay0; if (m > 0.f && m * m > thr2) out |= 1u;
    m = ax1 + ay0; if (m > 0.f && m * m > thr2) out |= 2u;
    m = ax0 + ay1; if (m > 0.f && m * m > thr2) out |= 4u;
    m = ax1 + ay1; if (m > 0.f && m * m > thr2) out |= 8u;
  }
  return out;
}

// Second-level cull + walk.  A wave covers 8x8 pixels as four 4x4 blocks, one per 16-lane
// group.  (A) 64 candidates at a time, lane i tests candidate i's box against each of the four
// blocks and the survivors are compacted (one ballot per group) into that group's own sub-list;
// with EDGE_CULL in two steps: (A1) box of the 8x8 block -> the wave's list, (A2) per 4x4 block the
// box and then the edge-line test above, which drops ~17 % of the pairs the boxes keep.  (B) the four groups then walk THEIR OWN lists side by side -- in one iteration the
// groups work on four different faces -- which keeps more lanes busy than walking the union of
// the lists (a 4x4 block meets ~25 faces, the 8x8 block ~41).  body(cand, in_box, ordinal) runs
// for every lane; in_box = the lane has a face this iteration and its pixel is inside the face's box.
// EDGE_CULL pays for itself only where a kept pair is expensive (the K-nearest forward walk:
// -4.5 %); the backward and nearest-face walks drop most pairs on a cheap key / depth compare and
// were measured slower with it (+7 %, +2 %).
// SHARE (the backward): the four sub-lists are of different lengths (50 of 64 lanes have a face in an average
// iteration), and a group that has run out used to idle until the longest list ended.  The groups are paired --
// longest with shortest, the two middle ones -- and the shorter group of a pair, once through its own list, takes the
// faces of its partner's list from the END, for the PARTNER's pixels (lane j of the helper stands in for lane j of the
// partner: same face for the 16 lanes of a row, so the row reduction of the gradient is unchanged); the pair then needs
// ceil((n_A + n_B) / 2) iterations instead of n_A.  prep(partner_lane) is called once per walk with the lane whose
// pixel this lane takes over when it helps (-1: never); body gets (cand, in_box, ordinal, helping, xf, yf).
struct NoPrep { __device__ __forceinline__ void operator()(int) const {} };
template <bool EDGE_CULL, bool SHARE = false, class LT, class Body, class Prep = NoPrep>
__device__ __forceinline__ void walk_wave(LT& L, const Tile& t, int H, int list_n, float blur,
                                          unsigned char* wl /* [2 CAP], EDGE_CULL only */,
                                          Body&& body, Prep&& prep = NoPrep()) {
  const int by = (t.yi & ~7), bx = (t.xi & ~7);
  const int grp = t.lane >> 4;
  // NDC extents (pixel centres) of the four 4x4 blocks: x by column pair, y by row pair
  const float xa0 = pix_to_ndc(H - 1 - bx, H), xi0 = pix_to_ndc(H - 1 - (bx + 3), H);
  const float xa1 = pix_to_ndc(H - 1 - (bx + 4), H), xi1 = pix_to_ndc(H - 1 - (bx + 7), H);
  const float ya0 = pix_to_ndc(H - 1 - by, H), yi0 = pix_to_ndc(H - 1 - (by + 3), H);
  const float ya1 = pix_to_ndc(H - 1 - (by + 4), H), yi1 = pix_to_ndc(H - 1 - (by + 7), H);
  static_assert(LT::CAP <= 256, "sub-list entries are bytes");
  unsigned char* sub0 = L.sub[0];
  const unsigned long long lt = (1ull << t.lane) - 1ull;
  int n0 = 0, n1 = 0, n2 = 0, n3 = 0;
  const bool split = t.sub >= 0;
  if (split) {
    // split role: the list was binned against this 4x4 group already; 16-lane group s takes the
    // candidates s, s+4, s+8, ... (no sub-lists)
    n0 = (list_n + 3) >> 2; n1 = (list_n + 2) >> 2; n2 = (list_n + 1) >> 2; n3 = list_n >> 2;
  } else if constexpr (!EDGE_CULL) {
    // one pass: lane i tests candidate i's box against the four 4x4 blocks
    for (int base = 0; base < list_n; base += 64) {
      const int c = base + t.lane;
      bool hx0 = false, hx1 = false, hy0 = false, hy1 = false;
      if (c < list_n) {
        const float4 b = L.box[c];
        hx0 = !((xi0 > b.y) | (xa0 < b.x)); hx1 = !((xi1 > b.y) | (xa1 < b.x));
        hy0 = !((yi0 > b.w) | (ya0 < b.z)); hy1 = !((yi1 > b.w) | (ya1 < b.z));
      }
      const unsigned long long b0 = __ballot(hx0 & hy0), b1 = __ballot(hx1 & hy0);
      const unsigned long long b2 = __ballot(hx0 & hy1), b3 = __ballot(hx1 & hy1);
      if (hx0 & hy0) sub0[0 * LT::CAP + n0 + __popcll(b0 & lt)] = (unsigned char)c;
      if (hx1 & hy0) sub0[1 * LT::CAP + n1 + __popcll(b1 & lt)] = (unsigned char)c;
      if (hx0 & hy1) sub0[2 * LT::CAP + n2 + __popcll(b2 & lt)] = (unsigned char)c;
      if (hx1 & hy1) sub0[3 * LT::CAP + n3 + __popcll(b3 & lt)] = (unsigned char)c;
      n0 += __popcll(b0); n1 += __popcll(b1); n2 += __popcll(b2); n3 += __popcll(b3);
    }
  } else {
    // (A1) box of the whole 8x8 block -> the wave's list
    int nw = 0;
    for (int base = 0; base < list_n; base += 64) {
      const int c = base + t.lane;
      bool hit = false;
      if (c < list_n) {
        const float4 b = L.box[c];
        hit = !((xi1 > b.y) | (xa0 < b.x) | (yi1 > b.w) | (ya0 < b.z));
      }
      const unsigned long long bw = __ballot(hit);
      if (hit) wl[nw + __popcll(bw & lt)] = (unsigned char)c;
      nw += __popcll(bw);
    }
    if (nw == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // (A1') front to back, roughly: the wave's list is dealt into 8 depth classes (nearest vertex of
    // the face, classes between the block's nearest and farthest) by a counting sort with ballots.
    // Results never depend on the order of the candidates, the cost of the walk does: a face that
    // arrives after the nearer ones enters a pixel's sorted list near its end, the bubble insertion
    // skips the leading 4-slot blocks it cannot touch, and once a list is full the faces behind it
    // fail the depth test before their edge distances are computed.
    if (nw > 8) {
      static_assert(LT::CAP <= 128, "two list entries per lane");
      const int i0 = t.lane, i1 = t.lane + 64;
      const int c0 = i0 < nw ? (int)wl[i0] : 0, c1 = i1 < nw ? (int)wl[i1] : 0;
      const float INF = __builtin_inff();
      const float z0 = i0 < nw ? min3f(L.b[c0].z, L.b[c0].w, L.c[c0].x) : INF;
      const float z1 = i1 < nw ? min3f(L.b[c1].z, L.b[c1].w, L.c[c1].x) : INF;
      const float zlo = wave_min(fminf(z0, z1));
      const float zhi = wave_max(fmaxf(i0 < nw ? z0 : -INF, i1 < nw ? z1 : -INF));
      const float sc = 8.0f / fmaxf(zhi - zlo, 1e-12f);
      const int b0 = i0 < nw ? min(7, (int)((z0 - zlo) * sc)) : -1;
      const int b1 = i1 < nw ? min(7, (int)((z1 - zlo) * sc)) : -1;
      unsigned char* wl2 = wl + LT::CAP;
      int base = 0;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const unsigned long long m0 = __ballot(b0 == c), m1 = __ballot(b1 == c);
        if (b0 == c) wl2[base + __popcll(m0 & lt)] = (unsigned char)c0;
        if (b1 == c) wl2[base + __popcll(m0) + __popcll(m1 & lt)] = (unsigned char)c1;
        base += __popcll(m0) + __popcll(m1);
      }
      wl = wl2;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // (A2) per 4x4 block: box, then the edge-line test
    const float cx0 = 0.5f * (xa0 + xi0), cx1 = 0.5f * (xa1 + xi1);
    const float cy0 = 0.5f * (ya0 + yi0), cy1 = 0.5f * (ya1 + yi1);
    const float hx = 0.5f * (xa0 - xi0), hy = 0.5f * (ya0 - yi0);
    const float r_cull = sqrtf(blur) * 1.001f;
    for (int base = 0; base < nw; base += 64) {
      const int i = base + t.lane;
      bool k0 = false, k1 = false, k2 = false, k3 = false;
      int c = 0;
      if (i < nw) {
        c = wl[i];
        const float4 b = L.box[c];
        const bool hx0 = !((xi0 > b.y) | (xa0 < b.x)), hx1 = !((xi1 > b.y) | (xa1 < b.x));
        const bool hy0 = !((yi0 > b.w) | (ya0 < b.z)), hy1 = !((yi1 > b.w) | (ya1 < b.z));
        const unsigned far = edge_cull4(L.a[c], L.b[c], cx0, cx1, cy0, cy1, hx, hy, r_cull);
        k0 = hx0 & hy0 & !(far & 1u); k1 = hx1 & hy0 & !(far & 2u);
        k2 = hx0 & hy1 & !(far & 4u); k3 = hx1 & hy1 & !(far & 8u);
      }
      const unsigned long long b0 = __ballot(k0), b1 = __ballot(k1);
      const unsigned long long b2 = __ballot(k2), b3 = __ballot(k3);
      if (k0) sub0[0 * LT::CAP + n0 + __popcll(b0 & lt)] = (unsigned char)c;
      if (k1) sub0[1 * LT::CAP + n1 + __popcll(b1 & lt)] = (unsigned char)c;
      if (k2) sub0[2 * LT::CAP + n2 + __popcll(b2 & lt)] = (unsigned char)c;
      if (k3) sub0[3 * LT::CAP + n3 + __popcll(b3 & lt)] = (unsigned char)c;
      n0 += __popcll(b0); n1 += __popcll(b1); n2 += __popcll(b2); n3 += __popcll(b3);
    }
  }
  const int n_max = max(max(n0, n1), max(n2, n3));
  if (n_max == 0) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int my_n = (grp == 0) ? n0 : (grp == 1) ? n1 : (grp == 2) ? n2 : n3;
  const unsigned char* sub = sub0 + grp * LT::CAP;
  if constexpr (SHARE) {
    if (!split) {
      // pairs (wave-uniform): a = the longest list, d = the shortest, b >= c the other two
      int a = 0, na = n0;
      if (n1 > na) { a = 1; na = n1; }
      if (n2 > na) { a = 2; na = n2; }
      if (n3 > na) { a = 3; na = n3; }
      int d = a == 0 ? 1 : 0, nd = a == 0 ? n1 : n0;
      if (a != 1 && n1 < nd) { d = 1; nd = n1; }
      if (a != 2 && n2 < nd) { d = 2; nd = n2; }
      if (a != 3 && n3 < nd) { d = 3; nd = n3; }
      int b = -1, c = -1;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (g != a && g != d) { if (b < 0) b = g; else c = g; }
      int nb = b == 0 ? n0 : b == 1 ? n1 : b == 2 ? n2 : n3, nc = c == 0 ? n0 : c == 1 ? n1 : c == 2 ? n2 : n3;
      if (nc > nb) { const int tg = b; b = c; c = tg; const int tn = nb; nb = nc; nc = tn; }
      const int xa = (na + nd + 1) >> 1, xb = (nb + nc + 1) >> 1;   // the longer group of a pair keeps its first x faces
      // per lane: own faces [0, own_n), then faces [h0, h0 + h_n) of group `pg`'s list for that group's pixels
      const int own_n = grp == a ? xa : grp == b ? xb : my_n;
      const int pg = grp == d ? a : grp == c ? b : -1;
      const int h0 = grp == d ? xa : xb;
      const int h_n = grp == d ? na - xa : grp == c ? nb - xb : 0;
      const int n_it = max(xa, xb);
      const int jj = t.lane & 15;
      prep(pg >= 0 ? pg * 16 + jj : -1);
      // the partner's pixel (4x4 block pg of the 8x8 block, same position inside it)
      const int pyi = by + ((pg >= 0 ? pg : grp) >> 1) * 4 + (jj >> 2), pxi = bx + ((pg >= 0 ? pg : grp) & 1) * 4 + (jj & 3);
      const float pxf = pix_to_ndc(H - 1 - pxi, H), pyf = pix_to_ndc(H - 1 - pyi, H);
      const bool pvalid = (pyi < H) && (pxi < H);
      const unsigned char* hsub = sub0 + (pg >= 0 ? pg : grp) * LT::CAP + h0;
      for (int i = 0; i < n_it; ++i) {
        const bool own = i < own_n;
        const bool help = !own && (i - own_n) < h_n;
        const int ci = own ? (int)sub[i] : (help ? (int)hsub[i - own_n] : 0);
        const Cand cur = load_cand(L, ci);
        const float exf = help ? pxf : t.xf, eyf = help ? pyf : t.yf;
        const bool have = own || (help && pvalid);
        const bool in_box = have && !((exf > cur.box.y) | (exf < cur.box.x) | (eyf > cur.box.w) | (eyf < cur.box.z));
        DIAG_ADD(3, 1); DIAG_ADD(4, __popcll(__ballot(in_box))); DIAG_ADD(10, __popcll(__ballot(have)));
        DIAG_ADD(12, __popcll(__ballot(have) & 0x0001000100010001ull));
        body(cur, in_box, i, help, exf, eyf);
      }
      return;
    }
  }
  // a group that has run out of faces (or has none) keeps loading its last (or the tile's
  // first) record: harmless, the lanes are masked by `have`.  (Prefetching the next record one
  // iteration ahead was measured: +16 VGPRs, no change in time.)
  const int last = max(my_n - 1, 0);
#ifndef ACFM_WALK_PREFETCH
#define ACFM_WALK_PREFETCH 0   // measured (64 frames, A/B on one box, twice): 192.7 / 193.4 us with it, 189.9 / 192.5 without
#endif
#if ACFM_WALK_PREFETCH
  // the list position of the NEXT iteration's candidate is read one iteration ahead: the walk's dependent chain per
  // iteration is then one LDS round trip (the record) instead of two (sub-list byte -> record)
  int nxt = my_n > 0 ? (split ? grp : (int)sub[0]) : 0;
#endif
  if constexpr (SHARE) prep(-1);
  for (int i = 0; i < n_max; ++i) {
#if ACFM_WALK_PREFETCH
    const int ci = nxt;
    {
      const int l2 = min(i + 1, last);
      nxt = my_n > 0 ? (split ? 4 * l2 + grp : (int)sub[l2]) : 0;
    }
    const Cand cur = load_cand(L, ci);
#else
    const int li = min(i, last);
    const Cand cur = load_cand(L, my_n > 0 ? (split ? 4 * li + grp : (int)sub[li]) : 0);
#endif
    const bool have = i < my_n;
    const bool in_box = have &&
        !((t.xf > cur.box.y) | (t.xf < cur.box.x) | (t.yf > cur.box.w) | (t.yf < cur.box.z));
    DIAG_ADD(3, 1); DIAG_ADD(4, __popcll(__ballot(in_box))); DIAG_ADD(10, __popcll(__ballot(have)));
    DIAG_ADD(12, __popcll(__ballot(have) & 0x0001000100010001ull));   // (group, face) pairs walked
    if constexpr (SHARE) body(cur, in_box, i, false, t.xf, t.yf);
    else body(cur, in_box, i);
  }
}

// Bins the faces of mesh t.n against the wave's 8x8 block and calls walk(count) whenever the LDS
// list is complete or could overflow.  The faces come from the bitmask of the block's 16x16
// coarse tile (k_setup): lane i expands mask word i into the wave's face-id list (ascending), then
// 64 ids per round lane i tests face i's box; survivors are compacted with one ballot, face
// order kept.  No barriers: the workgroup is one wave.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int wave_inclusive_scan(int x, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  return x;
}

// ACFM_MBOX_TEST: the mesh-box early-out in front of the mask loads.  It looks redundant beside k_order's empty flag,
// but the flag comes from counts per 16x16 pixels: without the test the K = 20 kernels measured 6-7 us slower each
// (blocks next to the mesh that bin a mask row to find nothing), the K = 1 kernel 2 us faster.
#ifndef ACFM_MBOX_TEST
#define ACFM_MBOX_TEST 1
#endif
template <class LT, class Walk>
__device__ __forceinline__ void bin_and_walk(const RasterWs& ws, const Tile& t, int F, int H, LT& L,
                                             fl_t* s_fl /* [FLCAP] */, float box_shrink, Walk&& walk) {
  if (t.empty) return;  // flagged by k_order: no face box near this block
#if ACFM_MBOX_TEST
  {
  // the mesh's box: the four slices' boxes (the usual case), or slot 0 where k_order joined 8 or 16 of them
  float4 mb = ws.mbox[(size_t)t.n * ws.slices];
  if (ws.slices == 4) {
#pragma unroll
    for (int i = 1; i < 4; ++i) {
      const float4 m2 = ws.mbox[(size_t)t.n * 4 + i];
      mb.x = fminf(mb.x, m2.x); mb.y = fmaxf(mb.y, m2.y); mb.z = fminf(mb.z, m2.z); mb.w = fmaxf(mb.w, m2.w);
    }
  }
  if (t.t_xmin > mb.y || t.t_xmax < mb.x || t.t_ymin > mb.w || t.t_ymax < mb.z) return;
  }
#endif
  const unsigned long long lt = (1ull << t.lane) - 1ull;
  const int ctiles = (H + CTILE - 1) / CTILE, words = (F + 63) / 64;
  const int cty = (t.yi & ~7) / CTILE, ctx = (t.xi & ~7) / CTILE;
  const unsigned long long* mrow = reinterpret_cast<const unsigned long long*>(ws.cmask) +
                                   ((size_t)t.n * ctiles * ctiles + (size_t)cty * ctiles + ctx) * words;
  int list_n = 0;
  // one loop with a single walk() site (the walk body is large and holds the per-pixel lists in
  // registers: a second inlined copy costs registers): refill the id list from the next mask
  // chunk when it runs dry, take up to 64 ids, test + append, walk when the list could overflow
  // or everything has been appended
  int w0 = -64, total = 0, i0 = 0, f0 = 0, fe = 0;
  bool direct = false, done = false;
#pragma unroll 1
  for (;;) {
#pragma unroll 1
    while (!done && !(direct ? f0 < fe : i0 < total)) {
      w0 += 64;
      if (w0 >= words) { done = true; break; }
      unsigned long long m = (w0 + t.lane < words) ? mrow[w0 + t.lane] : 0ull;
      const int cnt = __popcll(m);
      const int incl = wave_inclusive_scan(cnt, t.lane);
      total = __builtin_amdgcn_readlane(incl, 63);
      i0 = 0;
      direct = total > FLCAP;  // a coarse tile crowded beyond the id list: test these 4096 faces directly
      if (direct) {
        f0 = w0 * 64; fe = min(F, (w0 + 64) * 64); total = 0;
      } else if (total > 0) {
        int pos = incl - cnt;
        const int fbase = (w0 + t.lane) * 64;
#pragma unroll 1
        while (m != 0ull) {
          s_fl[pos++] = (fl_t)(fbase + (int)__ffsll((long long)m) - 1);
          m &= m - 1ull;
        }
        wave_lds_sync();
      }
    }
    if (!done) {
      int f;
      if (direct) { f = (f0 + t.lane < fe) ? f0 + t.lane : -1; f0 += RT; }
      else { f = (i0 + t.lane < total) ? (int)s_fl[i0 + t.lane] : -1; i0 += RT; }
      bool pass = false;
      float4 b = make_float4(0, 0, 0, 0);
      if (f >= 0) {
        // (the box first, the rest of the record only for a face that passes: loading the whole 64-byte record
        // up front removes a dependent gather but measured +2.5 us on the K = 1 kernel and +7 us on the K = 20
        // one: the binning is bound by gather transactions, not by their latency)
        b = ws.rec[(size_t)t.n * F + f].box;
        // a workspace shared with a render of larger blur: tighten the (margin-expanded) box; a
        // degenerate face's (inf, -inf, inf, -inf) stays what it is
        b.x += box_shrink; b.y -= box_shrink; b.z += box_shrink; b.w -= box_shrink;
        pass = !(t.t_xmin > b.y || t.t_xmax < b.x || t.t_ymin > b.w || t.t_ymax < b.z);
      }
      const unsigned long long bal = __ballot(pass);
      if (pass) {
        const int pos = list_n + __popcll(bal & lt);
        const size_t o = (size_t)t.n * F + f;
        L.box[pos] = b;
        L.a[pos] = ws.rec[o].a;
        L.b[pos] = ws.rec[o].b;
        const float4 c4 = ws.rec[o].c;
        L.c[pos] = make_float4(c4.x, c4.z, c4.w, __int_as_float(f));
#if ACFM_EDGE_CONST
        if constexpr (has_edge_const<LT>::value) { L.e0[pos] = ws.rec[o].e0; L.e1[pos] = ws.rec[o].e1; }
#endif
      }
      list_n += __popcll(bal);
    }
    if (list_n > LT::CAP - RT || (done && list_n > 0)) {
      wave_lds_sync();
      walk(list_n);
      wave_lds_sync();
      list_n = 0;
    }
    if (done) break;
  }
}

struct Hit { float pz, sd, c0, c1, c2, d01, d02, d12; };

// One pixel against one face, split in two stages so callers can drop a face after the cheap
// half.  Every rejection of the oracle (oracle_rasterize) is a pure filter, so evaluating
// them in a different order keeps the accepted set -- and every accepted value -- identical.
//   stage 1: barycentrics (IEEE divisions), depth pz, inside flag;  rejects pz < 0
//   stage 2: the three edge distances;  rejects !inside && d >= blur
// INSIDE_ONLY: the caller keeps only pixels inside the face (blur == 0): the others leave before
// the clipped barycentrics and the depth are computed.
// clip_barycentric_coords (the texture branch): clamp to [0,1], renormalise by max(sum, 1e-5); and the depth
// interpolated with whichever barycentrics apply.  One definition each: the K-nearest forward evaluates the
// clipped depth of a covering face too (ACFM_RECORD_COVER) and must land on the bits of the K = 1 render.
__device__ __forceinline__ void clip_bary(float& c0, float& c1, float& c2) {
  c0 = fmaxf(fminf(c0, 1.0f), 0.0f);
  c1 = fmaxf(fminf(c1, 1.0f), 0.0f);
  c2 = fmaxf(fminf(c2, 1.0f), 0.0f);
  const float s = fmaxf(c0 + c1 + c2, 1e-5f);
  const float rs = recip_refined(s);
  c0 = div_by(c0, s, rs); c1 = div_by(c1, s, rs); c2 = div_by(c2, s, rs);
}
__device__ __forceinline__ float bary_depth(float c0, float c1, float c2, float z0, float z1, float z2) {
  return c0 * z0 + c1 * z1 + c2 * z2;
}

template <bool CLIP, bool INSIDE_ONLY = false>
__device__ __forceinline__ bool test_face_depth(float xf, float yf, const float4& A, const float4& B,
                                                float z2, float denom, float r, Hit& h, bool& inside) {
  const float x0 = A.x, y0 = A.y, x1 = A.z, x2 = A.w, y1 = B.x, y2 = B.y;
  const float z0 = B.z, z1 = B.w;
  // three IEEE divisions by the same denominator denom = area + kEps share one refined reciprocal r (acfm_common.h),
  // both computed once per face by k_setup (FaceRec.c) with these very operations
  auto div = [&](float x) { return div_by(x, denom, r); };
  const float w0 = div(edge_fn(xf, yf, x1, y1, x2, y2));
  const float w1 = div(edge_fn(xf, yf, x2, y2, x0, y0));
  const float w2 = div(edge_fn(xf, yf, x0, y0, x1, y1));
  float c0 = w0, c1 = w1, c2 = w2;
  inside = (w0 > 0.0f) && (w1 > 0.0f) && (w2 > 0.0f);
  if (INSIDE_ONLY && !inside) return false;
  if (CLIP) clip_bary(c0, c1, c2);
  const float pz = bary_depth(c0, c1, c2, z0, z1, z2);
  h.pz = pz; h.c0 = c0; h.c1 = c1; h.c2 = c2;
  return !(pz < 0.0f);
}

// tpar (optional): the clamped segment parameters of the three edges (01, 02, 12), for the backward
__device__ __forceinline__ bool test_face_dist(float xf, float yf, const float4& A, const float4& B,
                                               float blur, bool inside, Hit& h, float* tpar = nullptr) {
  const float x0 = A.x, y0 = A.y, x1 = A.z, x2 = A.w, y1 = B.x, y2 = B.y;
  // edges 01 and 02 (both start at vertex 0) share the packed pipe, edge 12 goes through the scalar one
  const v2f ax = {x0, x0}, ay = {y0, y0}, bx = {A.z, A.w}, by = {B.x, B.y};
  v2f t2;
  const v2f d2 = point_line_dist2(xf, yf, ax, ay, bx, by, tpar ? &t2 : nullptr);
  h.d01 = d2.x;
  h.d02 = d2.y;
  h.d12 = point_line_dist(xf, yf, x1, y1, x2, y2, tpar ? tpar + 2 : nullptr);
  if (tpar) { tpar[0] = t2.x; tpar[1] = t2.y; }
  const float d = fminf(fminf(h.d01, h.d02), h.d12);
  h.sd = inside ? -d : d;
  return inside || !(d >= blur);
}

// The same with the per-edge constants of the candidate (ACFM_EDGE_CONST): E0 = (|e01|^2, |e02|^2, r01, r02),
// E1 = (|e12|^2, r12, degenerate flag, -).  Operation for operation point_line_dist / point_line_dist2 minus what does
// not depend on the pixel; a face with a degenerate edge (flag) must take test_face_dist (its distance-to-endpoint branch).
__device__ __forceinline__ bool test_face_dist_e(float xf, float yf, const float4& A, const float4& B, const float4& E0,
                                                 const float4& E1, float blur, bool inside, Hit& h, float* tpar = nullptr) {
  const float x0 = A.x, y0 = A.y, x1 = A.z, x2 = A.w, y1 = B.x, y2 = B.y;
  {
    const v2f ax = {x0, x0}, ay = {y0, y0}, bx = {A.z, A.w}, by = {B.x, B.y};
    const v2f l2 = {E0.x, E0.y}, r = {E0.z, E0.w};
    const v2f bax = bx - ax, bay = by - ay;
    const v2f num = bax * (xf - ax) + bay * (yf - ay);
    v2f t = num * r;
    t = fma2(fma2(-l2, t, num), r, t);
    t = fma2(fma2(-l2, t, num), r, t);
    t.x = fminf(fmaxf(t.x, 0.0f), 1.0f); t.y = fminf(fmaxf(t.y, 0.0f), 1.0f);
    const v2f qx = ax + t * bax, qy = ay + t * bay;
    const v2f dx = qx - xf, dy = qy - yf;
    const v2f d = dx * dx + dy * dy;
    h.d01 = d.x; h.d02 = d.y;
    if (tpar) { tpar[0] = t.x; tpar[1] = t.y; }
  }
  {
    const float bax = x2 - x1, bay = y2 - y1;
    float t = div_by(bax * (xf - x1) + bay * (yf - y1), E1.x, E1.y);
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    if (tpar) tpar[2] = t;
    const float qx = x1 + t * bax, qy = y1 + t * bay;
    const float dx = qx - xf, dy = qy - yf;
    h.d12 = dx * dx + dy * dy;
  }
  const float d = fminf(fminf(h.d01, h.d02), h.d12);
  h.sd = inside ? -d : d;
  return inside || !(d >= blur);
}

template <bool CLIP>
__device__ __forceinline__ bool test_face(float xf, float yf, const float4& A, const float4& B,
                                          float z2, float denom, float rden, float blur, Hit& h) {
  bool inside;
  if (!test_face_depth<CLIP>(xf, yf, A, B, z2, denom, rden, h, inside)) return false;
  return test_face_dist(xf, yf, A, B, blur, inside, h);
}

// blend probability sigmoid(-sd/sigma) = 1 / (1 + 2^(sd * log2(e)/sigma)) as mul + v_exp_f32 + add +
// v_rcp_f32 (4 instructions; the IEEE division sd/sigma + library expf + reciprocal were 33, most of them
// half-rate selects / compares -- tools/ubench/valu_rates.hip).  Error budget against the oracle's exact
// expf: the product rounds once (|arg| <= 13.3 up to the blur radius: 5.5e-7 relative on 2^arg where
// p ~ 1e-4, nothing where p ~ 0.5), v_exp_f32 and v_rcp_f32 are 1 ulp each: |dp| <= p (1 - p) 2e-7 + 6e-8 p
// <= 1e-7 per face; measured on the parity sweep: masks within 4e-7 (bar 1e-6).  Inside faces (sd < 0, far
// from the edge) underflow to p = 1 exactly like expf does.  Face ids never depend on p.
#ifndef ACFM_FAST_SIGMOID
#define ACFM_FAST_SIGMOID 1
#endif
__device__ __forceinline__ float sigmoid_scale(float sigma) {   // wave-uniform: kept in an SGPR
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(1.44269504088896341f / sigma)));
}
__device__ __forceinline__ float sigmoid_neg_fast(float sd, float sigma, float scale) {
#if ACFM_FAST_SIGMOID
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(sd * scale));
#else
  return __builtin_amdgcn_rcpf(1.0f + expf(sd / sigma));
#endif
}

// (depth, face) key: pz >= 0 so its bit pattern orders like the float; +0.0f folds -0.0 into
// +0.0.  Smaller face id wins a depth tie.
__device__ __forceinline__ unsigned long long make_key(float pz, int fid) {
  return ((unsigned long long)__float_as_uint(pz + 0.0f) << 32) | (unsigned)fid;
}

// ------------------------------------------------------------------------------- forward
// Storage type of images and masks: float, or IEEE half with ACFM_STORE_F16 (AcfmRasterTuning.flags bit 1, BASELINE
// config 5 "fp16 render with fp32 loss accumulate").  Only what is STORED changes: every accept / reject decision,
// depth, blend factor and loss sum is computed in fp32 exactly as in the fp32 build, so face ids are identical.
typedef _Float16 half_t;
__device__ __forceinline__ float ld_real(const void* p, size_t i, int h16) {
  return h16 ? (float)reinterpret_cast<const half_t*>(p)[i] : reinterpret_cast<const float*>(p)[i];
}
#ifndef ACFM_OUT_NT
#define ACFM_OUT_NT 0
#endif
__device__ __forceinline__ void st_real(void* p, size_t i, float v, int h16) {
#if ACFM_OUT_NT
  if (h16) __builtin_nontemporal_store((half_t)v, reinterpret_cast<half_t*>(p) + i);
  else __builtin_nontemporal_store(v, reinterpret_cast<float*>(p) + i);
#else
  if (h16) reinterpret_cast<half_t*>(p)[i] = (half_t)v;
  else reinterpret_cast<float*>(p)[i] = v;
#endif
}
__device__ __forceinline__ float4 ld4_real(const void* p, size_t i4, int h16) {   // elements 4 i4 .. 4 i4 + 3 (aligned)
  if (!h16) return reinterpret_cast<const float4*>(p)[i4];
  typedef half_t h4 __attribute__((ext_vector_type(4)));
  const h4 v = reinterpret_cast<const h4*>(p)[i4];
  return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
}
__device__ __forceinline__ void st_face(void* p, size_t i, long long id, int h16) {   // nearest-face plane
#if ACFM_OUT_NT
  if (h16) __builtin_nontemporal_store((int32_t)id, reinterpret_cast<int32_t*>(p) + i);
  else __builtin_nontemporal_store((int64_t)id, reinterpret_cast<int64_t*>(p) + i);
#else
  if (h16) reinterpret_cast<int32_t*>(p)[i] = (int32_t)id;
  else reinterpret_cast<int64_t*>(p)[i] = (int64_t)id;
#endif
}

struct FwdOut {
  unsigned long long* dbg;   // diagnostic build only: per-block (t_start, t_end, hw_id) stamps
  int h16;                   // ACFM_STORE_F16: mask / imgs / sil / atlas / references are IEEE half, p2f is an int32 [N,H,H] plane
  void* mask;                // soft: [N,H,H] (real_t = float, or half with h16)
  void* p2f;                 // [N,H,H,kout] int64; h16: [N,H,H] int32 (kout = 1)
  int kout;                  // soft: K (all kept faces) or 1 (nearest face only)
  unsigned long long* kth;   // soft, optional: [N,H,H] largest kept key if K faces kept, else ~0
  uint8_t* vis;              // optional: [N,V] vertices of every nearest face
  int V;
  // texture branch (TEX)
  const float* vrgb;         // optional [N,V,3]: per-vertex colours instead of an atlas (viz)
  const void* atlas;         // [N,F,R,R,3] real_t
  void* imgs;                // [N,3,H,H] real_t
  void* sil;                 // [N,H,H] real_t
  int32_t* tidx;             // [N,H,H]
  int R;
  float gamma;
  float box_shrink;          // > 0: the workspace was set up with a larger blur margin; boxes are tightened by this much
  int atlas_n;               // number of distinct atlases: mesh n samples atlas n % atlas_n
  float sig_scale;           // log2(e) / sigma (sigmoid_scale), computed on the host: a kernel argument can be re-read
                             // from the kernarg segment with a scalar load where a computed value would be spilled
  // fused render + silhouette losses (acfm_sil_loss_forward): the block's partial sums of the loss terms leave
  // with the mask; lpart == null: plain render
  const void* lgt;           // [lrb,H,H] real_t ground-truth masks (may be null)
  const void* ledt;          // [lrb,H,H] real_t distance transforms (may be null)
  int lrb;                   // references: mesh n is compared with reference n % lrb
  float4* lpart;             // [N,blocks^2,4] (ws.lpart)
  // fused texture render + masked MSE (acfm_tex_mse_forward): lpart[..].x takes the block's sum of
  // (tex m - img m)^2 - (img m)^2 over its covered pixels (elsewhere tex = 0 and the difference vanishes)
  const void* timg;          // [lrb,3,H,H] real_t reference images
  const void* tmask;         // [lrb,H,H] real_t reference masks
  // ACFM_RECORD_COVER: the K-nearest forward writes ws.cover (cover_out), the texture forward that takes the
  // workspace over reads it (cover_in) instead of walking the faces
  int* cover_out;
  const int* cover_in;
  // acfm_sil_forward_prefill: the K-nearest forward also stores the CONSTANT outputs of the texture render that will
  // take this workspace over (acfm_tex_forward ws_ready = 3) on the blocks no face comes near -- the same blocks that
  // render would fill (one emptiness rule: the cost counts of this workspace); here the stores drain behind the walk
  // of the blocks with work, there they were 24 of the kernel's 36 us.  float storage only.
  float* pf_imgs;            // [N,3,H,H] -> 0
  float* pf_sil;             // [N,H,H] -> 0
  int64_t* pf_p2f;           // [N,H,H,1] -> -1
  int32_t* pf_tidx;          // [N,H,H] -> -1
  int prefilled;             // texture forward from the cover plane: the empty blocks hold their constants already
};

__device__ __forceinline__ void mark_visible(const RasterWs& ws, const FwdOut& out, int n, int F, int f) {
  const int4 vi = ws.vidx[(size_t)n * F + f];
  uint8_t* v = out.vis + (size_t)n * out.V;
  v[vi.x] = 1; v[vi.y] = 1; v[vi.z] = 1;
}

// Insertion of (x, xq) into the sorted register list, four slots at a time.
// `lim` (wave-uniform) bounds the number of faces any lane of the wave can hold so far: slots
// at or beyond it are still empty in every lane, so blocks that lie wholly beyond it are skipped
// with scalar branches while every register index stays a compile-time constant.
//
// ACFM_INSERT_SHIFT = 1 (shipping): SHIFT form, top block first.  With P_k = (x < key[k]) the new list is
//   key'[k] = P_k ? (P_{k-1} ? key[k-1] : x) : key[k]        (P_{-1} = false; sorted list: P_{k-1} implies P_k)
// evaluated for k descending, in place: slot k reads only the OLD slots k and k-1 and x itself never changes.
// Per slot one 64-bit compare + six selects, like the compare-exchange of the bubble form (0), but
//   * no register copies: the bubble form carries the displaced element from slot to slot, and the compiler kept the
//     old and the new 64-bit key of a slot in different register pairs (their 32-bit halves overlap in time), which
//     cost 4 v_mov_b64 + 4 v_mov_b32 per 4-slot block -- a fifth of the insertion's instructions;
//   * the compares of a block are independent of its selects (no v_cmp -> s_nop -> v_cndmask chains);
//   * the walk is top-down, so the first block no lane's element enters ENDS the insertion (everything below holds
//     smaller keys): the bubble form tested every block below the insertion point one by one.
// Lanes that do not insert are masked by exec (their compare bits are 0).  Results are identical: both forms
// produce THE sorted list of the K smallest keys (keys are unique per pixel: the face id is part of the key).
#ifndef ACFM_ASM_SLOT
#define ACFM_ASM_SLOT 1
#endif
#ifndef ACFM_INSERT_SHIFT
#define ACFM_INSERT_SHIFT 1
#endif
__device__ __forceinline__ unsigned long long key_lt_mask(unsigned long long x, unsigned long long k) {
  return __builtin_amdgcn_uicmpl(x, k, 36 /* ICMP_ULT */);   // one v_cmp_lt_u64 into an SGPR pair (lanes off: 0)
}
template <int K, int B>   // block B = slots [4B, min(4B + 4, K)), called for B = top .. 0
__device__ __forceinline__ void shift_insert_block(unsigned long long (&key)[K], float (&q)[K], const unsigned long long x,
                                                   const float xq, int lim) {
  constexpr int LO = 4 * B, HI = (LO + 4 < K ? LO + 4 : K);
#ifndef ACFM_INSERT_FILLTEST
#define ACFM_INSERT_FILLTEST 0   // measured: 197.1 us with it, 197.5 without (A/B on one box): within noise, and it costs three registers
#endif
  bool beyond = lim <= LO;   // (the list holds at most lim entries after this insertion: slots >= lim stay empty in every lane)
#if ACFM_INSERT_FILLTEST
  // `lim` counts the faces WALKED, a loose bound on the faces a pixel KEPT: if slot LO - 1 is still empty in every
  // inserting lane, every insertion point lies below this block and it would only shift empties into empties
  // (one 32-bit compare -- the depth word of an empty slot is all ones, no depth >= 0 is -- instead of the block)
  if constexpr (LO > 0) {
    if (!beyond) beyond = __builtin_amdgcn_uicmp((unsigned)(key[LO - 1] >> 32), 0xffffffffu, 33 /* ICMP_NE */) == 0ull;
  }
#endif
  if (!beyond) {
    DIAG_ADD(9, 1);
    unsigned long long pk = key_lt_mask(x, key[HI - 1]);
    if (pk == 0ull) return;            // no lane's element enters this block, hence none enters a lower one
    DIAG_ADD(8, 1);
#pragma unroll
    for (int k = HI - 1; k >= LO; --k) {
      const unsigned long long pm = k > 0 ? key_lt_mask(x, key[k > 0 ? k - 1 : 0]) : 0ull;
      const bool below = __builtin_amdgcn_inverse_ballot_w64(pm);   // the element goes below slot k: slot k takes k-1's
      const bool here = __builtin_amdgcn_inverse_ballot_w64(pk);    // slot k changes at all
      const unsigned long long sk = below ? key[k > 0 ? k - 1 : 0] : x;
      const float sq = below ? q[k > 0 ? k - 1 : 0] : xq;
      key[k] = here ? sk : key[k];
      q[k] = here ? sq : q[k];
      pk = pm;
    }
  }
  if constexpr (B > 0) shift_insert_block<K, B - 1>(key, q, x, xq, lim);
}

template <int K, int LO>
__device__ __forceinline__ void bubble_insert(unsigned long long (&key)[K], float (&q)[K],
                                              unsigned long long& x, float& xq, int lim) {
#if ACFM_INSERT_SHIFT
  static_assert(LO == 0, "the shift form inserts into the whole list");
  shift_insert_block<K, (K - 1) / 4>(key, q, x, xq, lim);
#else
  // The list is sorted, so key[HI-1] is the largest of the block: if no active lane's element is
  // smaller, nothing moves in these four slots (the new face lies deeper than all of them in every
  // lane -- faces arrive in id order, not in depth order) and the block costs one compare.
  constexpr int HI = (LO + 4 < K ? LO + 4 : K);
  DIAG_ADD(9, 1);
  if (__ballot(x < key[HI - 1]) != 0ull) {
    DIAG_ADD(8, 1);
#pragma unroll
    for (int k = LO; k < HI; ++k) {
#if ACFM_ASM_SLOT
      // ONE 64-bit compare + six selects.  (With a plain `x < key[k]` the two selects of a pair are
      // canonicalised into a 64-bit umin / umax, which the backend expands into TWO compares plus register
      // copies: 9.5 half-rate instructions per slot instead of 7.  The wave-mask compare + inverse ballot
      // is opaque to that transformation and costs nothing: the mask stays in VCC.)
      const bool sw = __builtin_amdgcn_inverse_ballot_w64(__builtin_amdgcn_uicmpl(x, key[k], 36 /* ICMP_ULT */));
      const unsigned long long tk = key[k];
      const float tq = q[k];
      key[k] = sw ? x : tk; x = sw ? tk : x;
      q[k] = sw ? xq : tq;  xq = sw ? tq : xq;
#else
      const bool sw = x < key[k];
      const unsigned long long tk = key[k];
      const float tq = q[k];
      key[k] = sw ? x : tk; x = sw ? tk : x;
      q[k] = sw ? xq : tq;  xq = sw ? tq : xq;
#endif
    }
  }
  if constexpr (LO + 4 < K) {
    if (lim > LO + 4) bubble_insert<K, LO + 4>(key, q, x, xq, lim);
  }
#endif
}

// LDS of a forward workgroup (one wave).  The nearest-face kernels keep a 64-slot candidate list
// (5.5 KB: the register budget, not LDS, then bounds the waves per SIMD -- measured on the
// backward: 13.8 KB -> 6.9 KB per wave = 292 -> 256 us); the K-nearest kernels are register-bound
// at 4 waves per SIMD and get 10 240 B each (16 one-wave workgroups per CU): with every slot of pix_to_face stored
// (k_out = K) that is exactly the staging area of the block's ids, a union with the lists.
// (The per-pixel-list forward walk of round 2, ACFM_FWD_V2, lives in tools/variants/fwd_v2_per_pixel_lists.inc.)
//
// ACFM_EDGE_CONST (K-nearest kernels): a candidate carries, besides its record, what the exact per-pixel distance
// test needs per EDGE and not per pixel -- |e|^2 and the refined reciprocal 1/|e|^2 of the three edges (operands of
// the IEEE-exact division of point_line_dist) and a flag for an edge with |e|^2 <= kEps -- computed once per face by
// k_setup with the very operations the walk used to repeat for every (pixel, face) pair: two more 16-byte LDS
// entries per candidate (32 B), ~33 instruction slots fewer per walk iteration, bit-identical values.
#ifndef ACFM_EDGE_CONST
#define ACFM_EDGE_CONST 1
#endif
#ifndef ACFM_FWD_CAP
#define ACFM_FWD_CAP (ACFM_EDGE_CONST ? 88 : RCAP)   // 96 B x 88 + sub-lists + id list + cull lists = 10 208 B <= 10 240
#endif
template <int CAP_>
struct CandListET : CandListT<CAP_> {
  float4 e0[CAP_];   // (|e01|^2, 1/|e01|^2, |e02|^2, 1/|e02|^2)   -- the pair the packed pipe evaluates together
  float4 e1[CAP_];   // (|e12|^2, 1/|e12|^2, degenerate flag, -)
};
template <int CAP, int MIN_BYTES, bool EDGE>
struct FwdLdsT {
  struct Lists {
    typename std::conditional<EDGE, CandListET<CAP>, CandListT<CAP>>::type L;
    fl_t fl[FLCAP];
    unsigned char wl[2 * CAP];   // the wave's list of the edge cull, and the same in depth order (list positions < CAP <= 256)
  };
  union {
    Lists s;
    char stage[MIN_BYTES > 16 ? MIN_BYTES : 16];   // the block's K ids in image order (the lists are dead by then)
  };
};
template <int K> using FwdLdsK = FwdLdsT<(K > 1 ? ACFM_FWD_CAP : 64), (K > 1 ? 64 * K * 8 : 0), (K > 1 && ACFM_EDGE_CONST)>;
#ifndef ACFM_NO_LDS_ASSERT
static_assert(sizeof(FwdLdsK<20>) <= 10240, "the K = 20 forward runs 16 one-wave workgroups per CU: 10 240 B of LDS each");
#endif


// Constant outputs of a flagged-empty 8x8 block (no face box comes near it): exactly what
// fwd_block leaves for a block without candidates.  Lane i owns pixel (i / 8, i % 8) of the block;
// the K ids of the soft kernel go out as 16-byte pieces in image order (8 rows of 64 K bytes).
// The K-slot pix_to_face stores (160 B per pixel at K = 20: whole lines, never read by this library) go out
// non-temporal: -7 us on the K = 20 forward.  The 4-byte-per-pixel planes must NOT (ACFM_OUT_NT=1 measured 75 ->
// 123 us on the texture forward): a block row of such a plane is a 32-byte fragment and the four fragments of a
// line meet in L2.
#ifndef ACFM_P2F_NT
#define ACFM_P2F_NT 1
#endif
#if ACFM_P2F_NT
#define P2F_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define P2F_STORE(ptr, val) (*(ptr) = (val))
#endif
template <int K, bool TEX>
__device__ __forceinline__ void fwd_fill_block(const FwdOut& out, int n, int by, int bx, int H, int lane) {
  const int yi = by + (lane >> 3), xi = bx + (lane & 7);
  const bool valid = (yi < H) && (xi < H);
  const size_t pix = ((size_t)n * H + yi) * H + xi;
  if constexpr (K == 1) {
    if (TEX && out.lpart && lane == 0) {
      const int tiles = (H + RBLK - 1) / RBLK;
      out.lpart[(((size_t)n * tiles + by / RBLK) * tiles + bx / RBLK) * 4] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (!valid) return;
    st_face(out.p2f, pix, -1, out.h16);
    if (TEX) {
      const size_t HW = (size_t)H * H;
      const size_t io = (size_t)n * 3 * HW + (size_t)yi * H + xi;
      st_real(out.imgs, io, 0.f, out.h16); st_real(out.imgs, io + HW, 0.f, out.h16); st_real(out.imgs, io + 2 * HW, 0.f, out.h16);
      st_real(out.sil, pix, 0.f, out.h16);
      out.tidx[pix] = -1;
    }
  } else {
    if (valid) {
      // (kth is left alone: the backward reads it only where mask != 0, and mask is 0 on this whole block)
      st_real(out.mask, pix, 0.0f, out.h16);
      if (out.kout == 1) st_face(out.p2f, pix, -1, out.h16);
      if (out.pf_imgs) {
        const size_t HW = (size_t)H * H, io = (size_t)n * 3 * HW + (size_t)yi * H + xi;
        out.pf_imgs[io] = 0.f; out.pf_imgs[io + HW] = 0.f; out.pf_imgs[io + 2 * HW] = 0.f;
        out.pf_sil[pix] = 0.f; out.pf_tidx[pix] = -1; out.pf_p2f[pix] = -1;
      }
    }
    if (out.lpart && lane < 4) {   // mask = 0 on the whole block: nothing beyond the finish kernel's sum of gt
      const int tiles = (H + RBLK - 1) / RBLK;
      out.lpart[(((size_t)n * tiles + by / RBLK) * tiles + bx / RBLK) * 4 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (out.kout == 1) return;
    typedef long long ll2 __attribute__((ext_vector_type(2)));
    constexpr int CH = K / 2;
    ll2 v; v.x = -1; v.y = -1;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = i * 64 + lane;             // piece index: 8 rows x (8 pixels x CH pieces)
      const int r = c / (8 * CH), off = c % (8 * CH);
      if (by + r < H && bx + off / CH < H)
        P2F_STORE(&reinterpret_cast<ll2*>(reinterpret_cast<int64_t*>(out.p2f) + (((size_t)n * H + by + r) * H + bx) * K)[off], v);
    }
  }
}

// The same for images whose side is a multiple of 8 (every block whole), which is where the fills matter: 80 % of
// the blocks of a batch are empty and the generic routine above spends ~130 VALU instructions per block, a third of
// them 64-bit multiplies, on addresses that differ from block to block only by a wave-uniform base.  Here every
// lane's byte offsets inside a block are computed once per workgroup (FillLane) and a block costs one scalar base
// per output plus the stores (global_store with an SGPR base and a VGPR offset).
template <int K>
struct FillLane {
  unsigned pix;                        // (lane / 8) * H + lane % 8: the lane's pixel inside the block
  unsigned piece[K > 1 ? K / 2 : 1];   // K > 1: byte offset of the lane's i-th 16-byte piece of the K-slot id rows
};
template <int K>
__device__ __forceinline__ FillLane<K> make_fill_lane(int H, int lane) {
  FillLane<K> f;
  f.pix = (unsigned)((lane >> 3) * H + (lane & 7));
  if constexpr (K > 1) {
    constexpr int CH = K / 2;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = i * 64 + lane;             // piece index: 8 rows x (8 pixels x CH pieces)
      const int r = c / (8 * CH), off = c % (8 * CH);
      f.piece[i] = (unsigned)(r * H * K * 8 + off * 16);
    }
  } else {
    f.piece[0] = 0;
  }
  return f;
}
template <int K, bool TEX>
__device__ __forceinline__ void fwd_fill_block_whole(const FwdOut& out, int n, int by, int bx, int H, int lane,
                                                     const FillLane<K>& fl) {
  // wave-uniform: first pixel of the block, and the tiles index of its partial-sum record
  const size_t pix0 = ((size_t)n * H + by) * H + bx;
  const size_t pix = pix0 + fl.pix;
  if constexpr (K == 1) {
    if (TEX && out.lpart && lane == 0) {
      const int tiles = H / RBLK;
      out.lpart[(((size_t)n * tiles + by / RBLK) * tiles + bx / RBLK) * 4] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    st_face(out.p2f, pix, -1, out.h16);
    if (TEX) {
      const size_t HW = (size_t)H * H;
      const size_t io = pix + (size_t)n * 2 * HW;   // [N,3,H,W]: n 3 HW + y W + x
      st_real(out.imgs, io, 0.f, out.h16); st_real(out.imgs, io + HW, 0.f, out.h16); st_real(out.imgs, io + 2 * HW, 0.f, out.h16);
      st_real(out.sil, pix, 0.f, out.h16);
      out.tidx[pix] = -1;
    }
  } else {
    st_real(out.mask, pix, 0.0f, out.h16);   // (kth: see fwd_fill_block)
    if (out.kout == 1) st_face(out.p2f, pix, -1, out.h16);
    if (out.pf_imgs) {
      const size_t HW = (size_t)H * H, io = pix + (size_t)n * 2 * HW;
      out.pf_imgs[io] = 0.f; out.pf_imgs[io + HW] = 0.f; out.pf_imgs[io + 2 * HW] = 0.f;
      out.pf_sil[pix] = 0.f; out.pf_tidx[pix] = -1; out.pf_p2f[pix] = -1;
    }
    if (out.lpart && lane < 4) {   // mask = 0 on the whole block: nothing beyond the finish kernel's sum of gt
      const int tiles = H / RBLK;
      out.lpart[(((size_t)n * tiles + by / RBLK) * tiles + bx / RBLK) * 4 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (out.kout == 1) return;
    typedef long long ll2 __attribute__((ext_vector_type(2)));
    constexpr int CH = K / 2;
    ll2 v; v.x = -1; v.y = -1;
    char* base = reinterpret_cast<char*>(reinterpret_cast<int64_t*>(out.p2f) + pix0 * K);
#pragma unroll
    for (int i = 0; i < CH; ++i) P2F_STORE(reinterpret_cast<ll2*>(base + fl.piece[i]), v);
  }
}

// What the nearest-face (K = 1) forward does with a pixel's winner: face id, visibility, and for the texture branch
// the atlas lookup, blend and optional fused MSE partial.  Shared by the walking kernel (fwd_block) and by the
// kernel that reads the winner from the cover plane (k_tex_cover).
template <bool CLIP, bool TEX>
__device__ __forceinline__ void k1_finish(const RasterWs& ws, const Tile& t, int F, int H, float sigma, const FwdOut& out,
                                          unsigned long long bestkey, float bestsd, float bestb0, float bestb1,
                                          float bestb2, bool dist_late) {
  const int n = t.n;
  const int64_t fbase = (int64_t)n * F;
    float tacc = 0.f;     // fused texture MSE: this pixel's (tex m - img m)^2 - (img m)^2 over the three channels
    if (t.valid) {
    const bool hit = (bestkey != KEY_NONE);
    const int f = (int)(bestkey & 0xffffffffu);
    if (TEX && dist_late && hit) {
      const size_t o = (size_t)n * F + f;
      Hit h;
      test_face_dist(t.xf, t.yf, ws.rec[o].a, ws.rec[o].b, 0.0f, true, h);
      bestsd = h.sd;
    }
    st_face(out.p2f, t.pix, hit ? fbase + f : (int64_t)-1, out.h16);
    if (out.vis && hit) mark_visible(ws, out, n, F, f);
    if (TEX) {
      // TexturesAtlas.sample_textures + ambient-only Phong + softmax_rgb_blend, K = 1
      // (oracle_atlas_shade is the line-by-line spec)
      const size_t HW = (size_t)H * H;
      const size_t io = (size_t)n * 3 * HW + (size_t)t.yi * H + t.xi;
      if (!hit) {
        st_real(out.imgs, io, 0.f, out.h16); st_real(out.imgs, io + HW, 0.f, out.h16); st_real(out.imgs, io + 2 * HW, 0.f, out.h16);
        st_real(out.sil, t.pix, 0.f, out.h16);
        out.tidx[t.pix] = -1;
      } else {
        const int R = out.R;
        const float zb = __uint_as_float((unsigned)(bestkey >> 32));
        int ix = (int)(bestb0 * (float)R), iy = (int)(bestb1 * (float)R);
        const bool below = ((bestb0 + bestb1) * (float)R - ((float)ix + (float)iy)) <= 1.0f;
        if (!below) { ix = R - 1 - ix; iy = R - 1 - iy; }
        ix = min(max(ix, 0), R - 1); iy = min(max(iy, 0), R - 1);
        const size_t ti = ((((size_t)(n % out.atlas_n) * F + f) * R + iy) * R + ix);
        const float eps = 1e-10f, znear = 1.0f, zfar = 100.0f;
        const float prob = sigmoid_neg(bestsd, sigma);
        const float z_inv = (zfar - zb) / (zfar - znear);
        const float z_inv_max = fmaxf(z_inv, eps);
        const float wnum = prob * expf((z_inv - z_inv_max) / out.gamma);
        const float delta = fmaxf(expf((eps - z_inv_max) / out.gamma), eps);
        const float den = wnum + delta;
        float cr, cg, cb;
        if (out.vrgb) {
          // Textures(verts_rgb) (nmr.py:177-179): barycentric interpolation of the face's vertex colours
          const int4 vi = ws.vidx[(size_t)n * F + f];
          const float* c0 = out.vrgb + ((size_t)n * out.V + vi.x) * 3;
          const float* c1 = out.vrgb + ((size_t)n * out.V + vi.y) * 3;
          const float* c2 = out.vrgb + ((size_t)n * out.V + vi.z) * 3;
          const float b2 = bestb2;
          cr = bestb0 * c0[0] + bestb1 * c1[0] + b2 * c2[0];
          cg = bestb0 * c0[1] + bestb1 * c1[1] + b2 * c2[1];
          cb = bestb0 * c0[2] + bestb1 * c1[2] + b2 * c2[2];
        } else {
          cr = ld_real(out.atlas, ti * 3, out.h16); cg = ld_real(out.atlas, ti * 3 + 1, out.h16);
          cb = ld_real(out.atlas, ti * 3 + 2, out.h16);
        }
        const float vr = (wnum * cr + delta * 0.0f) / den, vg = (wnum * cg + delta * 0.0f) / den,
                    vb = (wnum * cb + delta * 0.0f) / den;
        st_real(out.imgs, io, vr, out.h16); st_real(out.imgs, io + HW, vg, out.h16); st_real(out.imgs, io + 2 * HW, vb, out.h16);
        st_real(out.sil, t.pix, 1.0f - (1.0f - prob), out.h16);
        out.tidx[t.pix] = (int32_t)ti;
        ws.fvis[(size_t)n * F + f] = 1;   // the atlas gradient (k_tex_bwd_faces) visits only faces that were seen
        if (out.lpart) {
          const size_t pp = (size_t)t.yi * H + t.xi, rn = (size_t)(n % out.lrb);
          const float mk = ld_real(out.tmask, rn * HW + pp, out.h16);
          const size_t ro = rn * 3 * HW + pp;
          const float b0 = ld_real(out.timg, ro, out.h16) * mk, b1 = ld_real(out.timg, ro + HW, out.h16) * mk,
                      b2 = ld_real(out.timg, ro + 2 * HW, out.h16) * mk;
          const float d0 = vr * mk - b0, d1 = vg * mk - b1, d2 = vb * mk - b2;
          tacc = (d0 * d0 - b0 * b0) + (d1 * d1 - b1 * b1) + (d2 * d2 - b2 * b2);
        }
      }
    }
    }   // t.valid
    if (TEX && out.lpart) {
      tacc = wave_sum(tacc);
      if (t.lane == 0) {
        const int tiles = (H + RBLK - 1) / RBLK;
        out.lpart[(((size_t)n * tiles + t.yi / RBLK) * tiles + t.xi / RBLK) * 4] = make_float4(tacc, 0.f, 0.f, 0.f);
      }
    }
}

template <int K, bool CLIP, bool TEX>
__device__ __forceinline__ void fwd_block(const RasterWs& ws, const Tile& t, int F, int H, float blur, float sigma,
                                          const FwdOut& out, FwdLdsK<K>& S) {
  auto& L = S.s.L;
  fl_t* s_fl = S.s.fl;
  const int n = t.n;
  const int64_t fbase = (int64_t)n * F;

  if constexpr (K == 1) {
    unsigned long long bestkey = KEY_NONE;
    float bestsd = 0.f, bestb0 = 0.f, bestb1 = 0.f, bestb2 = 0.f;
    // blur == 0 (every hard render of the reference): a face is accepted iff the pixel is inside it,
    // so the three edge distances decide nothing; only the winner's signed distance is ever used
    // (the blend weight of the texture branch) and is evaluated once per pixel after the walk.
    const bool dist_late = !(blur > 0.0f);
    bin_and_walk(ws, t, F, H, L, s_fl, out.box_shrink, [&](int list_n) {
      walk_wave<false>(L, t, H, list_n, blur, nullptr, [&](const Cand& cd, bool in_box, int ord) {
        if (!(in_box && t.valid)) return;
        Hit h;
        h.sd = 0.f;
        bool inside = false;
        if (dist_late) {
          if (!test_face_depth<CLIP, true>(t.xf, t.yf, cd.a, cd.b, cd.c.x, cd.c.y, cd.rden, h, inside)) return;
        } else {
          if (!test_face_depth<CLIP>(t.xf, t.yf, cd.a, cd.b, cd.c.x, cd.c.y, cd.rden, h, inside)) return;
          if (!test_face_dist(t.xf, t.yf, cd.a, cd.b, blur, inside, h)) return;
        }
        const unsigned long long key = make_key(h.pz, cd.fid);
        if (key < bestkey) { bestkey = key; bestsd = h.sd; bestb0 = h.c0; bestb1 = h.c1; bestb2 = h.c2; }
      });
    });
    k1_finish<CLIP, TEX>(ws, t, F, H, sigma, out, bestkey, bestsd, bestb0, bestb1, bestb2, dist_late);
  } else {
    // Per-pixel top-K list: K (depth|face) keys + their blend factors (1 - p), kept SORTED in
    // registers.  A new face is bubbled through the array with compare-exchanges on static
    // register indices (the displaced farthest entry falls off the end), so there is no LDS or
    // memory list, no final sort and the kept set is exactly the K nearest at every moment.
    const float sig_scale = out.sig_scale;
    unsigned long long key[K];
    float q[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { key[k] = KEY_NONE; q[k] = 1.0f; }
    int seen = 0;  // faces walked so far by this wave (uniform): no lane holds more than that
    unsigned long long cbest = KEY_NONE;   // ACFM_RECORD_COVER: nearest covering face so far
    unsigned char* s_wl = S.s.wl;  // first stage of the edge cull
    DIAG_ADD(0, 1);
    bin_and_walk(ws, t, F, H, L, s_fl, out.box_shrink, [&](int list_n) {
      DIAG_ADD(1, 1); DIAG_ADD(2, list_n);
#ifdef ACFM_DIAG_NO_WALK
      if (list_n >= 0) { seen += list_n; return; }
#endif
      walk_wave<true>(L, t, H, list_n, blur, s_wl, [&](const Cand& cd, bool in_box, int ord) {
#ifdef ACFM_DIAG_NO_BODY
        if (ord >= 0) return;
#endif
        // stage 1 (depth): a face that is not nearer than the K-th kept face of a full list
        // cannot enter it; when that holds for every lane of the wave the face is dropped
        // before its edge distances are computed (empty slots hold ~0, so x < key[K-1] is
        // always true for a list that is not full yet)
        Hit h;
        bool inside = false;
        bool live = in_box && t.valid &&
                    test_face_depth<CLIP>(t.xf, t.yf, cd.a, cd.b, cd.c.x, cd.c.y, cd.rden, h, inside);
        if (out.cover_out) {
          // the hard K = 1 render's candidate test for this pair (test_face_depth<true, true>): strictly inside,
          // depth from the CLIPPED barycentrics (h.c* hold the unclipped ones here), not negative.  Before the
          // K-th-key filter below: the nearest covering face need not be among the K nearest kept faces.
          // (a face that cannot win is dropped before the divisions: clipping an inside pixel's barycentrics divides
          // them by their sum s = area / (area + kEps) -- above 1 for a back-facing face -- so pzc = pz / s up to
          // rounding, and pz (1 - 1e-5) > best s means pzc > best.  With the walk roughly front to back this spares
          // most of the back layer.  cbest still empty: its depth bits are a NaN and the comparison is false.)
          const bool cin = in_box && t.valid && inside &&
                           !(h.pz * 0.99999f > __uint_as_float((unsigned)(cbest >> 32)) * (h.c0 + h.c1 + h.c2));
          if (__ballot(cin) != 0ull) {
            float c0 = h.c0, c1 = h.c1, c2 = h.c2;
            clip_bary(c0, c1, c2);
            const float pzc = bary_depth(c0, c1, c2, cd.b.z, cd.b.w, cd.c.x);
            const unsigned long long xc = make_key(pzc, cd.fid);
            if (cin && !(pzc < 0.0f) && xc < cbest) cbest = xc;
          }
        }
        unsigned long long x = make_key(h.pz, cd.fid);
        live = live && (x < key[K - 1]);
        if (__ballot(live) == 0ull) return;
        DIAG_ADD(5, 1); DIAG_ADD(6, __popcll(__ballot(live)));
        if (!live) return;
#ifdef ACFM_DIAG_NO_STAGE2
        h.sd = h.pz;
#else
#if ACFM_EDGE_CONST
        // (read here, not with the record: 8 registers that need not live through the depth stage)
        const float4 e1 = L.e1[cd.idx];
        if (__ballot(e1.z != 0.0f) != 0ull) {   // (rare: some lane's face has an edge shorter than sqrt(kEps))
          if (!test_face_dist(t.xf, t.yf, cd.a, cd.b, blur, inside, h)) return;
        } else {
          if (!test_face_dist_e(t.xf, t.yf, cd.a, cd.b, L.e0[cd.idx], e1, blur, inside, h)) return;
        }
#else
        if (!test_face_dist(t.xf, t.yf, cd.a, cd.b, blur, inside, h)) return;
#endif
#endif
        DIAG_ADD(7, __popcll(__ballot(true))); DIAG_ADD(11, 1);
#ifdef ACFM_DIAG_COUNT
        {   // (group, face) pairs with at least one accepting pixel
          const unsigned long long am = __ballot(true);
          DIAG_ADD(13, ((am & 0xffffull) != 0) + ((am & 0xffff0000ull) != 0) + ((am & 0xffff00000000ull) != 0) + ((am >> 48) != 0));
        }
#endif
        float xq = 1.0f - sigmoid_neg_fast(h.sd, sigma, sig_scale);
#ifdef ACFM_DIAG_NO_INSERT
        if (x < key[0]) { key[0] = x; q[0] = xq; }
#else
        bubble_insert<K, 0>(key, q, x, xq, __builtin_amdgcn_readfirstlane(seen + ord + 1));
#endif
      });
      seen += list_n;
    });
    const bool split = t.sub >= 0;
    if (split) {
      // The four 16-lane groups hold the K nearest of their quarter of the candidates for the same
      // 16 pixels.  Merge: group 0 takes group 1's entries and group 2 takes group 3's, then group 0
      // takes group 2's; an entry enters by the same bubble insertion (the K nearest of a union are
      // the K nearest of the two K-nearest lists).  Senders keep their lists untouched while they
      // are being read; lists are sorted, so a round ends at the first empty slot of every sender.
#pragma unroll
      for (int round = 0; round < 2; ++round) {
        const int mask = round == 0 ? 16 : 32;
        const bool recv = round == 0 ? ((t.lane & 16) == 0) : ((t.lane & 48) == 0);
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)(key[k] & 0xffffffffull), mask, 64);
          const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(key[k] >> 32), mask, 64);
          float oq = __shfl_xor(q[k], mask, 64);
          unsigned long long ok = recv ? (((unsigned long long)hi << 32) | lo) : KEY_NONE;
          if (__ballot(ok != KEY_NONE) != 0ull) bubble_insert<K, 0>(key, q, ok, oq, K);
        }
      }
    }
    const bool out_valid = t.valid && (!split || t.lane < 16);
    if (out.cover_out) {
      if (split) {   // the four 16-lane groups saw different candidates of the same 16 pixels
#pragma unroll
        for (int m = 16; m <= 32; m <<= 1) {
          const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)(cbest & 0xffffffffull), m, 64);
          const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(cbest >> 32), m, 64);
          const unsigned long long o = ((unsigned long long)hi << 32) | lo;
          if (o < cbest) cbest = o;
        }
      }
      if (out_valid) out.cover_out[t.pix] = cbest != KEY_NONE ? (int)(cbest & 0xffffffffu) : -1;
    }
    float lmask = 0.f, lg = 0.f, le = 0.f;
    if (out.lpart && out_valid) {
      const size_t rp = t.pix - (size_t)n * H * H + (size_t)(n % out.lrb) * H * H;
      if (out.lgt) lg = ld_real(out.lgt, rp, out.h16);
      if (out.ledt) le = ld_real(out.ledt, rp, out.h16);
    }
    if (out_valid) {
      float alpha = 1.0f;  // sigmoid_alpha_blend over the kept faces in ascending depth; empty slots hold 1
#pragma unroll
      for (int k = 0; k < K; ++k) alpha = alpha * q[k];
      st_real(out.mask, t.pix, 1.0f - alpha, out.h16);
      if (out.kth) out.kth[t.pix] = key[K - 1];  // ~0 unless K faces are kept
      lmask = 1.0f - alpha;
      if (out.vis && key[0] != KEY_NONE) mark_visible(ws, out, n, F, (int)(key[0] & 0xffffffffu));
      // lean output: only the nearest-face plane, the one slot any caller of the reference
      // reads (loss_utils.py:214, 431); the other K-1 ids stay in registers
      if (out.kout == 1)
        st_face(out.p2f, t.pix, (key[0] != KEY_NONE) ? fbase + (long long)(key[0] & 0xffffffffu) : (long long)-1, out.h16);
    }
    // Fused silhouette losses: with m = 0 outside the blocks that have work, sum|m - g| = sum g + sum(|m - g| - g),
    // sum(m + g - m g) = sum g + sum(m - m g); the finish kernels add sum g.  One 16-byte store per block (and
    // split role), no atomics: the per-mesh sums are formed in fixed order (deterministic).  Called last, after the
    // block's ids are on their way: the references were requested before the outputs (lg, le above).
    auto losses_out = [&]() {
      if (!out.lpart) return;
      float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
      if (out_valid) { t0 = fabsf(lmask - lg) - lg; t1 = lmask * lg; t2 = lmask - lmask * lg; t3 = le * lmask; }
      t0 = wave_sum(t0); t1 = wave_sum(t1); t2 = wave_sum(t2); t3 = wave_sum(t3);
      if (t.lane < 4) {
        const int tiles = (H + RBLK - 1) / RBLK;
        // (yi / 8, xi / 8 are the block's own for every lane)
        const size_t slot = (((size_t)n * tiles + t.yi / RBLK) * tiles + t.xi / RBLK) * 4;
        if (split) { if (t.lane == 0) out.lpart[slot + t.sub] = make_float4(t0, t1, t2, t3); }
        else out.lpart[slot + t.lane] = t.lane == 0 ? make_float4(t0, t1, t2, t3) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    if (out.kout == 1) { losses_out(); return; }
    typedef long long ll2 __attribute__((ext_vector_type(2)));  // K even -> 16-byte pieces
    constexpr int CH = K / 2;                                   // pieces per pixel
    constexpr bool STAGED = sizeof(FwdLdsK<K>) >= (size_t)64 * K * 8;
    if (STAGED && !split) {
      // The K ids of a pixel are 8K contiguous bytes, so a lane storing its own row hits 64
      // different cache lines per instruction.  The block's ids are therefore staged in LDS (the
      // candidate lists are dead by now) in image order -- 8 rows of 64K contiguous bytes -- and
      // written out with consecutive lanes on consecutive 16-byte pieces.
      wave_lds_sync();
      ll2* so = reinterpret_cast<ll2*>(&S.stage[0]);
      const int slot = (t.yi & 7) * 8 + (t.xi & 7);
#pragma unroll
      for (int k2 = 0; k2 < CH; ++k2) {
        ll2 v;
        v.x = (key[2 * k2] != KEY_NONE) ? fbase + (long long)(key[2 * k2] & 0xffffffffu) : (long long)-1;
        v.y = (key[2 * k2 + 1] != KEY_NONE) ? fbase + (long long)(key[2 * k2 + 1] & 0xffffffffu) : (long long)-1;
        so[slot * CH + k2] = v;
      }
      wave_lds_sync();
      const int by = t.yi & ~7, bx = t.xi & ~7;
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int c = i * 64 + t.lane;             // piece index: 8 rows x (8 pixels x CH pieces)
        const int r = c / (8 * CH), off = c % (8 * CH);
        if (by + r < H && bx + off / CH < H)
          P2F_STORE(&reinterpret_cast<ll2*>(reinterpret_cast<int64_t*>(out.p2f) + (((size_t)n * H + by + r) * H + bx) * K)[off], so[c]);
      }
    } else if (out_valid) {
      ll2* o2 = reinterpret_cast<ll2*>(reinterpret_cast<int64_t*>(out.p2f) + t.pix * K);
#pragma unroll
      for (int k2 = 0; k2 < CH; ++k2) {
        ll2 v;
        v.x = (key[2 * k2] != KEY_NONE) ? fbase + (long long)(key[2 * k2] & 0xffffffffu) : (long long)-1;
        v.y = (key[2 * k2 + 1] != KEY_NONE) ? fbase + (long long)(key[2 * k2 + 1] & 0xffffffffu) : (long long)-1;
        o2[k2] = v;
      }
    }
    losses_out();
  }
}

// waves per SIMD the K = 1 kernels are compiled for: 4, 5 and 6 measured the same 70 us on the texture forward (it is
// VALU-issue bound, 78 % busy, not latency bound), 7 and 8 spill (75 and 90 us)
#ifndef ACFM_K1_WAVES
#define ACFM_K1_WAVES 6
#endif
template <int K, bool CLIP, bool TEX>
__global__ __launch_bounds__(RT, K > 20 ? 2 : K == 1 ? ACFM_K1_WAVES : 4) void k_raster_fwd(RasterWs ws, int N, int F, int H, float blur,
                                                    float sigma, FwdOut out) {
  __shared__ __attribute__((aligned(16))) FwdLdsK<K> S;
  const Sched sc = make_sched(ws, N, H, K > 1);   // the K-nearest kernels split their heaviest blocks
  struct Stamp {   // diagnostic build: (t_start, t_end, hw | xcc << 32 | (sub + 1) << 36 | cost << 40, t_work_start, t_work_end) per workgroup
    unsigned long long* p; unsigned long long t0, tw, tf; int sub, cost;
    __device__ Stamp(unsigned long long* q) : p(q), t0(q ? __builtin_amdgcn_s_memrealtime() : 0), tw(0), tf(0), sub(-1), cost(0) {}
    __device__ void work(int sub_, int cost_) { if (p) { tw = __builtin_amdgcn_s_memrealtime(); sub = sub_; cost = cost_; } }
    __device__ void work_end() { if (p) tf = __builtin_amdgcn_s_memrealtime(); }
    __device__ ~Stamp() {
      if (p && threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
        unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID
        p[5 * (size_t)blockIdx.x] = t0; p[5 * (size_t)blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        p[5 * (size_t)blockIdx.x + 2] = ((unsigned long long)(cost & 0xffff) << 40) | ((unsigned long long)(sub + 1) << 36) |
                                        ((unsigned long long)xcc << 32) | hw;
        p[5 * (size_t)blockIdx.x + 3] = tw; p[5 * (size_t)blockIdx.x + 4] = tf;
      }
    }
  } stamp(out.dbg);
  const int lane = threadIdx.x & 63;
  if (sc.sub < 0) {
    // this workgroup's share of the flagged-empty blocks: fire-and-forget stores, issued first
    // a contiguous run of them: the empty entries end the order in ascending block index, so a wave's consecutive
    // blocks are neighbours in the image and their 32-byte row fragments meet in L2 as whole lines
    const int n_empty = sc.per - sc.n_work, chunk = (n_empty + sc.stride - 1) / sc.stride;
    const int e0 = sc.n_work + sc.j0 * chunk, e1 = min(sc.per, e0 + chunk);
#ifdef ACFM_DIAG_NO_FILL
    constexpr bool no_fill = K == 1;
#else
    constexpr bool no_fill = false;
#endif
    if ((H & (RBLK - 1)) == 0) {
      const FillLane<K> fl = make_fill_lane<K>(H, lane);
#pragma unroll 1
      for (int e = e0; e < e1 && !no_fill; ++e) {
        int n, by, bx;
        entry_block(ws.order[(size_t)sc.g * sc.per + e], sc, H, n, by, bx);
        fwd_fill_block_whole<K, TEX>(out, n, by, bx, H, lane, fl);
      }
    } else {
#pragma unroll 1
      for (int e = e0; e < e1 && !no_fill; ++e) {
        int n, by, bx;
        entry_block(ws.order[(size_t)sc.g * sc.per + e], sc, H, n, by, bx);
        fwd_fill_block<K, TEX>(out, n, by, bx, H, lane);
      }
    }
  }
#pragma unroll 1
  for (int e = sc.j0; e < sc.e_end; e += sc.stride) {
    const Tile t = make_tile(ws, sc, e, N, H, K > 1);
#ifdef ACFM_DIAG_NO_WORK
    if (K == 1) continue;
#endif
    if (!t.none) {
#ifdef ACFM_DIAG
      if (out.dbg) {
        const int tiles_ = (H + RBLK - 1) / RBLK;
        stamp.work(t.sub, block_cost(ws, t.n, __builtin_amdgcn_readfirstlane((t.yi / RBLK) * tiles_ + t.xi / RBLK), H));
      }
#endif
      fwd_block<K, CLIP, TEX>(ws, t, F, H, blur, sigma, out, S);
    }
    wave_lds_sync();   // the next block reuses the LDS lists
  }
#ifdef ACFM_DIAG
  stamp.work_end();
#endif
}

// Texture forward from the cover plane (ACFM_RECORD_COVER, acfm_tex_forward ws_ready = 2): the K-nearest render of this
// geometry left the nearest covering face of every pixel of its work blocks in ws.cover -- same inside test, same
// clipped depth, same tie-break as the K = 1 walk -- so a block is one dependent chain (id -> face record -> texel)
// per pixel and no binning.  Same schedule as k_raster_fwd (the order's work entries, then a contiguous run of the
// flagged-empty ones per wave); the unit is a WAVE of a four-wave workgroup and there is no LDS.  What is left is
// mostly the constant stores of the ~80 % empty blocks (24 us by themselves at 64 frames @256^2).
constexpr int COVER_WPB = 4;   // waves per workgroup
template <bool CLIP>
__global__ __launch_bounds__(64 * COVER_WPB) void k_tex_cover(RasterWs ws, int N, int F, int H, float sigma, FwdOut out) {
  const int tiles = (H + RBLK - 1) / RBLK;
  Sched sc;
  sc.G = (N & 7) == 0 ? 8 : 1;
  sc.per = (N / sc.G) * tiles * tiles;
  sc.g = sc.G == 8 ? ((int)blockIdx.x & 7) : 0;
  const int wg = sc.G == 8 ? ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  sc.j0 = wg * COVER_WPB + wave;
  sc.stride = ((int)gridDim.x / sc.G) * COVER_WPB;
  sc.n_work = ws.n_work[sc.g];
  sc.e_end = sc.n_work;
  sc.sub = -1;
  const int lane = threadIdx.x & 63;
  // Empty blocks.  Images whose side is a multiple of 32, float storage: by QUADS of four horizontally adjacent
  // blocks (32 x 8 pixels): a row of a quad is a whole 128-byte line of every 4-byte plane, so a plane of the quad
  // is ONE 16-byte store per lane instead of four 4-byte ones whose 32-byte fragments have to meet in L2.  A block
  // is empty iff its cost count is 0 (k_order's rule for the flag), read here by position instead of through the
  // order; quads with a working block fall back to single-block fills.  Waves at the low end of the group fill,
  // waves at the high end render (below), so that no wave queues both.
  const bool quads = (H & 31) == 0 && !out.h16;
  int work_j0 = sc.j0;
  if (out.prefilled) {
    // the silhouette render that left the cover plane stored these blocks' constants too (acfm_sil_forward_prefill);
    // only the fused MSE's partial-sum records of the empty blocks remain
    if (out.lpart) {
      const int n_empty = sc.per - sc.n_work;
      for (int e = sc.n_work + sc.j0 * 64 + lane; e < sc.per; e += sc.stride * 64) {
        int n, by, bx;
        entry_block(ws.order[(size_t)sc.g * sc.per + e], sc, H, n, by, bx);
        out.lpart[(((size_t)n * tiles + by / RBLK) * tiles + bx / RBLK) * 4] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      (void)n_empty;
    }
  } else if (quads) {
    const FillLane<1> fl = make_fill_lane<1>(H, lane);
    const int qrow = tiles / 4, units = (N / sc.G) * tiles * qrow;
    const size_t HW = (size_t)H * H;
    const unsigned off4 = (unsigned)((lane >> 3) * H + (lane & 7) * 4);
#pragma unroll 1
    for (int u = sc.j0; u < units; u += sc.stride) {
      const int m = u / (tiles * qrow), rem = u - m * (tiles * qrow);
      const int byb = rem / qrow, bxb = (rem - byb * qrow) * 4;
      const int n = m * sc.G + sc.g;
      const int cnt = lane < 4 ? ws.tile_cnt[((size_t)n * tiles + byb) * tiles + bxb + lane] : 1;
      const unsigned em = (unsigned)__ballot(cnt == 0) & 15u;
      const int by = byb * RBLK, bx = bxb * RBLK;
      if (em == 15u) {
        const size_t pix0 = ((size_t)n * H + by) * H + bx;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        float* im = reinterpret_cast<float*>(out.imgs) + pix0 + (size_t)n * 2 * HW + off4;
        *reinterpret_cast<float4*>(im) = z;
        *reinterpret_cast<float4*>(im + HW) = z;
        *reinterpret_cast<float4*>(im + 2 * HW) = z;
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(out.sil) + pix0 + off4) = z;
        *reinterpret_cast<int4*>(out.tidx + pix0 + off4) = make_int4(-1, -1, -1, -1);
        typedef long long ll2 __attribute__((ext_vector_type(2)));
        ll2 v; v.x = -1; v.y = -1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int i = lane + 64 * j;               // 16-byte piece of the quad's 8 rows x 256 bytes of ids
          *reinterpret_cast<ll2*>(reinterpret_cast<int64_t*>(out.p2f) + pix0 + (size_t)(i >> 4) * H + 2 * (i & 15)) = v;
        }
        if (out.lpart && lane < 4)
          out.lpart[(((size_t)n * tiles + byb) * tiles + bxb + lane) * 4] = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
#pragma unroll 1
        for (int b = 0; b < 4; ++b)
          if ((em >> b) & 1u) fwd_fill_block_whole<1, true>(out, n, by, bx + b * RBLK, H, lane, fl);
      }
    }
    work_j0 = sc.stride - 1 - sc.j0;
  } else {
    const int n_empty = sc.per - sc.n_work, chunk = (n_empty + sc.stride - 1) / sc.stride;
    const int e0 = sc.n_work + sc.j0 * chunk, e1 = min(sc.per, e0 + chunk);
    if ((H & (RBLK - 1)) == 0) {
      const FillLane<1> fl = make_fill_lane<1>(H, lane);
#pragma unroll 1
      for (int e = e0; e < e1; ++e) {
        int n, by, bx;
        entry_block(ws.order[(size_t)sc.g * sc.per + e], sc, H, n, by, bx);
        fwd_fill_block_whole<1, true>(out, n, by, bx, H, lane, fl);
      }
    } else {
#pragma unroll 1
      for (int e = e0; e < e1; ++e) {
        int n, by, bx;
        entry_block(ws.order[(size_t)sc.g * sc.per + e], sc, H, n, by, bx);
        fwd_fill_block<1, true>(out, n, by, bx, H, lane);
      }
    }
  }
#pragma unroll 1
  for (int e = work_j0; e < sc.e_end; e += sc.stride) {
    const Tile t = make_tile(ws, sc, e, N, H, false);
    unsigned long long bestkey = KEY_NONE;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f;
    if (t.valid) {
      const int f = out.cover_in[t.pix];
      if (f >= 0) {
        const FaceRec& r = ws.rec[(size_t)t.n * F + f];
        const float4 ra = r.a, rb = r.b, rc = r.c;
        Hit h;
        bool inside = false;
        test_face_depth<CLIP, true>(t.xf, t.yf, ra, rb, rc.x, rc.z, rc.w, h, inside);
        bestkey = make_key(h.pz, f); b0 = h.c0; b1 = h.c1; b2 = h.c2;
      }
    }
    k1_finish<CLIP, true>(ws, t, F, H, sigma, out, bestkey, 0.f, b0, b1, b2, true);
  }
}

// ------------------------------------------------------------------------------- backward
// Sum over the 16 lanes of a DPP row (= one 4x4 pixel block): four row shifts, the total lands
// in lane 15 of the row.
__device__ __forceinline__ float row_sum_dpp(float v) {
#define ACFM_DPP_ADD(ctrl) \
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, true))
  ACFM_DPP_ADD(0x111);  // row_shr:1
  ACFM_DPP_ADD(0x112);  // row_shr:2
  ACFM_DPP_ADD(0x114);  // row_shr:4
  ACFM_DPP_ADD(0x118);  // row_shr:8
#undef ACFM_DPP_ADD
  return v;
}

// PointLineDistanceBackward with the clamped t held constant.  t is the forward's own clamped
// parameter (point_line_dist, same expression; 1 for a degenerate segment: then q = b exactly, the gradient of a
// is g 0 e = 0 and that of b is g 2 (b - p) = -2 (p - b) g, the degenerate branch of the reference bit for bit).
__device__ __forceinline__ void point_line_dist_bwd(float px, float py, float ax, float ay, float bx,
                                                    float by, float t, float g, float& gax, float& gay,
                                                    float& gbx, float& gby) {
  const float qx = (1.0f - t) * ax + t * bx, qy = (1.0f - t) * ay + t * by;
  const float ex = 2.0f * (qx - px), ey = 2.0f * (qy - py);
  gax = g * (1.0f - t) * ex; gay = g * (1.0f - t) * ey;
  gbx = g * t * ex; gby = g * t * ey;
}

#ifndef ACFM_BWD_CAP
#define ACFM_BWD_CAP 64
#endif
constexpr int BWD_CAP = ACFM_BWD_CAP;   // candidate-list capacity of the backward (LDS per wave: 108 B per slot)
using BwdList = CandListT<BWD_CAP>;
// Upstream gradient of the mask: either given per pixel (grad_mask) or, for the fused render+loss operator,
// formed on the fly from the references and the per-mesh gradients of the four loss terms -- k_mask_losses_bwd's
// expression, operation for operation: go0 sign(m - g) / HW + go1 g + go2 (1 - g) + go3 e / HW.
struct BwdGrad {
  const float* grad_mask;    // [N,H,H] (always float), or null: fused
  const void* lgt;           // [lrb,H,H] real_t (may be null)
  const void* ledt;          // [lrb,H,H] real_t (may be null)
  const float* go;           // [N,4]
  int lrb;
  int h16;                   // mask / lgt / ledt are half
};
// Deterministic accumulation (AcfmRasterTuning.flags bit 0): every row sum (a fixed DPP tree of values that are
// themselves computed deterministically) is converted to 64-bit fixed point (2^-36 units) before it is added to
// the candidate's LDS accumulator and, from there, to the vertex's accumulator in memory -- integer addition is
// associative, so the result does not depend on the order in which blocks, rows and atomics happen to be
// served: two runs are bit-identical.  Rounding each contribution to 2^-36 (1.5e-11) keeps it within 1e-6 of the
// floating-point mode at the gradient scales of this problem (contributions up to ~1, sums up to ~1e3 of 2^27).
constexpr float FIX_SCALE = 68719476736.0f;          // 2^36
constexpr float FIX_INV = 1.0f / 68719476736.0f;
__device__ __forceinline__ void acc_add(float* p, float v) { atomicAdd(p, v); }
__device__ __forceinline__ void acc_add(long long* p, float v) {
  atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__float2ll_rn(v * FIX_SCALE));
}
__device__ __forceinline__ void acc_add_raw(float* p, float v) { atomicAdd(p, v); }
__device__ __forceinline__ void acc_add_raw(long long* p, long long v) {
  atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v);
}
#ifndef ACFM_BWD_SHARE
#define ACFM_BWD_SHARE 1
#endif
template <class AccT>
__device__ __forceinline__ void sil_bwd_block(const RasterWs& ws, const Tile& t, const void* __restrict__ mask,
                                              const unsigned long long* __restrict__ kth,
                                              const BwdGrad& bg, int V, int F, int H, float blur,
                                              float sigma, BwdList& L, fl_t* s_fl, AccT (*s_acc)[6]) {
  // d mask / d sd_k = -(1 - mask) * p_k / sigma   (robust form).  mask == 0
  // exactly means no face contributes (every p_k is 0 or the pixel is empty): no gradient.
  float coef = 0.f;
  unsigned long long kthkey = KEY_NONE;
  const float sig_scale = sigmoid_scale(sigma);
  if (t.empty) return;  // the forward wrote mask = 0 here
  if (t.valid) {
    const float m = ld_real(mask, t.pix, bg.h16);
    if (m != 0.0f) {
      float gm;
      if (bg.grad_mask) {
        gm = bg.grad_mask[t.pix];
      } else {
        const size_t HW = (size_t)H * H;
        const size_t rp = t.pix - (size_t)t.n * HW + (size_t)(t.n % bg.lrb) * HW;
        const float g = bg.lgt ? ld_real(bg.lgt, rp, bg.h16) : 0.f, e = bg.ledt ? ld_real(bg.ledt, rp, bg.h16) : 0.f;
        const float inv = 1.0f / (float)HW;
        const float* go = bg.go + 4 * (size_t)t.n;
        const float g0 = go[0] * inv, g1 = go[1], g2 = go[2], g3 = go[3] * inv;
        const float df = m - g;
        const float sgn = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);
        gm = g0 * sgn + g1 * g + g2 * (1.0f - g) + g3 * e;
      }
      coef = -gm * (1.0f - m) / sigma;
      kthkey = kth[t.pix];
    }
  }
  const bool work = (coef != 0.0f);
  if (__ballot(work) == 0ull) return;

  for (int i = t.tid; i < BWD_CAP * 6; i += RT) (&s_acc[0][0])[i] = (AccT)0;
  wave_lds_sync();
  AccT* gout;
  if constexpr (sizeof(AccT) == 8) gout = reinterpret_cast<AccT*>(ws.grad_fix) + (size_t)t.n * V * 2;
  else gout = reinterpret_cast<AccT*>(ws.grad_ndc) + (size_t)t.n * V * 2;

  // the pixel this lane stands in for when its group helps its partner group (walk_wave<.., SHARE>): that lane's
  // gradient coefficient and K-th key, fetched once per walk
  float coef_p = 0.f;
  unsigned long long kth_p = KEY_NONE;
  bin_and_walk(ws, t, F, H, L, s_fl, 0.f, [&](int list_n) {
#ifdef ACFM_DIAG_BWD_NO_WALK
    if (list_n >= 0) return;
#endif
    walk_wave<false, ACFM_BWD_SHARE != 0>(L, t, H, list_n, blur, nullptr,
#if ACFM_BWD_SHARE
                                          [&](const Cand& cd, bool in_box, int ord, bool helping, float exf, float eyf) {
      const float coef_e = helping ? coef_p : coef;
      const unsigned long long kth_e = helping ? kth_p : kthkey;
      bool member = (coef_e != 0.0f) && in_box;
#else
                                          [&](const Cand& cd, bool in_box, int ord) {
      const float exf = t.xf, eyf = t.yf, coef_e = coef;
      const unsigned long long kth_e = kthkey;
      bool member = work && in_box;
#endif
      if (__ballot(member) == 0ull) return;
#if ACFM_BWD_EDGE_GLOBAL
      const FaceRec& grec = ws.rec[(size_t)t.n * F + cd.fid];
      const float4 e0g = grec.e0, e1g = grec.e1;
#endif
      const float4 A = cd.a, B = cd.b;
      Hit h;
      h.pz = 0.f; h.sd = 0.f; h.d01 = 0.f; h.d02 = 0.f; h.d12 = 0.f;
      bool inside = false;
      // stage 1 (depth): only faces at or before the pixel's K-th kept face took part in the
      // blend; the others are dropped before their edge distances are computed
      member = member && test_face_depth<false>(exf, eyf, A, B, cd.c.x, cd.c.y, cd.rden, h, inside);
      member = member && (make_key(h.pz, cd.fid) <= kth_e);
      if (__ballot(member) == 0ull) return;
      float tpar[3] = {0.f, 0.f, 0.f};
#if ACFM_BWD_EDGE_GLOBAL
      // ACFM_BWD_EDGE_GLOBAL: the per-edge constants (|e|^2, refined 1/|e|^2) of the face from its record in memory
      // (L2-resident: 128 B x 1280 faces per mesh) instead of recomputing them per pixel; requested before the depth
      // stage so that its ~100 instructions cover the latency.  A face with a degenerate edge takes the unfactored path.
      if (__ballot(e1g.z != 0.0f) != 0ull) {
        if (member) member = test_face_dist(exf, eyf, A, B, blur, inside, h, tpar);
      } else {
        if (member) member = test_face_dist_e(exf, eyf, A, B, e0g, e1g, blur, inside, h, tpar);
      }
#else
      if (member) member = test_face_dist(exf, eyf, A, B, blur, inside, h, tpar);
#endif
      if (__ballot(member) == 0ull) return;
      float g0x = 0.f, g0y = 0.f, g1x = 0.f, g1y = 0.f, g2x = 0.f, g2y = 0.f;
      if (member) {
        const float x0 = A.x, y0 = A.y, x1 = A.z, x2 = A.w, y1 = B.x, y2 = B.y;
        // inside <=> sd < 0: an inside pixel lies on no edge, so d > 0 and sd = -d < 0
        const bool inside = h.sd < 0.0f;
        // (1-ulp reciprocal in the sigmoid: 1e-7 relative on a gradient checked to 1e-4)
        const float gs = coef_e * sigmoid_neg_fast(h.sd, sigma, sig_scale);   // dL / d sd
        const float gd = inside ? -gs : gs;                      // sd = inside ? -d : d
        // the arg-min edge (01 first, then 02, then 12: SURVEY App-A.4), chosen with selects so that the
        // wave runs ONE distance backward instead of up to three divergent copies of it
        const bool e01 = h.d01 <= h.d02 && h.d01 <= h.d12;
        const bool e02 = !e01 && h.d02 <= h.d01 && h.d02 <= h.d12;
        const float ax = (e01 || e02) ? x0 : x1, ay = (e01 || e02) ? y0 : y1;
        const float bx = e01 ? x1 : x2, by = e01 ? y1 : y2;
        const float tq = e01 ? tpar[0] : (e02 ? tpar[1] : tpar[2]);
        float ax_, ay_, bx_, by_;
        point_line_dist_bwd(exf, eyf, ax, ay, bx, by, tq, gd, ax_, ay_, bx_, by_);
        if (e01) { g0x = ax_; g0y = ay_; g1x = bx_; g1y = by_; }
        else if (e02) { g0x = ax_; g0y = ay_; g2x = bx_; g2y = by_; }
        else { g1x = ax_; g1y = ay_; g2x = bx_; g2y = by_; }
      }
      // The 16 lanes of a row share the face: sum their six contributions.  First the two lanes of a pair swap
      // halves -- the even lane keeps (g0x, g0y, g1x) of both, the odd lane (g1y, g2x, g2y) -- then three values
      // instead of six go through the row shifts (by 2, 4, 8: lanes of one parity): 6 selects + 12 DPP adds instead
      // of 24, and lanes 14 and 15 of the row add three sums each.  A fixed tree: deterministic.
      {
        const bool odd = (t.lane & 1) != 0;
        float k0 = odd ? g1y : g0x, k1 = odd ? g2x : g0y, k2 = odd ? g2y : g1x;
        const float s0 = odd ? g0x : g1y, s1 = odd ? g0y : g2x, s2 = odd ? g1x : g2y;
#define ACFM_DPP_GET(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, true))
        k0 += ACFM_DPP_GET(s0, 0xb1); k1 += ACFM_DPP_GET(s1, 0xb1); k2 += ACFM_DPP_GET(s2, 0xb1);   // quad_perm:[1,0,3,2]
        k0 += ACFM_DPP_GET(k0, 0x112); k1 += ACFM_DPP_GET(k1, 0x112); k2 += ACFM_DPP_GET(k2, 0x112); // row_shr:2
        k0 += ACFM_DPP_GET(k0, 0x114); k1 += ACFM_DPP_GET(k1, 0x114); k2 += ACFM_DPP_GET(k2, 0x114); // row_shr:4
        k0 += ACFM_DPP_GET(k0, 0x118); k1 += ACFM_DPP_GET(k1, 0x118); k2 += ACFM_DPP_GET(k2, 0x118); // row_shr:8
#undef ACFM_DPP_GET
        // (a row without a face this iteration, or without members, sums to exact zeros)
        if ((t.lane & 15) >= 14) {
          AccT* acc = s_acc[cd.idx] + (odd ? 3 : 0);   // two groups can hold the same face in one iteration: atomics
          if (k0 != 0.f) acc_add(&acc[0], k0);
          if (k1 != 0.f) acc_add(&acc[1], k1);
          if (k2 != 0.f) acc_add(&acc[2], k2);
        }
      }
    }
#if ACFM_BWD_SHARE
    , [&](int partner_lane) {
      const int src = partner_lane >= 0 ? partner_lane : t.lane;
      coef_p = __shfl(coef, src, 64);
      const unsigned lo = (unsigned)__shfl((int)(unsigned)(kthkey & 0xffffffffull), src, 64);
      const unsigned hi = (unsigned)__shfl((int)(unsigned)(kthkey >> 32), src, 64);
      kth_p = ((unsigned long long)hi << 32) | lo;
    }
#endif
    );
    wave_lds_sync();
#ifdef ACFM_DIAG_BWD_NO_FLUSH
    if (list_n >= 0) return;
#endif
    for (int c = t.lane; c < list_n; c += RT) {
      AccT* acc = s_acc[c];
      const AccT z = (AccT)0;
      const AccT a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3], a4 = acc[4], a5 = acc[5];
      if (a0 != z || a1 != z || a2 != z || a3 != z || a4 != z || a5 != z) {
        const int4 vi = ws.vidx[(size_t)t.n * F + __float_as_int(L.c[c].w)];
        if (a0 != z) acc_add_raw(&gout[2 * vi.x], a0);
        if (a1 != z) acc_add_raw(&gout[2 * vi.x + 1], a1);
        if (a2 != z) acc_add_raw(&gout[2 * vi.y], a2);
        if (a3 != z) acc_add_raw(&gout[2 * vi.y + 1], a3);
        if (a4 != z) acc_add_raw(&gout[2 * vi.z], a4);
        if (a5 != z) acc_add_raw(&gout[2 * vi.z + 1], a5);
        acc[0] = z; acc[1] = z; acc[2] = z; acc[3] = z; acc[4] = z; acc[5] = z;
      }
    }
  });
}

// waves per SIMD the backward is compiled for: with the shared walk 5 (no spills; 156.9 us in the benchmark step) beats
// 6 (80 VGPRs + 28 B of scratch; 160.5 us); without it 6 had measured best
#ifndef ACFM_BWD_WAVES
#define ACFM_BWD_WAVES 5
#endif
template <class AccT>
__global__ __launch_bounds__(RT, ACFM_BWD_WAVES) void k_sil_bwd(RasterWs ws, const void* __restrict__ mask,
                                                 const unsigned long long* __restrict__ kth,
                                                 BwdGrad grad_mask, int N, int V,
                                                 int F, int H, float blur, float sigma) {
  __shared__ BwdList L;
  __shared__ fl_t s_fl[FLCAP];
  // gradient accumulator per list slot: (d/dx0, d/dy0, d/dx1, d/dy1, d/dx2, d/dy2) of that face,
  // summed over the block's pixels; flushed (global float atomics on the face's three vertices)
  // and cleared after every walk.  (A [V][2] vertex accumulator per block merges more before
  // going to memory but costs 5 KB of LDS per wave at V = 642 and a clear + scan per block.)
  __shared__ AccT s_acc[BWD_CAP][6];
  const Sched sc = make_sched(ws, N, H, true);
#pragma unroll 1
  for (int e = sc.j0; e < sc.e_end; e += sc.stride) {
    const Tile t = make_tile(ws, sc, e, N, H, true);
    if (!t.none) sil_bwd_block(ws, t, mask, kth, grad_mask, V, F, H, blur, sigma, L, s_fl, s_acc);
    wave_lds_sync();
  }
}

// Fused render + loss, finish: per mesh, the block partials of the raster kernel and sum(gt) over the whole image
// (the blocks without work have m = 0: |m - g| = g, m + g - m g = g) -> out[n] = (mean|m - g|, sum m g,
// sum(m + g - m g), mean e m), the [N,4] vector of k_mask_losses.  Two short launches, FIN_CHUNKS workgroups per mesh
// in the first (one workgroup per mesh was latency-bound: 45 us for 64 meshes); every sum is formed in a fixed
// order (thread-strided partial sums, a fixed tree, then the chunks in order): deterministic, no atomics.
constexpr int FIN_MAX_CHUNKS = 64;   // (sizes ws.lpart2)
static int fin_chunks(int N) {       // enough workgroups to fill the chip at any batch size: ~2048 in all
  int c = 8;
  while (c < FIN_MAX_CHUNKS && c * N < 2048) c *= 2;
  return c;
}
__global__ __launch_bounds__(TPB) void k_sil_loss_finish1(const float4* __restrict__ lpart, const void* __restrict__ gt,
                                                           int tt, int HW, int RB, int h16, float* __restrict__ part2) {
  __shared__ float s_red[TPB][5];
  const int n = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x, FIN_CHUNKS = gridDim.x;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, gs = 0.f;
  const float4* p = lpart + (size_t)n * tt * 4;
  const int np = tt * 4, p_lo = (int)((long long)np * ch / FIN_CHUNKS), p_hi = (int)((long long)np * (ch + 1) / FIN_CHUNKS);
  for (int i = p_lo + tid; i < p_hi; i += TPB) {
    const float4 v = p[i];
    a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
  }
  if (gt) {
    const size_t go = (size_t)(n % RB) * HW;
    if ((HW & 3) == 0) {
      const int q = HW / 4, q_lo = (int)((long long)q * ch / FIN_CHUNKS), q_hi = (int)((long long)q * (ch + 1) / FIN_CHUNKS);
      constexpr int U = 8;                       // loads of a round in flight together
      for (int i0 = q_lo + tid; i0 < q_hi; i0 += TPB * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = i0 + u * TPB;
          v[u] = i < q_hi ? ld4_real(gt, go / 4 + i, h16) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) gs += (v[u].x + v[u].y) + (v[u].z + v[u].w);
      }
    } else {
      const int g_lo = (int)((long long)HW * ch / FIN_CHUNKS), g_hi = (int)((long long)HW * (ch + 1) / FIN_CHUNKS);
      for (int i = g_lo + tid; i < g_hi; i += TPB) gs += ld_real(gt, go + i, h16);
    }
  }
  s_red[tid][0] = a0; s_red[tid][1] = a1; s_red[tid][2] = a2; s_red[tid][3] = a3; s_red[tid][4] = gs;
  __syncthreads();
  for (int s = TPB / 2; s > 0; s >>= 1) {
    if (tid < s)
#pragma unroll
      for (int k = 0; k < 5; ++k) s_red[tid][k] += s_red[tid + s][k];
    __syncthreads();
  }
  if (tid < 5) part2[((size_t)n * FIN_CHUNKS + ch) * 5 + tid] = s_red[0][tid];
}
__global__ void k_sil_loss_finish2(const float* __restrict__ part2, int N, int HW, int FIN_CHUNKS, float* __restrict__ out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < FIN_CHUNKS; ++c)
#pragma unroll
    for (int k = 0; k < 5; ++k) a[k] += part2[((size_t)n * FIN_CHUNKS + c) * 5 + k];
  const float hw = (float)HW;
  out[4 * (size_t)n + 0] = (a[4] + a[0]) / hw;
  out[4 * (size_t)n + 1] = a[1];
  out[4 * (size_t)n + 2] = a[4] + a[2];
  out[4 * (size_t)n + 3] = a[3] / hw;
}

// Fused texture render + masked MSE, finish: out[n] = (sum over the mesh's blocks of their partial
// + sum_c sum_px (img_c m)^2) / (3 HW) -- the second term is what an uncovered pixel (tex = 0) contributes.
__global__ __launch_bounds__(TPB) void k_tex_loss_finish1(const float4* __restrict__ lpart, const void* __restrict__ timg,
                                                           const void* __restrict__ tmask, int tt, int HW, int RB,
                                                           int h16, float* __restrict__ part2) {
  __shared__ float s_red[TPB];
  const int n = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x, FIN_CHUNKS = gridDim.x;
  float acc = 0.f;
  const float4* p = lpart + (size_t)n * tt * 4;
  const int t_lo = (int)((long long)tt * ch / FIN_CHUNKS), t_hi = (int)((long long)tt * (ch + 1) / FIN_CHUNKS);
  for (int i = t_lo + tid; i < t_hi; i += TPB) acc += p[4 * (size_t)i].x;
  const size_t rn = (size_t)(n % RB);
  const size_t mo = rn * HW, io = rn * 3 * HW;
  if ((HW & 3) == 0) {
    const int q = HW / 4, q_lo = (int)((long long)q * ch / FIN_CHUNKS), q_hi = (int)((long long)q * (ch + 1) / FIN_CHUNKS);
    constexpr int U = 4;
    for (int i0 = q_lo + tid; i0 < q_hi; i0 += TPB * U) {
      float4 mk[U], c0[U], c1[U], c2[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * TPB;
        const bool in = i < q_hi;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        mk[u] = in ? ld4_real(tmask, mo / 4 + i, h16) : z;
        c0[u] = in ? ld4_real(timg, io / 4 + i, h16) : z;
        c1[u] = in ? ld4_real(timg, (io + HW) / 4 + i, h16) : z;
        c2[u] = in ? ld4_real(timg, (io + 2 * (size_t)HW) / 4 + i, h16) : z;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        auto sq = [](float a, float b) { const float v = a * b; return v * v; };
        acc += (sq(c0[u].x, mk[u].x) + sq(c1[u].x, mk[u].x) + sq(c2[u].x, mk[u].x)) +
               (sq(c0[u].y, mk[u].y) + sq(c1[u].y, mk[u].y) + sq(c2[u].y, mk[u].y)) +
               (sq(c0[u].z, mk[u].z) + sq(c1[u].z, mk[u].z) + sq(c2[u].z, mk[u].z)) +
               (sq(c0[u].w, mk[u].w) + sq(c1[u].w, mk[u].w) + sq(c2[u].w, mk[u].w));
      }
    }
  } else {
    const int g_lo = (int)((long long)HW * ch / FIN_CHUNKS), g_hi = (int)((long long)HW * (ch + 1) / FIN_CHUNKS);
    for (int i = g_lo + tid; i < g_hi; i += TPB) {
      const float mk = ld_real(tmask, mo + i, h16);
      const float b0 = ld_real(timg, io + i, h16) * mk, b1 = ld_real(timg, io + HW + i, h16) * mk,
                  b2 = ld_real(timg, io + 2 * (size_t)HW + i, h16) * mk;
      acc += b0 * b0 + b1 * b1 + b2 * b2;
    }
  }
  s_red[tid] = acc;
  __syncthreads();
  for (int s = TPB / 2; s > 0; s >>= 1) {
    if (tid < s) s_red[tid] += s_red[tid + s];
    __syncthreads();
  }
  if (tid == 0) part2[(size_t)n * FIN_CHUNKS + ch] = s_red[0];
}
__global__ void k_tex_loss_finish2(const float* __restrict__ part2, int N, int HW, int FIN_CHUNKS, float* __restrict__ out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float a = 0.f;
  for (int c = 0; c < FIN_CHUNKS; ++c) a += part2[(size_t)n * FIN_CHUNKS + c];
  out[n] = a / (3.0f * (float)HW);
}

// ------------------------------------------------------------------------------- projection
template <bool XY>   // XY: only (x, y) are stored, [N,V,2] (orthographic_proj / project_points)
__global__ __launch_bounds__(TPB) void k_project(const float* __restrict__ verts,
                                                 const float* __restrict__ cams, int V, float offset_z,
                                                 float* __restrict__ proj) {
  const int n = blockIdx.y;
  const int v = blockIdx.x * TPB + threadIdx.x;
  if (v >= V) return;
  const float* x = verts + ((size_t)n * V + v) * 3;
  float px, py, pz;
  project_point(cams + 7 * (size_t)n, x[0], x[1], x[2], offset_z, px, py, pz);
  if (XY) {
    float* o = proj + ((size_t)n * V + v) * 2;
    o[0] = px; o[1] = py;
  } else {
    float* o = proj + ((size_t)n * V + v) * 3;
    o[0] = px; o[1] = py; o[2] = pz;
  }
}

// Backward of proj = s * rot(q, X) + (tx, ty, offset_z), q not normalised here:
//   r      = (q0^2 - u.u) X + 2 (u.X) u + 2 q0 (u x X)
//   dL/ds  = g.r ; dL/dt = g.xy ; with G = s g:
//   dL/dq0 = 2 q0 (G.X) + 2 G.(u x X)
//   dL/du  = -2 (G.X) u + 2 (G.u) X + 2 (u.X) G + 2 q0 (X x G)
//   dL/dX  = (q0^2 - u.u) G + 2 (G.u) u + 2 q0 (G x u)
// MODE 1 (NDC2): the upstream gradient is grad_ndc [N,V,2] of the rasteriser
// (x_ndc = -x_p, y_ndc = -y_p, no z gradient on the silhouette path); MODE 2: the gradient of the
// (x, y) projection [N,V,2] as it is; MODE 0: all three components [N,V,3].
template <int MODE>
__global__ __launch_bounds__(TPB) void k_project_bwd(const float* __restrict__ verts,
                                                     const float* __restrict__ cams,
                                                     float* gin /* NDC2: cleared after reading */, int V,
                                                     float* __restrict__ grad_verts,
                                                     float* __restrict__ grad_cams,
                                                     const float* __restrict__ gproj = nullptr /* NDC modes: + [N,V,2] */) {
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
    if (MODE == 3) {            // NDC2 in 2^-36 fixed point (deterministic backward)
      long long* g = reinterpret_cast<long long*>(gin) + ((size_t)n * V + v) * 2;
      gx = -((float)g[0] * FIX_INV); gy = -((float)g[1] * FIX_INV); gz = 0.f;
      g[0] = 0; g[1] = 0;
      if (gproj) { gx += gproj[((size_t)n * V + v) * 2]; gy += gproj[((size_t)n * V + v) * 2 + 1]; }
    } else if (MODE == 1) {
      float* g = gin + ((size_t)n * V + v) * 2;
      gx = -g[0]; gy = -g[1]; gz = 0.f;
      g[0] = 0.f; g[1] = 0.f;   // the raster workspace's NDC-gradient scratch is left zeroed for the next backward
      // the gradient of the projection the forward handed out (AcfmSilExtras.proj_xy: a second consumer of the same
      // vertices and cameras, e.g. the boundary loss): one projection backward for both, no gradient sum afterwards
      if (gproj) { gx += gproj[((size_t)n * V + v) * 2]; gy += gproj[((size_t)n * V + v) * 2 + 1]; }
    } else if (MODE == 2) {
      const float* g = gin + ((size_t)n * V + v) * 2;
      gx = g[0]; gy = g[1]; gz = 0.f;
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
// Gather form of the same gradient, one wave per four (atlas, face) slots: for each it visits the pixels of the
// face's box in every mesh that samples this atlas (the G hypotheses of a frame), adds the
// gradients of the pixels whose texel belongs to the face into 3 R^2 LDS accumulators and stores
// the face's texels -- zeros included -- with plain coalesced stores.  No global atomics (agent-
// scope float atomics execute at the memory side on this multi-XCD part: 2 M of them took 87 us)
// and no zero fill of the 35 MB gradient.  Needs the face boxes of the forward's workspace.
// i / w and i % w for 0 <= i < 2^23, 0 < w < 2^12 without the ~35-instruction integer division
__device__ __forceinline__ void divmod_small(int i, int w, float rw, int& q, int& r) {
  q = (int)((float)i * rw);
  r = i - q * w;
  if (r < 0) { --q; r += w; }
  if (r >= w) { ++q; r -= w; }
}
constexpr int TEXG_MAX_R = 8;
#ifndef ACFM_TEXG_FPW
#define ACFM_TEXG_FPW 4
#endif
constexpr int TEXG_FPW = ACFM_TEXG_FPW;      // faces per wave: their boxes, texel indices and gradients are loaded side by side
#ifndef ACFM_TEXG_U
#define ACFM_TEXG_U 2    // (with 8 waves per SIMD below: 39.8 us per launch; U = 4 at 6 waves 41.4; U = 8 at 5 waves -- 92 VGPRs -- 45.2)
#endif
#ifndef ACFM_TEXG_WAVES
#define ACFM_TEXG_WAVES 8
#endif
constexpr int TEXG_U = ACFM_TEXG_U;        // big boxes: 64 U pixels per round, all their loads in flight together
// Upstream gradient of the rendered image: given ([N,3,H,H]) or, for the fused texture render + masked MSE, formed
// on the fly from the rendered image, the reference image and mask and the per-mesh gradient of the loss --
// k_tex_mse_bwd's expression: w (tex m - img m) m with w = go[n] 2 / (3 HW).
struct TexGrad {
  const float* grad_imgs;    // [N,3,H,H] (always float), or null: fused
  const void* imgs;          // [N,3,H,H] real_t, the forward's output
  const void* timg;          // [rb,3,H,H] real_t
  const void* tmask;         // [rb,H,H] real_t
  const float* go;           // [N]
  int rb;
  int h16;
};
struct TexGradN {            // the same for one mesh n
  const float* g;
  const void *im, *ri, *rm;
  size_t io, ro, mo;
  float w;
  size_t HW;
  int h16;
  __device__ __forceinline__ void load(size_t p, float& r, float& gg, float& b) const {
    if (g) { r = g[p]; gg = g[HW + p]; b = g[2 * HW + p]; return; }
    const float mk = ld_real(rm, mo + p, h16);
    r = w * (ld_real(im, io + p, h16) * mk - ld_real(ri, ro + p, h16) * mk) * mk;
    gg = w * (ld_real(im, io + HW + p, h16) * mk - ld_real(ri, ro + HW + p, h16) * mk) * mk;
    b = w * (ld_real(im, io + 2 * HW + p, h16) * mk - ld_real(ri, ro + 2 * HW + p, h16) * mk) * mk;
  }
};
__device__ __forceinline__ TexGradN tex_grad_of(const TexGrad& tg, int n, size_t HW) {
  TexGradN t = {};
  t.HW = HW;
  if (tg.grad_imgs) { t.g = tg.grad_imgs + (size_t)n * 3 * HW; return t; }
  const size_t rn = (size_t)(n % tg.rb);
  t.im = tg.imgs; t.ri = tg.timg; t.rm = tg.tmask; t.h16 = tg.h16;
  t.io = (size_t)n * 3 * HW; t.ro = rn * 3 * HW; t.mo = rn * HW;
  t.w = tg.go[n] * 2.0f / (3.0f * (float)HW);
  return t;
}
__global__ __launch_bounds__(256, ACFM_TEXG_WAVES) void k_tex_bwd_faces(RasterWs ws, TexGrad tgrad,
                                                       const int32_t* __restrict__ tidx, int N, int F, int H,
                                                       int R, int NA, float box_shrink,
                                                       float* __restrict__ grad_atlas) {
  __shared__ float s_acc[4][TEXG_FPW][3 * TEXG_MAX_R * TEXG_MAX_R];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // Wave q of the launch takes the faces q, q + Q, q + 2Q, q + 3Q of one atlas (Q = ceil(F / FPW)):
  // neighbouring faces of a mesh tend to be large together, and a wave walks its faces' boxes one
  // after the other, so they are dealt to different waves.
  const int Q = (F + TEXG_FPW - 1) / TEXG_FPW;
  const long long wq = (long long)blockIdx.x * 4 + wv;
  if (wq >= (long long)NA * Q) return;                       // (whole wave; no workgroup barriers below)
  const int a = (int)(wq / Q), f0 = (int)(wq % Q);
  const int R2 = R * R, n3 = 3 * R2;
  float (*acc)[3 * TEXG_MAX_R * TEXG_MAX_R] = s_acc[wv];
#pragma unroll
  for (int k = 0; k < TEXG_FPW; ++k)
    for (int i = lane; i < n3; i += 64) acc[k][i] = 0.f;
  wave_lds_sync();
  const size_t HW = (size_t)H * H;
  const float hf = (float)H;
  // lane k < FPW looks after face f0 + k Q
  const int my_f = f0 + (lane < TEXG_FPW ? lane : 0) * Q;
  const bool my_live = lane < TEXG_FPW && my_f < F;
  const int G = N / NA;
  for (int g = 0; g < G; ++g) {
    // the FPW boxes (mesh a + g NA), one per lane, turned into pixel ranges (k_setup's formula, one pixel of slack)
    int xa = 0, ya = 0, w = 1, cnt = 0;
    const int n = a + g * NA;
    if (my_live && ws.fvis[(size_t)n * F + my_f]) {          // (a face no pixel shows has no gradient: zeros)
      float4 b = ws.rec[(size_t)n * F + my_f].box;
      b.x += box_shrink; b.y -= box_shrink; b.z += box_shrink; b.w -= box_shrink;
      if (b.x <= b.y && b.z <= b.w) {                        // not a degenerate face (inf, -inf, ..) or an emptied box
        // pixel range of the box: k_setup's formula with one pixel of slack, then tightened to the
        // pixels that pass the forward's own test (pixel centre inside the box, same float expressions)
        xa = (int)floorf(hf - 1.0f - ((b.y + 1.0f) * hf - 1.0f) * 0.5f) - 1;
        int xb = (int)ceilf(hf - 1.0f - ((b.x + 1.0f) * hf - 1.0f) * 0.5f) + 1;
        ya = (int)floorf(hf - 1.0f - ((b.w + 1.0f) * hf - 1.0f) * 0.5f) - 1;
        int yb = (int)ceilf(hf - 1.0f - ((b.z + 1.0f) * hf - 1.0f) * 0.5f) + 1;
        if (!(xb < 0 || yb < 0 || xa >= H || ya >= H)) {
          xa = max(xa, 0); ya = max(ya, 0); xb = min(xb, H - 1); yb = min(yb, H - 1);
          for (int it = 0; it < 3 && xa <= xb && pix_to_ndc(H - 1 - xa, H) > b.y; ++it) ++xa;
          for (int it = 0; it < 3 && xa <= xb && pix_to_ndc(H - 1 - xb, H) < b.x; ++it) --xb;
          for (int it = 0; it < 3 && ya <= yb && pix_to_ndc(H - 1 - ya, H) > b.w; ++it) ++ya;
          for (int it = 0; it < 3 && ya <= yb && pix_to_ndc(H - 1 - yb, H) < b.z; ++it) --yb;
          if (xa <= xb && ya <= yb) { w = xb - xa + 1; cnt = w * (yb - ya + 1); }
        }
      }
    }
    const int32_t* tn = tidx + (size_t)n * HW;
    const TexGradN gn = tex_grad_of(tgrad, n, HW);
    int cmax = 0;
    int t[TEXG_FPW];
    size_t pp[TEXG_FPW];
#pragma unroll
    for (int k = 0; k < TEXG_FPW; ++k) {                     // all texel-index loads first ...
      const int kxa = __shfl(xa, k, 64), kya = __shfl(ya, k, 64), kw = __shfl(w, k, 64), kc = __shfl(cnt, k, 64);
      const int kbase = (a * F + f0 + k * Q) * R2;           // first texel index of the face (< 2^31: host check)
      cmax = max(cmax, kc);
      t[k] = -1;
      pp[k] = 0;
      if (lane < kc) {
        int qy, qx;
        divmod_small(lane, kw, __builtin_amdgcn_rcpf((float)kw), qy, qx);
        pp[k] = (size_t)(kya + qy) * H + (kxa + qx);
        t[k] = tn[pp[k]] - kbase;
      }
    }
#pragma unroll
    for (int k = 0; k < TEXG_FPW; ++k) {                     // ... then the gradients of the pixels that belong to the face
      if (t[k] >= 0 && t[k] < R2) {
        // d rgb / d texel = wnum / (wnum + delta) = 1 in fp32 (wnum >= 0.5, delta = 1e-10)
        float r, gg, bb;
        gn.load(pp[k], r, gg, bb);
        atomicAdd(&acc[k][3 * t[k] + 0], r);
        atomicAdd(&acc[k][3 * t[k] + 1], gg);
        atomicAdd(&acc[k][3 * t[k] + 2], bb);
      }
    }
    if (cmax > 64) {   // boxes of more than 64 pixels (a third of the bird's): the rest face by face, 64 U pixels per round
      for (int k = 0; k < TEXG_FPW; ++k) {
        const int kxa = __shfl(xa, k, 64), kya = __shfl(ya, k, 64), kw = __shfl(w, k, 64), kc = __shfl(cnt, k, 64);
        const int kbase = (a * F + f0 + k * Q) * R2;
        const float rw = __builtin_amdgcn_rcpf((float)kw);
        for (int i0 = 64 + lane; i0 < kc + lane; i0 += 64 * TEXG_U) {   // (i0 - lane is wave-uniform)
          int tt[TEXG_U];
          float cr[TEXG_U], cg[TEXG_U], cb[TEXG_U];
#pragma unroll
          for (int u = 0; u < TEXG_U; ++u) {
            const int i = i0 + 64 * u;
            tt[u] = -1; cr[u] = 0.f; cg[u] = 0.f; cb[u] = 0.f;
            if (i < kc) {
              int qy, qx;
              divmod_small(i, kw, rw, qy, qx);
              const size_t p = (size_t)(kya + qy) * H + (kxa + qx);
              tt[u] = tn[p] - kbase;
              gn.load(p, cr[u], cg[u], cb[u]);                              // unconditionally: one round trip per round
            }
          }
#pragma unroll
          for (int u = 0; u < TEXG_U; ++u)
            if (tt[u] >= 0 && tt[u] < R2) {
              atomicAdd(&acc[k][3 * tt[u] + 0], cr[u]);
              atomicAdd(&acc[k][3 * tt[u] + 1], cg[u]);
              atomicAdd(&acc[k][3 * tt[u] + 2], cb[u]);
            }
        }
      }
    }
  }
  wave_lds_sync();
#pragma unroll
  for (int k = 0; k < TEXG_FPW; ++k) {
    if (f0 + k * Q >= F) break;
    float* o = grad_atlas + (size_t)(a * F + f0 + k * Q) * n3;
    for (int i = lane; i < n3; i += 64) o[i] = acc[k][i];
  }
}

// ------------------------------------------------------------------------------- profiling
// One ring per process, shared by all devices and threads (a measurement aid, off by default): the
// flag is atomic, the ring is guarded by a mutex held from prof_begin to prof_end of a launch.
static std::atomic<bool> g_prof_on{false};
static std::mutex g_prof_mu;
static hipEvent_t g_ev[ACFM_PROF_RING][2];
static int g_ev_id[ACFM_PROF_RING];
static int g_ev_n = 0;
static bool g_ev_made = false;
static thread_local bool t_prof_open = false;

void prof_begin(int id, hipStream_t st) {
  if (!g_prof_on.load(std::memory_order_relaxed)) return;
  g_prof_mu.lock();
  if (!g_prof_on.load() || g_ev_n >= ACFM_PROF_RING) { g_prof_mu.unlock(); return; }
  t_prof_open = true;
  g_ev_id[g_ev_n] = id;
  (void)hipEventRecord(g_ev[g_ev_n][0], st);
}
void prof_end(hipStream_t st) {
  if (!t_prof_open) return;
  (void)hipEventRecord(g_ev[g_ev_n][1], st);
  g_ev_n++;
  t_prof_open = false;
  g_prof_mu.unlock();
}

// ------------------------------------------------------------------------------- host side
static int launch_setup(const float* verts, const int64_t* faces, const float* cams, int N, int V,
                        int F, int H, float offset_z, int mode, float blur, const RasterWs& ws,
                        const Tune& tn, hipStream_t st, uint8_t* vis = nullptr, float* proj_xy = nullptr) {
  const float margin = sqrtf(blur);
  const int tiles = (H + CNT_TILE - 1) / CNT_TILE;
  const int tt = tiles * tiles;                 // cost counters
  const int blocks = (H + RBLK - 1) / RBLK;
  const int ctiles = (H + CTILE - 1) / CTILE;
  const size_t mwords = 2 * (((size_t)F + 63) / 64);
  const size_t slice_words = (size_t)setup_slice_faces(F, ws.slices) / 32;
  const size_t slice_mask_bytes = sizeof(unsigned) * (size_t)ctiles * ctiles * (slice_words < mwords ? slice_words : mwords);
  const bool lds_mask = slice_mask_bytes <= (size_t)SETUP_LDS_MASK_BYTES;
  const size_t lds = sizeof(float) * 3 * (size_t)V + (tt <= SETUP_LDS_TILES ? sizeof(int) * (size_t)tt : 0) +
                     (lds_mask ? slice_mask_bytes : 0);
  if (lds > 150 * 1024) return ACFM_E_BADARG;
  if (!lds_mask && zero_async(ws.cmask, sizeof(unsigned) * (size_t)N * ctiles * ctiles * mwords, st)) return ACFM_E_LAUNCH;
  // (counters in LDS: every slice of k_setup stores its own plane of ws.tile_part, nothing to zero)
  if (tt > SETUP_LDS_TILES && zero_async(ws.tile_cnt, sizeof(int) * (size_t)N * tt, st)) return ACFM_E_LAUNCH;
  ProfScope ps(ACFM_PROF_SETUP, st);
  switch (ws.slices) {
    case 4: hipLaunchKernelGGL(k_setup<4>, dim3(N, 4), dim3(TPB), lds, st, verts, faces, cams, V, F, H, offset_z, mode,
                               margin, ws, vis, proj_xy); break;
    case 8: hipLaunchKernelGGL(k_setup<8>, dim3(N, 8), dim3(TPB), lds, st, verts, faces, cams, V, F, H, offset_z, mode,
                               margin, ws, vis, proj_xy); break;
    default: hipLaunchKernelGGL(k_setup<16>, dim3(N, 16), dim3(TPB), lds, st, verts, faces, cams, V, F, H, offset_z, mode,
                                margin, ws, vis, proj_xy); break;
  }
  if (ws.slices > 4)
    hipLaunchKernelGGL(k_order<true>, dim3((N & 7) == 0 ? 8 : 1), dim3(1024), 0, st, ws, N, blocks * blocks, H, tn.split);
  else
    hipLaunchKernelGGL(k_order<false>, dim3((N & 7) == 0 ? 8 : 1), dim3(1024), 0, st, ws, N, blocks * blocks, H, tn.split);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

static bool bad_dims(int N, int V, int F, int H) {
  return N <= 0 || N > 65535 || V <= 0 || F <= 0 || F > ACFM_MAX_FACES || H <= 0 || H > 4096 ||
         (size_t)N * F > 0x7fffffffull ||
         (size_t)N * ((H + RBLK - 1) / RBLK) * ((H + RBLK - 1) / RBLK) > 0x7fffffffull;
}

// workgroups of a raster launch: per XCD group ceil(entries / div) (+ 4 per split slot), see Sched
static unsigned tile_grid(int N, int H, int div, int split_slots = 0) {
  const int tiles = (H + RBLK - 1) / RBLK;
  const size_t G = (N & 7) == 0 ? 8 : 1;
  const size_t per = (size_t)tiles * tiles * (N / G);
  const size_t d = div < 1 ? 1 : (size_t)div;
  return (unsigned)(G * ((per + d - 1) / d + (size_t)split_slots * 4));
}

template <int K>
static int launch_sil_fwd(const RasterWs& ws, int N, int F, int H, float blur, float sigma,
                          const FwdOut& out, const Tune& tn, hipStream_t st) {
  ProfScope ps(ACFM_PROF_SIL_FWD, st);
  hipLaunchKernelGGL((k_raster_fwd<K, false, false>), dim3(tile_grid(N, H, tn.div[0], ws.split_slots)), dim3(RT), 0, st, ws, N, F,
                     H, blur, sigma, out);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

}  // namespace acfm

using namespace acfm;

extern "C" {

// Diagnostic builds only (make DIAG=1 -> libacfm_hip_diag.so, used by tools/stamps.py and tools/occ_probe.py):
// per-workgroup time stamps and the occupancy query.  The shipping library has no such state.
#ifdef ACFM_DIAG_COUNT
int acfm_debug_counters(unsigned long long* host16, int reset) {
  if (host16 && hipMemcpyFromSymbol(host16, HIP_SYMBOL(acfm::g_diag), sizeof(unsigned long long) * 16) != hipSuccess) return ACFM_E_LAUNCH;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(acfm::g_diag), z, sizeof(z)) != hipSuccess) return ACFM_E_LAUNCH;
  }
  return ACFM_OK;
}
#endif
#ifdef ACFM_DIAG
static unsigned long long* g_dbg = nullptr;
int acfm_debug_set_stamp_buffer(void* p) { g_dbg = (unsigned long long*)p; return 0; }
int acfm_debug_occupancy(int which, int dyn_lds) {
  int n = -1;
  hipError_t e = hipSuccess;
  if (which == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_raster_fwd<20, false, false>, RT, 0);
  else if (which == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_raster_fwd<1, true, true>, RT, 0);
  else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_sil_bwd<float>, RT, 0);
  return e == hipSuccess ? n : -1;
}
#else
static unsigned long long* const g_dbg = nullptr;
#endif
int acfm_version(void) { return 1001; }
const char* acfm_arch(void) { return "gfx950"; }

int acfm_stream_capture_id(void* stream, unsigned long long* id_host) {
  if (!id_host) return ACFM_E_BADARG;
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  if (hipStreamGetCaptureInfo((hipStream_t)stream, &status, &id) != hipSuccess) return ACFM_E_LAUNCH;
  *id_host = status == hipStreamCaptureStatusActive ? (id ? id : ~0ull) : 0ull;
  return ACFM_OK;
}

int acfm_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
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
  std::lock_guard<std::mutex> lk(g_prof_mu);
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
      "k_bds_loss", "k_bds_loss_bwd", "k_project", "k_tex_mse", "k_tex_mse_bwd", "k_deform_apply",
      "k_deform_bwd", "deform_solve", "deform_solve_bwd", "", "", "", "", ""};
  return (id >= 0 && id < ACFM_PROF_NKERNELS) ? names[id] : "";
}

size_t acfm_raster_workspace_bytes(int N, int V, int F, int H) {
  if (N <= 0 || V <= 0 || F <= 0 || H <= 0) return 0;
  return carve_ws(nullptr, N, V, F, H).bytes;
}

int acfm_project(const float* verts, const float* cams, int N, int V, float offset_z, float* proj,
                 void* stream) {
  if (!verts || !cams || !proj || N <= 0 || N > 65535 || V <= 0) return ACFM_E_BADARG;
  ProfScope ps(ACFM_PROF_PROJECT, (hipStream_t)stream);
  hipLaunchKernelGGL((k_project<false>), dim3((V + TPB - 1) / TPB, N), dim3(TPB), 0, (hipStream_t)stream, verts,
                     cams, V, offset_z, proj);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_project_backward(const float* verts, const float* cams, const float* grad_proj, int N, int V,
                          float* grad_verts, float* grad_cams, void* stream) {
  if (!verts || !cams || !grad_proj || N <= 0 || V <= 0) return ACFM_E_BADARG;
  ProfScope ps(ACFM_PROF_PROJ_BWD, (hipStream_t)stream);
  hipLaunchKernelGGL((k_project_bwd<0>), dim3(N), dim3(TPB), 0, (hipStream_t)stream, verts, cams,
                     const_cast<float*>(grad_proj), V, grad_verts, grad_cams);  // (read-only in this instantiation)
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_project_xy(const float* verts, const float* cams, int N, int V, float offset_z, float* proj_xy,
                    void* stream) {
  if (!verts || !cams || !proj_xy || N <= 0 || N > 65535 || V <= 0) return ACFM_E_BADARG;
  ProfScope ps(ACFM_PROF_PROJECT, (hipStream_t)stream);
  hipLaunchKernelGGL((k_project<true>), dim3((V + TPB - 1) / TPB, N), dim3(TPB), 0, (hipStream_t)stream, verts,
                     cams, V, offset_z, proj_xy);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_project_xy_backward(const float* verts, const float* cams, const float* grad_proj_xy, int N, int V,
                             float* grad_verts, float* grad_cams, void* stream) {
  if (!verts || !cams || !grad_proj_xy || N <= 0 || V <= 0) return ACFM_E_BADARG;
  ProfScope ps(ACFM_PROF_PROJ_BWD, (hipStream_t)stream);
  hipLaunchKernelGGL((k_project_bwd<2>), dim3(N), dim3(TPB), 0, (hipStream_t)stream, verts, cams,
                     const_cast<float*>(grad_proj_xy), V, grad_verts, grad_cams);  // (read-only in this instantiation)
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

static int sil_forward_impl(const float* verts_world, const int64_t* faces, const float* cams, int N, int V,
                            int F, int H, int K, int k_out, float blur_radius, float sigma, float offset_z,
                            void* mask, void* pix_to_face, uint64_t* kth, uint8_t* vis, void* wsp,
                            size_t ws_bytes, const AcfmRasterTuning* tuning, void* stream, bool fused,
                            const void* gt, const void* edt, int ref_batch, float* losses,
                            const AcfmSilExtras* ex = nullptr) {
  float* pf_imgs = ex ? ex->tex_imgs : nullptr;
  float* pf_sil = ex ? ex->tex_sil : nullptr;
  int64_t* pf_p2f = ex ? ex->tex_pix_to_face : nullptr;
  int32_t* pf_tidx = ex ? ex->tex_texel_idx : nullptr;
  if (!verts_world || !faces || !cams || !mask || !pix_to_face || !wsp) return ACFM_E_BADARG;
  if (fused && (!losses || ref_batch <= 0 || N % ref_batch != 0)) return ACFM_E_BADARG;
  if (bad_dims(N, V, F, H) || K < 2 || K > ACFM_MAX_K || !(sigma > 0.f) || blur_radius < 0.f ||
      (k_out != K && k_out != 1))
    return ACFM_E_BADARG;
  Tune tn;
  if (!tune_from(tuning, tn)) return ACFM_E_BADARG;
  if (tn.f16 && k_out != 1) return ACFM_E_BADARG;      // half storage goes with the int32 nearest-face plane
  const RasterWs ws = carve_ws(wsp, N, V, F, H, tn.split);
  if (ws.bytes > ws_bytes) return ACFM_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_setup(verts_world, faces, cams, N, V, F, H, offset_z, 0, blur_radius, ws, tn, st, vis,
                        ex ? ex->proj_xy : nullptr);
  if (rc) return rc;
  FwdOut out = {};
  out.dbg = g_dbg;
  out.h16 = tn.f16 ? 1 : 0;
  out.mask = mask;
  out.p2f = pix_to_face;
  out.kout = k_out;
  out.kth = reinterpret_cast<unsigned long long*>(kth);
  out.vis = vis;
  out.V = V;
  out.sig_scale = 1.44269504088896341f / sigma;
  out.lrb = 1;
  if (tn.cover) out.cover_out = ws.cover;
  if (pf_imgs || pf_sil || pf_p2f || pf_tidx) {   // all four or none; only with the cover plane (the texture render it prepares shades from it) and float storage
    if (!pf_sil || !pf_p2f || !pf_tidx || !tn.cover || tn.f16) return ACFM_E_BADARG;
    out.pf_imgs = pf_imgs; out.pf_sil = pf_sil; out.pf_p2f = pf_p2f; out.pf_tidx = pf_tidx;
  }
  if (fused) { out.lgt = gt; out.ledt = edt; out.lrb = ref_batch; out.lpart = ws.lpart; }
  switch (K) {
    case 20: rc = launch_sil_fwd<20>(ws, N, F, H, blur_radius, sigma, out, tn, st); break;
    case 10: rc = launch_sil_fwd<10>(ws, N, F, H, blur_radius, sigma, out, tn, st); break;
    case 8: rc = launch_sil_fwd<8>(ws, N, F, H, blur_radius, sigma, out, tn, st); break;
    case 4: rc = launch_sil_fwd<4>(ws, N, F, H, blur_radius, sigma, out, tn, st); break;
    case 2: rc = launch_sil_fwd<2>(ws, N, F, H, blur_radius, sigma, out, tn, st); break;
    case 32: rc = launch_sil_fwd<32>(ws, N, F, H, blur_radius, sigma, out, tn, st); break;
    default: return ACFM_E_BADARG;  // supported K: 2, 4, 8, 10, 20, 32
  }
  if (rc || !fused) return rc;
  const int tiles = (H + RBLK - 1) / RBLK;
  ProfScope ps(ACFM_PROF_MASK_LOSS, st);
  const int fc = fin_chunks(N);
  hipLaunchKernelGGL(k_sil_loss_finish1, dim3(fc, N), dim3(TPB), 0, st, ws.lpart, gt, tiles * tiles, H * H,
                     ref_batch, out.h16, ws.lpart2);
  hipLaunchKernelGGL(k_sil_loss_finish2, dim3((N + 63) / 64), dim3(64), 0, st, ws.lpart2, N, H * H, fc, losses);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_sil_forward(const float* verts_world, const int64_t* faces, const float* cams, int N, int V,
                     int F, int H, int K, int k_out, float blur_radius, float sigma, float offset_z,
                     void* mask, void* pix_to_face, uint64_t* kth, uint8_t* vis, void* wsp,
                     size_t ws_bytes, const AcfmRasterTuning* tuning, void* stream) {
  return sil_forward_impl(verts_world, faces, cams, N, V, F, H, K, k_out, blur_radius, sigma, offset_z, mask,
                          pix_to_face, kth, vis, wsp, ws_bytes, tuning, stream, false, nullptr, nullptr, 1, nullptr);
}

int acfm_sil_forward_ex(const float* verts_world, const int64_t* faces, const float* cams, int N, int V,
                        int F, int H, int K, int k_out, float blur_radius, float sigma, float offset_z,
                        void* mask, void* pix_to_face, uint64_t* kth, uint8_t* vis, void* wsp,
                        size_t ws_bytes, const AcfmRasterTuning* tuning, const AcfmSilExtras* extras, void* stream) {
  return sil_forward_impl(verts_world, faces, cams, N, V, F, H, K, k_out, blur_radius, sigma, offset_z, mask,
                          pix_to_face, kth, vis, wsp, ws_bytes, tuning, stream, false, nullptr, nullptr, 1, nullptr, extras);
}

int acfm_sil_loss_forward_ex(const float* verts_world, const int64_t* faces, const float* cams, const void* gt,
                             const void* edt, int ref_batch, int N, int V, int F, int H, int K, int k_out,
                             float blur_radius, float sigma, float offset_z, void* mask, void* pix_to_face,
                             uint64_t* kth, uint8_t* vis, float* losses, void* wsp, size_t ws_bytes,
                             const AcfmRasterTuning* tuning, const AcfmSilExtras* extras, void* stream) {
  return sil_forward_impl(verts_world, faces, cams, N, V, F, H, K, k_out, blur_radius, sigma, offset_z, mask,
                          pix_to_face, kth, vis, wsp, ws_bytes, tuning, stream, true, gt, edt, ref_batch, losses, extras);
}

int acfm_sil_loss_forward(const float* verts_world, const int64_t* faces, const float* cams, const void* gt,
                          const void* edt, int ref_batch, int N, int V, int F, int H, int K, int k_out,
                          float blur_radius, float sigma, float offset_z, void* mask, void* pix_to_face,
                          uint64_t* kth, uint8_t* vis, float* losses, void* wsp, size_t ws_bytes,
                          const AcfmRasterTuning* tuning, void* stream) {
  return sil_forward_impl(verts_world, faces, cams, N, V, F, H, K, k_out, blur_radius, sigma, offset_z, mask,
                          pix_to_face, kth, vis, wsp, ws_bytes, tuning, stream, true, gt, edt, ref_batch, losses);
}

static int sil_backward_impl(const float* verts_world, const int64_t* faces, const float* cams,
                             const void* mask, const uint64_t* kth, BwdGrad bg, int N, int V,
                             int F, int H, float blur_radius, float sigma, float offset_z, float* grad_verts,
                             float* grad_cams, void* wsp, size_t ws_bytes, int ws_from_forward,
                             const AcfmRasterTuning* tuning, void* stream, const float* gproj = nullptr) {
  if (!verts_world || !faces || !cams || !mask || !kth || !wsp) return ACFM_E_BADARG;
  if (!bg.grad_mask && (!bg.go || bg.lrb <= 0 || N % bg.lrb != 0)) return ACFM_E_BADARG;
  if (bad_dims(N, V, F, H) || !(sigma > 0.f) || blur_radius < 0.f) return ACFM_E_BADARG;
  Tune tn;
  if (!tune_from(tuning, tn)) return ACFM_E_BADARG;
  bg.h16 = tn.f16 ? 1 : 0;
  const RasterWs ws = carve_ws(wsp, N, V, F, H, tn.split);   // (same tuning as the forward whose workspace this is)
  if (ws.bytes > ws_bytes) return ACFM_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  if (!ws_from_forward) {
    int rc = launch_setup(verts_world, faces, cams, N, V, F, H, offset_z, 0, blur_radius, ws, tn, st);
    if (rc) return rc;
  }
  if (!grad_verts && !grad_cams) return ACFM_OK;   // nothing asked for
  // ws.grad_ndc is zero here: k_setup cleared it, and every k_project_bwd<1> clears it again after reading
  const size_t lds = 0;
  {
    ProfScope ps(ACFM_PROF_SIL_BWD, st);
    if (tn.deterministic)
      hipLaunchKernelGGL(k_sil_bwd<long long>, dim3(tile_grid(N, H, tn.div[2], ws.split_slots)), dim3(RT), lds, st, ws,
                         mask, reinterpret_cast<const unsigned long long*>(kth), bg, N, V, F, H, blur_radius, sigma);
    else
      hipLaunchKernelGGL(k_sil_bwd<float>, dim3(tile_grid(N, H, tn.div[2], ws.split_slots)), dim3(RT), lds, st, ws,
                         mask, reinterpret_cast<const unsigned long long*>(kth), bg, N, V, F, H, blur_radius, sigma);
  }
  ACFM_CHECK_LAUNCH();
  if (grad_verts || grad_cams) {
    ProfScope ps(ACFM_PROF_PROJ_BWD, st);
    if (tn.deterministic)
      hipLaunchKernelGGL((k_project_bwd<3>), dim3(N), dim3(TPB), 0, st, verts_world, cams,
                         reinterpret_cast<float*>(ws.grad_fix), V, grad_verts, grad_cams, gproj);
    else
      hipLaunchKernelGGL((k_project_bwd<1>), dim3(N), dim3(TPB), 0, st, verts_world, cams,
                         ws.grad_ndc, V, grad_verts, grad_cams, gproj);
    ACFM_CHECK_LAUNCH();
  }
  return ACFM_OK;
}

int acfm_sil_backward(const float* verts_world, const int64_t* faces, const float* cams,
                      const void* mask, const uint64_t* kth, const float* grad_mask, int N, int V,
                      int F, int H, float blur_radius, float sigma, float offset_z, float* grad_verts,
                      float* grad_cams, void* wsp, size_t ws_bytes, int ws_from_forward,
                      const AcfmRasterTuning* tuning, void* stream) {
  if (!grad_mask) return ACFM_E_BADARG;
  BwdGrad bg = {};
  bg.grad_mask = grad_mask;
  bg.lrb = 1;
  return sil_backward_impl(verts_world, faces, cams, mask, kth, bg, N, V, F, H, blur_radius, sigma, offset_z,
                           grad_verts, grad_cams, wsp, ws_bytes, ws_from_forward, tuning, stream);
}

int acfm_sil_backward_ex(const float* verts_world, const int64_t* faces, const float* cams,
                         const void* mask, const uint64_t* kth, const float* grad_mask, int N, int V,
                         int F, int H, float blur_radius, float sigma, float offset_z, float* grad_verts,
                         float* grad_cams, void* wsp, size_t ws_bytes, int ws_from_forward,
                         const AcfmRasterTuning* tuning, const AcfmSilExtras* extras, void* stream) {
  if (!grad_mask) return ACFM_E_BADARG;
  BwdGrad bg = {};
  bg.grad_mask = grad_mask;
  bg.lrb = 1;
  return sil_backward_impl(verts_world, faces, cams, mask, kth, bg, N, V, F, H, blur_radius, sigma, offset_z,
                           grad_verts, grad_cams, wsp, ws_bytes, ws_from_forward, tuning, stream,
                           extras ? extras->grad_proj_xy : nullptr);
}

int acfm_sil_loss_backward_ex(const float* verts_world, const int64_t* faces, const float* cams, const void* mask,
                              const uint64_t* kth, const void* gt, const void* edt, int ref_batch,
                              const float* grad_losses, int N, int V, int F, int H, float blur_radius, float sigma,
                              float offset_z, float* grad_verts, float* grad_cams, void* wsp, size_t ws_bytes,
                              int ws_from_forward, const AcfmRasterTuning* tuning, const AcfmSilExtras* extras,
                              void* stream) {
  if (!grad_losses) return ACFM_E_BADARG;
  BwdGrad bg = {};
  bg.lgt = gt; bg.ledt = edt; bg.go = grad_losses; bg.lrb = ref_batch;
  return sil_backward_impl(verts_world, faces, cams, mask, kth, bg, N, V, F, H, blur_radius, sigma, offset_z,
                           grad_verts, grad_cams, wsp, ws_bytes, ws_from_forward, tuning, stream,
                           extras ? extras->grad_proj_xy : nullptr);
}

int acfm_sil_loss_backward(const float* verts_world, const int64_t* faces, const float* cams, const void* mask,
                           const uint64_t* kth, const void* gt, const void* edt, int ref_batch,
                           const float* grad_losses, int N, int V, int F, int H, float blur_radius, float sigma,
                           float offset_z, float* grad_verts, float* grad_cams, void* wsp, size_t ws_bytes,
                           int ws_from_forward, const AcfmRasterTuning* tuning, void* stream) {
  if (!grad_losses) return ACFM_E_BADARG;
  BwdGrad bg = {};
  bg.lgt = gt; bg.ledt = edt; bg.go = grad_losses; bg.lrb = ref_batch;
  return sil_backward_impl(verts_world, faces, cams, mask, kth, bg, N, V, F, H, blur_radius, sigma, offset_z,
                           grad_verts, grad_cams, wsp, ws_bytes, ws_from_forward, tuning, stream);
}

int acfm_hard_raster(const float* verts_proj, const int64_t* faces, int N, int V, int F, int H,
                     int64_t* pix_to_face, uint8_t* vis, void* wsp, size_t ws_bytes,
                     const AcfmRasterTuning* tuning, void* stream) {
  if (!verts_proj || !faces || !pix_to_face || !wsp || bad_dims(N, V, F, H)) return ACFM_E_BADARG;
  Tune tn;
  if (!tune_from(tuning, tn) || tn.f16) return ACFM_E_BADARG;
  const RasterWs ws = carve_ws(wsp, N, V, F, H, tn.split);
  if (ws.bytes > ws_bytes) return ACFM_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_setup(verts_proj, faces, nullptr, N, V, F, H, 0.f, 1, 0.f, ws, tn, st, vis);
  if (rc) return rc;
  FwdOut out = {};
  out.dbg = g_dbg;
  out.p2f = pix_to_face;
  out.vis = vis;
  out.V = V;
  ProfScope ps(ACFM_PROF_HARD_FWD, st);
  hipLaunchKernelGGL((k_raster_fwd<1, false, false>), dim3(tile_grid(N, H, tn.div[1])), dim3(RT), 0, st, ws, N, F,
                     H, 0.f, 1e-4f, out);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

static int tex_forward_impl(const float* verts_world, const int64_t* faces, const float* cams,
                            const void* atlas, int N, int V, int F, int H, int R, float sigma, float gamma,
                            float offset_z, void* imgs, void* sil, void* pix_to_face, int32_t* texel_idx,
                            void* wsp, size_t ws_bytes, int ws_ready, float ws_blur, int atlas_batch,
                            const AcfmRasterTuning* tuning, void* stream, const void* ref_img,
                            const void* ref_mask, int ref_batch, float* loss) {
  if (!verts_world || !faces || !cams || !atlas || !imgs || !sil || !pix_to_face || !texel_idx || !wsp)
    return ACFM_E_BADARG;
  if (bad_dims(N, V, F, H) || R <= 0 || R > 256 || !(sigma > 0.f) || !(gamma > 0.f)) return ACFM_E_BADARG;
  if (atlas_batch <= 0 || N % atlas_batch != 0) return ACFM_E_BADARG;
  if ((size_t)atlas_batch * F * R * R > 0x7fffffffull) return ACFM_E_BADARG;  // texel_idx is int32
  Tune tn;
  if (!tune_from(tuning, tn)) return ACFM_E_BADARG;
  const RasterWs ws = carve_ws(wsp, N, V, F, H, tn.split);
  if (ws.bytes > ws_bytes) return ACFM_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  if (ws_ready && !(ws_blur >= 0.f)) return ACFM_E_BADARG;
  if (!ws_ready) {
    int rc = launch_setup(verts_world, faces, cams, N, V, F, H, offset_z, 0, 0.f, ws, tn, st);
    if (rc) return rc;
  }
  FwdOut out = {};
  out.dbg = g_dbg;
  out.p2f = pix_to_face;
  out.atlas = atlas; out.imgs = imgs; out.sil = sil; out.tidx = texel_idx; out.R = R; out.gamma = gamma;
  out.atlas_n = atlas_batch;
  out.h16 = tn.f16 ? 1 : 0;
  out.box_shrink = ws_ready ? sqrtf(ws_blur) * (1.0f - 1e-5f) : 0.f;
  out.lrb = 1;
  if (ws_ready < 0 || ws_ready > 3) return ACFM_E_BADARG;
  if (ws_ready >= 2) {
    if (!tn.cover) return ACFM_E_BADARG;   // the tuning of the render that filled the workspace says whether the plane is there
    out.cover_in = ws.cover;
    if (ws_ready == 3) {                   // ... and that render (acfm_sil_forward_prefill) stored the empty blocks' constants
      if (tn.f16) return ACFM_E_BADARG;
      out.prefilled = 1;
    }
  }
  if (loss) {
    if (!ref_img || !ref_mask || ref_batch <= 0 || N % ref_batch != 0) return ACFM_E_BADARG;
    out.timg = ref_img; out.tmask = ref_mask; out.lrb = ref_batch; out.lpart = ws.lpart;
  }
  {
    ProfScope ps(ACFM_PROF_TEX_FWD, st);
    if (out.cover_in) {
      // one wave per div entries of the order (measured at 64 frames @256^2, entries per wave 0.5 / 1 / 2 / 4 / 8:
      // 47 / 35 / 31 / 43 / 45 us: fewer waves leave the stores of the empty blocks to too few issuers)
      const size_t G = (N & 7) == 0 ? 8 : 1;
      const size_t per = (size_t)((H + RBLK - 1) / RBLK) * ((H + RBLK - 1) / RBLK) * (N / G);
      const size_t d = (size_t)(tn.div[1] < 1 ? 1 : tn.div[1]) * COVER_WPB;
      hipLaunchKernelGGL((k_tex_cover<true>), dim3((unsigned)(G * ((per + d - 1) / d))), dim3(64 * COVER_WPB), 0, st, ws,
                         N, F, H, sigma, out);
    } else {
      hipLaunchKernelGGL((k_raster_fwd<1, true, true>), dim3(tile_grid(N, H, tn.div[1])), dim3(RT), 0, st, ws, N, F, H,
                         0.f, sigma, out);
    }
    ACFM_CHECK_LAUNCH();
  }
  if (loss) {
    const int tiles = (H + RBLK - 1) / RBLK;
    ProfScope ps(ACFM_PROF_TEX_MSE, st);
    const int fc = fin_chunks(N);
    hipLaunchKernelGGL(k_tex_loss_finish1, dim3(fc, N), dim3(TPB), 0, st, ws.lpart, ref_img, ref_mask,
                       tiles * tiles, H * H, ref_batch, out.h16, ws.lpart2);
    hipLaunchKernelGGL(k_tex_loss_finish2, dim3((N + 63) / 64), dim3(64), 0, st, ws.lpart2, N, H * H, fc, loss);
    ACFM_CHECK_LAUNCH();
  }
  return ACFM_OK;
}

int acfm_tex_forward(const float* verts_world, const int64_t* faces, const float* cams,
                     const void* atlas, int N, int V, int F, int H, int R, float sigma, float gamma,
                     float offset_z, void* imgs, void* sil, void* pix_to_face, int32_t* texel_idx,
                     void* wsp, size_t ws_bytes, int ws_ready, float ws_blur, int atlas_batch,
                     const AcfmRasterTuning* tuning, void* stream) {
  return tex_forward_impl(verts_world, faces, cams, atlas, N, V, F, H, R, sigma, gamma, offset_z, imgs, sil,
                          pix_to_face, texel_idx, wsp, ws_bytes, ws_ready, ws_blur, atlas_batch, tuning, stream,
                          nullptr, nullptr, 1, nullptr);
}

int acfm_tex_mse_forward(const float* verts_world, const int64_t* faces, const float* cams, const void* atlas,
                         const void* ref_img, const void* ref_mask, int ref_batch, int N, int V, int F, int H, int R,
                         float sigma, float gamma, float offset_z, void* imgs, void* sil, void* pix_to_face,
                         int32_t* texel_idx, float* loss, void* wsp, size_t ws_bytes, int ws_ready, float ws_blur,
                         int atlas_batch, const AcfmRasterTuning* tuning, void* stream) {
  if (!loss) return ACFM_E_BADARG;
  return tex_forward_impl(verts_world, faces, cams, atlas, N, V, F, H, R, sigma, gamma, offset_z, imgs, sil,
                          pix_to_face, texel_idx, wsp, ws_bytes, ws_ready, ws_blur, atlas_batch, tuning, stream,
                          ref_img, ref_mask, ref_batch, loss);
}

int acfm_vertex_color_forward(const float* verts_world, const int64_t* faces, const float* cams,
                              const float* verts_rgb, int N, int V, int F, int H, float sigma, float gamma,
                              float offset_z, float* imgs, float* sil, int64_t* pix_to_face, void* wsp,
                              size_t ws_bytes, int ws_ready, float ws_blur, const AcfmRasterTuning* tuning,
                              void* stream) {
  if (!verts_world || !faces || !cams || !verts_rgb || !imgs || !sil || !pix_to_face || !wsp) return ACFM_E_BADARG;
  if (bad_dims(N, V, F, H) || !(sigma > 0.f) || !(gamma > 0.f)) return ACFM_E_BADARG;
  Tune tn;
  if (!tune_from(tuning, tn) || tn.f16) return ACFM_E_BADARG;
  const RasterWs ws = carve_ws(wsp, N, V, F, H, tn.split);
  if (ws.bytes + sizeof(int32_t) * (size_t)N * H * H > ws_bytes) return ACFM_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  if (ws_ready && !(ws_blur >= 0.f)) return ACFM_E_BADARG;
  if (!ws_ready) {
    int rc = launch_setup(verts_world, faces, cams, N, V, F, H, offset_z, 0, 0.f, ws, tn, st);
    if (rc) return rc;
  }
  FwdOut out = {};
  out.dbg = g_dbg;
  out.p2f = pix_to_face;
  out.vrgb = verts_rgb; out.V = V; out.atlas_n = N;
  out.box_shrink = ws_ready ? sqrtf(ws_blur) * (1.0f - 1e-5f) : 0.f;
  out.imgs = imgs; out.sil = sil; out.tidx = (int32_t*)((char*)wsp + ws.bytes); out.R = 1; out.gamma = gamma;
  out.atlas = verts_rgb;  // never dereferenced when vrgb is set
  ProfScope ps(ACFM_PROF_TEX_FWD, st);
  hipLaunchKernelGGL((k_raster_fwd<1, true, true>), dim3(tile_grid(N, H, tn.div[1])), dim3(RT), 0, st, ws, N, F, H,
                     0.f, sigma, out);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_tex_backward(const float* grad_imgs, const int32_t* texel_idx, int N, int F, int H, int R,
                      int atlas_batch, float* grad_atlas, void* stream) {
  if (!grad_imgs || !texel_idx || !grad_atlas || N <= 0 || F <= 0 || H <= 0 || R <= 0 || atlas_batch <= 0 ||
      N % atlas_batch != 0)
    return ACFM_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const size_t total = (size_t)N * H * H;
  if (zero_async(grad_atlas, sizeof(float) * 3 * (size_t)atlas_batch * F * R * R, st) != ACFM_OK)
    return ACFM_E_LAUNCH;
  ProfScope ps(ACFM_PROF_TEX_BWD, st);
  hipLaunchKernelGGL(k_tex_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, grad_imgs,
                     texel_idx, (size_t)H * H, total, grad_atlas);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

static int tex_backward_faces_impl(const TexGrad& tgrad, const int32_t* texel_idx, const void* wsp, size_t ws_bytes,
                                   float ws_blur, int N, int V, int F, int H, int R, int atlas_batch,
                                   float* grad_atlas, void* stream) {
  if (!texel_idx || !grad_atlas || !wsp) return ACFM_E_BADARG;
  if (bad_dims(N, V, F, H) || R <= 0 || R > TEXG_MAX_R || atlas_batch <= 0 || N % atlas_batch != 0 || !(ws_blur >= 0.f))
    return ACFM_E_BADARG;
  if ((size_t)atlas_batch * F * R * R > 0x7fffffffull) return ACFM_E_BADARG;
  const RasterWs ws = carve_ws(const_cast<void*>(wsp), N, V, F, H);
  if (ws.bytes > ws_bytes) return ACFM_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const size_t waves = (size_t)atlas_batch * ((F + TEXG_FPW - 1) / TEXG_FPW);
  ProfScope ps(ACFM_PROF_TEX_BWD, st);
  hipLaunchKernelGGL(k_tex_bwd_faces, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, ws, tgrad, texel_idx,
                     N, F, H, R, atlas_batch, ws_blur > 0.f ? sqrtf(ws_blur) * (1.0f - 1e-5f) : 0.f, grad_atlas);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_tex_backward_faces(const float* grad_imgs, const int32_t* texel_idx, const void* wsp, size_t ws_bytes,
                            float ws_blur, int N, int V, int F, int H, int R, int atlas_batch, float* grad_atlas,
                            void* stream) {
  if (!grad_imgs) return ACFM_E_BADARG;
  TexGrad tg = {};
  tg.grad_imgs = grad_imgs;
  tg.rb = 1;
  return tex_backward_faces_impl(tg, texel_idx, wsp, ws_bytes, ws_blur, N, V, F, H, R, atlas_batch, grad_atlas, stream);
}

int acfm_tex_mse_backward_faces(const void* imgs, const void* ref_img, const void* ref_mask, int ref_batch,
                                const float* grad_loss, const int32_t* texel_idx, const void* wsp, size_t ws_bytes,
                                float ws_blur, int N, int V, int F, int H, int R, int atlas_batch, float* grad_atlas,
                                const AcfmRasterTuning* tuning, void* stream) {
  if (!imgs || !ref_img || !ref_mask || !grad_loss || ref_batch <= 0 || N <= 0 || N % ref_batch != 0) return ACFM_E_BADARG;
  Tune tn;
  if (!tune_from(tuning, tn)) return ACFM_E_BADARG;
  TexGrad tg = {};
  tg.h16 = tn.f16 ? 1 : 0;
  tg.imgs = imgs; tg.timg = ref_img; tg.tmask = ref_mask; tg.go = grad_loss; tg.rb = ref_batch;
  return tex_backward_faces_impl(tg, texel_idx, wsp, ws_bytes, ws_blur, N, V, F, H, R, atlas_batch, grad_atlas, stream);
}

}  // extern "C"
