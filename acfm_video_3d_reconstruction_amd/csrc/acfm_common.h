// Shared device helpers for the gfx950 kernels of libacfm_hip.so.
//
// Arithmetic contract (DESIGN.md "Numerics"): fp32, every multiply and add rounds
// separately (-ffp-contract=off + the pragma below), IEEE division (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt).  That is what makes face indices bit-identical
// to the CPU oracle and to the reference's chain of separate torch kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/acfm_hip.h"

#pragma clang fp contract(off)

#ifndef ACFM_EDGE_CONST
#define ACFM_EDGE_CONST 1   // see acfm_raster.hip
#endif
#ifndef ACFM_BWD_EDGE_GLOBAL
#define ACFM_BWD_EDGE_GLOBAL 1   // see acfm_raster.hip (sil_bwd_block): measured 180.8 -> 171.4 us per 64-frame launch
#endif
#define ACFM_REC_EDGES (ACFM_EDGE_CONST || ACFM_BWD_EDGE_GLOBAL)

#define ACFM_K_EPS 1e-8f   // PyTorch3D kEpsilon
#define ACFM_EYE_Z 2.732f  // nmr.py:144: eye=(0,0,-2.732) -> T=(0,0,2.732)
#define ACFM_WAVE 64

#define ACFM_CHECK_LAUNCH()                        \
  do {                                             \
    if (hipGetLastError() != hipSuccess) return ACFM_E_LAUNCH; \
  } while (0)

namespace acfm {

// stream-ordered zero fill by a kernel (the library issues no hipMemsetAsync: see acfm_raster.hip)
int zero_async(void* p, size_t nbytes, hipStream_t st);

// per-kernel hipEvent bracketing (acfm_prof_* in include/acfm_hip.h); state lives in acfm_raster.hip
void prof_begin(int id, hipStream_t st);
void prof_end(hipStream_t st);
struct ProfScope {
  hipStream_t st;
  ProfScope(int id, hipStream_t s) : st(s) { prof_begin(id, s); }
  ~ProfScope() { prof_end(st); }
};

#ifndef ACFM_CTILE
#define ACFM_CTILE 16
#endif
constexpr int CTILE = ACFM_CTILE;  // coarse tile side (pixels): k_setup leaves one face bitmask per coarse tile

// Per-face record of the raster workspace (k_setup writes it, the binning of the raster kernels reads it).
// The first cache line is all the backward and nearest-face walks need; ACFM_EDGE_CONST adds what is constant per
// FACE in the exact per-pixel distance test of the K-nearest forward -- the squared lengths of the three edges and
// their refined reciprocals (operands of the IEEE-exact division of point_line_dist) -- computed once per face in
// k_setup with the very operations the per-pixel code used to repeat for every pixel, so every per-pixel value is
// bit-identical to the unfactored evaluation (and to the oracle).
struct __attribute__((aligned(64))) FaceRec {
  float4 box;   // (xmin,xmax,ymin,ymax), blur margin included; degenerate face = (inf,-inf,inf,-inf)
  float4 a;     // (x0,y0,x1,x2)   -- (x1,x2), (y1,y2) as register pairs for the packed fp32 pipe
  float4 b;     // (y1,y2,z0,z1)
  float4 c;     // (z2, area, denom = area + kEps, rden = refined 1/denom)
#if ACFM_REC_EDGES
  float4 e0;    // (|e01|^2, |e02|^2, 1/|e01|^2, 1/|e02|^2): edges v0->v1 and v0->v2, the pair the packed pipe evaluates
  float4 e1;    // (|e12|^2, 1/|e12|^2, flag = 1 if any |e|^2 <= kEps (that face takes the unfactored path), -)
  float4 pad_[2];
#endif
};

// Workspace carve-up shared by every raster entry point (acfm_raster_workspace_bytes).
struct RasterWs {
  float* ndc;      // [N,V,3] NDC x, NDC y, view z
  FaceRec* rec;    // [N,F] one 64-byte record per face: the box test and the copy of a passing face touch ONE cache line
  int4* vidx;      // [N,F] (i0,i1,i2,-)
  float4* mbox;    // [N,4] union of the face boxes of each of the 4 face slices of k_setup
  float* grad_ndc; // [N,V,2]
  long long* grad_fix; // [N,V,2] the same in 2^-36 fixed point (deterministic backward)
  int* tile_cnt;   // [N,blocks^2] faces whose box meets the 8x8 block (cost estimate for scheduling)
  int* tile_part;  // [slices,N,blocks^2] the same per face slice of k_setup (counters in LDS): k_order adds them into tile_cnt
  int slices;      // face slices (workgroups) per mesh in k_setup: 4, 8 or 16
  int* order;      // [N*blocks^2] heavy-first visiting order of (mesh, block) per XCD group
  uint8_t* fvis;   // [N,F] 1 = the face is the nearest one at some pixel of the last texture render on this workspace
  int* n_work;     // [8] per XCD group: entries of its order that have work (the flagged-empty ones follow them)
  int split_slots; // heaviest blocks per XCD group that may be rendered by four workgroups each (raster kernels)
  unsigned* cmask; // [N,ctiles^2,2*words] face bitmask of every CTILE x CTILE-pixel coarse tile (words = ceil(F/64) u64)
  float4* lpart;   // [N,blocks^2,4] fused render+loss: per 8x8 block (x 4 split roles) partial sums of the silhouette
                   // loss terms, written by the raster kernel, summed in fixed order by k_sil_loss_finish
  float* lpart2;   // [N,64,5] second-stage partial sums of the same
  int* cover;      // [N,H,H] nearest face that COVERS the pixel (the hard K = 1 render's answer: clipped-barycentric depth,
                   // pixel strictly inside), local face id or -1; written by the K-nearest forward on request (Tune::cover)
                   // for the blocks that have work, read by the texture forward that takes the workspace over
  size_t bytes;
};

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }


// Per-call tuning of the raster launches (AcfmRasterTuning of the C ABI; NULL = these defaults).  There is
// no process-global knob: every entry point takes its own copy.
struct Tune {
  int split = -5;            // split heuristic: < 0 automatic (k_order: a block is split when its cost > (-split / 4) x the mean work per wave slot), 0 never, 1 always
  int div[3] = {4, 2, 4};    // workgroups per XCD group = entries / div: [0] K-nearest forward, [1] nearest-face forward, [2] backward
                             // (measured at 64 frames @256^2: K-nearest forward 323 (div 1) / 292 (2) / 287 (4) us)
  bool deterministic = false; // flags bit 0: fixed-point accumulation in the silhouette backward
  bool f16 = false;           // flags bit 1: half storage of masks / images / atlases, int32 nearest-face plane
  bool cover = false;         // flags bit 2: the K-nearest forward also records the nearest COVERING face per pixel (ws.cover)
};
static inline bool tune_from(const AcfmRasterTuning* t, Tune& out) {
  if (!t) return true;
  if (t->split_mode < -64 || t->split_mode > 1) return false;
  out.split = t->split_mode;
  for (int i = 0; i < 3; ++i) {
    if (t->grid_div[i] < 0 || t->grid_div[i] > 64) return false;
    if (t->grid_div[i] > 0) out.div[i] = t->grid_div[i];   // 0 = keep the default
  }
  if (t->flags & ~7) return false;
  out.deterministic = (t->flags & 1) != 0;
  out.f16 = (t->flags & 2) != 0;
  out.cover = (t->flags & 4) != 0;
  return true;
}

static inline RasterWs carve_ws(void* base, int N, int V, int F, int H, int g_split_mode = -5) {
  RasterWs w;
  {
    // Splitting costs ~25 % more work per split block (four waves bin and merge) and shortens it ~3x.  It pays for the
    // blocks that would outlast the launch (k_order decides which, from the cost histogram): whole small launches --
    // one 8x8 block of a dense mesh region runs ~250 us, the work of a whole frame is ~4 us of the chip --, and the few
    // dozen heaviest blocks of a large one (64 frames @256^2: the blocks of >= 240 face boxes span the whole 208-us
    // launch).  Split slots are workgroups that exist whether used or not (4 per slot, an immediate exit if unused):
    // up to 1024 per XCD group for small launches, 4 per mesh of the group for large ones.
    const int per_group = (N & 7) == 0 ? N / 8 : N;      // meshes per XCD group
    const size_t blocks = (size_t)N * ((H + 7) / 8) * ((H + 7) / 8);
    if (g_split_mode == 0) w.split_slots = 0;
    else if (g_split_mode > 0 || blocks <= 40960) w.split_slots = per_group * 32 < 1024 ? per_group * 32 : 1024;
    else w.split_slots = per_group * 4 < 256 ? per_group * 4 : 256;
  }
  // face slices per mesh of k_setup (one workgroup each): enough workgroups to cover the chip with few meshes -- 4 at
  // >= 64 meshes, 8 at >= 32, else 16 -- but at least 64 faces per slice
  w.slices = N >= 64 ? 4 : N >= 32 ? 8 : 16;
  while (w.slices > 4 && (F + w.slices - 1) / w.slices < 64) w.slices >>= 1;
  char* p = (char*)base;
  size_t o = 0;
  w.ndc = (float*)(p + o);      o += align256(sizeof(float) * 3 * (size_t)N * V);
  w.rec = (FaceRec*)(p + o);    o += align256(sizeof(FaceRec) * (size_t)N * F);
  w.vidx = (int4*)(p + o);      o += align256(sizeof(int4) * (size_t)N * F);
  w.mbox = (float4*)(p + o);    o += align256(sizeof(float4) * (size_t)w.slices * (size_t)N);
  w.grad_ndc = (float*)(p + o); o += align256(sizeof(float) * 2 * (size_t)N * V);
  w.grad_fix = (long long*)(p + o); o += align256(sizeof(long long) * 2 * (size_t)N * V);
  const size_t tt = (size_t)((H + 7) / 8) * ((H + 7) / 8);  // 8x8-pixel blocks (RBLK)
  w.tile_cnt = (int*)(p + o);   o += align256(sizeof(int) * (size_t)N * tt);
  w.tile_part = (int*)(p + o);  o += align256(sizeof(int) * (size_t)w.slices * (size_t)N * tt);
  w.order = (int*)(p + o);      o += align256(sizeof(int) * (size_t)N * tt);
  w.n_work = (int*)(p + o);     o += align256(sizeof(int) * 8);
  w.fvis = (uint8_t*)(p + o);   o += align256((size_t)N * F);
  const size_t ct = (size_t)((H + CTILE - 1) / CTILE) * ((H + CTILE - 1) / CTILE), words = ((size_t)F + 63) / 64;
  w.cmask = (unsigned*)(p + o); o += align256(sizeof(unsigned) * 2 * (size_t)N * ct * words);
  w.lpart = (float4*)(p + o);   o += align256(sizeof(float4) * 4 * (size_t)N * tt);
  w.lpart2 = (float*)(p + o);   o += align256(sizeof(float) * 5 * 64 * (size_t)N);
  w.cover = (int*)(p + o);      o += align256(sizeof(int) * (size_t)N * H * H);
  w.bytes = o;
  return w;
}

__device__ __forceinline__ float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
  return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// x / y for normal-range y: refined reciprocal, quotient, two residual corrections -- the
// correctly rounded result of the IEEE division sequence without its scaling / fix-up steps (8
// instead of ~11 instructions; callers with several numerators over one denominator share r).
// Used where the operands cannot be denormal or overflow (areas, squared edge lengths > 1e-8 and
// coordinates of order 1); checked bit for bit against the oracle's C divisions by the parity tests.
__device__ __forceinline__ float recip_refined(float y) {
  const float r0 = __builtin_amdgcn_rcpf(y);
  return __builtin_fmaf(__builtin_fmaf(-y, r0, 1.0f), r0, r0);
}
__device__ __forceinline__ float div_by(float x, float y, float r) {
  float q = x * r;
  q = __builtin_fmaf(__builtin_fmaf(-y, q, x), r, q);
  q = __builtin_fmaf(__builtin_fmaf(-y, q, x), r, q);
  return q;
}

// PointLineDistanceForward: squared distance from p to segment ab.
// tc (optional): the clamped segment parameter, 1 for a degenerate segment -- what PointLineDistanceBackward
// recomputes with the same expression (the silhouette backward takes it from here instead).
__device__ __forceinline__ float point_line_dist(float px, float py, float ax, float ay, float bx,
                                                 float by, float* tc = nullptr) {
  const float bax = bx - ax, bay = by - ay;
  const float l2 = bax * bax + bay * bay;
  if (l2 <= ACFM_K_EPS) {
    const float dx = px - bx, dy = py - by;
    if (tc) *tc = 1.0f;
    return dx * dx + dy * dy;
  }
  float t = div_by(bax * (px - ax) + bay * (py - ay), l2, recip_refined(l2));
  t = fminf(fmaxf(t, 0.0f), 1.0f);
  if (tc) *tc = t;
  const float qx = ax + t * bax, qy = ay + t * bay;
  const float dx = qx - px, dy = qy - py;
  return dx * dx + dy * dy;
}

// Two point_line_dist at once on the packed fp32 pipe (v_pk_mul / v_pk_add / v_pk_fma_f32 do two
// lanes' worth of one IEEE operation per issue slot): the same operations in the same order per
// component, so each half is bit-identical to point_line_dist.  Segments (ax, ay)-(bx, by), one per component.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f point_line_dist2(float px, float py, v2f ax, v2f ay, v2f bx, v2f by, v2f* tc = nullptr) {
  const v2f bax = bx - ax, bay = by - ay;
  const v2f l2 = bax * bax + bay * bay;
  const v2f num = bax * (px - ax) + bay * (py - ay);
  v2f r0;
  r0.x = __builtin_amdgcn_rcpf(l2.x); r0.y = __builtin_amdgcn_rcpf(l2.y);
  const v2f one = {1.0f, 1.0f};
  const v2f r = fma2(fma2(-l2, r0, one), r0, r0);
  v2f t = num * r;
  t = fma2(fma2(-l2, t, num), r, t);
  t = fma2(fma2(-l2, t, num), r, t);
  t.x = fminf(fmaxf(t.x, 0.0f), 1.0f); t.y = fminf(fmaxf(t.y, 0.0f), 1.0f);
  const v2f qx = ax + t * bax, qy = ay + t * bay;
  const v2f dx = qx - px, dy = qy - py;
  v2f d = dx * dx + dy * dy;
  const v2f ex = px - bx, ey = py - by;       // degenerate segment (l2 <= kEps): distance to b
  const v2f de = ex * ex + ey * ey;
  d.x = l2.x <= ACFM_K_EPS ? de.x : d.x;
  d.y = l2.y <= ACFM_K_EPS ? de.y : d.y;
  if (tc) { tc->x = l2.x <= ACFM_K_EPS ? 1.0f : t.x; tc->y = l2.y <= ACFM_K_EPS ? 1.0f : t.y; }
  return d;
}

// PixToNdc
__device__ __forceinline__ float pix_to_ndc(int i, int S) {
  return -1.0f + (2.0f * (float)i + 1.0f) / (float)S;
}

// geom_utils.orthographic_proj_withz (geom_utils.py:62-79) for one point; c = cams row.
__device__ __forceinline__ void project_point(const float* __restrict__ c, float x, float y, float z,
                                              float offset_z, float& ox, float& oy, float& oz) {
  const float q0 = c[3], q1 = c[4], q2 = c[5], q3 = c[6];
  const float c1 = -1.0f * q1, c2 = -1.0f * q2, c3 = -1.0f * q3;
  const float X0 = x * 0.0f;
  const float t0 = X0 * q0 - x * c1 - y * c2 - z * c3;
  const float t1 = X0 * c1 + x * q0 + y * c3 - z * c2;
  const float t2 = X0 * c2 - x * c3 + y * q0 + z * c1;
  const float t3 = X0 * c3 + x * c2 - y * c1 + z * q0;
  const float r1 = q0 * t1 + q1 * t0 + q2 * t3 - q3 * t2;
  const float r2 = q0 * t2 - q1 * t3 + q2 * t0 + q3 * t1;
  const float r3 = q0 * t3 + q1 * t2 - q2 * t1 + q3 * t0;
  ox = c[0] * r1 + c[1];
  oy = c[0] * r2 + c[2];
  oz = c[0] * r3 + offset_z;
}

__device__ __forceinline__ float sigmoid_neg(float sd, float sigma) {
  // sigmoid(-sd / sigma) = 1 / (1 + exp(sd / sigma))
  return 1.0f / (1.0f + expf(sd / sigma));
}

// wave64 sum via DPP-free shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace acfm
