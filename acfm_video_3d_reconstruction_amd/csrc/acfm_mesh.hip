// Mesh priors of the hot path as fused gfx950 kernels:
//   acfm_cot_laplacian            <- geom_utils.mesh_laplacian(mesh, 'cot')   (geom_utils.py:158-324)
//   acfm_laplacian_smoothing{,_backward} <- pytorch3d.loss.mesh_laplacian_smoothing(meshes, 'cot' | 'uniform')
//                                     (called at multiframe/main.py:699-704; semantics SURVEY App-A.7)
//   acfm_edge_rigidity{,_backward} <- loss_utils.locally_rigid_fn            (loss_utils.py:150-164)
// All take PACKED meshes (verts [P,3], faces / edges with packed vertex ids), i.e. any mix of
// topologies.  The torch-op formulations launch 10-20 small kernels each (index_add, norm, ...);
// here each is one or two passes with float atomics (results differ from a sequential sum in the
// last bits only).
#include "acfm_common.h"

namespace acfm {

constexpr int MTPB = 256;

// cotangent weights / 4 of one face exactly as geom_utils.laplacian_cot (:277-298): side lengths,
// Heron's area clamped at 1e-12 before the sqrt, cot = (b^2 + c^2 - a^2) / area / 4.
struct FaceCot { float cota, cotb, cotc; };
__device__ __forceinline__ FaceCot face_cot(const float* __restrict__ v, long i0, long i1, long i2) {
  const float ax = v[3 * i0], ay = v[3 * i0 + 1], az = v[3 * i0 + 2];
  const float bx = v[3 * i1], by = v[3 * i1 + 1], bz = v[3 * i1 + 2];
  const float cx = v[3 * i2], cy = v[3 * i2 + 1], cz = v[3 * i2 + 2];
  const float A = sqrtf((bx - cx) * (bx - cx) + (by - cy) * (by - cy) + (bz - cz) * (bz - cz));  // |v1 - v2|
  const float B = sqrtf((ax - cx) * (ax - cx) + (ay - cy) * (ay - cy) + (az - cz) * (az - cz));  // |v0 - v2|
  const float C = sqrtf((ax - bx) * (ax - bx) + (ay - by) * (ay - by) + (az - bz) * (az - bz));  // |v0 - v1|
  const float s = 0.5f * (A + B + C);
  const float area = sqrtf(fmaxf(s * (s - A) * (s - B) * (s - C), 1e-12f));
  const float A2 = A * A, B2 = B * B, C2 = C * C;
  FaceCot r;
  r.cota = (B2 + C2 - A2) / area / 4.0f;   // weight of edge (v1, v2)
  r.cotb = (A2 + C2 - B2) / area / 4.0f;   // weight of edge (v2, v0)
  r.cotc = (A2 + B2 - C2) / area / 4.0f;   // weight of edge (v0, v1)
  return r;
}

// ---- a9: dense L = W - diag(rowsum W) of ONE mesh, L pre-zeroed -----------------------------
// Two passes, bit-reproducible: (1) W_ij on the off-diagonal -- an interior edge receives exactly two
// contributions and a + b = b + a, so the float atomics leave no trace of their order (a non-manifold edge with
// three or more faces would); (2) the diagonal as -rowsum(W), summed in a fixed order per row, which is also how
// the reference forms it (geom_utils.py:249-252: L = W - diag(W.sum(1))).  With the diagonal accumulated face by
// face (up to a dozen atomics per vertex in arrival order) L, and with it P and every deformed vertex, changed
// in the last bit from run to run -- enough to flip a K-truncation at a pixel now and then.
__global__ void k_cot_laplacian(const float* __restrict__ verts, const int64_t* __restrict__ faces, int V,
                                int F, float* __restrict__ L) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const long i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
  if (i0 < 0 || i1 < 0 || i2 < 0 || i0 >= V || i1 >= V || i2 >= V) return;
  const FaceCot c = face_cot(verts, i0, i1, i2);
  const long e[3][2] = {{i1, i2}, {i2, i0}, {i0, i1}};
  const float w[3] = {c.cota, c.cotb, c.cotc};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const long i = e[k][0], j = e[k][1];
    if (i == j) continue;   // (a face with a repeated vertex adds nothing to W - diag(rowsum W))
    atomicAdd(&L[(size_t)i * V + j], w[k]);
    atomicAdd(&L[(size_t)j * V + i], w[k]);
  }
}
// one wave per row: L_ii = -sum_{j != i} W_ij, lane-strided partial sums + a fixed shuffle tree
__global__ __launch_bounds__(256) void k_cot_laplacian_diag(int V, float* __restrict__ L) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= V) return;
  float* row = L + (size_t)i * V;
  float s = 0.f;
  for (int j = lane; j < V; j += 64) s += (j == i) ? 0.f : row[j];
  s = wave_sum(s);
  if (lane == 0) row[i] = -s;
}

// ---- a15: Laplacian smoothing -----------------------------------------------------------------
// pass 1 (per face): accumulate Wv[i] += w_ij v_j and rowsum[i] += w_ij for the six directed
// pairs; cot weights are kept (wface) for the backward.  uniform: w = 1 per directed edge of
// the UNIQUE edge list (pass 1 then runs over edges).
__global__ void k_lap_accum_faces(const float* __restrict__ verts, const int64_t* __restrict__ faces, int P,
                                  int F, float* __restrict__ Wv, float* __restrict__ rowsum,
                                  float* __restrict__ wface) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const long i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
  if (i0 < 0 || i1 < 0 || i2 < 0 || i0 >= P || i1 >= P || i2 >= P) {
    wface[3 * f] = 0.f; wface[3 * f + 1] = 0.f; wface[3 * f + 2] = 0.f;
    return;
  }
  const FaceCot c = face_cot(verts, i0, i1, i2);
  wface[3 * f] = c.cota; wface[3 * f + 1] = c.cotb; wface[3 * f + 2] = c.cotc;
  const long e[3][2] = {{i1, i2}, {i2, i0}, {i0, i1}};
  const float w[3] = {c.cota, c.cotb, c.cotc};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const long i = e[k][0], j = e[k][1];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      atomicAdd(&Wv[3 * i + d], w[k] * verts[3 * j + d]);
      atomicAdd(&Wv[3 * j + d], w[k] * verts[3 * i + d]);
    }
    atomicAdd(&rowsum[i], w[k]);
    atomicAdd(&rowsum[j], w[k]);
  }
}

__global__ void k_lap_accum_edges(const float* __restrict__ verts, const int64_t* __restrict__ edges, int P,
                                  int E, float* __restrict__ Wv, float* __restrict__ rowsum) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const long i = edges[2 * e], j = edges[2 * e + 1];
  if (i < 0 || j < 0 || i >= P || j >= P) return;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    atomicAdd(&Wv[3 * i + d], verts[3 * j + d]);
    atomicAdd(&Wv[3 * j + d], verts[3 * i + d]);
  }
  atomicAdd(&rowsum[i], 1.0f);
  atomicAdd(&rowsum[j], 1.0f);
}

// pass 2 (per vertex): lv = Wv * nw - v with nw = 1/rowsum (0 where rowsum <= 0);
// loss += |lv| * vweight[v];  glv = vweight * lv / |lv| is stored for the backward.
__global__ __launch_bounds__(MTPB) void k_lap_vertex(const float* __restrict__ verts,
                                                     const float* __restrict__ Wv,
                                                     const float* __restrict__ rowsum,
                                                     const float* __restrict__ vweight, int P,
                                                     float* __restrict__ glv, float* __restrict__ loss) {
  __shared__ float s_red[4];
  const int v = blockIdx.x * MTPB + threadIdx.x;
  float contrib = 0.f;
  if (v < P) {
    const float rs = rowsum[v];
    const float nw = rs > 0.f ? 1.0f / rs : 0.f;
    const float lx = Wv[3 * v] * nw - verts[3 * v];
    const float ly = Wv[3 * v + 1] * nw - verts[3 * v + 1];
    const float lz = Wv[3 * v + 2] * nw - verts[3 * v + 2];
    const float nrm = sqrtf(lx * lx + ly * ly + lz * lz);
    const float w = vweight[v];
    contrib = nrm * w;
    const float inv = nrm > 0.f ? w / nrm : 0.f;   // torch: subgradient 0 at the origin
    glv[3 * v] = lx * inv; glv[3 * v + 1] = ly * inv; glv[3 * v + 2] = lz * inv;
  }
  contrib = wave_sum(contrib);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = contrib;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
}

// backward: d loss / d v_j = sum_i w_ij nw_i glv_i - glv_j  (weights are constants)
__global__ void k_lap_bwd_init(const float* __restrict__ glv, const float* __restrict__ gop, int P3,
                               float* __restrict__ gv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < P3) gv[i] = -gop[0] * glv[i];
}
__global__ void k_lap_bwd_faces(const int64_t* __restrict__ faces, const float* __restrict__ wface,
                                const float* __restrict__ rowsum, const float* __restrict__ glv,
                                const float* __restrict__ gop, int P, int F, float* __restrict__ gv) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const float go = gop[0];
  const long i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
  if (i0 < 0 || i1 < 0 || i2 < 0 || i0 >= P || i1 >= P || i2 >= P) return;
  const long e[3][2] = {{i1, i2}, {i2, i0}, {i0, i1}};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const long i = e[k][0], j = e[k][1];
    const float w = go * wface[3 * f + k];
    const float ni = rowsum[i] > 0.f ? 1.0f / rowsum[i] : 0.f, nj = rowsum[j] > 0.f ? 1.0f / rowsum[j] : 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      atomicAdd(&gv[3 * j + d], w * ni * glv[3 * i + d]);
      atomicAdd(&gv[3 * i + d], w * nj * glv[3 * j + d]);
    }
  }
}
__global__ void k_lap_bwd_edges(const int64_t* __restrict__ edges, const float* __restrict__ rowsum,
                                const float* __restrict__ glv, const float* __restrict__ gop, int P, int E,
                                float* __restrict__ gv) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const float go = gop[0];
  const long i = edges[2 * e], j = edges[2 * e + 1];
  if (i < 0 || j < 0 || i >= P || j >= P) return;
  const float ni = rowsum[i] > 0.f ? 1.0f / rowsum[i] : 0.f, nj = rowsum[j] > 0.f ? 1.0f / rowsum[j] : 0.f;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    atomicAdd(&gv[3 * j + d], go * ni * glv[3 * i + d]);
    atomicAdd(&gv[3 * i + d], go * nj * glv[3 * j + d]);
  }
}

// Equal-sized meshes (the trainer's Meshes(verts [N,V,3], faces [N,F,3]): mesh m owns the packed vertices
// [m vpm, (m+1) vpm) and faces [m fpm, (m+1) fpm)): one workgroup per mesh keeps the mesh's W v and row
// sums in LDS, so the six directed pairs of a face cost LDS atomics instead of 24 scattered global ones
// (82 k faces: 64 us forward + 54 us backward before), and forward / backward are one launch each without
// zero fills.  A face with a vertex outside its mesh's range is skipped (as one outside [0, P) is).
constexpr int LTB = 512;
__global__ __launch_bounds__(LTB) void k_lap_mesh_fwd(const float* __restrict__ verts,
                                                      const int64_t* __restrict__ faces,
                                                      const float* __restrict__ vweight, int vpm, int fpm,
                                                      float* __restrict__ rowsum_out, float* __restrict__ glv,
                                                      float* __restrict__ wface, float* __restrict__ loss) {
  extern __shared__ float s_l[];           // Wv [vpm][3], rowsum [vpm]
  __shared__ float s_red[LTB / 64];
  float* s_wv = s_l;
  float* s_rs = s_l + 3 * (size_t)vpm;
  const int m = blockIdx.x, tid = threadIdx.x;
  const long vb = (long)m * vpm;
  for (int i = tid; i < 4 * vpm; i += LTB) s_l[i] = 0.f;
  __syncthreads();
  for (int fl = tid; fl < fpm; fl += LTB) {
    const size_t f = (size_t)m * fpm + fl;
    const long i0 = faces[3 * f] - vb, i1 = faces[3 * f + 1] - vb, i2 = faces[3 * f + 2] - vb;
    if (i0 < 0 || i1 < 0 || i2 < 0 || i0 >= vpm || i1 >= vpm || i2 >= vpm) {
      wface[3 * f] = 0.f; wface[3 * f + 1] = 0.f; wface[3 * f + 2] = 0.f;
      continue;
    }
    const FaceCot c = face_cot(verts, i0 + vb, i1 + vb, i2 + vb);
    wface[3 * f] = c.cota; wface[3 * f + 1] = c.cotb; wface[3 * f + 2] = c.cotc;
    // edge (i1,i2) carries cota, (i2,i0) cotb, (i0,i1) cotc: a vertex receives the two terms of its two
    // edges of this face in one go (12 LDS atomics per face instead of 24)
    const long vi[3] = {i0, i1, i2};
    const float wa[3] = {c.cotc, c.cota, c.cotb};   // weight towards the next vertex of the face
    const float wb[3] = {c.cotb, c.cotc, c.cota};   // weight towards the previous one
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const long i = vi[k], jn = vi[(k + 1) % 3] + vb, jp = vi[(k + 2) % 3] + vb;
#pragma unroll
      for (int d = 0; d < 3; ++d) atomicAdd(&s_wv[3 * i + d], wa[k] * verts[3 * jn + d] + wb[k] * verts[3 * jp + d]);
      atomicAdd(&s_rs[i], wa[k] + wb[k]);
    }
  }
  __syncthreads();
  float contrib = 0.f;
  for (int vl = tid; vl < vpm; vl += LTB) {
    const size_t v = (size_t)vb + vl;
    const float rs = s_rs[vl];
    rowsum_out[v] = rs;
    const float nw = rs > 0.f ? 1.0f / rs : 0.f;
    const float lx = s_wv[3 * vl] * nw - verts[3 * v];
    const float ly = s_wv[3 * vl + 1] * nw - verts[3 * v + 1];
    const float lz = s_wv[3 * vl + 2] * nw - verts[3 * v + 2];
    const float nrm = sqrtf(lx * lx + ly * ly + lz * lz);
    const float w = vweight[v];
    contrib += nrm * w;
    const float inv = nrm > 0.f ? w / nrm : 0.f;   // torch: subgradient 0 at the origin
    glv[3 * v] = lx * inv; glv[3 * v + 1] = ly * inv; glv[3 * v + 2] = lz * inv;
  }
  contrib = wave_sum(contrib);
  if ((tid & 63) == 0) s_red[tid >> 6] = contrib;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int i = 0; i < LTB / 64; ++i) t += s_red[i];
    atomicAdd(loss, t);
  }
}

__global__ __launch_bounds__(LTB) void k_lap_mesh_bwd(const int64_t* __restrict__ faces,
                                                      const float* __restrict__ wface,
                                                      const float* __restrict__ rowsum,
                                                      const float* __restrict__ glv,
                                                      const float* __restrict__ gop, int vpm, int fpm,
                                                      float* __restrict__ gv) {
  extern __shared__ float s_l[];           // gv [vpm][3], 1/rowsum [vpm]
  float* s_g = s_l;
  float* s_nw = s_l + 3 * (size_t)vpm;
  const int m = blockIdx.x, tid = threadIdx.x;
  const long vb = (long)m * vpm;
  const float go = gop[0];
  for (int i = tid; i < 3 * vpm; i += LTB) s_g[i] = -go * glv[3 * (size_t)vb + i];
  for (int i = tid; i < vpm; i += LTB) { const float rs = rowsum[vb + i]; s_nw[i] = rs > 0.f ? 1.0f / rs : 0.f; }
  __syncthreads();
  for (int fl = tid; fl < fpm; fl += LTB) {
    const size_t f = (size_t)m * fpm + fl;
    const long i0 = faces[3 * f] - vb, i1 = faces[3 * f + 1] - vb, i2 = faces[3 * f + 2] - vb;
    if (i0 < 0 || i1 < 0 || i2 < 0 || i0 >= vpm || i1 >= vpm || i2 >= vpm) continue;
    const long vi[3] = {i0, i1, i2};
    const float cota = go * wface[3 * f], cotb = go * wface[3 * f + 1], cotc = go * wface[3 * f + 2];
    const float wa[3] = {cotc, cota, cotb}, wb[3] = {cotb, cotc, cota};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const long i = vi[k], jn = vi[(k + 1) % 3], jp = vi[(k + 2) % 3];
      const float an = wa[k] * s_nw[jn], ap = wb[k] * s_nw[jp];
#pragma unroll
      for (int d = 0; d < 3; ++d)
        atomicAdd(&s_g[3 * i + d], an * glv[3 * (jn + vb) + d] + ap * glv[3 * (jp + vb) + d]);
    }
  }
  __syncthreads();
  for (int i = tid; i < 3 * vpm; i += LTB) gv[3 * (size_t)vb + i] = s_g[i];
}

// ---- a14: edge rigidity --------------------------------------------------------------------
__global__ __launch_bounds__(MTPB) void k_rigid(const float* __restrict__ v, const int64_t* __restrict__ e,
                                                const float* __restrict__ vt, const int64_t* __restrict__ et,
                                                int E, float* __restrict__ loss) {
  __shared__ float s_red[4];
  const int i = blockIdx.x * MTPB + threadIdx.x;
  float c = 0.f;
  if (i < E) {
    const long a = e[2 * i], b = e[2 * i + 1], at = et[2 * i], bt = et[2 * i + 1];
    const float dx = v[3 * a] - v[3 * b], dy = v[3 * a + 1] - v[3 * b + 1], dz = v[3 * a + 2] - v[3 * b + 2];
    const float tx = vt[3 * at] - vt[3 * bt], ty = vt[3 * at + 1] - vt[3 * bt + 1], tz = vt[3 * at + 2] - vt[3 * bt + 2];
    const float d = sqrtf(dx * dx + dy * dy + dz * dz) - sqrtf(tx * tx + ty * ty + tz * tz);
    c = d * d;
  }
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
}

__global__ void k_rigid_bwd(const float* __restrict__ v, const int64_t* __restrict__ e,
                            const float* __restrict__ vt, const int64_t* __restrict__ et, int E,
                            const float* __restrict__ gop, float* __restrict__ gv, float* __restrict__ gvt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E) return;
  const float go = gop[0];
  const long a = e[2 * i], b = e[2 * i + 1], at = et[2 * i], bt = et[2 * i + 1];
  const float dx = v[3 * a] - v[3 * b], dy = v[3 * a + 1] - v[3 * b + 1], dz = v[3 * a + 2] - v[3 * b + 2];
  const float tx = vt[3 * at] - vt[3 * bt], ty = vt[3 * at + 1] - vt[3 * bt + 1], tz = vt[3 * at + 2] - vt[3 * bt + 2];
  const float ld = sqrtf(dx * dx + dy * dy + dz * dz), lt = sqrtf(tx * tx + ty * ty + tz * tz);
  const float g = go * 2.0f * (ld - lt);
  if (gv) {
    const float s = ld > 0.f ? g / ld : 0.f;
    atomicAdd(&gv[3 * a], s * dx); atomicAdd(&gv[3 * a + 1], s * dy); atomicAdd(&gv[3 * a + 2], s * dz);
    atomicAdd(&gv[3 * b], -s * dx); atomicAdd(&gv[3 * b + 1], -s * dy); atomicAdd(&gv[3 * b + 2], -s * dz);
  }
  if (gvt) {
    const float s = lt > 0.f ? -g / lt : 0.f;
    atomicAdd(&gvt[3 * at], s * tx); atomicAdd(&gvt[3 * at + 1], s * ty); atomicAdd(&gvt[3 * at + 2], s * tz);
    atomicAdd(&gvt[3 * bt], -s * tx); atomicAdd(&gvt[3 * bt + 1], -s * ty); atomicAdd(&gvt[3 * bt + 2], -s * tz);
  }
}

static inline unsigned nblk(long n, int b) { return (unsigned)((n + b - 1) / b); }

// Equal-sized meshes with the edge list sorted by its first vertex (Meshes.edges_packed()): mesh m's
// edges are the contiguous run whose first vertex lies in [m vpm, (m+1) vpm).  One workgroup per mesh
// finds the run by bisection, sums the edge gradients per vertex in LDS and stores the mesh's [vpm,3]
// rows: no global atomics (740 k scattered ones took 27 us), no zero fill.
__global__ __launch_bounds__(LTB) void k_rigid_mesh_bwd(const float* __restrict__ v, const int64_t* __restrict__ e,
                                                        const float* __restrict__ vt, const int64_t* __restrict__ et,
                                                        int E, int vpm, const float* __restrict__ gop,
                                                        float* __restrict__ gv) {
  extern __shared__ float s_l[];           // gv [vpm][3]
  const int m = blockIdx.x, tid = threadIdx.x;
  const long vb = (long)m * vpm, ve = vb + vpm;
  for (int i = tid; i < 3 * vpm; i += LTB) s_l[i] = 0.f;
  auto lower = [&](long key) {             // first edge whose first vertex is >= key
    int lo = 0, hi = E;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (e[2 * (size_t)mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  // meshes of one topology have E / N edges each: try that run first (four loads) before the bisection
  // (17 dependent loads ~ 17 us)
  const int nm = gridDim.x;
  int e0 = (int)((long)E * m / nm), e1 = (int)((long)E * (m + 1) / nm);
  const bool ok0 = (e0 == 0 || e[2 * (size_t)(e0 - 1)] < vb) && (e0 == E || e[2 * (size_t)e0] >= vb);
  const bool ok1 = (e1 == 0 || e[2 * (size_t)(e1 - 1)] < ve) && (e1 == E || e[2 * (size_t)e1] >= ve);
  if (!ok0) e0 = lower(vb);
  if (!ok1) e1 = lower(ve);
  __syncthreads();
  const float go = gop[0];
  for (int i0 = e0 + tid; i0 < e1; i0 += 4 * LTB) {     // four edges per thread: index loads, then vertex loads, in flight together
    long a[4], b[4], at[4], bt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = min(i0 + u * LTB, e1 - 1);
      a[u] = e[2 * (size_t)i]; b[u] = e[2 * (size_t)i + 1]; at[u] = et[2 * (size_t)i]; bt[u] = et[2 * (size_t)i + 1];
    }
    float d[4][3], t[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        d[u][c] = v[3 * a[u] + c] - v[3 * b[u] + c];
        t[u][c] = vt[3 * at[u] + c] - vt[3 * bt[u] + c];
      }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (i0 + u * LTB >= e1 || b[u] < vb || b[u] >= ve) continue;   // (an edge never leaves its mesh)
      const float ld = sqrtf(d[u][0] * d[u][0] + d[u][1] * d[u][1] + d[u][2] * d[u][2]);
      const float lt = sqrtf(t[u][0] * t[u][0] + t[u][1] * t[u][1] + t[u][2] * t[u][2]);
      const float g = go * 2.0f * (ld - lt);
      const float sc = ld > 0.f ? g / ld : 0.f;
      float* ga = s_l + 3 * (a[u] - vb);
      float* gb = s_l + 3 * (b[u] - vb);
      atomicAdd(&ga[0], sc * d[u][0]); atomicAdd(&ga[1], sc * d[u][1]); atomicAdd(&ga[2], sc * d[u][2]);
      atomicAdd(&gb[0], -sc * d[u][0]); atomicAdd(&gb[1], -sc * d[u][1]); atomicAdd(&gb[2], -sc * d[u][2]);
    }
  }
  __syncthreads();
  for (int i = tid; i < 3 * vpm; i += LTB) gv[3 * (size_t)vb + i] = s_l[i];
}

}  // namespace acfm

using namespace acfm;

extern "C" {

int acfm_cot_laplacian(const float* verts, const int64_t* faces, int V, int F, float* L, void* stream) {
  if (!verts || !faces || !L || V <= 0 || F <= 0) return ACFM_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (zero_async(L, sizeof(float) * (size_t)V * V, st) != ACFM_OK) return ACFM_E_LAUNCH;
  hipLaunchKernelGGL(k_cot_laplacian, dim3(nblk(F, 256)), dim3(256), 0, st, verts, faces, V, F, L);
  hipLaunchKernelGGL(k_cot_laplacian_diag, dim3((V + 3) / 4), dim3(256), 0, st, V, L);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

size_t acfm_laplacian_smoothing_state_floats(int P, int F) {
  if (P <= 0 || F < 0) return 0;
  return (size_t)7 * P + (size_t)3 * F;  // Wv [P,3] | rowsum [P] | glv [P,3] | wface [F,3]
}

// method 0: cot (faces [F,3] packed ids); method 1: uniform (edges [E,2] unique packed edges, F = E).
static bool lap_blocked(int P, int F, int method, int vpm, int fpm) {
  return method == 0 && vpm > 0 && fpm > 0 && P % vpm == 0 && F % fpm == 0 && P / vpm == F / fpm &&
         sizeof(float) * 4 * (size_t)vpm <= 150 * 1024;
}

int acfm_laplacian_smoothing(const float* verts, const int64_t* conn, const float* vweight, int P, int F,
                             int method, int verts_per_mesh, int faces_per_mesh, float* loss, float* state,
                             void* stream) {
  if (!verts || !conn || !vweight || !loss || !state || P <= 0 || F <= 0 || (method != 0 && method != 1))
    return ACFM_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  float* Wv = state; float* rowsum = state + 3 * (size_t)P; float* glv = state + 4 * (size_t)P;
  float* wface = state + 7 * (size_t)P;
  if (lap_blocked(P, F, method, verts_per_mesh, faces_per_mesh)) {
    if (zero_async(loss, sizeof(float), st) != ACFM_OK) return ACFM_E_LAUNCH;
    hipLaunchKernelGGL(k_lap_mesh_fwd, dim3(P / verts_per_mesh), dim3(LTB), sizeof(float) * 4 * (size_t)verts_per_mesh,
                       st, verts, conn, vweight, verts_per_mesh, faces_per_mesh, rowsum, glv, wface, loss);
    ACFM_CHECK_LAUNCH();
    return ACFM_OK;
  }
  if (zero_async(state, sizeof(float) * 4 * (size_t)P, st) != ACFM_OK) return ACFM_E_LAUNCH;
  if (zero_async(loss, sizeof(float), st) != ACFM_OK) return ACFM_E_LAUNCH;
  if (method == 0)
    hipLaunchKernelGGL(k_lap_accum_faces, dim3(nblk(F, 256)), dim3(256), 0, st, verts, conn, P, F, Wv, rowsum, wface);
  else
    hipLaunchKernelGGL(k_lap_accum_edges, dim3(nblk(F, 256)), dim3(256), 0, st, verts, conn, P, F, Wv, rowsum);
  hipLaunchKernelGGL(k_lap_vertex, dim3(nblk(P, MTPB)), dim3(MTPB), 0, st, verts, (const float*)Wv,
                     (const float*)rowsum, vweight, P, glv, loss);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_laplacian_smoothing_backward(const int64_t* conn, const float* state, const float* grad_loss, int P, int F,
                                      int method, int verts_per_mesh, int faces_per_mesh, float* grad_verts,
                                      void* stream) {
  if (!conn || !state || !grad_loss || !grad_verts || P <= 0 || F <= 0 || (method != 0 && method != 1))
    return ACFM_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const float* rowsum = state + 3 * (size_t)P; const float* glv = state + 4 * (size_t)P;
  const float* wface = state + 7 * (size_t)P;
  if (lap_blocked(P, F, method, verts_per_mesh, faces_per_mesh)) {
    hipLaunchKernelGGL(k_lap_mesh_bwd, dim3(P / verts_per_mesh), dim3(LTB), sizeof(float) * 4 * (size_t)verts_per_mesh,
                       st, conn, wface, rowsum, glv, grad_loss, verts_per_mesh, faces_per_mesh, grad_verts);
    ACFM_CHECK_LAUNCH();
    return ACFM_OK;
  }
  hipLaunchKernelGGL(k_lap_bwd_init, dim3(nblk(3L * P, 256)), dim3(256), 0, st, glv, grad_loss, 3 * P, grad_verts);
  if (method == 0)
    hipLaunchKernelGGL(k_lap_bwd_faces, dim3(nblk(F, 256)), dim3(256), 0, st, conn, wface, rowsum, glv, grad_loss,
                       P, F, grad_verts);
  else
    hipLaunchKernelGGL(k_lap_bwd_edges, dim3(nblk(F, 256)), dim3(256), 0, st, conn, rowsum, glv, grad_loss, P, F,
                       grad_verts);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_edge_rigidity(const float* verts, const int64_t* edges, const float* verts_t, const int64_t* edges_t,
                       int E, float* loss, void* stream) {
  if (!verts || !edges || !verts_t || !edges_t || !loss || E <= 0) return ACFM_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (zero_async(loss, sizeof(float), st) != ACFM_OK) return ACFM_E_LAUNCH;
  hipLaunchKernelGGL(k_rigid, dim3(nblk(E, MTPB)), dim3(MTPB), 0, st, verts, edges, verts_t, edges_t, E, loss);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_edge_rigidity_backward(const float* verts, const int64_t* edges, const float* verts_t,
                                const int64_t* edges_t, int E, int P, int Pt, int verts_per_mesh,
                                const float* grad_loss, float* grad_verts, float* grad_verts_t, void* stream) {
  if (!verts || !edges || !verts_t || !edges_t || !grad_loss || E <= 0 || P <= 0 || Pt <= 0) return ACFM_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (grad_verts && !grad_verts_t && verts_per_mesh > 0 && P % verts_per_mesh == 0 &&
      sizeof(float) * 3 * (size_t)verts_per_mesh <= 150 * 1024) {
    hipLaunchKernelGGL(k_rigid_mesh_bwd, dim3(P / verts_per_mesh), dim3(LTB), sizeof(float) * 3 * (size_t)verts_per_mesh,
                       st, verts, edges, verts_t, edges_t, E, verts_per_mesh, grad_loss, grad_verts);
    ACFM_CHECK_LAUNCH();
    return ACFM_OK;
  }
  if (grad_verts && zero_async(grad_verts, sizeof(float) * 3 * (size_t)P, st) != ACFM_OK) return ACFM_E_LAUNCH;
  if (grad_verts_t && zero_async(grad_verts_t, sizeof(float) * 3 * (size_t)Pt, st) != ACFM_OK) return ACFM_E_LAUNCH;
  hipLaunchKernelGGL(k_rigid_bwd, dim3(nblk(E, 256)), dim3(256), 0, st, verts, edges, verts_t, edges_t, E, grad_loss,
                     grad_verts, grad_verts_t);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

}  // extern "C"
