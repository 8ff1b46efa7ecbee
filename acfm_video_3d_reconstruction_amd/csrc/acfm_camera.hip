// Camera hypothesis pipeline of the multiframe trainer in one kernel:
//   decode     embedding e[7] -> (s = relu(decay*e0 + 1) + 1e-12, t = e1..2, q = normalize(e3..6))
//              (multiframe/main.py:572-577, == :452-457, predictor.py:248-252)
//   mirror     blend with the pose of the horizontally flipped image by the frame's mirror flag:
//              (s, -tx, ty, q_y(pi) * standardize(q)), standardized      (main.py:97-125)
//   transform  crop/scale augmentation (a, dx, dy, flag) applied to scale and translation
//              (main.py:128-138)
// The reference runs this as ~60 elementwise torch kernels on [G*N, 7] tensors per step plus ~100
// in the backward (the camera embeddings are optimised through it, train_utils.py:186-213); here
// it is one thread per camera, forward and backward.  Row r of the [G*N, 7] batch belongs to frame
// r % N (the reference repeats the per-frame flags G times).
#include "acfm_common.h"

namespace acfm {

struct CamFwd {
  float s, q[4], b_sign, raw0_sign, nrm;
  bool relu_on;
};

__device__ __forceinline__ void cam_forward(const float* __restrict__ e, float decay, float m, const float* __restrict__ tr,
                                            float* __restrict__ o, CamFwd& c) {
  const float pre = decay * e[0] + 1.0f;
  c.relu_on = pre > 0.0f;
  c.s = fmaxf(pre, 0.0f) + 1e-12f;
  c.nrm = sqrtf(e[3] * e[3] + e[4] * e[4] + e[5] * e[5] + e[6] * e[6]);
  const float d = fmaxf(c.nrm, 1e-12f);
#pragma unroll
  for (int k = 0; k < 4; ++k) c.q[k] = e[3 + k] / d;
  // mirrored pose: q_m = standardize(q_y(pi) * standardize(q)), q_y(pi) = (0, 0, 1, 0)
  c.b_sign = c.q[0] < 0.0f ? -1.0f : 1.0f;
  const float b0 = c.b_sign * c.q[0], b1 = c.b_sign * c.q[1], b2 = c.b_sign * c.q[2], b3 = c.b_sign * c.q[3];
  float raw[4] = {-b2, b3, b0, -b1};
  c.raw0_sign = raw[0] < 0.0f ? -1.0f : 1.0f;
  const float pose[7] = {c.s, e[1], e[2], c.q[0], c.q[1], c.q[2], c.q[3]};
  const float mir[7] = {c.s, -e[1], e[2], c.raw0_sign * raw[0], c.raw0_sign * raw[1], c.raw0_sign * raw[2],
                        c.raw0_sign * raw[3]};
  float p1[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) p1[k] = (1.0f - m) * pose[k] + mir[k] * m;
  const float a = tr[0], dx = tr[1], dy = tr[2], f = tr[3];
  const float nw[7] = {p1[0] * a, p1[1] * a + dx, p1[2] * a + dy, p1[3], p1[4], p1[5], p1[6]};
#pragma unroll
  for (int k = 0; k < 7; ++k) o[k] = (1.0f - f) * p1[k] + nw[k] * f;
}

__global__ void k_camera_fwd(const float* __restrict__ emb, const int64_t* __restrict__ mirror,
                             const float* __restrict__ transforms, int R, int N, float decay,
                             float* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const int n = r % N;
  float e[7], o[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) e[k] = emb[(size_t)r * 7 + k];
  const float tr[4] = {transforms[4 * n], transforms[4 * n + 1], transforms[4 * n + 2], transforms[4 * n + 3]};
  CamFwd c;
  cam_forward(e, decay, (float)mirror[n], tr, o, c);
#pragma unroll
  for (int k = 0; k < 7; ++k) out[(size_t)r * 7 + k] = o[k];
}

// d embedding from d camera (the forward is re-evaluated)
__device__ __forceinline__ void cam_backward(const float* __restrict__ e, const float* __restrict__ g, float decay, float m,
                                             const float* __restrict__ tr, float* __restrict__ ge) {
  float o[7];
  CamFwd c;
  cam_forward(e, decay, m, tr, o, c);
  const float a = tr[0], f = tr[3];
  // out = (1-f) p1 + f new(p1)
  float gp1[7];
  gp1[0] = (1.0f - f) * g[0] + f * g[0] * a;
  gp1[1] = (1.0f - f) * g[1] + f * g[1] * a;
  gp1[2] = (1.0f - f) * g[2] + f * g[2] * a;
#pragma unroll
  for (int k = 3; k < 7; ++k) gp1[k] = (1.0f - f) * g[k] + f * g[k];
  // p1 = (1-m) pose + m mir
  float gpose[7], gmir[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) { gpose[k] = (1.0f - m) * gp1[k]; gmir[k] = m * gp1[k]; }
  float gs = gpose[0] + gmir[0];
  const float ge1 = gpose[1] - gmir[1], ge2 = gpose[2] + gmir[2];
  // mir q = raw0_sign * raw, raw = (-b2, b3, b0, -b1), b = b_sign * q
  const float gr0 = c.raw0_sign * gmir[3], gr1 = c.raw0_sign * gmir[4], gr2 = c.raw0_sign * gmir[5],
              gr3 = c.raw0_sign * gmir[6];
  float gq[4] = {gpose[3] + c.b_sign * gr2, gpose[4] - c.b_sign * gr3, gpose[5] - c.b_sign * gr0,
                 gpose[6] + c.b_sign * gr1};
  // q = e / max(|e|, eps)
  if (c.nrm > 1e-12f) {
    const float dot = c.q[0] * gq[0] + c.q[1] * gq[1] + c.q[2] * gq[2] + c.q[3] * gq[3];
#pragma unroll
    for (int k = 0; k < 4; ++k) ge[3 + k] = (gq[k] - c.q[k] * dot) / c.nrm;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) ge[3 + k] = gq[k] / 1e-12f;
  }
  ge[0] = c.relu_on ? decay * gs : 0.0f;
  ge[1] = ge1;
  ge[2] = ge2;
}


// The same straight from the per-hypothesis embedding tables (mesh_net.py:436-444: one nn.Embedding(frames, 7) per
// hypothesis): row r = g N + n reads tables[sel ? sel[r] : g][frames_idx[n]]; the backward adds into dense per-table
// gradients (zeroed by the caller; float atomics: frames of a batch are distinct, so every cell gets one add).
// A frame id outside [0, n_frames) or a table id outside [0, n_tables) -- nn.Embedding raises on those (main.py:551-570)
// -- never becomes an address: the forward writes a NaN camera for that row (the loss of the step is then NaN, loudly,
// without a host synchronisation inside a captured step), the backward skips it; ops.camera_pipeline_tables(check=True)
// is the host-side check for data-loader batches.
constexpr int CAM_MAX_TABLES = 32;
struct CamTables {
  const float* t[CAM_MAX_TABLES];
  float* g[CAM_MAX_TABLES];
};
__global__ void k_camera_fwd_tables(CamTables tb, const int64_t* __restrict__ frames_idx, const int64_t* __restrict__ sel,
                                    const int64_t* __restrict__ mirror, const float* __restrict__ transforms, int R, int N,
                                    int n_tables, int n_frames, float decay, float* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const int n = r % N;
  const int64_t table = sel ? sel[r] : (int64_t)(r / N);
  const int64_t frame = frames_idx[n];
  if (table < 0 || table >= n_tables || frame < 0 || frame >= n_frames) {
#pragma unroll
    for (int k = 0; k < 7; ++k) out[(size_t)r * 7 + k] = __builtin_nanf("");
    return;
  }
  const float* src = tb.t[table] + (size_t)frame * 7;
  float e[7], o[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) e[k] = src[k];
  const float tr[4] = {transforms[4 * n], transforms[4 * n + 1], transforms[4 * n + 2], transforms[4 * n + 3]};
  CamFwd c;
  cam_forward(e, decay, (float)mirror[n], tr, o, c);
#pragma unroll
  for (int k = 0; k < 7; ++k) out[(size_t)r * 7 + k] = o[k];
}

__global__ void k_camera_bwd(const float* __restrict__ emb, const int64_t* __restrict__ mirror,
                             const float* __restrict__ transforms, const float* __restrict__ gout, int R,
                             int N, float decay, float* __restrict__ gemb) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const int n = r % N;
  float e[7], g[7], ge[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) { e[k] = emb[(size_t)r * 7 + k]; g[k] = gout[(size_t)r * 7 + k]; }
  const float tr[4] = {transforms[4 * n], transforms[4 * n + 1], transforms[4 * n + 2], transforms[4 * n + 3]};
  cam_backward(e, g, decay, (float)mirror[n], tr, ge);
#pragma unroll
  for (int k = 0; k < 7; ++k) gemb[(size_t)r * 7 + k] = ge[k];
}

__global__ void k_camera_bwd_tables(CamTables tb, const int64_t* __restrict__ frames_idx, const int64_t* __restrict__ sel,
                                    const int64_t* __restrict__ mirror, const float* __restrict__ transforms,
                                    const float* __restrict__ gout, int R, int N, int n_tables, int n_frames,
                                    float decay) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const int n = r % N;
  const int64_t table = sel ? sel[r] : (int64_t)(r / N);
  const int64_t frame = frames_idx[n];
  if (table < 0 || table >= n_tables || frame < 0 || frame >= n_frames) return;   // (the forward wrote NaN for this row)
  if (!tb.g[table]) return;
  const size_t row = (size_t)frame * 7;
  const float* src = tb.t[table] + row;
  float e[7], g[7], ge[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) { e[k] = src[k]; g[k] = gout[(size_t)r * 7 + k]; }
  const float tr[4] = {transforms[4 * n], transforms[4 * n + 1], transforms[4 * n + 2], transforms[4 * n + 3]};
  cam_backward(e, g, decay, (float)mirror[n], tr, ge);
#pragma unroll
  for (int k = 0; k < 7; ++k) atomicAdd(tb.g[table] + row + k, ge[k]);
}

// cam = (s, tx, ty, q / max(|q|, 1e-12)): the camera the refinement loop renders with while it
// optimises scale, translation and an unnormalised quaternion (predictor.py:301-308:
// torch.cat([scale, trans, F.normalize(quat)])); one thread per camera, forward and backward
__global__ void k_camera_normalize(const float* __restrict__ raw, int N, float* __restrict__ out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* e = raw + (size_t)n * 7;
  const float nrm = sqrtf(e[3] * e[3] + e[4] * e[4] + e[5] * e[5] + e[6] * e[6]);
  const float d = fmaxf(nrm, 1e-12f);
  float* o = out + (size_t)n * 7;
  o[0] = e[0]; o[1] = e[1]; o[2] = e[2];
#pragma unroll
  for (int k = 3; k < 7; ++k) o[k] = e[k] / d;
}

__global__ void k_camera_normalize_bwd(const float* __restrict__ raw, const float* __restrict__ gout, int N,
                                       float* __restrict__ graw) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* e = raw + (size_t)n * 7;
  const float* g = gout + (size_t)n * 7;
  float* o = graw + (size_t)n * 7;
  o[0] = g[0]; o[1] = g[1]; o[2] = g[2];
  const float nrm = sqrtf(e[3] * e[3] + e[4] * e[4] + e[5] * e[5] + e[6] * e[6]);
  if (nrm > 1e-12f) {
    const float q0 = e[3] / nrm, q1 = e[4] / nrm, q2 = e[5] / nrm, q3 = e[6] / nrm;
    const float dot = q0 * g[3] + q1 * g[4] + q2 * g[5] + q3 * g[6];
    o[3] = (g[3] - q0 * dot) / nrm; o[4] = (g[4] - q1 * dot) / nrm;
    o[5] = (g[5] - q2 * dot) / nrm; o[6] = (g[6] - q3 * dot) / nrm;
  } else {
#pragma unroll
    for (int k = 3; k < 7; ++k) o[k] = g[k] / 1e-12f;
  }
}

// pose of the horizontally flipped image of an already decoded camera (multiframe/main.py:97-125 with flag 1):
// (s, tx, ty, q) -> (s, -tx, ty, standardize(q_y(pi) * standardize(q))), q_y(pi) = (0, 0, 1, 0); one thread per camera.
// (The reference's chain of pytorch3d.transforms calls is ~35 elementwise launches on [R,4] tensors.)
__global__ void k_camera_mirror(const float* __restrict__ cams, int R, float* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float* c = cams + (size_t)r * 7;
  const float sg = c[3] < 0.0f ? -1.0f : 1.0f;
  const float b0 = sg * c[3], b1 = sg * c[4], b2 = sg * c[5], b3 = sg * c[6];
  const float raw[4] = {-b2, b3, b0, -b1};
  const float s2 = raw[0] < 0.0f ? -1.0f : 1.0f;
  float* o = out + (size_t)r * 7;
  o[0] = c[0]; o[1] = -c[1]; o[2] = c[2];
  o[3] = s2 * raw[0]; o[4] = s2 * raw[1]; o[5] = s2 * raw[2]; o[6] = s2 * raw[3];
}

}  // namespace acfm

using namespace acfm;

extern "C" {

int acfm_camera_pipeline(const float* emb, const int64_t* mirror_flag, const float* transforms, int R, int N,
                         float scale_lr_decay, float* cams, void* stream) {
  if (!emb || !mirror_flag || !transforms || !cams || R <= 0 || N <= 0 || R % N != 0) return ACFM_E_BADARG;
  hipLaunchKernelGGL(k_camera_fwd, dim3((R + 127) / 128), dim3(128), 0, (hipStream_t)stream, emb, mirror_flag,
                     transforms, R, N, scale_lr_decay, cams);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_camera_pipeline_backward(const float* emb, const int64_t* mirror_flag, const float* transforms,
                                  const float* grad_cams, int R, int N, float scale_lr_decay, float* grad_emb,
                                  void* stream) {
  if (!emb || !mirror_flag || !transforms || !grad_cams || !grad_emb || R <= 0 || N <= 0 || R % N != 0)
    return ACFM_E_BADARG;
  hipLaunchKernelGGL(k_camera_bwd, dim3((R + 127) / 128), dim3(128), 0, (hipStream_t)stream, emb, mirror_flag,
                     transforms, grad_cams, R, N, scale_lr_decay, grad_emb);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_camera_pipeline_tables(const void* const* tables, int n_tables, int n_frames, const int64_t* frames_idx,
                                const int64_t* selected, const int64_t* mirror_flag, const float* transforms, int R,
                                int N, float scale_lr_decay, float* cams, void* stream) {
  if (!tables || !frames_idx || !mirror_flag || !transforms || !cams || n_tables <= 0 || n_tables > CAM_MAX_TABLES ||
      n_frames <= 0 || R <= 0 || N <= 0 || R % N != 0 || (!selected && R / N > n_tables))
    return ACFM_E_BADARG;
  CamTables tb;
  for (int i = 0; i < CAM_MAX_TABLES; ++i) {
    tb.t[i] = i < n_tables ? (const float*)tables[i] : nullptr;
    tb.g[i] = nullptr;
    if (i < n_tables && !tb.t[i]) return ACFM_E_BADARG;
  }
  hipLaunchKernelGGL(k_camera_fwd_tables, dim3((R + 127) / 128), dim3(128), 0, (hipStream_t)stream, tb, frames_idx,
                     selected, mirror_flag, transforms, R, N, n_tables, n_frames, scale_lr_decay, cams);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_camera_pipeline_tables_backward(const void* const* tables, int n_tables, int n_frames, const int64_t* frames_idx,
                                         const int64_t* selected, const int64_t* mirror_flag, const float* transforms,
                                         const float* grad_cams, int R, int N, float scale_lr_decay,
                                         void* const* grad_tables, void* stream) {
  if (!tables || !grad_tables || !frames_idx || !mirror_flag || !transforms || !grad_cams || n_tables <= 0 ||
      n_tables > CAM_MAX_TABLES || n_frames <= 0 || R <= 0 || N <= 0 || R % N != 0 || (!selected && R / N > n_tables))
    return ACFM_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  CamTables tb;
  for (int i = 0; i < CAM_MAX_TABLES; ++i) {
    tb.t[i] = i < n_tables ? (const float*)tables[i] : nullptr;
    tb.g[i] = i < n_tables ? (float*)grad_tables[i] : nullptr;
    if (i < n_tables && !tb.t[i]) return ACFM_E_BADARG;
    if (tb.g[i] && zero_async(tb.g[i], sizeof(float) * 7 * (size_t)n_frames, st) != ACFM_OK) return ACFM_E_LAUNCH;
  }
  hipLaunchKernelGGL(k_camera_bwd_tables, dim3((R + 127) / 128), dim3(128), 0, st, tb, frames_idx, selected,
                     mirror_flag, transforms, grad_cams, R, N, n_tables, n_frames, scale_lr_decay);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_camera_mirror(const float* cams, int R, float* out, void* stream) {
  if (!cams || !out || R <= 0) return ACFM_E_BADARG;
  hipLaunchKernelGGL(k_camera_mirror, dim3((R + 127) / 128), dim3(128), 0, (hipStream_t)stream, cams, R, out);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_camera_normalize(const float* cam_raw, int N, float* cams, void* stream) {
  if (!cam_raw || !cams || N <= 0) return ACFM_E_BADARG;
  hipLaunchKernelGGL(k_camera_normalize, dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream, cam_raw, N, cams);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_camera_normalize_backward(const float* cam_raw, const float* grad_cams, int N, float* grad_raw,
                                   void* stream) {
  if (!cam_raw || !grad_cams || !grad_raw || N <= 0) return ACFM_E_BADARG;
  hipLaunchKernelGGL(k_camera_normalize_bwd, dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream, cam_raw,
                     grad_cams, N, grad_raw);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

}  // extern "C"
