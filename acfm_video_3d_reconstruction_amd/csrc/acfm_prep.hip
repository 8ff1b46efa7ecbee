// On-device input preparation: what multiframe/main.py:365-377 does per
// batch on the CPU with scipy / scikit-image (multiframe/utils/image.py:94-146) followed by a
// device->host->device round trip of the masks.
//   acfm_edt          <- scipy.ndimage.distance_transform_edt(1 - mask)      (image.py:94-102)
//   acfm_boundaries   <- skimage.segmentation.find_boundaries(mask) + the point list /
//                        normalisation / padding of compute_boundaries       (image.py:122-146)
#include "acfm_common.h"

namespace acfm {

constexpr int PTPB = 256;
constexpr int EDT_INF = 1 << 28;

// pass 1: per column, distance (in rows) to the nearest foreground pixel of that column.
__global__ __launch_bounds__(PTPB) void k_edt_cols(const float* __restrict__ mask, int H, int W,
                                                   int* __restrict__ g, int* __restrict__ any_fg) {
  const int n = blockIdx.y, x = blockIdx.x * PTPB + threadIdx.x;
  if (x >= W) return;
  const float* m = mask + (size_t)n * H * W;
  int* go = g + (size_t)n * H * W;
  int d = EDT_INF;
  bool fg = false;
  for (int y = 0; y < H; ++y) {                 // downward sweep
    const bool f = m[(size_t)y * W + x] != 0.0f;  // scipy: zero elements of (1 - mask) are the targets
    const bool target = (1.0f - m[(size_t)y * W + x]) == 0.0f;
    (void)f;
    d = target ? 0 : (d >= EDT_INF ? EDT_INF : d + 1);
    fg |= target;
    go[(size_t)y * W + x] = d;
  }
  d = EDT_INF;
  for (int y = H - 1; y >= 0; --y) {            // upward sweep
    const int cur = go[(size_t)y * W + x];
    d = (cur == 0) ? 0 : (d >= EDT_INF ? EDT_INF : d + 1);
    go[(size_t)y * W + x] = min(cur, d);
  }
  if (fg) any_fg[n] = 1;
}

// pass 2: per row, d2(x) = min_x' (x - x')^2 + g(x')^2 with the row's g staged in LDS (exact
// integer arithmetic), then sqrt in fp64 and one rounding to fp32 like scipy's float64 result
// cast by torch.tensor(...).float() (main.py:371-372).
__global__ __launch_bounds__(PTPB) void k_edt_rows(const int* __restrict__ g,
                                                   const int* __restrict__ any_fg, int H, int W,
                                                   int divisor, float* __restrict__ out) {
  extern __shared__ int s_g[];  // [W]
  const int n = blockIdx.y, y = blockIdx.x;
  const int* gr = g + ((size_t)n * H + y) * W;
  for (int x = threadIdx.x; x < W; x += PTPB) s_g[x] = gr[x];
  __syncthreads();
  const bool has = any_fg[n] != 0;
  for (int x = threadIdx.x; x < W; x += PTPB) {
    long long best;
    if (has) {
      best = (long long)EDT_INF * EDT_INF;
      for (int xp = 0; xp < W; ++xp) {
        const int gg = s_g[xp];
        if (gg >= EDT_INF) continue;
        const long long dx = x - xp;
        const long long v = dx * dx + (long long)gg * gg;
        best = v < best ? v : best;
      }
    } else {
      // no foreground anywhere: scipy 1.6's transform answers as if (row -1, column 0) were one
      best = (long long)(y + 1) * (y + 1) + (long long)x * x;
    }
    out[((size_t)n * H + y) * W + x] = (float)(sqrt((double)best) / (double)divisor);
  }
}

// find_boundaries(mode='thick', connectivity=1) of skimage 0.18.1: grey dilation != grey erosion
// over the 4-neighbourhood cross (ndimage mode='reflect' = edge replication at radius 1), then the
// (row, col) list in row-major order, normalised to [-1,1) and stored as (x, y, valid).
__global__ __launch_bounds__(PTPB) void k_boundaries(const float* __restrict__ mask, int H, int W, int cap,
                                                     float* __restrict__ out, int* __restrict__ counts) {
  __shared__ int s_w[4];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* m = mask + (size_t)n * H * W;
  float* o = out + (size_t)n * cap * 3;
  int total = 0;
  const int npix = H * W;
  for (int base = 0; base < npix; base += PTPB) {
    const int p = base + tid;
    bool b = false;
    int y = 0, x = 0;
    if (p < npix) {
      y = p / W; x = p % W;
      const float c = m[p];
      const float u = m[(size_t)max(y - 1, 0) * W + x], d = m[(size_t)min(y + 1, H - 1) * W + x];
      const float l = m[(size_t)y * W + max(x - 1, 0)], r = m[(size_t)y * W + min(x + 1, W - 1)];
      const float mx = fmaxf(fmaxf(fmaxf(c, u), fmaxf(d, l)), r);
      const float mn = fminf(fminf(fminf(c, u), fminf(d, l)), r);
      b = mx != mn;
    }
    const unsigned long long bal = __ballot(b);
    if (lane == 0) s_w[wv] = __popcll(bal);
    __syncthreads();
    int off = total, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < wv) off += s_w[i];
      tot += s_w[i];
    }
    if (b) {
      const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
      if (pos < cap) {
        o[3 * pos + 0] = (float)(((double)x / (double)W - 0.5) * 2.0);
        o[3 * pos + 1] = (float)(((double)y / (double)H - 0.5) * 2.0);
        o[3 * pos + 2] = 1.0f;
      }
    }
    total += tot;
    __syncthreads();
  }
  // the reference pads the shorter lists with (0, 0) BEFORE normalising: (0/H - 0.5)*2 = -1
  for (int i = min(total, cap) + tid; i < cap; i += PTPB) {
    o[3 * i + 0] = -1.0f; o[3 * i + 1] = -1.0f; o[3 * i + 2] = 0.0f;
  }
  if (tid == 0) counts[n] = total;
}

}  // namespace acfm

using namespace acfm;

extern "C" {

size_t acfm_edt_workspace_bytes(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  return align256(sizeof(int) * (size_t)N * H * W) + align256(sizeof(int) * (size_t)N);
}

int acfm_edt(const float* mask, int N, int H, int W, int divisor, float* out, void* wsp, size_t ws_bytes,
             void* stream) {
  if (!mask || !out || !wsp || N <= 0 || N > 65535 || H <= 0 || W <= 0 || H > 65535 || W > 8192 || divisor < 1)
    return ACFM_E_BADARG;
  if (acfm_edt_workspace_bytes(N, H, W) > ws_bytes) return ACFM_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int* g = (int*)wsp;
  int* any_fg = (int*)((char*)wsp + align256(sizeof(int) * (size_t)N * H * W));
  if (zero_async(any_fg, sizeof(int) * (size_t)N, st) != ACFM_OK) return ACFM_E_LAUNCH;
  hipLaunchKernelGGL(k_edt_cols, dim3((W + PTPB - 1) / PTPB, N), dim3(PTPB), 0, st, mask, H, W, g, any_fg);
  hipLaunchKernelGGL(k_edt_rows, dim3(H, N), dim3(PTPB), sizeof(int) * (size_t)W, st, (const int*)g,
                     (const int*)any_fg, H, W, divisor, out);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

int acfm_boundaries(const float* mask, int N, int H, int W, int cap, float* out, int* counts, void* stream) {
  if (!mask || !out || !counts || N <= 0 || N > 65535 || H <= 0 || W <= 0 || cap <= 0 ||
      (size_t)H * W > 0x7fffffffull)
    return ACFM_E_BADARG;
  hipLaunchKernelGGL(k_boundaries, dim3(N), dim3(PTPB), 0, (hipStream_t)stream, mask, H, W, cap, out, counts);
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}

}  // extern "C"
