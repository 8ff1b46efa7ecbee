// Correlation cost volume of the frozen optical-flow network, forward
// only: the one native extension in the reference tree (multiframe/data/optical_flow/model/
// correlation_package/correlation_cuda_kernel.cu:73-147, called by MaskFlownet.py:116, 416 with
// pad_size = max_displacement = md, kernel_size = 1, stride1 = stride2 = 1):
//   out[n, (tj+md)(2md+1) + (ti+md), y, x] = 1/C * sum_c f1[n,c,y,x] * f2[n,c,y+tj,x+ti]   (zero outside)
// The reference first copies both inputs into zero-padded NHWC buffers and gives every output
// pixel a 32-thread block that loops over the (2md+1)^2 displacements with a shuffle reduction
// each.  Here a 256-thread workgroup owns 16x16 output pixels: the f2 halo tile of 8 channels at
// a time goes through LDS (zeros outside the image: no padded copies) and every thread keeps all
// (2md+1)^2 accumulators of its pixel in registers (~120 VGPRs at md = 4: 4 waves per SIMD; two
// pixels per thread halve the LDS reads per product but need 256 VGPRs and ran 1.5x slower).
// Small feature maps (the coarse pyramid levels) do not fill the chip with tiles alone: the channel
// range is then split over several workgroups that add their partial sums with float atomics.
// One LDS read per product makes this LDS-bandwidth-bound (~7 TFLOP/s); a banded-GEMM formulation
// on the matrix cores is the next step if the flow network is ever run on-box (it is not in ACFM:
// flows are precomputed and the network's weights are not in the repository).
#include "acfm_common.h"

namespace acfm {

constexpr int CT = 16;   // output tile edge
constexpr int CCH = 8;   // channels per LDS chunk

template <int MD, bool ATOMIC>
__global__ __launch_bounds__(256) void k_correlation_fwd(const float* __restrict__ f1, const float* __restrict__ f2,
                                                         int C, int H, int W, int csplit, float* __restrict__ out) {
  constexpr int D1 = 2 * MD + 1, TW = CT + 2 * MD;
  __shared__ float s2[CCH][TW][TW + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int x0 = blockIdx.x * CT, y0 = blockIdx.y * CT;
  const int n = blockIdx.z / csplit, part = blockIdx.z % csplit;
  const int cper = ((C + csplit - 1) / csplit + CCH - 1) / CCH * CCH;
  const int c_lo = part * cper, c_hi = min(C, c_lo + cper);
  const int x = x0 + tx, y = y0 + ty;
  const size_t HW = (size_t)H * W;
  const float* b1 = f1 + (size_t)n * C * HW;
  const float* b2 = f2 + (size_t)n * C * HW;
  float acc[D1 * D1];
#pragma unroll
  for (int d = 0; d < D1 * D1; ++d) acc[d] = 0.f;
  for (int c0 = c_lo; c0 < c_hi; c0 += CCH) {
    __syncthreads();
    for (int i = threadIdx.x; i < CCH * TW * TW; i += 256) {
      const int cc = i / (TW * TW), r = i % (TW * TW), hy = r / TW, hx = r % TW;
      const int gy = y0 + hy - MD, gx = x0 + hx - MD, c = c0 + cc;
      s2[cc][hy][hx] = (c < c_hi && gy >= 0 && gy < H && gx >= 0 && gx < W) ? b2[(size_t)c * HW + (size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    float a[CCH];
#pragma unroll
    for (int cc = 0; cc < CCH; ++cc)
      a[cc] = (c0 + cc < c_hi && y < H && x < W) ? b1[(size_t)(c0 + cc) * HW + (size_t)y * W + x] : 0.f;
#pragma unroll
    for (int cc = 0; cc < CCH; ++cc)
#pragma unroll
      for (int tj = 0; tj < D1; ++tj) {
        const float* row = &s2[cc][ty + tj][tx];
#pragma unroll
        for (int ti = 0; ti < D1; ++ti) acc[tj * D1 + ti] += a[cc] * row[ti];
      }
  }
  if (y >= H || x >= W) return;
  const float inv = 1.0f / (float)C;
  float* o = out + (size_t)n * D1 * D1 * HW + (size_t)y * W + x;
#pragma unroll
  for (int d = 0; d < D1 * D1; ++d) {
    if (ATOMIC) atomicAdd(&o[(size_t)d * HW], acc[d] * inv);
    else o[(size_t)d * HW] = acc[d] * inv;
  }
}

}  // namespace acfm

using namespace acfm;

template <int MD>
static void launch_corr(const float* f1, const float* f2, int N, int C, int H, int W, float* out, int csplit,
                        hipStream_t st) {
  const dim3 grid((W + CT - 1) / CT, (H + CT - 1) / CT, N * csplit);
  if (csplit > 1)
    hipLaunchKernelGGL((k_correlation_fwd<MD, true>), grid, dim3(256), 0, st, f1, f2, C, H, W, csplit, out);
  else
    hipLaunchKernelGGL((k_correlation_fwd<MD, false>), grid, dim3(256), 0, st, f1, f2, C, H, W, csplit, out);
}

extern "C" int acfm_correlation_forward(const float* f1, const float* f2, int N, int C, int H, int W, int md,
                                        float* out, void* stream) {
  if (!f1 || !f2 || !out || N <= 0 || N > 4096 || C <= 0 || H <= 0 || W <= 0 || md < 1 || md > 4) return ACFM_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  // enough workgroups for 256 CUs: split the channels when the tiles alone are few
  const long tiles = (long)((W + CT - 1) / CT) * ((H + CT - 1) / CT) * N;
  int csplit = (int)((768 + tiles - 1) / tiles);
  const int max_split = (C + CCH - 1) / CCH;
  if (csplit > max_split) csplit = max_split;
  if (csplit < 1) csplit = 1;
  if (csplit > 1 &&
      zero_async(out, sizeof(float) * (size_t)N * (2 * md + 1) * (2 * md + 1) * H * W, st) != ACFM_OK)
    return ACFM_E_LAUNCH;
  switch (md) {
    case 1: launch_corr<1>(f1, f2, N, C, H, W, out, csplit, st); break;
    case 2: launch_corr<2>(f1, f2, N, C, H, W, out, csplit, st); break;
    case 3: launch_corr<3>(f1, f2, N, C, H, W, out, csplit, st); break;
    default: launch_corr<4>(f1, f2, N, C, H, W, out, csplit, st); break;
  }
  ACFM_CHECK_LAUNCH();
  return ACFM_OK;
}
