// Latency of dependent fp64 operations on one wave (the serial chain of the in-tile Cholesky, acfm_solve.hip):
//   hipcc -O3 --offload-arch=gfx950 f64_chain.hip -o f64_chain && ./f64_chain
// Each test runs ITER dependent steps in one wave of one workgroup and reports ns per step from the 100 MHz
// wall clock; "x4" rows run four independent chains interleaved (issue rate rather than latency).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITER = 4096;

template <int MODE>
__global__ void k(double* out, long long* ticks, double a, double b) {
  double x = a + threadIdx.x * 1e-9, y = b, z = a * 0.5, w2 = b * 0.25;
  long long t0 = wall_clock64();
#pragma unroll 16
  for (int i = 0; i < ITER; ++i) {
    if (MODE == 0) x = __builtin_fma(x, y, b);                                   // dependent fma
    if (MODE == 1) { x = __builtin_fma(x, y, b); y = __builtin_fma(y, a, b); z = __builtin_fma(z, a, b); w2 = __builtin_fma(w2, a, b); }
    if (MODE == 2) x = __builtin_amdgcn_rsq(x) + 1.5;                            // rsq + add
    if (MODE == 3) {                                                             // readlane round trip + fma
      int lo = __builtin_amdgcn_readlane(__double2loint(x), 7), hi = __builtin_amdgcn_readlane(__double2hiint(x), 7);
      x = __builtin_fma(__hiloint2double(hi, lo), y, x);
    }
    if (MODE == 4) {  // one pivot of the chain: bcast, rsq, correction, multiply, bcast, fma
      int lo = __builtin_amdgcn_readlane(__double2loint(x), 7), hi = __builtin_amdgcn_readlane(__double2hiint(x), 7);
      const double piv = __hiloint2double(hi, lo);
      const double y0 = __builtin_amdgcn_rsq(piv);
      const double s0 = x * y0;
      const double e = 0.5 - (0.5 * piv) * y0 * y0;
      const double h = e * (1.0 + 1.5 * e);
      const double v = s0 + s0 * h;
      int lo2 = __builtin_amdgcn_readlane(__double2loint(v), 9), hi2 = __builtin_amdgcn_readlane(__double2hiint(v), 9);
      x = z - v * __hiloint2double(hi2, lo2) + 2.0;
    }
    if (MODE == 5) x = x * y;                                                    // dependent mul
    if (MODE == 6) { float f = (float)x; f = __builtin_fmaf(f, 1.0001f, 0.5f); x = f; }  // cvt round trip
  }
  long long t1 = wall_clock64();
  out[threadIdx.x] = x + y + z + w2;
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
}

int main() {
  double* out; long long* ticks;
  hipMalloc(&out, 64 * 8); hipMalloc(&ticks, 8);
  const char* names[] = {"dependent v_fma_f64", "4 independent v_fma_f64 chains (per 4)", "v_rsq_f64 + v_add_f64", "2 v_readlane + v_fma_f64",
                         "one pivot of the chain", "dependent v_mul_f64", "cvt f64->f32, fma f32, cvt back"};
  for (int m = 0; m < 7; ++m) {
    long long best = 1ll << 60;
    for (int rep = 0; rep < 5; ++rep) {
      switch (m) {
        case 0: k<0><<<1, 64>>>(out, ticks, 1.0000001, 0.9999999); break;
        case 1: k<1><<<1, 64>>>(out, ticks, 1.0000001, 0.9999999); break;
        case 2: k<2><<<1, 64>>>(out, ticks, 1.0000001, 0.9999999); break;
        case 3: k<3><<<1, 64>>>(out, ticks, 1.0000001, 1e-9); break;
        case 4: k<4><<<1, 64>>>(out, ticks, 1.0000001, 0.9999999); break;
        case 5: k<5><<<1, 64>>>(out, ticks, 1.0000001, 0.9999999); break;
        case 6: k<6><<<1, 64>>>(out, ticks, 1.0000001, 0.9999999); break;
      }
      hipDeviceSynchronize();
      long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
      if (t < best) best = t;
    }
    printf("%-44s %7.2f ns per step\n", names[m], best * 10.0 / ITER);
  }
  return 0;
}
