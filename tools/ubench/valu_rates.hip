// Issue cost of the VALU instructions the K-nearest insertion is made of, relative to v_fma_f32
// (one wave per SIMD, independent instructions, 64 per loop iteration).  Build + run:
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define KERNEL(name, body)                                                              \
  __global__ __launch_bounds__(64) void name(int iters, unsigned* out) {                \
    unsigned a = threadIdx.x, b = threadIdx.x * 3u + 1u, c = 7u, d = 9u, e = 11u, f = 13u; \
    unsigned long long p = a, q = b;                                                    \
    float x = (float)a, y = 1.5f, z = 0.25f;                                            \
    typedef float v2 __attribute__((ext_vector_type(2)));                               \
    v2 u = {x, y}, v = {y, z}, w = {z, x};                                              \
    for (int i = 0; i < iters; ++i) { R64(body) }                                       \
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + e + f + (unsigned)p + (unsigned)q + \
        (unsigned)x + (unsigned)u.x + (unsigned)u.y;                                    \
  }
KERNEL(k_fma, asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));)
KERNEL(k_pkfma, asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(u) : "v"(v), "v"(w));)
KERNEL(k_pkmul, asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(u) : "v"(v), "v"(w));)
KERNEL(k_cnd, asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a) : "v"(b), "v"(c) : );)
KERNEL(k_cmp32, asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");)
KERNEL(k_cmp64, asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(p), "v"(q) : "vcc");)
KERNEL(k_swap, asm volatile("v_swap_b32 %0, %1" : "+v"(a), "+v"(b));)
KERNEL(k_mov64, asm volatile("v_mov_b64 %0, %1" : "=v"(p) : "v"(q));)
KERNEL(k_min32, asm volatile("v_min_u32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c));)
KERNEL(k_rcp, asm volatile("v_rcp_f32 %0, %1" : "=v"(x) : "v"(y));)
KERNEL(k_exp, asm volatile("v_exp_f32 %0, %1" : "=v"(x) : "v"(y));)
KERNEL(k_cmpcnd, asm volatile("v_cmp_lt_u32 vcc, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a) : "v"(b), "v"(c) : "vcc");)
KERNEL(k_cmpcnd3, asm volatile("v_cmp_lt_u64 vcc, %3, %4\n v_cndmask_b32 %0, %5, %6, vcc\n v_cndmask_b32 %1, %6, %5, vcc\n v_cndmask_b32 %2, %0, %5, vcc" : "=&v"(a), "=&v"(d), "=&v"(e) : "v"(p), "v"(q), "v"(b), "v"(c) : "vcc");)
KERNEL(k_minmax, asm volatile("v_min_u32 %0, %2, %3\n v_max_u32 %1, %2, %3" : "=&v"(a), "=&v"(d) : "v"(b), "v"(c));)
KERNEL(k_add, asm volatile("v_add_f32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(z));)
KERNEL(k_mul, asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x) : "v"(y), "v"(z));)
KERNEL(k_sub_u32, asm volatile("v_sub_u32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c));)
KERNEL(k_ballot, asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(p) : "v"(a), "v"(b));)

static int g_wps = 1;   // waves per SIMD
template <class K>
static double run(K k, const char* name, double base) {
  unsigned* out;
  hipMalloc(&out, 1024 * 64 * 4 * g_wps);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(1024 * g_wps), dim3(64), 0, 0, 100, out);   // warm-up; 1024 waves = one per SIMD
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(1024 * g_wps), dim3(64), 0, 0, iters, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double ns = ms * 1e6 / (iters * 64.0 * g_wps);   // per instruction and SIMD
  printf("%-10s %7.3f ns/inst  (%.2f x v_fma_f32)\n", name, ns, base > 0 ? ns / base : 1.0);
  hipFree(out);
  return ns;
}
int main(int argc, char** argv) {
  if (argc > 1) g_wps = atoi(argv[1]);
  printf("waves per SIMD: %d\n", g_wps);
  const double b = run(k_fma, "v_fma_f32", 0);
  run(k_pkfma, "v_pk_fma", b); run(k_pkmul, "v_pk_mul", b); run(k_cnd, "v_cndmask", b);
  run(k_cmp32, "cmp_lt_u32", b); run(k_cmp64, "cmp_lt_u64", b); run(k_swap, "v_swap_b32", b);
  run(k_mov64, "v_mov_b64", b); run(k_min32, "v_min_u32", b); run(k_rcp, "v_rcp_f32", b);
  run(k_exp, "v_exp_f32", b); run(k_ballot, "cmp->sgpr", b);
  run(k_cmpcnd, "cmp+cnd", b); run(k_cmpcnd3, "cmp64+3cnd", b); run(k_minmax, "min+max", b);
  run(k_add, "v_add_f32", b); run(k_mul, "v_mul_f32", b); run(k_sub_u32, "v_sub_u32", b);
  return 0;
}
