// VALU issue rates on gfx950, per SIMD, with INDEPENDENT instruction streams inside every wave
// (eight destination registers / eight SGPR-pair masks in rotation: no instruction waits for the
// previous one, and compares do not funnel through VCC) at 1, 2, 4 and 8 waves per SIMD.
// This is the calibration of bench.py's `roofline.valu.peak`: the rate of the instruction mix
// the raster kernels are made of, not a nominal "4 cycles per wave64 instruction".
//
// Build + run (on the GPU box):
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
// Output columns: ns per wave-instruction and SIMD (1024 SIMDs), the same in shader cycles
// (s_memtime ticks of the waves themselves), and G wave-instructions / s for the whole chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

#define R8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define REP4(x) x x x x
// one loop iteration = 8 streams x 4 = 32 instructions (or instruction groups)
#define BODY(M) REP4(R8(M))

struct St {
  float x[8], y, z;
  unsigned a[8], b, c;
  u64 p[8], q, m[8];
  v2 u[8], v, w;
};

#define KERNEL(name, M)                                                                      \
  __global__ __launch_bounds__(256) void name(int iters, unsigned* out, u64* cyc) {          \
    St s;                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                          \
      s.x[i] = (float)(threadIdx.x + i); s.a[i] = threadIdx.x * 7u + i;                      \
      s.p[i] = ((u64)threadIdx.x << 32) | (unsigned)i; s.m[i] = 0x5555555555555555ull << (i & 1); \
      s.u[i] = (v2){(float)i, 1.0f};                                                         \
    }                                                                                        \
    s.y = 1.0000001f; s.z = 0.25f; s.b = threadIdx.x * 3u + 1u; s.c = 77u;                   \
    s.q = ((u64)(threadIdx.x ^ 5u) << 32) | 3u; s.v = (v2){1.0000001f, 0.999f}; s.w = (v2){0.5f, 0.25f}; \
    const u64 t0 = __builtin_readcyclecounter();                                             \
    for (int it = 0; it < iters; ++it) { BODY(M) }                                           \
    const u64 t1 = __builtin_readcyclecounter();                                             \
    unsigned acc = 0;                                                                        \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                            \
      acc += (unsigned)s.x[i] + s.a[i] + (unsigned)s.p[i] + (unsigned)s.u[i].x + (unsigned)s.u[i].y + (unsigned)__popcll(s.m[i]); \
    out[blockIdx.x * 256 + threadIdx.x] = acc;                                               \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;         \
  }

#define M_FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s.x[i]) : "v"(s.y), "v"(s.z));
#define M_ADD(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(s.x[i]) : "v"(s.y));
#define M_MUL(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(s.x[i]) : "v"(s.y));
#define M_SUB(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s.x[i]) : "v"(s.z));
#define M_MAX(i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(s.x[i]) : "v"(s.y));
#define M_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(s.x[i]) : "v"(s.z), "v"(s.y));
#define M_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(s.u[i]) : "v"(s.v), "v"(s.w));
#define M_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(s.u[i]) : "v"(s.v));
#define M_PKADD(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(s.u[i]) : "v"(s.w));
#define M_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(s.a[i]) : "v"(s.b), "s"(s.m[i]));
#define M_CMP32(i) asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(s.m[i]) : "v"(s.a[i]), "v"(s.b));
#define M_CMP64(i) asm volatile("v_cmp_lt_u64 %0, %1, %2" : "=s"(s.m[i]) : "v"(s.p[i]), "v"(s.q));
#define M_CMPF(i) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(s.m[i]) : "v"(s.x[i]), "v"(s.y));
#define M_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(s.a[i]) : "v"(s.b));
#define M_MOV64(i) asm volatile("v_mov_b64 %0, %1" : "=v"(s.p[i]) : "v"(s.q));
#define M_ADDU(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(s.a[i]) : "v"(s.b));
#define M_AND(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(s.a[i]) : "v"(s.b));
#define M_LSHL(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(s.a[i]));
#define M_MINU(i) asm volatile("v_min_u32 %0, %1, %0" : "+v"(s.a[i]) : "v"(s.b));
#define M_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(s.x[i]));
#define M_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(s.x[i]));
#define M_SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(s.x[i]));
#define M_SWAP(i) asm volatile("v_swap_b32 %0, %1" : "+v"(s.a[i]), "+v"(s.a[(i + 1) & 7]));
// DPP row shift add (the backward's row sums)
#define M_DPP(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(s.x[i]));
// the sorted insertion's slot: one 64-bit compare + six selects (key lo/hi of the slot and of the
// travelling element, the two blend factors); counted as 7 instructions
#define M_SLOT(i)                                                                                     \
  asm volatile("v_cmp_lt_u64 %0, %3, %4\n v_cndmask_b32 %1, %1, %5, %0\n v_cndmask_b32 %2, %2, %6, %0\n" \
               "v_cndmask_b32 %1, %1, %6, %0\n v_cndmask_b32 %2, %2, %5, %0\n"                         \
               "v_cndmask_b32 %1, %1, %5, %0\n v_cndmask_b32 %2, %2, %6, %0"                            \
               : "=&s"(s.m[i]), "+v"(s.a[i]), "+v"(s.a[(i + 4) & 7]) : "v"(s.p[i]), "v"(s.q), "v"(s.b), "v"(s.c));
// fp32 arithmetic mix of the walk: 2 sub, 2 mul, 1 sub (an edge function), 5 instructions
#define M_EDGE(i)                                                                                     \
  asm volatile("v_sub_f32 %0, %0, %1\n v_mul_f32 %0, %0, %2\n v_sub_f32 %0, %0, %1\n v_mul_f32 %0, %0, %2\n v_sub_f32 %0, %0, %1" \
               : "+v"(s.x[i]) : "v"(s.z), "v"(s.y));

KERNEL(k_fma, M_FMA) KERNEL(k_add, M_ADD) KERNEL(k_mul, M_MUL) KERNEL(k_sub, M_SUB) KERNEL(k_max, M_MAX)
KERNEL(k_med3, M_MED3) KERNEL(k_pkfma, M_PKFMA) KERNEL(k_pkmul, M_PKMUL) KERNEL(k_pkadd, M_PKADD)
KERNEL(k_cnd, M_CND) KERNEL(k_cmp32, M_CMP32) KERNEL(k_cmp64, M_CMP64) KERNEL(k_cmpf, M_CMPF)
KERNEL(k_mov, M_MOV) KERNEL(k_mov64, M_MOV64) KERNEL(k_addu, M_ADDU) KERNEL(k_and, M_AND) KERNEL(k_lshl, M_LSHL)
KERNEL(k_minu, M_MINU) KERNEL(k_rcp, M_RCP) KERNEL(k_exp, M_EXP) KERNEL(k_sqrt, M_SQRT) KERNEL(k_swap, M_SWAP)
KERNEL(k_dpp, M_DPP) KERNEL(k_slot, M_SLOT) KERNEL(k_edge, M_EDGE)

typedef void (*kern_t)(int, unsigned*, u64*);
struct Row { const char* name; kern_t k; int per; };   // per = instructions per macro expansion

int main(int argc, char** argv) {
  const Row rows[] = {
      {"v_fma_f32", k_fma, 1}, {"v_add_f32", k_add, 1}, {"v_mul_f32", k_mul, 1}, {"v_sub_f32", k_sub, 1},
      {"v_max_f32", k_max, 1}, {"v_med3_f32", k_med3, 1}, {"v_pk_fma_f32", k_pkfma, 1}, {"v_pk_mul_f32", k_pkmul, 1},
      {"v_pk_add_f32", k_pkadd, 1}, {"v_cndmask(sgpr)", k_cnd, 1}, {"v_cmp_lt_u32->s", k_cmp32, 1},
      {"v_cmp_lt_u64->s", k_cmp64, 1}, {"v_cmp_lt_f32->s", k_cmpf, 1}, {"v_mov_b32", k_mov, 1}, {"v_mov_b64", k_mov64, 1},
      {"v_add_u32", k_addu, 1}, {"v_and_b32", k_and, 1}, {"v_lshlrev_b32", k_lshl, 1}, {"v_min_u32", k_minu, 1},
      {"v_rcp_f32", k_rcp, 1}, {"v_exp_f32", k_exp, 1}, {"v_sqrt_f32", k_sqrt, 1}, {"v_swap_b32", k_swap, 1},
      {"v_add_f32 dpp", k_dpp, 1}, {"insert slot (cmp64+6cnd)", k_slot, 7}, {"edge fn (2sub 2mul sub)", k_edge, 5}};
  const int nrows = sizeof(rows) / sizeof(rows[0]);
  const int wps_list[] = {1, 2, 4, 8};
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  printf("device %s, %d CUs (%d SIMDs), nominal clock %d MHz\n", pr.gcnArchName, cus, 4 * cus, pr.clockRate / 1000);
  printf("independent streams (8 destination registers / 8 SGPR masks in rotation), 256-thread workgroups = one wave per SIMD of a CU\n");
  printf("%-26s", "instruction");
  for (int w : wps_list) printf(" | %d w/SIMD: ns/inst  cyc/inst  Ginst/s", w);
  printf("\n");
  unsigned* out; u64* cyc;
  hipMalloc(&out, sizeof(unsigned) * 256 * cus * 8);
  hipMalloc(&cyc, sizeof(u64) * 4 * cus * 8);
  std::vector<u64> h(4 * cus * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  for (int r = 0; r < nrows; ++r) {
    printf("%-26s", rows[r].name);
    for (int wps : wps_list) {
      const int grid = cus * wps;
      hipLaunchKernelGGL(rows[r].k, dim3(grid), dim3(256), 0, 0, 50, out, cyc);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(rows[r].k, dim3(grid), dim3(256), 0, 0, iters, out, cyc);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h.data(), cyc, sizeof(u64) * 4 * grid, hipMemcpyDeviceToHost);
      double csum = 0;
      for (int i = 0; i < 4 * grid; ++i) csum += (double)h[i];
      const double per_simd = (double)iters * 32.0 * rows[r].per * wps;       // wave-instructions issued on one SIMD
      const double ns = ms * 1e6 / per_simd;
      const double cy = csum / (4.0 * grid) / per_simd;                        // mean wave lifetime in s_memtime ticks / instructions on its SIMD
      const double ginst = per_simd * 4.0 * cus / (ms * 1e-3) / 1e9;
      printf(" |            %6.3f   %6.2f   %7.1f", ns, cy, ginst);
    }
    printf("\n");
  }
  return 0;
}
