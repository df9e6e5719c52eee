// What does one wave-instruction cost the SIMD it is issued on?  (the fused solver kernels are priced in "issue
// slots"; the microarchitecture guide gives 4 cycles for v_add / v_fma and 8 for the transcendental unit -- this lab adds the
// instructions the exact-division and lane-shift code is made of.)  Every kernel repeats ONE instruction on 16 independent register
// sets inside a loop, four waves per SIMD, all CUs; cycles = elapsed x clock / instructions per SIMD, with the clock taken from
// s_memrealtime-free arithmetic: a v_add_f32 run in the same process is the 4-cycle yardstick, and everything is reported relative to it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

// one kernel per instruction: BODY(i) is an asm statement on register set i
#define KERNEL(NAME, DECL, BODY, SINK)                                                      \
  __global__ __launch_bounds__(256) void NAME(float* out, int iters, float seed)            \
  {                                                                                          \
    DECL                                                                                     \
    for (int it = 0; it < iters; ++it) { REP16(BODY) }                                      \
    float s = 0.f;                                                                           \
    SINK                                                                                     \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                                 \
  }

#define DECL_F float a[16]; float b = seed * 0.999f, c = seed * 1e-3f; for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x * 1e-3f;
#define SINK_F for (int i = 0; i < 16; ++i) s += a[i];
#define DECL_D double d[16]; float a[16]; float b = seed; const double r = 1.0 / (seed * 3.3); for (int i = 0; i < 16; ++i) { a[i] = seed + i; d[i] = a[i] * 1.5; }
#define SINK_D for (int i = 0; i < 16; ++i) s += a[i] + static_cast<float>(d[i]);

#define B_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define B_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define B_RSQ(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
#define B_SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
#define B_DSCALE(i) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(b) : "vcc");
#define B_DFMAS(i) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
#define B_DFIXUP(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
#define B_DPP(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
#define B_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
#define B_READLANE(i) { unsigned sx; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sx) : "v"(a[i])); asm volatile("" :: "s"(sx)); }
#define B_MIN3U(i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define B_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 1, -1" : "+v"(a[i]));
#define B_CMP(i) asm volatile("v_cmp_gt_u32 vcc, %0, %1" :: "v"(a[i]), "v"(b) : "vcc");
#define B_CVT_D_F(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
#define B_CVT_F_D(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
#define B_MUL_D(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(r));
#define B_FMA_D(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(r));
#define B_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(r));
#define B_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(r));
#define B_SNOP(i) asm volatile("s_nop 0");
#define B_SNOP4(i) asm volatile("s_nop 4");

KERNEL(k_add, DECL_F, B_ADD, SINK_F)
KERNEL(k_mul, DECL_F, B_MUL, SINK_F)
KERNEL(k_fma, DECL_F, B_FMA, SINK_F)
KERNEL(k_rcp, DECL_F, B_RCP, SINK_F)
KERNEL(k_rsq, DECL_F, B_RSQ, SINK_F)
KERNEL(k_sqrt, DECL_F, B_SQRT, SINK_F)
KERNEL(k_dscale, DECL_F, B_DSCALE, SINK_F)
KERNEL(k_dfmas, DECL_F, B_DFMAS, SINK_F)
KERNEL(k_dfixup, DECL_F, B_DFIXUP, SINK_F)
KERNEL(k_mov, DECL_F, B_MOV, SINK_F)
KERNEL(k_dpp, DECL_F, B_DPP, SINK_F)
KERNEL(k_cndmask, DECL_F, B_CNDMASK, SINK_F)
KERNEL(k_readlane, DECL_F, B_READLANE, SINK_F)
KERNEL(k_min3u, DECL_F, B_MIN3U, SINK_F)
KERNEL(k_lshladd, DECL_F, B_LSHLADD, SINK_F)
KERNEL(k_cmp, DECL_F, B_CMP, SINK_F)
KERNEL(k_cvt_d_f, DECL_D, B_CVT_D_F, SINK_D)
KERNEL(k_cvt_f_d, DECL_D, B_CVT_F_D, SINK_D)
KERNEL(k_mul_d, DECL_D, B_MUL_D, SINK_D)
KERNEL(k_fma_d, DECL_D, B_FMA_D, SINK_D)
KERNEL(k_pkfma, DECL_D, B_PKFMA, SINK_D)
KERNEL(k_pkmul, DECL_D, B_PKMUL, SINK_D)
KERNEL(k_snop, DECL_F, B_SNOP, SINK_F)
KERNEL(k_snop4, DECL_F, B_SNOP4, SINK_F)

typedef void (*kern_t)(float*, int, float);
static float time_kernel(kern_t k, int waves_per_simd, float* out)
{
  const int blocks = 256 * waves_per_simd, iters = 4000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 50, 1.0f);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  // wave-instructions per SIMD = waves_per_simd x iters x 16; ns per instruction
  return best * 1e6f / (static_cast<float>(waves_per_simd) * iters * 16.f);
}

int main()
{
  float* out; CK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
  struct { const char* name; kern_t k; } ks[] = {
    {"v_add_f32", k_add}, {"v_mul_f32", k_mul}, {"v_fma_f32", k_fma}, {"v_mov_b32", k_mov}, {"v_cndmask_b32", k_cndmask},
    {"v_mov_b32_dpp wave_shr", k_dpp}, {"v_readlane_b32", k_readlane}, {"v_min3_u32", k_min3u}, {"v_lshl_add_u32", k_lshladd},
    {"v_cmp_gt_u32", k_cmp}, {"v_rcp_f32", k_rcp}, {"v_rsq_f32", k_rsq}, {"v_sqrt_f32", k_sqrt}, {"v_div_scale_f32", k_dscale},
    {"v_div_fmas_f32", k_dfmas}, {"v_div_fixup_f32", k_dfixup}, {"v_cvt_f64_f32", k_cvt_d_f}, {"v_cvt_f32_f64", k_cvt_f_d},
    {"v_mul_f64", k_mul_d}, {"v_fma_f64", k_fma_d}, {"v_pk_fma_f32", k_pkfma}, {"v_pk_mul_f32", k_pkmul}, {"s_nop 0", k_snop}, {"s_nop 4", k_snop4}};
  for (int w : {4, 1}) {
    const float yard = time_kernel(k_add, w, out);
    std::printf("waves per SIMD = %d: v_add_f32 %.3f ns per wave-instruction per SIMD (= 4 cycles => %.2f GHz)\n", w, yard, 4.f / yard);
    for (auto& k : ks) {
      const float ns = time_kernel(k.k, w, out);
      std::printf("  %-24s %7.3f ns  = %5.2f cycles (v_add = 4)\n", k.name, ns, 4.f * ns / yard);
    }
  }
  return 0;
}
