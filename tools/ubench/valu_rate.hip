// Issue rate of single VALU instructions on gfx950: cycles per wave64 instruction and SIMD, from a kernel that runs one
// instruction 8 x 4096 times per wavefront in 8 independent chains, 8 wavefronts per SIMD on every CU.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL(NAME, ASMSTR)                                                                                     \
  __global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned seed) {                                     \
    unsigned r0 = threadIdx.x + seed, r1 = r0 * 3u, r2 = r0 * 5u, r3 = r0 * 7u, r4 = r0 * 11u, r5 = r0 * 13u, r6 = r0 * 17u, r7 = r0 * 19u; \
    unsigned a = seed | 0x3f800000u, b = (seed * 3u) | 0x3f000000u;                                              \
    for (int i = 0; i < 4096; ++i) {                                                                             \
      asm volatile(ASMSTR(0) ASMSTR(1) ASMSTR(2) ASMSTR(3) ASMSTR(4) ASMSTR(5) ASMSTR(6) ASMSTR(7)               \
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)              \
                   : "v"(a), "v"(b) : "vcc", "s20", "s21");                                                         \
    }                                                                                                            \
    out[blockIdx.x * 256 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                                 \
  }
#define S_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define S_FMAC(k) "v_fmac_f32 %" #k ", %8, %9\n"
#define S_ADD(k) "v_add_f32 %" #k ", %" #k ", %8\n"
#define S_MUL24(k) "v_mul_i32_i24 %" #k ", %" #k ", %8\n"
#define S_MAD24(k) "v_mad_i32_i24 %" #k ", %8, %9, %" #k "\n"
#define S_MADU24(k) "v_mad_u32_u24 %" #k ", %8, %9, %" #k "\n"
#define S_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n"
#define S_SUBSDWA(k) "v_sub_u32_sdwa %" #k ", %8, %" #k " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define S_CVTSDWA(k) "v_cvt_f32_u32_sdwa %" #k ", %" #k " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
#define S_CVT(k) "v_cvt_f32_u32 %" #k ", %" #k "\n"
#define S_DOT2(k) "v_dot2_i32_i16 %" #k ", %8, %9, %" #k "\n"
#define S_DOT2U(k) "v_dot2_u32_u16 %" #k ", %8, %9, %" #k "\n"
#define S_PKSUB16(k) "v_pk_sub_i16 %" #k ", %" #k ", %8\n"
#define S_MED3(k) "v_med3_u32 %" #k ", %" #k ", %8, %9\n"
#define S_MIN(k) "v_min_u32 %" #k ", %" #k ", %8\n"
#define S_ANDOR(k) "v_and_or_b32 %" #k ", %" #k ", %8, %9\n"
#define S_LSHLOR(k) "v_lshl_or_b32 %" #k ", %" #k ", 1, %9\n"
#define S_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define S_CNDMASK64(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[20:21]\n"
#define S_CMP(k) "v_cmp_eq_u32 vcc, %" #k ", %8\n"
#define S_CMP64(k) "v_cmp_eq_u32_e64 s[20:21], %" #k ", %8\n"
#define S_CMPCND(k) "v_cmp_eq_u32 vcc, %" #k ", %8\ns_nop 1\nv_cndmask_b32 %" #k ", %" #k ", %9, vcc\n"
#define S_MOV(k) "v_mov_b32 %" #k ", %8\n"
#define S_XOR(k) "v_xor_b32 %" #k ", %" #k ", %8\n"
#define S_ADDU(k) "v_add_u32 %" #k ", %" #k ", %8\n"
#define S_LSHLADD64(k) ""
#define S_RCP(k) "v_rcp_f32 %" #k ", %" #k "\n"
#define S_FLOOR(k) "v_floor_f32 %" #k ", %" #k "\n"
#define S_CVTI(k) "v_cvt_i32_f32 %" #k ", %" #k "\n"
#define S_MAX3F(k) "v_max3_f32 %" #k ", %" #k ", %8, %9\n"
#define S_MULF(k) "v_mul_f32 %" #k ", %" #k ", %8\n"
#define S_SUBF(k) "v_sub_f32 %" #k ", %8, %" #k "\n"
#define S_READLANE(k) ""
#define S_SQRT(k) "v_sqrt_f32 %" #k ", %" #k "\n"
#define S_ADD3(k) "v_add3_u32 %" #k ", %" #k ", %8, %9\n"
#define S_SAD(k) "v_sad_u32 %" #k ", %8, %9, %" #k "\n"
#define S_PERM(k) "v_perm_b32 %" #k ", %" #k ", %8, %9\n"
#define S_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 16, 16\n"
#define S_MADU16(k) "v_mad_u16 %" #k ", %8, %9, %" #k "\n"
#define S_FMAMIX(k) "v_fma_mix_f32 %" #k ", %8, %9, %" #k " op_sel_hi:[1,1,0]\n"
#define S_ADDF64(k) ""
KERNEL(k_fma, S_FMA)
KERNEL(k_fmac, S_FMAC)
KERNEL(k_add, S_ADD)
KERNEL(k_mul24, S_MUL24)
KERNEL(k_mad24, S_MAD24)
KERNEL(k_madu24, S_MADU24)
KERNEL(k_mullo, S_MULLO)
KERNEL(k_subsdwa, S_SUBSDWA)
KERNEL(k_cvtsdwa, S_CVTSDWA)
KERNEL(k_cvt, S_CVT)
KERNEL(k_dot2, S_DOT2)
KERNEL(k_dot2u, S_DOT2U)
KERNEL(k_pksub16, S_PKSUB16)
KERNEL(k_med3, S_MED3)
KERNEL(k_min, S_MIN)
KERNEL(k_andor, S_ANDOR)
KERNEL(k_lshlor, S_LSHLOR)
KERNEL(k_cndmask, S_CNDMASK)
KERNEL(k_cndmask64, S_CNDMASK64)
KERNEL(k_cmp, S_CMP)
KERNEL(k_cmp64, S_CMP64)
KERNEL(k_cmpcnd, S_CMPCND)
KERNEL(k_mov, S_MOV)
KERNEL(k_xor, S_XOR)
KERNEL(k_addu, S_ADDU)
KERNEL(k_rcp, S_RCP)
KERNEL(k_floor, S_FLOOR)
KERNEL(k_cvti, S_CVTI)
KERNEL(k_max3f, S_MAX3F)
KERNEL(k_mulf, S_MULF)
KERNEL(k_subf, S_SUBF)
KERNEL(k_sqrt, S_SQRT)
KERNEL(k_add3, S_ADD3)
KERNEL(k_sad, S_SAD)
KERNEL(k_perm, S_PERM)
KERNEL(k_bfe, S_BFE)
KERNEL(k_fmamix, S_FMAMIX)
// packed float and double: 64-bit register pairs
__global__ __launch_bounds__(256) void k_pkfma(unsigned* out, unsigned seed) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 r[8];
  for (int k = 0; k < 8; ++k) r[k] = f2{(float)(threadIdx.x + k), (float)seed};
  const f2 a = {1.0001f, 0.9999f}, b = {0.5f, 0.25f};
  for (int i = 0; i < 4096; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b));
  }
  float s = 0;
  for (int k = 0; k < 8; ++k) s += r[k].x + r[k].y;
  out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(s);
}
__global__ __launch_bounds__(256) void k_addf64(unsigned* out, unsigned seed) {
  double r[8];
  for (int k = 0; k < 8; ++k) r[k] = (double)(threadIdx.x + k + seed);
  const double a = 1.0001;
  for (int i = 0; i < 4096; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(r[k]) : "v"(a));
  }
  double s = 0;
  for (int k = 0; k < 8; ++k) s += r[k];
  out[blockIdx.x * 256 + threadIdx.x] = (unsigned)(long long)s;
}
__global__ __launch_bounds__(256) void k_fmaf64(unsigned* out, unsigned seed) {
  double r[8];
  for (int k = 0; k < 8; ++k) r[k] = (double)(threadIdx.x + k + seed);
  const double a = 1.0001, b = 0.5;
  for (int i = 0; i < 4096; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b));
  }
  double s = 0;
  for (int k = 0; k < 8; ++k) s += r[k];
  out[blockIdx.x * 256 + threadIdx.x] = (unsigned)(long long)s;
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const int blocks = cus * 8;  // 8 blocks x 4 waves per CU = 8 waves per SIMD
  unsigned* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  struct K { const char* name; void (*f)(unsigned*, unsigned); };
  const K ks[] = {{"v_fma_f32", k_fma}, {"v_fmac_f32", k_fmac}, {"v_add_f32", k_add}, {"v_pk_fma_f32", k_pkfma}, {"v_add_f64", k_addf64}, {"v_fma_f64", k_fmaf64},
                  {"v_mul_i32_i24", k_mul24}, {"v_mad_i32_i24", k_mad24}, {"v_mad_u32_u24", k_madu24}, {"v_mul_lo_u32", k_mullo}, {"v_sub_u32_sdwa", k_subsdwa},
                  {"v_cvt_f32_u32_sdwa", k_cvtsdwa}, {"v_cvt_f32_u32", k_cvt}, {"v_dot2_i32_i16", k_dot2}, {"v_dot2_u32_u16", k_dot2u}, {"v_pk_sub_i16", k_pksub16},
                  {"v_med3_u32", k_med3}, {"v_min_u32", k_min}, {"v_and_or_b32", k_andor}, {"v_lshl_or_b32", k_lshlor}, {"v_cndmask_b32", k_cndmask},
                  {"v_cndmask_b32_e64 sgpr", k_cndmask64}, {"v_cmp_eq_u32 vcc", k_cmp}, {"v_cmp_eq_u32_e64 sgpr", k_cmp64}, {"v_cmp+s_nop+v_cndmask", k_cmpcnd}, {"v_mov_b32", k_mov}, {"v_xor_b32", k_xor}, {"v_add_u32", k_addu}, {"v_rcp_f32", k_rcp}, {"v_floor_f32", k_floor}, {"v_cvt_i32_f32", k_cvti}, {"v_max3_f32", k_max3f}, {"v_mul_f32", k_mulf}, {"v_sub_f32", k_subf},
                  {"v_sqrt_f32", k_sqrt}, {"v_add3_u32", k_add3}, {"v_sad_u32", k_sad}, {"v_perm_b32", k_perm}, {"v_bfe_u32", k_bfe}, {"v_fma_mix_f32", k_fmamix}};
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  int clk_khz = 0;
  hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  std::printf("# %s, %d CUs, nominal clock %d MHz; 8 waves per SIMD, 8 independent chains, 32768 instructions per wave\n", p.name, cus, clk_khz / 1000);
  std::printf("# cycles per wave64 instruction and SIMD at the nominal clock (4.0 = full rate)\n");
  for (const K& k : ks) {
    hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, out, 1u);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, out, 1u);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double inst_per_simd = 8.0 * 8 * 4096;  // 8 waves x 8 chains x 4096
    const double cyc = best * 1e-3 * (clk_khz * 1e3) / inst_per_simd;
    std::printf("%-22s %8.3f ms  %6.2f cycles\n", k.name, best, cyc);
  }
  return 0;
}
