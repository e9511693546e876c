// Scratch microbenchmark: issue cost (shader cycles per wave-instruction, from s_memtime) of the instruction kinds the
// tile pass's resolve is made of, at 1 / 2 / 4 / 5 waves per SIMD, eight independent chains per wave.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/valu_cost.hip -o tools/micro/valu_cost
// Output feeds DESIGN.md 4 ("what the vector pipes can issue") and bench.py's roofline_valu.peak.
// Round 4 (calibration): every figure is now derived THREE ways - from s_memtime ticks of the median wave (as before), from the
// kernel's wall time (HIP events) at the shader clock the same launch measured (s_memtime / s_memrealtime x 100 MHz), and
// checked against a census of where the waves really ran (HW_ID: waves per SIMD) - because round 3's table showed v_fma_f64
// and v_pk_fma_f32 at twice the datasheet's vector peak.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum { M_MUL, M_ADD, M_FMA, M_FMA3, M_MAD24, M_MULLO, M_AND, M_LSHL_ADD, M_CNDMASK, M_CMP, M_CVT_I32, M_FLOOR, M_MED3, M_RCP, M_SQRT,
       M_FMA64, M_ADD64, M_CVT_F32_F64, M_PK_FMA, M_PK_MUL, M_PK_ADD, M_MOV, M_BFE, M_MIX, M_MUL_SGPR, M_FMA_SGPR, M_CNDMASK_E64, M_CNDMASK_DEP, M_MIN_I32, M_ADD_U32, M_CVT_F32_I32, M_LDS_B64, M_SUB, M_FMAC, M_MAX, M_MUL_LIT, M_MUL_INL, M_ADD_LIT, M_LSHLREV, M_OR, M_MUL_I24, M_CVT_F32_U32, M_FMAMK, M_MUL_SGPR_MIX, M_FMA_NEG, M_ADD_INL, M_SUB_U32, M_COUNT };
static const char* kNames[M_COUNT] = { "v_mul_f32", "v_add_f32", "v_fma_f32 (a*s+s)", "v_fma_f32 (3 vgpr src)", "v_mad_u32_u24", "v_mul_lo_u32", "v_and_b32",
    "v_lshl_add_u32", "v_cndmask_b32", "v_cmp_gt_f32 (vcc)", "v_cvt_i32_f32", "v_floor_f32", "v_med3_f32", "v_rcp_f32", "v_sqrt_f32",
    "v_fma_f64", "v_add_f64", "v_cvt_f32_f64", "v_pk_fma_f32 (2 results)", "v_pk_mul_f32 (2 results)", "v_pk_add_f32 (2 results)", "v_mov_b32",
    "v_bfe_u32", "mix: 4 mul + 2 add + 2 fma", "v_mul_f32 (sgpr src)", "v_fma_f32 (v, sgpr, v)", "v_cndmask_b32 e64 (sgpr pair)",
    "v_cmp + v_cndmask pairs (per instr)", "v_min_i32", "v_add_u32", "v_cvt_f32_i32", "ds_read_b64 (issue only)",
    "v_sub_f32", "v_fmac_f32 (vop2)", "v_max_f32", "v_mul_f32 (literal 0.1)", "v_mul_f32 (inline 2.0)", "v_add_f32 (literal 0.1)", "v_lshlrev_b32 (imm 4)", "v_or_b32",
    "v_mul_i32_i24 (vop2)", "v_cvt_f32_u32", "v_fmamk_f32 (literal)", "4 x v_mul v,v + 4 x v_mul s,v", "v_fma_f32 (-a, b, c: 3 vgpr)", "v_add_f32 (inline 1.0)", "v_sub_u32" };

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, float s, int iters)
{
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    float a0 = threadIdx.x + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    const double ds = s;
    const float t = s * 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const f2 ss = {s, s};
    const unsigned long long msk = 0x5555555555555555ull ^ (unsigned long long)iters;
    __shared__ double ldsbuf[1024];
    ldsbuf[threadIdx.x] = a0; ldsbuf[threadIdx.x + 256] = a1; ldsbuf[threadIdx.x + 512] = a2; ldsbuf[threadIdx.x + 768] = a3;
    __syncthreads();
    const unsigned ldsaddr = (unsigned)(threadIdx.x & 63) * 64u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#define A(n) "+v"(a##n)
#define D(n) "+v"(d##n)
#define P(n) "+v"(p##n)
#define ALLA A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7)
#define ALLD D(0), D(1), D(2), D(3), D(4), D(5), D(6), D(7)
#define ALLP P(0), P(1), P(2), P(3), P(4), P(5), P(6), P(7)
#define OP2(op) asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" \
                             op " %6, %6, %8\n" op " %7, %7, %8\n" : ALLA : "v"(s))
#define OP1(op) asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n" : ALLA)
            if (MODE == M_MUL) OP2("v_mul_f32");
            else if (MODE == M_ADD) OP2("v_add_f32");
            else if (MODE == M_FMA) asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                                                 "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n" : ALLA : "v"(s));
            else if (MODE == M_FMA3) asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                                  "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" : ALLA : "v"(s), "v"(t));
            else if (MODE == M_MAD24) asm volatile("v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n"
                                                   "v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9\n" : ALLA : "v"(s), "v"(t));
            else if (MODE == M_MULLO) OP2("v_mul_lo_u32");
            else if (MODE == M_AND) OP2("v_and_b32");
            else if (MODE == M_LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %1, %1, 1, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_lshl_add_u32 %3, %3, 1, %8\n"
                                                      "v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8\n" : ALLA : "v"(s));
            else if (MODE == M_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                                                     "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n" : ALLA : "v"(s) : "vcc");
            else if (MODE == M_CMP) asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cmp_gt_f32 vcc, %1, %8\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_gt_f32 vcc, %3, %8\n"
                                                 "v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %5, %8\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_gt_f32 vcc, %7, %8\n" : ALLA : "v"(s) : "vcc");
            else if (MODE == M_CVT_I32) OP1("v_cvt_i32_f32");
            else if (MODE == M_FLOOR) OP1("v_floor_f32");
            else if (MODE == M_MED3) asm volatile("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n"
                                                  "v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9\n" : ALLA : "v"(s), "v"(t));
            else if (MODE == M_RCP) OP1("v_rcp_f32");
            else if (MODE == M_SQRT) OP1("v_sqrt_f32");
            else if (MODE == M_FMA64) asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n"
                                                   "v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8\n" : ALLD : "v"(ds));
            else if (MODE == M_ADD64) asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                                                   "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n" : ALLD : "v"(ds));
            else if (MODE == M_CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %8\n v_cvt_f32_f64 %1, %9\n v_cvt_f32_f64 %2, %10\n v_cvt_f32_f64 %3, %11\n"
                                                         "v_cvt_f32_f64 %4, %12\n v_cvt_f32_f64 %5, %13\n v_cvt_f32_f64 %6, %14\n v_cvt_f32_f64 %7, %15\n"
                                                         : ALLA : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));
            else if (MODE == M_PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                                                    "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n" : ALLP : "v"(ss));
            else if (MODE == M_PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                                                    "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n" : ALLP : "v"(ss));
            else if (MODE == M_PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                                                    "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n" : ALLP : "v"(ss));
            else if (MODE == M_MOV) asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n"
                                                 "v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n" : ALLA : "v"(s));
            else if (MODE == M_BFE) asm volatile("v_bfe_u32 %0, %0, 3, 8\n v_bfe_u32 %1, %1, 3, 8\n v_bfe_u32 %2, %2, 3, 8\n v_bfe_u32 %3, %3, 3, 8\n"
                                                 "v_bfe_u32 %4, %4, 3, 8\n v_bfe_u32 %5, %5, 3, 8\n v_bfe_u32 %6, %6, 3, 8\n v_bfe_u32 %7, %7, 3, 8\n" : ALLA);
            else if (MODE == M_MUL_SGPR) asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                                                      "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n" : ALLA : "s"(s));
            else if (MODE == M_FMA_SGPR) asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                                      "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" : ALLA : "s"(s), "v"(t));
            else if (MODE == M_CNDMASK_E64) asm volatile("v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n"
                                                         "v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9\n" : ALLA : "v"(s), "s"(msk));
            else if (MODE == M_CNDMASK_DEP) asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                                                         "v_cmp_gt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_gt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n" : ALLA : "v"(s) : "vcc");
            else if (MODE == M_MIN_I32) OP2("v_min_i32");
            else if (MODE == M_ADD_U32) OP2("v_add_u32");
            else if (MODE == M_CVT_F32_I32) OP1("v_cvt_f32_i32");
            else if (MODE == M_LDS_B64) { asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:8\n ds_read_b64 %2, %8 offset:16\n ds_read_b64 %3, %8 offset:24\n"
                                                       "ds_read_b64 %4, %8 offset:32\n ds_read_b64 %5, %8 offset:40\n ds_read_b64 %6, %8 offset:48\n ds_read_b64 %7, %8 offset:56\n s_waitcnt lgkmcnt(0)\n"
                                                       : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3), "=v"(d4), "=v"(d5), "=v"(d6), "=v"(d7) : "v"(ldsaddr) : "memory"); }
            else if (MODE == M_SUB) OP2("v_sub_f32");
            else if (MODE == M_FMAC) asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                                                  "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n" : ALLA : "v"(s), "v"(t));
            else if (MODE == M_MAX) OP2("v_max_f32");
            else if (MODE == M_MUL_LIT) asm volatile("v_mul_f32 %0, 0x3dcccccd, %0\n v_mul_f32 %1, 0x3dcccccd, %1\n v_mul_f32 %2, 0x3dcccccd, %2\n v_mul_f32 %3, 0x3dcccccd, %3\n"
                                                     "v_mul_f32 %4, 0x3dcccccd, %4\n v_mul_f32 %5, 0x3dcccccd, %5\n v_mul_f32 %6, 0x3dcccccd, %6\n v_mul_f32 %7, 0x3dcccccd, %7\n" : ALLA);
            else if (MODE == M_MUL_INL) asm volatile("v_mul_f32 %0, 2.0, %0\n v_mul_f32 %1, 2.0, %1\n v_mul_f32 %2, 2.0, %2\n v_mul_f32 %3, 2.0, %3\n"
                                                     "v_mul_f32 %4, 2.0, %4\n v_mul_f32 %5, 2.0, %5\n v_mul_f32 %6, 2.0, %6\n v_mul_f32 %7, 2.0, %7\n" : ALLA);
            else if (MODE == M_ADD_LIT) asm volatile("v_add_f32 %0, 0x3dcccccd, %0\n v_add_f32 %1, 0x3dcccccd, %1\n v_add_f32 %2, 0x3dcccccd, %2\n v_add_f32 %3, 0x3dcccccd, %3\n"
                                                     "v_add_f32 %4, 0x3dcccccd, %4\n v_add_f32 %5, 0x3dcccccd, %5\n v_add_f32 %6, 0x3dcccccd, %6\n v_add_f32 %7, 0x3dcccccd, %7\n" : ALLA);
            else if (MODE == M_ADD_INL) asm volatile("v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 1.0, %1\n v_add_f32 %2, 1.0, %2\n v_add_f32 %3, 1.0, %3\n"
                                                     "v_add_f32 %4, 1.0, %4\n v_add_f32 %5, 1.0, %5\n v_add_f32 %6, 1.0, %6\n v_add_f32 %7, 1.0, %7\n" : ALLA);
            else if (MODE == M_LSHLREV) asm volatile("v_lshlrev_b32 %0, 4, %0\n v_lshlrev_b32 %1, 4, %1\n v_lshlrev_b32 %2, 4, %2\n v_lshlrev_b32 %3, 4, %3\n"
                                                     "v_lshlrev_b32 %4, 4, %4\n v_lshlrev_b32 %5, 4, %5\n v_lshlrev_b32 %6, 4, %6\n v_lshlrev_b32 %7, 4, %7\n" : ALLA);
            else if (MODE == M_OR) OP2("v_or_b32");
            else if (MODE == M_SUB_U32) OP2("v_sub_u32");
            else if (MODE == M_MUL_I24) OP2("v_mul_i32_i24");
            else if (MODE == M_CVT_F32_U32) OP1("v_cvt_f32_u32");
            else if (MODE == M_FMAMK) asm volatile("v_fmamk_f32 %0, %0, 0x3dcccccd, %8\n v_fmamk_f32 %1, %1, 0x3dcccccd, %8\n v_fmamk_f32 %2, %2, 0x3dcccccd, %8\n v_fmamk_f32 %3, %3, 0x3dcccccd, %8\n"
                                                   "v_fmamk_f32 %4, %4, 0x3dcccccd, %8\n v_fmamk_f32 %5, %5, 0x3dcccccd, %8\n v_fmamk_f32 %6, %6, 0x3dcccccd, %8\n v_fmamk_f32 %7, %7, 0x3dcccccd, %8\n" : ALLA : "v"(s));
            else if (MODE == M_MUL_SGPR_MIX) asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %9, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %9, %3\n"
                                                          "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %9, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %9, %7\n" : ALLA : "v"(s), "s"(t));
            else if (MODE == M_FMA_NEG) asm volatile("v_fma_f32 %0, -%0, %8, %9\n v_fma_f32 %1, -%1, %8, %9\n v_fma_f32 %2, -%2, %8, %9\n v_fma_f32 %3, -%3, %8, %9\n"
                                                     "v_fma_f32 %4, -%4, %8, %9\n v_fma_f32 %5, -%5, %8, %9\n v_fma_f32 %6, -%6, %8, %9\n v_fma_f32 %7, -%7, %8, %9\n" : ALLA : "v"(s), "v"(t));
            else if (MODE == M_MIX) asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_fma_f32 %3, %3, %8, %9\n"
                                                 "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_fma_f32 %7, %7, %8, %9\n" : ALLA : "v"(s), "v"(t));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 4;
        cyc[w] = t1 - t0;
        cyc[nw + w] = rt1 - rt0;                                             // 100 MHz ticks over the same interval (+ prologue)
        // where this wave ran: XCC id, and HW_ID's SE / SH / CU / SIMD fields (wave slot, queue and VM ids masked off)
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        cyc[2 * nw + w] = ((unsigned long long)(xcc & 0xfu) << 16) | (hw & 0xff30u);
        cyc[3 * nw + w] = t0;                                                // start stamp: were all waves resident at once?
        cyc[4 * nw + w] = t1;
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7)
        + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.x + p6.x + p7.x;
}

typedef void (*kern_t)(float*, unsigned long long*, float, int);
template <int M> struct Tab { static void fill(kern_t* t) { t[M] = k<M>; Tab<M - 1>::fill(t); } };
template <> struct Tab<-1> { static void fill(kern_t*) {} };

int main()
{
    kern_t tab[M_COUNT]; Tab<M_COUNT - 1>::fill(tab);
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    unsigned long long* c; hipMalloc(&c, 5 * 256 * 8 * 4 * sizeof(unsigned long long));
    std::vector<unsigned long long> h(5 * 256 * 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2048;
    printf("%-28s", "cycles per wave-instruction");
    const int wps[] = { 1, 2, 4, 5, 8 };
    for (int w : wps) printf("  %d w/SIMD", w);
    printf("   (median wave: s_memtime ticks x waves per SIMD / instructions; 5 -> 5 blocks of 256 threads per CU)\n");
    // keep the device busy for a while first: the clock it holds under load is part of the answer
    for (int r = 0; r < 200; r++) hipLaunchKernelGGL(tab[M_FMA3], dim3(2048), dim3(256), 0, 0, d, c, 1.0001f, 2048);
    hipDeviceSynchronize();
    std::vector<double> wall_cost(M_COUNT * 5), tick_cost(M_COUNT * 5), clk_ghz(M_COUNT * 5), overlap(M_COUNT * 5);
    std::vector<int> simds(M_COUNT * 5), wmin(M_COUNT * 5), wmax(M_COUNT * 5);
    for (int m = 0; m < M_COUNT; m++) {
        printf("%-28s", kNames[m]);
        for (int wi = 0; wi < 5; wi++) {
            const int w = wps[wi];
            const int grid = 256 * w;
            const size_t nw = (size_t)grid * 4;
            hipLaunchKernelGGL(tab[m], dim3(grid), dim3(256), 0, 0, d, c, 1.0001f, 16);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(tab[m], dim3(grid), dim3(256), 0, 0, d, c, 1.0001f, iters);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0.0f; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), c, sizeof(unsigned long long) * nw * 5, hipMemcpyDeviceToHost);
            std::vector<unsigned long long> t(h.begin(), h.begin() + nw), rt(h.begin() + nw, h.begin() + 2 * nw);
            std::sort(t.begin(), t.end()); std::sort(rt.begin(), rt.end());
            const double ticks = (double)t[nw / 2], rticks = (double)rt[nw / 2];
            // census: waves per (xcc, se, sh, cu, simd)
            std::vector<unsigned long long> key(h.begin() + 2 * nw, h.begin() + 3 * nw);
            std::sort(key.begin(), key.end());
            int n_simd = 0, lo = 1 << 30, hi = 0;
            for (size_t i = 0; i < nw;) { size_t j = i; while (j < nw && key[j] == key[i]) j++; n_simd++; lo = std::min(lo, (int)(j - i)); hi = std::max(hi, (int)(j - i)); i = j; }
            // were the waves resident together?  latest start vs earliest end, as a share of the median lifetime
            const unsigned long long last_start = *std::max_element(h.begin() + 3 * nw, h.begin() + 4 * nw), first_end = *std::min_element(h.begin() + 4 * nw, h.begin() + 5 * nw);
            const int q = m * 5 + wi;
            const double ghz = ticks / (rticks * 10.0);                      // s_memrealtime: 100 MHz = 10 ns per tick
            clk_ghz[q] = ghz; simds[q] = n_simd; wmin[q] = lo; wmax[q] = hi;
            overlap[q] = ((double)first_end - (double)last_start) / ticks;
            // one wave's lifetime covers the instructions of all w waves sharing its SIMD
            tick_cost[q] = ticks / ((double)iters * 64.0 * w);
            // wall time: all waves' instructions / SIMDs that held waves, in cycles of the measured clock
            wall_cost[q] = (double)ms * 1e6 * ghz / ((double)iters * 64.0 * (double)nw / (double)n_simd);
            printf("  %8.2f", tick_cost[q]);
        }
        printf("\n");
    }
    printf("\n# the same from the kernel's WALL time (HIP events) x the clock of that launch (s_memtime / s_memrealtime x 100 MHz), per SIMD that held waves\n");
    printf("%-28s", "cycles per wave-instruction");
    for (int w : wps) printf("  %d w/SIMD", w);
    printf("   | clock GHz, SIMDs with waves, waves per SIMD min..max, co-residence (1 = all waves alive together) at 8 w/SIMD\n");
    for (int m = 0; m < M_COUNT; m++) {
        printf("%-28s", kNames[m]);
        for (int wi = 0; wi < 5; wi++) printf("  %8.2f", wall_cost[m * 5 + wi]);
        const int q = m * 5 + 4;
        printf("   | %.3f  %d  %d..%d  %.2f\n", clk_ghz[q], simds[q], wmin[q], wmax[q], overlap[q]);
    }
    printf("\n# census per column of the first row (v_mul_f32): SIMDs with waves, waves per SIMD min..max, clock GHz, co-residence\n");
    for (int wi = 0; wi < 5; wi++) printf("#   %d w/SIMD asked: %d SIMDs, %d..%d waves each, %.3f GHz, co-residence %.2f\n", wps[wi], simds[wi], wmin[wi], wmax[wi], clk_ghz[wi], overlap[wi]);
    return 0;
}
