// am_fe_cmpx.h -- first-stage preamble test with EXEC-narrowing compares, shared by the fused front ends
// (am_fe2.hip: tile kernel; am_fe3.hip: streaming kernel).  Not part of the public ABI.
#ifndef AM_FE_CMPX_H
#define AM_FE_CMPX_H

#include <stdint.h>

// First-stage test with EXEC-narrowing compares (gfx9 v_cmpx: EXEC &= condition), eight samples per
// block: three VALU instructions and one scalar move per sample.  The compiler's own form of the same
// predicate keeps every partial result as a lane mask in scalar registers and spends ~9 scalar
// instructions per sample on combining them -- the scalar unit, not the vector ALU, then bounds the phase.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FE2_NO_CMPX)
#define FE2_CMPX 1
// bit SH+k of cm |= (x[k] > thr[k]) & !(x[k+1] > x[k])        preamble_impl.cc:174-175
template <int SH>
__device__ __forceinline__ void fe2_peak8(uint32_t &cm, const float *x, float x8, const float *thr)
{
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmpx_gt_f32_e32 vcc, %[x0], %[t0]\n\t" "v_cmpx_ngt_f32_e32 vcc, %[x1], %[x0]\n\t"
        "v_or_b32_e32 %[cm], %[b0], %[cm]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_gt_f32_e32 vcc, %[x1], %[t1]\n\t" "v_cmpx_ngt_f32_e32 vcc, %[x2], %[x1]\n\t"
        "v_or_b32_e32 %[cm], %[b1], %[cm]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_gt_f32_e32 vcc, %[x2], %[t2]\n\t" "v_cmpx_ngt_f32_e32 vcc, %[x3], %[x2]\n\t"
        "v_or_b32_e32 %[cm], %[b2], %[cm]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_gt_f32_e32 vcc, %[x3], %[t3]\n\t" "v_cmpx_ngt_f32_e32 vcc, %[x4], %[x3]\n\t"
        "v_or_b32_e32 %[cm], %[b3], %[cm]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_gt_f32_e32 vcc, %[x4], %[t4]\n\t" "v_cmpx_ngt_f32_e32 vcc, %[x5], %[x4]\n\t"
        "v_or_b32_e32 %[cm], %[b4], %[cm]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_gt_f32_e32 vcc, %[x5], %[t5]\n\t" "v_cmpx_ngt_f32_e32 vcc, %[x6], %[x5]\n\t"
        "v_or_b32_e32 %[cm], %[b5], %[cm]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_gt_f32_e32 vcc, %[x6], %[t6]\n\t" "v_cmpx_ngt_f32_e32 vcc, %[x7], %[x6]\n\t"
        "v_or_b32_e32 %[cm], %[b6], %[cm]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_gt_f32_e32 vcc, %[x7], %[t7]\n\t" "v_cmpx_ngt_f32_e32 vcc, %[x8], %[x7]\n\t"
        "v_or_b32_e32 %[cm], %[b7], %[cm]\n\t" "s_mov_b64 exec, %[sv]"
        : [cm] "+v"(cm), [sv] "=&s"(sv)
        : [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]),
          [x6] "v"(x[6]), [x7] "v"(x[7]), [x8] "v"(x8), [t0] "v"(thr[0]), [t1] "v"(thr[1]), [t2] "v"(thr[2]),
          [t3] "v"(thr[3]), [t4] "v"(thr[4]), [t5] "v"(thr[5]), [t6] "v"(thr[6]), [t7] "v"(thr[7]),
          [b0] "n"(1u << (SH + 0)), [b1] "n"(1u << (SH + 1)), [b2] "n"(1u << (SH + 2)), [b3] "n"(1u << (SH + 3)),
          [b4] "n"(1u << (SH + 4)), [b5] "n"(1u << (SH + 5)), [b6] "n"(1u << (SH + 6)), [b7] "n"(1u << (SH + 7))
        : "vcc");
}
// bit SH+k of cm &= !(w[k] < thr[k])                            preamble_impl.cc:177-179 (w = weakest later pulse)
template <int SH>
__device__ __forceinline__ void fe2_weak8(uint32_t &cm, const float *w, const float *thr)
{
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmpx_lt_f32_e32 vcc, %[w0], %[t0]\n\t" "v_and_b32_e32 %[cm], %[b0], %[cm]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_lt_f32_e32 vcc, %[w1], %[t1]\n\t" "v_and_b32_e32 %[cm], %[b1], %[cm]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_lt_f32_e32 vcc, %[w2], %[t2]\n\t" "v_and_b32_e32 %[cm], %[b2], %[cm]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_lt_f32_e32 vcc, %[w3], %[t3]\n\t" "v_and_b32_e32 %[cm], %[b3], %[cm]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_lt_f32_e32 vcc, %[w4], %[t4]\n\t" "v_and_b32_e32 %[cm], %[b4], %[cm]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_lt_f32_e32 vcc, %[w5], %[t5]\n\t" "v_and_b32_e32 %[cm], %[b5], %[cm]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_lt_f32_e32 vcc, %[w6], %[t6]\n\t" "v_and_b32_e32 %[cm], %[b6], %[cm]\n\t" "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_lt_f32_e32 vcc, %[w7], %[t7]\n\t" "v_and_b32_e32 %[cm], %[b7], %[cm]\n\t" "s_mov_b64 exec, %[sv]"
        : [cm] "+v"(cm), [sv] "=&s"(sv)
        : [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [w4] "v"(w[4]), [w5] "v"(w[5]),
          [w6] "v"(w[6]), [w7] "v"(w[7]), [t0] "v"(thr[0]), [t1] "v"(thr[1]), [t2] "v"(thr[2]), [t3] "v"(thr[3]),
          [t4] "v"(thr[4]), [t5] "v"(thr[5]), [t6] "v"(thr[6]), [t7] "v"(thr[7]),
          [b0] "n"(~(1u << (SH + 0))), [b1] "n"(~(1u << (SH + 1))), [b2] "n"(~(1u << (SH + 2))),
          [b3] "n"(~(1u << (SH + 3))), [b4] "n"(~(1u << (SH + 4))), [b5] "n"(~(1u << (SH + 5))),
          [b6] "n"(~(1u << (SH + 6))), [b7] "n"(~(1u << (SH + 7)))
        : "vcc");
}
#endif

#endif
