// am_fe_stream.h -- device helpers shared by the streaming fused front ends (am_fe3.hip: 64 Msps, one 32-sample chip per
// lane, 16-byte LDS rows; am_fe4.hip: 2 .. 40 Msps, a unit of G chips per lane, 8-byte LDS rows).  Not part of the public ABI.
#ifndef AM_FE_STREAM_H
#define AM_FE_STREAM_H

#include "am_internal.h"

// everything this workgroup wrote to LDS is visible to it after this (global loads / stores stay in flight)
__device__ __forceinline__ void fes_barrier()
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}

// value of lane-1 (lane 0 of a wave gets `first`) / of lane+1 (lane 63 gets `last`): DPP wave_shr:1 / wave_shl:1 on gfx9
__device__ __forceinline__ float fes_from_prev_lane(float v, float first, int lane)
{
#if defined(__HIP_DEVICE_COMPILE__)
    (void)lane;
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, first), __builtin_bit_cast(int, v),
                                                                 0x138, 0xf, 0xf, false));
#else
    const float s = __shfl_up(v, 1, AM_WAVE);
    return lane == 0 ? first : s;
#endif
}
__device__ __forceinline__ float fes_from_next_lane(float v, float last, int lane)
{
#if defined(__HIP_DEVICE_COMPILE__)
    (void)lane;
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, last), __builtin_bit_cast(int, v),
                                                                 0x130, 0xf, 0xf, false));
#else
    const float s = __shfl_down(v, 1, AM_WAVE);
    return lane == AM_WAVE - 1 ? last : s;
#endif
}

// the same inside rows of 16 lanes: the first / last lane of a row gets 0 (DPP row_shr:1 / row_shl:1 with bound_ctrl: the
// move folds into the addition that follows)
__device__ __forceinline__ float fes_from_prev_lane_row16(float v, int lane)
{
#if defined(__HIP_DEVICE_COMPILE__)
    (void)lane;
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
#else
    const float s = __shfl_up(v, 1, AM_WAVE);
    return (lane & 15) == 0 ? 0.0f : s;
#endif
}
__device__ __forceinline__ float fes_from_next_lane_row16(float v, int lane)
{
#if defined(__HIP_DEVICE_COMPILE__)
    (void)lane;
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));
#else
    const float s = __shfl_down(v, 1, AM_WAVE);
    return (lane & 15) == 15 ? 0.0f : s;
#endif
}

// value of lane-1, lane 0 gets lane 63's (DPP wave_ror:1: every lane has a source, so the move folds into the consumer)
__device__ __forceinline__ float fes_from_prev_lane_ror(float v, int lane)
{
#if defined(__HIP_DEVICE_COMPILE__)
    (void)lane;
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x13C, 0xf, 0xf, true));
#else
    return __shfl(v, (lane + AM_WAVE - 1) & (AM_WAVE - 1), AM_WAVE);
#endif
}

// Two independent fp32 operations per instruction (v_pk_add_f32 / v_pk_mul_f32: same rounding as the scalar forms; the
// streaming front ends are bound by VALU issue, not by HBM).  On the host (CPU-fiber tests) a plain pair.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float fes_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ fes_f2 fes_mk2(float a, float b) { fes_f2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ fes_f2 fes_pk_add(fes_f2 a, fes_f2 b) { return a + b; }
__device__ __forceinline__ fes_f2 fes_pk_mul(fes_f2 a, fes_f2 b) { return a * b; }
#else
struct fes_f2 { float x, y; };
__device__ __forceinline__ fes_f2 fes_mk2(float a, float b) { fes_f2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ fes_f2 fes_pk_add(fes_f2 a, fes_f2 b) { fes_f2 r; r.x = a.x + b.x; r.y = a.y + b.y; return r; }
__device__ __forceinline__ fes_f2 fes_pk_mul(fes_f2 a, fes_f2 b) { fes_f2 r; r.x = a.x * b.x; r.y = a.y * b.y; return r; }
#endif

// 24-bit multiply (full rate; v_mul_lo_u32 is quarter rate): operands are ring slots, lane and chip indices
__device__ __forceinline__ int fes_mul24(int a, int b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __mul24(a, b);
#else
    return a * b;
#endif
}

// x / D for 0 <= x < 1024 and D <= 64 (thread and lane indices by lanes per row / per block) without the quarter-rate 32-bit
// multiply: exact there (x (D - 1) < 65536; tests/test_fe_stream_helpers.py runs the whole domain)
template <int D>
__device__ __forceinline__ int fes_div_small(int x)
{
    static_assert(D >= 1 && D <= 64, "divisor");
    return fes_mul24(x, (65536 + D - 1) / D) >> 16;
}

// 16 bytes of raw IQ with the streaming (nt) policy: every sample is read once
#if defined(__clang__)
typedef float fes_f4 __attribute__((ext_vector_type(4)));
#endif
__device__ __forceinline__ float4 fes_gload16(const void *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const fes_f4 t = __builtin_nontemporal_load(reinterpret_cast<const fes_f4 *>(p));
    float4 r; r.x = t.x; r.y = t.y; r.z = t.z; r.w = t.w;
    return r;
#else
    return *reinterpret_cast<const float4 *>(p);
#endif
}

// the same from a wave-uniform base (held in scalar registers) + a 32-bit lane offset: global_load with saddr + voffset
__device__ __forceinline__ float4 fes_gload16_at(unsigned long long base, unsigned off)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const fes_f4 __attribute__((address_space(1))) *gp;
    const fes_f4 t = __builtin_nontemporal_load(reinterpret_cast<gp>(base + (unsigned long long)off));
    float4 r; r.x = t.x; r.y = t.y; r.z = t.z; r.w = t.w;
    return r;
#else
    return *reinterpret_cast<const float4 *>(base + (unsigned long long)off);
#endif
}

// the same with the default cache policy (samples that another kernel reads again soon: am_k_gather_wg's rows)
__device__ __forceinline__ float4 fes_gload16_cached_at(unsigned long long base, unsigned off)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const fes_f4 __attribute__((address_space(1))) *gp;
    const fes_f4 t = *reinterpret_cast<gp>(base + (unsigned long long)off);
    float4 r; r.x = t.x; r.y = t.y; r.z = t.z; r.w = t.w;
    return r;
#else
    return *reinterpret_cast<const float4 *>(base + (unsigned long long)off);
#endif
}

// Wave priority by step (round 5).  A CU's arbiters serve the OLDEST wave first, so of the six persistent workgroups of a CU the
// first started ran ahead of the others all the way: it ended at ~78 us of a 110 us launch, the last started at ~104 (profiling
// build's timeline, profiles/r5_prio) -- every CU worked through its last quarter with ever fewer workgroups to hide latency
// behind, and the chip's memory system with ever fewer requests in flight.  A priority that falls with the step index modulo 4
// puts a workgroup that is one step behind one level ABOVE its neighbour three times out of four: the six stay within a step
// of each other and end together (first / last end of a CU: 88 / 97 us).  Scheduling only: no result depends on it.
#ifndef FES_STEP_PRIO
#define FES_STEP_PRIO 1
#endif
__device__ __forceinline__ void fes_step_priority(unsigned k)
{
#if FES_STEP_PRIO && defined(__HIP_DEVICE_COMPILE__)
    const unsigned lvl = 3u - (k & 3u);
    if (lvl == 0u) __builtin_amdgcn_s_setprio(0);
    else if (lvl == 1u) __builtin_amdgcn_s_setprio(1);
    else if (lvl == 2u) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
#else
    (void)k;
#endif
}

static inline long long fes_floor_div(long long x, long long d) { return x >= 0 ? x / d : -((-x + d - 1) / d); }
static inline long long fes_ceil_div(long long x, long long d) { return -fes_floor_div(-x, d); }

#endif
