// wave_common.hpp -- device helpers shared by the wave-autonomous kernels (front_wave.hip, cascade_wave.hip):
// register-prefetched raw loads, the unpack of prefetched words, packed-f32 primitives, NCO lookups.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/iqgpu.h"
#include "dsp_device.hpp"
#include "kernels.hpp"

namespace iqgpu {

constexpr int kRowB = 48;                                   // LDS row: 4 cf32 + 16 B pad (odd number of 16-byte slots)

// k_cascade keeps stage 0's input (rows written by lane PAIRS: lane 2i holds samples 4i, 4i+1 and lane 2i+1
// samples 4i+2, 4i+3 of a parity stream) as two planes of 16-byte half rows instead: plane h, row r = samples
// 4r + 2h, 4r + 2h + 1 at byte 16 r of the plane, the planes 64 (mod 128) bytes apart.  A third less LDS than
// the padded rows (which is what lets a 4-stage cascade run 16 waves per CU) and fewer write conflicts: the
// four rows an 8-lane write group touches are 64 contiguous bytes in each plane.  A ds_read_b128 of "row
// lane + c" is 16 contiguous bytes per lane.  (k_front_s1 keeps the padded rows: there the planes measured
// 4 % slower, SQ_LDS_IDX_ACTIVE 290 -> 310 cycles per tile.)
__host__ __device__ constexpr int plane_stride(int rows)
{
    return (rows * 16 + 63) / 128 * 128 + 64;               // smallest value >= 16 rows that is 64 (mod 128)
}

struct RawChunk { uint32_t w[8]; };

// NT: with the non-temporal hint -- for streams a kernel reads exactly once (tools/stream_pattern.hip: the access pattern
// of these kernels, nothing computed, reads 6.1 TB/s as plain loads and 7.0 TB/s as non-temporal ones)
#ifndef IQGPU_NT_S1
#define IQGPU_NT_S1 1
#endif
#ifndef IQGPU_NT_CASC
#define IQGPU_NT_CASC 1
#endif
#ifndef IQGPU_NT_FAT
#define IQGPU_NT_FAT 0
#endif
template <int BPS, bool NT = false>
__device__ __forceinline__ void load_chunk(const char *p, RawChunk &r)
{
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    typedef uint32_t u2v __attribute__((ext_vector_type(2)));
    if (BPS == 4) {
        const u4v v = NT ? __builtin_nontemporal_load((const u4v *)p) : *(const u4v *)p;
        r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
    } else if (BPS == 2) {
        const u2v v = NT ? __builtin_nontemporal_load((const u2v *)p) : *(const u2v *)p;
        r.w[0] = v.x; r.w[1] = v.y;
    } else {
        const u4v v0 = NT ? __builtin_nontemporal_load((const u4v *)p) : *(const u4v *)p;
        const u4v v1 = NT ? __builtin_nontemporal_load((const u4v *)(p + 16)) : *(const u4v *)(p + 16);
        r.w[0] = v0.x; r.w[1] = v0.y; r.w[2] = v0.z; r.w[3] = v0.w;
        r.w[4] = v1.x; r.w[5] = v1.y; r.w[6] = v1.z; r.w[7] = v1.w;
    }
}

// four frames from prefetched words; arithmetic identical to unpack_one (src/sample_convert.c:75-96).
// unit_gain skips the multiply by 1.0f (exact).
template <int BPS>
__device__ __forceinline__ void unpack_chunk(const RawChunk &r, int fmt, float gain, bool unit_gain, cf2 x[4])
{
    if (BPS == 4) {
        if (fmt == IQGPU_FMT_CU16) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                x[s].x = ((float)(r.w[s] & 0xffffu) - 32767.5f) * (1.0f / 32768.0f);
                x[s].y = ((float)(r.w[s] >> 16) - 32767.5f) * (1.0f / 32768.0f);
            }
        } else {
            const float norm = (fmt == IQGPU_FMT_CS16) ? 1.0f / 32768.0f : 1.0f / 2048.0f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                x[s].x = (float)(short)(r.w[s] & 0xffffu) * norm;
                x[s].y = (float)(short)(r.w[s] >> 16) * norm;
            }
        }
    } else if (BPS == 2) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const unsigned h = r.w[s >> 1] >> ((s & 1) * 16);
            if (fmt == IQGPU_FMT_CU8) {
                x[s].x = ((float)(h & 0xffu) - 127.5f) * (1.0f / 128.0f);
                x[s].y = ((float)((h >> 8) & 0xffu) - 127.5f) * (1.0f / 128.0f);
            } else {
                x[s].x = (float)(signed char)(h & 0xffu) * (1.0f / 128.0f);
                x[s].y = (float)(signed char)((h >> 8) & 0xffu) * (1.0f / 128.0f);
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            x[s].x = __uint_as_float(r.w[2 * s]);
            x[s].y = __uint_as_float(r.w[2 * s + 1]);
        }
    }
    if (!unit_gain) {
#pragma unroll
        for (int s = 0; s < 4; ++s) { x[s].x *= gain; x[s].y *= gain; }
    }
}

__device__ __forceinline__ float4 ld4(const char *p) { return *(const float4 *)p; }

// ---- packed-f32 primitives (VOP3P).  hipcc's SLP vectoriser does find v_pk_fma_f32 on its own but
// pays for every scalar broadcast with v_mov pairs and serialises the polyphase chains; the three
// hot inner products are therefore written with explicit op_sel forms (this file is compiled with
// -fno-slp-vectorize).  A v2f is one {re, im} sample in an aligned VGPR pair.
typedef float v2f __attribute__((ext_vector_type(2)));

// acc += t.lo * x   /   acc += t.hi * x      (t: a pair of real taps, x: {re, im})
__device__ __forceinline__ void pk_fma_lo(v2f &acc, v2f t, v2f x)
{
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(t), "v"(x));
}
__device__ __forceinline__ void pk_fma_hi(v2f &acc, v2f t, v2f x)
{
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(t), "v"(x));
}
// the same with the tap pair in SGPRs (wave-uniform half-band taps)
__device__ __forceinline__ void pk_fma_lo_s(v2f &acc, v2f t, v2f x)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(t), "v"(x));
}
__device__ __forceinline__ void pk_fma_hi_s(v2f &acc, v2f t, v2f x)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(t), "v"(x));
}
// x * (c + j s) with cs = {c, s}:  t = {-xi s, xi c};  y = {xr c, xr s} + t
// (one asm statement for the pair: between two dependent single-instruction asm statements the compiler, which cannot
//  see inside them, puts an s_nop -- 53 of them per tile in round 1's loop)
__device__ __forceinline__ v2f pk_cmul(v2f x, v2f cs)
{
    v2f y;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "=&v"(y) : "v"(x), "v"(cs));
    return y;
}
// Dependent packed FMAs are issued from multi-instruction asm blocks: hipcc pads every boundary between two
// single-instruction asm statements that depend on each other with an s_nop (it cannot see into the asm to
// count wait states), which cost 53 nops per tile when every FMA was its own statement.
// the eight FMAs of one half-band tap pair {h[2q], h[2q+1]} on four accumulators: acc[r] += h[2q] e[r+1] + h[2q+1] e[r]
#define IQGPU_HB8(T, EA, EB, EC, ED, EE)                                               \
    "v_pk_fma_f32 %[a0], %[" T "], %[" EB "], %[a0] op_sel_hi:[0,1,1]\n\t"              \
    "v_pk_fma_f32 %[a1], %[" T "], %[" EC "], %[a1] op_sel_hi:[0,1,1]\n\t"              \
    "v_pk_fma_f32 %[a2], %[" T "], %[" ED "], %[a2] op_sel_hi:[0,1,1]\n\t"              \
    "v_pk_fma_f32 %[a3], %[" T "], %[" EE "], %[a3] op_sel_hi:[0,1,1]\n\t"              \
    "v_pk_fma_f32 %[a0], %[" T "], %[" EA "], %[a0] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t" \
    "v_pk_fma_f32 %[a1], %[" T "], %[" EB "], %[a1] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t" \
    "v_pk_fma_f32 %[a2], %[" T "], %[" EC "], %[a2] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t" \
    "v_pk_fma_f32 %[a3], %[" T "], %[" ED "], %[a3] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
// five consecutive tap pairs t[0..4] (SGPR pairs) against the 13 even-stream registers e[0..12]; pair i reads e[8 - 2i .. 12 - 2i]
__device__ __forceinline__ void pk_fma_hb40(v2f acc[4], const v2f *t, const v2f *e)
{
    asm(IQGPU_HB8("t0", "e8", "e9", "e10", "e11", "e12") IQGPU_HB8("t1", "e6", "e7", "e8", "e9", "e10")
        IQGPU_HB8("t2", "e4", "e5", "e6", "e7", "e8") IQGPU_HB8("t3", "e2", "e3", "e4", "e5", "e6")
        IQGPU_HB8("t4", "e0", "e1", "e2", "e3", "e4")
        : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3])
        : [t0] "s"(t[0]), [t1] "s"(t[1]), [t2] "s"(t[2]), [t3] "s"(t[3]), [t4] "s"(t[4]),
          [e0] "v"(e[0]), [e1] "v"(e[1]), [e2] "v"(e[2]), [e3] "v"(e[3]), [e4] "v"(e[4]), [e5] "v"(e[5]), [e6] "v"(e[6]),
          [e7] "v"(e[7]), [e8] "v"(e[8]), [e9] "v"(e[9]), [e10] "v"(e[10]), [e11] "v"(e[11]), [e12] "v"(e[12]));
}
// the four FMAs of one polyphase tap pair on two slots: ya += ta.lo h1 + ta.hi h0,  yb += tb.lo h2 + tb.hi h1
#define IQGPU_PP4(TA, TB, HA, HB_, HC)                                                  \
    "v_pk_fma_f32 %[ya], %[" TA "], %[" HB_ "], %[ya] op_sel_hi:[0,1,1]\n\t"             \
    "v_pk_fma_f32 %[yb], %[" TB "], %[" HC "], %[yb] op_sel_hi:[0,1,1]\n\t"              \
    "v_pk_fma_f32 %[ya], %[" TA "], %[" HA "], %[ya] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t" \
    "v_pk_fma_f32 %[yb], %[" TB "], %[" HB_ "], %[yb] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
// the same with ya, yb STARTED by the first two products (no zero-initialised accumulators: 8 v_mov per tile;
// a sum of products that are all -0 now ends as -0 instead of +0, nothing else changes)
#define IQGPU_PP4_FIRST(TA, TB, HA, HB_, HC)                                            \
    "v_pk_mul_f32 %[ya], %[" TA "], %[" HB_ "] op_sel_hi:[0,1]\n\t"                      \
    "v_pk_mul_f32 %[yb], %[" TB "], %[" HC "] op_sel_hi:[0,1]\n\t"                       \
    "v_pk_fma_f32 %[ya], %[" TA "], %[" HA "], %[ya] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t" \
    "v_pk_fma_f32 %[yb], %[" TB "], %[" HB_ "], %[yb] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
// tap pairs 0..3 of both slots against g[0..8]; pair i reads g[6 - 2i .. 8 - 2i]; ya, yb are outputs only
__device__ __forceinline__ void pk_fma_pp16(v2f &ya, v2f &yb, const v2f *ta, const v2f *tb, const v2f *g)
{
    asm(IQGPU_PP4_FIRST("p0", "q0", "g6", "g7", "g8") IQGPU_PP4("p1", "q1", "g4", "g5", "g6")
        IQGPU_PP4("p2", "q2", "g2", "g3", "g4") IQGPU_PP4("p3", "q3", "g0", "g1", "g2")
        : [ya] "=&v"(ya), [yb] "=&v"(yb)
        : [p0] "v"(ta[0]), [p1] "v"(ta[1]), [p2] "v"(ta[2]), [p3] "v"(ta[3]),
          [q0] "v"(tb[0]), [q1] "v"(tb[1]), [q2] "v"(tb[2]), [q3] "v"(tb[3]),
          [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [g5] "v"(g[5]), [g6] "v"(g[6]),
          [g7] "v"(g[7]), [g8] "v"(g[8]));
}
// tap pairs 0..2 of both slots against g[0..6]; pair i reads g[4 - 2i .. 6 - 2i]
__device__ __forceinline__ void pk_fma_pp12(v2f &ya, v2f &yb, const v2f *ta, const v2f *tb, const v2f *g)
{
    asm(IQGPU_PP4("p0", "q0", "g4", "g5", "g6") IQGPU_PP4("p1", "q1", "g2", "g3", "g4") IQGPU_PP4("p2", "q2", "g0", "g1", "g2")
        : [ya] "+v"(ya), [yb] "+v"(yb)
        : [p0] "v"(ta[0]), [p1] "v"(ta[1]), [p2] "v"(ta[2]), [q0] "v"(tb[0]), [q1] "v"(tb[1]), [q2] "v"(tb[2]),
          [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [g5] "v"(g[5]), [g6] "v"(g[6]));
}
// the table sits at an LDS address that is a multiple of its 8 KiB (the kernels put it first and check): index and
// base meet in ONE v_and_or_b32 instead of an and and an add (eight lookups per tile)
// (copy: which 8 KiB copy of the table behind `tab` -- a constant that lands in the instruction's offset field)
__device__ __forceinline__ v2f nco_phasor2(const cf2 *tab, uint32_t theta, int copy = 0)
{
    typedef __attribute__((address_space(3))) const char lds_char;
    typedef __attribute__((address_space(3))) const v2f lds_v2f;
    const uint32_t base = (uint32_t)(size_t)(__attribute__((address_space(3))) const void *)tab;
    lds_char *p = (lds_char *)(size_t)((((theta + (1u << 21)) >> 19) & 0x1ff8u) | base);
    return *(lds_v2f *)(p + copy * 8192);
}

// y = x * (c + j s); the sign of s for mix-down is folded into the LDS copy of the table
__device__ __forceinline__ cf2 cmul_tab(cf2 x, cf2 cs)
{
    cf2 y;
    y.x = fmaf(x.x, cs.x, -(x.y * cs.y));
    y.y = fmaf(x.x, cs.y, x.y * cs.x);
    return y;
}

// Per-lane constants of the dc-blocker scan: c^(4 lane), and the weights of the two cross-row steps.
struct DcLane { float pw, wa, wb; };
__device__ __forceinline__ DcLane dc_lane_init(const FrontArgs &a, int lane)
{
    DcLane d{1.0f, 1.0f, 1.0f};
    const int ia = (lane & 15) + 1, ib = lane - 31;          // c^(4 ia): from lane 15 of the row before; c^(4 ib): from lane 31
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (lane & (1 << k)) d.pw *= a.dc_cpow[k];
        if (ia & (1 << k)) d.wa *= a.dc_cpow[k];
        if (ib > 0 && (ib & (1 << k))) d.wb *= a.dc_cpow[k];
    }
    return d;
}
template <int CTRL, int ROW_MASK, bool BOUND0>
__device__ __forceinline__ float dpp_f(float old, float src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, ROW_MASK, 0xf, BOUND0));
}

// DC blocker over one 256-frame chunk (lane: frames 4 lane .. 4 lane + 3).  hist: bit s set = frame s is an
// already-processed history frame (enters the recurrence as zero and is left untouched).
// (vr, vi) = v[n-1] at the chunk's first frame on entry, at the next chunk's first frame on exit.
// The scan over the 64 lanes -- B_l = sum_{j <= l} c^(4 (l - j)) b_j -- runs on DPP moves (round 4; until then six ds_bpermute
// pairs with their index arithmetic, 16 LDS-crossbar operations per chunk: two thirds of this function's instructions):
// four row_shr steps inside the rows of 16 (lanes without a source read 0), then lane 15 of rows 0 / 2 into rows 1 / 3
// (row_bcast15) and lane 31 into rows 2 / 3 (row_bcast31), each weighted by the distance to the receiving lane.
__device__ __forceinline__ void dc_chunk(const FrontArgs &a, int lane, const DcLane &dl, cf2 x[4], unsigned hist, float &vr, float &vi)
{
    const float cc = a.dc_c, aa = a.dc_a;
    cf2 xd[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) xd[s] = (hist & (1u << s)) ? cf2{0.0f, 0.0f} : x[s];
    float br = xd[0].x, bi = xd[0].y;
#pragma unroll
    for (int s = 1; s < 4; ++s) { br = fmaf(br, cc, xd[s].x); bi = fmaf(bi, cc, xd[s].y); }
    br = fmaf(a.dc_cpow[0], dpp_f<0x111, 0xf, true>(0.0f, br), br); bi = fmaf(a.dc_cpow[0], dpp_f<0x111, 0xf, true>(0.0f, bi), bi);   // row_shr:1
    br = fmaf(a.dc_cpow[1], dpp_f<0x112, 0xf, true>(0.0f, br), br); bi = fmaf(a.dc_cpow[1], dpp_f<0x112, 0xf, true>(0.0f, bi), bi);   // row_shr:2
    br = fmaf(a.dc_cpow[2], dpp_f<0x114, 0xf, true>(0.0f, br), br); bi = fmaf(a.dc_cpow[2], dpp_f<0x114, 0xf, true>(0.0f, bi), bi);   // row_shr:4
    br = fmaf(a.dc_cpow[3], dpp_f<0x118, 0xf, true>(0.0f, br), br); bi = fmaf(a.dc_cpow[3], dpp_f<0x118, 0xf, true>(0.0f, bi), bi);   // row_shr:8
    br = fmaf(dl.wa, dpp_f<0x142, 0xa, false>(0.0f, br), br);       bi = fmaf(dl.wa, dpp_f<0x142, 0xa, false>(0.0f, bi), bi);         // row_bcast15 -> rows 1, 3
    br = fmaf(dl.wb, dpp_f<0x143, 0xc, false>(0.0f, br), br);       bi = fmaf(dl.wb, dpp_f<0x143, 0xc, false>(0.0f, bi), bi);         // row_bcast31 -> rows 2, 3
    const float er = dpp_f<0x138, 0xf, true>(0.0f, br), ei = dpp_f<0x138, 0xf, true>(0.0f, bi);                                         // wave_shr:1: exclusive
    const float tr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, br), 63));                            // the chunk's aggregate
    const float ti = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bi), 63));
    float sr = fmaf(dl.pw, vr, er), si = fmaf(dl.pw, vi, ei);         // v[n-1] of the lane's first frame
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float yr = fmaf(-aa, sr, xd[s].x), yi = fmaf(-aa, si, xd[s].y);
        sr = fmaf(cc, sr, xd[s].x); si = fmaf(cc, si, xd[s].y);
        if (!(hist & (1u << s))) { x[s].x = yr; x[s].y = yi; }
    }
    const float c256 = a.dc_cpow[6];
    vr = fmaf(c256, vr, tr); vi = fmaf(c256, vi, ti);
}

} // namespace iqgpu
