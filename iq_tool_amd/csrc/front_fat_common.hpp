// front_fat_common.hpp -- what k_front_fat (front_fat.hip: 8 half-band outputs per lane, 8 waves per CU) and k_front_mid
// (front_mid.hip: 6 per lane, 12 waves) share: float2 arithmetic the compiler can schedule, aligned LDS accesses, the tap planes
// of the shifted-tap polyphase slots and the slots themselves.
#pragma once
#include "front_tiles.hpp"

namespace iqgpu {

// polyphase taps: per arm R = 0 0 tap13 .. tap0 0 0 (18 floats); a slot whose output sits d in {0, 1, 2} samples past the slot's
// first possible one multiplies its 16-sample window by T_d[w] = R[w + 2 - d].  17 planes of 8 bytes per arm, 2056 bytes apart
// (reads of one slot are 4112 apart: neither ds_read2_b64 nor ds_read2st64_b64 can fuse them into a half-rate instruction);
// plane pi holds (R[16 - pi], R[17 - pi]) of every arm, so that pair q of T_d sits in plane d + 14 - 2 q: the address is linear
// in the position and, with arm a in slot a of a plane, in the arm too -- tap_row() is then a shift, an add and a shift-add on
// the top bits of the phase.  Which lanes of a half-wave meet in a bank pair depends on the step alone, so the placement is
// chosen per chain (front_tap_fold(), front_mid.hip): slot a, or slot a ^ (a >> 5) (two more instructions) for steps whose
// lanes would otherwise walk the arms in strides of 16 or 32 (s = 1.625: 15.5 modelled cycles per read against 4.0).
// (Rounds 2 - 3 always folded, and picked planes by the parity of d: 11 VALU instructions per slot, one of them a quarter-rate
// integer multiply; a build without ANY conflict in the gather measured -2.4 %, profiles/r03_ab.md.)
constexpr int kFTapPlaneB = 2048 + 8;
constexpr int kFTapPlanes = 17;
constexpr int kFTapLds = (kFTapPlanes * kFTapPlaneB + 15) / 16 * 16;
__host__ __device__ constexpr unsigned tap_pair_off(int q) { return (unsigned)((14 - 2 * q) * kFTapPlaneB); }   // pair q of a slot, from tap_row()

// fills the planes from the [256][16] table of the chain (all threads of the workgroup)
__device__ __forceinline__ void fill_tap_planes(float *s_tap, const float *arb_table, const int tid, const int nthreads, const bool fold)
{
    for (int i = tid; i < 256 * kFTapPlanes; i += nthreads) {           // R[k] = tap[15 - k] for k = 2 .. 15, else 0
        const int arm = i & 255, pl = i >> 8, row = fold ? (arm ^ (arm >> 5)) : arm;
        const int k0 = 16 - pl;
        const float r0 = (k0 >= 2 && k0 < 16) ? arb_table[arm * 16 + 15 - k0] : 0.0f;
        const float r1 = (k0 + 1 >= 2 && k0 + 1 < 16) ? arb_table[arm * 16 + 14 - k0] : 0.0f;
        float *d = (float *)((char *)s_tap + pl * kFTapPlaneB + row * 8);
        d[0] = r0; d[1] = r1;
    }
}
// LDS address of plane 0's entry for an output with phase P from the lane's first sample, in the slot whose first possible sample
// is LOJ: position p = P >> 24, arm = the next 8 bits, d = p - LOJ in {0, 1, 2}; entry = d * 2056 + slot(arm) * 8 =
// 8 * (x + p) - 2056 * LOJ with x = P >> 16 = 256 p + arm  (unsigned arithmetic: the constant part may wrap, the sum does not);
// folded: x ^ bits 5 .. 7 of the arm in x's place
template <bool FOLD>
__device__ __forceinline__ unsigned tap_row(const unsigned tap_lds, const uint32_t P, const int LOJ)
{
    uint32_t x = P >> 16;
    if (FOLD) x ^= (P >> 21) & 7u;
    return ((x + (P >> 24)) << 3) + (tap_lds - (unsigned)(LOJ * kFTapPlaneB));
}

__device__ __forceinline__ v2f fma2(float t, v2f x, v2f acc) { return __builtin_elementwise_fma(v2f{t, t}, x, acc); }
__device__ __forceinline__ v2f mul2(float t, v2f x) { return v2f{t, t} * x; }
__device__ __forceinline__ float4 ldq(const char *p) { return *(const float4 *)__builtin_assume_aligned(p, 16); }
__device__ __forceinline__ void stq(char *p, float4 v) { *(float4 *)__builtin_assume_aligned(p, 16) = v; }
// a loaded register that no FMA touches still counts as used: hipcc would otherwise trim the 16-byte read and re-chunk it
template <typename T> __device__ __forceinline__ void keep(const T &v) { asm volatile("" :: "v"(v)); }

// One polyphase slot: its output sits at half-band sample LOJ + d of the lane's NL, d in {0, 1, 2}; t = the arm's taps shifted
// by d between zeros, T[w] = tap[13 + d - w] (0 outside 0 .. 13), w = 0 .. 15 <-> sample LOJ - 13 + w.  Sum in ascending tap
// order = descending w, started by the first product (k_front_s1's order; the zero taps in front leave +-0).
// Hw[i] = sample i - 14 (the 13 in front of the lane's own; Hw[0] unused), own[m] = sample m.
template <int NL, int LOJ> struct PpGeom { static constexpr int HI = LOJ + 2 < NL - 1 ? LOJ + 2 : NL - 1, W = HI - LOJ + 14; };
// step wi of a slot's chain (wi counts down from 15; a slot whose window is shorter starts later)
// SEL (k_front_p0): the tap is picked out of its register pair by the instruction's op_sel bits, written out -- left to the compiler,
// one of the five slots of a step had its 16 taps copied into {t, t} pairs first (32 v_mov_b32 per step)
template <int NL, int LOJ, bool SEL = false>
__device__ __forceinline__ void pp_step(const int wi, const v2f (&Hw)[14], const v2f (&own)[NL], const v2f (&t)[8], v2f &y)
{
    constexpr int W = PpGeom<NL, LOJ>::W;
    if (wi >= W) return;
    const int m = LOJ - 13 + wi;
    const v2f h = m < 0 ? Hw[m + 14] : own[m];
    if (SEL) {
        const v2f tp = t[wi >> 1];
        if (wi == W - 1) {
            if (wi & 1) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(y) : "v"(tp), "v"(h));
            else        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(y) : "v"(tp), "v"(h));
        } else {
            if (wi & 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(y) : "v"(tp), "v"(h));
            else        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(y) : "v"(tp), "v"(h));
        }
        return;
    }
    const float tw = (wi & 1) ? t[wi >> 1].y : t[wi >> 1].x;
    y = wi == W - 1 ? mul2(tw, h) : fma2(tw, h, y);
}
// one slot by itself (a dependent chain: the other waves of the SIMD fill its issue slots)
template <int NL, int LA>
__device__ __forceinline__ void pp_slot1(const v2f (&Hw)[14], const v2f (&own)[NL], const v2f (&ta)[8], v2f &ya)
{
#pragma unroll
    for (int wi = 15; wi >= 0; --wi) pp_step<NL, LA>(wi, Hw, own, ta, ya);
}
// two / three slots side by side: their chains are independent, so that no FMA waits for the one before it
template <int NL, int LA, int LB, bool SEL = false>
__device__ __forceinline__ void pp_slots2(const v2f (&Hw)[14], const v2f (&own)[NL], const v2f (&ta)[8], const v2f (&tb)[8], v2f &ya, v2f &yb)
{
#pragma unroll
    for (int wi = 15; wi >= 0; --wi) { pp_step<NL, LA, SEL>(wi, Hw, own, ta, ya); pp_step<NL, LB, SEL>(wi, Hw, own, tb, yb); }
}
template <int NL, int LA, int LB, int LC, bool SEL = false>
__device__ __forceinline__ void pp_slots3(const v2f (&Hw)[14], const v2f (&own)[NL], const v2f (&ta)[8], const v2f (&tb)[8], const v2f (&tc)[8],
                                          v2f &ya, v2f &yb, v2f &yc)
{
#pragma unroll
    for (int wi = 15; wi >= 0; --wi) {
        pp_step<NL, LA, SEL>(wi, Hw, own, ta, ya); pp_step<NL, LB, SEL>(wi, Hw, own, tb, yb); pp_step<NL, LC, SEL>(wi, Hw, own, tc, yc);
    }
}


} // namespace iqgpu
