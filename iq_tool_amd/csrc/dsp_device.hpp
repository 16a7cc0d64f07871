// dsp_device.hpp -- device-side operator primitives shared by the gfx950 kernels:
// sample_convert unpack / pack (bit-exact with src/sample_convert.c; translation units that include
// this header are compiled with -ffp-contract=off), LIQUID_NCO phasor lookup and mix, and the
// closed-form output index of the fixed-point polyphase resampler.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/iqgpu.h"
#include "kernels.hpp"

namespace iqgpu {

// ============================================================================================
// sample_convert (src/sample_convert.c:75-96, 127-208): normaliser first, then gain, each a
// separately rounded float multiply.
// ============================================================================================
__device__ __forceinline__ float up_s(float v, float norm, float gain)
{
    return __fmul_rn(__fmul_rn(v, norm), gain);
}
__device__ __forceinline__ float up_u(float v, float off, float norm, float gain)
{
    return __fmul_rn(__fmul_rn(__fsub_rn(v, off), norm), gain);
}

// one frame, any format (slow path: tile edges, unaligned calls, rare formats)
__device__ __forceinline__ cf2 unpack_one(const void *raw, int64_t j, int fmt, float gain)
{
    cf2 r;
    switch (fmt) {
    case IQGPU_FMT_CS8: {
        const signed char *p = (const signed char *)raw + 2 * j;
        r.x = up_s((float)p[0], 1.0f / 128.0f, gain); r.y = up_s((float)p[1], 1.0f / 128.0f, gain);
        break; }
    case IQGPU_FMT_CU8: {
        const unsigned char *p = (const unsigned char *)raw + 2 * j;
        r.x = up_u((float)p[0], 127.5f, 1.0f / 128.0f, gain); r.y = up_u((float)p[1], 127.5f, 1.0f / 128.0f, gain);
        break; }
    case IQGPU_FMT_CS16: {
        const short *p = (const short *)raw + 2 * j;
        r.x = up_s((float)p[0], 1.0f / 32768.0f, gain); r.y = up_s((float)p[1], 1.0f / 32768.0f, gain);
        break; }
    case IQGPU_FMT_SC16Q11: {
        const short *p = (const short *)raw + 2 * j;
        r.x = up_s((float)p[0], 1.0f / 2048.0f, gain); r.y = up_s((float)p[1], 1.0f / 2048.0f, gain);
        break; }
    case IQGPU_FMT_CU16: {
        const unsigned short *p = (const unsigned short *)raw + 2 * j;
        r.x = up_u((float)p[0], 32767.5f, 1.0f / 32768.0f, gain); r.y = up_u((float)p[1], 32767.5f, 1.0f / 32768.0f, gain);
        break; }
    case IQGPU_FMT_CS24: {
        const unsigned char *p = (const unsigned char *)raw + 6 * j;
        int a = (int)(((unsigned)p[0] << 8) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 24)) >> 8;
        int b = (int)(((unsigned)p[3] << 8) | ((unsigned)p[4] << 16) | ((unsigned)p[5] << 24)) >> 8;
        r.x = up_s((float)a, 1.0f / 8388608.0f, gain); r.y = up_s((float)b, 1.0f / 8388608.0f, gain);
        break; }
    case IQGPU_FMT_CS32: {
        const int *p = (const int *)raw + 2 * j;
        r.x = (float)__dmul_rn(__dmul_rn((double)p[0], 1.0 / 2147483648.0), (double)gain);
        r.y = (float)__dmul_rn(__dmul_rn((double)p[1], 1.0 / 2147483648.0), (double)gain);
        break; }
    case IQGPU_FMT_CU32: {
        const unsigned *p = (const unsigned *)raw + 2 * j;
        r.x = (float)__dmul_rn(__dmul_rn(__dsub_rn((double)p[0], 2147483647.5), 1.0 / 2147483648.0), (double)gain);
        r.y = (float)__dmul_rn(__dmul_rn(__dsub_rn((double)p[1], 2147483647.5), 1.0 / 2147483648.0), (double)gain);
        break; }
    default: { // IQGPU_FMT_CF32
        const cf2 *p = (const cf2 *)raw + j;
        r.x = __fmul_rn(p->x, gain); r.y = __fmul_rn(p->y, gain);
        break; }
    }
    return r;
}

// four consecutive frames starting at j with one coalesced vector load (aligned fast path)
__device__ __forceinline__ bool unpack_four_fast(const void *raw, int64_t j, int fmt, float gain, cf2 x[4])
{
    switch (fmt) {
    case IQGPU_FMT_CS16: case IQGPU_FMT_SC16Q11: {
        const float norm = (fmt == IQGPU_FMT_CS16) ? 1.0f / 32768.0f : 1.0f / 2048.0f;
        const uint4 v = *(const uint4 *)((const char *)raw + 4 * j);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            x[s].x = up_s((float)(short)(w[s] & 0xffffu), norm, gain);
            x[s].y = up_s((float)(short)(w[s] >> 16), norm, gain);
        }
        return true; }
    case IQGPU_FMT_CU16: {
        const uint4 v = *(const uint4 *)((const char *)raw + 4 * j);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            x[s].x = up_u((float)(w[s] & 0xffffu), 32767.5f, 1.0f / 32768.0f, gain);
            x[s].y = up_u((float)(w[s] >> 16), 32767.5f, 1.0f / 32768.0f, gain);
        }
        return true; }
    case IQGPU_FMT_CU8: {
        const uint2 v = *(const uint2 *)((const char *)raw + 2 * j);
        const unsigned w[2] = {v.x, v.y};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const unsigned h = w[s >> 1] >> ((s & 1) * 16);
            x[s].x = up_u((float)(h & 0xffu), 127.5f, 1.0f / 128.0f, gain);
            x[s].y = up_u((float)((h >> 8) & 0xffu), 127.5f, 1.0f / 128.0f, gain);
        }
        return true; }
    case IQGPU_FMT_CS8: {
        const uint2 v = *(const uint2 *)((const char *)raw + 2 * j);
        const unsigned w[2] = {v.x, v.y};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const unsigned h = w[s >> 1] >> ((s & 1) * 16);
            x[s].x = up_s((float)(signed char)(h & 0xffu), 1.0f / 128.0f, gain);
            x[s].y = up_s((float)(signed char)((h >> 8) & 0xffu), 1.0f / 128.0f, gain);
        }
        return true; }
    case IQGPU_FMT_CF32: {
        const float4 v0 = *(const float4 *)((const char *)raw + 8 * j);
        const float4 v1 = *(const float4 *)((const char *)raw + 8 * j + 16);
        x[0].x = __fmul_rn(v0.x, gain); x[0].y = __fmul_rn(v0.y, gain);
        x[1].x = __fmul_rn(v0.z, gain); x[1].y = __fmul_rn(v0.w, gain);
        x[2].x = __fmul_rn(v1.x, gain); x[2].y = __fmul_rn(v1.y, gain);
        x[3].x = __fmul_rn(v1.z, gain); x[3].y = __fmul_rn(v1.w, gain);
        return true; }
    default:
        return false;
    }
}

// src/sample_convert.c:40-57: scale, +-0.5 by sign, clamp, truncate
__device__ __forceinline__ int pk_signed(float x, float scale, float lo, float hi)
{
    float v = __fmul_rn(x, scale);
    v = (v > 0.0f) ? __fadd_rn(v, 0.5f) : __fsub_rn(v, 0.5f);
    if (v > hi) v = hi;
    if (v < lo) v = lo;
    return (int)v;
}
// src/sample_convert.c:59-73: scale, offset, clamp, +0.5, truncate
__device__ __forceinline__ unsigned pk_unsigned(float x, float scale, float off, float hi)
{
    // (one v_med3_f32 instead of two compare / select pairs: the same value for every finite input -- fminf / fmaxf would add a
    //  canonicalising v_max_f32 each under IEEE mode)
    const float v = __builtin_amdgcn_fmed3f(__fadd_rn(__fmul_rn(x, scale), off), 0.0f, hi);
    return (unsigned)__fadd_rn(v, 0.5f);
}

// one cs16 frame as 32 bits: same result as
// src/sample_convert.c:40-57 for every finite input -- +-0.5 by sign is a copysign (0 gives 0 either
// way), truncation then int16 saturation equals float clamp then truncation.
__device__ __forceinline__ uint32_t pack_cs16(cf2 v)
{
    typedef float f2 __attribute__((ext_vector_type(2)));           // (both components in one packed multiply and one packed add)
    f2 s = f2{v.x, v.y} * f2{32767.0f, 32767.0f};
    s = s + f2{copysignf(0.5f, s.x), copysignf(0.5f, s.y)};
    typedef short s2 __attribute__((ext_vector_type(2)));
    const s2 pk = __builtin_amdgcn_cvt_pk_i16((int)s.x, (int)s.y);
    return __builtin_bit_cast(uint32_t, pk);
}
// cu8 / cs8: one frame as 16 bits (src/sample_convert.c:40-73, the arithmetic of pack_store)
__device__ __forceinline__ uint32_t pack_b8(cf2 v, bool is_unsigned)
{
    if (is_unsigned) return pk_unsigned(v.x, 127.0f, 127.5f, 255.0f) | (pk_unsigned(v.y, 127.0f, 127.5f, 255.0f) << 8);
    return ((unsigned)pk_signed(v.x, 127.0f, -128.0f, 127.0f) & 0xffu) | (((unsigned)pk_signed(v.y, 127.0f, -128.0f, 127.0f) & 0xffu) << 8);
}
// one frame -> out[idx] in any format (src/sample_convert.c:213-309)
__device__ __forceinline__ void pack_store(void *out, int64_t idx, int fmt, cf2 v)
{
    switch (fmt) {
    case IQGPU_FMT_CS16: case IQGPU_FMT_SC16Q11: {
        const float s = (fmt == IQGPU_FMT_CS16) ? 32767.0f : 2048.0f;
        const unsigned a = (unsigned)pk_signed(v.x, s, -32768.0f, 32767.0f) & 0xffffu;
        const unsigned b = (unsigned)pk_signed(v.y, s, -32768.0f, 32767.0f) & 0xffffu;
        ((unsigned *)out)[idx] = a | (b << 16);
        break; }
    case IQGPU_FMT_CU16: {
        const unsigned a = pk_unsigned(v.x, 32767.0f, 32767.5f, 65535.0f);
        const unsigned b = pk_unsigned(v.y, 32767.0f, 32767.5f, 65535.0f);
        ((unsigned *)out)[idx] = a | (b << 16);
        break; }
    case IQGPU_FMT_CS8: {
        const unsigned a = (unsigned)pk_signed(v.x, 127.0f, -128.0f, 127.0f) & 0xffu;
        const unsigned b = (unsigned)pk_signed(v.y, 127.0f, -128.0f, 127.0f) & 0xffu;
        ((unsigned short *)out)[idx] = (unsigned short)(a | (b << 8));
        break; }
    case IQGPU_FMT_CU8: {
        const unsigned a = pk_unsigned(v.x, 127.0f, 127.5f, 255.0f);
        const unsigned b = pk_unsigned(v.y, 127.0f, 127.5f, 255.0f);
        ((unsigned short *)out)[idx] = (unsigned short)(a | (b << 8));
        break; }
    case IQGPU_FMT_CS24: {
        const float fa = __fmul_rn(v.x, 8388607.0f), fb = __fmul_rn(v.y, 8388607.0f);
        int a = (int)((fa > 0.0f) ? __fadd_rn(fa, 0.5f) : __fsub_rn(fa, 0.5f));
        int b = (int)((fb > 0.0f) ? __fadd_rn(fb, 0.5f) : __fsub_rn(fb, 0.5f));
        a = a > 8388607 ? 8388607 : (a < -8388608 ? -8388608 : a);
        b = b > 8388607 ? 8388607 : (b < -8388608 ? -8388608 : b);
        unsigned char *o = (unsigned char *)out + 6 * idx;
        o[0] = (unsigned char)(a & 0xff); o[1] = (unsigned char)((a >> 8) & 0xff); o[2] = (unsigned char)((a >> 16) & 0xff);
        o[3] = (unsigned char)(b & 0xff); o[4] = (unsigned char)((b >> 8) & 0xff); o[5] = (unsigned char)((b >> 16) & 0xff);
        break; }
    case IQGPU_FMT_CS32: {
        const double hi = 2147483647.0, lo = -2147483648.0;
        double a = __dmul_rn((double)v.x, hi), b = __dmul_rn((double)v.y, hi);
        a = (a > 0.0) ? __dadd_rn(a, 0.5) : __dsub_rn(a, 0.5);
        b = (b > 0.0) ? __dadd_rn(b, 0.5) : __dsub_rn(b, 0.5);
        a = a > hi ? hi : (a < lo ? lo : a);
        b = b > hi ? hi : (b < lo ? lo : b);
        ((int2 *)out)[idx] = make_int2((int)a, (int)b);
        break; }
    case IQGPU_FMT_CU32: {
        const double hi = 4294967295.0;
        double a = __dadd_rn(__dmul_rn((double)v.x, 2147483647.0), 2147483647.5);
        double b = __dadd_rn(__dmul_rn((double)v.y, 2147483647.0), 2147483647.5);
        a = a > hi ? hi : (a < 0.0 ? 0.0 : a);
        b = b > hi ? hi : (b < 0.0 ? 0.0 : b);
        ((uint2 *)out)[idx] = make_uint2((unsigned)__dadd_rn(a, 0.5), (unsigned)__dadd_rn(b, 0.5));
        break; }
    default: // IQGPU_FMT_CF32: memcpy (src/sample_convert.c:301-303)
        ((cf2 *)out)[idx] = v;
        break;
    }
}

// LIQUID_NCO phasor of phase theta: table index = rounded top 10 bits (SPEC B.4)
__device__ __forceinline__ cf2 nco_phasor(const cf2 *tab, uint32_t theta)
{
    return tab[(theta + (1u << 21)) >> 22];
}
// y = x * (c + j s) for mode +1, x * (c - j s) for mode -1 (src/frequency_shift.c:91-95)
__device__ __forceinline__ cf2 nco_mix(cf2 x, cf2 cs, int mode)
{
    const float s = (mode > 0) ? cs.y : -cs.y, c = cs.x;
    cf2 y;
    y.x = fmaf(x.x, c, -(x.y * s));
    y.y = fmaf(x.x, s, x.y * c);
    return y;
}

__device__ __forceinline__ uint64_t first_k_at(uint64_t target, uint64_t phi0, uint32_t step)
{
    return target > phi0 ? (target - phi0 + (uint64_t)step - 1) / (uint64_t)step : 0;
}

} // namespace iqgpu
