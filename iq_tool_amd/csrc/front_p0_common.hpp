// front_p0_common.hpp -- what the output-major polyphase kernels share: k_front_p0 (front_p0.hip: chains without a half-band stage) and
// k_p0fft16 (fftconv.hip, round 6: the same polyphase steps computed straight into the overlap-save window of the user filter behind
// the resampler).  The unpack of one window sample from a lane's raw words, and the step classes.
#pragma once
#include "front_fat_common.hpp"

namespace iqgpu {

constexpr int kP0Step = 320;                                // outputs per step of a wave: five per lane

// one window sample from the lane's raw words (frame i of the window; BPS = 2: two frames per word)
template <int FMT>
__device__ __forceinline__ v2f p0_unpack(const uint32_t *r, const int i)
{
    if (FMT == IQGPU_FMT_CU8) {
        // ((float)u - 127.5) * (1 / 128) (src/sample_convert.c:75-96; gain 1): both steps are exact in float, and so is
        // u * 2^-7 - 127.5 * 2^-7 in one fused multiply-add -- the same value, one packed instruction per frame
        const uint32_t w = r[i >> 1];
        const v2f u = (i & 1) ? v2f{(float)((w >> 16) & 0xffu), (float)(w >> 24)} : v2f{(float)(w & 0xffu), (float)((w >> 8) & 0xffu)};
        return __builtin_elementwise_fma(u, v2f{1.0f / 128.0f, 1.0f / 128.0f}, v2f{-127.5f / 128.0f, -127.5f / 128.0f});
    } else if (FMT == IQGPU_FMT_CS8) {
        const uint32_t h = r[i >> 1] >> (16 * (i & 1));
        return v2f{(float)(signed char)(h & 0xffu) * (1.0f / 128.0f), (float)(signed char)((h >> 8) & 0xffu) * (1.0f / 128.0f)};
    } else {
        return v2f{(float)(short)(r[i] & 0xffffu) * (1.0f / 32768.0f), (float)(short)(r[i] >> 16) * (1.0f / 32768.0f)};
    }
}

// step classes of five slots per lane: 1 <= s = step / 2^24 < 2, by (lo_2, lo_3, lo_4) = floor(2 s), floor(3 s), floor(4 s).  A lane's five
// outputs span less than 4 s + 1 < 9 samples behind its first (own[9]); the slots' shifted tap rows cover position offsets 0 and 1
// past lo_j.  (Until late round 5: 1.6 <= s only -- the bound of the eight-samples-per-lane kernel this one grew out of, which the
// output-major form does not have: a 2.048 MS/s capture to the cu8-nrsc5 preset's 1.488375 MS/s has s = 1.376.)
inline bool p0_class(uint32_t step, int *l3, int *l4, int *l2 = nullptr)
{
    const uint64_t one = (uint64_t)1 << 24;
    if ((uint64_t)step >= 2 * one || (uint64_t)step < one + one / 64) return false;     // (s >= 1.016: clear of the no-resampling edge)
    const int k2 = (int)(((uint64_t)step * 2) >> 24);
    *l3 = (int)(((uint64_t)step * 3) >> 24); *l4 = (int)(((uint64_t)step * 4) >> 24);
    if (l2) *l2 = k2;
    if (k2 == 3) return (*l3 == 4 && *l4 == 6) || (*l3 == 5 && *l4 == 6) || (*l3 == 5 && *l4 == 7);
    return k2 == 2 && ((*l3 == 3 && *l4 == 4) || (*l3 == 3 && *l4 == 5) || (*l3 == 4 && *l4 == 5));
}


} // namespace iqgpu
