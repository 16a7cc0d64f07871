// front_wave.hip -- k_front_s1: the wave-autonomous fast path of the front kernel for chains with
// ONE half-band stage (0.25 <= r < 0.5: the NRSC-5 preset 2.4 MS/s -> 744.1875 kS/s is the
// headline case).  Same arithmetic and stream bookkeeping as k_front (kernels.hip); what differs is
// the mapping onto CDNA4:
//
//   * one wavefront (64 lanes) owns a contiguous run of 512-sample tiles and carries every piece of
//     state itself (stage windows in its private LDS slice, next output index / phase in SGPRs), so
//     after the table load there is no workgroup barrier at all: 12 waves per CU drift apart and
//     cover each other's HBM / LDS latency;
//   * raw frames arrive by one coalesced 16-byte load per lane per 256 frames, issued one tile
//     ahead (register prefetch);
//   * the NCO-mixed samples are written to LDS split into even / odd streams in rows of 4 cf32
//     padded to 48 bytes (an odd number of 16-byte slots), so that a lane that owns 4 consecutive
//     half-band outputs reads its 24-sample window with 12 conflict-free ds_read_b128 at constant
//     offsets and keeps it in registers (20 taps x 4 outputs x 2 components of v_fma with the tap
//     in an SGPR);
//   * the polyphase stage gives each lane the 4 half-band samples it just produced: the output
//     (if any) that falls on each of them is found in closed form from the 24-bit phase, its 14
//     taps come from the 256-arm table in LDS, the 18-sample window again sits in registers.
#include <hip/hip_runtime.h>

#include "../../include/iqgpu.h"
#include "dsp_device.hpp"
#include "kernels.hpp"

namespace iqgpu {

constexpr int kRowB = 48;                                   // 4 cf32 + 16 B pad
constexpr int kXRows = 5 + 64;                              // 20 history + 256 samples per parity stream
constexpr int kHRows = 4 + 64;                              // 16 history + 256 half-band outputs
constexpr int kWaveLds = (2 * kXRows + kHRows) * kRowB;     // 9888 B per wave
constexpr int kTabLds = 1024 * 8 + 256 * 14 * 4;            // NCO {cos,sin} + polyphase taps [256][14]

size_t front_s1_lds_bytes() { return (size_t)kTabLds + (size_t)kWaves * kWaveLds; }

struct RawChunk { uint32_t w[8]; };

template <int BPS>
__device__ __forceinline__ void load_chunk(const void *raw, int64_t j, RawChunk &r)
{
    const char *p = (const char *)raw + (int64_t)BPS * j;
    if (BPS == 4) {
        const uint4 v = *(const uint4 *)p;
        r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
    } else if (BPS == 2) {
        const uint2 v = *(const uint2 *)p;
        r.w[0] = v.x; r.w[1] = v.y;
    } else {
        const uint4 v0 = *(const uint4 *)p, v1 = *(const uint4 *)(p + 16);
        r.w[0] = v0.x; r.w[1] = v0.y; r.w[2] = v0.z; r.w[3] = v0.w;
        r.w[4] = v1.x; r.w[5] = v1.y; r.w[6] = v1.z; r.w[7] = v1.w;
    }
}

// four frames from prefetched words; arithmetic identical to unpack_one (src/sample_convert.c:75-96)
template <int BPS>
__device__ __forceinline__ void unpack_chunk(const RawChunk &r, int fmt, float gain, cf2 x[4])
{
    if (BPS == 4) {
        if (fmt == IQGPU_FMT_CU16) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                x[s].x = up_u((float)(r.w[s] & 0xffffu), 32767.5f, 1.0f / 32768.0f, gain);
                x[s].y = up_u((float)(r.w[s] >> 16), 32767.5f, 1.0f / 32768.0f, gain);
            }
        } else {
            const float norm = (fmt == IQGPU_FMT_CS16) ? 1.0f / 32768.0f : 1.0f / 2048.0f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                x[s].x = up_s((float)(short)(r.w[s] & 0xffffu), norm, gain);
                x[s].y = up_s((float)(short)(r.w[s] >> 16), norm, gain);
            }
        }
    } else if (BPS == 2) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const unsigned h = r.w[s >> 1] >> ((s & 1) * 16);
            if (fmt == IQGPU_FMT_CU8) {
                x[s].x = up_u((float)(h & 0xffu), 127.5f, 1.0f / 128.0f, gain);
                x[s].y = up_u((float)((h >> 8) & 0xffu), 127.5f, 1.0f / 128.0f, gain);
            } else {
                x[s].x = up_s((float)(signed char)(h & 0xffu), 1.0f / 128.0f, gain);
                x[s].y = up_s((float)(signed char)((h >> 8) & 0xffu), 1.0f / 128.0f, gain);
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            x[s].x = __uint_as_float(r.w[2 * s]) * gain;
            x[s].y = __uint_as_float(r.w[2 * s + 1]) * gain;
        }
    }
}

// smallest q with q * d >= x, for quotients below 2^9 (x < 2^33, d > 2^24)
__device__ __forceinline__ uint32_t ceil_div_small(uint64_t x, uint32_t d, float inv_d)
{
    uint32_t q = (uint32_t)((float)x * inv_d);
    while ((uint64_t)q * d < x) ++q;
    while (q > 0 && (uint64_t)(q - 1) * d >= x) --q;
    return q;
}

__device__ __forceinline__ float4 ld4(const char *p) { return *(const float4 *)p; }

// BPS: bytes per input frame on the vector-load path (2, 4, 8); 0 = scalar loads only
template <int BPS>
__global__ __launch_bounds__(kWThreads) void k_front_s1(const FrontArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    cf2   *s_nco = (cf2 *)smem;
    float *s_arb = (float *)(smem + 1024 * 8);
    char *XE = (char *)smem + kTabLds + wave * kWaveLds;
    char *XO = XE + kXRows * kRowB;
    char *HB = XO + kXRows * kRowB;

    if (a.nco_mode != 0 || a.pnco_mode != 0)
        for (int i = tid; i < 1024; i += kWThreads) s_nco[i] = a.nco_tab[i];
    for (int i = tid; i < 256 * 14; i += kWThreads) s_arb[i] = a.arb_table[(i / 14) * 16 + (i % 14)];
    for (int i = lane; i < kWaveLds / 16; i += 64) ((float4 *)XE)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    const int64_t gw = (int64_t)blockIdx.x * kWaves + wave;
    const int64_t t_emit0 = gw * a.w_tiles_per_wave;
    if (t_emit0 >= a.w_total_tiles) return;
    int64_t t_emit1 = t_emit0 + a.w_tiles_per_wave;
    if (t_emit1 > a.w_total_tiles) t_emit1 = a.w_total_tiles;
    const int64_t t_begin = t_emit0 - a.w_warm_tiles;

    if (gw == 0 && a.frames_in < (int64_t)a.hist_cap) {
        const int keep = a.hist_cap - (int)a.frames_in;
        for (int i = lane; i < keep; i += 64) a.hist_out[i] = a.hist_in[i + (int)a.frames_in];
    }

    // output bookkeeping (uniform): first output whose half-band sample is >= this wave's first one
    const uint32_t step = a.step;
    const float inv_step = 1.0f / (float)step;
    uint64_t k_tile0 = first_k_at((uint64_t)(t_emit0 * 256) << 24, a.phi0, step);
    uint64_t delta0 = a.phi0 + k_tile0 * (uint64_t)step - ((uint64_t)(t_emit0 * 256) << 24);   // < step

    const bool vec_ok = (BPS != 0) && a.raw_aligned;
    RawChunk nxt[2];
    bool have_next = false;
    {
        const int64_t j0 = t_begin * kWTile - a.rem0;
        if (vec_ok && j0 >= 0 && j0 + kWTile <= a.frames_in && (((j0 * BPS) & 15) == 0)) {
            load_chunk<BPS ? BPS : 4>(a.raw, j0 + 4 * lane, nxt[0]);
            load_chunk<BPS ? BPS : 4>(a.raw, j0 + 256 + 4 * lane, nxt[1]);
            have_next = true;
        }
    }

    for (int64_t t = t_begin; t < t_emit1; ++t) {
        const int64_t i0 = t * kWTile;
        const int64_t j0 = i0 - a.rem0;
        const bool emit = t >= t_emit0;
        const bool fast = have_next;
        RawChunk cur[2];
        if (fast) { cur[0] = nxt[0]; cur[1] = nxt[1]; }
        {   // prefetch the next tile's frames
            const int64_t jn = j0 + kWTile;
            have_next = false;
            if (t + 1 < t_emit1 && vec_ok && jn >= 0 && jn + kWTile <= a.frames_in && (((jn * BPS) & 15) == 0)) {
                load_chunk<BPS ? BPS : 4>(a.raw, jn + 4 * lane, nxt[0]);
                load_chunk<BPS ? BPS : 4>(a.raw, jn + 256 + 4 * lane, nxt[1]);
                have_next = true;
            }
        }

        // ------------------------------------------------------------ pointwise -> LDS
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int l4 = 256 * c + 4 * lane;
            const int64_t j = j0 + l4;
            cf2 x[4];
            bool is_hist[4] = {false, false, false, false};
            bool is_new[4] = {true, true, true, true};
            if (fast) {
                unpack_chunk<BPS ? BPS : 4>(cur[c], a.in_fmt, a.gain, x);
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int64_t js = j + s;
                    if (js < 0) {
                        const int64_t h = (int64_t)a.hist_cap + js;
                        x[s] = (h >= 0) ? a.hist_in[h] : cf2{0.0f, 0.0f};
                        is_hist[s] = true; is_new[s] = false;
                    } else if (js >= a.frames_in) {
                        x[s] = cf2{0.0f, 0.0f};
                        is_new[s] = false;
                    } else {
                        x[s] = unpack_one(a.raw, js, a.in_fmt, a.gain);
                    }
                }
            }
            if (a.iq_enable) {
#pragma unroll
                for (int s = 0; s < 4; ++s) if (!is_hist[s]) {
                    const float re = x[s].x;
                    x[s].x = re * a.iq_magp1;
                    x[s].y = fmaf(a.iq_phase, re, x[s].y);
                }
            }
            if (a.nco_mode != 0) {
                uint32_t th = a.nco_theta0 + (uint32_t)(i0 + l4) * a.nco_dtheta;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (!is_hist[s]) x[s] = nco_mix(x[s], nco_phasor(s_nco, th), a.nco_mode);
                    th += a.nco_dtheta;
                }
            }
            if (emit && j + 4 > a.frames_in - (int64_t)a.hist_cap) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int64_t back = a.frames_in - (j + s);
                    if (is_new[s] && back <= (int64_t)a.hist_cap) a.hist_out[(int64_t)a.hist_cap - back] = x[s];
                }
            }
            const int off = (5 + 32 * c + (lane >> 1)) * kRowB + (lane & 1) * 16;
            *(float4 *)(XE + off) = make_float4(x[0].x, x[0].y, x[2].x, x[2].y);
            *(float4 *)(XO + off) = make_float4(x[1].x, x[1].y, x[3].x, x[3].y);
        }
        __builtin_amdgcn_wave_barrier();

        // ------------------------------------------------------------ half-band: 4 outputs per lane
        {
            const char *we = XE + lane * kRowB;
            cf2 E[24];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const float4 v0 = ld4(we + r * kRowB), v1 = ld4(we + r * kRowB + 16);
                E[4 * r + 0] = cf2{v0.x, v0.y}; E[4 * r + 1] = cf2{v0.z, v0.w};
                E[4 * r + 2] = cf2{v1.x, v1.y}; E[4 * r + 3] = cf2{v1.z, v1.w};
            }
            const char *wo = XO + lane * kRowB;
            const float4 o0 = ld4(wo + 2 * kRowB + 16), o1 = ld4(wo + 3 * kRowB);
            float ar[4] = {0.5f * o0.x, 0.5f * o0.z, 0.5f * o1.x, 0.5f * o1.z};
            float ai[4] = {0.5f * o0.y, 0.5f * o0.w, 0.5f * o1.y, 0.5f * o1.w};
#pragma unroll
            for (int q = 0; q < 20; ++q) {
                const float h = a.hb0[q];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ar[r] = fmaf(h, E[20 + r - q].x, ar[r]);
                    ai[r] = fmaf(h, E[20 + r - q].y, ai[r]);
                }
            }
            char *ph = HB + (lane + 4) * kRowB;
            *(float4 *)ph = make_float4(ar[0], ai[0], ar[1], ai[1]);
            *(float4 *)(ph + 16) = make_float4(ar[2], ai[2], ar[3], ai[3]);
        }
        __builtin_amdgcn_wave_barrier();

        // ------------------------------------------------------------ polyphase + pack
        if (emit) {
            const int64_t q_tile0 = t * 256;
            if (q_tile0 < a.n_groups) {
                const char *wh = HB + lane * kRowB;
                cf2 H[18];
                {
                    const float4 v = ld4(wh + 16);
                    H[0] = cf2{v.x, v.y}; H[1] = cf2{v.z, v.w};
                }
#pragma unroll
                for (int r = 1; r < 5; ++r) {
                    const float4 v0 = ld4(wh + r * kRowB), v1 = ld4(wh + r * kRowB + 16);
                    H[4 * r - 2] = cf2{v0.x, v0.y}; H[4 * r - 1] = cf2{v0.z, v0.w};
                    H[4 * r + 0] = cf2{v1.x, v1.y}; H[4 * r + 1] = cf2{v1.z, v1.w};
                }
                // first output at or after this lane's first half-band sample (4*lane)
                const uint64_t tgt = (uint64_t)(4 * lane) << 24;
                uint32_t n0 = 0;
                if (tgt > delta0) n0 = ceil_div_small(tgt - delta0, step, inv_step);
                uint32_t Pl = (uint32_t)(delta0 + (uint64_t)n0 * step - tgt);     // phase relative to 4*lane
                uint32_t kk = n0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool hit = (Pl >> 24) == (uint32_t)r;
                    const int arm = (int)((Pl >> 16) & 255u);
                    const float2 *tp = (const float2 *)(s_arb + arm * 14);
                    float yr = 0.0f, yi = 0.0f;
#pragma unroll
                    for (int n2 = 0; n2 < 7; ++n2) {
                        const float2 tt = tp[n2];
                        yr = fmaf(tt.x, H[14 + r - 2 * n2].x, yr); yi = fmaf(tt.x, H[14 + r - 2 * n2].y, yi);
                        yr = fmaf(tt.y, H[13 + r - 2 * n2].x, yr); yi = fmaf(tt.y, H[13 + r - 2 * n2].y, yi);
                    }
                    if (hit && (q_tile0 + 4 * lane + r) < a.n_groups) {
                        const uint64_t k = k_tile0 + kk;
                        cf2 y{yr, yi};
                        if (a.pnco_mode != 0)
                            y = nco_mix(y, nco_phasor(s_nco, a.pnco_theta0 + (uint32_t)k * a.pnco_dtheta), a.pnco_mode);
                        pack_store(a.out, (int64_t)k, a.out_fmt, y);
                    }
                    if (hit) { Pl += step; ++kk; }
                }
            }
            // outputs of this tile: those with phase below 256 * 2^24
            const uint32_t n_tile = ceil_div_small(((uint64_t)1 << 32) - delta0, step, inv_step);
            k_tile0 += n_tile;
            delta0 = delta0 + (uint64_t)n_tile * step - ((uint64_t)1 << 32);
        }

        // ------------------------------------------------------------ slide the windows
        {
            float4 ve, vo, vh;
            if (lane < 15) { ve = ld4(XE + 64 * kRowB + lane * 16); vo = ld4(XO + 64 * kRowB + lane * 16); }
            if (lane < 12) vh = ld4(HB + 64 * kRowB + lane * 16);
            __builtin_amdgcn_wave_barrier();
            if (lane < 15) { *(float4 *)(XE + lane * 16) = ve; *(float4 *)(XO + lane * 16) = vo; }
            if (lane < 12) *(float4 *)(HB + lane * 16) = vh;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

hipError_t launch_front_s1(const FrontArgs &a, hipStream_t s)
{
    const size_t lds = front_s1_lds_bytes();
    const int64_t n_sub = (a.w_total_tiles + a.w_tiles_per_wave - 1) / a.w_tiles_per_wave;
    const unsigned grid = (unsigned)((n_sub + kWaves - 1) / kWaves);
    int cls;
    switch (a.in_fmt) {
    case IQGPU_FMT_CS8: case IQGPU_FMT_CU8: cls = 2; break;
    case IQGPU_FMT_CS16: case IQGPU_FMT_CU16: case IQGPU_FMT_SC16Q11: cls = 4; break;
    case IQGPU_FMT_CF32: cls = 8; break;
    default: cls = 0; break;
    }
#define IQGPU_LAUNCH_S1(BPS)                                                                                          \
    do {                                                                                                              \
        hipError_t e = hipFuncSetAttribute((const void *)k_front_s1<BPS>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                           (int)lds);                                                                 \
        if (e != hipSuccess) return e;                                                                                \
        hipLaunchKernelGGL(k_front_s1<BPS>, dim3(grid), dim3(kWThreads), lds, s, a);                                  \
    } while (0)
    if (cls == 2) IQGPU_LAUNCH_S1(2);
    else if (cls == 4) IQGPU_LAUNCH_S1(4);
    else if (cls == 8) IQGPU_LAUNCH_S1(8);
    else IQGPU_LAUNCH_S1(0);
#undef IQGPU_LAUNCH_S1
    return hipGetLastError();
}

} // namespace iqgpu
