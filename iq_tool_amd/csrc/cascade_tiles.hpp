// cascade_tiles.hpp -- device side of the wave-autonomous half-band cascade (k_cascade, cascade_wave.hip; the edge waves of
// k_front_s2, front_s2.hip, run the same tile routine): stage windows, the skewed tile loop casc_tiles, the LDS geometry.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>

#include "../../include/iqgpu.h"
#include "dsp_device.hpp"
#include "kernels.hpp"
#include "wave_common.hpp"

namespace iqgpu {

// 16-byte LDS accesses of this file: every slice, row and array starts on a 16-byte boundary, but the wave's slice sits at a
// run-time multiple of casc_wave_lds, which hides that from the compiler -- without the hint it emits ds_read2_b64 /
// ds_write2_b64 pairs (two 8-byte accesses at a 16-byte lane stride: two-way bank conflicts) instead of b128
__device__ __forceinline__ float4 ld4a(const char *p) { return *(const float4 *)__builtin_assume_aligned(p, 16); }
// ... and every component of a window load counts as used: left alone, the compiler trims a 16-byte load to the dwords the
// taps touch and re-chunks the rest into ds_read2_b32 / unaligned ds_read2_b64 (slow, and conflict-prone at a 16-byte lane stride)
template <typename T> __device__ __forceinline__ void keep(const T &v) { asm volatile("" :: "v"(v)); }
__device__ __forceinline__ v2f ld2(const char *p) { const float2 v = *(const float2 *)p; return v2f{v.x, v.y}; }
__device__ __forceinline__ void st4a(char *p, float4 v) { *(float4 *)__builtin_assume_aligned(p, 16) = v; }

__host__ __device__ constexpr int casc_hist_rows(int m) { return (2 * m - 1 + 3) / 4; }   // older rows a lane reads

// One stage: 4 outputs per lane for lanes < n_act.
//   out j = 0.5 O[j - M] + sum_q h[q] E[j - q], q < 2M   (E[i] = x[2i], O[i] = x[2i+1]; h pre-scaled by 0.5)
// XE / XO: the stage's input rows; row (H + l) holds samples 4l .. 4l+3 of the tile, rows 0 .. H-1 the history.
template <int M> struct CascWin0 { v2f E[4 * (casc_hist_rows(M) + 1)]; v2f O[8]; };
template <int M>
__device__ __forceinline__ void casc_stage_load(const char *XE, const char *XO, int lane, CascWin0<M> &wn)
{
    // Every load below is used in full (the taps touch E[4H - 2M + 1 .. 4H + 3] and four consecutive O's): a load with unused
    // components is trimmed by the compiler and re-chunked into ds_read2_b32 / unaligned ds_read2_b64, which conflict
    constexpr int H = casc_hist_rows(M);
    constexpr int PS = plane_stride(H + 64 + 1);      // two planes of 16-byte half rows (wave_common.hpp)
    constexpr int lo = 4 * H - 2 * M + 1;             // first window entry a tap touches (3 for M = 3 and M = 5)
    const char *we = XE + lane * 16;
#pragma unroll
    for (int r = 0; r <= H; ++r) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {                 // half row h: entries 4r + 2h, 4r + 2h + 1
            const int e0 = 4 * r + 2 * h;
            const char *q = we + r * 16 + h * PS;
            // (a half row with one used entry is read whole and the other entry declared used: an 8-byte read in front of
            //  aligned 16-byte ones makes the vectoriser re-chunk the whole run at the 8-byte phase)
            if (e0 + 1 >= lo) { const float4 v = ld4a(q); wn.E[e0] = v2f{v.x, v.y}; wn.E[e0 + 1] = v2f{v.z, v.w}; if (e0 < lo) keep(wn.E[e0]); }
        }
    }
    // centre taps: O[4l + i - M] = window index 4H + i - M  ->  rows r0, r0 + 1; entries (c0 & 3) .. (c0 & 3) + 3 of O[0 .. 7]
    constexpr int c0 = 4 * H - M;                     // window index of i = 0
    constexpr int r0 = c0 / 4;
    constexpr int u0 = c0 & 3, u1 = u0 + 3;           // used entries u0 .. u1
    const char *wo = XO + (lane + r0) * 16;
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {                  // pair pr: entries 2 pr, 2 pr + 1 = row r0 + pr / 2, plane pr & 1
        const char *q = wo + (pr >> 1) * 16 + (pr & 1) * PS;
        const bool a0 = 2 * pr >= u0 && 2 * pr <= u1, a1 = 2 * pr + 1 >= u0 && 2 * pr + 1 <= u1;
        if (a0 || a1) {
            const float4 v = ld4a(q); wn.O[2 * pr] = v2f{v.x, v.y}; wn.O[2 * pr + 1] = v2f{v.z, v.w};
            if (!a0) keep(wn.O[2 * pr]);
            if (!a1) keep(wn.O[2 * pr + 1]);
        }
    }
}
template <int M>
__device__ __forceinline__ void casc_stage_fma(const CascWin0<M> &wn, const float *taps_sgpr, v2f y[4])
{
    constexpr int H = casc_hist_rows(M);
    constexpr int c0 = 4 * H - M;
    const v2f *hbp = (const v2f *)taps_sgpr;          // M SGPR pairs {h[2i], h[2i+1]}
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const v2f o = wn.O[(c0 & 3) + i];
        y[i] = v2f{0.5f * o.x, 0.5f * o.y};
    }
#pragma unroll
    for (int q2 = 0; q2 < M; ++q2) {
        const v2f tp = hbp[q2];
#pragma unroll
        for (int i = 0; i < 4; ++i) pk_fma_lo_s(y[i], tp, wn.E[4 * H + i - 2 * q2]);
#pragma unroll
        for (int i = 0; i < 4; ++i) pk_fma_hi_s(y[i], tp, wn.E[4 * H + i - 2 * q2 - 1]);
    }
}
template <int M>
__device__ __forceinline__ void casc_stage(const char *XE, const char *XO, int lane, const float *taps_sgpr, v2f y[4])
{
    CascWin0<M> wn;
    casc_stage_load<M>(XE, XO, lane, wn);
    casc_stage_fma<M>(wn, taps_sgpr, y);
}

// ---- stages k >= 1: linear even / odd arrays.  E[hs + i] / O[ho + i] hold samples 2i / 2i+1 of the tile, the hs / ho
// entries in front the history; G outputs per lane (2 in stage 1, 1 behind it).
__host__ __device__ constexpr int casc_lin_hs(int m) { return 2 * m; }                 // >= 2m - 1, even
__host__ __device__ constexpr int casc_lin_ho(int m) { return (m + 1) & ~1; }          // >= m, even
__host__ __device__ constexpr int casc_lin_g(int k) { return k == 1 ? 2 : 1; }         // outputs per lane
__host__ __device__ constexpr int casc_lin_lanes(int k) { return (256 >> k) / casc_lin_g(k); }


template <int M, int G> struct CascWinLin { v2f W[G == 2 ? 2 * M + 2 : 2 * M]; v2f o[2]; };
template <int M, int G>
__device__ __forceinline__ void casc_stage_lin_load(const char *E, const char *O, int lane, CascWinLin<M, G> &wn)
{
    constexpr int HS = casc_lin_hs(M), HO = casc_lin_ho(M);
    if (G == 2) {
        // outputs 2l, 2l+1: E[HS + 2l - (2M-1) .. HS + 2l + 1] = array entries 2l + 1 .. 2l + 2M + 1 -> 16-byte reads from entry 2l
        const char *we = E + lane * 16;
#pragma unroll
        for (int r = 0; r <= M; ++r) {
            const float4 v = ld4a(we + r * 16);
            wn.W[2 * r] = v2f{v.x, v.y}; wn.W[2 * r + 1] = v2f{v.z, v.w};
        }
        wn.o[0] = ld2(O + (HO + 2 * lane - M) * 8); wn.o[1] = ld2(O + (HO + 2 * lane + 1 - M) * 8);
    } else {
        // output l: E[l - (2M-1) .. l] = array entries l + 1 .. l + 2M (8-byte reads, consecutive lanes consecutive words)
        const char *we = E + (lane + HS - (2 * M - 1)) * 8;
#pragma unroll
        for (int i = 0; i < 2 * M; ++i) wn.W[i] = ld2(we + i * 8);
        wn.o[0] = ld2(O + (HO + lane - M) * 8);
    }
}
template <int M, int G>
__device__ __forceinline__ void casc_stage_lin_fma(const CascWinLin<M, G> &wn, const float *taps_sgpr, v2f y[2])
{
    constexpr int HS = casc_lin_hs(M);
    const v2f *hbp = (const v2f *)taps_sgpr;          // M SGPR pairs {h[2i], h[2i+1]}
    if (G == 2) {
#pragma unroll
        for (int i = 0; i < 2 * M + 2; ++i) keep(wn.W[i]);
        // W[i] = array entry 2l + i = E index 2l + i - HS; output j = 2l + g uses E[j - q] = W[HS + g - q]
        y[0] = v2f{0.5f * wn.o[0].x, 0.5f * wn.o[0].y}; y[1] = v2f{0.5f * wn.o[1].x, 0.5f * wn.o[1].y};
#pragma unroll
        for (int q2 = 0; q2 < M; ++q2) {
            const v2f tp = hbp[q2];
            pk_fma_lo_s(y[0], tp, wn.W[HS - 2 * q2]);     pk_fma_lo_s(y[1], tp, wn.W[HS + 1 - 2 * q2]);
            pk_fma_hi_s(y[0], tp, wn.W[HS - 2 * q2 - 1]); pk_fma_hi_s(y[1], tp, wn.W[HS - 2 * q2]);
        }
    } else {
        y[0] = v2f{0.5f * wn.o[0].x, 0.5f * wn.o[0].y};
        // W[i] = E[l - (2M-1) + i]; tap q multiplies E[l - q] = W[2M - 1 - q]
#pragma unroll
        for (int q2 = 0; q2 < M; ++q2) {
            const v2f tp = hbp[q2];
            pk_fma_lo_s(y[0], tp, wn.W[2 * M - 1 - 2 * q2]);
            pk_fma_hi_s(y[0], tp, wn.W[2 * M - 2 - 2 * q2]);
        }
    }
}
template <int M, int G>
__device__ __forceinline__ void casc_stage_lin(const char *E, const char *O, int lane, const float *taps_sgpr, v2f y[2])
{
    CascWinLin<M, G> wn;
    casc_stage_lin_load<M, G>(E, O, lane, wn);
    casc_stage_lin_fma<M, G>(wn, taps_sgpr, y);
}

// ---- stage 0 on RAW frames (RAW0: 8-bit input, unit gain, no dc blocker / iq correction / mixer in front -- BASELINE
// configs[3]).  The streaming waves keep the tile's frames in LDS as they came (2 bytes a frame: 1 KiB per tile instead of
// 4 KiB of cf32, one 16-byte read per 8 frames instead of four) and a lane unpacks the 4M + 5 frames its four outputs need
// in registers: the unpacked values are the ones unpack_chunk produces (cu8: (u - 127.5) / 128 as ONE fused multiply-add --
// product and sum are exact, so the rounding of the two-step form never happens), the taps meet them in the order of
// casc_stage, and the bits are the same.  Layout: byte 64 + 2 f holds frame f of the tile, bytes 0 .. 63 the last 32 frames
// of the tile before.  Edge waves keep the cf32 rows (their history comes as processed samples).
constexpr int kRawHist = 64;
__host__ __device__ constexpr int casc_raw_nb(int m) { return (2 * m - 1 + 3) / 4 + 1; }   // 16-byte blocks: the lane's own and the ones before it
template <int M> struct CascWinRaw { uint32_t W[4 * casc_raw_nb(M)]; };
template <int M>
__device__ __forceinline__ void casc_stage_raw8_load(const char *RB, int lane, CascWinRaw<M> &wn)
{
    constexpr int NB = casc_raw_nb(M);
    const char *wb = RB + kRawHist + (lane - (NB - 1)) * 16;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const uint4 v = *(const uint4 *)__builtin_assume_aligned(wb + b * 16, 16);
        wn.W[4 * b + 0] = v.x; wn.W[4 * b + 1] = v.y; wn.W[4 * b + 2] = v.z; wn.W[4 * b + 3] = v.w;
    }
}
template <int M, bool UNS>
__device__ __forceinline__ void casc_stage_raw8_fma(const CascWinRaw<M> &wn, const float *taps_sgpr, v2f y[4])
{
    // even sample n / odd sample n of the lane (frames 8 lane + 2n, + 2n + 1), n = -(2M-1) .. 3: dword n of the lane's blocks
    constexpr int NB = casc_raw_nb(M);
    auto unpack = [&](uint32_t h) {                         // h: one frame in the low 16 bits
        v2f x;
        if (UNS) {
            x.x = __builtin_fmaf((float)(h & 0xffu), 1.0f / 128.0f, -127.5f / 128.0f);
            x.y = __builtin_fmaf((float)((h >> 8) & 0xffu), 1.0f / 128.0f, -127.5f / 128.0f);
        } else {
            x.x = (float)(signed char)(h & 0xffu) * (1.0f / 128.0f);
            x.y = (float)(signed char)((h >> 8) & 0xffu) * (1.0f / 128.0f);
        }
        return x;
    };
    constexpr int Z = 4 * (NB - 1);                         // dword index of n = 0
#pragma unroll
    for (int i = 0; i < 4 * NB; ++i) keep(wn.W[i]);
    v2f E[2 * M + 3];                                       // E[k] = even sample n = k - (2M - 1)
#pragma unroll
    for (int k = 0; k < 2 * M + 3; ++k) E[k] = unpack(wn.W[Z + k - (2 * M - 1)] & 0xffffu);
    const v2f *hbp = (const v2f *)taps_sgpr;                // M SGPR pairs {h[2i], h[2i+1]}
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const v2f o = unpack(wn.W[Z + i - M] >> 16);        // O[j - M], j = 4 lane + i
        y[i] = v2f{0.5f * o.x, 0.5f * o.y};
    }
#pragma unroll
    for (int q2 = 0; q2 < M; ++q2) {
        const v2f tp = hbp[q2];
#pragma unroll
        for (int i = 0; i < 4; ++i) pk_fma_lo_s(y[i], tp, E[2 * M - 1 + i - 2 * q2]);
#pragma unroll
        for (int i = 0; i < 4; ++i) pk_fma_hi_s(y[i], tp, E[2 * M - 1 + i - 2 * q2 - 1]);
    }
}
template <int M, bool UNS>
__device__ __forceinline__ void casc_stage_raw8(const char *RB, int lane, const float *taps_sgpr, v2f y[4])
{
    CascWinRaw<M> wn;
    casc_stage_raw8_load<M>(RB, lane, wn);
    casc_stage_raw8_fma<M, UNS>(wn, taps_sgpr, y);
}

struct CascLds { char *XE[kCascMaxK], *XO[kCascMaxK]; const cf2 *nco; };

// KT: 0 = stage count and semi-lengths from the arguments; 1 .. 4 = that many stages with liquid's 60 dB semi-lengths
// (3 everywhere, 5 in the last one) resolved at compile time, so that the whole tile is straight-line code and the
// window reads of all stages are issued together
template <int BPS, bool EDGE, bool RAW0 = false, int KT = 0>
__device__ __forceinline__ void casc_tiles(const FrontArgs &a, const CascLds &w, const int lane,
                                           const int64_t t_begin, const int64_t t_emit0, const int64_t t_emit1, const int seg)
{
    constexpr int VB = BPS ? BPS : 4;
    const int K = KT ? KT : a.casc_K;
    auto stage_m = [&](int k) { return KT ? (k == KT - 1 ? 5 : 3) : a.m[k]; };
    // dc blocker (SPEC B.5): v[n] = x[n] + c v[n-1], y[n] = x[n] - (1 - c) v[n-1]; v carried as a wave-uniform pair
    float dc_vr = 0.0f, dc_vi = 0.0f;
    DcLane lane_pow{1.0f, 1.0f, 1.0f};                     // c^(4 lane) and the scan's cross-row weights
    bool dc_started = false;
    if (a.dc_enable) {
        lane_pow = dc_lane_init(a, lane);
    }
    const bool unit_gain = a.gain == 1.0f;
    char *XE0 = w.XE[0], *XO0 = w.XO[0];
    const int H0 = casc_hist_rows(stage_m(0));
    const int PS0 = plane_stride(H0 + 64 + 1);
    const int woff = (H0 + (lane >> 1)) * 16 + (lane & 1) * PS0;       // this lane's write slot in stage 0 (plane lane & 1)

    RawChunk nxt[2];
    const bool nco_on = !EDGE && a.nco_mode != 0;
    v2f cs_n[2][4];
    auto nco_lookup = [&](int64_t tile_first) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            uint32_t th = a.nco_theta0 + ((uint32_t)tile_first + (uint32_t)(256 * c + 4 * lane)) * a.nco_dtheta;
#pragma unroll
            for (int s = 0; s < 4; ++s) { cs_n[c][s] = nco_phasor2(w.nco, th); th += a.nco_dtheta; }
        }
    };
    if (!EDGE) {
        const char *src = (const char *)a.raw + (t_begin * kWTile - a.rem0) * VB + 4 * VB * lane;
        load_chunk<VB, IQGPU_NT_CASC != 0>(src, nxt[0]);
        load_chunk<VB, IQGPU_NT_CASC != 0>(src + 256 * VB, nxt[1]);
        if (nco_on) nco_lookup(t_begin * kWTile);
    }

    // The stages run SKEWED by one tile each: what iteration t hands to LDS (the pointwise samples of tile t, stage k's
    // outputs) is read at the top of iteration t + 1, so that an iteration is ONE LDS round trip -- every stage's window
    // reads are issued together, then the FMAs, then all writes -- instead of a chain of K + 1.  Stage k works on tile
    // t - 1 - k, the last stage's outputs of tile t - K go to memory; K more iterations drain the pipe (their pointwise
    // input is never used by an emitted output: every stage is causal).
    for (int64_t t = t_begin; t < t_emit1 + K; ++t) {
        const int64_t i0 = t * kWTile;
        const int64_t j0 = i0 - a.rem0;
        const bool fresh = t < t_emit1;                   // tile t is part of this run (else: drain)
        const bool emit = t >= t_emit0 && fresh;          // pointwise side effects (hist_out)

        // ------------------------------------------------------------ pointwise -> stage 0 rows
        cf2 x[2][4];
        uint32_t rw[4] = {0u, 0u, 0u, 0u};
        if (RAW0 && !EDGE) {
            // the frames go to LDS as they are; nothing stands between the unpack and stage 0 (launch_cascade checks)
            rw[0] = nxt[0].w[0]; rw[1] = nxt[0].w[1]; rw[2] = nxt[1].w[0]; rw[3] = nxt[1].w[1];
            if (fresh) {
                const char *src = (const char *)a.raw + (j0 + kWTile) * VB + 4 * VB * lane;
                load_chunk<VB, IQGPU_NT_CASC != 0>(src, nxt[0]);
                load_chunk<VB, IQGPU_NT_CASC != 0>(src + 256 * VB, nxt[1]);
            }
        } else if (!EDGE) {
            unpack_chunk<VB>(nxt[0], a.in_fmt, a.gain, unit_gain, x[0]);
            unpack_chunk<VB>(nxt[1], a.in_fmt, a.gain, unit_gain, x[1]);
            if (fresh) {
                const char *src = (const char *)a.raw + (j0 + kWTile) * VB + 4 * VB * lane;
                load_chunk<VB, IQGPU_NT_CASC != 0>(src, nxt[0]);
                load_chunk<VB, IQGPU_NT_CASC != 0>(src + 256 * VB, nxt[1]);
            }
            if (a.dc_enable) {
                if (!dc_started) { const cd2 cv = a.dc_carry[seg]; dc_vr = (float)cv.x; dc_vi = (float)cv.y; dc_started = true; }
                dc_chunk(a, lane, lane_pow, x[0], 0u, dc_vr, dc_vi);
                dc_chunk(a, lane, lane_pow, x[1], 0u, dc_vr, dc_vi);
            }
            if (a.iq_enable) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float re = x[c][s].x;
                        x[c][s].x = re * a.iq_magp1;
                        x[c][s].y = fmaf(a.iq_phase, re, x[c][s].y);
                    }
            }
            if (nco_on) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const v2f y = pk_cmul(v2f{x[c][s].x, x[c][s].y}, cs_n[c][s]);
                        x[c][s] = cf2{y.x, y.y};
                    }
            }
        } else {
            // edge tiles: per-frame loads; history frames (js < 0) are already fully processed and skip
            // every operator, frames past the end of the call are zeros
            unsigned hist_mask[2] = {0u, 0u}, new_mask[2] = {0u, 0u};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int64_t j = j0 + 256 * c + 4 * lane;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int64_t js = j + s;
                    cf2 v{0.0f, 0.0f};
                    if (js < 0) {
                        const int64_t h = (int64_t)a.hist_cap + js;
                        if (h >= 0) v = a.hist_in[h];
                        hist_mask[c] |= 1u << s;
                    } else if (js < a.frames_in) {
                        v = unpack_one(a.raw, js, a.in_fmt, a.gain);
                        new_mask[c] |= 1u << s;
                    }
                    x[c][s] = v;
                }
            }
            if (a.dc_enable && (dc_started || j0 + kWTile > 0)) {
                if (!dc_started) {
                    // state before the run's first new sample, moved back over the history positions of this
                    // tile that precede it (they feed zeros into the recurrence)
                    const cd2 cv = a.dc_carry[seg];
                    const int64_t n_h = (j0 < 0) ? -j0 : 0;
                    const double back = exp(-(double)n_h * a.dc_logc);
                    dc_vr = (float)(cv.x * back); dc_vi = (float)(cv.y * back);
                    dc_started = true;
                }
                dc_chunk(a, lane, lane_pow, x[0], hist_mask[0], dc_vr, dc_vi);
                dc_chunk(a, lane, lane_pow, x[1], hist_mask[1], dc_vr, dc_vi);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int l4 = 256 * c + 4 * lane;
                uint32_t th = a.nco_theta0 + ((uint32_t)i0 + (uint32_t)l4) * a.nco_dtheta;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (new_mask[c] & (1u << s)) {
                        cf2 v = x[c][s];
                        if (a.iq_enable) {
                            const float re = v.x;
                            v.x = re * a.iq_magp1;
                            v.y = fmaf(a.iq_phase, re, v.y);
                        }
                        if (a.nco_mode != 0) v = cmul_tab(v, nco_phasor(w.nco, th));
                        const int64_t back = a.frames_in - (j0 + l4 + s);   // 1 .. hist_cap for kept frames
                        if (emit && back <= (int64_t)a.hist_cap) a.hist_out[(int64_t)a.hist_cap - back] = v;
                        x[c][s] = v;
                    }
                    th += a.nco_dtheta;
                }
            }
        }
        // ------------------------------------------------------------ the stages: reads and FMAs (tile t - 1 - k in stage k)
        if (!EDGE) __builtin_amdgcn_s_setprio(1);       // feeding the LDS pipe goes ahead of FMA runs (as in k_front_s1)
        if (nco_on && fresh) nco_lookup(i0 + kWTile);
        v2f ys[kCascMaxK][4];
        float se[kCascMaxK], so[kCascMaxK];
        uint32_t hv = 0;
        if constexpr (KT > 0) {
            // compile-time stage list: every window (and every history tail) is read first, then all the FMAs run
            constexpr int M0 = KT == 1 ? 5 : 3, M1 = KT == 2 ? 5 : 3, M2 = KT == 3 ? 5 : 3, M3 = 5;
            CascWinRaw<M0> r0; CascWin0<M0> f0; CascWinLin<M1, 2> l1; CascWinLin<M2, 1> l2; CascWinLin<M3, 1> l3;
#pragma unroll
            for (int k = 0; k < kCascMaxK; ++k) {
                se[k] = 0.f; so[k] = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) ys[k][i] = v2f{0.f, 0.f};
            }
            if (RAW0 && !EDGE) {
                casc_stage_raw8_load<M0>(w.XE[0], lane, r0);
                hv = *(const uint32_t *)(w.XE[0] + 1024 + (lane & (kRawHist / 4 - 1)) * 4);
            } else {
                casc_stage_load<M0>(w.XE[0], w.XO[0], lane, f0);
                constexpr int Hk = casc_hist_rows(M0);
                const int ls = lane < 8 * Hk ? lane : 8 * Hk - 1;
                const int so_ = (ls >= 4 * Hk ? PS0 - 16 * Hk : 0) + ls * 4;
                se[0] = *(const float *)(w.XE[0] + 64 * 16 + so_); so[0] = *(const float *)(w.XO[0] + 64 * 16 + so_);
            }
            auto tails = [&](int k, int m) {
                const int hs = casc_lin_hs(m), ho = casc_lin_ho(m), pk_ = 256 >> k;
                se[k] = *(const float *)(w.XE[k] + pk_ * 8 + (lane < 2 * hs ? lane : 2 * hs - 1) * 4);
                so[k] = *(const float *)(w.XO[k] + pk_ * 8 + (lane < 2 * ho ? lane : 2 * ho - 1) * 4);
            };
            // stage groups whose windows are read together: {0, 1} {2, 3} on raw frames (all four at once do not fit 128
            // registers); with the cf32 rows (48 registers for stage 0's window alone) every stage reads for itself
            constexpr bool G01 = RAW0 && !EDGE;
            if (G01 && KT > 1) { casc_stage_lin_load<M1, 2>(w.XE[1], w.XO[1], lane, l1); tails(1, M1); }
            __builtin_amdgcn_sched_barrier(0);
            if (!EDGE) __builtin_amdgcn_s_setprio(0);
            if (RAW0 && !EDGE) casc_stage_raw8_fma<M0, true>(r0, a.casc_taps[0], ys[0]);
            else casc_stage_fma<M0>(f0, a.casc_taps[0], ys[0]);
            if (!G01 && KT > 1) { casc_stage_lin_load<M1, 2>(w.XE[1], w.XO[1], lane, l1); tails(1, M1); }
            if (KT > 1) casc_stage_lin_fma<M1, 2>(l1, a.casc_taps[1], ys[1]);
            if (KT > 2) { casc_stage_lin_load<M2, 1>(w.XE[2], w.XO[2], lane, l2); tails(2, M2); }
            if (G01 && KT > 3) { const int lc = lane < 32 ? lane : 31; casc_stage_lin_load<M3, 1>(w.XE[3], w.XO[3], lc, l3); tails(3, M3); }
            if (KT > 2) casc_stage_lin_fma<M2, 1>(l2, a.casc_taps[2], ys[2]);
            if (!G01 && KT > 3) { const int lc = lane < 32 ? lane : 31; casc_stage_lin_load<M3, 1>(w.XE[3], w.XO[3], lc, l3); tails(3, M3); }
            if (KT > 3) casc_stage_lin_fma<M3, 1>(l3, a.casc_taps[3], ys[3]);
        } else {
#pragma unroll
        for (int k = 0; k < kCascMaxK; ++k) {
            se[k] = 0.f; so[k] = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) ys[k][i] = v2f{0.f, 0.f};
            if (k < K) {
                const int m = stage_m(k);
                const int n_act = k == 0 ? 64 : casc_lin_lanes(k);
                const int lc = lane < n_act ? lane : n_act - 1;       // idle lanes read what the last active one reads
                if (k == 0 && RAW0 && !EDGE) {
                    if (BPS == 2) {          // RAW0 stands for "cu8" here, RAW0 with BPS 0 for "cs8" (launch_cascade)
                        if (m == 3) casc_stage_raw8<3, true>(w.XE[0], lane, a.casc_taps[0], ys[0]);
                        else        casc_stage_raw8<5, true>(w.XE[0], lane, a.casc_taps[0], ys[0]);
                    }
                    hv = *(const uint32_t *)(w.XE[0] + 1024 + (lane & (kRawHist / 4 - 1)) * 4);   // the last 32 frames: next tile's history (unconditional reads: straight-line code)
                } else if (k == 0) {
                    if (m == 3) casc_stage<3>(w.XE[0], w.XO[0], lane, a.casc_taps[0], ys[0]);
                    else        casc_stage<5>(w.XE[0], w.XO[0], lane, a.casc_taps[0], ys[0]);
                    const int Hk = casc_hist_rows(m);                  // Hk rows = 16 Hk bytes in each of the two planes
                    const int ls = lane < 8 * Hk ? lane : 8 * Hk - 1;
                    const int so_ = (ls >= 4 * Hk ? PS0 - 16 * Hk : 0) + ls * 4;
                    se[0] = *(const float *)(w.XE[0] + 64 * 16 + so_); so[0] = *(const float *)(w.XO[0] + 64 * 16 + so_);
                } else {
                    if (k == 1) { if (m == 3) casc_stage_lin<3, 2>(w.XE[k], w.XO[k], lc, a.casc_taps[k], ys[k]); else casc_stage_lin<5, 2>(w.XE[k], w.XO[k], lc, a.casc_taps[k], ys[k]); }
                    else        { if (m == 3) casc_stage_lin<3, 1>(w.XE[k], w.XO[k], lc, a.casc_taps[k], ys[k]); else casc_stage_lin<5, 1>(w.XE[k], w.XO[k], lc, a.casc_taps[k], ys[k]); }
                    const int hs = casc_lin_hs(m), ho = casc_lin_ho(m), pk_ = 256 >> k;     // samples per parity and tile
                    se[k] = *(const float *)(w.XE[k] + pk_ * 8 + (lane < 2 * hs ? lane : 2 * hs - 1) * 4);
                    so[k] = *(const float *)(w.XO[k] + pk_ * 8 + (lane < 2 * ho ? lane : 2 * ho - 1) * 4);
                }
            }
        }
        }
        __builtin_amdgcn_wave_barrier();

        // ------------------------------------------------------------ the writes: histories slide, every stage hands its tile on
        if (!EDGE) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int k = 0; k < kCascMaxK; ++k) {
            if (k < K) {
                const int m = stage_m(k);
                const int g_out = k == 0 ? 4 : casc_lin_g(k);     // outputs per lane of this stage
                const int n_act = k == 0 ? 64 : casc_lin_lanes(k);
                const v2f *y = ys[k];
                // slide this stage's history to the front of its buffers (one dword per lane), then its new input behind it
                if (k == 0 && RAW0 && !EDGE) {
                    if (lane < kRawHist / 4) *(uint32_t *)(w.XE[0] + lane * 4) = hv;
                    *(uint2 *)(XE0 + kRawHist + 8 * lane) = make_uint2(rw[0], rw[1]);
                    *(uint2 *)(XE0 + kRawHist + 512 + 8 * lane) = make_uint2(rw[2], rw[3]);
                } else if (k == 0) {
                    const int Hk = casc_hist_rows(m);
                    const int so_ = (lane >= 4 * Hk ? PS0 - 16 * Hk : 0) + lane * 4;
                    if (lane < 8 * Hk) { *(float *)(w.XE[0] + so_) = se[0]; *(float *)(w.XO[0] + so_) = so[0]; }
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int off = woff + 32 * c * 16;
                        st4a(XE0 + off, make_float4(x[c][0].x, x[c][0].y, x[c][2].x, x[c][2].y));
                        st4a(XO0 + off, make_float4(x[c][1].x, x[c][1].y, x[c][3].x, x[c][3].y));
                    }
                } else {
                    const int hs = casc_lin_hs(m), ho = casc_lin_ho(m);
                    if (lane < 2 * hs) *(float *)(w.XE[k] + lane * 4) = se[k];
                    if (lane < 2 * ho) *(float *)(w.XO[k] + lane * 4) = so[k];
                }
                if (k + 1 < K) {
                    // this stage's outputs -> the even / odd arrays of the next one (output j: even -> E[hs + j/2], odd -> O[ho + j/2])
                    const int hn = casc_lin_hs(stage_m(k + 1)), on = casc_lin_ho(stage_m(k + 1));
                    if (lane < n_act) {
                        if (g_out == 4) {
                            st4a(w.XE[k + 1] + (hn + 2 * lane) * 8, make_float4(y[0].x, y[0].y, y[2].x, y[2].y));
                            st4a(w.XO[k + 1] + (on + 2 * lane) * 8, make_float4(y[1].x, y[1].y, y[3].x, y[3].y));
                        } else if (g_out == 2) {
                            *(float2 *)(w.XE[k + 1] + (hn + lane) * 8) = make_float2(y[0].x, y[0].y);
                            *(float2 *)(w.XO[k + 1] + (on + lane) * 8) = make_float2(y[1].x, y[1].y);
                        } else {
                            char *dst = (lane & 1) ? w.XO[k + 1] + (on + (lane >> 1)) * 8 : w.XE[k + 1] + (hn + (lane >> 1)) * 8;
                            *(float2 *)dst = make_float2(y[0].x, y[0].y);
                        }
                    }
                } else if (t - K >= t_emit0 && lane < n_act) {
                    // the last stage's outputs (tile t - K) go to memory: g_out contiguous cf32 per lane
                    const int64_t o = ((i0 - (int64_t)K * kWTile) >> K) + (int64_t)g_out * lane;
                    if (g_out == 4) {
                        if (!EDGE || o + 4 <= a.casc_n_out) {
                            float4 *dst = (float4 *)(a.casc_out + o);
                            dst[0] = make_float4(y[0].x, y[0].y, y[1].x, y[1].y);
                            dst[1] = make_float4(y[2].x, y[2].y, y[3].x, y[3].y);
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i) if (o + i < a.casc_n_out) a.casc_out[o + i] = cf2{y[i].x, y[i].y};
                        }
                    } else if (g_out == 2) {
                        if (!EDGE || o + 2 <= a.casc_n_out) *(float4 *)(a.casc_out + o) = make_float4(y[0].x, y[0].y, y[1].x, y[1].y);
                        else if (o < a.casc_n_out) a.casc_out[o] = cf2{y[0].x, y[0].y};
                    } else {
                        if (!EDGE || o < a.casc_n_out) a.casc_out[o] = cf2{y[0].x, y[0].y};
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (!EDGE) __builtin_amdgcn_s_setprio(0);
    }
}

// bytes of one stage's two buffers
__host__ __device__ inline int casc_stage_bytes(int k, int m)
{
    if (k == 0) return 4 * plane_stride(casc_hist_rows(m) + 64 + 1);
    const int pk_ = 256 >> k;
    return (((casc_lin_hs(m) + pk_) * 8 + 15) & ~15) + (((casc_lin_ho(m) + pk_) * 8 + 15) & ~15);
}

} // namespace iqgpu
