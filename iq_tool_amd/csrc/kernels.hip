// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the I/Q chain.
//
//   k_front      raw -> unpack/gain -> dc block -> iq correct -> NCO -> half-band cascade ->
//                256-arm polyphase -> post NCO -> pack           (one launch per process() call)
//   k_dc_prefix  per-segment aggregate of the DC blocker's first-order recurrence
//   k_dc_scan    carries of the (few) segments
//   k_fir        time-domain FIR with LDS tap tiling over the cf32 filter-input buffer
//
// Reference semantics and file:line citations are in DESIGN.md; the per-operator arithmetic
// follows sample_convert.c (bit-exact: this file is compiled with -ffp-contract=off, so a*b+c is
// two roundings unless written as fmaf(); HIP's __fmul_rn/__fadd_rn are plain operators),
// frequency_shift.c / dc_block.c / iq_correct.c / resampler.c / filter.c call sites.
#include <hip/hip_runtime.h>

#include "../../include/iqgpu.h"
#include "dsp_device.hpp"
#include "kernels.hpp"

#ifndef IQGPU_NT_DC
#define IQGPU_NT_DC 1      // k_dc_prefix reads the call's frames once, with the non-temporal hint (0.116 -> 0.105 ms on config 3)
#endif

namespace iqgpu {

__device__ __forceinline__ int lvl_hist(const FrontArgs &a, int i) { return (i < a.S) ? 4 * a.m[i] : kArbHist; }

// one half-band decimator stage over a level buffer: dst[j] = 0.5 x[2j+1-2m] + sum_q h[2q+1] x[2j-2q]
// (taps pre-scaled by 0.5).  M > 0: compile-time semi-length, two accumulator chains.
template <int M>
__device__ __forceinline__ void hb_stage(const cf2 *src, cf2 *dst, const float *taps, int n_out, int tid, int m_rt = 0)
{
    const int m = M > 0 ? M : m_rt;
    float h[M > 0 ? 2 * M : 1];
    if (M > 0) {
#pragma unroll
        for (int q = 0; q < 2 * M; ++q) h[q] = taps[q];
    }
    for (int jl = tid; jl < n_out; jl += kThreads) {
        const cf2 d = src[2 * jl + 1 - 2 * m];           // centre tap (delay branch), gain 0.5
        const cf2 *p = src + 2 * jl;
        float ar = 0.5f * d.x, ai = 0.5f * d.y;
        if (M > 0) {
            cf2 sv[M > 0 ? 2 * M : 1];
#pragma unroll
            for (int q = 0; q < 2 * M; ++q) sv[q] = p[-2 * q];
            float br = 0.0f, bi = 0.0f;
#pragma unroll
            for (int q = 0; q < 2 * M; q += 2) {             // same order within each chain as the rolled loop
                ar = fmaf(h[q], sv[q].x, ar); ai = fmaf(h[q], sv[q].y, ai);
                br = fmaf(h[q + 1], sv[q + 1].x, br); bi = fmaf(h[q + 1], sv[q + 1].y, bi);
            }
            ar += br; ai += bi;
        } else {
            for (int q = 0; q < 2 * m; ++q) {                // odd taps h[2q+1] on x[2jl - 2q]
                const cf2 sv = p[-2 * q];
                const float h = taps[q];
                ar = fmaf(h, sv.x, ar); ai = fmaf(h, sv.y, ai);
            }
        }
        dst[jl] = cf2{ar, ai};
    }
}

// ============================================================================================
// k_front
//   Block b owns tiles [b*tpb, (b+1)*tpb) of kTile input samples (tile boundaries are aligned to
//   the stream's 2^S groups) and first re-runs warm_tiles tiles to rebuild every stage window, so
//   blocks are independent; inside a block the tiles run in order and all state (dc blocker
//   value, stage histories, next output index) is carried in registers / LDS.
// ============================================================================================
// one frame of a 4-byte / 2-byte output format as the word pack_store would write (dsp_device.hpp: the same expressions, case by case)
__device__ __forceinline__ uint32_t pack_word32(int fmt, cf2 v)
{
    if (fmt == IQGPU_FMT_CU16) return pk_unsigned(v.x, 32767.0f, 32767.5f, 65535.0f) | (pk_unsigned(v.y, 32767.0f, 32767.5f, 65535.0f) << 16);
    const float s = (fmt == IQGPU_FMT_CS16) ? 32767.0f : 2048.0f;
    return ((unsigned)pk_signed(v.x, s, -32768.0f, 32767.0f) & 0xffffu) | (((unsigned)pk_signed(v.y, s, -32768.0f, 32767.0f) & 0xffffu) << 16);
}
__device__ __forceinline__ uint32_t pack_word16(int fmt, cf2 v)
{
    if (fmt == IQGPU_FMT_CU8) return pk_unsigned(v.x, 127.0f, 127.5f, 255.0f) | (pk_unsigned(v.y, 127.0f, 127.5f, 255.0f) << 8);
    return ((unsigned)pk_signed(v.x, 127.0f, -128.0f, 127.0f) & 0xffu) | (((unsigned)pk_signed(v.y, 127.0f, -128.0f, 127.0f) & 0xffu) << 8);
}

__global__ __launch_bounds__(kThreads) void k_front(const FrontArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = a.S;

    cf2   *s_nco = (cf2 *)smem;                              // 1024 {cos, sin}
    float *s_arb = (float *)(s_nco + 1024);                  // [256][16]
    float *s_hb  = s_arb + 256 * 16;                         // branch taps
    cf2   *s_wtot = (cf2 *)(s_hb + ((a.n_hb_taps + 3) & ~3)); // [2][4] dc wave totals
    cf2   *s_lvl = s_wtot + 8;                               // level buffers

    // ---- tables and zeroed histories ----
    if (a.nco_mode != 0 || a.pnco_mode != 0)
        for (int i = tid; i < 1024; i += kThreads) s_nco[i] = a.nco_tab[i];
    if (a.mode == 1) {
        for (int i = tid; i < 256 * 16; i += kThreads) s_arb[i] = a.arb_table[i];
        for (int i = tid; i < a.n_hb_taps; i += kThreads) s_hb[i] = a.hb_taps[i];
        const int total = a.lvl_off[S + 1];
        for (int i = tid; i < total; i += kThreads) s_lvl[i] = cf2{0.0f, 0.0f};
    }

    const int b = blockIdx.x;
    const int64_t t_emit0 = (int64_t)b * a.tiles_per_block;
    int64_t t_emit1 = t_emit0 + a.tiles_per_block;
    if (t_emit1 > a.total_tiles || b == (int)gridDim.x - 1) t_emit1 = a.total_tiles;
    const int64_t t_begin = t_emit0 - a.warm_tiles;
    const int TG = kTile >> S;
    const int bps = (a.in_fmt == IQGPU_FMT_CS8 || a.in_fmt == IQGPU_FMT_CU8) ? 2
                  : (a.in_fmt == IQGPU_FMT_CS24) ? 6
                  : (a.in_fmt == IQGPU_FMT_CS32 || a.in_fmt == IQGPU_FMT_CU32 || a.in_fmt == IQGPU_FMT_CF32) ? 8 : 4;

    // ---- block 0 keeps the part of the old history that this (short) call does not replace ----
    if (b == 0 && a.frames_in < (int64_t)a.hist_cap) {
        const int keep = a.hist_cap - (int)a.frames_in;
        for (int i = tid; i < keep; i += kThreads) a.hist_out[i] = a.hist_in[i + (int)a.frames_in];
    }

    // ---- dc blocker state ----
    float vr = 0.0f, vi = 0.0f;       // v[n-1] at the start of the next chunk (uniform)
    bool dc_started = false;
    float lane_pow = 1.0f;            // c^(4*lane)
    if (a.dc_enable) {
#pragma unroll
        for (int k = 0; k < 6; ++k) if (lane & (1 << k)) lane_pow *= a.dc_cpow[k];
    }

    uint64_t k_next = 0;
    if (a.mode == 1) k_next = first_k_at((uint64_t)(t_emit0 * TG) << 24, a.phi0, a.step);

    __syncthreads();

    for (int64_t t = t_begin; t < t_emit1; ++t) {
        const int64_t i0 = t * kTile;             // group-aligned index of the tile's first sample
        const int64_t j0 = i0 - a.rem0;           // index into this call's new samples
        const bool emit = t >= t_emit0;
        const bool interior = (j0 >= 0) && (j0 + kTile <= a.frames_in) && a.raw_aligned &&
                              (((j0 * bps) & 15) == 0);
        cf2 *lv0 = s_lvl + a.lvl_off[0] + lvl_hist(a, 0);

        if (a.dc_enable && !dc_started && j0 + kTile > 0) {
            // state before the block's first new sample, moved back over the n_h history
            // positions of this tile that precede it (they feed zeros into the recurrence)
            const cd2 cv = a.dc_carry[b];
            const int64_t n_h = (j0 < 0) ? -j0 : 0;
            const double back = exp(-(double)n_h * a.dc_logc);
            vr = (float)(cv.x * back); vi = (float)(cv.y * back);
            dc_started = true;
        }

        // ------------------------------------------------------------ phase 1: pointwise
#pragma unroll
        for (int c = 0; c < kTile / 1024; ++c) {
            const int l = c * 1024 + 4 * tid;
            const int64_t j = j0 + l;
            cf2 x[4];
            bool is_hist[4] = {false, false, false, false};
            bool is_new[4] = {true, true, true, true};
            bool fast = false;
            if (interior) fast = unpack_four_fast(a.raw, j, a.in_fmt, a.gain, x);
            if (!fast) {
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

            if (a.dc_enable) {
                // v[n] = x[n] + c v[n-1];  y[n] = v[n] - v[n-1] = x[n] - (1-c) v[n-1]   (SPEC B.5)
                const float cc = a.dc_c;
                cf2 xd[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) xd[s] = is_hist[s] ? cf2{0.0f, 0.0f} : x[s];
                float br = xd[0].x, bi = xd[0].y;
#pragma unroll
                for (int s = 1; s < 4; ++s) { br = fmaf(br, cc, xd[s].x); bi = fmaf(bi, cc, xd[s].y); }
                // inclusive scan over the 64 lanes: B_l += c^(4*2^k) B_(l-2^k)
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const float ur = __shfl_up(br, 1 << k), ui = __shfl_up(bi, 1 << k);
                    if (lane >= (1 << k)) { br = fmaf(a.dc_cpow[k], ur, br); bi = fmaf(a.dc_cpow[k], ui, bi); }
                }
                cf2 *wt = s_wtot + 4 * (c & 1);
                if (lane == 63) wt[wave] = cf2{br, bi};
                float er = __shfl_up(br, 1), ei = __shfl_up(bi, 1);   // exclusive within the wave
                if (lane == 0) { er = 0.0f; ei = 0.0f; }
                __syncthreads();
                // state at this wave's first sample, and at the end of the chunk
                const float c256 = a.dc_cpow[6];
                float pr = vr, pi = vi;           // running state across the 4 waves
                float wr = 0.0f, wi = 0.0f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    if (w == wave) { wr = pr; wi = pi; }
                    const cf2 tot = wt[w];
                    pr = fmaf(c256, pr, tot.x); pi = fmaf(c256, pi, tot.y);
                }
                vr = pr; vi = pi;
                float sr = fmaf(lane_pow, wr, er), si = fmaf(lane_pow, wi, ei);  // v[n-1] of x[0]
                const float aa = a.dc_a;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float yr = fmaf(-aa, sr, xd[s].x), yi = fmaf(-aa, si, xd[s].y);
                    sr = fmaf(cc, sr, xd[s].x); si = fmaf(cc, si, xd[s].y);
                    if (!is_hist[s]) { x[s].x = yr; x[s].y = yi; }
                }
            }

            if (a.iq_enable) { // src/iq_correct.c:307-313
#pragma unroll
                for (int s = 0; s < 4; ++s) if (!is_hist[s]) {
                    const float re = x[s].x;
                    x[s].x = re * a.iq_magp1;
                    x[s].y = fmaf(a.iq_phase, re, x[s].y);
                }
            }

            if (a.nco_mode != 0) {
                uint32_t th = a.nco_theta0 + (uint32_t)(i0 + l) * a.nco_dtheta;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (!is_hist[s]) x[s] = nco_mix(x[s], nco_phasor(s_nco, th), a.nco_mode);
                    th += a.nco_dtheta;
                }
            }

            if (emit && a.hist_cap > 0 && j + 4 > a.frames_in - (int64_t)a.hist_cap) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int64_t back = a.frames_in - (j + s);      // 1 .. hist_cap for kept samples
                    if (is_new[s] && back <= (int64_t)a.hist_cap) a.hist_out[(int64_t)a.hist_cap - back] = x[s];
                }
            }

            if (a.mode == 1) {
                float4 *dst = (float4 *)(lv0 + l);
                dst[0] = make_float4(x[0].x, x[0].y, x[1].x, x[1].y);
                dst[1] = make_float4(x[2].x, x[2].y, x[3].x, x[3].y);
            } else if (emit) {
                // no resampler: frames map one to one (src/pipeline.c:516-519)
                uint32_t th = a.pnco_theta0 + (uint32_t)j * a.pnco_dtheta;
                cf2 yv[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    cf2 y = x[s];
                    if (a.pnco_mode != 0) y = nco_mix(y, nco_phasor(s_nco, th), a.pnco_mode);
                    yv[s] = y;
                    th += a.pnco_dtheta;
                }
                // Round 6: a thread's four frames leave as ONE piece where the format allows it (cf32: two 16-byte stores, 4-byte frames:
                // one 16-byte store, 2-byte frames: one 8-byte store) instead of four stores of one frame each -- four instructions
                // that each touched 64 pieces 4 frames apart, every cache line written four times (the store shape that cost
                // k_front_p0's cf32 output 21 %).  Same arithmetic as pack_store (pack_word32 / pack_word16 restate its
                // cases); the stream's first and last frames keep the per-frame path.
                const bool all4 = is_new[0] && is_new[1] && is_new[2] && is_new[3];
                const int of = a.out_fmt;
                if (all4 && of == IQGPU_FMT_CF32) {
                    typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));
                    char *o = (char *)a.out + 8 * j;
                    *(f4a8 *)o = f4a8{yv[0].x, yv[0].y, yv[1].x, yv[1].y};
                    *(f4a8 *)(o + 16) = f4a8{yv[2].x, yv[2].y, yv[3].x, yv[3].y};
                } else if (all4 && (of == IQGPU_FMT_CS16 || of == IQGPU_FMT_SC16Q11 || of == IQGPU_FMT_CU16)) {
                    typedef uint32_t u4a4 __attribute__((ext_vector_type(4), aligned(4)));
                    *(u4a4 *)((char *)a.out + 4 * j) = u4a4{pack_word32(of, yv[0]), pack_word32(of, yv[1]), pack_word32(of, yv[2]), pack_word32(of, yv[3])};
                } else if (all4 && (of == IQGPU_FMT_CS8 || of == IQGPU_FMT_CU8)) {
                    typedef uint32_t u2a2 __attribute__((ext_vector_type(2), aligned(2)));
                    *(u2a2 *)((char *)a.out + 2 * j) = u2a2{pack_word16(of, yv[0]) | (pack_word16(of, yv[1]) << 16), pack_word16(of, yv[2]) | (pack_word16(of, yv[3]) << 16)};
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) if (is_new[s]) pack_store(a.out, j + s, of, yv[s]);
                }
            }
        }
        if (a.mode != 1) continue;
        __syncthreads();

        // ------------------------------------------------------------ phase 2: half-band cascade
        for (int i = 0; i < S; ++i) {
            const int m = a.m[i], n_out = kTile >> (i + 1);
            const cf2 *src = s_lvl + a.lvl_off[i] + 4 * m;
            cf2 *dst = s_lvl + a.lvl_off[i + 1] + lvl_hist(a, i + 1);
            const float *taps = s_hb + a.tap_off[i];
            // liquid's designs at 60 dB use m = 10, 5, 3, 3, ...: those get fully unrolled bodies (all window
            // reads in flight at once); any other semi-length runs the rolled loop
            switch (m) {
            case 3:  hb_stage<3>(src, dst, a.hb_taps + a.tap_off[i], n_out, tid); break;    // uniform global reads:
            case 5:  hb_stage<5>(src, dst, a.hb_taps + a.tap_off[i], n_out, tid); break;    // s_load, taps in SGPRs
            case 10: hb_stage<10>(src, dst, a.hb_taps + a.tap_off[i], n_out, tid); break;
            default: hb_stage<0>(src, dst, taps, n_out, tid, m); break;
            }
            __syncthreads();
        }

        // ------------------------------------------------------------ phase 3: polyphase + pack
        if (emit) {
            const int64_t qa = t * TG;
            int64_t qb = qa + TG; if (qb > a.n_groups) qb = a.n_groups;
            if (qb > qa) {
                const uint64_t k_lo = k_next;
                uint64_t k_hi;
                if (qb - qa == TG) {
                    k_hi = k_lo + a.n_est;
                    while (a.phi0 + k_hi * (uint64_t)a.step < ((uint64_t)qb << 24)) ++k_hi;
                } else {
                    k_hi = first_k_at((uint64_t)qb << 24, a.phi0, a.step);
                }
                const cf2 *hb = s_lvl + a.lvl_off[S] + kArbHist;
                for (uint64_t k = k_lo + tid; k < k_hi; k += kThreads) {
                    const uint64_t P = a.phi0 + k * (uint64_t)a.step;
                    const int ql = (int)((int64_t)(P >> 24) - qa);
                    const int arm = (int)((P >> 16) & 255u);
                    const cf2 *w = hb + ql;
                    const float *tp = s_arb + arm * 16;
                    float ar = 0.0f, ai = 0.0f;
#pragma unroll
                    for (int n = 0; n < 14; ++n) {
                        const cf2 sv = w[-n];
                        ar = fmaf(tp[n], sv.x, ar); ai = fmaf(tp[n], sv.y, ai);
                    }
                    cf2 y{ar, ai};
                    if (a.pnco_mode != 0)
                        y = nco_mix(y, nco_phasor(s_nco, a.pnco_theta0 + (uint32_t)k * a.pnco_dtheta), a.pnco_mode);
                    pack_store(a.out, (int64_t)k, a.out_fmt, y);
                }
                k_next = k_hi;
            }
        }
        __syncthreads();

        // ------------------------------------------------------------ phase 4: slide histories
        for (int i = wave; i <= S; i += 4) {
            const int H = lvl_hist(a, i), n_i = kTile >> i;
            cf2 *buf = s_lvl + a.lvl_off[i];
            if (lane < H) { const cf2 v = buf[n_i + lane]; buf[lane] = v; }
        }
        __syncthreads();
    }
}

size_t front_lds_bytes(const FrontArgs &a)
{
    size_t bytes = 1024 * sizeof(cf2) + 256 * 16 * sizeof(float) + (size_t)((a.n_hb_taps + 3) & ~3) * sizeof(float) + 8 * sizeof(cf2);
    if (a.mode == 1) bytes += (size_t)a.lvl_off[a.S + 1] * sizeof(cf2);
    return bytes;
}

hipError_t launch_front(const FrontArgs &a, int n_blocks, hipStream_t s)
{
    const size_t lds = front_lds_bytes(a);
    static LdsAttrCache cache;
    if (lds > 64 * 1024) { const hipError_t e = cache.ensure((const void *)k_front, lds); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(k_front, dim3((unsigned)n_blocks), dim3(kThreads), lds, s, a);
    return hipGetLastError();
}

// ============================================================================================
// DC-blocker carries
// ============================================================================================
// One workgroup per segment: A = sum_k c^(end-1-k) x[k] over the segment's new samples.
__global__ __launch_bounds__(kThreads) void k_dc_prefix(const DcPrefixArgs a)
{
    __shared__ float red[2 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sgm = blockIdx.x;
    const int64_t beg = dc_seg_start(a.geom, sgm);
    const int64_t end = (sgm == a.geom.n_seg - 1) ? a.geom.frames_in : dc_seg_start(a.geom, sgm + 1);
    const int64_t len = end - beg;
    float accr = 0.0f, acci = 0.0f;
    const int vb = (a.in_fmt == IQGPU_FMT_CS8 || a.in_fmt == IQGPU_FMT_CU8) ? 2
                 : (a.in_fmt == IQGPU_FMT_CS16 || a.in_fmt == IQGPU_FMT_CU16 || a.in_fmt == IQGPU_FMT_SC16Q11) ? 4
                 : (a.in_fmt == IQGPU_FMT_CF32) ? 8 : 0;
    if (len > 0) {
        // left-pad the segment to whole chunks of 1024 so that the last chunk ends at `end`
        const int64_t n_chunks = (len + 1023) >> 10;
        const int64_t pad = (n_chunks << 10) - len;
        const float c = a.c;
        const float c1024 = (float)exp(1024.0 * a.logc);
        int64_t ch = 0;
        // 16-bit formats, 16-byte aligned chunks: four chunks' loads in flight per thread (one at a time left the kernel at
        // 4.4 TB/s: 32 KiB in flight per CU; same arithmetic, same order)
        if (vb == 4 && a.raw_aligned && (((beg - pad) * 4) & 15) == 0 && n_chunks > 5) {
            {   // the first chunk holds the padding: the general path below, once
                const int64_t u = 4 * tid;
                float lr = 0.0f, li = 0.0f;
                const int64_t k0 = beg + u - pad;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    cf2 x{0.0f, 0.0f};
                    if (u + s >= pad) x = unpack_one(a.raw, k0 + s, a.in_fmt, a.gain);
                    lr = fmaf(lr, c, x.x); li = fmaf(li, c, x.y);
                }
                accr = fmaf(accr, c1024, lr); acci = fmaf(acci, c1024, li);
                ch = 1;
            }
            const float norm = (a.in_fmt == IQGPU_FMT_SC16Q11) ? 1.0f / 2048.0f : 1.0f / 32768.0f;
            const bool uns = a.in_fmt == IQGPU_FMT_CU16;
            const char *base = (const char *)a.raw + (beg - pad + 4 * tid) * 4;
            for (; ch + 4 <= n_chunks; ch += 4) {
                uint4 v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#if IQGPU_NT_DC
                    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
                    const u4v q = __builtin_nontemporal_load((const u4v *)(base + ((ch + i) << 12)));
                    v[i] = make_uint4(q.x, q.y, q.z, q.w);
#else
                    v[i] = *(const uint4 *)(base + ((ch + i) << 12));
#endif
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
                    float lr = 0.0f, li = 0.0f;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        float xr, xi;
                        if (uns) { xr = up_u((float)(w[s] & 0xffffu), 32767.5f, 1.0f / 32768.0f, a.gain); xi = up_u((float)(w[s] >> 16), 32767.5f, 1.0f / 32768.0f, a.gain); }
                        else { xr = up_s((float)(short)(w[s] & 0xffffu), norm, a.gain); xi = up_s((float)(short)(w[s] >> 16), norm, a.gain); }
                        lr = fmaf(lr, c, xr); li = fmaf(li, c, xi);
                    }
                    accr = fmaf(accr, c1024, lr); acci = fmaf(acci, c1024, li);
                }
            }
        }
        // 8-bit formats (late round 5: the loop below, one 8-byte load at a time, read an RTL-SDR's cu8 capture at 1.2 TB/s --
        // 0.42 ms per 2^28 frames, more than the resampler behind it): eight chunks' loads in flight per thread, same arithmetic, same order
        if (vb == 2 && a.raw_aligned && (((beg - pad) * 2) & 15) == 0 && n_chunks > 9) {
            {   // the first chunk holds the padding: the general path, once
                const int64_t u = 4 * tid;
                float lr = 0.0f, li = 0.0f;
                const int64_t k0 = beg + u - pad;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    cf2 x{0.0f, 0.0f};
                    if (u + s >= pad) x = unpack_one(a.raw, k0 + s, a.in_fmt, a.gain);
                    lr = fmaf(lr, c, x.x); li = fmaf(li, c, x.y);
                }
                accr = fmaf(accr, c1024, lr); acci = fmaf(acci, c1024, li);
                ch = 1;
            }
            const bool uns = a.in_fmt == IQGPU_FMT_CU8;
            const char *base = (const char *)a.raw + (beg - pad + 4 * tid) * 2;
            for (; ch + 8 <= n_chunks; ch += 8) {
                uint2 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    typedef uint32_t u2v __attribute__((ext_vector_type(2)));
                    const u2v q = __builtin_nontemporal_load((const u2v *)(base + ((ch + i) << 11)));
                    v[i] = make_uint2(q.x, q.y);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float lr = 0.0f, li = 0.0f;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const unsigned h = ((s & 2) ? v[i].y : v[i].x) >> (16 * (s & 1));
                        float xr, xi;
                        if (uns) { xr = up_u((float)(h & 0xffu), 127.5f, 1.0f / 128.0f, a.gain); xi = up_u((float)((h >> 8) & 0xffu), 127.5f, 1.0f / 128.0f, a.gain); }
                        else { xr = up_s((float)(signed char)(h & 0xffu), 1.0f / 128.0f, a.gain); xi = up_s((float)(signed char)((h >> 8) & 0xffu), 1.0f / 128.0f, a.gain); }
                        lr = fmaf(lr, c, xr); li = fmaf(li, c, xi);
                    }
                    accr = fmaf(accr, c1024, lr); acci = fmaf(acci, c1024, li);
                }
            }
        }
        for (; ch < n_chunks; ++ch) {
            const int64_t u = (ch << 10) + 4 * tid;       // padded position of this thread's 4 samples
            float lr = 0.0f, li = 0.0f;
            const int64_t k0 = beg + u - pad;
            cf2 xv[4];
            // one coalesced vector load when the four frames are real and 16-byte aligned
            const bool vec = vb != 0 && a.raw_aligned && u >= pad && ((k0 * vb) & 15) == 0 && unpack_four_fast(a.raw, k0, a.in_fmt, a.gain, xv);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                cf2 x{0.0f, 0.0f};
                if (vec) x = xv[s];
                else if (u + s >= pad) x = unpack_one(a.raw, k0 + s, a.in_fmt, a.gain);
                lr = fmaf(lr, c, x.x); li = fmaf(li, c, x.y);
            }
            accr = fmaf(accr, c1024, lr); acci = fmaf(acci, c1024, li);
        }
        const float wt = (float)exp((double)(1020 - 4 * tid) * a.logc);  // to the end of the chunk
        accr *= wt; acci *= wt;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { accr += __shfl_down(accr, o); acci += __shfl_down(acci, o); }
    if (lane == 0) { red[2 * wave] = accr; red[2 * wave + 1] = acci; }
    __syncthreads();
    if (tid == 0) {
        float r = 0.0f, i = 0.0f;
        for (int w = 0; w < 4; ++w) { r += red[2 * w]; i += red[2 * w + 1]; }
        a.agg[sgm] = cf2{r, i};
    }
}

hipError_t launch_dc_prefix(const DcPrefixArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(k_dc_prefix, dim3((unsigned)a.geom.n_seg), dim3(kThreads), 0, s, a);
    return hipGetLastError();
}

constexpr int kDcScanThreads = 1024;      // one workgroup, 3 segments a thread for a launch of 3 072 runs: 11.7 us (256 threads: 17.8 us; loading the
                                          // aggregates ahead of the folds changes nothing -- the folds' 64-bit index arithmetic and the barriers of the scan do)
// carry[s] = state before segment s; state <- state after the last sample.  The per-segment maps
// v -> f_s v + g_s (f_s = c^len_s, g_s = the segment's aggregate) compose associatively: each of the
// threads folds a contiguous slice of segments, the slices are scanned through LDS, then every thread
// replays its slice from its prefix.  All in double.
__global__ __launch_bounds__(kDcScanThreads) void k_dc_scan(const DcScanArgs a)
{
    __shared__ double sf[kDcScanThreads], sr[kDcScanThreads], si[kDcScanThreads];
    const int tid = threadIdx.x;
    const int n = a.geom.n_seg;
    const int per = (n + kDcScanThreads - 1) / kDcScanThreads;
    const int s0 = tid * per, s1 = (s0 + per < n) ? s0 + per : n;
    auto seg_len = [&](int sgm) {
        const int64_t beg = dc_seg_start(a.geom, sgm);
        const int64_t end = (sgm == n - 1) ? a.geom.frames_in : dc_seg_start(a.geom, sgm + 1);
        return end > beg ? end - beg : (int64_t)0;
    };
    // fold the slice: v_out = F v_in + (Gr, Gi)
    double F = 1.0, Gr = 0.0, Gi = 0.0;
    {
        int64_t prev_len = -1; double f = 1.0;
        for (int sgm = s0; sgm < s1; ++sgm) {
            const int64_t len = seg_len(sgm);
            if (len != prev_len) { f = exp((double)len * a.logc); prev_len = len; }
            const cf2 g = a.agg[sgm];
            F *= f; Gr = Gr * f + (double)g.x; Gi = Gi * f + (double)g.y;
        }
    }
    sf[tid] = F; sr[tid] = Gr; si[tid] = Gi;
    __syncthreads();
    // inclusive scan of the maps (later map applied after the earlier one)
    for (int o = 1; o < kDcScanThreads; o <<= 1) {
        double pf = 1.0, pr = 0.0, pi = 0.0;
        const bool has = tid >= o;
        if (has) { pf = sf[tid - o]; pr = sr[tid - o]; pi = si[tid - o]; }
        __syncthreads();
        if (has) { sr[tid] = sr[tid] + sf[tid] * pr; si[tid] = si[tid] + sf[tid] * pi; sf[tid] = sf[tid] * pf; }
        __syncthreads();
    }
    const double v0r = a.state->x, v0i = a.state->y;
    // state before this thread's slice = (inclusive scan of the previous thread) applied to v0
    double vr = v0r, vi = v0i;
    if (tid > 0) { vr = sf[tid - 1] * v0r + sr[tid - 1]; vi = sf[tid - 1] * v0i + si[tid - 1]; }
    {
        int64_t prev_len = -1; double f = 1.0;
        for (int sgm = s0; sgm < s1; ++sgm) {
            a.carry[sgm] = cd2{vr, vi};
            const int64_t len = seg_len(sgm);
            if (len != prev_len) { f = exp((double)len * a.logc); prev_len = len; }
            const cf2 g = a.agg[sgm];
            vr = vr * f + (double)g.x; vi = vi * f + (double)g.y;
        }
    }
    __syncthreads();
    if (tid == kDcScanThreads - 1) { a.state->x = sf[tid] * v0r + sr[tid]; a.state->y = sf[tid] * v0i + si[tid]; }
}

hipError_t launch_dc_scan(const DcScanArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(k_dc_scan, dim3(1), dim3(kDcScanThreads), 0, s, a);
    return hipGetLastError();
}

// ============================================================================================
// k_fir: y[i] = sum_k h[k] f[(L-1) + i - k]   (firfilt / fftfilt are the same linear convolution,
// SPEC B.2/B.3); taps and the matching data window are staged through LDS in chunks of
// kFirTapChunk taps; each thread owns four outputs 256 apart so LDS reads are conflict free.
// ============================================================================================
__global__ __launch_bounds__(kThreads) void k_fir(const FirArgs a)
{
    __shared__ __align__(16) cf2 s_w[kFirOutTile + kFirTapChunk];
    __shared__ __align__(16) cf2 s_h[kFirTapChunk];
    __shared__ __align__(16) cf2 s_nco[1024];
    const int tid = threadIdx.x;
    const int64_t o0 = (int64_t)blockIdx.x * kFirOutTile;
    const int L = a.ntaps;
    if (a.pnco_mode != 0) for (int i = tid; i < 1024; i += kThreads) s_nco[i] = a.nco_tab[i];
    if (a.move_n > 0 && blockIdx.x == gridDim.x - 1)
        for (int64_t i = tid; i < a.move_n; i += kThreads) a.move_dst[i] = a.move_src[i];

    float ar[4] = {0, 0, 0, 0}, ai[4] = {0, 0, 0, 0};
    for (int k0 = 0; k0 < L; k0 += kFirTapChunk) {
        __syncthreads();
        // window element e <-> f index base + e, base = (L-1) + o0 - k0 - (C-1)
        const int64_t base = (int64_t)(L - 1) + o0 - k0 - (kFirTapChunk - 1);
        const int64_t fmax = (int64_t)(L - 1) + a.n_emit;   // valid f indices are [0, fmax)
        for (int e = tid; e < kFirOutTile + kFirTapChunk - 1; e += kThreads) {
            const int64_t fi = base + e;
            s_w[e] = (fi >= 0 && fi < fmax) ? a.fbuf[fi] : cf2{0.0f, 0.0f};
        }
        for (int e = tid; e < kFirTapChunk; e += kThreads)
            s_h[e] = (k0 + e < L) ? a.taps[k0 + e] : cf2{0.0f, 0.0f};
        __syncthreads();
        const int kn = (L - k0 < kFirTapChunk) ? L - k0 : kFirTapChunk;
        if (a.is_complex) {
            for (int kk = 0; kk < kn; ++kk) {
                const cf2 h = s_h[kk];
                const cf2 *p = s_w + tid + (kFirTapChunk - 1) - kk;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const cf2 x = p[256 * r];
                    ar[r] = fmaf(h.x, x.x, ar[r]); ar[r] = fmaf(-h.y, x.y, ar[r]);
                    ai[r] = fmaf(h.x, x.y, ai[r]); ai[r] = fmaf(h.y, x.x, ai[r]);
                }
            }
        } else {
            for (int kk = 0; kk < kn; ++kk) {
                const float h = s_h[kk].x;
                const cf2 *p = s_w + tid + (kFirTapChunk - 1) - kk;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const cf2 x = p[256 * r];
                    ar[r] = fmaf(h, x.x, ar[r]); ai[r] = fmaf(h, x.y, ai[r]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t i = o0 + tid + 256 * r;
        if (i < a.n_emit) {
            cf2 y{ar[r], ai[r]};
            if (a.pnco_mode != 0)
                y = nco_mix(y, nco_phasor(s_nco, a.pnco_theta0 + (uint32_t)i * a.pnco_dtheta), a.pnco_mode);
            pack_store(a.out, i, a.out_fmt, y);
        }
    }
}

hipError_t launch_fir(const FirArgs &a, hipStream_t s)
{
    if (a.n_emit <= 0) return hipSuccess;
    const unsigned nb = (unsigned)((a.n_emit + kFirOutTile - 1) / kFirOutTile);
    hipLaunchKernelGGL(k_fir, dim3(nb), dim3(kThreads), 0, s, a);
    return hipGetLastError();
}

__global__ void k_copy_cf(cf2 *dst, const cf2 *src, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

hipError_t launch_copy_cf(cf2 *dst, const cf2 *src, int64_t n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    unsigned nb = (unsigned)((n + 255) / 256); if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_copy_cf, dim3(nb), dim3(256), 0, s, dst, src, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// k_iq_probe: the block the reference hands to its I/Q optimiser -- the first 1024 samples of a chunk after
// unpack -> [dc block] -> [iq correct] -> [pre NCO]  (src/pipeline.c:468-476 copies them out of buffer A right
// behind pre_processor_apply_chain).  One wavefront; the dc recurrence (1024 steps) runs on one lane.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_iq_probe(const IqProbeArgs a)
{
    __shared__ cf2 x[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) x[i] = unpack_one(a.raw, i, a.in_fmt, a.gain);
    __syncthreads();
    if (a.dc_enable && lane == 0) {
        const cd2 st = *a.dc_state;                // v[n-1] in front of this call's first sample
        float vr = (float)st.x, vi = (float)st.y;
        const float aa = 1.0f - a.dc_c;
        for (int i = 0; i < 1024; ++i) {
            const cf2 v = x[i];
            x[i] = cf2{fmaf(-aa, vr, v.x), fmaf(-aa, vi, v.y)};
            vr = fmaf(a.dc_c, vr, v.x); vi = fmaf(a.dc_c, vi, v.y);
        }
    }
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) {
        cf2 v = x[i];
        if (a.iq_enable) { const float re = v.x; v.x = re * a.iq_magp1; v.y = fmaf(a.iq_phase, re, v.y); }
        if (a.nco_mode != 0) v = nco_mix(v, nco_phasor(a.nco_tab, a.nco_theta0 + (uint32_t)i * a.nco_dtheta), a.nco_mode);
        a.out[i] = v;
    }
}

hipError_t launch_iq_probe(const IqProbeArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(k_iq_probe, dim3(1), dim3(64), 0, s, a);
    return hipGetLastError();
}

} // namespace iqgpu
