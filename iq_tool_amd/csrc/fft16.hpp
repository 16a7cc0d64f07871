// fft16.hpp -- the in-LDS radix-16 transforms of the overlap-save user filter and everything of a filter block behind its window
// (spectrum product, inverse transform, epilogue): shared by k_fftconv16 (fftconv.hip: window read from the cf32 stream) and
// k_p0fft16 (p0fft.hpp: window computed by the resampler inside the kernel).  Device code only; see fftconv.hip for the design.
#pragma once
#include <cstdlib>
#include <type_traits>
#include <hip/hip_runtime.h>

#include "../../include/iqgpu.h"
#include "dsp_device.hpp"
#include "kernels.hpp"

namespace iqgpu {

// LDS index swizzle: one pad element per 32.  The autosort writes of the early passes are strided
// (stride 4 Ns elements); without the pad 65 % of the kernel's LDS cycles were bank conflicts
// (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE on config 3).
__device__ __forceinline__ int sw(int i) { return i + (i >> 5); }

__device__ __forceinline__ cf2 cmulf(cf2 a, cf2 b)
{
    return cf2{fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x)};
}

// ---------------------------------------------------------------------------------------------
// k_fftconv16<log2 N>: the same overlap-save block for N >= 1024 with radix-16 passes held in registers:
// N / 16 threads, one 16-point butterfly per thread per pass, so a 4096-point transform is 3 LDS
// round trips (+ barriers) instead of 6, and the 15 twiddles of a butterfly are fetched together
// with its 16 points.  N = 2^a 16^b: the leading factor (2, 4 or 8) runs first as radix-2 / radix-4.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void dft4(cf2 &a0, cf2 &a1, cf2 &a2, cf2 &a3)
{
    const cf2 s02{a0.x + a2.x, a0.y + a2.y}, d02{a0.x - a2.x, a0.y - a2.y};
    const cf2 s13{a1.x + a3.x, a1.y + a3.y};
    const cf2 d13{a1.y - a3.y, a3.x - a1.x};                       // (a1 - a3) * (-i)
    a0 = cf2{s02.x + s13.x, s02.y + s13.y};
    a1 = cf2{d02.x + d13.x, d02.y + d13.y};
    a2 = cf2{s02.x - s13.x, s02.y - s13.y};
    a3 = cf2{d02.x - d13.x, d02.y - d13.y};
}

// in place; X[m + 4n] ends up in v[n + 4m]
__device__ __forceinline__ void dft16(cf2 v[16])
{
    // W16^e = exp(-2 pi i e / 16)
    const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
#pragma unroll
    for (int c = 0; c < 4; ++c) dft4(v[c], v[c + 4], v[c + 8], v[c + 12]);     // u[c][m] -> v[c + 4m]
    // u[c][m] *= W16^(c m)
    v[1 + 4] = cmulf(v[1 + 4], cf2{c1, -s1});          // e = 1
    v[2 + 4] = cmulf(v[2 + 4], cf2{h, -h});            // e = 2
    v[3 + 4] = cmulf(v[3 + 4], cf2{s1, -c1});          // e = 3
    v[1 + 8] = cmulf(v[1 + 8], cf2{h, -h});            // e = 2
    v[2 + 8] = cf2{v[2 + 8].y, -v[2 + 8].x};           // e = 4: * (-i)
    v[3 + 8] = cmulf(v[3 + 8], cf2{-h, -h});           // e = 6
    v[1 + 12] = cmulf(v[1 + 12], cf2{s1, -c1});        // e = 3
    v[2 + 12] = cmulf(v[2 + 12], cf2{-h, -h});         // e = 6
    v[3 + 12] = cmulf(v[3 + 12], cf2{-c1, s1});        // e = 9
#pragma unroll
    for (int m = 0; m < 4; ++m) dft4(v[4 * m], v[4 * m + 1], v[4 * m + 2], v[4 * m + 3]);
}

// In place in ONE buffer: every pass pulls its points into registers, all threads meet, then the
// autosorted results go back to the same buffer (half the LDS of a ping-pong pair -> twice the
// workgroups per CU; one more barrier per pass).  N is a template parameter: with T = N / 16 and the
// sub-transform size Ns known at compile time, the padded index of point r of a butterfly is
//   sw(j + r T)    = sw(j)  + r T  + (r T  >> 5)      (T a multiple of 32)
//   sw(j0 + r Ns)  = sw(j0) + r Ns + (r Ns >> 5)      (Ns a power of two, k < Ns, 16 (j - k) a multiple of 16 Ns)
// i.e. one base address per pass and constant offsets (the run-time-N version spent more VALU
// instructions on these indices than on the butterflies: 1250 integer against 770 floating-point).
// The 15 twiddles of a radix-16 butterfly, W^r with W = exp(-2 pi i k / (16 Ns)): four of them (r = 1, 2, 4, 8) come
// from the table in global memory -- fetched one pass AHEAD, before the barriers of the pass in front, so that
// their latency never sits between two passes -- and the other eleven are products of two or three of those
// (error <= 3 roundings of exactly rounded table values, ~2e-7).  Fetching all fifteen where they are used made
// the kernel latency-bound: 0.27 ms on config 4 against 0.15 ms with no twiddle loads at all.
struct Tw4 { cf2 w1, w2, w4, w8; };

template <int N, int Ns>
__device__ __forceinline__ Tw4 load_tw4(const cf2 *tw, int tid)
{
    Tw4 t{cf2{1.f, 0.f}, cf2{1.f, 0.f}, cf2{1.f, 0.f}, cf2{1.f, 0.f}};
    if constexpr (Ns > 1 && Ns < N) {
        constexpr int tstride = N / (Ns * 16);
        const int kt = (tid & (Ns - 1)) * tstride;
        t.w1 = tw[kt]; t.w2 = tw[2 * kt]; t.w4 = tw[4 * kt]; t.w8 = tw[8 * kt];
    }
    return t;
}

// FIRST: the pass's input is in registers -- io[r] = point tid + r T, which is what the window load (and the last pass of the
// transform in front) leaves in a thread -- instead of LDS; WAIT: other threads may still be reading the buffer (the inverse
// transform: the forward one's last pass), so the writes wait for a barrier.  The LAST pass (Ns = T: k = j, the autosorted
// destination of thread j is point j + r T again) leaves its results in io[] in natural order and writes nothing.  Window load,
// spectrum product, and output therefore never touch LDS: 4 LDS round trips and 7 barriers per block instead of 8 and 14.
template <int N, int Ns, bool FIRST, bool WAIT>
__device__ __forceinline__ void r16_passes(cf2 *buf, const cf2 *tw, int tid, const Tw4 cur, cf2 (&io)[16])
{
    if constexpr (Ns < N) {
        constexpr int T = N / 16;
        constexpr bool LAST = Ns * 16 == N;
        const int j = tid, k = j & (Ns - 1);
        cf2 v[16];
        if constexpr (FIRST) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = io[r];
        } else {
            const cf2 *src = buf + sw(j);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = src[r * T + ((r * T) >> 5)];
        }
        if (Ns > 1) {
            const cf2 w3 = cmulf(cur.w1, cur.w2), w5 = cmulf(cur.w4, cur.w1), w6 = cmulf(cur.w4, cur.w2), w7 = cmulf(cur.w4, w3);
            v[1] = cmulf(v[1], cur.w1); v[2] = cmulf(v[2], cur.w2); v[3] = cmulf(v[3], w3); v[4] = cmulf(v[4], cur.w4);
            v[5] = cmulf(v[5], w5); v[6] = cmulf(v[6], w6); v[7] = cmulf(v[7], w7); v[8] = cmulf(v[8], cur.w8);
            v[9] = cmulf(v[9], cmulf(cur.w8, cur.w1)); v[10] = cmulf(v[10], cmulf(cur.w8, cur.w2));
            v[11] = cmulf(v[11], cmulf(cur.w8, w3)); v[12] = cmulf(v[12], cmulf(cur.w8, cur.w4));
            v[13] = cmulf(v[13], cmulf(cur.w8, w5)); v[14] = cmulf(v[14], cmulf(cur.w8, w6));
            v[15] = cmulf(v[15], cmulf(cur.w8, w7));
        }
        dft16(v);
        if constexpr (LAST) {
            static_assert(Ns == T, "the last radix-16 pass has Ns = N / 16");
#pragma unroll
            for (int r = 0; r < 16; ++r) io[r] = v[(r >> 2) + 4 * (r & 3)];
        } else {
            const Tw4 nxt = load_tw4<N, Ns * 16>(tw, tid);           // issued in front of the barriers
            cf2 *dst = buf + sw((j - k) * 16 + k);
            if constexpr (!FIRST || WAIT) __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[r * Ns + ((r * Ns) >> 5)] = v[(r >> 2) + 4 * (r & 3)];
            __syncthreads();
            r16_passes<N, Ns * 16, false, false>(buf, tw, tid, nxt, io);
        }
    }
}

// io[r]: in = point tid + r T of the sequence, out = point tid + r T of its transform
template <int LOG2N, bool WAIT>
__device__ __forceinline__ void fft16_lds(cf2 *buf, const cf2 *tw, int tid, cf2 (&io)[16])
{
    constexpr int N = 1 << LOG2N, T = N / 16;
    constexpr int Ns2 = (LOG2N & 1) ? 2 : 1;                          // after the radix-2 pass
    constexpr int Ns4 = (LOG2N & 2) ? Ns2 * 4 : Ns2;                  // after the radix-4 pass
    static_assert(Ns4 * 16 <= N, "at least one radix-16 pass");
    const Tw4 first = load_tw4<N, Ns4>(tw, tid);                      // of the first radix-16 pass: in flight under the passes in front
    if constexpr ((LOG2N & 1) != 0) {                                 // radix-2, Ns = 1: no twiddles; points tid + i T and N/2 + tid + i T
        if constexpr (WAIT) __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = tid + i * T;
            buf[sw(2 * j)] = cf2{io[i].x + io[8 + i].x, io[i].y + io[8 + i].y};
            buf[sw(2 * j + 1)] = cf2{io[i].x - io[8 + i].x, io[i].y - io[8 + i].y};
        }
        __syncthreads();
    }
    if constexpr ((LOG2N & 2) != 0) {                                 // radix-4: 4 butterflies per thread
        constexpr int Ns = Ns2, nb = N >> 2, tstride = N / (Ns * 4);
        constexpr bool from_regs = (LOG2N & 1) == 0;                  // the first pass: point tid + i T + r N/4 = io[i + 4 r]
        cf2 v[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = tid + i * T, k = j & (Ns - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (from_regs) v[i][r] = io[i + 4 * r];
                else v[i][r] = buf[sw(j) + r * nb + ((r * nb) >> 5)];
            }
            if (Ns > 1) {
#pragma unroll
                for (int r = 1; r < 4; ++r) v[i][r] = cmulf(v[i][r], tw[k * r * tstride]);
            }
            dft4(v[i][0], v[i][1], v[i][2], v[i][3]);
        }
        if constexpr (!from_regs || WAIT) __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = tid + i * T, k = j & (Ns - 1), j0 = (j - k) * 4 + k;
#pragma unroll
            for (int r = 0; r < 4; ++r) buf[sw(j0 + r * Ns)] = v[i][r];
        }
        __syncthreads();
    }
    if constexpr (Ns4 == 1) r16_passes<N, 1, true, WAIT>(buf, tw, tid, first, io);
    else r16_passes<N, Ns4, false, false>(buf, tw, tid, first, io);
}

// everything of a block behind its window: io[i] = window sample tid + i T  ->  forward transform, spectrum product, inverse, and the
// epilogue over the block's outputs o0 .. o0 + V - 1 (result points L1 .. L1 + V - 1): [post NCO], [fused AGC], pack.
// WAIT0: other threads may still be reading the transform buffer when the forward transform's first pass wants to write it
template <int LOG2N, bool WAIT0>
__device__ __forceinline__ void fftconv16_tail(const FftConvArgs &a, cf2 *X, const cf2 *s_nco, const int tid, cf2 (&io)[16], const int64_t o0, const int V, const int L1)
{
    constexpr int N = 1 << LOG2N, T = N / 16;
    fft16_lds<LOG2N, WAIT0>(X, a.twiddle, tid, io);
    {
        const cf2 *ph = a.hfreq + tid;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const cf2 z = cmulf(io[i], ph[i * T]);
            io[i] = cf2{z.x, -z.y};
        }
    }
    fft16_lds<LOG2N, true>(X, a.twiddle, tid, io);
    const int64_t left = a.n_emit - o0;
    const int nv = left < (int64_t)V ? (int)left : V;
    // fused AGC of the locked phase: the chunk of the block's first output and where the next one starts, in closed form from the
    // block index alone (wave-uniform: the scalar unit's work); at most two chunks meet in a block (launch_fftconv's condition)
    float agc_g = 1.0f, m0 = 0.0f, m1 = 0.0f;
    int64_t agc_c0 = 0, agc_c1 = 0, agc_b1 = 0;
    if (a.agc_fused) {
        agc_g = a.agc_state->gain;
        agc_c0 = agc_chunk_of_output(a.agc_geom, o0);
        agc_b1 = agc_out_end(a.agc_geom, agc_c0);
        agc_c1 = agc_b1 < o0 + nv ? agc_chunk_of_output(a.agc_geom, agc_b1) : agc_c0 + 1;      // (chunks without an output lie between)
    }
    // the output format is chosen ONCE per block (the switch inside the loop was a chain of scalar compares and branches per output:
    // sixteen times per thread)
    auto emit = [&](auto fmt_tag) {
        constexpr int F = decltype(fmt_tag)::value;                  // -1: whatever a.out_fmt says (pack_store)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = tid + r * T - L1;                          // output i of the block is point L1 + i of the result
            if (i < 0 || i >= nv) continue;
            cf2 y = io[r];
            y.y = -y.y;
            const int64_t k = o0 + i;
            if (a.pnco_mode != 0)
                y = nco_mix(y, nco_phasor(s_nco, a.pnco_theta0 + (uint32_t)k * a.pnco_dtheta), a.pnco_mode);
            if (a.agc_fused) {
                // agc_apply: the chunk's peak over the samples BEFORE the gain, then samples[i] *= g (src/agc.c:169-214)
                const float m2 = fmaf(y.x, y.x, y.y * y.y);
                if (k < agc_b1) m0 = fmaxf(m0, m2); else m1 = fmaxf(m1, m2);
                y = cf2{y.x * agc_g, y.y * agc_g};
            }
            if (F == IQGPU_FMT_CS16) ((uint32_t *)a.out)[k] = pack_cs16(y);
            else if (F == IQGPU_FMT_CU8) ((uint16_t *)a.out)[k] = (uint16_t)pack_b8(y, true);
            else if (F == IQGPU_FMT_CS8) ((uint16_t *)a.out)[k] = (uint16_t)pack_b8(y, false);
            else if (F == IQGPU_FMT_CF32) ((cf2 *)a.out)[k] = y;
            else pack_store(a.out, k, a.out_fmt, y);
        }
    };
    switch (a.out_fmt) {
    case IQGPU_FMT_CS16: emit(std::integral_constant<int, IQGPU_FMT_CS16>{}); break;
    case IQGPU_FMT_CU8:  emit(std::integral_constant<int, IQGPU_FMT_CU8>{}); break;
    case IQGPU_FMT_CS8:  emit(std::integral_constant<int, IQGPU_FMT_CS8>{}); break;
    case IQGPU_FMT_CF32: emit(std::integral_constant<int, IQGPU_FMT_CF32>{}); break;
    default:             emit(std::integral_constant<int, -1>{}); break;
    }
    if (a.agc_fused) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { m0 = fmaxf(m0, __shfl_xor(m0, o)); m1 = fmaxf(m1, __shfl_xor(m1, o)); }
        if ((tid & 63) == 0) {
            if (m0 > 0.0f) atomicMax(a.agc_peak2 + agc_c0, (unsigned long long)__double_as_longlong((double)m0));
            if (m1 > 0.0f) atomicMax(a.agc_peak2 + agc_c1, (unsigned long long)__double_as_longlong((double)m1));
        }
    }
}

} // namespace iqgpu
