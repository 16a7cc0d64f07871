// fftconv.hip -- k_fftconv: block convolution for the FFT-kind user filter (fftfilt_crcf / fftfilt_cccf,
// src/filter.c:339-342, 464-526) as overlap-SAVE in LDS.
//
// liquid's fftfilt is overlap-add with a 2n-point FFT per n-sample block and firfilt is the direct form
// (SPEC B.3); both are the same linear convolution, only fftfilt's block-quantised output COUNT is
// observable and that is applied by the host (plan_call).  So the kernel's transform size N is a
// free tuning choice, independent of the reference's block size, and long FIR-kind filters take
// this path too.  One workgroup owns V = N - (L-1) consecutive outputs:
//   window  = the N filter-input samples ending with them (the first L-1 are history),
//   X       = FFT_N(window)             Stockham autosort, radix 4 (+ one radix-2 pass), ping-pong in LDS
//   Y       = IFFT_N(X . H)             H = FFT_N(taps) / N, computed once on the host in double
//   outputs = Y[L-1 .. N)               (the part of the circular convolution without wrap-around),
//                                       then [post NCO] and pack.
// Twiddles come from an N-entry table in global memory (L2-resident; double-precision values
// rounded to float -- v_sin/v_cos are not accurate enough for the 1e-5 budget).
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

// Stockham autosort passes over N points in LDS, sub-transform size Ns -> Ns*R per pass.
// tw[k] = exp(-2 pi i k / N).  A thread owns at most kFftMaxB butterflies per pass (the launcher
// sizes the workgroup accordingly); the twiddles of pass p+1 are fetched from the (L2-resident)
// table while pass p computes, so that their latency is off the pass-to-pass critical path.
constexpr int kFftMaxB = 2;

struct TwSet { cf2 w[kFftMaxB][3]; };

__device__ __forceinline__ void load_twiddles(TwSet &t, const cf2 *tw, int N, int Ns, int tid, int nthr)
{
    const int nb = N >> 2, tstride = N / (Ns * 4);
#pragma unroll
    for (int i = 0; i < kFftMaxB; ++i) {
        const int j = tid + i * nthr;
        const int k = (j < nb) ? (j & (Ns - 1)) : 0;
#pragma unroll
        for (int r = 1; r < 4; ++r) t.w[i][r - 1] = tw[k * r * tstride];
    }
}

__device__ __forceinline__ void radix4_pass(const cf2 *src, cf2 *dst, const TwSet &t, int N, int Ns, int tid, int nthr)
{
    const int nb = N >> 2;
#pragma unroll
    for (int i = 0; i < kFftMaxB; ++i) {
        const int j = tid + i * nthr;
        if (j < nb) {
            const int k = j & (Ns - 1);
            cf2 v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = src[sw(j + r * nb)];
            if (Ns > 1) {
#pragma unroll
                for (int r = 1; r < 4; ++r) v[r] = cmulf(v[r], t.w[i][r - 1]);
            }
            const int j0 = (j - k) * 4 + k;
            const cf2 a{v[0].x + v[2].x, v[0].y + v[2].y}, b{v[0].x - v[2].x, v[0].y - v[2].y};
            const cf2 c{v[1].x + v[3].x, v[1].y + v[3].y};
            const cf2 d{v[1].y - v[3].y, v[3].x - v[1].x};             // (v1 - v3) * (-i)
            dst[sw(j0)] = cf2{a.x + c.x, a.y + c.y};
            dst[sw(j0 + Ns)] = cf2{b.x + d.x, b.y + d.y};
            dst[sw(j0 + 2 * Ns)] = cf2{a.x - c.x, a.y - c.y};
            dst[sw(j0 + 3 * Ns)] = cf2{b.x - d.x, b.y - d.y};
        }
    }
}

// forward transform of the N points in buf0 (ping-pong with buf1); returns the buffer holding the result
__device__ __forceinline__ cf2 *fft_lds(cf2 *buf0, cf2 *buf1, const cf2 *tw, int N, int log2n, int tid, int nthr)
{
    cf2 *src = buf0, *dst = buf1;
    int Ns = 1;
    TwSet cur, nxt;
    if (log2n & 1) {
        // one radix-2 pass first (no twiddles at Ns = 1); meanwhile fetch the twiddles of the Ns = 2 pass
        load_twiddles(cur, tw, N, 2, tid, nthr);
        const int nb = N >> 1;
        for (int j = tid; j < nb; j += nthr) {
            const cf2 v0 = src[sw(j)], v1 = src[sw(j + nb)];
            dst[sw(2 * j)] = cf2{v0.x + v1.x, v0.y + v1.y};
            dst[sw(2 * j + 1)] = cf2{v0.x - v1.x, v0.y - v1.y};
        }
        __syncthreads();
        Ns = 2;
        cf2 *t = src; src = dst; dst = t;
    } else {
        load_twiddles(cur, tw, N, 1, tid, nthr);
    }
    while (Ns < N) {
        if (Ns * 4 < N) load_twiddles(nxt, tw, N, Ns * 4, tid, nthr);
        radix4_pass(src, dst, cur, N, Ns, tid, nthr);
        __syncthreads();
        Ns *= 4;
        cur = nxt;
        cf2 *t = src; src = dst; dst = t;
    }
    return src;
}

__global__ __launch_bounds__(kFftMaxThreads) void k_fftconv(const FftConvArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int N = 1 << a.log2n, L1 = a.ntaps - 1, V = N - L1;
    const int NP = N + (N >> 5) + 2;                        // padded length (sw)
    cf2 *buf0 = (cf2 *)smem, *buf1 = buf0 + NP;
    cf2 *s_nco = buf1 + NP;                                  // only when the post NCO is on
    if (a.pnco_mode != 0) for (int i = tid; i < 1024; i += nthr) s_nco[i] = a.nco_tab[i];
    if (a.move_n > 0 && blockIdx.x == gridDim.x - 1)
        for (int64_t i = tid; i < a.move_n; i += nthr) a.move_dst[i] = a.move_src[i];

    // window sample p <-> filter-input stream index s = blk*V - L1 + p <-> fbuf[L1 + s]
    const int64_t o0 = (int64_t)blockIdx.x * V;
    for (int p = tid; p < N; p += nthr) {
        const int64_t fi = o0 + p;
        buf0[sw(p)] = (fi < a.fbuf_len) ? a.fbuf[fi] : cf2{0.0f, 0.0f};
    }
    __syncthreads();
    cf2 *X = fft_lds(buf0, buf1, a.twiddle, N, a.log2n, tid, nthr);
    cf2 *other = (X == buf0) ? buf1 : buf0;
    // Y = conj(FFT(conj(X . H))), H already carries the 1/N
    for (int p = tid; p < N; p += nthr) {
        const cf2 z = cmulf(X[sw(p)], a.hfreq[p]);
        X[sw(p)] = cf2{z.x, -z.y};
    }
    __syncthreads();
    cf2 *Y = fft_lds(X, other, a.twiddle, N, a.log2n, tid, nthr);
    const int64_t left = a.n_emit - o0;
    const int nv = left < (int64_t)V ? (int)left : V;
    for (int i = tid; i < nv; i += nthr) {
        cf2 y = Y[sw(L1 + i)];
        y.y = -y.y;
        const int64_t k = o0 + i;
        if (a.pnco_mode != 0)
            y = nco_mix(y, nco_phasor(s_nco, a.pnco_theta0 + (uint32_t)k * a.pnco_dtheta), a.pnco_mode);
        pack_store(a.out, k, a.out_fmt, y);
    }
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

// LOOP: the conditional fallback behind a fused launch (run_if): a bounded grid whose workgroups walk the blocks -- when the verdict
// stands (the rule) a few thousand workgroups leave at once, where one workgroup per block of a 2^28-frame call -- 179 000 single
// waves at N = 1024 -- took 20 - 45 us to come and go
template <int LOG2N, bool LOOP = false>
// (four waves per SIMD: 4 workgroups of N = 4096, 2 of N = 8192 -- what their LDS allows -- need 128 VGPRs or fewer)
__global__ __launch_bounds__((1 << LOG2N) / 16) __attribute__((amdgpu_waves_per_eu(4))) void k_fftconv16(const FftConvArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int N = 1 << LOG2N, T = N / 16;                        // blockDim.x = T
    if (a.run_if && *a.run_if == 0) return;                          // (the cf32 fallback behind a fused launch whose verdict stood)
    const int tid = threadIdx.x;
    const int L1 = a.ntaps - 1, V = N - L1;
    constexpr int NP = N + (N >> 5) + 2;
    cf2 *X = (cf2 *)smem;
    cf2 *s_nco = X + NP;
    if (a.pnco_mode != 0) for (int i = tid; i < 1024; i += T) s_nco[i] = a.nco_tab[i];   // (read behind the barriers of the transforms)
    const int64_t n_blocks = LOOP ? (a.n_emit + V - 1) / V : (int64_t)gridDim.x;
    if (a.move_n > 0 && blockIdx.x == gridDim.x - 1)                 // the next call's history, into the other buffer of the pair
        for (int64_t i = tid; i < a.move_n; i += T) a.move_dst[i] = a.move_src[i];

  int64_t blk = blockIdx.x;
  do {
    const int64_t o0 = blk * V;
    cf2 io[16];                                                      // point tid + i T of the window / spectrum / result
    {
        const cf2 *srcg = a.fbuf + o0 + tid;
        const bool nt = 4 * L1 <= N;
        const int64_t room = a.fbuf_len - o0 - tid;                  // window samples this thread may read: p = tid + i T < room
#pragma unroll
        for (int i = 0; i < 16; ++i)
        {
            // a window shares its first ntaps - 1 samples with the block in front: where that is a small part of it (nt: at most
            // a quarter) the stream is as good as read once and takes the non-temporal hint (config 3: step 0.767 -> 0.744 ms;
            // with half of every window shared -- config 4 -- the hint costs 1.5 %)
            typedef float f2v __attribute__((ext_vector_type(2)));
            cf2 vin{0.0f, 0.0f};
            if ((int64_t)(i * T) < room) {
                if (nt) { const f2v q = __builtin_nontemporal_load((const f2v *)(srcg + i * T)); vin = cf2{q.x, q.y}; }
                else vin = srcg[i * T];
            }
            io[i] = vin;
        }
    }
    fft16_lds<LOG2N, false>(X, a.twiddle, tid, io);
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
    blk += (int64_t)gridDim.x;
    if (LOOP) __syncthreads();                                       // (the walk: the transform buffer is free again)
  } while (LOOP && blk < n_blocks);
}

bool fftconv_agc_fusable(int log2n, int ntaps, uint32_t dbg)
{
    return log2n >= 10 && (1 << log2n) <= kMaxFftN && (1 << log2n) > ntaps - 1 && !(dbg & (kDbgFftNoR16 | kDbgAgcNoFuse));
}

template <int LOG2N>
static hipError_t launch_fftconv16(const FftConvArgs &a, unsigned nb, size_t lds, hipStream_t s)
{
    static LdsAttrCache cache16, cache16l;
    if (a.run_if) {
        // the conditional fallback: about four rounds of resident workgroups, each walking its share of the blocks
        const unsigned cap = 256u * 16u * 1024u / (unsigned)(1 << LOG2N) * 4u;
        if (lds > 64 * 1024) { const hipError_t e = cache16l.ensure((const void *)k_fftconv16<LOG2N, true>, lds); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL((k_fftconv16<LOG2N, true>), dim3(nb < cap ? nb : cap), dim3((1 << LOG2N) / 16), lds, s, a);
        return hipGetLastError();
    }
    if (lds > 64 * 1024) { const hipError_t e = cache16.ensure((const void *)k_fftconv16<LOG2N>, lds); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(k_fftconv16<LOG2N>, dim3(nb), dim3((1 << LOG2N) / 16), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_fftconv(const FftConvArgs &a, hipStream_t s)
{
    if ((a.agc_fused || a.run_if) && !(a.log2n >= 10 && !(a.dbg & kDbgFftNoR16))) return hipErrorInvalidValue;   // (the radix-16 kernel only)
    if (a.n_emit > 0 && a.log2n >= 10 && !(a.dbg & kDbgFftNoR16)) {
        const int N = 1 << a.log2n, V = N - (a.ntaps - 1);
        if (V <= 0 || N > kMaxFftN) return hipErrorInvalidValue;
        const unsigned nb = (unsigned)((a.n_emit + V - 1) / V);
        const size_t lds = (size_t)(N + (N >> 5) + 2) * sizeof(cf2) + (a.pnco_mode != 0 ? 1024 * sizeof(cf2) : 0);
        switch (a.log2n) {
        case 10: return launch_fftconv16<10>(a, nb, lds, s);
        case 11: return launch_fftconv16<11>(a, nb, lds, s);
        case 12: return launch_fftconv16<12>(a, nb, lds, s);
        case 13: return launch_fftconv16<13>(a, nb, lds, s);
        case 14: return launch_fftconv16<14>(a, nb, lds, s);
        default: return hipErrorInvalidValue;
        }
    }
    if (a.n_emit <= 0) return hipSuccess;
    const int N = 1 << a.log2n, V = N - (a.ntaps - 1);
    if (V <= 0 || N > kMaxFftN4) return hipErrorInvalidValue;
    const unsigned nb = (unsigned)((a.n_emit + V - 1) / V);
    const size_t lds = (size_t)2 * (N + (N >> 5) + 2) * sizeof(cf2) + (a.pnco_mode != 0 ? 1024 * sizeof(cf2) : 0);
    static LdsAttrCache cache;
    if (lds > 64 * 1024) { const hipError_t e = cache.ensure((const void *)k_fftconv, lds); if (e != hipSuccess) return e; }
    int nthr = a.threads > 0 ? a.threads : (N >= 8192 ? 1024 : (N >= 4096 ? 512 : 256));
    if (nthr > kFftMaxThreads) nthr = kFftMaxThreads;
    while (nthr * kFftMaxB < (N >> 2)) nthr *= 2;                   // at most kFftMaxB radix-4 butterflies per thread
    hipLaunchKernelGGL(k_fftconv, dim3(nb), dim3(nthr), lds, s, a);
    return hipGetLastError();
}

} // namespace iqgpu
