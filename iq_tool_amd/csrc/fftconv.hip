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
// k_fftconv16: the same overlap-save block for N >= 1024 with radix-16 passes held in registers:
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

// in place in ONE buffer: every pass pulls its points into registers, all threads meet, then the
// autosorted results go back to the same buffer (half the LDS of a ping-pong pair -> twice the
// workgroups per CU; one more barrier per pass)
__device__ __forceinline__ void fft16_lds(cf2 *buf, const cf2 *tw, int N, int log2n, int tid, int T)
{
    int Ns = 1;
    if (log2n & 1) {                                                  // radix-2, Ns = 1: no twiddles
        const int nb = N >> 1;                                        // 8 butterflies per thread
        cf2 v0[8], v1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int j = tid + i * T; v0[i] = buf[sw(j)]; v1[i] = buf[sw(j + nb)]; }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = tid + i * T;
            buf[sw(2 * j)] = cf2{v0[i].x + v1[i].x, v0[i].y + v1[i].y};
            buf[sw(2 * j + 1)] = cf2{v0[i].x - v1[i].x, v0[i].y - v1[i].y};
        }
        __syncthreads();
        Ns = 2;
    }
    if (log2n & 2) {                                                  // radix-4: 4 butterflies per thread
        const int nb = N >> 2, tstride = N / (Ns * 4);
        cf2 v[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = tid + i * T, k = j & (Ns - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[i][r] = buf[sw(j + r * nb)];
            if (Ns > 1) {
#pragma unroll
                for (int r = 1; r < 4; ++r) v[i][r] = cmulf(v[i][r], tw[k * r * tstride]);
            }
            dft4(v[i][0], v[i][1], v[i][2], v[i][3]);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = tid + i * T, k = j & (Ns - 1), j0 = (j - k) * 4 + k;
#pragma unroll
            for (int r = 0; r < 4; ++r) buf[sw(j0 + r * Ns)] = v[i][r];
        }
        __syncthreads();
        Ns *= 4;
    }
    while (Ns < N) {                                                  // radix-16, one butterfly per thread
        const int j = tid, k = j & (Ns - 1), tstride = N / (Ns * 16);
        cf2 v[16], w[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = buf[sw(j + r * T)];
        if (Ns > 1) {
#pragma unroll
            for (int r = 1; r < 16; ++r) w[r] = tw[k * r * tstride];
#pragma unroll
            for (int r = 1; r < 16; ++r) v[r] = cmulf(v[r], w[r]);
        }
        dft16(v);
        const int j0 = (j - k) * 16 + k;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) buf[sw(j0 + r * Ns)] = v[(r >> 2) + 4 * (r & 3)];
        __syncthreads();
        Ns *= 16;
    }
}

__global__ __launch_bounds__(1024) void k_fftconv16(const FftConvArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, T = blockDim.x;                     // T = N / 16
    const int N = 1 << a.log2n, L1 = a.ntaps - 1, V = N - L1;
    const int NP = N + (N >> 5) + 2;
    cf2 *X = (cf2 *)smem;
    cf2 *s_nco = X + NP;
    if (a.pnco_mode != 0) for (int i = tid; i < 1024; i += T) s_nco[i] = a.nco_tab[i];

    const int64_t o0 = (int64_t)blockIdx.x * V;
#pragma unroll 4
    for (int p = tid; p < N; p += T) {
        const int64_t fi = o0 + p;
        X[sw(p)] = (fi < a.fbuf_len) ? a.fbuf[fi] : cf2{0.0f, 0.0f};
    }
    __syncthreads();
    fft16_lds(X, a.twiddle, N, a.log2n, tid, T);
#pragma unroll 4
    for (int p = tid; p < N; p += T) {
        const cf2 z = cmulf(X[sw(p)], a.hfreq[p]);
        X[sw(p)] = cf2{z.x, -z.y};
    }
    __syncthreads();
    fft16_lds(X, a.twiddle, N, a.log2n, tid, T);
    cf2 *Y = X;
    const int64_t left = a.n_emit - o0;
    const int nv = left < (int64_t)V ? (int)left : V;
    for (int i = tid; i < nv; i += T) {
        cf2 y = Y[sw(L1 + i)];
        y.y = -y.y;
        const int64_t k = o0 + i;
        if (a.pnco_mode != 0)
            y = nco_mix(y, nco_phasor(s_nco, a.pnco_theta0 + (uint32_t)k * a.pnco_dtheta), a.pnco_mode);
        pack_store(a.out, k, a.out_fmt, y);
    }
}

hipError_t launch_fftconv(const FftConvArgs &a, hipStream_t s)
{
    if (a.n_emit > 0 && a.log2n >= 10 && !getenv("IQGPU_FFT_NO_R16")) {
        const int N = 1 << a.log2n, V = N - (a.ntaps - 1);
        if (V <= 0 || N > kMaxFftN) return hipErrorInvalidValue;
        const unsigned nb = (unsigned)((a.n_emit + V - 1) / V);
        const size_t lds = (size_t)(N + (N >> 5) + 2) * sizeof(cf2) + (a.pnco_mode != 0 ? 1024 * sizeof(cf2) : 0);
        static LdsAttrCache cache16;
        if (lds > 64 * 1024) { const hipError_t e = cache16.ensure((const void *)k_fftconv16, lds); if (e != hipSuccess) return e; }
        hipLaunchKernelGGL(k_fftconv16, dim3(nb), dim3(N / 16), lds, s, a);
        return hipGetLastError();
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
