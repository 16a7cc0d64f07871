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
#include "fft16.hpp"

namespace iqgpu {

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
    const int L1 = a.ntaps - 1;
    // (win / vout: a window of fewer stream samples than points and fewer outputs than it could give -- k_p0fft16's geometry, for
    //  the byte-for-byte comparison of the two paths; 0 = the whole transform)
    const int WIN = a.win > 0 ? a.win : N, V = a.vout > 0 ? a.vout : N - L1;
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
        // window samples this thread may read: p = tid + i T < room
        int64_t room = a.fbuf_len - o0 - tid;
        if (room > (int64_t)(WIN - tid)) room = WIN - tid;
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
    fftconv16_tail<LOG2N, false>(a, X, s_nco, tid, io, o0, V, L1);
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

bool p0fft_geometry(int log2n, int ntaps, int *win, int *vout)
{
    if (log2n != 12) return false;                                   // (the instantiated transform: four waves per workgroup, two workgroups per CU)
    const int N = 1 << log2n, L1 = ntaps - 1;
    constexpr int kStep = 320;                                       // (k_front_p0's / k_p0fft16's outputs per step of a wave: front_p0_common.hpp)
    const int w = N / kStep * kStep;
    const int v = w - L1;                                            // (the tap rows are let go of block by block: nothing ties the blocks to the step grid)
    if (L1 < 1 || v < 4 * kStep) return false;                     // (more than a quarter of every window computed twice: not worth it)
    *win = w; *vout = v;
    return true;
}

bool p0fft_shape(const FrontArgs &front, int log2n, int ntaps)
{
    int w, v;
    FrontArgs fa = front;
    fa.out_fmt = IQGPU_FMT_CF32; fa.agc_fused = 0;
    return front_p0_shape(fa) && p0fft_geometry(log2n, ntaps, &w, &v);
}

// the fused launch: every operand shape the kernel and its grid assume is checked HERE, on the host
static hipError_t launch_p0fft(const FftConvArgs &a, hipStream_t s)
{
    int w = 0, v = 0;
    const P0Feed &f = a.feed;
    if (!p0fft_geometry(a.log2n, a.ntaps, &w, &v) || a.win != w || a.vout != v) return hipErrorInvalidValue;
    if (f.frames_in < 0 || f.n_res < 0 || f.pre < (int64_t)(a.ntaps - 1) || f.k_a < 0 || f.k_b > f.n_res || f.grid < 1 || f.hist_cap < 0) return hipErrorInvalidValue;
    if (a.fbuf_len != f.pre + f.n_res || a.n_emit < 0 || a.n_emit > a.fbuf_len) return hipErrorInvalidValue;
    if (f.write_state && (a.move_n < 0 || (a.move_n > 0 && (a.move_src - a.fbuf) + a.move_n != a.fbuf_len))) return hipErrorInvalidValue;
    if (a.n_emit == 0 && !f.write_state) return hipSuccess;
    const int N = 1 << a.log2n;
    const size_t lds = (size_t)(((N + (N >> 5) + 2) * 8 + 15) / 16 * 16) + p0fft_tap_lds() + (a.pnco_mode != 0 ? 1024 * sizeof(cf2) : 0);
    if (f.in_fmt == IQGPU_FMT_CU8) return launch_p0fft_cu8(a, lds, s);
    if (f.in_fmt == IQGPU_FMT_CS8) return launch_p0fft_cs8(a, lds, s);
    if (f.in_fmt == IQGPU_FMT_CS16) return launch_p0fft_cs16(a, lds, s);
    return hipErrorInvalidValue;
}

hipError_t launch_fftconv(const FftConvArgs &a, hipStream_t s)
{
    if (a.feed.raw) return launch_p0fft(a, s);
    if ((a.win > 0 || a.vout > 0) && !(a.log2n >= 10 && !(a.dbg & kDbgFftNoR16) && a.win >= a.ntaps && a.win <= (1 << a.log2n) && a.vout >= 1
        && a.vout <= a.win - (a.ntaps - 1))) return hipErrorInvalidValue;
    if ((a.agc_fused || a.run_if) && !(a.log2n >= 10 && !(a.dbg & kDbgFftNoR16))) return hipErrorInvalidValue;   // (the radix-16 kernel only)
    if (a.n_emit > 0 && a.log2n >= 10 && !(a.dbg & kDbgFftNoR16)) {
        const int N = 1 << a.log2n, V = a.vout > 0 ? a.vout : N - (a.ntaps - 1);
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
