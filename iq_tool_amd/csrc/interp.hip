// interp.hip -- k_interp: msresamp_crcf for ratios r >= 1 (src/resampler.c:27, 51; SPEC B.6):
//   arbitrary 256-arm polyphase resampler at rate_arb in [1, 2]  ->  S half-band interpolators
//   -> [post NCO] -> pack.
// It runs behind the pointwise front stage (and the pre-resample user filter, src/filter.c:43-92
// places the filter BEFORE the resampler whenever the chain does not decimate), on a cf32 buffer
// laid out like the filter's: [hist samples of the previous calls][this call's new samples].
//
// Every stage is a pure function of the stream position, so a workgroup rebuilds what it needs:
// for a tile of kInterpTile final outputs it walks the cascade backwards once (ext[s] = extra
// samples of level s in front of the tile; host, make_interp_geometry) and then forwards through
// LDS, level by level:
//   level 0      = arbitrary-resampler outputs k:  P_k = phi0 + k step (24-bit fraction),
//                  input index q = P_k >> 24, arm = (P_k >> 16) & 255, 14 taps ending at x[q]
//   level s+1[u] = u even: level s[i - m]                                   (delay branch)
//                  u odd : sum_t h[2t+1] level s[i - 2m + 1 + t], i = u>>1  (filter branch)
// Outputs with k < 0 belong to earlier calls and are recomputed from the history samples;
// before the start of the stream everything is zero, as the reference's zeroed windows are.
#include <hip/hip_runtime.h>

#include "../../include/iqgpu.h"
#include "dsp_device.hpp"
#include "kernels.hpp"

namespace iqgpu {

// filter branch of one half-band interpolator: sum_t h[2t+1] x[i - 2m + 1 + t], t < 2m.
// M > 0: compile-time semi-length (all 2M window reads in flight at once, two accumulator chains).
template <int M>
__device__ __forceinline__ cf2 interp_branch(const cf2 *p, const float *taps, int m_rt)
{
    float ar = 0.0f, ai = 0.0f;
    if (M > 0) {
        cf2 sv[M > 0 ? 2 * M : 1];
#pragma unroll
        for (int q = 0; q < 2 * M; ++q) sv[q] = p[q];
        float br = 0.0f, bi = 0.0f;
#pragma unroll
        for (int q = 0; q < 2 * M; q += 2) {
            ar = fmaf(taps[q], sv[q].x, ar); ai = fmaf(taps[q], sv[q].y, ai);
            br = fmaf(taps[q + 1], sv[q + 1].x, br); bi = fmaf(taps[q + 1], sv[q + 1].y, bi);
        }
        ar += br; ai += bi;
    } else {
        for (int q = 0; q < 2 * m_rt; ++q) {
            const cf2 sv = p[q];
            ar = fmaf(taps[q], sv.x, ar); ai = fmaf(taps[q], sv.y, ai);
        }
    }
    return cf2{ar, ai};
}

// one frame of a 4-byte / 2-byte output format as the word pack_store would write (dsp_device.hpp: the same expressions, case by case)
__device__ __forceinline__ uint32_t interp_word32(int fmt, cf2 v)
{
    if (fmt == IQGPU_FMT_CU16) return pk_unsigned(v.x, 32767.0f, 32767.5f, 65535.0f) | (pk_unsigned(v.y, 32767.0f, 32767.5f, 65535.0f) << 16);
    const float s = (fmt == IQGPU_FMT_CS16) ? 32767.0f : 2048.0f;
    return ((unsigned)pk_signed(v.x, s, -32768.0f, 32767.0f) & 0xffffu) | (((unsigned)pk_signed(v.y, s, -32768.0f, 32767.0f) & 0xffffu) << 16);
}
__device__ __forceinline__ uint32_t interp_word16(int fmt, cf2 v)
{
    if (fmt == IQGPU_FMT_CU8) return pk_unsigned(v.x, 127.0f, 127.5f, 255.0f) | (pk_unsigned(v.y, 127.0f, 127.5f, 255.0f) << 8);
    return ((unsigned)pk_signed(v.x, 127.0f, -128.0f, 127.0f) & 0xffu) | (((unsigned)pk_signed(v.y, 127.0f, -128.0f, 127.0f) & 0xffu) << 8);
}

__global__ __launch_bounds__(kThreads) void k_interp(const InterpArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const int S = a.S;
    cf2   *s_nco = (cf2 *)smem;                              // 1024 {cos, sin}
    float *s_arb = (float *)(s_nco + 1024);                  // [256][16]
    float *s_hb  = s_arb + 256 * 16;                         // branch taps of every stage
    cf2   *s_in  = (cf2 *)(s_hb + ((a.n_hb_taps + 3) & ~3)); // staged resampler input
    cf2   *s_lvl = s_in + a.in_cap;                          // levels 0 .. S-1 (level S goes to memory)

    if (a.pnco_mode != 0) for (int i = tid; i < 1024; i += kThreads) s_nco[i] = a.nco_tab[i];
    // polyphase taps in rows of 14 floats, read as seven 8-byte pairs (the XOR-folded rows of k_front_s1 measured 3 % slower
    // here: consecutive outputs of an interpolator sit on few distinct arms)
    for (int i = tid; i < 256 * kArbWin; i += kThreads) {
        const int arm = i / kArbWin, k = i % kArbWin;
        s_arb[arm * kArbWin + k] = a.arb_table[arm * 16 + k];
    }
    for (int i = tid; i < a.n_hb_taps; i += kThreads) s_hb[i] = a.hb_taps[i];
    if (a.move_n > 0 && blockIdx.x == gridDim.x - 1)
        for (int64_t i = tid; i < a.move_n; i += kThreads) a.move_dst[i] = a.move_src[i];

    const int n0 = kInterpTile >> S;                         // resampler outputs per tile
    const int64_t step = (int64_t)a.step;
    const int64_t phi = (int64_t)a.phi0;

    for (int64_t t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
        __syncthreads();                                     // previous tile done with the LDS
        const int64_t j0 = t * kInterpTile;                  // first final output of the tile
        const int64_t k0 = j0 >> S;
        // ---- stage the input window of resampler outputs [k0 - ext0, k0 + n0) ----
        const int64_t k_lo = k0 - a.ext[0];
        const int n_k = n0 + a.ext[0];
        const int64_t q_base = ((phi + k_lo * step) >> 24) - (kArbWin - 1);
        int64_t q_top = (phi + (k_lo + n_k - 1) * step) >> 24;
        if (q_top > a.n_in - 1) q_top = a.n_in - 1;
        const int n_q = (int)(q_top - q_base + 1);
        for (int i = tid; i < n_q; i += kThreads) {
            const int64_t bi = q_base + i + a.hist;          // index into xbuf
            s_in[i] = (bi >= 0) ? a.xbuf[bi] : cf2{0.0f, 0.0f};
        }
        __syncthreads();
        // ---- level 0: arbitrary resampler ----
        // Round 6: a thread takes the PAIR of outputs (2 idx, 2 idx + 1).  Their windows are the same 14 samples or one sample apart
        // (the stage's step is at most one sample: rate_arb >= 1), so 15 window reads serve both where 28 did -- the kernel is bound
        // by its LDS reads (14 window + 7 tap reads per output until now) -- and the second output picks its samples out of the
        // shared registers by the one-bit distance d.  Same products in the same order (tap 0 first, started from zero).
        for (int idx = tid; 2 * idx < n_k; idx += kThreads) {
            const int i0 = 2 * idx;
            const bool has1 = i0 + 1 < n_k;
            const int64_t k0 = k_lo + i0;
            const int64_t P0 = phi + k0 * step, P1 = P0 + step;
            const int64_t q0 = P0 >> 24, q1 = P1 >> 24;
            cf2 y0{0.0f, 0.0f}, y1{0.0f, 0.0f};
            if (q0 <= q_top) {                               // later inputs have not arrived yet
                const cf2 *w = s_in + (int)(q0 - q_base) - (kArbWin - 1);          // w[j] = x[q0 - 13 + j], j = 0 .. 14
                cf2 sr[kArbWin + 1];
#pragma unroll
                for (int j = 0; j <= kArbWin; ++j) sr[j] = w[j];
                const float2 *ta = (const float2 *)(s_arb + (int)((P0 >> 16) & 255) * kArbWin);     // 56-byte rows: 8-byte reads
                float ar = 0.0f, ai = 0.0f;
#pragma unroll
                for (int n2 = 0; n2 < kArbWin / 2; ++n2) {
                    const float2 t2 = ta[n2];
                    const cf2 s0 = sr[kArbWin - 1 - 2 * n2], s1 = sr[kArbWin - 2 - 2 * n2];
                    ar = fmaf(t2.x, s0.x, ar); ai = fmaf(t2.x, s0.y, ai);
                    ar = fmaf(t2.y, s1.x, ar); ai = fmaf(t2.y, s1.y, ai);
                }
                y0 = cf2{ar, ai};
                if (has1 && q1 <= q_top) {
                    const bool d = q1 != q0;                 // the second output's newest sample: x[q0 + 1] or x[q0]
                    const float2 *tb = (const float2 *)(s_arb + (int)((P1 >> 16) & 255) * kArbWin);
                    float br = 0.0f, bi = 0.0f;
#pragma unroll
                    for (int n2 = 0; n2 < kArbWin / 2; ++n2) {
                        const float2 t2 = tb[n2];
                        const cf2 s0 = d ? sr[kArbWin - 2 * n2] : sr[kArbWin - 1 - 2 * n2];
                        const cf2 s1 = d ? sr[kArbWin - 1 - 2 * n2] : sr[kArbWin - 2 - 2 * n2];
                        br = fmaf(t2.x, s0.x, br); bi = fmaf(t2.x, s0.y, bi);
                        br = fmaf(t2.y, s1.x, br); bi = fmaf(t2.y, s1.y, bi);
                    }
                    y1 = cf2{br, bi};
                }
            }
            if (S == 0) {
                const bool st0 = k0 < a.n_arb, st1 = has1 && k0 + 1 < a.n_arb;      // ext[0] = 0: k0 >= 0
                if (a.pnco_mode != 0) {
                    y0 = nco_mix(y0, nco_phasor(s_nco, a.pnco_theta0 + (uint32_t)k0 * a.pnco_dtheta), a.pnco_mode);
                    y1 = nco_mix(y1, nco_phasor(s_nco, a.pnco_theta0 + (uint32_t)(k0 + 1) * a.pnco_dtheta), a.pnco_mode);
                }
                const int of = a.out_fmt;
                if (st0 && st1 && of == IQGPU_FMT_CF32) {
                    typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));
                    *(f4a8 *)((char *)a.out + 8 * k0) = f4a8{y0.x, y0.y, y1.x, y1.y};
                } else if (st0 && st1 && (of == IQGPU_FMT_CS16 || of == IQGPU_FMT_SC16Q11 || of == IQGPU_FMT_CU16)) {
                    typedef uint32_t u2a4 __attribute__((ext_vector_type(2), aligned(4)));
                    *(u2a4 *)((char *)a.out + 4 * k0) = u2a4{interp_word32(of, y0), interp_word32(of, y1)};
                } else if (st0 && st1 && (of == IQGPU_FMT_CS8 || of == IQGPU_FMT_CU8)) {
                    typedef uint32_t u1a2 __attribute__((aligned(2)));
                    *(u1a2 *)((char *)a.out + 2 * k0) = interp_word16(of, y0) | (interp_word16(of, y1) << 16);
                } else {
                    if (st0) pack_store(a.out, k0, of, y0);
                    if (st1) pack_store(a.out, k0 + 1, of, y1);
                }
            } else {
                s_lvl[a.lvl_off[0] + i0] = y0;
                if (has1) s_lvl[a.lvl_off[0] + i0 + 1] = y1;
            }
        }
        // ---- half-band interpolators ----
        for (int s = 0; s < S; ++s) {
            __syncthreads();
            const int m = a.m[s];
            const cf2 *src = s_lvl + a.lvl_off[s];
            const float *taps = s_hb + a.tap_off[s];
            const bool last = (s + 1 == S);
            const int e1 = last ? 0 : a.ext[s + 1];
            const int n1 = (kInterpTile >> (S - s - 1)) + e1;
            const int64_t u_lo = (j0 >> (S - s - 1)) - e1;            // first index at level s+1
            const int64_t i_base = (j0 >> (S - s)) - a.ext[s];        // level-s index of src[0]
            cf2 *dst = last ? nullptr : s_lvl + a.lvl_off[s + 1];
            // a thread takes the PAIR (u even, u + 1): the delay branch and the filter branch of one input index -- every lane
            // runs the filter branch once (interleaved even / odd lanes had each wave execute it with half its lanes idle)
            const int64_t u_al = u_lo & ~(int64_t)1;
            const int n_pair = (int)((u_lo + n1 - u_al + 1) >> 1);
            for (int idx = tid; idx < n_pair; idx += kThreads) {
                const int64_t u = u_al + 2 * (int64_t)idx;               // even
                const int li = (int)((u >> 1) - i_base);
                const bool has_e = u >= u_lo, has_o = u + 1 < u_lo + n1;
                cf2 ye = cf2{0.0f, 0.0f}, yo = cf2{0.0f, 0.0f};
                if (has_e) ye = src[li - m];
                if (has_o) {
                    const cf2 *p = src + li - 2 * m + 1;
                    const float *tg = a.hb_taps + a.tap_off[s];        // uniform global reads: taps in SGPRs
                    switch (m) {
                    case 3:  yo = interp_branch<3>(p, tg, m); break;
                    case 5:  yo = interp_branch<5>(p, tg, m); break;
                    case 10: yo = interp_branch<10>(p, tg, m); break;
                    default: yo = interp_branch<0>(p, taps, m); break;
                    }
                }
                if (last) {
                    const bool st_e = has_e && u < a.n_emit, st_o = has_o && u + 1 < a.n_emit;
                    cf2 y0 = ye, y1 = yo;
                    if (a.pnco_mode != 0) {
                        y0 = nco_mix(y0, nco_phasor(s_nco, a.pnco_theta0 + (uint32_t)u * a.pnco_dtheta), a.pnco_mode);
                        y1 = nco_mix(y1, nco_phasor(s_nco, a.pnco_theta0 + (uint32_t)(u + 1) * a.pnco_dtheta), a.pnco_mode);
                    }
                    // Round 6: the pair leaves as ONE piece where the format allows it (two stores of one frame each had every cache line
                    // written twice: the store shape of DESIGN 3.3); the same expressions as pack_store
                    const int of = a.out_fmt;
                    if (st_e && st_o && of == IQGPU_FMT_CF32) {
                        typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));
                        *(f4a8 *)((char *)a.out + 8 * u) = f4a8{y0.x, y0.y, y1.x, y1.y};
                    } else if (st_e && st_o && (of == IQGPU_FMT_CS16 || of == IQGPU_FMT_SC16Q11 || of == IQGPU_FMT_CU16)) {
                        typedef uint32_t u2a4 __attribute__((ext_vector_type(2), aligned(4)));
                        *(u2a4 *)((char *)a.out + 4 * u) = u2a4{interp_word32(of, y0), interp_word32(of, y1)};
                    } else if (st_e && st_o && (of == IQGPU_FMT_CS8 || of == IQGPU_FMT_CU8)) {
                        typedef uint32_t u1a2 __attribute__((aligned(2)));
                        *(u1a2 *)((char *)a.out + 2 * u) = interp_word16(of, y0) | (interp_word16(of, y1) << 16);
                    } else {
                        if (st_e) pack_store(a.out, u, of, y0);
                        if (st_o) pack_store(a.out, u + 1, of, y1);
                    }
                } else {
                    if (has_e) dst[u - u_lo] = ye;
                    if (has_o) dst[u + 1 - u_lo] = yo;
                }
            }
        }
    }
}

// ext[s], LDS offsets and capacities for a chain of S interpolators with semi-lengths m[0..S)
// (run order: m[0] is the lowest-rate stage).  Returns the input history the kernel needs.
int make_interp_geometry(InterpArgs &a)
{
    const int S = a.S;
    a.ext[S] = 0;
    for (int s = S - 1; s >= 0; --s) a.ext[s] = (a.ext[s + 1] + 1) / 2 + 2 * a.m[s] - 1;
    int off = 0;
    for (int s = 0; s < S; ++s) {
        a.lvl_off[s] = off;
        off += (kInterpTile >> (S - s)) + a.ext[s];
        off = (off + 1) & ~1;
    }
    a.lvl_off[S] = off;
    const uint64_t n_k = (uint64_t)((kInterpTile >> S) + a.ext[0]);
    a.in_cap = (int)((((n_k * a.step) >> 24) + kArbWin + 4) & ~(uint64_t)1);
    return (int)((((uint64_t)a.ext[0] * a.step) >> 24) + 1 + kArbWin);
}

hipError_t launch_interp(const InterpArgs &a, int n_cu, hipStream_t s)
{
    if (a.n_emit <= 0) return hipSuccess;
    const size_t lds = 1024 * sizeof(cf2) + 256 * 16 * sizeof(float) + (size_t)((a.n_hb_taps + 3) & ~3) * sizeof(float) +
                       (size_t)(a.in_cap + a.lvl_off[a.S]) * sizeof(cf2);
    int64_t grid = a.n_tiles;
    const int64_t cap = (int64_t)n_cu * 4;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(k_interp, dim3((unsigned)grid), dim3(kThreads), lds, s, a);
    return hipGetLastError();
}

} // namespace iqgpu
