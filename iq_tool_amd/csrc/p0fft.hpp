// p0fft.hpp -- k_p0fft16: resampler (no half-band stage) -> overlap-save user filter in ONE kernel (round 6).  A header, because the
// instantiations (three input formats x six step classes) are spread over three translation units -- p0fft_cu8.hip, p0fft_cs8.hip,
// p0fft_cs16.hip -- that compile side by side (one unit took longer than the rest of the library together).  The host-side checks
// and the dispatch live in fftconv.hip (launch_fftconv).
#pragma once
#include "fft16.hpp"
#include "front_p0_common.hpp"

namespace iqgpu {

// ---------------------------------------------------------------------------------------------
// k_p0fft16<log2 N, input format, step class> (round 6): resampler -> user filter WITHOUT the cf32 stream between
// them.  The shipped cu8-nrsc5-usb / -lsb presets (iq_tool_presets.conf:198-239; placement src/filter.c:53-90; post_processor.c:9-36)
// ran k_front_p0 with cf32 output (1.33 GB written per 2^28 frames) and k_fftconv16 reading it back with its overlap (1.64 GB) -- for
// a chain whose own input + output is 0.83 GB: 0.09 of the HBM rate.  A window sample of the filter IS a polyphase output, so the
// workgroup that owns a block of the filter computes its window itself, the way k_front_p0 computes outputs -- a step of a wave =
// 320 consecutive window samples, five per lane, each lane loading and unpacking the 22 raw frames around its five, tap rows held in
// registers and re-read under an EXEC mask only where the (position, arm) key moved -- straight into the transform's LDS buffer;
// one more LDS round trip hands every thread its sixteen points, and the block goes on as k_fftconv16's (fftconv16_tail: the same
// transforms, product and epilogue, hence the same bytes as k_fftconv16 on the same windows).
//   * Geometry: a block's window holds `win` = a whole number of steps of stream samples (N = 4096: 12 steps = 3840, the last 256
//     points zero) and emits `vout` = the largest multiple of 320 <= win - (L - 1): from block to block and from step to step a
//     lane-slot moves on by a multiple of 320 outputs, so for the NRSC-5 step (320 x 1.6125 = 516 - 0.001 samples) its arm moves by a
//     quarter of an arm and the masked re-reads stay rare (k_front_p0's observation); the L - 1 + (win - L + 1 - vout) samples two
//     neighbouring windows share are computed twice (9 % at 193 taps).
//   * Persistent workgroups (the tap planes -- 35 KB of LDS -- are filled once): two per CU of four waves each, walking the blocks.
//   * Edges: window samples in front of the call's first output are the filter's history and pending samples (fbuf, KB); outputs
//     whose frames reach into the stream history (hist_in) or past the call's end are computed one by one from guarded frames
//     (pp_generic: the slot routines' sum without their zero taps) -- a handful of steps per call.  The grid's last workgroup also leaves the next call's state: hist_out
//     and the filter's buffer front (move_dst), recomputed the slow way.
// ---------------------------------------------------------------------------------------------
template <int LOG2N, int FMT, int L3, int L4, int L2>
// (two waves per SIMD: two workgroups of N = 4096, one of N = 8192 -- what their LDS allows -- at 256 VGPRs or fewer)
__global__ __launch_bounds__((1 << LOG2N) / 16) __attribute__((amdgpu_waves_per_eu(2))) void k_p0fft16(const FftConvArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int N = 1 << LOG2N, T = N / 16, NWAVES = T / 64;
    constexpr int BPS = (FMT == IQGPU_FMT_CS16) ? 4 : 2;
    constexpr int NW = (BPS == 2) ? 12 : 24;                         // raw words per window: 24 frames loaded, 22 used
    constexpr int NS = 5;
    constexpr int LO[5] = {0, 1, L2, L3, L4};
    constexpr int NP = N + (N >> 5) + 2;
    constexpr int XB = (NP * 8 + 15) / 16 * 16;
    typedef __attribute__((address_space(3))) const v2f lds_v2f;
    typedef uint32_t u4v __attribute__((ext_vector_type(4), aligned(2)));
    if (a.run_if && *a.run_if == 0) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const P0Feed &f = a.feed;
    const int L1 = a.ntaps - 1, W = a.win, V = a.vout;
    cf2 *X = (cf2 *)smem;
    float *s_tap = (float *)(smem + XB);
    cf2 *s_nco = (cf2 *)(smem + XB + kFTapLds);
    fill_tap_planes(s_tap, f.arb_table, tid, T, f.tap_fold != 0);
    if (a.pnco_mode != 0) for (int i = tid; i < 1024; i += T) s_nco[i] = a.nco_tab[i];
    __syncthreads();
    const unsigned tap_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)s_tap;
    const uint32_t step = f.step;

    v2f t[NS][8];                                                    // the slots' shifted tap rows, kept from step to step of a block
    uint32_t held[NS];                                               // ... and the (position, arm) each was loaded for
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        held[j] = 0xffffffffu;
#pragma unroll
        for (int i = 0; i < 8; ++i) t[j][i] = v2f{0.f, 0.f};
    }
    // which tap rows the outputs at phase Pq need anew: re-read under their lanes' mask (front_p0.hip)
    auto reload = [&](const uint64_t Pq) {
        const uint32_t Fq = (uint32_t)Pq & 0xffffffu;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const uint32_t pj = Fq + (uint32_t)j * step;
            const uint32_t key = pj >> 16;
            if (key != held[j]) {
                const unsigned row = f.tap_fold ? tap_row<true>(tap_lds, pj, LO[j]) : tap_row<false>(tap_lds, pj, LO[j]);
#pragma unroll
                for (int i = 0; i < 8; ++i) t[j][i] = *(lds_v2f *)(size_t)(row + tap_pair_off(i));
                held[j] = key;
            }
        }
    };
    auto fetch = [&](const uint64_t Pq, uint32_t (&r)[NW]) {
        const int64_t f0 = (int64_t)(Pq >> 24) - 13;
        const char *src = (const char *)f.raw + f0 * BPS;
#pragma unroll
        for (int q = 0; q < NW / 4; ++q) {
            const u4v v = *(const u4v *)(src + 16 * q);
            r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
        }
    };
    // five outputs from their window: the chains of the other kernels, slot by slot (same products, same order)
    auto compute5 = [&](const v2f (&Hw)[14], const v2f (&own)[9], v2f (&y)[NS]) {
        pp_slots3<9, 0, 1, L2, true>(Hw, own, t[0], t[1], t[2], y[0], y[1], y[2]);
        pp_slots2<9, L3, L4, true>(Hw, own, t[3], t[4], y[3], y[4]);
    };
    // processed sample i of the stream, call-relative: the history of earlier calls in front of 0, nothing behind the call's end
    auto edge_sample = [&](const int64_t i) -> v2f {
        if (i < 0) {
            const int64_t h = (int64_t)f.hist_cap + i;
            if (h < 0) return v2f{0.f, 0.f};
            const cf2 v = f.hist_in[h];
            return v2f{v.x, v.y};
        }
        if (i >= f.frames_in) return v2f{0.f, 0.f};
        uint32_t w[1];
        if (BPS == 2) { w[0] = *(const uint16_t *)((const char *)f.raw + 2 * i); return p0_unpack<FMT>(w, 0); }
        w[0] = (uint32_t)*(const uint16_t *)((const char *)f.raw + 4 * i) | ((uint32_t)*(const uint16_t *)((const char *)f.raw + 4 * i + 2) << 16);
        return p0_unpack<FMT>(w, 0);
    };
    // One output of the resampler the slow way: every frame of its window fetched by itself (history, input, or nothing), the arm's
    // taps from the chain's table in global memory, a rolled loop -- the sum in ascending tap order started by the first product,
    // which is what the slot routines compute once their zero taps are left out (front_fat_common.hpp: T_d[w] = tap[13 + d - w] on
    // sample LOJ - 13 + w, w descending).  A few hundred outputs per call; kept small so that it costs the fast path no registers.
    auto pp_generic = [&](const int64_t k) -> v2f {
        const uint64_t P = f.phi0 + (uint64_t)k * (uint64_t)step;
        const int64_t q = (int64_t)(P >> 24);
        const float *tp = f.arb_table + (((uint32_t)P >> 16) & 255u) * 16;
        v2f acc = mul2(tp[0], edge_sample(q));
#pragma unroll 1
        for (int i = 1; i < 14; ++i) acc = fma2(tp[i], edge_sample(q - i), acc);
        return acc;
    };
    // window entry fi (index of the filter's input buffer) the slow way: history / pending samples from fbuf, then the resampler's
    // outputs, then nothing
    auto slow1 = [&](const int64_t fi) -> v2f {
        if (fi < f.pre) { const cf2 q = a.fbuf[fi]; return v2f{q.x, q.y}; }
        const int64_t k = fi - f.pre;
        if (k < f.n_res) return pp_generic(k);
        return v2f{0.f, 0.f};
    };

    // ---- the next call's state, by the grid's last workgroup (in front of its blocks) ----
    if (f.write_state && blockIdx.x == gridDim.x - 1) {
        for (int i = tid; i < f.hist_cap; i += T) {
            const v2f v = edge_sample(f.frames_in - (int64_t)f.hist_cap + i);
            f.hist_out[i] = cf2{v.x, v.y};
        }
        // the filter's buffer front: its last L - 1 input samples + the samples the block quantisation leaves pending
        const int64_t fi_move = a.move_src - a.fbuf;
#pragma unroll 1
        for (int64_t i = tid; i < a.move_n; i += T) {
            const v2f v = slow1(fi_move + i);
            a.move_dst[i] = cf2{v.x, v.y};
        }
    }

    const int64_t n_blocks = (a.n_emit + V - 1) / V;
    const int nsteps = W / kP0Step;
    // A wave's steps of a block: s = wave, wave + NWAVES, ... (N = 4096: twelve steps, three per wave).  ONE frame buffer: the unpack of
    // a step empties it and the next step's frames take it, in flight under the multiply-adds.  Measured alternatives (same box,
    // gpurun_out/r6/): all three steps' frames asked for at the block's start (36 registers more in the fill) 2.35 ms against 1.43;
    // the next block's frames asked for ahead of this block's transforms (36 registers held across them: 87 spilled) 2.14 ms.
    for (int64_t blk = blockIdx.x; blk < n_blocks; blk += (int64_t)gridDim.x) {
        const int64_t o0 = blk * V;
        for (int p = W + tid; p < N; p += T) X[sw(p)] = cf2{0.0f, 0.0f};           // the points behind the window's stream samples
        uint32_t rc[NW];
        bool have = false;                                           // rc holds the frames of the step about to run (wave-uniform)
#ifndef IQGPU_DIAG_P0FFT_NOFILL                                      // (DIAGNOSTIC build, timing only: the transforms alone, on whatever the buffer holds)
        for (int s = wave; s < nsteps; s += NWAVES) {
            // the step's 320 window entries <-> outputs k_first .. k_first + 319 of the call
            const int64_t k_first = o0 + (int64_t)kP0Step * s - f.pre;
            const bool fast = k_first >= f.k_a && k_first + kP0Step <= f.k_b;
            const int pb = kP0Step * s + NS * lane;
            if (fast) {
                const uint64_t P = f.phi0 + (uint64_t)(k_first + NS * lane) * (uint64_t)step;
                if (!have) fetch(P, rc);
                reload(P);                                           // (nothing to do where the step before has looked ahead)
                v2f Hw[14], own[9];
                Hw[0] = v2f{0.f, 0.f};
#pragma unroll
                for (int i = 1; i < 14; ++i) Hw[i] = p0_unpack<FMT>(rc, i - 1);
#pragma unroll
                for (int m = 0; m < 9; ++m) own[m] = p0_unpack<FMT>(rc, 13 + m);
                // the frames of this wave's next step of the block take the registers the unpack has emptied: in flight under the
                // multiply-adds
                const int sn = s + NWAVES;
                const int64_t kn = k_first + (int64_t)kP0Step * NWAVES;
                const bool next_fast = sn < nsteps && kn >= f.k_a && kn + kP0Step <= f.k_b;
                const uint64_t Pn = P + (uint64_t)(kP0Step * NWAVES) * (uint64_t)step;
                have = next_fast;
                if (next_fast) fetch(Pn, rc);
                v2f y[NS];
                compute5(Hw, own, y);
#pragma unroll
                for (int j = 0; j < NS; ++j) X[sw(pb + j)] = cf2{y[j].x, y[j].y};
                if (next_fast) reload(Pn);                           // the re-reads' round trip runs beside the stores
            } else {
#pragma unroll 1
                for (int j = 0; j < NS; ++j) {
                    const v2f v = slow1(o0 + pb + j);
                    X[sw(pb + j)] = cf2{v.x, v.y};
                }
                have = false;
            }
        }
#endif
        // the tap rows are let go of here: 80 registers that the transforms need (kept across them the kernel spilled; a wave re-reads
        // its five rows once per block -- 40 reads against the four LDS round trips of the two transforms)
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            held[j] = 0xffffffffu;
#pragma unroll
            for (int i = 0; i < 8; ++i) t[j][i] = v2f{0.f, 0.f};
        }
        __syncthreads();
        cf2 io[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) io[i] = X[sw(tid + i * T)];
#ifdef IQGPU_DIAG_P0FFT_NOFFT
        // DIAGNOSTIC build (timing only, wrong bytes): the window fill alone -- the block's points leave as they are
        for (int i = 0; i < 16; ++i) if (tid + i * T < V && o0 + tid + i * T < a.n_emit) pack_store(a.out, o0 + tid + i * T, a.out_fmt, io[i]);
#else
        fftconv16_tail<LOG2N, true>(a, X, s_nco, tid, io, o0, V, L1);
#endif
        __syncthreads();                                             // the transform buffer is free again
    }
}

template <int LOG2N, int FMT>
static hipError_t launch_p0fft_n(const FftConvArgs &a, size_t lds, hipStream_t s)
{
    int l2 = 0, l3 = 0, l4 = 0;
    if (!p0_class(a.feed.step, &l3, &l4, &l2)) return hipErrorInvalidValue;
    const int64_t n_blocks = (a.n_emit + a.vout - 1) / a.vout;
    unsigned grid = (unsigned)(n_blocks < (int64_t)a.feed.grid ? n_blocks : (int64_t)a.feed.grid);
    if (grid == 0) grid = 1;                                          // (nothing to emit: the state of the next call is still due)
    constexpr int T = (1 << LOG2N) / 16;
#define IQGPU_LAUNCH_PF(FMT, L3, L4, L2)                                                                            \
    do {                                                                                                              \
        static LdsAttrCache cache;                /* per instantiation */                                          \
        { const hipError_t e = cache.ensure((const void *)k_p0fft16<LOG2N, FMT, L3, L4, L2>, lds); if (e != hipSuccess) return e; } \
        hipLaunchKernelGGL((k_p0fft16<LOG2N, FMT, L3, L4, L2>), dim3(grid), dim3(T), lds, s, a);                    \
    } while (0)
#define IQGPU_LAUNCH_PF_CLS(FMT)                                                                                    \
    do {                                                                                                              \
        if (l2 == 2 && l3 == 3 && l4 == 4) IQGPU_LAUNCH_PF(FMT, 3, 4, 2);                                           \
        else if (l2 == 2 && l3 == 3) IQGPU_LAUNCH_PF(FMT, 3, 5, 2);                                                 \
        else if (l2 == 2) IQGPU_LAUNCH_PF(FMT, 4, 5, 2);                                                            \
        else if (l3 == 4) IQGPU_LAUNCH_PF(FMT, 4, 6, 3);                                                            \
        else if (l4 == 6) IQGPU_LAUNCH_PF(FMT, 5, 6, 3);                                                            \
        else IQGPU_LAUNCH_PF(FMT, 5, 7, 3);                                                                         \
    } while (0)
#ifdef IQGPU_P0FFT_QUICK
    // (development builds: the one instantiation of the cu8-nrsc5-usb / -lsb presets -- the full set takes minutes to compile)
    if (FMT == IQGPU_FMT_CU8 && l2 == 3 && l3 == 4 && l4 == 6) IQGPU_LAUNCH_PF(FMT, 4, 6, 3);
    else return hipErrorInvalidValue;
#else
    IQGPU_LAUNCH_PF_CLS(FMT);
#endif
#undef IQGPU_LAUNCH_PF_CLS
#undef IQGPU_LAUNCH_PF
    return hipGetLastError();
}


// the launch for one input format (operands checked by launch_p0fft, fftconv.hip)
template <int FMT>
static hipError_t launch_p0fft_fmt(const FftConvArgs &a, size_t lds, hipStream_t s)
{
    return launch_p0fft_n<12, FMT>(a, lds, s);
}

} // namespace iqgpu
