// front_fat.hip -- k_front_fat: the NRSC-5 preset shape (cs16 in, unit gain, pre NCO or none, one half-band m = 10,
// 256-arm polyphase with 1.6 <= step / 2^24 < 2, cs16 out) as FEWER, FATTER waves.
//
// k_front_s1 (front_wave.hip) runs this chain with 16 waves per CU, 4 half-band outputs per lane and five LDS round
// trips per 512-frame tile; both the vector pipe and the LDS pipe sit at ~65 % and the in-order waves wait on each
// other.  This kernel trades occupancy for registers:
//
//   * 8 waves per CU (2 per SIMD, up to 256 VGPRs), 1024-frame tiles, a lane owns 8 consecutive half-band outputs:
//     every even sample is read 27 / 8 = 3.4 times instead of 23 / 4 = 5.75, the polyphase window 21 / 8 instead of
//     17 / 4, rows of 8 cf32 at an 80-byte pitch (an odd number of 16-byte slots: conflict-free ds_read_b128);
//   * the tile loop is software-pipelined so that every batch of LDS reads is issued one phase before it is used:
//         P1  pointwise(T)  -> write X(T)   -> issue the half-band window reads of tile T
//         P2  polyphase(T-1) + pack + store          (its window and taps were issued in P3 of the iteration before)
//         P3  half-band(T)  -> write HB(T)  -> issue the polyphase window + tap reads of tile T, NCO phasors of T + 1
//     a wave waits at most once per phase, and what it waits for has had a whole phase of its own FMAs to arrive;
//   * the polyphase stage runs FIVE slots per 8 half-band samples instead of eight (a lane's 8 samples hold 4 or 5
//     outputs when step / 2^24 >= 1.6): slot j's output sits at sample lo_j + d, d in {0, 1, 2}, lo_j = floor(j step)
//     a compile-time constant of the step class.  The FMAs of a slot run over a FIXED 16-sample register window
//     [lo_j - 13, lo_j + 2]; where the output really is becomes a shift of the taps: the arm's taps are stored
//     reversed between two zeros on either side (two copies one float apart, so that every shift is an 8-byte
//     aligned start), and a zero tap leaves the sum as it is -- same products in the same order, same bits;
//   * all lanes gather (no EXEC masks, no empty slots): 40 ds_read_b64 per 1024 frames instead of 56 half-empty ones,
//     79 polyphase FMAs instead of 112, 5 packs instead of 8, and a lane's 4 or 5 outputs leave as one 16-byte
//     store plus at most one dword.
//
// Everything is plain C++ on float2 values: hipcc folds the tap broadcasts into op_sel and -- because every LDS
// access is one it can see -- places exact counted lgkmcnt waits itself.  The file is built with the load / store
// merging of the back end switched off (build.py): merged, the 8-byte tap reads become half-rate ds_read2_b64.
//
// Arithmetic, summation order and stream bookkeeping are those of k_front_s1<4, fast>: the outputs are bit-identical
// (tests/test_gpu_parity.py::test_fat_kernel_equals_the_sixteen_wave_kernel).  Edge tiles (stream history, end of the
// call, unaligned buffers) are run by the scalar-load instantiation of run_tiles (front_tiles.hpp) on a few extra
// waves, two 512-frame tiles per 1024-frame tile.
#include "front_tiles.hpp"
#include "front_fat_common.hpp"

namespace iqgpu {

constexpr int kFatWaves = 8;
constexpr int kFatThreads = kFatWaves * 64;
constexpr int kFRowB = 80;                                  // LDS row: 8 cf32 + 16 B pad (5 slots of 16 bytes)
constexpr int kFXRows = 3 + 64;                             // 24 history + 512 samples per parity stream
constexpr int kFatWaveLds = 2 * kFXRows * kFRowB;           // XE, XO; the half-band output rows (2 + 64) live on top of XE
constexpr int kFatNcoLds = 2 * 1024 * 8;
constexpr int kFatArbLds = 256 * 14 * 4;                    // the edge waves' table (layout of k_front_s1)
constexpr int kFatTabLds = kFatNcoLds + kFatArbLds + kFTapLds;
static_assert(kFatWaveLds >= kWaveLds, "an edge wave uses the slice with k_front_s1's layout");
static_assert(kFatTabLds + kFatWaves * kFatWaveLds <= 160 * 1024, "LDS");

int front_fat_waves() { return kFatWaves; }
size_t front_fat_lds_bytes() { return (size_t)kFatTabLds + (size_t)kFatWaves * kFatWaveLds; }

struct FatLds { char *XE, *XO; const cf2 *nco; unsigned tap_lds; };

// Tiles [T_begin, T_emit1) of 1024 frames; those from T_emit0 on produce output.  Every tile, and the one behind the
// last (prefetch), lies inside the call's new, 16-byte aligned frames and outside the history the call leaves behind.
// L3, L4: lo_3 = floor(3 step / 2^24), lo_4 = floor(4 step / 2^24) of the step class (lo_0 .. lo_2 = 0, 1, 3 for all of them).
template <bool NONCO, int L3, int L4>
__device__ __forceinline__ void run_fat(const FrontArgs &a, const FatLds &w, const int lane,
                                        const int64_t T_begin, const int64_t T_emit0, const int64_t T_emit1)
{
    constexpr int LO[5] = {0, 1, 3, L3, L4};
    char *XE = w.XE, *XO = w.XO, *HB = w.XE;
    const uint32_t step = a.step;
    float hb[20];
#pragma unroll
    for (int k = 0; k < 20; ++k) hb[k] = a.hb0[k];

    // ---- output bookkeeping of the polyphase tile T_emit0 (wave-uniform), then per lane
    uint64_t k_tile0 = first_k_at(((uint64_t)T_emit0 * 512) << 24, a.phi0, step);
    uint32_t delta0 = (uint32_t)(a.phi0 + k_tile0 * (uint64_t)step - (((uint64_t)T_emit0 * 512) << 24));   // < step
    const uint32_t n_est = (uint32_t)(((uint64_t)1 << 33) / step);       // outputs of a 512-sample tile: n_est or n_est + 1
    const uint64_t c_est = (uint64_t)n_est * step;
    uint32_t n0, Pl;                                   // the lane's first output of the tile: index in the tile, phase from sample 8 lane
    {
        const uint64_t tgt = (uint64_t)(8 * lane) << 24;
        const uint64_t nn = tgt > delta0 ? (tgt - delta0 + step - 1) / step : 0;
        n0 = (uint32_t)nn;
        Pl = (uint32_t)((uint64_t)delta0 + nn * step - tgt);
    }

    // ---- per-lane LDS offsets
    const int wq = (3 + (lane >> 2)) * kFRowB + (lane & 3) * 16;                // this lane's write slot of chunk 0 (chunk c: 16 c rows on)
    const int sl_src = (64 + (lane >> 4)) * kFRowB + (lane & 15) * 4;          // one dword per lane of the last rows ...
    const int sl_dst = (lane >> 4) * kFRowB + (lane & 15) * 4;                 // ... becomes the history rows of the next tile
    const char *we = XE + lane * kFRowB, *wo = XO + lane * kFRowB, *wh = HB + lane * kFRowB;

    RawChunk nxt[4];
    v2f cs_n[4][4];
    auto load_tile = [&](int64_t T) {
        const char *src = (const char *)a.raw + (T * 1024 - a.rem0) * 4 + 16 * lane;
#pragma unroll
        for (int c = 0; c < 4; ++c) load_chunk<4, IQGPU_NT_FAT != 0>(src + 1024 * c, nxt[c]);
    };
    auto nco_lookup = [&](int64_t T) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t th = a.nco_theta0 + ((uint32_t)(T * 1024) + (uint32_t)(256 * c + 4 * lane)) * a.nco_dtheta;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                cs_n[c][s] = nco_phasor2(w.nco, th, s & 1);     // the odd stream only meets the centre tap 0.5: half-scaled copy
                th += a.nco_dtheta;
            }
        }
    };
    load_tile(T_begin);
    if (!NONCO) nco_lookup(T_begin);

    float sl_e = 0.f, sl_o = 0.f, sl_h = 0.f;
    v2f own[8];                                        // the lane's own half-band outputs of the tile before (row lane + 2 of HB)
    v2f Hw[14];                                        // the 13 half-band samples in front of them (+ one unused)
    v2f tp[2][8];                                      // taps of slots 0, 1 of the next polyphase tile
    unsigned trow[5];                                  // LDS address of each slot's (shifted) tap row
#pragma unroll
    for (int i = 0; i < 8; ++i) own[i] = v2f{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 14; ++i) Hw[i] = v2f{0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) tp[j][i] = v2f{0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 5; ++j) trow[j] = w.tap_lds;
    typedef __attribute__((address_space(3))) const v2f lds_v2f;

    v2f E[28], acc[8];
    v2f x[4][4];
    v2f t2[8], t3[8], t4[8];                           // taps of slots 2 .. 4
    v2f y[5];
#define FENCE() __builtin_amdgcn_sched_barrier(0)
    auto taps = [&](v2f (&t)[8], unsigned row) {
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = *(lds_v2f *)(size_t)(row + tap_pair_off(i));
    };
    // The tile loop as a fixed sequence of small batches, LDS batches (at most ~16 operations: a wave can have 15 in flight)
    // alternating with the FMA runs that cover them; every batch of reads is issued at least one FMA run before its first use.
    // ---- polyphase window of the tile before: Hw[i] = half-band sample at row coordinate 8 lane + 2 + i (m = i - 14)
    auto L_slide_h = [&]() { if (lane < 32) sl_h = *(const float *)(HB + sl_src); };     // (every tile: the warm-up ones too)
    auto L_hw = [&]() {
#pragma unroll
        for (int q = 1; q < 8; ++q) {
            const float4 v = ldq(wh + (q >> 2) * kFRowB + 16 * (q & 3));
            Hw[2 * q - 2] = v2f{v.x, v.y}; Hw[2 * q - 1] = v2f{v.z, v.w};
        }
        keep(Hw[0]);
    };
    // ---- pointwise: unpack, mix
    auto V_point = [&](const int64_t T) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                x[c][s] = v2f{(float)(short)(nxt[c].w[s] & 0xffffu), (float)(short)(nxt[c].w[s] >> 16)};   // 2^-15: in the table (taps when NONCO)
        load_tile(T + 1);                              // (tile T_emit1 is readable too: the plan keeps one tile behind every run)
        if (!NONCO) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int s = 0; s < 4; ++s) x[c][s] = pk_cmul(x[c][s], cs_n[c][s]);
        }
    };
    // ---- the mixed samples -> XE / XO (on top of the half-band rows: L_hw has read them)
    auto L_xwrite = [&]() {
        if (lane < 48) { *(float *)(XE + sl_dst) = sl_e; *(float *)(XO + sl_dst) = sl_o; }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            stq(XE + wq + 16 * c * kFRowB, make_float4(x[c][0].x, x[c][0].y, x[c][2].x, x[c][2].y));
            stq(XO + wq + 16 * c * kFRowB, make_float4(x[c][1].x, x[c][1].y, x[c][3].x, x[c][3].y));
        }
        __builtin_amdgcn_wave_barrier();
    };
    // ---- half-band window: E[j] = even sample at row coordinate 8 lane + 4 + j; output i uses E[20 + i - k], k = 0 .. 19
    auto L_erows = [&](const int r0, const int r1) {
#pragma unroll
        for (int r = r0; r < r1; ++r)
#pragma unroll
            for (int q = (r == 0 ? 2 : 0); q < 4; ++q) {
                const float4 v = ldq(we + r * kFRowB + 16 * q);
                const int j = 8 * r + 2 * q - 4;
                E[j] = v2f{v.x, v.y}; E[j + 1] = v2f{v.z, v.w};
            }
        if (r0 == 0) keep(E[0]);
    };
    auto L_slide_x = [&]() { if (lane < 48) { sl_e = *(const float *)(XE + sl_src); sl_o = *(const float *)(XO + sl_src); } };
    // ---- centre tap: odd sample at row coordinate 8 lane + 14 + i
    auto L_odd = [&]() {
        const float4 o0 = ldq(wo + 1 * kFRowB + 48), o1 = ldq(wo + 2 * kFRowB), o2 = ldq(wo + 2 * kFRowB + 16), o3 = ldq(wo + 2 * kFRowB + 32);
        acc[0] = v2f{o0.x, o0.y}; acc[1] = v2f{o0.z, o0.w}; acc[2] = v2f{o1.x, o1.y}; acc[3] = v2f{o1.z, o1.w};
        acc[4] = v2f{o2.x, o2.y}; acc[5] = v2f{o2.z, o2.w}; acc[6] = v2f{o3.x, o3.y}; acc[7] = v2f{o3.z, o3.w};
        if (NONCO) {                                   // with a mixer the odd stream is stored as 0.5 x
            const float hc = 0.5f / 32768.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = v2f{hc * acc[i].x, hc * acc[i].y};
        }
    };
    // ---- pack + store of the polyphase tile, on to the next one, its tap rows
    auto V_emit = [&]() {
        uint32_t pk[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) pk[j] = pack_cs16(cf2{y[j].x, y[j].y});
        // the lane's 4 or 5 outputs are consecutive: one 16-byte store and at most one dword
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
        char *ob = (char *)a.out + ((int64_t)k_tile0 + n0) * 4;
        *(u32x4 *)ob = u32x4{pk[0], pk[1], pk[2], pk[3]};
        if (Pl + 4u * step < (8u << 24)) *(uint32_t *)(ob + 16) = pk[4];
        const uint32_t nt = n_est + (((uint64_t)delta0 + c_est) < ((uint64_t)1 << 33) ? 1u : 0u);
        k_tile0 += nt;
        const int32_t e = (int32_t)((int64_t)((uint64_t)nt * step) - ((int64_t)1 << 33));      // |e| < step
        delta0 = (uint32_t)((int32_t)delta0 + e);
        int32_t pl = (int32_t)Pl + e;
        if (pl < 0) { pl += (int32_t)step; n0 += 1u; }
        else if (pl >= (int32_t)step) { pl -= (int32_t)step; n0 -= 1u; }
        Pl = (uint32_t)pl;
    };
    // tap rows of the polyphase tile the state stands at: slot j's output has phase Pl + j step from the lane's first sample:
    // position p = phase >> 24, arm = the next 8 bits, shift d = p - lo_j in {0, 1, 2}:
    // d = 2 -> planes 0 .. 7, d = 0 -> planes 1 .. 8, d = 1 -> the planes that start one float on (9 .. 16)
    auto V_taprows = [&]() {
        uint32_t P = Pl;
        if (a.tap_fold) {       // (wave-uniform: a scalar branch)
#pragma unroll
            for (int j = 0; j < 5; ++j) { trow[j] = tap_row<true>(w.tap_lds, P, LO[j]); P += step; }
        } else {
#pragma unroll
            for (int j = 0; j < 5; ++j) { trow[j] = tap_row<false>(w.tap_lds, P, LO[j]); P += step; }
        }
    };
    auto V_hb = [&](const int q0, const int q1) {
#pragma unroll
        for (int q2 = q0; q2 < q1; ++q2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fma2(hb[2 * q2], E[20 + i - 2 * q2], acc[i]);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fma2(hb[2 * q2 + 1], E[19 + i - 2 * q2], acc[i]);
        }
    };
    // ---- half-band rows (on top of XE: its reads for this tile are long issued); the lane keeps its own row
    auto L_hbwrite = [&]() {
        if (lane < 32) *(float *)(HB + sl_dst) = sl_h;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            stq(HB + (2 + lane) * kFRowB + 16 * q, make_float4(acc[2 * q].x, acc[2 * q].y, acc[2 * q + 1].x, acc[2 * q + 1].y));
        __builtin_amdgcn_wave_barrier();
    };

    // front half of a tile: pointwise -> X -> half-band -> HB rows; with PP = true the polyphase of the tile before runs between them
    auto tile = [&](const int64_t T, const bool PP) {
        L_slide_h(); if (PP) L_hw();
        FENCE();
        V_point(T); FENCE();
        L_xwrite(); if (PP) { taps(t3, trow[3]); taps(t4, trow[4]); } FENCE();
        if (PP) { pp_slots2<8, 0, 1>(Hw, own, tp[0], tp[1], y[0], y[1]); FENCE(); }
        L_slide_x(); L_erows(0, 2); FENCE();
        if (PP) { pp_slots3<8, 3, L3, L4>(Hw, own, t2, t3, t4, y[2], y[3], y[4]); FENCE(); }
        L_erows(2, 4); L_odd(); FENCE();
        if (PP) { keep(y[4]); V_emit(); }       // (slot 4 is computed by every lane beside the others, not as a chain of its own under the store's branch)
        V_taprows(); FENCE();
        if (!NONCO) { nco_lookup(T + 1); FENCE(); }
        V_hb(0, 3); FENCE();
        taps(tp[0], trow[0]); FENCE();
        V_hb(3, 6); FENCE();
        taps(tp[1], trow[1]); FENCE();
        V_hb(6, 10); FENCE();
        taps(t2, trow[2]);
        L_hbwrite();
#pragma unroll
        for (int i = 0; i < 8; ++i) own[i] = acc[i];
        FENCE();
    };
    for (int64_t T = T_begin; T <= T_emit0; ++T) tile(T, false);          // warm-up tiles and the first emitting one: no polyphase yet
    for (int64_t T = T_emit0 + 1; T < T_emit1; ++T) tile(T, true);        // steady state
    // the last tile's polyphase
    L_hw(); taps(t3, trow[3]); taps(t4, trow[4]); FENCE();        // (no slide: nothing follows)
    pp_slots2<8, 0, 1>(Hw, own, tp[0], tp[1], y[0], y[1]);
    pp_slots3<8, 3, L3, L4>(Hw, own, t2, t3, t4, y[2], y[3], y[4]);
    keep(y[4]);
    V_emit();
#undef FENCE
}

// NONCO: the same shape without a shift (no mixer; the 2^-15 rides on the half-band taps, launch_front_fat scales hb0)
template <bool NONCO, int L3, int L4>
__global__ __launch_bounds__(kFatThreads) void k_front_fat(const FrontArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    cf2   *s_nco = (cf2 *)smem, *s_nco_half = s_nco + 1024;
    float *s_arb = (float *)(smem + kFatNcoLds);
    float *s_tap = (float *)(smem + kFatNcoLds + kFatArbLds);
    char *slice = (char *)smem + kFatTabLds + wave * kFatWaveLds;
    if (((unsigned)(size_t)(__attribute__((address_space(3))) const void *)s_nco & 8191u) != 0u) __builtin_trap();   // nco_phasor2 ORs the index into the base

    if (!NONCO) {
        const float sgn = a.nco_mode < 0 ? -1.0f : 1.0f;               // mix down: conj(phasor)
        const float scl = 1.0f / 32768.0f;                             // the cs16 normaliser, folded into the table (exact)
        for (int i = tid; i < 1024; i += kFatThreads) {
            const cf2 v = a.nco_tab[i];
            s_nco[i] = cf2{v.x * scl, sgn * v.y * scl};
            s_nco_half[i] = cf2{v.x * (0.5f * scl), sgn * v.y * (0.5f * scl)};
        }
    }
    for (int i = tid; i < 256 * 14; i += kFatThreads) {                 // the edge waves' rows: arm a in row a ^ (a >> 5) of 56 B
        const int arm = i / 14, k = i % 14;
        s_arb[(arm ^ (arm >> 5)) * 14 + k] = a.arb_table[arm * 16 + k];
    }
    fill_tap_planes(s_tap, a.arb_table, tid, kFatThreads, a.tap_fold != 0);
    for (int i = lane; i < kFatWaveLds / 16; i += 64) ((float4 *)slice)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    const int64_t gw = (int64_t)blockIdx.x * kFatWaves + wave;
    if (gw == 0 && a.frames_in < (int64_t)a.hist_cap) {
        const int keep_n = a.hist_cap - (int)a.frames_in;
        for (int i = lane; i < keep_n; i += 64) a.hist_out[i] = a.hist_in[i + (int)a.frames_in];
    }
    if (gw < a.w_n_edge) {
        // edge work in 1024-frame tiles [0, w_edge_ta) and [w_edge_tb, w_total_tiles): two tiles of run_tiles each
        int64_t t0, t1;
        if (gw < a.w_n_edge1) { t0 = gw * a.w_edge_tpw; t1 = t0 + a.w_edge_tpw; if (t1 > a.w_edge_ta) t1 = a.w_edge_ta; }
        else { t0 = a.w_edge_tb + (gw - a.w_n_edge1) * a.w_edge_tpw; t1 = t0 + a.w_edge_tpw; if (t1 > a.w_total_tiles) t1 = a.w_total_tiles; }
        WaveLds w;
        w.XE = slice; w.XO = w.XE + kXRows * kRowB; w.HB = w.XO + kHBOff * kRowB;
        w.nco = s_nco; w.arb = s_arb;
        w.arb_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)s_arb;
        run_tiles<4, true, true, false, false, NONCO>(a, w, lane, 2 * t0 - 1, 2 * t0, 2 * t1, 0);
    } else {
        const int64_t r = gw - a.w_n_edge;
        if (r >= a.w_n_stream) return;
        const int64_t t0 = w_run_start(a, r), t1 = w_run_start(a, r + 1);
        FatLds w;
        w.XE = slice; w.XO = slice + kFXRows * kFRowB; w.nco = s_nco;
        w.tap_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)s_tap;
        run_fat<NONCO, L3, L4>(a, w, lane, t0 - a.w_warm_tiles, t0, t1);
    }
}

// step class of the five-slot polyphase: 1.6 <= s < 2 (a lane's 8 samples hold 4 or 5 outputs), lo_3 = floor(3 s), lo_4 = floor(4 s)
static int fat_step_class(uint32_t step)
{
    const uint64_t one = (uint64_t)1 << 24;
    if ((uint64_t)step * 5 < 8 * one || (uint64_t)step >= 2 * one) return 0;      // s < 1.6 or s >= 2
    const int l3 = (int)(((uint64_t)step * 3) >> 24), l4 = (int)(((uint64_t)step * 4) >> 24);
    if (l3 == 4 && l4 == 6) return 1;
    if (l3 == 5 && l4 == 6) return 2;
    if (l3 == 5 && l4 == 7) return 3;
    return 0;
}

// the shape k_front_fat exists for: the specialised (FAST) shape of k_front_s1 without the fused AGC, in a step class above
bool front_fat_shape(const FrontArgs &a)
{
    return a.S == 1 && a.in_fmt == IQGPU_FMT_CS16 && a.out_fmt == IQGPU_FMT_CS16 && a.gain == 1.0f && !a.iq_enable && !a.dc_enable &&
           a.pnco_mode == 0 && !a.agc_fused && !(a.dbg & (kDbgNoFast | kDbgNoFat)) && fat_step_class(a.step) != 0;
}

hipError_t launch_front_fat(const FrontArgs &a_in, hipStream_t s)
{
    const bool nonco = a_in.nco_mode == 0;
    FrontArgs a = a_in;
    if (nonco) for (float &h : a.hb0) h *= 1.0f / 32768.0f;           // the cs16 normaliser rides on the half-band taps (exact: a power of two)
    const size_t lds = front_fat_lds_bytes();
    const int64_t n_items = a.w_n_edge + a.w_n_stream;
    const unsigned grid = (unsigned)((n_items + kFatWaves - 1) / kFatWaves);
    if (grid == 0) return hipSuccess;
#define IQGPU_LAUNCH_FAT(NONCO, L3, L4)                                                                              \
    do {                                                                                                              \
        static LdsAttrCache cache;                /* per instantiation */                                          \
        { const hipError_t e = cache.ensure((const void *)k_front_fat<NONCO, L3, L4>, lds); if (e != hipSuccess) return e; } \
        hipLaunchKernelGGL((k_front_fat<NONCO, L3, L4>), dim3(grid), dim3(kFatThreads), lds, s, a);                 \
    } while (0)
    switch (fat_step_class(a.step) * 2 + (nonco ? 1 : 0)) {
    case 2: IQGPU_LAUNCH_FAT(false, 4, 6); break;
    case 3: IQGPU_LAUNCH_FAT(true, 4, 6); break;
    case 4: IQGPU_LAUNCH_FAT(false, 5, 6); break;
    case 5: IQGPU_LAUNCH_FAT(true, 5, 6); break;
    case 6: IQGPU_LAUNCH_FAT(false, 5, 7); break;
    case 7: IQGPU_LAUNCH_FAT(true, 5, 7); break;
    default: return hipErrorInvalidValue;
    }
#undef IQGPU_LAUNCH_FAT
    return hipGetLastError();
}

} // namespace iqgpu
