// front_tiles.hpp -- the tile routine of the wave-autonomous front kernels (run_tiles and its helpers), shared by
// front_wave.hip (k_front_s1) and front_fat.hip (k_front_fat, whose edge waves run the scalar-load instantiation of it).
// See front_wave.hip for the design notes.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/iqgpu.h"
#include "dsp_device.hpp"
#include "kernels.hpp"
#include "wave_common.hpp"

namespace iqgpu {

constexpr int kXRows = 5 + 64;                              // 20 history + 256 samples per parity stream
constexpr int kHRows = 4 + 64;                              // 16 history + 256 half-band outputs, aliased onto
constexpr int kHBOff = 5;                                   //   XO rows kHBOff .. kHBOff + kHRows - 1
constexpr int kWaveLds = (kXRows + kHBOff + kHRows) * kRowB; // 6816 B per wave
constexpr int kTabLds = 2 * 1024 * 8 + 256 * 14 * 4;        // NCO {cos,sin}, its half-scaled copy (FAST), polyphase taps [256][14]


// smallest q with q * d >= x, for 0 < x < 2^32, 2^24 <= d <= 2^25 (quotient below 2^8).  The float
// estimate of x / d is within 1e-4 of the truth, so its truncation is the exact floor or one off in
// either direction; make it the exact floor first, then round up.
__device__ __forceinline__ uint32_t ceil_div_small(uint32_t x, uint32_t d, float inv_d)
{
    uint32_t f = (uint32_t)((float)x * inv_d);
    f -= ((uint64_t)f * d > (uint64_t)x) ? 1u : 0u;                 // estimate one too high
    f += ((uint64_t)(f + 1) * d <= (uint64_t)x) ? 1u : 0u;          // estimate one too low
    return f + (((uint64_t)f * d < (uint64_t)x) ? 1u : 0u);
}

// (pack_cs16 / pack_b8: dsp_device.hpp)
__device__ __forceinline__ void pack_store_at(char *base, uint32_t idx, int fmt, cf2 v)
{
    if (fmt == IQGPU_FMT_CS16) {
        *(uint32_t *)(base + 4u * idx) = pack_cs16(v);
    } else if (fmt == IQGPU_FMT_CF32) {
        *(cf2 *)(base + 8u * idx) = v;
    } else {
        pack_store(base, (int64_t)idx, fmt, v);
    }
}

__device__ __forceinline__ int out_bytes(int fmt)
{
    return (fmt == IQGPU_FMT_CS8 || fmt == IQGPU_FMT_CU8) ? 2 : (fmt == IQGPU_FMT_CS24) ? 6
         : (fmt == IQGPU_FMT_CS32 || fmt == IQGPU_FMT_CU32 || fmt == IQGPU_FMT_CF32) ? 8 : 4;
}

// ---- optional in-kernel stamps (diagnostic build only: -DIQGPU_STAMPS; never in the shipped .so).
// Per-phase cycle sums of every wave are added into a.sink[32 KiB ...] as u64 counters.
#ifdef IQGPU_STAMPS
#define STAMP_DECL unsigned long long st_last = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP_BEGIN do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st_last = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
#define STAMP(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st_acc[i] += now_ - st_last; st_last = now_; } while (0)
#define STAMP_FLUSH(sinkbase) do { if (lane == 0) { unsigned long long *d_ = (unsigned long long *)((char *)(sinkbase) + 32768); for (int i_ = 0; i_ < 8; ++i_) atomicAdd(d_ + i_, st_acc[i_]); } } while (0)
#else
#define STAMP_DECL
#define STAMP_BEGIN do { } while (0)
#define STAMP(i) do { } while (0)
#define STAMP_FLUSH(sinkbase) do { } while (0)
#endif

// in-kernel shader clock of the streaming loop (diagnostic build only: -DIQGPU_CLOCKSTAMP): per wave,
// cycles (s_memtime) and 100 MHz ticks (s_memrealtime) of its whole run, summed into a.sink[32 KiB + 128 ...]
#ifdef IQGPU_CLOCKSTAMP
#define CLOCK_BEGIN const unsigned long long ck_c0 = __builtin_amdgcn_s_memtime(), ck_r0 = __builtin_amdgcn_s_memrealtime()
#define CLOCK_END(sinkbase) do { const unsigned long long c1_ = __builtin_amdgcn_s_memtime(), r1_ = __builtin_amdgcn_s_memrealtime(); \
        if (lane == 0) { unsigned long long *d_ = (unsigned long long *)((char *)(sinkbase) + 32768 + 128); \
            if (!EDGE) { atomicAdd(d_, c1_ - ck_c0); atomicAdd(d_ + 1, r1_ - ck_r0); atomicAdd(d_ + 2, 1ull); } \
            const unsigned gw_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                               \
            if (gw_ < 4096u) { unsigned *w_ = (unsigned *)(sinkbase);                                               \
                w_[gw_] = (unsigned)ck_r0; w_[4096 + gw_] = (unsigned)r1_;                                         \
                unsigned hw_, xcc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));             \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                \
                w_[(32768 + 256) / 4 + gw_] = (hw_ & 0xffffu) | (xcc_ << 16) | (EDGE ? 0x80000000u : 0u); } } } while (0)
#else
#define CLOCK_BEGIN do { } while (0)
#define CLOCK_END(sinkbase) do { } while (0)
#endif

struct WaveLds { char *XE, *XO, *HB; const cf2 *nco; const float *arb; unsigned arb_lds; };   // nco: two 8 KiB copies (FAST)

// first output at or after the lane's first half-band sample (4*lane), for a tile whose first
// output has phase delta0 (< step): n0 = its index within the tile, Pl = its phase relative to 4*lane
__device__ __forceinline__ void tap_phase(int lane, uint32_t delta0, uint32_t step, float inv_step, uint32_t &n0, uint32_t &Pl)
{
    const uint32_t tgt = (uint32_t)(4 * lane) << 24;
    n0 = 0;
    if (tgt > delta0) n0 = ceil_div_small(tgt - delta0, step, inv_step);
    Pl = (uint32_t)((uint64_t)delta0 + (uint64_t)n0 * step - (uint64_t)tgt);
}

// hit pattern and LDS row address of each of the lane's four slots
__device__ __forceinline__ void tap_rows(const WaveLds &w, uint32_t Pl, uint32_t step, bool hit[4], unsigned row[4])
{
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        hit[r] = Pl < ((uint32_t)(r + 1) << 24);       // Pl >= r << 24 here: the pending output never lies behind the slot
        const unsigned arm = (Pl >> 16) & 255u;
        row[r] = w.arb_lds + (arm ^ (arm >> 5)) * 56u;
        if (hit[r]) Pl += step;
    }
}

// the taps of TWO slots: 7 x ds_read_b64 from each arm's 56-byte row (asm, so that they are not fused into
// half-rate ds_read2_b64), ending with the wait -- no instruction of the compiler's can touch a tap
// register while its read is in flight (tools/check_isa.py).  The four slots are gathered as two such
// pairs with the first pair's 28 FMAs in between: 28 tap registers live instead of 56, which is what
// lets the kernel run 4 waves per SIMD (128 VGPRs) without scratch spills.
// Only lanes whose slot holds an output read (EXEC = the slot's hit mask for its seven reads): the rows
// are as good as random, so the reads are bank-conflict bound and every idle lane taken out of the
// access shortens it (for a step of 1.61 some 38 % of the slots are empty).  The registers of a masked
// lane keep stale values; its products are dropped by the same hit test further down.
__device__ __forceinline__ void gather_taps2(unsigned row_a, unsigned row_b, uint64_t hit_a, uint64_t hit_b, v2f ta[7], v2f tb[7])
{
    uint64_t ex;
    asm volatile(
        "s_mov_b64 %[ex], exec\n\t"
        "s_and_b64 exec, %[ex], %[ha]\n\t"
        "ds_read_b64 %0, %[ra]\n\tds_read_b64 %1, %[ra] offset:8\n\tds_read_b64 %2, %[ra] offset:16\n\t"
        "ds_read_b64 %3, %[ra] offset:24\n\tds_read_b64 %4, %[ra] offset:32\n\tds_read_b64 %5, %[ra] offset:40\n\t"
        "ds_read_b64 %6, %[ra] offset:48\n\t"
        "s_and_b64 exec, %[ex], %[hb]\n\t"
        "ds_read_b64 %7, %[rb]\n\tds_read_b64 %8, %[rb] offset:8\n\tds_read_b64 %9, %[rb] offset:16\n\t"
        "ds_read_b64 %10, %[rb] offset:24\n\tds_read_b64 %11, %[rb] offset:32\n\tds_read_b64 %12, %[rb] offset:40\n\t"
        "ds_read_b64 %13, %[rb] offset:48\n\t"
        "s_mov_b64 exec, %[ex]\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(ta[0]), "=&v"(ta[1]), "=&v"(ta[2]), "=&v"(ta[3]), "=&v"(ta[4]), "=&v"(ta[5]), "=&v"(ta[6]),
          "=&v"(tb[0]), "=&v"(tb[1]), "=&v"(tb[2]), "=&v"(tb[3]), "=&v"(tb[4]), "=&v"(tb[5]), "=&v"(tb[6]), [ex] "=&s"(ex)
        : [ra] "v"(row_a), [rb] "v"(row_b), [ha] "s"(hit_a), [hb] "s"(hit_b)
        : "memory", "scc");
}

// Tiles [t_begin, t_emit1) of kWTile frames; those from t_emit0 on produce output.
// EDGE = false: every tile (and the one after the last, for the prefetch) lies inside the call's
//               new, aligned frames and outside the history the call leaves behind.
// EDGE = true : per-frame scalar loads; handles history, end of call, alignment, history save.
// max over the wave of a non-negative double (rare: once per chunk boundary and run end)
__device__ __forceinline__ double wave_max_d(double m)
{
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) { const double o = __shfl_xor(m, k); m = o > m ? o : m; }
    return m;
}

// FEED (k_front_s2, front_s2.hip): the tile's samples do not come from memory but from an earlier stage of the SAME wave --
// feed->produce(t, x) leaves sample 256 c + 4 lane + s of tile t in x[c][s] (what unpack_chunk delivers for cf32 input)
struct NoFeed { __device__ void produce(int64_t, cf2 (&)[2][4]) {} };
template <int BPS, bool EDGE, bool FAST, bool S0, bool AGC = false, bool NONCO = false, typename FEED = NoFeed>
__device__ __forceinline__ void run_tiles(const FrontArgs &a, const WaveLds &w, const int lane,
                                          const int64_t t_begin, const int64_t t_emit0, const int64_t t_emit1, const int seg,
                                          FEED *const feed = nullptr)
{
    constexpr bool kFeed = !__is_same(FEED, NoFeed);
    static_assert(!kFeed || (!EDGE && !FAST && !S0 && BPS == 8), "a feeder stands in for the cf32 vector loads of a streaming run");
    // dc blocker (never in the FAST instantiation): wave-uniform state, carries per run as in k_cascade
    float dc_vr = 0.0f, dc_vi = 0.0f;
    DcLane lane_pow{1.0f, 1.0f, 1.0f};                     // c^(4 lane) and the scan's cross-row weights
    bool dc_started = false;
    if (!FAST && a.dc_enable) {
        lane_pow = dc_lane_init(a, lane);
    }
    constexpr int VB = BPS ? BPS : 4;
    constexpr int NC = S0 ? 1 : 2;                  // 256-frame chunks per tile: S0 = no half-band stage, a tile
    constexpr int TILE = 256 * NC;                  // is 256 input frames = 256 polyphase-input samples
    char *XE = w.XE, *XO = w.XO, *HB = w.HB;

    // output bookkeeping (wave-uniform): first output whose half-band sample is >= this run's first
    const uint32_t step = a.step;
    const float inv_step = 1.0f / (float)step;
    uint64_t k_tile0 = first_k_at((uint64_t)(t_emit0 * 256) << 24, a.phi0, step);
    uint32_t delta0 = (uint32_t)(a.phi0 + k_tile0 * (uint64_t)step - ((uint64_t)(t_emit0 * 256) << 24));   // < step
    // (the streaming FAST instantiations write cs16; their EDGE siblings may be asked for cf32 as well: k_front_mid in front of a user filter)
    const int obps = (FAST && !EDGE) ? 4 : out_bytes(a.out_fmt);
    const bool unit_gain = FAST || a.gain == 1.0f;
    // FAST: the same outputs-per-tile count and lane phases without a division in the loop
    const uint32_t n_est = (uint32_t)(((uint64_t)1 << 32) / step);
    const uint64_t c_est = (uint64_t)n_est * step;                       // <= 2^32
    uint32_t n0_st = 0, Pl_st = 0;
    bool taps_ready = false;
    const int woff = (5 + (lane >> 1)) * kRowB + (lane & 1) * 16;       // this lane's LDS write slot

    // register prefetch of the next tile's frames (compiler-managed loads: hipcc waits for them with
    // vmcnt(0) at the top of the next iteration, a whole tile after they were issued)
    RawChunk nxt[2];
    // ... and of its eight NCO phasors: the table lookups depend on the stream position only, so they are
    // issued a tile ahead too and their LDS round trip never sits on the tile's critical path
    // NONCO (FAST only): the same preset shape without a shift -- no mixer at all; the samples stay unnormalised in LDS
    // and the 2^-15 rides on the half-band taps (launch_front_s1 scales hb0; exact, a power of two)
    const bool nco_on = !EDGE && ((FAST && !NONCO) || (!FAST && a.nco_mode != 0));
    v2f cs_n[2][4];
    auto nco_lookup = [&](int64_t tile_first) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            uint32_t th = a.nco_theta0 + ((uint32_t)tile_first + (uint32_t)(256 * c + 4 * lane)) * a.nco_dtheta;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                // FAST: the odd stream only ever meets the half-band's centre tap 0.5 -- its samples are mixed with the
                // half-scaled copy of the table and stored as 0.5 x (exactly: a power of two), which saves the 8 multiplies
                cs_n[c][s] = nco_phasor2(w.nco, th, (FAST && (s & 1)) ? 1 : 0);
                th += a.nco_dtheta;
            }
        }
    };
    if (!EDGE && !kFeed) {
        const char *src = (const char *)a.raw + (t_begin * TILE - a.rem0) * VB + 4 * VB * lane;
        load_chunk<VB, IQGPU_NT_S1 != 0>(src, nxt[0]);
        if (NC == 2) load_chunk<VB, IQGPU_NT_S1 != 0>(src + 256 * VB, nxt[1]);
        if (nco_on) nco_lookup(t_begin * TILE);
    }

    // cs16 output of the streaming variant: a tile's four packed dwords are held in registers and
    // stored at the top of the NEXT iteration, right after the wait for the prefetched frames, so
    // that this wait (vmcnt(0)) only ever covers loads and stores issued a whole tile earlier
    // ... and the 2-byte formats (cu8 / cs8: the cu8-nrsc5 presets) likewise: one store per slot was a third of that shape's time
    const bool out_b8 = !FAST && (a.out_fmt == IQGPU_FMT_CU8 || a.out_fmt == IQGPU_FMT_CS8);
    const bool defer = !EDGE && (FAST || a.out_fmt == IQGPU_FMT_CS16 || out_b8);
    // ... and cf32 (the last stage in front of a user filter): the lane's 2 .. 4 frames as one 16-byte store + at most one more, a tile late --
    // stored where they were computed, the wait for the next tile's frames (vmcnt(0)) stood behind four fresh 8-byte stores every tile
    // (cf32 input only: the last stage behind k_cascade; the 8-bit-input kernels have no eight registers to spare at 16 waves)
    const bool defer_f = !EDGE && !FAST && BPS == 8 && a.out_fmt == IQGPU_FMT_CF32;
    // A lane's outputs of a tile are consecutive (2 to 4 of them: one per 1 .. 2 half-band samples), so
    // they are compacted and leave as one 8-byte store plus at most one more, instead of four predicated
    // dword stores -- the CU's vector-memory issue path is one of the three pipes this kernel loads.
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2), aligned(4)));
    uint32_t pend_c[4] = {0, 0, 0, 0}, pend_n0 = 0, pend_cnt = 0;
    cf2 pend_f[4] = {cf2{0.f, 0.f}, cf2{0.f, 0.f}, cf2{0.f, 0.f}, cf2{0.f, 0.f}};
    typedef float f32x4a8 __attribute__((ext_vector_type(4), aligned(8)));
    char *pend_base = (char *)a.out;
    typedef uint32_t u32a2 __attribute__((aligned(2)));
    auto flush_pending = [&]() {
        if (defer_f) {
            if (pend_cnt != 0) {
                char *b = pend_base + 8u * pend_n0;
                if (pend_cnt == 1) *(cf2 *)b = pend_f[0];
                else *(f32x4a8 *)b = f32x4a8{pend_f[0].x, pend_f[0].y, pend_f[1].x, pend_f[1].y};
                if (pend_cnt == 3) *(cf2 *)(b + 16) = pend_f[2];
                if (pend_cnt == 4) *(f32x4a8 *)(b + 16) = f32x4a8{pend_f[2].x, pend_f[2].y, pend_f[3].x, pend_f[3].y};
                pend_cnt = 0;
            }
            return;
        }
        if (out_b8) {
            // 2-byte frames: the lane's 2 .. 4 outputs as one or two dwords at a 2-byte-aligned address, an odd one as a short
            if (pend_cnt != 0) {
                char *b = pend_base + 2u * pend_n0;
                if (pend_cnt & 1u) *(uint16_t *)(b + 2u * (pend_cnt - 1u)) = (uint16_t)pend_c[pend_cnt == 1u ? 0 : 2];
                if (pend_cnt >= 2u) *(u32a2 *)b = pend_c[0] | (pend_c[1] << 16);
                if (pend_cnt == 4u) *(u32a2 *)(b + 4) = pend_c[2] | (pend_c[3] << 16);
                pend_cnt = 0;
            }
            return;
        }
        if (pend_cnt != 0) {
            char *b = pend_base + 4u * pend_n0;
            if (pend_cnt == 1) *(uint32_t *)b = pend_c[0];
            else *(u32x2 *)b = u32x2{pend_c[0], pend_c[1]};
            if (pend_cnt == 3) *(uint32_t *)(b + 8) = pend_c[2];
            if (pend_cnt == 4) *(u32x2 *)(b + 8) = u32x2{pend_c[2], pend_c[3]};
            pend_cnt = 0;
        }
    };
    float sl_hist = 0.0f;         // one dword per lane (< 48) of the last 4 rows of the polyphase input
    // fused AGC (locked phase): gain from the device state, per-lane max |y|^2 (exact, in double) of the chunk the
    // run is in and of the next one; a chunk ends where the input frame (c + 1) * chunk_frames - 1 completes a
    // half-band sample (agc_out_end, kernels.hpp): at most one boundary per tile
    float agc_g = 1.0f;
    double agc_m0 = 0.0, agc_m1 = 0.0;
    int64_t agc_c = 0, agc_B = 0;
    bool agc_any = false;
    // polyphase-input sample q of the call needs the chain's input frames up to ((q + 1) << AS) - agc_rem - 1; a tile holds 256 of them
    const int AS = AGC ? a.agc_shift : 0;
    if (AGC) {
        agc_g = a.agc_state->gain;
        const int64_t F0 = (((int64_t)256 * t_emit0 + 1) << AS) - 1 - a.agc_rem;
        agc_c = F0 > 0 ? F0 / a.agc_chunk_frames : 0;
        agc_B = (agc_c + 1) * a.agc_chunk_frames;
    }
    STAMP_DECL
    STAMP_BEGIN;
    CLOCK_BEGIN;
    for (int64_t t = t_begin; t < t_emit1; ++t) {
        const int64_t i0 = t * TILE;
        const int64_t j0 = i0 - a.rem0;
        const bool emit = t >= t_emit0;

        // ------------------------------------------------------------ pointwise -> LDS
        cf2 x[2][4];
        if (kFeed) {
            if (defer || defer_f) flush_pending();
            feed->produce(t, x);
        } else if (!EDGE) {
            // consume the prefetched frames first: the wait for them lands here, before the next
            // tile's loads are issued, so those stay in flight across the whole tile
            if (FAST) {
                // cs16 left unnormalised: the 2^-15 lives in this kernel's copy of the NCO table (exact)
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        x[c][s].x = (float)(short)(nxt[c].w[s] & 0xffffu);
                        x[c][s].y = (float)(short)(nxt[c].w[s] >> 16);
                    }
            } else {
                unpack_chunk<VB>(nxt[0], a.in_fmt, a.gain, unit_gain, x[0]);
                if (NC == 2) unpack_chunk<VB>(nxt[1], a.in_fmt, a.gain, unit_gain, x[1]);
            }
            STAMP(0);
            if (defer || defer_f) flush_pending();
            {
                const char *src = (const char *)a.raw + (j0 + TILE) * VB + 4 * VB * lane;
                load_chunk<VB, IQGPU_NT_S1 != 0>(src, nxt[0]);
                if (NC == 2) load_chunk<VB, IQGPU_NT_S1 != 0>(src + 256 * VB, nxt[1]);
            }
            if (!FAST && a.dc_enable) {
                if (!dc_started) { const cd2 cv = a.dc_carry[seg]; dc_vr = (float)cv.x; dc_vi = (float)cv.y; dc_started = true; }
                dc_chunk(a, lane, lane_pow, x[0], 0u, dc_vr, dc_vi);
                if (NC == 2) dc_chunk(a, lane, lane_pow, x[1], 0u, dc_vr, dc_vi);
            }
            if (!FAST && a.iq_enable) {
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float re = x[c][s].x;
                        x[c][s].x = re * a.iq_magp1;
                        x[c][s].y = fmaf(a.iq_phase, re, x[c][s].y);
                    }
            }
            if (nco_on) {
                // the eight phasors were looked up while the previous tile was in flight (below)
#pragma unroll
                for (int c = 0; c < NC; ++c)
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
            for (int c = 0; c < NC; ++c) {
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
                        if (FAST) { const short *pr = (const short *)a.raw + 2 * js; v = cf2{(float)pr[0], (float)pr[1]}; }
                        else v = unpack_one(a.raw, js, a.in_fmt, a.gain);
                        new_mask[c] |= 1u << s;
                    }
                    x[c][s] = v;
                }
            }
            if (!FAST && a.dc_enable && (dc_started || j0 + TILE > 0)) {
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
                if (NC == 2) dc_chunk(a, lane, lane_pow, x[1], hist_mask[1], dc_vr, dc_vi);
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int l4 = 256 * c + 4 * lane;
                uint32_t th = a.nco_theta0 + ((uint32_t)i0 + (uint32_t)l4) * a.nco_dtheta;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (new_mask[c] & (1u << s)) {
                        cf2 v = x[c][s];
                        if (!FAST && a.iq_enable) {
                            const float re = v.x;
                            v.x = re * a.iq_magp1;
                            v.y = fmaf(a.iq_phase, re, v.y);
                        }
                        if (FAST && NONCO) { v.x *= 1.0f / 32768.0f; v.y *= 1.0f / 32768.0f; }      // what hist_out keeps: normalised
                        else if (FAST || a.nco_mode != 0) v = cmul_tab(v, nco_phasor(w.nco, th));
                        const int64_t back = a.frames_in - (j0 + l4 + s);   // 1 .. hist_cap for kept frames
                        if (emit && back <= (int64_t)a.hist_cap) a.hist_out[(int64_t)a.hist_cap - back] = v;
                        x[c][s] = v;
                    }
                    th += a.nco_dtheta;
                }
            }
        }
        // wave priority: a wave that is about to feed the LDS pipe (the busiest of the three) goes ahead of waves
        // that are in their FMA runs (measured -3 % on the NRSC-5 chain)
        if (!EDGE) __builtin_amdgcn_s_setprio(1);
        v2f own[4];                                 // the lane's own polyphase-input row (row lane + 4 of HB): kept, not re-read
        if (S0) {
            // no half-band stage: the lane's four samples ARE its polyphase-input row
            if (lane < 48) *(float *)(HB + lane * 4) = sl_hist;         // history rows of the polyphase input
            char *ph = HB + (lane + 4) * kRowB;
            *(float4 *)ph = make_float4(x[0][0].x, x[0][0].y, x[0][1].x, x[0][1].y);
            *(float4 *)(ph + 16) = make_float4(x[0][2].x, x[0][2].y, x[0][3].x, x[0][3].y);
#pragma unroll
            for (int i = 0; i < 4; ++i) own[i] = v2f{x[0][i].x, x[0][i].y};
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int off = woff + 32 * c * kRowB;
                if (FAST && EDGE && NONCO) {  // history and new samples alike are normalised here: back to the LDS domain
#pragma unroll
                    for (int q = 0; q < 4; ++q) { x[c][q].x *= 32768.0f; x[c][q].y *= 32768.0f; }
                } else if (FAST && EDGE) {    // the scalar path mixes with the full table (and keeps its samples for hist_out)
                    x[c][1].x *= 0.5f; x[c][1].y *= 0.5f; x[c][3].x *= 0.5f; x[c][3].y *= 0.5f;
                }
                *(float4 *)(XE + off) = make_float4(x[c][0].x, x[c][0].y, x[c][2].x, x[c][2].y);
                *(float4 *)(XO + off) = make_float4(x[c][1].x, x[c][1].y, x[c][3].x, x[c][3].y);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // the rows that become the next tile's history are read back NOW (queued right behind the writes)
        // and stored at the end of the tile: by then the data is long there, so the slide costs no LDS
        // round trip of its own
        float sl_e = 0.f, sl_o = 0.f;
        if (!S0 && lane < 60) { sl_e = *(const float *)(XE + 64 * kRowB + lane * 4); sl_o = *(const float *)(XO + 64 * kRowB + lane * 4); }
        if (nco_on) nco_lookup(i0 + TILE);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(1);

        // ------------------------------------------------------------ half-band: 4 outputs per lane
        if (!S0) {
            const char *we = XE + lane * kRowB;
            v2f E[24];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const float4 v1 = ld4(we + r * kRowB + 16);
                if (r == 0) {                  // E[0] is not a tap of any of the lane's four outputs: 8 bytes do
                    // (through a pointer the compiler cannot see through: next to the 16-byte read at + 16 the vectoriser would
                    //  re-chunk the 24 bytes as 16 at + 8 -- a misaligned ds_read2_b64 -- and 8 at + 24)
                    typedef __attribute__((address_space(3))) const v2f lds_v2f_t;
                    unsigned we8 = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)(we + 8);
                    asm("" : "+v"(we8));
                    E[0] = v2f{0.f, 0.f}; E[1] = *(lds_v2f_t *)(size_t)we8;
                } else {
                    const float4 v0 = ld4(we + r * kRowB);
                    E[4 * r + 0] = v2f{v0.x, v0.y}; E[4 * r + 1] = v2f{v0.z, v0.w};
                }
                E[4 * r + 2] = v2f{v1.x, v1.y}; E[4 * r + 3] = v2f{v1.z, v1.w};
            }
            const char *wo = XO + lane * kRowB;
            const float4 o0 = ld4(wo + 2 * kRowB + 16), o1 = ld4(wo + 3 * kRowB);
            const float hc = FAST ? (NONCO ? 0.5f / 32768.0f : 1.0f) : 0.5f;   // FAST: the odd stream is stored as 0.5 x (NONCO: as 2^15 x)
            v2f acc[4] = {v2f{hc * o0.x, hc * o0.y}, v2f{hc * o0.z, hc * o0.w},
                          v2f{hc * o1.x, hc * o1.y}, v2f{hc * o1.z, hc * o1.w}};
            if (!EDGE) __builtin_amdgcn_s_setprio(0);
            const v2f *hbp = (const v2f *)a.hb0;          // 10 SGPR pairs {h[2i], h[2i+1]}
            // acc[r] += h[2 q2] E[20 + r - 2 q2] + h[2 q2 + 1] E[19 + r - 2 q2], q2 = 0 .. 9
            pk_fma_hb40(acc, hbp, E + 11);
            pk_fma_hb40(acc, hbp + 5, E + 1);
            // The half-band output rows live ON TOP of the odd-stream rows (HB row h = XO row h + 5): every read of
            // the odd stream for this tile has been issued above, so its data rows are dead.  The 4 history rows of
            // the half-band output (the previous tile's last rows, kept in sl_hist) are put back first: they share
            // XO rows 5 .. 8, which this tile's pointwise phase has just used.
            if (!EDGE) __builtin_amdgcn_s_setprio(1);
            if (lane < 48) *(float *)(HB + lane * 4) = sl_hist;
            char *ph = HB + (lane + 4) * kRowB;
            *(float4 *)ph = make_float4(acc[0].x, acc[0].y, acc[1].x, acc[1].y);
            *(float4 *)(ph + 16) = make_float4(acc[2].x, acc[2].y, acc[3].x, acc[3].y);
#pragma unroll
            for (int i = 0; i < 4; ++i) own[i] = acc[i];
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 48) sl_hist = *(const float *)(HB + 64 * kRowB + lane * 4);
        __builtin_amdgcn_sched_barrier(0);

        // ------------------------------------------------------------ polyphase + pack
        if (emit) {
            const int64_t q_tile0 = t * 256;
            if (!EDGE || q_tile0 < a.n_groups) {
                const char *wh = HB + lane * kRowB;
                v2f H[18];
                H[0] = v2f{0.f, 0.f}; H[1] = *(const v2f *)(wh + 24);          // H[0] is not a tap of any slot: 8 bytes do
#pragma unroll
                for (int r = 1; r < 4; ++r) {
                    const float4 v0 = ld4(wh + r * kRowB), v1 = ld4(wh + r * kRowB + 16);
                    H[4 * r - 2] = v2f{v0.x, v0.y}; H[4 * r - 1] = v2f{v0.z, v0.w};
                    H[4 * r + 0] = v2f{v1.x, v1.y}; H[4 * r + 1] = v2f{v1.z, v1.w};
                }
                H[14] = own[0]; H[15] = own[1]; H[16] = own[2]; H[17] = own[3];     // row lane + 4: what this lane wrote above
                // (gathered right before use: values written by asm loads must not sit in registers
                // that the register allocator may copy before the wait below)
                uint32_t n0, Pl;
                if (FAST && !EDGE) {
                    if (!taps_ready) { tap_phase(lane, delta0, step, inv_step, n0_st, Pl_st); taps_ready = true; }
                    n0 = n0_st; Pl = Pl_st;
                } else {
                    tap_phase(lane, delta0, step, inv_step, n0, Pl);
                }
                bool hit[4];
                unsigned row[4];
                tap_rows(w, Pl, step, hit, row);
                v2f y[4];                                    // started by pk_fma_pp16's first products
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    v2f ta[7], tb[7];
                    gather_taps2(row[2 * half], row[2 * half + 1], __builtin_amdgcn_ballot_w64(hit[2 * half]),
                                 __builtin_amdgcn_ballot_w64(hit[2 * half + 1]), ta, tb);
                    STAMP(3);
                    // y[a] += ta[n2].lo H[14 + 2 half - 2 n2] + ta[n2].hi H[13 + 2 half - 2 n2]; y[b] one sample later
                    pk_fma_pp16(y[2 * half], y[2 * half + 1], ta, tb, H + 7 + 2 * half);
                    pk_fma_pp12(y[2 * half], y[2 * half + 1], ta + 4, tb + 4, H + 1 + 2 * half);
                }
                STAMP(4);
                // half-band samples of this tile that exist in this call
                uint32_t q_lim = 256u;
                if (EDGE) { const int64_t left = a.n_groups - q_tile0; if (left < 256) q_lim = (uint32_t)left; }
                char *obase = (char *)a.out + (int64_t)k_tile0 * obps;
                const uint32_t pth0 = a.pnco_theta0 + (uint32_t)k_tile0 * a.pnco_dtheta;
                if (!EDGE) __builtin_amdgcn_s_setprio(0);
                uint32_t kk = n0;
                uint32_t pk[4] = {0, 0, 0, 0};
                cf2 pf[4] = {cf2{0.f, 0.f}, cf2{0.f, 0.f}, cf2{0.f, 0.f}, cf2{0.f, 0.f}};
                uint32_t agc_qb = 256u;                           // half-band samples of this tile below it are in chunk agc_c
                if (AGC) {
                    const int64_t F0 = (((int64_t)256 * t + 1) << AS) - 1 - a.agc_rem;   // last input frame that polyphase-input sample 0 of the tile needs
                    if (F0 >= agc_B) {                                      // the boundary fell between two tiles
                        const double m = wave_max_d(agc_m0);
                        if (lane == 0 && m > 0.0) atomicMax(a.agc_peak2 + agc_c, (unsigned long long)__double_as_longlong(m));
                        agc_m0 = 0.0; agc_c += 1; agc_B += a.agc_chunk_frames;
                    }
                    const int64_t d = agc_B - F0;
                    if (d < ((int64_t)256 << AS)) agc_qb = (uint32_t)((d + ((int64_t)1 << AS) - 1) >> AS);
                    agc_any = true;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (FAST && !EDGE && !AGC) {
                        // every lane packs every slot (a slot without an output holds junk that the compaction below
                        // never picks): no EXEC games, no zero-initialised words
                        pk[r] = pack_cs16(cf2{y[r].x, y[r].y});
                    } else if (hit[r] && (!EDGE || (uint32_t)(4 * lane + r) < q_lim)) {
                        v2f yy = y[r];
                        if (!FAST && a.pnco_mode != 0) yy = pk_cmul(yy, nco_phasor2(w.nco, pth0 + kk * a.pnco_dtheta));
                        if (AGC) {
                            // agc_apply: peak of the chunk over the samples BEFORE the gain, then samples[i] *= g (src/agc.c:169-214)
                            const double re = (double)yy.x, im = (double)yy.y;
                            const double m2 = fma(re, re, im * im);          // exact: products of floats, sum below 2^53 ulps
                            if (agc_qb >= 256u) agc_m0 = fmax(agc_m0, m2);
                            else if ((uint32_t)(4 * lane + r) < agc_qb) agc_m0 = fmax(agc_m0, m2);
                            else agc_m1 = fmax(agc_m1, m2);
                            yy = v2f{yy.x * agc_g, yy.y * agc_g};
                        }
                        if (defer) pk[r] = out_b8 ? pack_b8(cf2{yy.x, yy.y}, a.out_fmt == IQGPU_FMT_CU8) : pack_cs16(cf2{yy.x, yy.y});
                        else if (defer_f) pf[r] = cf2{yy.x, yy.y};
                        else pack_store_at(obase, kk, (FAST && !EDGE) ? (int)IQGPU_FMT_CS16 : a.out_fmt, cf2{yy.x, yy.y});
                    }
                    kk += hit[r] ? 1u : 0u;
                }
                if (defer_f) {
                    const bool h01 = hit[0] && hit[1];
                    pend_f[0] = hit[0] ? pf[0] : pf[1];
                    pend_f[1] = h01 ? pf[1] : (hit[2] ? pf[2] : pf[3]);
                    pend_f[2] = (h01 && hit[2]) ? pf[2] : pf[3];
                    pend_f[3] = pf[3];
                    pend_n0 = n0;
                    pend_cnt = kk - n0;
                }
                if (defer) {
                    // compact the hit slots (gaps between hits are 1 or 2 samples, so a streaming lane has >= 2)
                    const bool h01 = hit[0] && hit[1];
                    pend_c[0] = hit[0] ? pk[0] : pk[1];
                    pend_c[1] = h01 ? pk[1] : (hit[2] ? pk[2] : pk[3]);
                    pend_c[2] = (h01 && hit[2]) ? pk[2] : pk[3];
                    pend_c[3] = pk[3];
                    pend_n0 = n0;
                    pend_cnt = kk - n0;
                }
                pend_base = obase;
                if (AGC && agc_qb < 256u) {                       // the tile held a boundary: chunk agc_c is complete for this run
                    const double m = wave_max_d(agc_m0);
                    if (lane == 0 && m > 0.0) atomicMax(a.agc_peak2 + agc_c, (unsigned long long)__double_as_longlong(m));
                    agc_m0 = agc_m1; agc_m1 = 0.0; agc_c += 1; agc_B += a.agc_chunk_frames;
                }
            }
            // outputs of this tile, ceil((2^32 - delta0) / step) = floor((2^32 - 1 - delta0) / step) + 1
            uint32_t nt;
            if (FAST) {
                // outputs k with delta0 + k step < 2^32: n_est of them, one more iff the (n_est)-th still fits
                nt = n_est + (((uint64_t)delta0 + c_est) < ((uint64_t)1 << 32) ? 1u : 0u);
            } else {
                const uint32_t xm = 0xffffffffu - delta0;
                uint32_t nfl = (uint32_t)((float)xm * inv_step);
                nfl -= ((uint64_t)nfl * step > (uint64_t)xm) ? 1u : 0u;
                nfl += ((uint64_t)(nfl + 1) * step <= (uint64_t)xm) ? 1u : 0u;
                nt = nfl + 1u;
            }
            k_tile0 += nt;
            // e = shift of every phase from this tile to the next, |e| < step
            const int32_t e = (int32_t)((int64_t)((uint64_t)nt * step) - ((int64_t)1 << 32));
            delta0 = (uint32_t)((int32_t)delta0 + e);
            if (FAST && !EDGE && taps_ready) {
                // the lane's first output keeps its index unless its phase leaves [0, step)
                int32_t pl = (int32_t)Pl_st + e;
                if (pl < 0) { pl += (int32_t)step; n0_st += 1u; }
                else if (pl >= (int32_t)step) { pl -= (int32_t)step; n0_st -= 1u; }
                Pl_st = (uint32_t)pl;
            }
        }

        STAMP(5);
        // ------------------------------------------------------------ slide the windows
        {
            // the last 5 rows of XE / XO become the history rows of the next tile.  One dword per
            // lane: a ds_write_b32 costs 4 LDS cycles whatever the lane count, a ds_write_b128 13.
            if (!S0 && lane < 60) { *(float *)(XE + lane * 4) = sl_e; *(float *)(XO + lane * 4) = sl_o; }
        }
        __builtin_amdgcn_wave_barrier();
        STAMP(6);
    }
    if (defer || defer_f) flush_pending();
    if (AGC && agc_any) {
        const double m = wave_max_d(agc_m0);
        if (lane == 0 && m > 0.0) atomicMax(a.agc_peak2 + agc_c, (unsigned long long)__double_as_longlong(m));
    }
    CLOCK_END(a.sink);
    STAMP_FLUSH(a.sink);
}

} // namespace iqgpu
