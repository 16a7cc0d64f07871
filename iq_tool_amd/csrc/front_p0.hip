// front_p0.hip -- k_front_p0: chains WITHOUT a half-band stage (0.5 <= r < 1: the shipped cu8-nrsc5 presets, 2.4 MS/s ->
// 1.488375 MS/s) as an OUTPUT-major polyphase kernel whose taps stay in registers (round 5).
//
// k_front_s1<.., S0> (front_wave.hip) runs this shape sample-major: 256-frame tiles through LDS, one polyphase slot per input
// sample (38 % of them empty), 14 lane-random tap reads per slot -- 0.58 ms per 2^28 cu8 frames, 0.18 of the HBM rate, with the LDS
// pipe (the tap gather) as its limit.  Without a half-band stage nothing ties the polyphase stage to tiles of INPUT samples, so:
//
//   * a step of a wave is 320 consecutive OUTPUTS, five per lane: lane l of step s owns outputs K_a + 320 s + 5 l + (0 .. 4).
//     Their positions P = phi + k step are a closed form; the lane's window is the 22 input samples [p0 - 13, p0 + 8] around
//     them (p0 = position of its first output), which it LOADS ITSELF from the raw stream -- 44 bytes of cu8 at a 2-byte
//     aligned address, three dwordx4 -- unpacks, and keeps: no sample ever enters LDS, there is no history to slide and no
//     warm-up tile (a step depends on nothing but the stream position).  Neighbouring lanes' windows overlap (13 of 22
//     samples): the L1 serves the overlap, HBM sees every frame once;
//   * slot j's output sits LO_j + d samples behind p0, LO_j = floor(j step / 2^24) a compile-time constant of the step class, d in
//     {0, 1}: the same shifted, zero-padded tap rows and fixed 16-sample register windows as k_front_fat / k_front_mid
//     (front_fat_common.hpp), hence the same products in the same order and the same bits as k_front_s1;
//   * from one step to the next every lane-slot moves on by exactly 320 outputs.  Where 320 step / 2^24 is close to a whole
//     number of samples -- NRSC-5: 1.6125 x 320 = 516 - 0.001 -- its arm moves by a fraction (0.26 arms), so the slot's 16 taps
//     are the ones it already holds four steps out of five: the slot's (position, arm) key is compared with the one its registers
//     were loaded for and the tap rows are re-read under an EXEC mask only where it changed (a ds_read_b64 with a dozen active
//     lanes costs about a cycle of the LDS pipe, tools/lds_mask_bench.hip, against 5.8 for the full gather).  For a step without
//     such a period every slot reloads every time: the full gather, and still no sample traffic in LDS.
//
// Measured (round 5, profiles/r05_presets.md; round 6: 0.34 ms and 0.40 ms): cu8-nrsc5 front end 0.561 -> 0.441 ms per 2^28 cu8 frames, 0.66 -> 0.61 ms with cf32
// output in front of the -usb / -lsb filter (where the 1.3 GB of cf32 it writes set the pace).  What it took beyond the design:
// the tap re-reads of step s + 1 are issued at the END of step s (their LDS round trip runs beside the stores and the next unpack:
// 0.585 -> 0.465 ms); the frames of step s + 2 are fetched BEHIND the stores of step s (vmcnt counts in order, and hipcc waits
// for a store wherever its registers are written again); the folded arm placement (consecutive lanes are 16 arms apart for
// this step: linear planes put a slot's re-reading lanes into two bank pairs); the chunk sorting of the fused AGC only in the
// one step in thirty that holds a boundary.  368 VALU instructions and 179 LDS cycles per step: at two waves per SIMD (round 5) the
// kernel took their sum, at three (round 6) it overlaps them.  IQGPU_NO_P0=1 keeps k_front_s1<S0>.
//
// Three waves per SIMD since round 6 (80 VGPRs of taps + 44 of window + ONE step of frames in flight; two waves with two steps until then).  Edge tiles -- the stream
// history in front of the call, the tail that becomes the next call's history -- are run by the scalar-load instantiation of
// run_tiles (front_tiles.hpp, 256-frame tiles) on a few extra waves, as in the other wave-autonomous kernels; the streaming part is
// the outputs whose position lies in tiles [w_edge_ta, w_edge_tb).  Fused digital AGC as in k_front_mid (float peaks per chunk).
#include <type_traits>

#include "front_tiles.hpp"
#include "front_fat_common.hpp"
#include "front_p0_common.hpp"

namespace iqgpu {

// Round 6: TWELVE waves per CU (three per SIMD; 8 until then) with ONE frame buffer (two until then): every instantiation fits 156
// VGPRs without scratch, and the third wave per SIMD is worth more than the second step of frames in flight -- cu8-nrsc5 front end
// 0.4087 -> 0.3444 ms (-15.7 %), with cf32 output 0.477 -> 0.401; twelve waves with two buffers 0.363 / 0.407 (same box, three
// rounds of 40 steps, tools/gpu/r6_p0_waves.sh).  Bytes unchanged.
#ifndef IQGPU_P0_WAVES
#define IQGPU_P0_WAVES 12
#endif
constexpr int kP0Waves = IQGPU_P0_WAVES;
constexpr int kP0Threads = kP0Waves * 64;
constexpr int kP0EdgeMax = 6;                               // edge waves of a launch (k_front_s1's slice layout, an arena of their own)
constexpr int kP0EdgeTpw = 4;                               // 256-frame tiles per edge run
constexpr int kP0ArbLds = 256 * 14 * 4;                     // the edge waves' table (layout of k_front_s1)

constexpr int kP0StripB = kP0Step * 8;                      // cf32 output: a step's 320 outputs pass through a strip of the wave's own (coalesced stores)
static_assert(kP0ArbLds + kFTapLds + kP0EdgeMax * kWaveLds + kP0Waves * kP0StripB <= 160 * 1024, "LDS");

int front_p0_waves() { return kP0Waves; }
int front_p0_max_edge_waves() { return kP0EdgeMax; }
int front_p0_edge_tpw() { return kP0EdgeTpw; }
static size_t p0_lds_bytes() { return (size_t)kP0ArbLds + kFTapLds + (size_t)kP0EdgeMax * kWaveLds + (size_t)kP0Waves * kP0StripB; }

// Steps [s_begin, s_end) of the launch's streaming outputs [k_a, k_b) (call-relative output indices).
// FMT: cu8 / cs8 (2 bytes per frame) or cs16; OUTF: the output format (cu8 / cs8: 2 bytes per frame, cs16: 4, cf32: 8)
// L2, L3, L4 = floor(2 s), floor(3 s), floor(4 s) of the step class (s = step / 2^24; floor(s) = 1)
template <int FMT, int L3, int L4, int OUTF, bool AGC, int L2 = 3>
__device__ __forceinline__ void run_p0(const FrontArgs &a, const unsigned tap_lds, const unsigned strip_lds, const int lane, const int64_t s_begin, const int64_t s_end)
{
    constexpr int BPS = (FMT == IQGPU_FMT_CS16) ? 4 : 2;
    constexpr int OUTB = OUTF == IQGPU_FMT_CF32 ? 8 : OUTF == IQGPU_FMT_CS16 ? 4 : 2;
    constexpr int NW = (BPS == 2) ? 12 : 24;                 // raw words per window: 24 frames loaded, 22 used
    constexpr int NS = 5;
    constexpr int LO[5] = {0, 1, L2, L3, L4};
    typedef __attribute__((address_space(3))) const v2f lds_v2f;
    typedef uint32_t u4v __attribute__((ext_vector_type(4), aligned(2)));
    const uint32_t step = a.step;
    const uint64_t adv = (uint64_t)kP0Step * step;           // what a lane's phase moves on by from step to step

    // phase of the lane's first output of step s_begin: phi + k step, in samples << 24 from the call's first frame
    uint64_t P = a.phi0 + (uint64_t)(a.p0_k_a + s_begin * kP0Step + 5 * lane) * (uint64_t)step;
    // the frames of a step are fetched NB steps ahead, behind the stores of the step that frees their buffer (round 5: NB = 2, two
    // waves per SIMD cover little latency by themselves; round 6: NB = 1 and three waves; and vmcnt counts in order: a wait for a store -- hipcc places one wherever a register that a
    // store reads is written again -- must not stand behind younger loads, or every step waits for the frames it has just asked for)
#ifndef IQGPU_P0_NB
#define IQGPU_P0_NB 1
#endif
    constexpr int NB = IQGPU_P0_NB;                          // (round 6: ONE buffer and a third wave per SIMD instead of two and two)
    uint32_t rb[NB][NW];
    auto fetch = [&](uint64_t Pq, uint32_t (&r)[NW]) {
        int64_t f0 = (int64_t)(Pq >> 24) - 13;
        // (lanes of the last step that own no output any more would read past what the plan guarantees: pulled back)
        if (f0 > a.p0_f_max) f0 = a.p0_f_max;
        const char *src = (const char *)a.raw + f0 * BPS;
#pragma unroll
        for (int q = 0; q < NW / 4; ++q) {
            const u4v v = *(const u4v *)(src + 16 * q);
            r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
        }
    };
#pragma unroll
    for (int b = 0; b < NB; ++b) fetch(P + (uint64_t)b * adv, rb[b]);

    v2f t[NS][8];                                            // the slots' shifted tap rows, kept from step to step
    uint32_t held[NS];                                       // ... and the (position, arm) each was loaded for
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        held[j] = 0xffffffffu;
#pragma unroll
        for (int i = 0; i < 8; ++i) t[j][i] = v2f{0.f, 0.f};
    }

    // fused AGC of the locked phase (as k_front_mid): position p of the stream belongs to chunk p / chunk_frames (S = 0, no open
    // group); a step spans about 520 samples, a chunk at least 1024: one boundary per step at most
    float agc_g = 1.0f, m0 = 0.0f, m1 = 0.0f;
    int64_t agc_c = 0, agc_B = 0;
    // (phase of the step's first output -- lane 0, slot 0 -- from wave-uniform arguments only: the scalar unit keeps it)
    uint64_t Pw = a.phi0 + (uint64_t)(a.p0_k_a + s_begin * kP0Step) * (uint64_t)step;
    if (AGC) {
        agc_g = a.agc_state->gain;
        agc_c = (int64_t)(Pw >> 24) / a.agc_chunk_frames;
        agc_B = (agc_c + 1) * a.agc_chunk_frames;
    }
    auto flush_peak = [&](float m, int64_t c) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0 && m > 0.0f) atomicMax(a.agc_peak2 + c, (unsigned long long)__double_as_longlong((double)m));
    };

    // which tap rows the step at phase Pq needs anew: re-read those under their lanes' mask.  Issued at the END of the step before
    // (its multiply-adds have let go of the registers), so that the reads' round trip runs beside that step's stores and the next
    // one's unpack instead of in front of its first multiply-add
    auto reload = [&](const uint64_t Pq) {
        const uint32_t Fq = (uint32_t)Pq & 0xffffffu;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const uint32_t pj = Fq + (uint32_t)j * step;
            const uint32_t key = pj >> 16;
            if (key != held[j]) {
                const unsigned row = a.tap_fold ? tap_row<true>(tap_lds, pj, LO[j]) : tap_row<false>(tap_lds, pj, LO[j]);
#pragma unroll
                for (int i = 0; i < 8; ++i) t[j][i] = *(lds_v2f *)(size_t)(row + tap_pair_off(i));
                held[j] = key;
            }
        }
    };
    reload(P);

    // PARTIAL: the launch's last step, whose outputs from p0_k_b on do not exist (its own copy of the code: the loop has no such test)
    auto one_step = [&](auto partial, const int64_t s, uint32_t (&rc)[NW]) {
        constexpr bool PARTIAL = decltype(partial)::value;
        const uint32_t F = (uint32_t)P & 0xffffffu;           // phase of the lane's first output inside its sample
        const int64_t p0 = (int64_t)(P >> 24);
        uint32_t Pj[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) Pj[j] = F + (uint32_t)j * step;
        // ---- the window: Hw[i] = sample p0 - 14 + i (i = 1 .. 13), own[m] = sample p0 + m (m = 0 .. 8)
        v2f Hw[14], own[9];
        Hw[0] = v2f{0.f, 0.f};
#pragma unroll
        for (int i = 1; i < 14; ++i) Hw[i] = p0_unpack<FMT>(rc, i - 1);
#pragma unroll
        for (int m = 0; m < 9; ++m) own[m] = p0_unpack<FMT>(rc, 13 + m);
        // ---- five outputs: the chains of the other kernels, slot by slot
        v2f y[NS];
        pp_slots3<9, 0, 1, L2, true>(Hw, own, t[0], t[1], t[2], y[0], y[1], y[2]);
        pp_slots2<9, L3, L4, true>(Hw, own, t[3], t[4], y[3], y[4]);
        const int64_t k0 = a.p0_k_a + s * kP0Step + 5 * lane;                  // the lane's first output (call-relative)
        if (AGC) {
            const int64_t first = (int64_t)(Pw >> 24);     // position of the step's first output: the others lie at most 520 samples behind it
            if (first >= agc_B) {                           // the boundary fell between two steps
                flush_peak(m0, agc_c);
                m0 = 0.0f; agc_c += 1; agc_B += a.agc_chunk_frames;
            }
            const uint32_t b_rel = (uint32_t)(agc_B - first < 4096 ? agc_B - first : 4096);   // boundary, in samples behind `first`
            if (b_rel > 1024u) {
                // (no chunk boundary inside this step -- all but one step in thirty: nothing to sort)
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    const float m2 = fmaf(y[j].x, y[j].x, y[j].y * y[j].y);
                    if (!PARTIAL || k0 + j < a.p0_k_b) m0 = fmaxf(m0, m2);
                    y[j] = v2f{y[j].x * agc_g, y[j].y * agc_g};
                }
            } else {
                const uint32_t mine = (uint32_t)p0 - (uint32_t)first;
                bool crossed = false;
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    const float m2 = fmaf(y[j].x, y[j].x, y[j].y * y[j].y);
                    const bool valid = !PARTIAL || k0 + j < a.p0_k_b;
                    const bool late = mine + (Pj[j] >> 24) >= b_rel;
                    if (valid) { if (late) m1 = fmaxf(m1, m2); else m0 = fmaxf(m0, m2); }
                    crossed = crossed || (valid && late);
                    y[j] = v2f{y[j].x * agc_g, y[j].y * agc_g};
                }
                if (__builtin_amdgcn_ballot_w64(crossed) != 0ull) {   // the step held a boundary: chunk agc_c is complete for this run
                    flush_peak(m0, agc_c);
                    m0 = m1; m1 = 0.0f; agc_c += 1; agc_B += a.agc_chunk_frames;
                }
            }
        }
        // ---- pack + store: the lane's five outputs are consecutive
        char *ob = (char *)a.out + k0 * OUTB;
        const bool whole = !PARTIAL;
        if (OUTB == 8) {
            typedef float f2v __attribute__((ext_vector_type(2), aligned(8)));
            if (whole) {
                // Round 6: a lane's five outputs are 40 contiguous bytes, so the three stores above each touched 64 pieces 40 bytes
                // apart -- twenty-odd cache lines per instruction, every line of the step written by three instructions -- and the
                // kernel with cf32 output ran 0.59 - 0.65 ms for work its instruction and LDS counters price at 0.21 (profiles/
                // r06_fused_pmc.txt).  The step's 320 outputs pass through a 2560-byte strip of the wave's own in LDS and leave as five
                // stores of 64 consecutive outputs each: 512 contiguous bytes per instruction, every cache line written once.
                // (non-temporal stores on the old pattern: measured in round 5, no change.  The same strip for the 2- and 4-byte outputs:
                //  measured, +2.5 % -- 0.4235 against 0.4132 ms for the cu8-nrsc5 preset: those forms are bound by their instructions, not their stores)
                typedef __attribute__((address_space(3))) v2f lds_wv2f;
                const unsigned wa = strip_lds + 40u * (unsigned)lane;
#pragma unroll
                for (int j = 0; j < NS; ++j) *(lds_wv2f *)(size_t)(wa + 8u * j) = y[j];
                __builtin_amdgcn_wave_barrier();
                const unsigned ra = strip_lds + 8u * (unsigned)lane;
                char *ow = (char *)a.out + (a.p0_k_a + s * kP0Step) * OUTB + 8 * lane;       // (the step's first output: wave-uniform base)
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    const v2f v = *(lds_v2f *)(size_t)(ra + 512u * j);
                    *(f2v *)(ow + 512 * j) = f2v{v.x, v.y};
                }
                __builtin_amdgcn_wave_barrier();             // (the strip is the next step's too)
            } else {
#pragma unroll
                for (int j = 0; j < NS; ++j) if (k0 + j < a.p0_k_b) *(f2v *)(ob + 8 * j) = f2v{y[j].x, y[j].y};
            }
        } else if (OUTB == 4) {
            typedef uint32_t w4v __attribute__((ext_vector_type(4), aligned(4)));
            uint32_t pk[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) pk[j] = pack_cs16(cf2{y[j].x, y[j].y});
            if (whole) { *(w4v *)ob = w4v{pk[0], pk[1], pk[2], pk[3]}; *(uint32_t *)(ob + 16) = pk[4]; }
            else {
#pragma unroll
                for (int j = 0; j < NS; ++j) if (k0 + j < a.p0_k_b) *(uint32_t *)(ob + 4 * j) = pk[j];
            }
        } else {
            typedef uint32_t u32a2 __attribute__((aligned(2)));
            uint32_t pk[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) pk[j] = pack_b8(cf2{y[j].x, y[j].y}, OUTF == IQGPU_FMT_CU8);
            if (whole) {
                *(u32a2 *)ob = pk[0] | (pk[1] << 16);
                *(u32a2 *)(ob + 4) = pk[2] | (pk[3] << 16);
                *(uint16_t *)(ob + 8) = (uint16_t)pk[4];
            } else {
#pragma unroll
                for (int j = 0; j < NS; ++j) if (k0 + j < a.p0_k_b) *(uint16_t *)(ob + 2 * j) = (uint16_t)pk[j];
            }
        }
        // the frames of step s + NB take the buffer this step has emptied (the buffers take turns, NB steps per trip of the loop:
        // moving a queue up by register copies would wait for the load it copies)
        fetch(P + (uint64_t)NB * adv, rc);
        P += adv; Pw += adv;
        reload(P);
    };
    // the launch's very last step may be partial: it is taken out of the loop (s_last = the run's end when the run holds it)
    const bool has_partial = s_end * kP0Step > a.p0_k_b - a.p0_k_a;
    const int64_t s_full = has_partial ? s_end - 1 : s_end;
    int64_t s = s_begin;
    for (; s + NB <= s_full; s += NB) {
#pragma unroll
        for (int b = 0; b < NB; ++b) one_step(std::false_type{}, s + b, rb[b]);
    }
    int b = 0;
    if constexpr (NB == 2) {
        // (written out for the shipped two buffers: the general form below costs the instantiations with the fused AGC up to 60 spilled
        //  registers -- a second copy of the partial step)
        for (; s < s_full; ++s, ++b) one_step(std::false_type{}, s, rb[0]);        // at most one whole step is left over, in the first buffer
        if (has_partial) { if (b == 0) one_step(std::true_type{}, s, rb[0]); else one_step(std::true_type{}, s, rb[1]); }
    } else {
        // at most NB - 1 whole steps are left over, in the first buffers in turn
#pragma unroll
        for (int q = 0; q < NB - 1; ++q) if (s < s_full) { one_step(std::false_type{}, s, rb[q]); ++s; b = q + 1; }
        if (has_partial) {
#pragma unroll
            for (int q = 0; q < NB; ++q) if (b == q) one_step(std::true_type{}, s, rb[q]);
        }
    }
    if (AGC) flush_peak(m0, agc_c);
}

template <int FMT, int L3, int L4, int OUTF, bool AGC, int L2 = 3>
__global__ __launch_bounds__(kP0Threads) void k_front_p0(const FrontArgs a)
{
    constexpr int BPS = (FMT == IQGPU_FMT_CS16) ? 4 : 2;
    if (a.run_if && *a.run_if == 0) return;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *s_arb = (float *)smem;
    float *s_tap = (float *)(smem + kP0ArbLds);
    char *arena = (char *)smem + kP0ArbLds + kFTapLds;
    for (int i = tid; i < 256 * 14; i += kP0Threads) {                  // the edge waves' rows: arm a in row a ^ (a >> 5) of 56 B
        const int arm = i / 14, k = i % 14;
        s_arb[(arm ^ (arm >> 5)) * 14 + k] = a.arb_table[arm * 16 + k];
    }
    fill_tap_planes(s_tap, a.arb_table, tid, kP0Threads, a.tap_fold != 0);
    for (int i = tid; i < kP0EdgeMax * kWaveLds / 16; i += kP0Threads) ((float4 *)arena)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    const int64_t gw = (int64_t)blockIdx.x * kP0Waves + wave;
    if (gw == 0 && a.frames_in < (int64_t)a.hist_cap) {
        const int keep_n = a.hist_cap - (int)a.frames_in;
        for (int i = lane; i < keep_n; i += 64) a.hist_out[i] = a.hist_in[i + (int)a.frames_in];
    }
    if (gw < a.w_n_edge) {
        int64_t t0, t1;
        if (gw < a.w_n_edge1) { t0 = gw * a.w_edge_tpw; t1 = t0 + a.w_edge_tpw; if (t1 > a.w_edge_ta) t1 = a.w_edge_ta; }
        else { t0 = a.w_edge_tb + (gw - a.w_n_edge1) * a.w_edge_tpw; t1 = t0 + a.w_edge_tpw; if (t1 > a.w_total_tiles) t1 = a.w_total_tiles; }
        if (gw >= kP0EdgeMax) __builtin_trap();          // (the host keeps launches with more edge runs on k_front_s1)
        WaveLds w;
        w.XE = arena + (int)gw * kWaveLds; w.XO = w.XE + kXRows * kRowB; w.HB = w.XO + kHBOff * kRowB;
        w.nco = nullptr; w.arb = s_arb;
        w.arb_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)s_arb;
        run_tiles<BPS, true, false, true, AGC, false>(a, w, lane, t0 - a.w_warm_tiles, t0, t1, 0);
    } else {
        const int64_t r = gw - a.w_n_edge;
        if (r >= a.w_n_stream) return;
        const int64_t s0 = r * a.w_run_q + (r < a.w_run_r ? r : a.w_run_r), s1 = s0 + a.w_run_q + (r < a.w_run_r ? 1 : 0);
        const unsigned tap_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)s_tap;
        const unsigned strip_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)(arena + kP0EdgeMax * kWaveLds + wave * kP0StripB);
        run_p0<FMT, L3, L4, OUTF, AGC, L2>(a, tap_lds, strip_lds, lane, s0, s1);
    }
}

// which chains: no half-band stage, nothing pointwise but the unpack (unit gain, no dc blocker / iq correction / mixer on either
// side), cu8 / cs8 / cs16 in, cu8 / cs8 / cs16 / cf32 out, a five-slot step class (1.016 <= s < 2); the fused AGC with chunks of 1024 frames and more
bool front_p0_shape(const FrontArgs &a)
{
    int l3, l4;
    if (a.S != 0 || a.gain != 1.0f || a.iq_enable || a.dc_enable || a.nco_mode != 0 || a.pnco_mode != 0 || (a.dbg & (kDbgNoFast | kDbgNoFat))) return false;
    if (!(a.in_fmt == IQGPU_FMT_CU8 || a.in_fmt == IQGPU_FMT_CS8 || a.in_fmt == IQGPU_FMT_CS16)) return false;
    if (!(a.out_fmt == IQGPU_FMT_CU8 || a.out_fmt == IQGPU_FMT_CS8 || a.out_fmt == IQGPU_FMT_CS16 || a.out_fmt == IQGPU_FMT_CF32)) return false;
    if (a.agc_fused && (a.out_fmt == IQGPU_FMT_CF32 || a.agc_chunk_frames < 1024 || a.agc_shift != 0)) return false;
    return p0_class(a.step, &l3, &l4);
}

// the streaming part of a plan made by plan_front_s1 for 256-frame tiles: outputs whose position lies in tiles [ta, tb), as steps
// of 320 dealt out evenly over the wave slots the edge runs leave free
void plan_front_p0(FrontArgs &a, int64_t wave_slots)
{
    a.p0_k_a = a.p0_k_b = 0; a.p0_f_max = 0;
    const int64_t ta = a.w_edge_ta, tb = a.w_edge_tb;
    a.w_n_stream = 0; a.w_run_q = 0; a.w_run_r = 0;
    if (tb <= ta) return;
    auto first_k = [&](int64_t pos) {
        const uint64_t target = (uint64_t)pos << 24;
        return (int64_t)(target > a.phi0 ? (target - a.phi0 + (uint64_t)a.step - 1) / (uint64_t)a.step : 0);
    };
    a.p0_k_a = first_k(ta * 256); a.p0_k_b = first_k(tb * 256);
    a.p0_f_max = a.frames_in - 24;                                   // the last frame a 24-frame window load may start at
    const int64_t n_steps = (a.p0_k_b - a.p0_k_a + kP0Step - 1) / kP0Step;
    if (n_steps <= 0) return;
    int64_t w = wave_slots - a.w_n_edge;
    if (w > n_steps) w = n_steps;
    if (w < 1) w = 1;
    a.w_n_stream = w; a.w_run_q = n_steps / w; a.w_run_r = n_steps % w;
}

hipError_t launch_front_p0(const FrontArgs &a, hipStream_t s)
{
    int l2 = 0, l3 = 0, l4 = 0;
    if (!front_p0_shape(a) || !p0_class(a.step, &l3, &l4, &l2) || a.w_n_edge > kP0EdgeMax || a.rem0 != 0) return hipErrorInvalidValue;
    const size_t lds = p0_lds_bytes();
    const int64_t n_items = a.w_n_edge + a.w_n_stream;
    const unsigned grid = (unsigned)((n_items + kP0Waves - 1) / kP0Waves);
    if (grid == 0) return hipSuccess;
#define IQGPU_LAUNCH_P0(FMT, L3, L4, OUTF, AGC, L2)                                                                 \
    do {                                                                                                              \
        static LdsAttrCache cache;                /* per instantiation */                                          \
        { const hipError_t e = cache.ensure((const void *)k_front_p0<FMT, L3, L4, OUTF, AGC, L2>, lds); if (e != hipSuccess) return e; } \
        hipLaunchKernelGGL((k_front_p0<FMT, L3, L4, OUTF, AGC, L2>), dim3(grid), dim3(kP0Threads), lds, s, a);      \
    } while (0)
#define IQGPU_LAUNCH_P0_AGC(FMT, L3, L4, OUTF, L2)                                                                  \
    do { if (a.agc_fused) IQGPU_LAUNCH_P0(FMT, L3, L4, OUTF, true, L2); else IQGPU_LAUNCH_P0(FMT, L3, L4, OUTF, false, L2); } while (0)
#define IQGPU_LAUNCH_P0_OUT(FMT, L3, L4, L2)                                                                        \
    do {                                                                                                              \
        if (a.out_fmt == IQGPU_FMT_CF32) IQGPU_LAUNCH_P0(FMT, L3, L4, IQGPU_FMT_CF32, false, L2);                   \
        else if (a.out_fmt == IQGPU_FMT_CS16) IQGPU_LAUNCH_P0_AGC(FMT, L3, L4, IQGPU_FMT_CS16, L2);                 \
        else if (a.out_fmt == IQGPU_FMT_CU8) IQGPU_LAUNCH_P0_AGC(FMT, L3, L4, IQGPU_FMT_CU8, L2);                   \
        else IQGPU_LAUNCH_P0_AGC(FMT, L3, L4, IQGPU_FMT_CS8, L2);                                                   \
    } while (0)
#define IQGPU_LAUNCH_P0_CLS(FMT)                                                                                    \
    do {                                                                                                              \
        if (l2 == 2 && l3 == 3 && l4 == 4) IQGPU_LAUNCH_P0_OUT(FMT, 3, 4, 2);                                       \
        else if (l2 == 2 && l3 == 3) IQGPU_LAUNCH_P0_OUT(FMT, 3, 5, 2);                                             \
        else if (l2 == 2) IQGPU_LAUNCH_P0_OUT(FMT, 4, 5, 2);                                                        \
        else if (l3 == 4) IQGPU_LAUNCH_P0_OUT(FMT, 4, 6, 3);                                                        \
        else if (l4 == 6) IQGPU_LAUNCH_P0_OUT(FMT, 5, 6, 3);                                                        \
        else IQGPU_LAUNCH_P0_OUT(FMT, 5, 7, 3);                                                                     \
    } while (0)
    if (a.in_fmt == IQGPU_FMT_CU8) IQGPU_LAUNCH_P0_CLS(IQGPU_FMT_CU8);
    else if (a.in_fmt == IQGPU_FMT_CS8) IQGPU_LAUNCH_P0_CLS(IQGPU_FMT_CS8);
    else IQGPU_LAUNCH_P0_CLS(IQGPU_FMT_CS16);
#undef IQGPU_LAUNCH_P0_CLS
#undef IQGPU_LAUNCH_P0_OUT
#undef IQGPU_LAUNCH_P0_AGC
#undef IQGPU_LAUNCH_P0
    return hipGetLastError();
}

} // namespace iqgpu
