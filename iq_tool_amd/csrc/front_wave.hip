// front_wave.hip -- k_front_s1: the wave-autonomous front kernel for chains with ONE half-band stage
// (0.25 <= r < 0.5: the NRSC-5 preset 2.4 MS/s -> 744.1875 kS/s is the headline case), with NO
// half-band stage (0.5 <= r < 1, template flag S0), and as the last stage behind k_cascade for S >= 2.
// Same arithmetic and stream bookkeeping as k_front (kernels.hip); what differs is the mapping onto
// CDNA4:
//
//   * one wavefront (64 lanes) owns a contiguous run of 512-sample tiles and carries every piece of
//     state itself (stage windows in its private LDS slice, next output index / phase in SGPRs), so
//     after the table load there is no workgroup barrier at all: 12 - 16 waves per CU drift apart and
//     cover each other's HBM / LDS latency;
//   * raw frames arrive by one coalesced 16-byte load per lane per 256 frames, issued one tile
//     ahead (register prefetch) and consumed before the next prefetch is issued, so the only
//     vmcnt wait of the loop lands on loads that are a whole tile old;
//   * the NCO-mixed samples are written to LDS split into even / odd streams in rows of 4 cf32
//     padded to 48 bytes (an odd number of 16-byte slots), so that a lane that owns 4 consecutive
//     half-band outputs reads its 24-sample window with 12 conflict-free ds_read_b128 at constant
//     offsets and keeps it in registers (20 taps x 4 outputs, re/im packed: v_pk_fma_f32 with the
//     tap in an SGPR pair);
//   * the half-band outputs go back to LDS on top of the odd-stream rows (dead by then), which keeps a
//     wave's slice at 6.8 KB;
//   * the polyphase stage gives each lane the 4 half-band samples it just produced: the output
//     (if any) that falls on each of them is found in closed form from the 24-bit phase, its 14
//     taps come from the 256-arm table in LDS, the 18-sample window again sits in registers;
//   * a lane's 2 .. 4 outputs of a tile are consecutive: they are compacted and (cs16) stored as one
//     8-byte store plus at most one more, a tile later.
//
// Tiles that touch the stream history, the end of the call, or an unaligned buffer ("edge" tiles)
// are handled by a scalar-load instantiation of the same tile routine; the host gives those tiles
// to a few extra waves (a handful of tiles each) so that the streaming waves run a loop with no
// special cases in it.
#include <hip/hip_runtime.h>

#include "../../include/iqgpu.h"
#include "dsp_device.hpp"
#include "kernels.hpp"
#include "wave_common.hpp"
#include "front_tiles.hpp"

namespace iqgpu {

size_t front_s1_lds_bytes() { return (size_t)kTabLds + (size_t)kS1Waves * kWaveLds; }   // the larger of the two shapes

// BPS: bytes per input frame on the vector-load path (2, 4, 8); 0 = no vector path for this format
//      FAST: cs16 in, unit gain, no iq correction, pre NCO on, no post NCO, cs16 out (the NRSC-5 preset
//      shape) -- the same arithmetic with every run-time switch resolved at compile time
//      S0: no half-band stage at all (0.5 <= r < 1, e.g. the cu8-nrsc5 preset 2.4 MS/s -> 1.488375 MS/s):
//      256-frame tiles, the mixed samples go straight to the polyphase rows
//      The FAST instantiation (121 VGPRs) and the 8-bit-input ones run 16 waves per workgroup, the others 12.
//      AGC: output AGC fused (gain before the pack, exact per-chunk peaks); those of the run-time-switched kernels run 12 waves
//      VAR: 0 = as the flags say; 1 = FAST without a mixer (NONCO); 4 = the last stage of a multi-stage chain (cf32 from k_cascade,
//      nothing pointwise, cs16 / cu8 / cf32 out: 16 waves instead of 12); 2, 3 = the cu8-nrsc5 preset shapes (S0, no shift, unit gain,
//      no dc blocker / iq correction, cu8 out) from cu8 resp. cs16 input with their run-time switches resolved at compile time;
//      5, 6 (late round 5) = the headline chain with a dc blocker (cs16 in and out, unit gain, no iq correction; 5: mixer in front, 6: none);
//      7 = the cu8-nrsc5 preset shape with a mixer in front (S0, cu8 in and out): 16 waves also with the fused AGC; 8, 9 = with a dc blocker (and a mixer / none)
template <int BPS, bool FAST, bool S0 = false, bool AGC = false, int VAR = 0>
__global__ __launch_bounds__((FAST || (BPS == 2 && !AGC) || (VAR >= 2 && VAR <= 4) || VAR >= 7) ? kS1Threads : kWThreads) void k_front_s1(const FrontArgs a_in)
{
    constexpr bool NONCO = VAR == 1;
    FrontArgs a = a_in;
    if (VAR == 4) {        // the last stage behind k_cascade: cf32 in, nothing pointwise; cs16 or cu8 out, or cf32 for a filter behind it (VAR 4 with S0 = false only)
        a.gain = 1.0f; a.iq_enable = 0; a.dc_enable = 0; a.nco_mode = 0; a.pnco_mode = 0; a.in_fmt = IQGPU_FMT_CF32;
        if (a.out_fmt != IQGPU_FMT_CU8 && a.out_fmt != IQGPU_FMT_CF32) a.out_fmt = IQGPU_FMT_CS16;
    } else if (VAR == 2 || VAR == 3) {        // constants instead of arguments: the compiler folds every switch they feed (128 -> 93 VGPRs, -17 % time)
        a.gain = 1.0f; a.iq_enable = 0; a.dc_enable = 0; a.nco_mode = 0; a.pnco_mode = 0;
        a.in_fmt = VAR == 2 ? (int)IQGPU_FMT_CU8 : (int)IQGPU_FMT_CS16; a.out_fmt = IQGPU_FMT_CU8;
    }
    if (VAR == 7) {               // the cu8-nrsc5 preset shape WITH a mixer in front (S0): cu8 in and out, unit gain, nothing else pointwise
        a.gain = 1.0f; a.iq_enable = 0; a.dc_enable = 0; a.pnco_mode = 0; a.in_fmt = IQGPU_FMT_CU8; a.out_fmt = IQGPU_FMT_CU8;
        if (a.nco_mode == 0) a.nco_mode = 1;               // (the mixer's direction stays the argument's)
    }
    if (VAR == 8 || VAR == 9) {   // ... and WITH a dc blocker (8: mixer in front too, 9: none)
        a.gain = 1.0f; a.iq_enable = 0; a.dc_enable = 1; a.pnco_mode = 0; a.in_fmt = IQGPU_FMT_CU8; a.out_fmt = IQGPU_FMT_CU8;
        if (VAR == 9) a.nco_mode = 0; else if (a.nco_mode == 0) a.nco_mode = 1;
    }
    if (VAR == 5 || VAR == 6) {   // the headline chain WITH a dc blocker (5: mixer in front, 6: none): cs16 in and out, unit gain, no iq correction
        a.gain = 1.0f; a.iq_enable = 0; a.dc_enable = 1; a.pnco_mode = 0; a.in_fmt = IQGPU_FMT_CS16; a.out_fmt = IQGPU_FMT_CS16;
        if (VAR == 6) a.nco_mode = 0; else if (a.nco_mode == 0) a.nco_mode = 1;       // (the mixer's direction stays the argument's)
    }
    if (a.run_if && *a.run_if == 0) return;         // a fallback launch whose fused predecessor stood
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool k16 = FAST || (BPS == 2 && !AGC) || (VAR >= 2 && VAR <= 4) || VAR >= 7;
    constexpr int kThr = k16 ? kS1Threads : kWThreads, kWv = k16 ? kS1Waves : kWaves;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    cf2   *s_nco = (cf2 *)smem, *s_nco_half = s_nco + 1024;
    float *s_arb = (float *)(smem + 2 * 1024 * 8);
    WaveLds w;
    w.XE = (char *)smem + kTabLds + wave * kWaveLds;
    w.XO = w.XE + kXRows * kRowB;
    w.HB = w.XO + kHBOff * kRowB;
    w.nco = s_nco; w.arb = s_arb;
    if (((unsigned)(size_t)(__attribute__((address_space(3))) const void *)s_nco & 8191u) != 0u) __builtin_trap();   // nco_phasor2 ORs the index into the base
    w.arb_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)s_arb;   // LDS byte address

    if (a.nco_mode != 0 || a.pnco_mode != 0) {
        const float sgn = (a.nco_mode < 0 || a.pnco_mode < 0) ? -1.0f : 1.0f;   // mix down: conj(phasor)
        // FAST: the cs16 normaliser 2^-15 is folded into the table -- power-of-two scaling commutes with
        // every rounding of x * (c + j s), so the mixed samples are bit-identical
        const float scl = FAST ? 1.0f / 32768.0f : 1.0f;
        for (int i = tid; i < 1024; i += kThr) {
            const cf2 v = a.nco_tab[i];
            s_nco[i] = cf2{v.x * scl, sgn * v.y * scl};
            if (FAST) s_nco_half[i] = cf2{v.x * (0.5f * scl), sgn * v.y * (0.5f * scl)};
        }
    }
    // polyphase taps: arm a lives in row a ^ (a >> 5).  The arms that the lanes of one gather touch
    // form an arithmetic progression (mod 256); with plain 56-byte rows that lands 3.3x the cycles of
    // a conflict-free ds_read_b64 on MI355X for the NRSC-5 step, with the XOR-folded rows 1.1x
    // (tools/lds_gather_bench.hip).
    for (int i = tid; i < 256 * 14; i += kThr) {
        const int arm = i / 14, k = i % 14;
        s_arb[(arm ^ (arm >> 5)) * 14 + k] = a.arb_table[arm * 16 + k];
    }
    for (int i = lane; i < kWaveLds / 16; i += 64) ((float4 *)w.XE)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    const int64_t gw = (int64_t)blockIdx.x * kWv + wave;
#ifdef IQGPU_STAGGER
    // de-synchronise the waves of the CU: they run the same phases (LDS-heavy, VALU-heavy) and
    // otherwise march through them in lockstep, so that LDS time and VALU time add up
    for (int i = 0; i < wave * IQGPU_STAGGER; ++i) __builtin_amdgcn_s_sleep(8);
#endif
    if (gw == 0 && a.frames_in < (int64_t)a.hist_cap) {
        const int keep = a.hist_cap - (int)a.frames_in;
        for (int i = lane; i < keep; i += 64) a.hist_out[i] = a.hist_in[i + (int)a.frames_in];
    }

    if (gw < a.w_n_edge) {
        // edge work: tiles [0, w_edge_ta) and [w_edge_tb, w_total_tiles) in runs of w_edge_tpw
        int64_t t0, t1;
        if (gw < a.w_n_edge1) { t0 = gw * a.w_edge_tpw; t1 = t0 + a.w_edge_tpw; if (t1 > a.w_edge_ta) t1 = a.w_edge_ta; }
        else { t0 = a.w_edge_tb + (gw - a.w_n_edge1) * a.w_edge_tpw; t1 = t0 + a.w_edge_tpw; if (t1 > a.w_total_tiles) t1 = a.w_total_tiles; }
        const int seg = (gw < a.w_n_edge1) ? (int)gw : (int)(gw + a.w_n_stream);   // DcGeom mode 1 order
        run_tiles<BPS, true, FAST, S0, AGC, NONCO>(a, w, lane, t0 - a.w_warm_tiles, t0, t1, seg);
    } else {
        const int64_t r = gw - a.w_n_edge;
        if (r >= a.w_n_stream) return;
        const int64_t t0 = w_run_start(a, r), t1 = w_run_start(a, r + 1);
        const int seg = (int)(a.w_n_edge1 + r);
        if (BPS != 0) run_tiles<BPS, false, FAST, S0, AGC, NONCO>(a, w, lane, t0 - a.w_warm_tiles, t0, t1, seg);
    }
}

// the specialised instantiations: the cs16 NRSC-5 preset shape, with a pre-resample shift or with none at all
static bool front_s1_fast_shape(const FrontArgs &a)
{
    return a.S == 1 && a.in_fmt == IQGPU_FMT_CS16 && a.out_fmt == IQGPU_FMT_CS16 && a.gain == 1.0f && !a.iq_enable &&
           !a.dc_enable && a.pnco_mode == 0 && !(a.dbg & kDbgNoFast);
}
static bool front_s1_fast_nonco(const FrontArgs &a) { return front_s1_fast_shape(a) && a.nco_mode == 0; }
// the fused AGC exists for the 8- and 16-bit-input instantiations, with or without the half-band stage (the shipped
// NRSC-5 presets: cs16 / cu8 in, cs16 or cu8 out, shift or none); chunks at least a tile long (one boundary per tile)
bool front_s1_agc_fusable(const FrontArgs &a)
{
    const bool in8 = a.in_fmt == IQGPU_FMT_CU8 || a.in_fmt == IQGPU_FMT_CS8;
    const bool in16 = a.in_fmt == IQGPU_FMT_CS16 || a.in_fmt == IQGPU_FMT_CU16 || a.in_fmt == IQGPU_FMT_SC16Q11;
    // (cf32 input: the last stage behind k_cascade, whose chunks are counted in the chain's input frames)
    const bool mid = a.in_fmt == IQGPU_FMT_CF32 && a.S == 1 && a.agc_shift >= 2;
    return (a.S == 0 || a.S == 1) && (in8 || in16 || mid) && a.agc_chunk_frames >= ((int64_t)256 << a.agc_shift) && !(a.dbg & kDbgAgcNoFuse);
}

// the cu8-nrsc5 preset shapes (iq_tool_presets.conf:190-214): 0.5 <= r < 1, nothing but unpack -> polyphase -> pack cu8
static int front_s1_plain_var(const FrontArgs &a)
{
    if (a.S != 0 || a.out_fmt != IQGPU_FMT_CU8 || a.gain != 1.0f || a.iq_enable || a.dc_enable || a.nco_mode != 0 || a.pnco_mode != 0 ||
        (a.dbg & kDbgNoFast)) return 0;
    return a.in_fmt == IQGPU_FMT_CU8 ? 2 : a.in_fmt == IQGPU_FMT_CS16 ? 3 : 0;
}
// the last stage behind k_cascade (or any cf32 stream) with nothing pointwise in it and 16- or 8-bit frames out: VAR 4
static bool front_s1_mid_var(const FrontArgs &a)
{
    return a.S == 1 && a.in_fmt == IQGPU_FMT_CF32 && a.gain == 1.0f && !a.iq_enable && !a.dc_enable && a.nco_mode == 0 && a.pnco_mode == 0 &&
           (a.out_fmt == IQGPU_FMT_CS16 || a.out_fmt == IQGPU_FMT_CU8 || a.out_fmt == IQGPU_FMT_CF32) && !(a.dbg & kDbgNoFast);
}
// the headline chain with `--dc-block`: S = 1, cs16 in and out, unit gain, dc blocker on, no iq correction, no mixer behind: VAR 5 / 6
static int front_s1_dc_var(const FrontArgs &a)
{
    if (a.S != 1 || a.in_fmt != IQGPU_FMT_CS16 || a.out_fmt != IQGPU_FMT_CS16 || a.gain != 1.0f || a.iq_enable || !a.dc_enable || a.pnco_mode != 0 ||
        (a.dbg & kDbgNoFast)) return 0;
    return a.nco_mode != 0 ? 5 : 6;
}
// the cu8-nrsc5 preset with `--freq-shift`: S = 0, cu8 in and out, unit gain, a mixer in front, nothing else pointwise: VAR 7
static int front_s1_s0_mixer_var(const FrontArgs &a)
{
    if (!(a.S == 0 && a.in_fmt == IQGPU_FMT_CU8 && a.out_fmt == IQGPU_FMT_CU8 && a.gain == 1.0f && !a.iq_enable && a.pnco_mode == 0 && !(a.dbg & kDbgNoFast)))
        return 0;
    if (a.dc_enable) return a.nco_mode != 0 ? 8 : 9;      // ... with `--dc-block`
    return a.nco_mode != 0 ? 7 : 0;
}
// wavefronts per workgroup of the instantiation that launch_front_s1() will pick for these arguments
static bool front_s1_sixteen(const FrontArgs &a)
{
    if (front_s1_plain_var(a) != 0 || front_s1_mid_var(a) || front_s1_s0_mixer_var(a) != 0) return true;
    // 4 waves per SIMD where the instantiation fits 128 VGPRs (nearly) without scratch: the specialised one,
    // and the 8-bit-input ones (2 - 4 spilled dwords; measured -8 % on the cu8-nrsc5 shape, -4 % on cu8 -> cs16).
    // The cs16 / cf32-input run-time-switched ones spill 7 - 21 dwords there and are faster with 12 waves.
    return front_s1_fast_shape(a) || ((a.in_fmt == IQGPU_FMT_CU8 || a.in_fmt == IQGPU_FMT_CS8) && !a.agc_fused);
}
int front_s1_waves(const FrontArgs &a) { return front_s1_sixteen(a) ? kS1Waves : kWaves; }

hipError_t launch_front_s1(const FrontArgs &a_in, hipStream_t s)
{
    const bool fast = front_s1_fast_shape(a_in), nonco = front_s1_fast_nonco(a_in);
    FrontArgs a_scaled;
    if (nonco) {                      // the cs16 normaliser rides on the half-band taps (exact: a power of two)
        a_scaled = a_in;
        for (float &h : a_scaled.hb0) h *= 1.0f / 32768.0f;
    }
    const FrontArgs &a = nonco ? a_scaled : a_in;
    const int waves = front_s1_sixteen(a) ? kS1Waves : kWaves;
    const size_t lds = (size_t)kTabLds + (size_t)waves * kWaveLds;
    const int64_t n_items = a.w_n_edge + a.w_n_stream;
    const unsigned grid = (unsigned)((n_items + waves - 1) / waves);
    if (grid == 0) return hipSuccess;
    int cls;
    switch (a.in_fmt) {
    case IQGPU_FMT_CS8: case IQGPU_FMT_CU8: cls = 2; break;
    case IQGPU_FMT_CS16: case IQGPU_FMT_CU16: case IQGPU_FMT_SC16Q11: cls = 4; break;
    case IQGPU_FMT_CF32: cls = 8; break;
    default: cls = 0; break;
    }
#define IQGPU_LAUNCH_S1Y(BPS, FAST, S0, AGC, VAR)                                                                     \
    do {                                                                                                              \
        static LdsAttrCache cache;                /* per instantiation */                                          \
        { const hipError_t e = cache.ensure((const void *)k_front_s1<BPS, FAST, S0, AGC, VAR>, lds); if (e != hipSuccess) return e; } \
        hipLaunchKernelGGL((k_front_s1<BPS, FAST, S0, AGC, VAR>), dim3(grid), dim3(waves * 64), lds, s, a);         \
    } while (0)
#define IQGPU_LAUNCH_S1X(BPS, FAST, S0, AGC) IQGPU_LAUNCH_S1Y(BPS, FAST, S0, AGC, 0)
#define IQGPU_LAUNCH_S1(BPS, FAST, S0) IQGPU_LAUNCH_S1X(BPS, FAST, S0, false)
    if (a.agc_fused && !(cls == 2 || cls == 4 || (cls == 8 && a.S == 1))) return hipErrorInvalidValue;     // (front_s1_agc_fusable)
    const int pvar = front_s1_plain_var(a);
    if (pvar == 2 && a.agc_fused) IQGPU_LAUNCH_S1Y(2, false, true, true, 2);
    else if (pvar == 2) IQGPU_LAUNCH_S1Y(2, false, true, false, 2);
    else if (pvar == 3 && a.agc_fused) IQGPU_LAUNCH_S1Y(4, false, true, true, 3);
    else if (pvar == 3) IQGPU_LAUNCH_S1Y(4, false, true, false, 3);
    else if (front_s1_s0_mixer_var(a) == 7 && a.agc_fused) IQGPU_LAUNCH_S1Y(2, false, true, true, 7);
    else if (front_s1_s0_mixer_var(a) == 7) IQGPU_LAUNCH_S1Y(2, false, true, false, 7);
    else if (front_s1_s0_mixer_var(a) == 8 && a.agc_fused) IQGPU_LAUNCH_S1Y(2, false, true, true, 8);
    else if (front_s1_s0_mixer_var(a) == 8) IQGPU_LAUNCH_S1Y(2, false, true, false, 8);
    else if (front_s1_s0_mixer_var(a) == 9 && a.agc_fused) IQGPU_LAUNCH_S1Y(2, false, true, true, 9);
    else if (front_s1_s0_mixer_var(a) == 9) IQGPU_LAUNCH_S1Y(2, false, true, false, 9);
    else if (a.S == 0) {
        if (cls == 2 && a.agc_fused) IQGPU_LAUNCH_S1X(2, false, true, true);
        else if (cls == 2) IQGPU_LAUNCH_S1(2, false, true);
        else if (cls == 4 && a.agc_fused) IQGPU_LAUNCH_S1X(4, false, true, true);
        else if (cls == 4) IQGPU_LAUNCH_S1(4, false, true);
        else if (cls == 8) IQGPU_LAUNCH_S1(8, false, true);
        else IQGPU_LAUNCH_S1(0, false, true);
    }
    else if (cls == 2 && a.agc_fused) IQGPU_LAUNCH_S1X(2, false, false, true);
    else if (cls == 2) IQGPU_LAUNCH_S1(2, false, false);
    else if (cls == 4 && nonco && a.agc_fused) IQGPU_LAUNCH_S1Y(4, true, false, true, 1);
    else if (cls == 4 && nonco) IQGPU_LAUNCH_S1Y(4, true, false, false, 1);
    else if (cls == 4 && fast && a.agc_fused) IQGPU_LAUNCH_S1X(4, true, false, true);
    else if (cls == 4 && front_s1_dc_var(a) == 5 && a.agc_fused) IQGPU_LAUNCH_S1Y(4, false, false, true, 5);
    else if (cls == 4 && front_s1_dc_var(a) == 5) IQGPU_LAUNCH_S1Y(4, false, false, false, 5);
    else if (cls == 4 && front_s1_dc_var(a) == 6 && a.agc_fused) IQGPU_LAUNCH_S1Y(4, false, false, true, 6);
    else if (cls == 4 && front_s1_dc_var(a) == 6) IQGPU_LAUNCH_S1Y(4, false, false, false, 6);
    else if (cls == 4 && a.agc_fused) IQGPU_LAUNCH_S1X(4, false, false, true);
    else if (cls == 4 && fast) IQGPU_LAUNCH_S1(4, true, false);
    else if (cls == 4) IQGPU_LAUNCH_S1(4, false, false);
    else if (cls == 8 && front_s1_mid_var(a) && a.agc_fused) IQGPU_LAUNCH_S1Y(8, false, false, true, 4);
    else if (cls == 8 && front_s1_mid_var(a)) IQGPU_LAUNCH_S1Y(8, false, false, false, 4);
    else if (cls == 8 && a.agc_fused) IQGPU_LAUNCH_S1X(8, false, false, true);
    else if (cls == 8) IQGPU_LAUNCH_S1(8, false, false);
    else IQGPU_LAUNCH_S1(0, false, false);
#undef IQGPU_LAUNCH_S1
#undef IQGPU_LAUNCH_S1X
#undef IQGPU_LAUNCH_S1Y
    return hipGetLastError();
}

// Splits the call's tiles into streaming runs (all tiles vector-loadable) and edge runs.  With fixed_tpw == 0
// the streaming tiles are dealt out as evenly as possible over the wave slots the edge runs leave free, so
// that the whole call is ONE round of resident workgroups (no workgroup waits for a CU to come free).
// Measured in round 2 (tools/clock.py, per-wave timeline): runs of equal length end between 0.31 and 0.47 ms and
// the XCDs finish 10 - 25 % apart, but evening that out inside a workgroup (a wave out of tiles taking half of
// the longest remaining run over an LDS compare-and-swap) made the launch 15 % SLOWER: a CU's throughput does
// not fall while its waves retire, so the static deal stays.
void plan_front_s1(FrontArgs &a, int64_t wave_slots, int fixed_tpw, int warm_tiles, int edge_tpw, int tile_frames, int align, int lead)
{
    const int kWTile = tile_frames;                 // 512 with a half-band stage in the kernel, 256 without
    const int64_t total = a.w_total_tiles;
    int vb;
    switch (a.in_fmt) {
    case IQGPU_FMT_CS8: case IQGPU_FMT_CU8: vb = 2; break;
    case IQGPU_FMT_CS16: case IQGPU_FMT_CU16: case IQGPU_FMT_SC16Q11: vb = 4; break;
    case IQGPU_FMT_CF32: vb = 8; break;
    default: vb = 0; break;
    }
    a.w_warm_tiles = warm_tiles;
    a.w_edge_tpw = edge_tpw;
    a.w_wsum = 0; a.w_wpw = 0;                       // equal runs unless the caller weights them afterwards (weight_runs)
    int64_t ta = total, tb = total;
    // (cs24 output takes six byte stores per frame: the streaming loop counts on one)
    if (vb != 0 && a.out_fmt != IQGPU_FMT_CS24 && a.raw_aligned && (((int64_t)a.rem0 * vb) & 15) == 0) {
        // tile t is streamable iff 512 t - rem0 >= 0 and 512 (t + 1) - rem0 <= frames_in - hist_cap
        const int64_t t_min = (a.rem0 + lead + kWTile - 1) / kWTile;     // (lead: frames in front of its first tile that a streaming run reads as well)
        const int64_t lim = a.frames_in - (int64_t)a.hist_cap + a.rem0;
        const int64_t t_max = lim >= kWTile ? lim / kWTile - 1 : -1;              // last streamable tile
        // a run over tiles [t0, t1) touches tiles [t0 - warm, t1] (one past its end for the prefetch)
        if (t_max > t_min + warm_tiles) { ta = t_min + warm_tiles; tb = t_max; }
        // (k_front_mid's 768-frame tiles: the edge runs are handed to the 512-frame tile routine, so the streaming part starts
        //  and ends at an even tile)
        if (align > 1 && ta < total) {
            ta = (ta + align - 1) / align * align; tb = tb / align * align;
            if (tb <= ta) { ta = total; tb = total; }
        }
    }
    if (ta > total) ta = total;
    if (tb > total) tb = total;
    a.w_edge_ta = ta; a.w_edge_tb = tb;
    a.w_n_edge1 = (ta + edge_tpw - 1) / edge_tpw;
    a.w_n_edge = a.w_n_edge1 + (total - tb + edge_tpw - 1) / edge_tpw;
    const int64_t ns = tb - ta;
    a.w_n_stream = 0; a.w_run_q = 0; a.w_run_r = 0;
    if (ns > 0) {
        int64_t w;
        if (fixed_tpw > 0) {
            w = (ns + fixed_tpw - 1) / fixed_tpw;
        } else {
            // one run per free wave slot; a short call gets one tile per run (the slots are free anyway, and a
            // wave's time is warm-up + run: what counts for a small batch is its latency)
            w = wave_slots - a.w_n_edge;
            if (w > ns) w = ns;
            if (w < 1) w = 1;
        }
        a.w_n_stream = w; a.w_run_q = ns / w; a.w_run_r = ns % w;
    }
}

} // namespace iqgpu
