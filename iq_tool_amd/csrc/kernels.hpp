// kernels.hpp -- argument blocks and launchers of the gfx950 kernels (defined in kernels.hip).
#pragma once
#include <atomic>
#include <mutex>

#include <cstdint>
#include <hip/hip_runtime_api.h>

#if defined(__HIPCC__)
#define IQGPU_HD __host__ __device__
#else
#define IQGPU_HD
#endif

namespace iqgpu {

constexpr int kTile = 2048;        // input samples per workgroup tile
constexpr int kThreads = 256;      // 4 wavefronts of 64
constexpr int kMaxS = 12;
constexpr int kArbHist = 16;       // 13 needed, padded
constexpr int kWTile = 512;        // input samples per wavefront tile (k_front_s1)
constexpr int kWaves = 12;         // wavefronts per workgroup in k_cascade (3 per SIMD, 1 workgroup per CU)
constexpr int kWThreads = kWaves * 64;
constexpr int kCascMaxWaves = 16;  // k_cascade with two or more stages (latency-bound: 4 waves per SIMD when the LDS slices allow)
#ifndef IQGPU_S1_WAVES
#define IQGPU_S1_WAVES 16
#endif
constexpr int kS1Waves = IQGPU_S1_WAVES;   // wavefronts per workgroup in k_front_s1 (4 per SIMD, 1 workgroup per CU)
constexpr int kS1Threads = kS1Waves * 64;
constexpr int kFirOutTile = 1024;  // outputs per workgroup tile in the FIR kernel
constexpr int kFirTapChunk = 256;  // taps staged in LDS per pass
constexpr int kFftMinTaps = 96;    // FIR-kind filters at least this long run as overlap-save (k_fftconv)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device and costs a few microseconds: launchers
// call it once per (kernel instantiation, device, size) through this little cache.  Handles on one device may
// be driven by different threads (the harness runs one thread per shard): the size is published only after
// the attribute call has succeeded, under a lock, so no thread can launch ahead of it and a failure is retried.
struct LdsAttrCache {
    std::mutex mu;
    std::atomic<size_t> configured[64];
    LdsAttrCache() { for (auto &c : configured) c.store(0, std::memory_order_relaxed); }
    hipError_t ensure(const void *func, size_t lds)
    {
        int dev = 0;
        const bool cached = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
        if (cached && lds <= configured[dev].load(std::memory_order_acquire)) return hipSuccess;
        std::lock_guard<std::mutex> g(mu);
        if (cached && lds <= configured[dev].load(std::memory_order_relaxed)) return hipSuccess;
        const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess && cached) configured[dev].store(lds, std::memory_order_release);
        return e;
    }
};

struct cf2 { float x, y; };
struct AgcState;
struct cd2 { double x, y; };

// ---------------------------------------------------------------------------------------------
// k_front: raw -> [unpack, gain] -> [dc block] -> [iq correct] -> [pre NCO] -> [half-band cascade ->
//          arbitrary polyphase] -> [post NCO] -> [pack] -> out
// ---------------------------------------------------------------------------------------------
// diagnostic switches (IQGPU_NO_FAST, IQGPU_AGC_NOFUSE, IQGPU_NO_RAW0, IQGPU_NO_KT, IQGPU_FFT_NO_R16): read from the
// environment ONCE, in iqgpu_chain_create, and carried in the launch arguments -- the launch path itself never calls getenv
enum : uint32_t { kDbgNoFast = 1u, kDbgAgcNoFuse = 2u, kDbgNoRaw0 = 4u, kDbgNoKT = 8u, kDbgFftNoR16 = 16u, kDbgNoFat = 32u, kDbgForceFat = 64u, kDbgUseFat = 128u, kDbgMid8 = 256u,
                  kDbgNoS2 = 512u,         // IQGPU_NO_S2=1: two-stage chains keep k_cascade + k_front_s1 instead of the fused k_front_s2
                  kDbgNoMid8bit = 8192u,   // IQGPU_NO_MID_8BIT=1: S = 1 chains with 8-bit frames on either side keep k_front_s1 instead of k_front_mid
                  kDbgNoCasc2 = 4096u,     // IQGPU_NO_CASC2=1: raw cu8 cascades keep k_cascade's one tile per trip instead of k_cascade2's two
                  kDbgNoP0 = 2048u,        // IQGPU_NO_P0=1: chains without a half-band stage keep k_front_s1<S0> instead of k_front_p0
                  kDbgFuseFilter = 16384u, // fuse_filter=1: S = 0 chains with a filter behind the resampler run k_p0fft16 (one kernel, no cf32 stream) instead of
                                           // k_front_p0 + k_fftconv16 -- opt-in: parity-green and SLOWER than the two kernels (profiles/r06_fused_filter.md)
                  kDbgNoFusedMove = 1024u }; // IQGPU_NO_FUSED_MOVE=1: the filter's history moves by a copy kernel, not inside the filter kernel

struct FrontArgs {
    uint32_t    dbg;          // kDbg* switches of the chain
    uint32_t    tap_fold;                  // k_front_fat / k_front_mid: arm a in slot a ^ (a >> 5) of a tap plane instead of slot a (front_tap_fold())
    // input
    const void *raw;          // frames_in new samples, in_fmt
    const cf2  *hist_in;      // hist_cap processed samples that precede this call
    cf2        *hist_out;     // the same for the next call
    int64_t     frames_in;
    int32_t     hist_cap;
    int32_t     rem0;         // samples of the open 2^S group carried in from the previous call
    int32_t     in_fmt;
    float       gain;
    int32_t     raw_aligned;  // raw pointer is 16-byte aligned
    // dc blocker
    int32_t     dc_enable;
    float       dc_c;         // 1 - alpha
    float       dc_a;         // alpha' = 1 - c (exact)
    float       dc_cpow[8];   // c^(4*2^k), k=0..5; [6] = c^256; [7] = c^1024
    double      dc_logc;      // log(c), for c^(-n) at block start
    const cd2  *dc_carry;     // per block: state before the block's first new sample
    // iq correction
    int32_t     iq_enable;
    float       iq_magp1, iq_phase;
    // pre-resample NCO
    int32_t     nco_mode;     // 0 off, +1 mix up, -1 mix down
    uint32_t    nco_theta0;   // phase of input sample i_rel = 0
    uint32_t    nco_dtheta;
    const cf2  *nco_tab;      // 1024 x {cos, sin}
    // resampler
    int32_t     mode;         // 0 = no resampler (pointwise only), 1 = decimating msresamp
    int32_t     S;
    int32_t     m[kMaxS];
    int32_t     tap_off[kMaxS];
    int32_t     lvl_off[kMaxS + 2]; // LDS offset (in cf32) of each level buffer; [S+1] = total
    int32_t     n_hb_taps;
    const float *hb_taps;     // branch taps of every stage, pre-scaled by 0.5
    const float *arb_table;   // [256][16]
    uint32_t    step;
    uint32_t    n_est;        // floor((kTile >> S) * 2^24 / step): outputs of a full tile, or one more
    uint64_t    phi0;         // phase of output k_rel = 0 relative to group q_rel = 0
    int64_t     n_groups;     // complete groups available in this call
    int64_t     n_out;        // outputs of this call
    int64_t     total_tiles;
    int32_t     tiles_per_block;
    int32_t     warm_tiles;
    // geometry of the wave-autonomous fast path (k_front_s1): tiles of kWTile samples
    int64_t     w_total_tiles;
    int32_t     w_warm_tiles;
    int32_t     w_edge_tpw;     // tiles per wave for edge runs
    // streaming runs: run w (0 <= w < w_n_stream) covers tiles [w_run_start(w), w_run_start(w + 1)) of
    // [w_edge_ta, w_edge_tb): w_run_q tiles each, one more for the first w_run_r runs
    int64_t     w_n_stream, w_run_q, w_run_r;
    // ... or, with w_wsum != 0, runs WEIGHTED by the wave's slot in its workgroup: global wave g (edge runs first) sits in slot
    // g % w_wpw, slots come in classes of four (the first / second / third wave of each SIMD) and class c weighs w_wt[c].  The
    // oldest wave of a SIMD wins every arbitration: of three equal runs it finishes the first at 260 us, the second at 312, the
    // third at 357 (profiles/r03_wave_timeline.txt), and the SIMD spends the last quarter of the launch with one or two waves.
    // Runs of 1.3 : 1 : 0.7 bring the three ends within 15 % of each other (more skew turns the order around): -2 % of kernel time.
    // w_wsum = the weight of all streaming waves.
    int32_t     w_wpw, w_wt[4];
    int64_t     w_wsum;
    // run stealing (k_front_mid, front_mid.hip): one 8-byte descriptor per wave of the launch, {end : 32 | next : 32} in tiles from
    // w_edge_ta.  The owner claims tile `next` with a returning agent-scope add at the start of every tile and learns its current
    // `end` from the value that comes back; a wave out of work halves the longest remaining run among 64 sampled descriptors by a
    // compare-and-swap on the whole word (so a tile is handed to exactly one wave) and re-runs one warm-up tile.  NULL = static runs.
    // All descriptors are exhausted (next >= end) when a launch ends, which is the state the next launch needs.
    unsigned long long *w_steal;
    int32_t     w_steal_min, w_steal_rounds;   // a run is split only while it has at least w_steal_min unclaimed tiles; sampling rounds before a wave gives up
    int64_t     w_run_stride;                  // > 0 (fixed-length runs, block_samples != 0, more of them than resident waves): the launch holds
                                               // ONE round of workgroups (tables and tap planes filled once per workgroup instead of once per 12
                                               // runs) of w_run_stride + w_n_edge waves; workgroup b owns runs b, b + gridDim.x, ... and its waves
                                               // -- the edge waves too, once their edge runs are done -- take them in order through a counter in
                                               // LDS (the oldest wave of a SIMD is the fastest: it ends up with more runs).  0 = one run per wave
    int32_t     w_steal_stride, w_steal_lanes; // descriptor w sits at w_steal[w * w_steal_stride] (spread over the memory channels: the claims of
                                               // 3072 waves and the thieves' samples otherwise all land on the few channels that hold 24 KB);
                                               // lanes that sample per round (<= 64)
    int64_t     w_edge_ta, w_edge_tb;   // edge tiles: [0, ta) and [tb, total)
    int64_t     w_n_edge1, w_n_edge;    // edge runs in the first region / in both
    // k_front_p0 (front_p0.hip): the streaming outputs of the launch [p0_k_a, p0_k_b) (call-relative), dealt out as steps of 320 --
    // w_n_stream runs of w_run_q steps, one more for the first w_run_r; the last frame a window load may start at
    int64_t     p0_k_a, p0_k_b, p0_f_max;
    float       hb0[24];      // branch taps of stage 0 (pre-scaled by 0.5) for s_load access
    // fused output AGC of the locked phase (front_wave.hip, agc.hip): the kernel multiplies by the gain in *agc_state
    // before the pack and records max |y|^2 per chunk; k_agc_verify then confirms that no chunk changes the gain
    int32_t     agc_fused;
    const struct AgcState *agc_state;
    unsigned long long *agc_peak2;      // [n_chunks], double bits, zeroed before the launch
    int64_t     agc_chunk_frames;       // >= 256 << agc_shift (at most one chunk boundary per tile)
    int32_t     agc_shift, agc_rem;     // polyphase-input sample q of the call needs the call's input frames up to
                                        // ((q + 1) << agc_shift) - agc_rem - 1  (S and rem of the WHOLE chain: behind
                                        // k_cascade the kernel itself sees a one-stage chain on the intermediate stream)
    const int32_t *run_if;              // not NULL: the launch does nothing unless *run_if != 0 (fallback launches)
    void       *sink;         // 64 KiB diagnostic scratch (per-phase cycle counters of -DIQGPU_STAMPS builds)
    // k_cascade (cascade_wave.hip): the first casc_K stages of an S >= 2 chain, cf32 out
    int32_t     casc_K;
    int32_t     casc_wave_lds;          // bytes of LDS per wavefront
    float       casc_taps[4][12];       // branch taps of stages 0 .. K-1 (pre-scaled by 0.5), 2m <= 10 each
    cf2        *casc_out;               // intermediate stream, index 0 = first sample this call completes
    int64_t     casc_n_out;             // (rem0 + frames_in) >> K
    // post-resample NCO
    int32_t     pnco_mode;
    uint32_t    pnco_theta0, pnco_dtheta;
    // output
    int32_t     out_fmt;
    void       *out;
};

// weight of the global waves [0, g) of a launch with wpw waves per workgroup (slot classes of four: the first / second / ... wave
// of each SIMD)
IQGPU_HD inline int64_t w_cum_weight(const int32_t *wt, int wpw, int64_t g)
{
    const int nc = wpw / 4;
    int64_t per = 0;
    for (int c = 0; c < nc; ++c) per += 4 * (int64_t)wt[c];
    const int64_t b = g / wpw;
    int64_t s = g % wpw, cw = b * per;
    for (int c = 0; c < nc; ++c) { const int64_t n = s < 4 ? s : 4; cw += n * wt[c]; s -= n; }
    return cw;
}
IQGPU_HD inline int64_t w_run_start(const FrontArgs &a, int64_t w)
{
    return a.w_edge_ta + w * a.w_run_q + (w < a.w_run_r ? w : a.w_run_r);
}
// ... of a launch whose plan may be weighted (k_front_mid only: the 64-bit division below costs the other kernels registers they
// do not have -- with it inlined the last-stage instantiation of k_front_s1 spilled 968 bytes and ran at half its speed)
IQGPU_HD inline int64_t w_run_start_weighted(const FrontArgs &a, int64_t w)
{
    if (a.w_wsum != 0) {
        if (w >= a.w_n_stream) return a.w_edge_tb;
        const int64_t cw = w_cum_weight(a.w_wt, a.w_wpw, w + a.w_n_edge) - w_cum_weight(a.w_wt, a.w_wpw, a.w_n_edge);
        return a.w_edge_ta + cw * (a.w_edge_tb - a.w_edge_ta) / a.w_wsum;
    }
    return w_run_start(a, w);
}
// switches a plan (plan_front_s1) to weighted runs; wt[c] > 0 for the wpw / 4 slot classes.  Every run keeps at least two tiles.
inline void weight_runs(FrontArgs &a, int wpw, const int32_t *wt)
{
    a.w_wsum = 0;
    if (a.w_n_stream < 16 * wpw || wpw % 4 != 0 || wpw > 16) return;
    int32_t lo = wt[0], hi = wt[0];
    for (int c = 0; c < wpw / 4; ++c) { a.w_wt[c] = wt[c]; if (wt[c] < lo) lo = wt[c]; if (wt[c] > hi) hi = wt[c]; }
    if (lo <= 0 || (a.w_edge_tb - a.w_edge_ta) / a.w_n_stream * lo / hi < 2) return;      // (short runs: leave them equal)
    a.w_wpw = wpw;
    a.w_wsum = w_cum_weight(a.w_wt, wpw, a.w_n_stream + a.w_n_edge) - w_cum_weight(a.w_wt, wpw, a.w_n_edge);
}

size_t front_lds_bytes(const FrontArgs &a);
hipError_t launch_front(const FrontArgs &a, int n_blocks, hipStream_t s);
// leading stages of a multi-stage decimation as a wave-autonomous kernel (cascade_wave.hip)
constexpr int kCascMaxK = 4;
bool cascade_supported(const int *m_run_order, int S);
size_t cascade_wave_lds(const FrontArgs &a, bool two_tile_trips = true);   // (false: k_cascade's own layout only)
hipError_t launch_cascade(const FrontArgs &a, hipStream_t s);
int cascade_waves(const FrontArgs &a);    // needs casc_wave_lds
// ... with two tiles per trip of a streaming wave for raw cu8 frames (cascade2.hip): the chain shape (needs in_fmt, gain, the
// pointwise switches, casc_K, m[]); the call (needs the run geometry and casc_wave_lds too); bytes of a wave's slice
bool cascade2_shape(const FrontArgs &a);
bool cascade2_applies(const FrontArgs &a);
void cascade2_set_min_run(int n);                       // diagnostics (iqgpu_debug_set "casc2_min_run"); 0 = the built-in bound
int cascade2_wave_lds(int K, int in_fmt);
hipError_t launch_cascade2(const FrontArgs &a, hipStream_t s);
// one half-band stage (m = 10), no dc blocker: wave-autonomous kernel (front_wave.hip)
size_t front_s1_lds_bytes();
hipError_t launch_front_s1(const FrontArgs &a, hipStream_t s);
int front_s1_waves(const FrontArgs &a);   // needs S, formats, gain, iq / dc / nco switches
// the same shape as fewer, fatter waves (front_fat.hip): 8 waves per CU, 1024-frame tiles, software-pipelined LDS reads
constexpr int kFatTile = 1024;
constexpr int kFatMinTilesPerWave = 8;      // calls shorter than this many tiles per fat wave keep k_front_s1
int front_fat_waves();
bool front_fat_shape(const FrontArgs &a);  // needs S, formats, gain, iq / dc / nco switches, step, agc_fused, dbg
size_t front_fat_lds_bytes();
hipError_t launch_front_fat(const FrontArgs &a, hipStream_t s);
// ... and with 6 half-band outputs per lane, 12 waves per CU, 768-frame tiles (front_mid.hip)
constexpr int kMidLead = 20;                 // a streaming run of k_front_mid reads this many frames in front of its first tile
int front_mid_waves();
int front_mid_max_edge_waves();          // launches with more edge runs than this stay on k_front_s1
int front_mid_nl(const FrontArgs &a);    // half-band outputs per lane of the instantiation for these arguments (8, 6; 0 = not its shape)
int front_mid_tile(int nl);              // its tile: 128 nl frames
// placement of the arms in the tap planes for a chain's step and nl outputs per lane: 1 = folded, 0 = linear (whichever a model of
// the half-wave's bank pairs prices lower, the fold's two extra instructions per slot counted)
int front_tap_fold(uint32_t step, int nl);
hipError_t launch_front_mid(const FrontArgs &a, hipStream_t s);
// chains without a half-band stage as an output-major polyphase kernel with register-resident taps (front_p0.hip)
int front_p0_waves();
int front_p0_max_edge_waves();
int front_p0_edge_tpw();
bool front_p0_shape(const FrontArgs &a);        // needs S, formats, gain, iq / dc / nco switches, step, agc_fused, agc_chunk_frames, agc_shift, dbg
void plan_front_p0(FrontArgs &a, int64_t wave_slots);   // after plan_front_s1 (256-frame tiles): needs phi0, step, frames_in, in_fmt
hipError_t launch_front_p0(const FrontArgs &a, hipStream_t s);
// two-stage chains (S = 2) with both half-bands and the polyphase in one kernel (front_s2.hip): a1 = the chain as k_cascade sees
// it (K = 1), a2 = the last stage as k_front_s1 sees it, planned in ITS tiles (512 intermediate samples = 1024 input frames)
int front_s2_waves();
int64_t front_s2_mid_samples(const FrontArgs &a2);  // cf32 samples of the intermediate buffer it needs (edge waves only): needs the plan
bool front_s2_shape(const FrontArgs &a1);          // needs casc_K, m[0], in_fmt
hipError_t launch_front_s2(const FrontArgs &a1, const FrontArgs &a2, hipStream_t s);
// fills the w_* geometry from frames_in / rem0 / hist_cap / alignment (w_total_tiles must be set): at most
// wave_slots runs in all (edge runs included) when fixed_tpw == 0, else streaming runs of fixed_tpw tiles
void plan_front_s1(FrontArgs &a, int64_t wave_slots, int fixed_tpw, int warm_tiles, int edge_tiles_per_wave, int tile_frames = kWTile, int align = 1, int lead = 0);

// ---------------------------------------------------------------------------------------------
// DC-blocker carry: per-segment aggregates, then a sequential scan over the (few) segments
// ---------------------------------------------------------------------------------------------
// Where the front kernel's independent pieces start in the call's new samples.  Segment s covers
// [start(s), start(s+1)); start(0) = 0; starts clamp to [0, frames_in].
struct DcGeom {
    int32_t mode;          // 0: blocks of k_front (regular spacing); 1: wave runs of k_cascade
    int32_t n_seg;
    int64_t frames_in;
    // mode 0: start(s) = seg_first + (s - 1) seg_len for s >= 1
    int64_t seg_first, seg_len;
    // mode 1: runs in stream order -- n_edge1 edge runs of edge_tpw tiles from tile 0, n_stream streaming runs
    // from tile ta (run_q tiles each, one more for the first run_r), then edge runs from tile tb; a run
    // starts warm tiles early
    int64_t n_edge1, n_stream, edge_tpw, run_q, run_r, ta, tb;
    int32_t warm, rem0, tile;   // tile: input frames per tile of the wave kernel (512, or 256 without a half-band)
};
IQGPU_HD inline int64_t dc_seg_start(const DcGeom &g, int s)
{
    if (s <= 0) return 0;
    int64_t v;
    if (g.mode == 0) {
        v = g.seg_first + (int64_t)(s - 1) * g.seg_len;
    } else {
        int64_t t0;
        if (s < g.n_edge1) t0 = (int64_t)s * g.edge_tpw;
        else if (s < g.n_edge1 + g.n_stream) { const int64_t w = s - g.n_edge1; t0 = g.ta + w * g.run_q + (w < g.run_r ? w : g.run_r); }
        else t0 = g.tb + (s - g.n_edge1 - g.n_stream) * g.edge_tpw;
        v = (t0 - g.warm) * g.tile - g.rem0;
    }
    if (v < 0) v = 0;
    if (v > g.frames_in) v = g.frames_in;
    return v;
}

struct DcPrefixArgs {
    const void *raw;
    int32_t     in_fmt;
    float       gain;
    int32_t     raw_aligned;
    float       c;            // 1 - alpha
    double      logc;
    DcGeom      geom;
    cf2        *agg;          // [n_seg]
};
hipError_t launch_dc_prefix(const DcPrefixArgs &a, hipStream_t s);

struct DcScanArgs {
    const cf2 *agg;
    cd2       *carry;         // [n_seg]
    cd2       *state;         // in: state before sample 0; out: state after the last sample
    DcGeom     geom;
    double     logc;
};
hipError_t launch_dc_scan(const DcScanArgs &a, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// k_fir: time-domain FIR over the cf32 filter-input buffer, [post NCO], pack
// ---------------------------------------------------------------------------------------------
struct FirArgs {
    const cf2 *fbuf;          // [ntaps-1 history][pending + new samples]
    const cf2 *taps;          // ntaps complex taps h[k]
    int32_t    ntaps;
    int32_t    is_complex;
    int64_t    n_emit;        // outputs to produce
    int32_t    pnco_mode;
    uint32_t   pnco_theta0, pnco_dtheta;
    const cf2 *nco_tab;
    int32_t    out_fmt;
    void      *out;
    // the next call's buffer front (history + still-pending samples) lives in the OTHER buffer of the pair: nobody reads that
    // one during this launch, so the last workgroup copies it on the side (one launch fewer per step than a copy kernel)
    cf2       *move_dst;
    const cf2 *move_src;
    int64_t    move_n;
};
hipError_t launch_fir(const FirArgs &a, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// the chunk map of the output AGC (agc.hip; the fused epilogue of k_fftconv16 needs it too)
// ---------------------------------------------------------------------------------------------
// How one process() call maps its input chunks (chunk_frames input frames each, counted from the
// start of the call) to ranges of its output: closed form, so kernels and host agree without a table.
struct AgcGeom {
    int64_t  frames_in;
    int64_t  chunk_frames;
    int32_t  n_chunks;
    int32_t  mode;         // 0 = one output per input, 1 = decimating front kernel, 2 = k_interp path
    int32_t  rem;          // mode 1: samples of the open 2^S group before the call
    int32_t  S;            // mode 1: group shift; mode 2: interpolator stages
    uint64_t phi;          // resampler phase at the start of the call
    uint32_t step;
    uint32_t block;        // fftfilt block size (0 = none)
    uint64_t fpending;     // samples waiting in front of the block filter
};
// outputs of the call produced by its chunks 0 .. c (c = -1 -> 0)
IQGPU_HD inline int64_t agc_out_end(const AgcGeom &g, int64_t c)
{
    if (c < 0) return 0;
    int64_t f = (c + 1) * g.chunk_frames;
    if (f > g.frames_in) f = g.frames_in;
    uint64_t n = (uint64_t)f;
    if (g.mode == 2 && g.block) n = ((g.fpending + n) / g.block) * g.block;      // pre filter, then resampler
    if (g.mode != 0) {
        const uint64_t span = (g.mode == 1 ? (((uint64_t)g.rem + n) >> g.S) : n) << 24;
        n = span > g.phi ? (span - g.phi + g.step - 1) / g.step : 0;
        if (g.mode == 2) n <<= g.S;
    }
    if (g.mode != 2 && g.block) n = ((g.fpending + n) / g.block) * g.block;      // filter behind the resampler
    return (int64_t)n;
}

// the chunk of the call that output k (< the call's n_emit) belongs to: the smallest c with agc_out_end(g, c) > k, in closed form
// (the inverse of the chain above, one division by the chunk length) -- modes 0 and 1, what the fused AGC behind a user filter needs
IQGPU_HD inline int64_t agc_chunk_of_output(const AgcGeom &g, int64_t k)
{
    // outputs of the resampler (or input frames, mode 0) that must exist before output k does
    uint64_t n_min = (uint64_t)k + 1;
    if (g.block) {
        const uint64_t need = ((uint64_t)k / g.block + 1) * g.block;        // emitted count that covers k: a whole number of blocks
        n_min = need > g.fpending ? need - g.fpending : 0;
    }
    uint64_t f_min = n_min;                                                   // input frames of the call that must have arrived
    if (g.mode == 1) {
        if (n_min == 0) return 0;
        const uint64_t x = (n_min - 1) * (uint64_t)g.step + g.phi;           // complete groups G with G 2^24 > x
        const uint64_t g_min = (x >> 24) + 1;
        const uint64_t have = g_min << g.S;
        f_min = have > (uint64_t)g.rem ? have - (uint64_t)g.rem : 0;
    }
    if (f_min == 0) return 0;
    return (int64_t)((f_min + (uint64_t)g.chunk_frames - 1) / (uint64_t)g.chunk_frames) - 1;
}

// ---------------------------------------------------------------------------------------------
// k_fftconv: FFT-kind user filter as overlap-save block convolution in LDS (fftconv.hip)
// ---------------------------------------------------------------------------------------------
constexpr int kMaxFftN = 16384;    // k_fftconv16 transforms in place: N cf32 = 128 KiB (+ pad) of LDS
constexpr int kMaxFftN4 = 8192;    // the radix-4 ping-pong kernel (only used below N = 1024)
constexpr int kFftMaxThreads = 1024;
// k_p0fft16 (round 6): the resampler of a chain WITHOUT a half-band stage computed straight into the filter's overlap-save windows
// -- a window sample IS a polyphase output -- so that no cf32 stream stands between the two in HBM.  What the window fill needs of
// the front kernel's arguments (k_front_p0's: front_p0.hip):
struct P0Feed {
    const void  *raw;         // frames_in new frames, in_fmt (cu8 / cs8 / cs16); NULL = not fused (k_fftconv16 reads fbuf)
    const cf2   *hist_in;     // hist_cap processed samples that precede this call
    cf2         *hist_out;    // the same for the next call (written when write_state)
    int64_t      frames_in;
    int32_t      hist_cap, in_fmt;
    const float *arb_table;   // [256][16]
    uint32_t     step, tap_fold;
    uint64_t     phi0;        // phase of output 0 of the call
    int64_t      n_res;       // resampler outputs of this call: fbuf entry pre + k <-> output k
    int64_t      pre;         // fbuf entries in front of them: the filter's L - 1 history + the samples still pending (read from fbuf)
    int64_t      k_a, k_b;    // outputs [k_a, k_b): every one of the 24 frames a lane loads around them lies inside the call's input
    int32_t      write_state; // 1: the launch also leaves hist_out and the next call's buffer front (move_dst); 0: the AGC fallback
    int32_t      grid;        // persistent workgroups of the launch (two per CU: what their LDS -- transform buffer + tap planes -- allows)
};
struct FftConvArgs {
    uint32_t   dbg;           // kDbg* switches of the chain
    const cf2 *fbuf;          // [ntaps-1 history][pending + new samples]
    int64_t    fbuf_len;      // valid cf32 entries in fbuf
    const cf2 *hfreq;         // FFT_N(taps) / N
    const cf2 *twiddle;       // exp(-2 pi i k / N), k < N
    int32_t    ntaps;
    int32_t    log2n;         // N = 1 << log2n >= 2 (ntaps - 1); each workgroup emits N - (ntaps - 1) outputs
    int32_t    threads;       // 0 = auto
    int64_t    n_emit;
    int32_t    pnco_mode;
    uint32_t   pnco_theta0, pnco_dtheta;
    const cf2 *nco_tab;
    int32_t    out_fmt;
    void      *out;
    cf2       *move_dst;      // as FirArgs: the next call's buffer front, copied by the last workgroup
    const cf2 *move_src;
    int64_t    move_n;
    // fused output AGC of the locked phase (round 5; k_fftconv16 only): the shipped -usb / -lsb presets run filter -> digital AGC ->
    // pack.  As in the front kernels the epilogue multiplies by the gain it finds in *agc_state before the pack and leaves max |y|^2
    // per chunk (float, as k_front_mid: peak_approx) in agc_peak2; k_agc_classify confirms every chunk afterwards or raises the flag
    // behind which the same launch with cf32 output (run_if) and the unfused AGC kernels redo the call.  A workgroup's outputs lie in
    // at most two chunks (the host fuses only when a chunk's outputs outnumber a workgroup's).
    int32_t    agc_fused;
    const AgcState *agc_state;
    unsigned long long *agc_peak2;
    AgcGeom    agc_geom;
    const int32_t *run_if;    // not NULL: the launch does nothing unless *run_if != 0 (the fallback behind a fused launch)
    // overlap-save geometry other than the default (0 = default: win = N, vout = N - (ntaps - 1)): a block's window holds `win` stream
    // samples -- the rest of its N points is zero -- and the block emits `vout` <= win - (ntaps - 1) outputs.  k_p0fft16 fills windows
    // in whole polyphase steps (win a multiple of 320) and moves from block to block by a whole number of steps (vout too), so that a
    // lane-slot's arm -- and with it the taps it holds -- repeats; k_fftconv16 takes the same two numbers so that the two paths can be
    // compared byte for byte (radix-16 kernels only)
    int32_t    win, vout;
    P0Feed     feed;
};
hipError_t launch_fftconv(const FftConvArgs &a, hipStream_t s);
// the geometry k_p0fft16 runs a filter of ntaps at N = 2^log2n with: false when no whole number of steps fits
bool p0fft_geometry(int log2n, int ntaps, int *win, int *vout);
// which chains (the front kernel's shape test on the FRONT arguments -- k_front_p0's, cf32 out -- plus the filter's transform)
bool p0fft_shape(const FrontArgs &front, int log2n, int ntaps);
hipError_t launch_p0fft_cu8(const FftConvArgs &a, size_t lds, hipStream_t s);     // p0fft_cu8.hip / _cs8 / _cs16: the instantiations per input format
hipError_t launch_p0fft_cs8(const FftConvArgs &a, size_t lds, hipStream_t s);
hipError_t launch_p0fft_cs16(const FftConvArgs &a, size_t lds, hipStream_t s);
size_t p0fft_tap_lds();                                                               // bytes of the tap planes a workgroup carries (p0fft_cu8.hip)
constexpr int64_t kP0FftMaxKeep = 1 << 15;     // history + pending samples one workgroup recomputes for the next call: longer -> not fused
bool fftconv_agc_fusable(int log2n, int ntaps, uint32_t dbg);   // which filters have the epilogue (the radix-16 kernel)

// ---------------------------------------------------------------------------------------------
// k_interp: msresamp for r >= 1 -- arbitrary polyphase, then S half-band interpolators,
//           [post NCO], pack (interp.hip)
// ---------------------------------------------------------------------------------------------
constexpr int kInterpTile = 2048;  // final outputs per workgroup tile
constexpr int kArbWin = 14;        // taps per polyphase arm
struct InterpArgs {
    const cf2 *xbuf;          // [hist samples of earlier calls][n_in new samples]
    int32_t    hist;
    int64_t    n_in;
    uint64_t   phi0;          // phase of output k = 0 relative to new sample 0
    uint32_t   step;
    int64_t    n_arb;         // polyphase outputs of this call
    int64_t    n_emit;        // n_arb << S
    int64_t    n_tiles;
    int32_t    S;
    int32_t    m[kMaxS];      // run order: [0] = lowest rate
    int32_t    tap_off[kMaxS];
    int32_t    ext[kMaxS + 1];     // samples of level s rebuilt in front of a tile
    int32_t    lvl_off[kMaxS + 1]; // LDS offsets (cf32) of levels 0..S-1; [S] = total
    int32_t    in_cap;        // staged input window (cf32)
    int32_t    n_hb_taps;
    const float *hb_taps;     // filter-branch taps h[2t+1] of every stage
    const float *arb_table;   // [256][16]
    int32_t    pnco_mode;
    uint32_t   pnco_theta0, pnco_dtheta;
    const cf2 *nco_tab;
    int32_t    out_fmt;
    void      *out;
    cf2       *move_dst;      // as FirArgs: the next call's history, into the other buffer of the pair, by the last workgroup
    const cf2 *move_src;
    int64_t    move_n;
};
// fills ext / lvl_off / in_cap from S, m[], step; returns the input history (samples) k_interp needs
int make_interp_geometry(InterpArgs &a);
hipError_t launch_interp(const InterpArgs &a, int n_cu, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Output AGC, "digital" profile (agc.hip): per-chunk peak -> gain scan over chunks -> scale + pack
// ---------------------------------------------------------------------------------------------
struct AgcState {          // == iqgpu_agc_state
    int32_t  locked;
    float    peak_memory;
    float    gain;
    int32_t  reserved;
    double   last_strong;
    uint64_t seen;
};
struct AgcArgs {
    AgcGeom    geom;
    const cf2 *x;             // the call's output samples before the AGC (cf32)
    int64_t    n_out;
    unsigned long long *peak2;// [n_chunks] max |x|^2 per chunk, double bits (zeroed before k_agc_peak)
    float     *gain;          // [n_chunks]
    int32_t   *chunk_len;     // [n_chunks] outputs per chunk, written by k_agc_peak, read by k_agc_scan
    int32_t   *wg_last;       // [ceil(n_chunks / 256)] k_agc_classify: the last healthy chunk of each of its workgroups (-1: none)
    int32_t   *wg_pend;       // [ceil(n_chunks / 256)] ... and its last weak chunk with no healthy one in front of it inside the workgroup
    int64_t   *chunk_b;       // [n_chunks] k_agc_classify: the call's outputs in front of the chunk (agc_out_end(c - 1)), for the verdict's times
    AgcState  *state;
    float      target;
    double     rate;          // config->target_rate
    int32_t    clock_wall;    // 1: every chunk of the call sees time t_wall
    double     t_wall;
    int32_t    splits;        // workgroups per chunk
    int32_t    out_fmt;
    void      *out;
    const int32_t *run_if;    // not NULL: the three kernels do nothing unless *run_if != 0
    int32_t   *verify_flag;   // k_agc_classify's verdict: [0] set to 1 when the fused pass cannot stand (see agc.hip; 8 ints)
    unsigned long long *peak2_fallback;   // k_agc_classify zeroes this one too: the peak array of the conditional fallback launches
    int32_t    peak_approx;   // the fused kernel left float-accumulated peaks (k_front_mid): a chunk within a few ulp of a threshold
                              // is sent to the exact, unfused kernels instead of being classified
    int32_t   *verdict_host;  // not NULL: a word of pinned host memory that receives the verdict too (system-scope store): on the paths
                              // where the host waits for the call anyway it reads the verdict there and launches the fallback kernels
                              // only when it is set, instead of queueing four launches that return at once (agc_host.cpp)
};
hipError_t launch_agc(const AgcArgs &a, hipStream_t s);

// RMS profiles dx / local (liquid agc_crcf): chunk-parallel with warm-up, then verified / repaired (agc.hip)
struct AgcRmsArgs {
    const cf2 *x;             // the call's output samples before the AGC (cf32); x[-hist_valid .. -1] = the samples in front of them
    int64_t    n;
    int64_t    pos0;          // stream position (samples since the last reset) of x[0]: the chunk grid is the STREAM's
    int64_t    hist_valid;    // min(warm, pos0)
    float      alpha;         // loop bandwidth: AGC_DX_BANDWIDTH 1e-4, AGC_LOCAL_BANDWIDTH 1e-2
    AgcState  *state;         // gain = g, peak_memory = y2_prime
    int64_t    chunk, warm;   // outputs per lane; samples a speculative lane starts ahead
    int32_t    n_chunks;
    float     *st;            // [n_chunks][4]: {g, p} a chunk arrived with, {g, p} it left
    int32_t    out_fmt;
    void      *out;
};
void agc_rms_geometry(float alpha, int64_t pos0, int64_t n, int64_t *chunk, int64_t *warm, int32_t *n_chunks);
hipError_t launch_agc_rms(const AgcRmsArgs &a, hipStream_t s);   // peak, scan, apply
// after a fused launch of the front kernel: all chunks healthy at the unchanged gain -> state advanced, *verify_flag = 0;
// otherwise state untouched and *verify_flag = 1 (the caller has queued the unfused kernels behind it, run_if = verify_flag)
hipError_t launch_agc_verify(const AgcArgs &a, hipStream_t s);
bool front_s1_agc_fusable(const FrontArgs &a);

// dst[i] = src[i], i < n (cf32)
hipError_t launch_copy_cf(cf2 *dst, const cf2 *src, int64_t n, hipStream_t s);

// k_iq_probe: first 1024 pre-processed samples of a call for the I/Q optimiser (src/pipeline.c:468-476)
struct IqProbeArgs {
    const void *raw; int32_t in_fmt; float gain;
    int32_t dc_enable; float dc_c; const cd2 *dc_state;
    int32_t iq_enable; float iq_magp1, iq_phase;
    int32_t nco_mode; uint32_t nco_theta0, nco_dtheta; const cf2 *nco_tab;
    cf2 *out;                 // [1024]
};
hipError_t launch_iq_probe(const IqProbeArgs &a, hipStream_t s);

} // namespace iqgpu
