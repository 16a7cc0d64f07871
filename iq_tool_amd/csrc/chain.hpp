// chain.hpp -- internal to libiqgpu's host side: the chain object behind the opaque iqgpu_chain handle, the per-call plan and
// what the translation units of the C ABI share.  Not installed; include/iqgpu.h is the interface.
//   abi.cpp        library / lifecycle / state entry points, create-time design (design_chain)
//   plan.cpp       stream-position arithmetic (plan_call), per-call run geometry of the wave kernels (Call::plan_geometry)
//   process.cpp    one process() call: buffers, the stages in stream order, iqgpu_chain_process[_device]
//   agc_host.cpp   host side of the output AGC: chunk map, fused / unfused split, verifier + fallback launches
//   pipeline.cpp   iqgpu_chain_submit / _collect (pinned host buffers, three stages moved along by the host)
#pragma once
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/iqgpu.h"
#include "design.hpp"
#include "kernels.hpp"

using namespace iqgpu;

// ---- error reporting (abi.cpp) ----
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
const char *last_error_text();
std::string debug_value(const char *name);             // abi.cpp: the table iqgpu_debug_set keeps ("" when unset)
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(IQGPU_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

inline double monotonic_sec() // get_monotonic_time_sec, src/utils.c
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

inline size_t bytes_per_frame(int fmt)
{
    switch (fmt) { // get_bytes_per_sample, src/sample_convert.c:102-122 (complex formats)
    case IQGPU_FMT_CS8: case IQGPU_FMT_CU8: return 2;
    case IQGPU_FMT_CS16: case IQGPU_FMT_CU16: case IQGPU_FMT_SC16Q11: return 4;
    case IQGPU_FMT_CS24: return 6;
    case IQGPU_FMT_CS32: case IQGPU_FMT_CU32: case IQGPU_FMT_CF32: return 8;
    default: return 0;
    }
}
// ------------------------------------------------------------------------------------------------
// the chain object
// ------------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return IQGPU_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return fail(IQGPU_ENOMEM, "hipMalloc(%zu) failed", want); }
        cap = want;
        return IQGPU_OK;
    }
    // grow, keeping the first keep_bytes (synchronises the stream once per growth)
    int ensure_keep(size_t bytes, size_t keep_bytes, hipStream_t s)
    {
        if (bytes <= cap) return IQGPU_OK;
        DevBuf nb;
        int rc = nb.ensure(bytes); if (rc) return rc;
        if (p && keep_bytes) {
            if (hipMemcpyAsync(nb.p, p, keep_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
                nb.release(); return fail(IQGPU_EHIP, "device copy failed while growing a stream buffer");
            }
        }
        release();
        p = nb.p; cap = nb.cap;
        return IQGPU_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct iqgpu_chain {
    static constexpr int kPipeSlots = 8;
    iqgpu_chain_desc desc;
    int device = 0;
    float ratio = 1.0f;
    double target_rate = 0.0;
    bool resample = false;
    bool decim = false;          // resampler fused into the front kernel (r < 1, filter after it or none)
    bool late = false;           // resampler behind the front stage / pre filter: k_interp (r >= 1)
    ResamplePlan rp;
    FilterPlan fp;
    // operator constants
    bool dc = false; float dc_alpha = 0.0f, dc_c = 1.0f; double dc_logc = 0.0;
    float iq_mag = 0.0f, iq_phase = 0.0f;
    int nco_mode = 0, pnco_mode = 0; uint32_t nco_dtheta = 0;
    // geometry
    int S = 0, D = 1, TG = kTile;
    int warm_tiles = 0, hist_cap = 0, tiles_per_block = 128;
    bool auto_block = true;      // block_samples == 0: size the per-wave runs from the call and the CU count
    int n_cu = 256;
    uint32_t n_est = 0;
    int lvl_off[kMaxS + 2] = {0};
    int tap_off[kMaxS] = {0};
    int n_hb_taps = 0;
    // stream position (since the last reset)
    int rem = 0;                 // samples of the open group
    uint64_t phi = 0;            // phase of the next output relative to the next group
    uint32_t nco_theta = 0;      // pre-NCO phase of the next input sample
    uint32_t pnco_theta = 0;     // post-NCO phase of the next output sample
    uint64_t fpending = 0;       // FFT-mode filter input samples not yet emitted
    // device state
    hipStream_t own_stream = nullptr, stream = nullptr;
    cf2 *d_nco_tab = nullptr; float *d_arb = nullptr; float *d_hb = nullptr; cf2 *d_ftaps = nullptr;
    cf2 *d_hfreq = nullptr, *d_twiddle = nullptr; int fft_log2n = 0, fft_threads = 0;   // overlap-save path of FFT-kind filters
    cf2 *d_hist[2] = {nullptr, nullptr}; int hist_cur = 0;
    // S >= 2 without a dc blocker: k_cascade (stages 0 .. S-2) -> mid -> k_front_s1 (last stage + polyphase)
    bool cascade = false; int hist2_cap = 0, casc_warm = 1;
    cf2 *d_hist2[2] = {nullptr, nullptr}; int hist2_cur = 0;
    DevBuf mid;
    cd2 *d_dc_state = nullptr;
    void *d_sink = nullptr;      // diagnostic scratch of k_front_s1 (iqgpu_chain_debug_read_scratch)
    // run stealing in k_front_mid (kernels.hpp, FrontArgs::w_steal): one descriptor per wave of a launch; all exhausted between
    // launches (zeroed when the array is (re)allocated).  IQGPU_STEAL=1 turns it on, IQGPU_STEAL_MIN / _ROUNDS / _LANES / _STRIDE tune it.
    DevBuf steal_buf;
    // (off by default: measured neutral, profiles/r04_steal.md -- the launch tail it removes turned out to be free)
    bool steal = false; int steal_min = 6, steal_rounds = 6, steal_stride = 544, steal_lanes = 16;   // stride in 8-byte words: 4352 B
    DevBuf dc_agg, dc_carry;
    DevBuf fbuf[2]; int fcur = 0;
    // output AGC (digital profile)
    bool agc = false; float agc_target = 0.9f; int64_t agc_chunk = 16384;
    AgcState *d_agc_state = nullptr; AgcState agc_init{};
    float agc_rms_alpha = 0.0f;     // > 0: profile dx / local (liquid agc_crcf), AgcState.gain = g, .peak_memory = y2_prime
    DevBuf abuf, agc_peak, agc_gain, agc_peak_b;
    // dx / local: the chunk grid of the parallel scheme belongs to the stream -- its position since the last reset, and the last
    // `agc_rms_warm` samples of the AGC's input (what a chunk that begins early in the next call warms up on), agc.hip
    DevBuf agc_hist; int64_t agc_rms_warm = 0; uint64_t agc_rms_pos = 0;
    // agc_peak is all zero: what a fused front launch needs (k_agc_classify hands it back zeroed, agc_peak_b too; the unfused kernels do not)
    bool agc_peak_clean = false;
    // fused AGC of the locked phase (k_front_s1<.., AGC> + k_agc_verify): which chains qualify, the host's mirror of
    // "has the stream locked" (a closed form: the first chunk that starts after AGC_DIGITAL_LOCK_TIME of output), the
    // flag the verifier leaves for the fallback launches
    bool agc_fusable = false, agc_locked_host = false; uint64_t agc_seen_host = 0;
    int32_t *d_agc_flag = nullptr;
    // The verdict on the HOST (round 5).  On the paths where the host waits for a call's output anyway -- iqgpu_chain_process, and
    // submit / collect, whose host moves every batch from stage to stage -- the fallback kernels are not queued behind a fused
    // launch as four launches that normally return at once (~19 us of a 400 us step): k_agc_classify's verdict also lands in a
    // word of pinned host memory, the prepared fallback launches wait here, and whoever next needs the stream to be final
    // (the next call into the chain, the batch's D2H copy, reset / synchronize / get_agc_state) reads the word and launches them
    // only when it is set.  iqgpu_chain_process_device keeps the queued scheme: its caller owns the stream and its synchronisation.
    // h_agc_verdict[0]: -1 = a verdict is awaited, 0 / 1 = k_agc_classify's answer
    volatile int32_t *h_agc_verdict = nullptr; int32_t *d_agc_verdict = nullptr;   // the same word, host and device address
    bool defer_fallback = false;                // set around process_device_impl by the host-ordered entry points
    struct PendingVerdict { bool valid = false; bool filter = false; FrontArgs fb; FftConvArgs fc; AgcArgs ga; } pend;
    // ... and the AGC fused into the user filter's epilogue (k_fftconv16) where a filter stands between the resampler and the AGC:
    // the shipped -usb / -lsb presets
    bool agc_fusable_filter = false;
    // round 6: resampler -> post-resample overlap-save filter in ONE kernel (k_p0fft16, fftconv.hip): the chain's shape allows it,
    // with the window geometry it runs (fuse_win stream samples per block, fuse_vout outputs)
    bool fuse_filter = false; int fuse_win = 0, fuse_vout = 0; bool fft_keep_geometry = false;
    DevBuf ibuf[2]; int icur = 0;  // k_interp input: [ihist history][new samples]
    InterpArgs ia{};              // geometry of the r >= 1 path
    int ihist = 0;
    float *d_ihb = nullptr;
    DevBuf stage_in, stage_out;
    // pipelined host entry point (iqgpu_chain_submit / _collect): kPipeSlots batches in flight.  H2D copies, kernels
    // (the chain's stream) and D2H copies each have their own stream; a batch moves to the next stage inside a later
    // submit / collect, once the host has seen the previous stage finish (no device-side event waits: see pipe_advance)
    struct PipeSlot {
        hipEvent_t in_done = nullptr, k_done = nullptr, all_done = nullptr;
        DevBuf d_in, d_out; uint64_t ticket = 0; bool busy = false;
        size_t frames_in = 0, n_emit = 0; void *out = nullptr;
        float iq_mag = 0.0f, iq_phase = 0.0f;          // correction factors as of submit()
    };
    PipeSlot pipe[kPipeSlots];
    static constexpr int kCopyStreams = 4;                // small copies rotate over them, large ones keep to the first
    hipStream_t pipe_h2d[kCopyStreams] = {}, pipe_d2h[kCopyStreams] = {};
    bool pipe_ready = false;
    uint64_t pipe_seq = 0;        // tickets handed out
    uint64_t pipe_launched = 0;   // tickets whose kernels have been queued (<= pipe_seq)
    uint64_t pipe_copied = 0;     // tickets whose D2H copy has been queued (<= pipe_launched)
    // stream position behind the last ticket (valid while pipe_launched < pipe_seq)
    int pipe_rem = 0; uint64_t pipe_phi = 0, pipe_fpending = 0;
    bool iq_pinned = false; float iq_pin_mag = 0.0f, iq_pin_phase = 0.0f;   // factors of the batch being launched
    // I/Q optimiser probe: first 1024 pre-processed samples of a call (device -> pinned host), src/pipeline.c:468-476
    // (the optimiser runs on ITS OWN thread beside the stage thread: aux_mu guards the factors and the probe state;
    //  a block in flight or not yet read is never overwritten -- the optimiser takes at most two a second)
    std::mutex aux_mu;
    bool probe_on = false, probe_pending = false, probe_valid = false;
    cf2 *d_probe = nullptr; cf2 *h_probe = nullptr; hipEvent_t probe_done = nullptr;
    cf2 probe_last[1024];
    char front_kernel[48] = "";   // which front kernel the last call launched (iqgpu_chain_front_kernel)
    bool poisoned = false;        // a call failed after device state had been touched: reset() clears it
    bool force_generic = false;   // IQGPU_FORCE_GENERIC=1: always use the workgroup-tiled k_front
    uint32_t dbg = 0;             // kDbg* diagnostic switches, read from the environment once at create
    // weights of the runs of the first / second / third wave of a SIMD in k_front_mid (IQGPU_RUN_WEIGHTS=a,b,c; 0 = equal runs)
    int32_t run_wt[4] = {1300, 1000, 700, 0};
    // placement of the arms in the tap planes of k_front_mid / k_front_fat for this chain's step (front_tap_fold);
    // IQGPU_TAP_FOLD=0|1 overrides
    int tap_fold6 = 0, tap_fold8 = 0, tap_fold_env = -1;
    // profiling
    bool profiling = false;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> pending_events;
    std::vector<hipEvent_t> event_pool;
    iqgpu_profile prof{};
};

// ---- create-time design (abi.cpp): validation, ratio, operator constants, plans, launch geometry; touches no device ----
int design_chain(iqgpu_chain *c, const iqgpu_chain_desc *d);

// ------------------------------------------------------------------------------------------------
// stream-position arithmetic (closed forms; SPEC B.6)
// ------------------------------------------------------------------------------------------------
struct CallPlan {
    int64_t n_groups = 0;      // complete 2^S groups this call
    int64_t n_res = 0;         // front-kernel outputs this call (resampled, or one per input)
    int64_t n_emit = 0;        // frames written to the caller
    int64_t n_x = 0;           // r >= 1 path: samples entering k_interp (after the pre filter)
    int64_t n_arb = 0;         //              polyphase outputs; n_emit = n_arb << S
    uint64_t phi_next = 0;
    int rem_next = 0;
    uint64_t fpending_next = 0;
};

// the stream position a call starts from: what plan_call reads of the chain's host-side state
struct StreamPos { int rem = 0; uint64_t phi = 0; uint64_t fpending = 0; };
CallPlan plan_call_at(const iqgpu_chain *c, const StreamPos &at, size_t frames_in);      // plan.cpp
CallPlan plan_call(const iqgpu_chain *c, size_t frames_in);                               // ... from the chain's own position

// ---- profiling (process.cpp): HIP events around every launch while profiling is on ----
hipEvent_t get_event(iqgpu_chain *c);
void drain_events(iqgpu_chain *c);
struct KernelTimer {
    iqgpu_chain *c; int kind; hipEvent_t a = nullptr, b = nullptr;
    KernelTimer(iqgpu_chain *c_, int kind_) : c(c_), kind(kind_)
    {
        if (c->profiling) { a = get_event(c); b = get_event(c); (void)hipEventRecord(a, c->stream); }
    }
    ~KernelTimer()
    {
        if (c->profiling) { (void)hipEventRecord(b, c->stream); c->pending_events.push_back({kind, {a, b}}); }
    }
};

// ------------------------------------------------------------------------------------------------
// one process() call
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// one process() call: per-call geometry, then the stages in stream order
//   [dc carries] -> front (k_front | k_front_s1 | k_cascade + k_front_s1) -> [filter] -> [k_interp] -> [agc]
// ------------------------------------------------------------------------------------------------
struct Call {
    iqgpu_chain *c;
    const void *d_raw_in; size_t frames_in; void *d_out;
    CallPlan p;
    bool filt; size_t L1; uint64_t fpending0;
    void *fin_out; int fin_fmt;              // where the LAST stage writes (d_out, or the AGC's cf32 buffer)
    int64_t total_tiles; int tpb, n_blocks;  // geometry of the workgroup-tiled k_front
    bool casc, fast_s0, fast_s1;             // which front path runs
    bool fat = false;                        // fast_s1 as k_front_fat (front_fat.hip): 8 waves per CU, 1024-frame tiles
    bool mid = false;                        // ... or as k_front_mid (front_mid.hip): 12 waves per CU, 768-frame tiles
    bool p0 = false;                         // fast_s0 as k_front_p0 (front_p0.hip): output-major steps, taps kept in registers
    bool fusef = false;                      // ... or, with a post-resample filter behind it, resampler AND filter as k_p0fft16 (no front launch at all)
    bool s2 = false;                         // casc with both stages fused into k_front_s2 (front_s2.hip); cplan is then the LAST stage's plan
    int64_t s2_in_tiles = 0;                 //   ... and this the number of 512-frame input tiles of the call
    int wtile, casc_K, rem_k;
    float iq_mag = 0.0f, iq_phase = 0.0f;    // the correction factors this call applies (snapshot under aux_mu)
    bool agc_fused = false;                  // this call: gain applied in the front kernel -- or, with a filter behind it, in the filter's epilogue -- and verified behind it
    bool front_fused() const { return agc_fused && !filt; }
    FrontArgs cplan;                         // run geometry of the wave-autonomous kernel that sees the raw input
    cf2 *fcur = nullptr, *icur = nullptr;    // filter-input / k_interp-input buffers of this call

    // run rule of the wave-autonomous kernels: one run per resident wave when auto (wave slots of ONE round of
    // workgroups), else fixed-length runs from block_samples
    int64_t wave_slots(int waves) const { return (int64_t)c->n_cu * waves; }
    int fixed_tpw() const
    {
        if (c->auto_block) return 0;
        int64_t t = (int64_t)c->tiles_per_block * kTile / (16 * kWTile);
        if (t < 1) t = 1;
        if (t > (1 << 30)) t = 1 << 30;
        return (int)t;
    }
    void copy_plan(FrontArgs &dst) const
    {
        dst.w_total_tiles = cplan.w_total_tiles;
        dst.w_warm_tiles = cplan.w_warm_tiles; dst.w_edge_tpw = cplan.w_edge_tpw;
        dst.w_n_stream = cplan.w_n_stream; dst.w_run_q = cplan.w_run_q; dst.w_run_r = cplan.w_run_r;
        dst.w_edge_ta = cplan.w_edge_ta; dst.w_edge_tb = cplan.w_edge_tb;
        dst.w_n_edge1 = cplan.w_n_edge1; dst.w_n_edge = cplan.w_n_edge;
        dst.w_wpw = cplan.w_wpw; dst.w_wsum = cplan.w_wsum;
        for (int i = 0; i < 4; ++i) dst.w_wt[i] = cplan.w_wt[i];
        dst.p0_k_a = cplan.p0_k_a; dst.p0_k_b = cplan.p0_k_b; dst.p0_f_max = cplan.p0_f_max;
    }
    int raw_aligned() const { return (((uintptr_t)d_raw_in) & 15u) == 0 ? 1 : 0; }
    // the per-chunk peaks a fused front launch accumulates into start from zero: k_agc_classify zeroes what it has read (agc_peak
    // and agc_peak_b), so only the first fused call behind an unfused one (or behind a reallocation) pays for a fill
    hipError_t clean_agc_peaks()
    {
        if (c->agc_peak_clean) return hipSuccess;
        const hipError_t e = hipMemsetAsync(c->agc_peak.p, 0, c->agc_peak.cap, c->stream);
        if (e == hipSuccess) c->agc_peak_clean = true;
        return e;
    }

    void plan_geometry();
    DcGeom dc_geom() const;
    AgcGeom agc_geom() const;
    int stage_dc_carries();
    int prepare_buffers();
    int stage_front();
    int stage_filter();
    int stage_late_resampler();
    int stage_agc();
    int stage_agc_verify_and_fallback(const FrontArgs &spec);
    int stage_agc_verify_and_fallback_filter(const FftConvArgs &spec);
    AgcArgs agc_args() const;
};

// ---- entry points shared across translation units ----
int process_device_impl(iqgpu_chain *c, const void *d_raw_in, size_t frames_in, void *d_out, size_t out_capacity_bytes,
                        size_t *frames_out);                                              // process.cpp
size_t agc_unfused_head(const iqgpu_chain *c, size_t frames_in, bool *locks);             // agc_host.cpp
// behind a fused launch whose fallback waits for the verdict on the host: waits for the word, launches the fallback when it is set
// (*ran = true then).  No-op without a pending verdict.
int agc_resolve_pending(iqgpu_chain *c, bool *ran = nullptr);                             // agc_host.cpp
int pipe_advance(iqgpu_chain *c, uint64_t upto);   // pipeline.cpp: queues the kernels of every submitted batch up to ticket `upto`
int pipe_drain(iqgpu_chain *c, uint64_t upto);     // ... and their D2H copies
