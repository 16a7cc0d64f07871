// iqgpu_api.cpp -- the C ABI of libiqgpu (include/iqgpu.h): chain lifecycle, stream-position
// bookkeeping and kernel launches.  No CPU compute path exists here: every entry point that moves
// samples launches the gfx950 kernels of kernels.hip or fails.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <new>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/iqgpu.h"
#include "design.hpp"
#include "kernels.hpp"

using namespace iqgpu;

// ------------------------------------------------------------------------------------------------
// error reporting
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static double monotonic_sec() // get_monotonic_time_sec, src/utils.c
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(IQGPU_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

static size_t bytes_per_frame(int fmt)
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
    DevBuf dc_agg, dc_carry;
    DevBuf fbuf[2]; int fcur = 0;
    // output AGC (digital profile)
    bool agc = false; float agc_target = 0.9f; int64_t agc_chunk = 16384;
    AgcState *d_agc_state = nullptr; AgcState agc_init{};
    float agc_rms_alpha = 0.0f;     // > 0: profile dx / local (liquid agc_crcf), AgcState.gain = g, .peak_memory = y2_prime
    DevBuf abuf, agc_peak, agc_gain, agc_peak_b;
    bool agc_peak_clean = false;  // agc_peak is all zero: what a fused front launch needs (k_agc_verify leaves it so; the unfused kernels do not)
    // fused AGC of the locked phase (k_front_s1<.., AGC> + k_agc_verify): which chains qualify, the host's mirror of
    // "has the stream locked" (a closed form: the first chunk that starts after AGC_DIGITAL_LOCK_TIME of output), the
    // flag the verifier leaves for the fallback launches
    bool agc_fusable = false, agc_locked_host = false; uint64_t agc_seen_host = 0;
    int32_t *d_agc_flag = nullptr;
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
    int pipe_rem = 0; uint64_t pipe_phi = 0, pipe_fpending = 0;   // stream position behind the last ticket (valid while pipe_launched < pipe_seq)
    bool iq_pinned = false; float iq_pin_mag = 0.0f, iq_pin_phase = 0.0f;   // factors of the batch being launched
    // I/Q optimiser probe: first 1024 pre-processed samples of a call (device -> pinned host), src/pipeline.c:468-476
    // (the optimiser runs on ITS OWN thread beside the stage thread: aux_mu guards the factors and the probe state;
    //  a block in flight or not yet read is never overwritten -- the optimiser takes at most two a second)
    std::mutex aux_mu;
    bool probe_on = false, probe_pending = false, probe_valid = false;
    cf2 *d_probe = nullptr; cf2 *h_probe = nullptr; hipEvent_t probe_done = nullptr;
    cf2 probe_last[1024];
    bool poisoned = false;        // a call failed after device state had been touched: reset() clears it
    bool force_generic = false;   // IQGPU_FORCE_GENERIC=1: always use the workgroup-tiled k_front
    uint32_t dbg = 0;             // kDbg* diagnostic switches, read from the environment once at create
    int32_t run_wt[4] = {1300, 1000, 700, 0};      // weights of the runs of the first / second / third wave of a SIMD in k_front_mid (IQGPU_RUN_WEIGHTS=a,b,c; 0 = equal runs)
    int tap_fold6 = 0, tap_fold8 = 0, tap_fold_env = -1;    // placement of the arms in the tap planes of k_front_mid / k_front_fat for this chain's step (front_tap_fold); IQGPU_TAP_FOLD=0|1 overrides
    // profiling
    bool profiling = false;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> pending_events;
    std::vector<hipEvent_t> event_pool;
    iqgpu_profile prof{};
};

static int pipe_advance(iqgpu_chain *c, uint64_t upto);   // queues the kernels of every submitted batch up to ticket `upto`
static int pipe_drain(iqgpu_chain *c, uint64_t upto);     // ... and their D2H copies

// ------------------------------------------------------------------------------------------------
// library-level
// ------------------------------------------------------------------------------------------------
extern "C" int iqgpu_abi_version(void) { return IQGPU_ABI_VERSION; }
extern "C" const char *iqgpu_last_error(void) { return g_err; }

extern "C" int iqgpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" size_t iqgpu_get_bytes_per_sample(int format)
{
    switch (format) { // the reference also sizes its real scalar formats (include/common_types.h:33-37)
    case 1: case 2: return 1;           // U8, S8
    case 3: case 4: return 2;           // U16, S16
    case 5: case 6: case 7: return 4;   // U32, S32, F32
    default: return bytes_per_frame(format);
    }
}

extern "C" void iqgpu_chain_desc_init(iqgpu_chain_desc *d)
{
    if (!d) return;
    memset(d, 0, sizeof(*d));
    d->in_format = IQGPU_FMT_CS16;
    d->out_format = IQGPU_FMT_CS16;
    d->gain = 1.0f;              // src/main.c:145
    d->no_resample = 0;
    d->block_samples = 0;        // auto: one contiguous run per resident wavefront
}

// ------------------------------------------------------------------------------------------------
// create / destroy
// ------------------------------------------------------------------------------------------------
static void free_device_state(iqgpu_chain *c)
{
    (void)hipSetDevice(c->device);
    if (c->d_nco_tab) (void)hipFree(c->d_nco_tab);
    if (c->d_arb) (void)hipFree(c->d_arb);
    if (c->d_hb) (void)hipFree(c->d_hb);
    if (c->d_ftaps) (void)hipFree(c->d_ftaps);
    if (c->d_hfreq) (void)hipFree(c->d_hfreq);
    if (c->d_ihb) (void)hipFree(c->d_ihb);
    if (c->d_agc_state) (void)hipFree(c->d_agc_state);
    c->abuf.release(); c->agc_peak.release(); c->agc_gain.release(); c->agc_peak_b.release();
    if (c->d_agc_flag) (void)hipFree(c->d_agc_flag);
    if (c->d_twiddle) (void)hipFree(c->d_twiddle);
    for (int i = 0; i < 2; ++i) if (c->d_hist[i]) (void)hipFree(c->d_hist[i]);
    for (int i = 0; i < 2; ++i) if (c->d_hist2[i]) (void)hipFree(c->d_hist2[i]);
    c->mid.release();
    if (c->d_dc_state) (void)hipFree(c->d_dc_state);
    if (c->d_sink) (void)hipFree(c->d_sink);
    c->dc_agg.release(); c->dc_carry.release();
    c->fbuf[0].release(); c->fbuf[1].release();
    c->ibuf[0].release(); c->ibuf[1].release();
    c->stage_in.release(); c->stage_out.release();
    if (c->d_probe) (void)hipFree(c->d_probe);
    if (c->h_probe) (void)hipHostFree(c->h_probe);
    if (c->probe_done) (void)hipEventDestroy(c->probe_done);
    for (auto &ps : c->pipe) {
        ps.d_in.release(); ps.d_out.release();
        if (ps.in_done) (void)hipEventDestroy(ps.in_done);
        if (ps.k_done) (void)hipEventDestroy(ps.k_done);
        if (ps.all_done) (void)hipEventDestroy(ps.all_done);
    }
    for (hipStream_t st : c->pipe_h2d) if (st) (void)hipStreamDestroy(st);
    for (hipStream_t st : c->pipe_d2h) if (st) (void)hipStreamDestroy(st);
    for (auto &pe : c->pending_events) { (void)hipEventDestroy(pe.second.first); (void)hipEventDestroy(pe.second.second); }
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
}

template <typename T>
static int upload(T **dst, const T *src, size_t n)
{
    if (n == 0) { *dst = nullptr; return IQGPU_OK; }
    if (hipMalloc((void **)dst, n * sizeof(T)) != hipSuccess) return fail(IQGPU_ENOMEM, "hipMalloc(%zu) failed", n * sizeof(T));
    HIP_TRY(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return IQGPU_OK;
}

// Everything create() derives on the host: validation (the reference's fatal paths), ratio,
// operator constants, resampler / filter plans and launch geometry.  Touches no device.
static int design_chain(iqgpu_chain *c, const iqgpu_chain_desc *d)
{
    if (!bytes_per_frame(d->in_format)) return fail(IQGPU_EFORMAT, "Unhandled input format: %d", d->in_format);
    if (!bytes_per_frame(d->out_format)) return fail(IQGPU_EFORMAT, "Unhandled output format: %d", d->out_format);
    if (!(d->input_rate_hz > 0.0) && !(d->resample_ratio > 0.0f) && !d->no_resample)
        return fail(IQGPU_EINVAL, "input_rate_hz must be positive");
    c->desc = *d;
    c->device = d->device_ordinal;
    {   // every IQGPU_* switch is read HERE, once per chain: process() and the launch functions never touch the environment
        const char *fg = getenv("IQGPU_FORCE_GENERIC"); c->force_generic = fg && fg[0] == '1';
        c->dbg = (getenv("IQGPU_NO_FAST") ? kDbgNoFast : 0u) | (getenv("IQGPU_AGC_NOFUSE") ? kDbgAgcNoFuse : 0u) |
                 (getenv("IQGPU_NO_RAW0") ? kDbgNoRaw0 : 0u) | (getenv("IQGPU_NO_KT") ? kDbgNoKT : 0u) |
                 (getenv("IQGPU_FFT_NO_R16") ? kDbgFftNoR16 : 0u) | (getenv("IQGPU_NO_FAT") ? kDbgNoFat : 0u) |
                 (getenv("IQGPU_FORCE_FAT") ? kDbgForceFat : 0u) | (getenv("IQGPU_FAT") ? kDbgUseFat : 0u) | (getenv("IQGPU_MID8") ? kDbgMid8 : 0u);
        if (const char *tf = getenv("IQGPU_TAP_FOLD")) c->tap_fold_env = atoi(tf) != 0 ? 1 : 0;
        if (const char *rw = getenv("IQGPU_RUN_WEIGHTS")) { int x = 0, y = 0, z = 0; if (sscanf(rw, "%d,%d,%d", &x, &y, &z) == 3 && x >= 0 && y >= 0 && z >= 0) { c->run_wt[0] = x; c->run_wt[1] = y; c->run_wt[2] = z; } }
    }

    // ---- ratio (src/setup.c:91-122) ----
    const double in_rate = d->input_rate_hz > 0.0 ? d->input_rate_hz : 1.0;
    c->target_rate = d->no_resample ? in_rate : d->target_rate_hz;
    if (d->resample_ratio > 0.0f && !d->no_resample) {
        c->ratio = d->resample_ratio;
        if (!(d->target_rate_hz > 0.0)) c->target_rate = in_rate * (double)c->ratio;
    } else {
        c->ratio = (float)(c->target_rate / in_rate);
    }
    if (!std::isfinite(c->ratio) || c->ratio < 0.001f || c->ratio > 1000.0f) return fail(IQGPU_ERATIO, "Calculated resampling ratio (%.6f) is invalid or outside acceptable range.", (double)c->ratio);
    c->resample = !d->no_resample;

    // ---- dc blocker (src/dc_block.c:32) ----
    if (d->dc_block_enable) {
        c->dc = true;
        c->dc_alpha = (float)(2.0 * 3.14159265358979323846 * 10.0f / in_rate);
        if (!(c->dc_alpha > 0.0f)) return fail(IQGPU_EINVAL, "DC Block: Calculated normalized alpha is invalid.");
        const float a1 = -1.0f + c->dc_alpha;       // liquid: a = {1, -1 + alpha}
        c->dc_c = -a1;
        c->dc_logc = std::log((double)c->dc_c);
    }
    c->iq_mag = d->iq_mag; c->iq_phase = d->iq_phase;

    // ---- frequency shift (src/frequency_shift.c:24-81) ----
    if (d->shift_after_resample && std::fabs(d->shift_hz) < 1e-9) return fail(IQGPU_ESHIFT, "Option --shift-after-resample was used, but no effective frequency shift was requested or calculated.");
    if (std::fabs(d->shift_hz) >= 1e-9) {
        const double rate = d->shift_after_resample ? c->target_rate : in_rate;
        if (std::fabs(d->shift_hz) > 5.0 * rate) return fail(IQGPU_ESHIFT, "Requested frequency shift %.2f Hz exceeds sanity limit for the rate of %.1f Hz.", d->shift_hz, rate);
        const float w = (float)(2.0 * 3.14159265358979323846 * std::fabs(d->shift_hz) / rate);
        c->nco_dtheta = nco_constrain(w);
        const int mode = d->shift_hz >= 0 ? +1 : -1;
        if (d->shift_after_resample) c->pnco_mode = mode; else c->nco_mode = mode;
    }

    // ---- resampler (src/resampler.c:20-34, 60 dB include/constants.h:137) ----
    std::string err;
    if (c->resample) {
        if (!make_resample_plan(c->ratio, 60.0f, c->rp, err)) return fail(IQGPU_ERATIO, "%s", err.c_str());
        if (c->rp.S >= kMaxS) return fail(IQGPU_ERATIO, "too many half-band stages");
        c->tap_fold6 = c->tap_fold_env >= 0 ? c->tap_fold_env : front_tap_fold(c->rp.step, 6);
        c->tap_fold8 = c->tap_fold_env >= 0 ? c->tap_fold_env : front_tap_fold(c->rp.step, 8);
    }

    // ---- user filter (src/filter.c:138-393) ----
    {
        int rc = make_filter_plan(*d, in_rate, c->target_rate, c->fp, err);
        if (rc != IQGPU_OK) return fail(rc, "%s", err.c_str());
    }
    // r < 1 decimates inside the front kernel (filter, if any, behind it); otherwise the filter
    // comes first (src/filter.c:43-92) and the resampler runs last, in k_interp
    c->late = c->resample && (c->rp.interp || (c->fp.enabled && !c->fp.post_resample));
    c->decim = c->resample && !c->late;
    c->S = c->decim ? c->rp.S : 0;
    c->D = 1 << c->S;
    c->TG = kTile >> c->S;
    if (c->decim && c->S >= 2 && !c->force_generic) {
        int mm[kMaxS];
        for (int i = 0; i < c->S; ++i) mm[i] = c->rp.stages[(size_t)i].m;
        c->cascade = cascade_supported(mm, c->S);
        if (c->cascade) {
            uint64_t h = 0;                                   // input history the first S-1 stages need
            for (int k = c->S - 2; k >= 0; --k) h = 2 * h + 4u * (unsigned)mm[k];
            c->casc_warm = (int)((h + kWTile - 1) / kWTile); if (c->casc_warm < 1) c->casc_warm = 1;
            c->hist2_cap = kTile + 2;                         // last stage: 66 samples of history, one warm-up tile
        }
    }
    // ---- output AGC (src/agc.c:21-83, src/config.c:306-330) ----
    if (d->agc_enable) {
        if (d->agc_profile != IQGPU_AGC_DIGITAL && d->agc_profile != IQGPU_AGC_DX && d->agc_profile != IQGPU_AGC_LOCAL)
            return fail(IQGPU_EINVAL, "Invalid AGC profile %d. Must be 'dx', 'local', or 'digital'.", d->agc_profile);   // src/config.c:318
        if (d->agc_target != 0.0f && (d->agc_target <= 0.0f || d->agc_target > 1.0f))
            return fail(IQGPU_EINVAL, "Invalid AGC target level %.2f. Must be between 0.0 and 1.0.", (double)d->agc_target);
        if (d->agc_clock != IQGPU_AGC_CLOCK_SAMPLES && d->agc_clock != IQGPU_AGC_CLOCK_WALL) return fail(IQGPU_EINVAL, "agc_clock must be IQGPU_AGC_CLOCK_SAMPLES or IQGPU_AGC_CLOCK_WALL");
        c->agc = true;
        // dx / local: liquid agc_crcf with AGC_DX_BANDWIDTH / AGC_LOCAL_BANDWIDTH (src/agc.c:45-57, constants.h:169,175);
        // the target level does not reach the loop (agc_crcf_set_gain(1.0f) behind set_signal_level, agc.c:56-59)
        c->agc_rms_alpha = d->agc_profile == IQGPU_AGC_DX ? 1e-4f : d->agc_profile == IQGPU_AGC_LOCAL ? 1e-2f : 0.0f;
        c->agc_target = d->agc_target > 0.0f ? d->agc_target : 0.9f;      // AGC_DIGITAL_PEAK_TARGET
        c->agc_chunk = d->agc_chunk_frames ? (int64_t)d->agc_chunk_frames : 16384;   // PIPELINE_CHUNK_BASE_SAMPLES
        // k_agc_scan adds the output lengths of 64 chunks in 32 bits
        if ((double)c->agc_chunk * (double)(c->ratio > 1.0f ? c->ratio : 1.0f) * 64.0 >= 2147483648.0)
            return fail(IQGPU_EINVAL, "agc_chunk_frames %lld is too large for this ratio (64 chunks must stay below 2^31 output frames)", (long long)c->agc_chunk);
    }
    if (c->late) {
        InterpArgs &ia = c->ia;
        ia.S = c->rp.S; ia.step = c->rp.step;
        int off = 0;
        for (int s2 = 0; s2 < ia.S; ++s2) {      // run order of the interpolators: lowest rate first
            const HalfbandStage &st = c->rp.stages[(size_t)(ia.S - 1 - s2)];
            ia.m[s2] = st.m; ia.tap_off[s2] = off; off += 2 * st.m;
        }
        ia.n_hb_taps = off;
        c->ihist = (make_interp_geometry(ia) + 15) & ~15;
    }

    // ---- geometry ----
    c->auto_block = d->block_samples == 0;
    size_t block = d->block_samples ? d->block_samples : 262144;
    if (block % kTile != 0 || block == 0) return fail(IQGPU_EINVAL, "block_samples must be a multiple of %d", kTile);
    c->tiles_per_block = (int)(block / kTile);
    if (c->decim) {
        c->warm_tiles = (int)((c->rp.history_in + kTile - 1) / kTile);
        if (c->warm_tiles < 1) c->warm_tiles = 1;
        c->hist_cap = c->warm_tiles * kTile + c->D;
        int off = 0;
        for (int i = 0; i <= c->S; ++i) {
            const int H = (i < c->S) ? 4 * c->rp.stages[(size_t)i].m : kArbHist;
            c->lvl_off[i] = off;
            off += H + (kTile >> i);
            off = (off + 1) & ~1;
        }
        c->lvl_off[c->S + 1] = off;
        c->n_est = (uint32_t)((((uint64_t)c->TG) << 24) / c->rp.step);
    } else {
        c->warm_tiles = 0; c->hist_cap = 0;
        c->lvl_off[0] = 0; c->lvl_off[1] = 0;
    }

    return IQGPU_OK;
}

extern "C" int iqgpu_chain_create(const iqgpu_chain_desc *d, iqgpu_chain **out)
{
    if (!d || !out) return fail(IQGPU_EINVAL, "iqgpu_chain_create: NULL argument");
    *out = nullptr;
    iqgpu_chain *c = new (std::nothrow) iqgpu_chain();
    if (!c) return fail(IQGPU_ENOMEM, "out of host memory");
    { const int drc = design_chain(c, d); if (drc != IQGPU_OK) { delete c; return drc; } }

    // ---- device ----
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { int rc = fail(IQGPU_ENODEV, "no HIP device available"); delete c; return rc; }
    if (c->device < 0 || c->device >= ndev) { int rc = fail(IQGPU_ENODEV, "device_ordinal %d out of range (%d devices)", c->device, ndev); delete c; return rc; }
    int rc = IQGPU_OK;
#define CREATE_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = fail(IQGPU_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); goto bad; } } while (0)
#define CREATE_RC(expr) do { rc = (expr); if (rc != IQGPU_OK) goto bad; } while (0)
    {
        CREATE_TRY(hipSetDevice(c->device));
        {
            int ncu = 0;
            if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && ncu > 0) c->n_cu = ncu;
        }
        CREATE_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
        c->stream = c->own_stream;
        std::vector<cfloat> tab(1024);
        nco_fill_sincos(tab.data());
        CREATE_RC(upload(&c->d_nco_tab, (const cf2 *)tab.data(), 1024));
        if (c->late) {
            CREATE_RC(upload(&c->d_arb, c->rp.arb_table.data(), c->rp.arb_table.size()));
            std::vector<float> hb;
            for (int s2 = 0; s2 < c->ia.S; ++s2)
                for (float v : c->rp.stages[(size_t)(c->ia.S - 1 - s2)].branch) hb.push_back(v);
            if (hb.empty()) hb.push_back(0.0f);
            CREATE_RC(upload(&c->d_ihb, hb.data(), hb.size()));
            for (int i = 0; i < 2; ++i) {
                CREATE_RC(c->ibuf[i].ensure(((size_t)c->ihist + 1) * sizeof(cf2)));
                CREATE_TRY(hipMemset(c->ibuf[i].p, 0, c->ibuf[i].cap));
            }
        }
        if (c->decim) {
            CREATE_RC(upload(&c->d_arb, c->rp.arb_table.data(), c->rp.arb_table.size()));
            std::vector<float> hb;
            for (int i = 0; i < c->S; ++i) {
                c->tap_off[i] = (int)hb.size();
                for (float v : c->rp.stages[(size_t)i].branch) hb.push_back(0.5f * v);   // per-stage gain 1/2 (exact)
            }
            c->n_hb_taps = (int)hb.size();
            if (hb.empty()) hb.push_back(0.0f);
            CREATE_RC(upload(&c->d_hb, hb.data(), hb.size()));
            for (int i = 0; i < 2; ++i) {
                CREATE_TRY(hipMalloc((void **)&c->d_hist[i], (size_t)c->hist_cap * sizeof(cf2)));
                CREATE_TRY(hipMemset(c->d_hist[i], 0, (size_t)c->hist_cap * sizeof(cf2)));
                if (c->cascade) {
                    CREATE_TRY(hipMalloc((void **)&c->d_hist2[i], (size_t)c->hist2_cap * sizeof(cf2)));
                    CREATE_TRY(hipMemset(c->d_hist2[i], 0, (size_t)c->hist2_cap * sizeof(cf2)));
                }
            }
        }
        if (c->agc) {
            CREATE_TRY(hipMalloc((void **)&c->d_agc_state, sizeof(AgcState)));
            c->agc_init = AgcState{0, c->agc_rms_alpha > 0.0f ? 1.0f : 0.05f, 1.0f, 0, c->desc.agc_clock == IQGPU_AGC_CLOCK_WALL ? monotonic_sec() : 0.0, 0};
            CREATE_TRY(hipMemcpy(c->d_agc_state, &c->agc_init, sizeof(AgcState), hipMemcpyHostToDevice));
            {   // [0] verdict of the verifier, [1] ratchet seen, [2] weak chunk seen, [3] last healthy chunk (agc.hip)
                const int32_t init[4] = {0, 0, 0, -1};
                CREATE_TRY(hipMalloc((void **)&c->d_agc_flag, sizeof(init)));
                CREATE_TRY(hipMemcpy(c->d_agc_flag, init, sizeof(init), hipMemcpyHostToDevice));
            }
            // the fused path exists for the specialised front kernel: the shipped cs16 NRSC-5 preset shape
            FrontArgs fa{};
            fa.dbg = c->dbg;
            fa.S = c->S; fa.in_fmt = c->desc.in_format; fa.out_fmt = c->desc.out_format; fa.gain = c->desc.gain;
            fa.iq_enable = c->desc.iq_correct_enable ? 1 : 0; fa.dc_enable = c->dc ? 1 : 0;
            fa.nco_mode = c->nco_mode; fa.pnco_mode = c->pnco_mode; fa.agc_chunk_frames = c->agc_chunk; fa.agc_shift = c->S;
            if (c->cascade) { fa.S = 1; fa.in_fmt = IQGPU_FMT_CF32; }      // k_cascade in front: the last stage sees cf32, one half-band
            c->agc_fusable = c->agc_rms_alpha == 0.0f && c->decim && !c->late && !c->force_generic && !c->fp.enabled &&
                             (c->cascade || c->S == 0 || (c->S == 1 && c->rp.stages[0].m == 10)) && front_s1_agc_fusable(fa);
        }
        if (c->fp.enabled) CREATE_RC(upload(&c->d_ftaps, (const cf2 *)c->fp.taps.data(), c->fp.taps.size()));
        // overlap-save path: every FFT-kind filter, and FIR-kind ones long enough that two transforms
        // per window beat the direct form (the two are the same linear convolution, SPEC B.3)
        const size_t Lt = c->fp.taps.size();
        if (c->fp.enabled && !c->force_generic && Lt >= 2 && 2 * (Lt - 1) <= (size_t)kMaxFftN &&
            (c->fp.block > 0 || Lt >= (size_t)kFftMinTaps)) {
            // N = 4 (L-1) rounded up to a power of two in [256, 4096], larger (up to 16384, in place in LDS) only
            // when the taps need it; measured with k_fftconv16 on config 3 (1025 taps): N 2048 0.31 ms, 4096 0.22, 8192 0.24,
            // 16384 0.29; on config 4 (4097 taps): N 8192 0.146 ms, 16384 0.144
            int lg = 8;
            while ((size_t)(1 << lg) < 4 * (Lt - 1) && (1 << lg) < 4096) ++lg;
            while ((size_t)(1 << lg) < 2 * (Lt - 1)) ++lg;
            if (const char *e = getenv("IQGPU_FFT_LOG2N")) { const int v = atoi(e); if (v >= 1 && (1 << v) <= kMaxFftN && (size_t)(1 << v) >= 2 * (Lt - 1)) lg = v; }
            if (const char *e = getenv("IQGPU_FFT_THREADS")) c->fft_threads = atoi(e);
            const int N = 1 << lg;
            c->fft_log2n = lg;
            // H = FFT_N(taps) / N and the twiddle table, in double on the host (once per chain)
            std::vector<double> ct((size_t)N), st((size_t)N);
            const double w0 = -2.0 * 3.14159265358979323846 / (double)N;
            for (int k = 0; k < N; ++k) { ct[(size_t)k] = std::cos(w0 * k); st[(size_t)k] = std::sin(w0 * k); }
            std::vector<cf2> tw((size_t)N), hf((size_t)N);
            for (int k = 0; k < N; ++k) tw[(size_t)k] = cf2{(float)ct[(size_t)k], (float)st[(size_t)k]};
            for (int p = 0; p < N; ++p) {
                double hr = 0.0, hi = 0.0;
                unsigned idx = 0;                                  // p k mod N
                for (size_t k = 0; k < Lt; ++k) {
                    const double cr = ct[idx], ci = st[idx];
                    hr += c->fp.taps[k].re * cr - c->fp.taps[k].im * ci;
                    hi += c->fp.taps[k].re * ci + c->fp.taps[k].im * cr;
                    idx = (idx + (unsigned)p) & (unsigned)(N - 1);
                }
                hf[(size_t)p] = cf2{(float)(hr / N), (float)(hi / N)};
            }
            CREATE_RC(upload(&c->d_twiddle, tw.data(), tw.size()));
            CREATE_RC(upload(&c->d_hfreq, hf.data(), hf.size()));
        }
        CREATE_TRY(hipMalloc((void **)&c->d_dc_state, sizeof(cd2)));
        CREATE_TRY(hipMemset(c->d_dc_state, 0, sizeof(cd2)));
        CREATE_TRY(hipMalloc(&c->d_sink, 64 * 1024));
        CREATE_TRY(hipMemset(c->d_sink, 0, 64 * 1024));
        if (c->fp.enabled) {
            // the filter-input buffer starts as ntaps-1 zeros of history
            const size_t h = c->fp.taps.size() - 1;
            for (int i = 0; i < 2; ++i) {
                CREATE_RC(c->fbuf[i].ensure((h + 1) * sizeof(cf2)));
                CREATE_TRY(hipMemset(c->fbuf[i].p, 0, c->fbuf[i].cap));
            }
        }
        CREATE_TRY(hipDeviceSynchronize());
    }
    *out = c;
    return IQGPU_OK;
bad:
    free_device_state(c);
    delete c;
    return rc;
#undef CREATE_TRY
#undef CREATE_RC
}

extern "C" void iqgpu_chain_destroy(iqgpu_chain *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)pipe_advance(c, c->pipe_seq);            // batches submitted and never collected still run to completion
    (void)pipe_drain(c, c->pipe_seq);
    (void)hipStreamSynchronize(c->stream);
    for (hipStream_t st : c->pipe_d2h) if (st) (void)hipStreamSynchronize(st);
    free_device_state(c);
    delete c;
}

static void fill_info(const iqgpu_chain *c, iqgpu_chain_info *info);

extern "C" int iqgpu_design_probe(const iqgpu_chain_desc *d, iqgpu_chain_info *info,
                                  float *filter_taps_re_im, size_t cap_taps,
                                  float *hb_taps, size_t cap_hb, float *arb_proto, size_t cap_arb)
{
    if (!d || !info) return fail(IQGPU_EINVAL, "iqgpu_design_probe: NULL argument");
    iqgpu_chain *c = new (std::nothrow) iqgpu_chain();
    if (!c) return fail(IQGPU_ENOMEM, "out of host memory");
    const int rc = design_chain(c, d);
    if (rc == IQGPU_OK) {
        fill_info(c, info);
        if (filter_taps_re_im) {
            const size_t n = c->fp.taps.size() < cap_taps ? c->fp.taps.size() : cap_taps;
            memcpy(filter_taps_re_im, c->fp.taps.data(), n * sizeof(cfloat));
        }
        if (hb_taps) {
            size_t o = 0;
            for (int i = 0; i < c->rp.S; ++i)
                for (float v : c->rp.stages[(size_t)i].proto) { if (o < cap_hb) hb_taps[o] = v; ++o; }
        }
        if (arb_proto && c->resample) {
            const size_t n = c->rp.arb_proto.size() < cap_arb ? c->rp.arb_proto.size() : cap_arb;
            memcpy(arb_proto, c->rp.arb_proto.data(), n * sizeof(float));
        }
    }
    delete c;
    return rc;
}

extern "C" int iqgpu_chain_get_info(const iqgpu_chain *c, iqgpu_chain_info *info)
{
    if (!c || !info) return fail(IQGPU_EINVAL, "iqgpu_chain_get_info: NULL argument");
    fill_info(c, info);
    return IQGPU_OK;
}

static void fill_info(const iqgpu_chain *c, iqgpu_chain_info *info)
{
    memset(info, 0, sizeof(*info));
    info->ratio = c->ratio;
    info->interp = c->rp.interp ? 1 : 0;
    info->num_halfband_stages = c->resample ? c->rp.S : 0;
    for (int i = 0; i < info->num_halfband_stages && i < 16; ++i) info->stage_m[i] = c->rp.stages[(size_t)i].m;
    info->rate_arb = c->rp.rate_arb;
    info->arb_step = c->rp.step;
    info->nco_dtheta = c->nco_dtheta;
    info->dc_alpha = c->dc_alpha;
    info->filter_post_resample = c->fp.post_resample ? 1 : 0;
    info->filter_impl = c->fp.impl;
    info->filter_ntaps = (uint32_t)c->fp.taps.size();
    info->filter_block = c->fp.block;
    info->history_samples = (uint32_t)c->hist_cap;
}

extern "C" int iqgpu_chain_get_filter_taps(const iqgpu_chain *c, float *re_im, size_t cap_taps)
{
    if (!c) return fail(IQGPU_EINVAL, "NULL chain");
    const size_t n = c->fp.taps.size();
    if (re_im) memcpy(re_im, c->fp.taps.data(), (n < cap_taps ? n : cap_taps) * sizeof(cfloat));
    return (int)n;
}

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

static CallPlan plan_call_at(const iqgpu_chain *c, const StreamPos &at, size_t frames_in)
{
    CallPlan p;
    if (c->late) {
        p.n_res = (int64_t)frames_in;
        p.n_x = p.n_res;
        if (c->fp.enabled && c->fp.block) { // src/filter.c:503-525
            const uint64_t total = at.fpending + (uint64_t)p.n_res;
            p.n_x = (int64_t)((total / c->fp.block) * c->fp.block);
            p.fpending_next = total - (uint64_t)p.n_x;
        }
        const uint64_t span = (uint64_t)p.n_x << 24;
        const uint64_t step = c->rp.step;
        if (span > at.phi) { p.n_arb = (int64_t)((span - at.phi + step - 1) / step); p.phi_next = at.phi + (uint64_t)p.n_arb * step - span; }
        else { p.n_arb = 0; p.phi_next = at.phi - span; }
        p.n_emit = p.n_arb << c->ia.S;
        return p;
    }
    if (c->decim) {
        const uint64_t avail = (uint64_t)at.rem + frames_in;
        p.n_groups = (int64_t)(avail >> c->S);
        p.rem_next = (int)(avail & (uint64_t)(c->D - 1));
        const uint64_t span = (uint64_t)p.n_groups << 24;
        const uint64_t step = c->rp.step;
        if (span > at.phi) { p.n_res = (int64_t)((span - at.phi + step - 1) / step); p.phi_next = at.phi + (uint64_t)p.n_res * step - span; }
        else { p.n_res = 0; p.phi_next = at.phi - span; }
    } else {
        p.n_res = (int64_t)frames_in;
    }
    if (c->fp.enabled && c->fp.block) { // src/filter.c:503-525
        const uint64_t total = at.fpending + (uint64_t)p.n_res;
        p.n_emit = (int64_t)((total / c->fp.block) * c->fp.block);
        p.fpending_next = total - (uint64_t)p.n_emit;
    } else {
        p.n_emit = p.n_res;
    }
    return p;
}

static CallPlan plan_call(const iqgpu_chain *c, size_t frames_in)
{
    StreamPos at; at.rem = c->rem; at.phi = c->phi; at.fpending = c->fpending;
    return plan_call_at(c, at, frames_in);
}

extern "C" size_t iqgpu_chain_next_out_frames(const iqgpu_chain *c, size_t frames_in)
{
    if (!c) return 0;
    if (c->pipe_launched < c->pipe_seq) {      // behind the batches submitted and not yet launched
        StreamPos at; at.rem = c->pipe_rem; at.phi = c->pipe_phi; at.fpending = c->pipe_fpending;
        return (size_t)plan_call_at(c, at, frames_in).n_emit;
    }
    return (size_t)plan_call(c, frames_in).n_emit;
}

// frames a FRESH chain of this description emits for frames_in input frames in one stream: the same closed form
// the calls use (resampler law either way round, FFT-block quantisation in front of or behind the resampler),
// without a device -- what a sharding writer needs to place shard outputs (BASELINE configs[4])
extern "C" int iqgpu_design_out_frames(const iqgpu_chain_desc *d, size_t frames_in, size_t *frames_out)
{
    if (!d || !frames_out) return fail(IQGPU_EINVAL, "iqgpu_design_out_frames: NULL argument");
    *frames_out = 0;
    iqgpu_chain *c = new (std::nothrow) iqgpu_chain();
    if (!c) return fail(IQGPU_ENOMEM, "out of host memory");
    const int rc = design_chain(c, d);
    if (rc == IQGPU_OK) *frames_out = (size_t)plan_call(c, frames_in).n_emit;
    delete c;
    return rc;
}

extern "C" size_t iqgpu_chain_max_out_frames(const iqgpu_chain *c, size_t frames_in)
{
    if (!c) return 0;
    // src/pipeline.c:246-258, generalised from PIPELINE_CHUNK_BASE_SAMPLES to frames_in
    double r = c->resample ? (double)c->ratio : 1.0;
    if (r < 1.0) r = 1.0;
    size_t cap = (size_t)std::ceil((double)frames_in * r) + 128;
    if (cap < frames_in) cap = frames_in;
    if (c->fp.enabled && c->fp.block) cap += c->fp.block;
    if (c->late) cap += ((size_t)2 << c->ia.S) + (c->fp.block ? (size_t)std::ceil((double)c->fp.block * r) : 0);
    return cap;
}

// ------------------------------------------------------------------------------------------------
// profiling helpers
// ------------------------------------------------------------------------------------------------
static hipEvent_t get_event(iqgpu_chain *c)
{
    if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
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

static void drain_events(iqgpu_chain *c)
{
    for (auto &pe : c->pending_events) {
        float ms = 0.0f;
        (void)hipEventSynchronize(pe.second.second);
        if (hipEventElapsedTime(&ms, pe.second.first, pe.second.second) == hipSuccess) {
            c->prof.ms[pe.first] += (double)ms;
            c->prof.launches[pe.first] += 1;
        }
        c->event_pool.push_back(pe.second.first);
        c->event_pool.push_back(pe.second.second);
    }
    c->pending_events.clear();
}

// ------------------------------------------------------------------------------------------------
// process
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// one process() call: per-call geometry, then the stages in stream order
//   [dc carries] -> front (k_front | k_front_s1 | k_cascade + k_front_s1) -> [filter] -> [k_interp] -> [agc]
// ------------------------------------------------------------------------------------------------
namespace {

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
    int wtile, casc_K, rem_k;
    float iq_mag = 0.0f, iq_phase = 0.0f;    // the correction factors this call applies (snapshot under aux_mu)
    bool agc_fused = false;                  // this call: gain applied in the front kernel, verified behind it
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
    }
    int raw_aligned() const { return (((uintptr_t)d_raw_in) & 15u) == 0 ? 1 : 0; }
    // the per-chunk peaks a fused front launch accumulates into start from zero: k_agc_verify zeroes what it has read, so only
    // the first fused call behind an unfused one (or behind a reallocation) pays for a fill
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
    AgcArgs agc_args() const;
};

void Call::plan_geometry()
{
    const int64_t span_samples = (int64_t)c->rem + (int64_t)frames_in;
    total_tiles = (span_samples + kTile - 1) / kTile;
    // blocks of the workgroup-tiled k_front: with block_samples = 0 sized from the call -- about eight blocks
    // per CU, at least 16 tiles each when a block has to re-run a warm-up tile (decimating chains), any
    // size for pointwise chains
    tpb = c->tiles_per_block;
    if (c->auto_block) {
        int64_t t = (total_tiles + (int64_t)c->n_cu * 8 - 1) / ((int64_t)c->n_cu * 8);
        const int64_t t_min = c->decim ? 16 : 1;
        if (t < t_min) t = t_min;
        if (t > 128) t = 128;
        tpb = (int)t;
    }
    n_blocks = (int)((total_tiles + tpb - 1) / tpb);
    if (n_blocks < 1) n_blocks = 1;

    // run geometry of the wave-autonomous kernels (needed by the dc carries as well)
    //   S >= 2: k_cascade (stages 0 .. S-2) + k_front_s1 (last stage);  S == 1: k_front_s1;  S == 0: its S0 variant
    casc = c->cascade && !c->force_generic;
    fast_s0 = c->decim && c->S == 0 && !c->force_generic;          // polyphase only, 256-frame tiles
    fast_s1 = fast_s0 || (c->decim && c->S == 1 && c->rp.stages[0].m == 10 && !c->force_generic);
    wtile = fast_s0 ? 256 : kWTile;
    casc_K = c->S - 1;
    rem_k = casc ? (c->rem & ((1 << casc_K) - 1)) : c->rem;
    cplan = FrontArgs{};
    cplan.dbg = c->dbg;
    if (casc || fast_s1) {
        cplan.frames_in = (int64_t)frames_in; cplan.rem0 = rem_k; cplan.hist_cap = c->hist_cap;
        cplan.in_fmt = c->desc.in_format; cplan.out_fmt = (casc || filt) ? (int)IQGPU_FMT_CF32 : fin_fmt;
        cplan.raw_aligned = raw_aligned();
        if (casc) {
            cplan.casc_K = casc_K;
            for (int k = 0; k < casc_K; ++k) cplan.m[k] = c->rp.stages[(size_t)k].m;
            cplan.casc_wave_lds = (int)cascade_wave_lds(cplan);
        }
        cplan.agc_fused = agc_fused ? 1 : 0; cplan.agc_shift = c->S; cplan.agc_chunk_frames = c->agc_chunk;
        cplan.S = c->S; cplan.gain = c->desc.gain; cplan.iq_enable = c->desc.iq_correct_enable ? 1 : 0;
        cplan.dc_enable = c->dc ? 1 : 0; cplan.nco_mode = c->nco_mode;
        cplan.pnco_mode = (!filt && !c->late) ? c->pnco_mode : 0;
        cplan.step = c->rp.step;
        // the preset shape on a call long enough to give every one of the 8 x CUs fat waves a run of tiles: k_front_fat
        // (shorter calls keep k_front_s1's 16 x CUs waves of 512-frame tiles: what counts for them is latency; same bytes either way)
        const bool fat_ok = !casc && !fast_s0 && front_fat_shape(cplan) &&
              ((c->dbg & kDbgForceFat) || (int64_t)frames_in >= (int64_t)kFatMinTilesPerWave * kFatTile * wave_slots(front_fat_waves()));
        const int mid_nl = (!casc && !fast_s0) ? front_mid_nl(cplan) : 0;
        const bool mid_ok = mid_nl != 0 &&
              ((c->dbg & kDbgForceFat) || (int64_t)frames_in >= (int64_t)kFatMinTilesPerWave * front_mid_tile(mid_nl) * wave_slots(front_mid_waves()));
        // (measured on one box, 2^28 frames: k_front_s1 0.437 ms, k_front_fat 0.404, k_front_mid 0.381: the 12-wave kernel is the
        //  default; IQGPU_FAT=1 selects the 8-wave one where its step class applies)
        fat = fat_ok && ((c->dbg & kDbgUseFat) || !mid_ok);
        mid = mid_ok && !fat;
        if (fat) wtile = kFatTile;
        if (mid) wtile = front_mid_tile(mid_nl);
        const int mid_align = (mid && mid_nl == 6) ? 2 : 1;      // 768-frame tiles: edge runs of two = three 512-frame tiles
        cplan.w_total_tiles = ((int64_t)rem_k + (int64_t)frames_in + wtile - 1) / wtile;
        int warm = casc ? c->casc_warm : (int)((c->rp.history_in + wtile - 1) / wtile);
        if (warm < 1) warm = 1;
        int ftpw = fixed_tpw();
        if (ftpw > 1 && (fat || mid)) { ftpw = ftpw * kWTile / wtile; if (ftpw < 1) ftpw = 1; }
        plan_front_s1(cplan, wave_slots(casc ? cascade_waves(cplan) : fat ? front_fat_waves() : mid ? front_mid_waves() : front_s1_waves(cplan)),
                      ftpw, warm, mid_align, wtile, mid_align, mid ? kMidLead : 0);
        // k_front_mid: the three waves of a SIMD get runs in proportion to the speed their age buys them (kernels.hpp, weight_runs)
        if (mid && ftpw == 0 && cplan.w_n_edge <= front_mid_max_edge_waves() && c->run_wt[0] > 0) weight_runs(cplan, front_mid_waves(), c->run_wt);
        if (mid && cplan.w_n_edge > front_mid_max_edge_waves()) {
            // (an unaligned buffer, a call that is all edges: k_front_mid keeps LDS for a handful of edge waves only)
            mid = false; wtile = kWTile;
            cplan.w_total_tiles = ((int64_t)rem_k + (int64_t)frames_in + wtile - 1) / wtile;
            warm = (int)((c->rp.history_in + wtile - 1) / wtile); if (warm < 1) warm = 1;
            plan_front_s1(cplan, wave_slots(front_s1_waves(cplan)), fixed_tpw(), warm, 1, wtile);
        }
    }
}

// where every independent piece of the front kernel starts (the dc blocker needs its state there)
DcGeom Call::dc_geom() const
{
    DcGeom dg{};
    dg.frames_in = (int64_t)frames_in;
    if (casc || fast_s1) {
        dg.mode = 1;
        dg.n_edge1 = cplan.w_n_edge1; dg.n_stream = cplan.w_n_stream;
        dg.edge_tpw = cplan.w_edge_tpw; dg.run_q = cplan.w_run_q; dg.run_r = cplan.w_run_r; dg.ta = cplan.w_edge_ta; dg.tb = cplan.w_edge_tb;
        dg.warm = cplan.w_warm_tiles; dg.rem0 = rem_k; dg.tile = wtile;
        dg.n_seg = (int)(cplan.w_n_edge + dg.n_stream);
        if (dg.n_seg < 1) dg.n_seg = 1;
    } else {
        dg.mode = 0; dg.n_seg = n_blocks;
        dg.seg_first = ((int64_t)tpb - c->warm_tiles) * kTile - c->rem;
        dg.seg_len = (int64_t)tpb * kTile;
    }
    return dg;
}

// state of the dc blocker at the start of every independent piece of the front kernel
int Call::stage_dc_carries()
{
    const DcGeom dg = dc_geom();
    DcPrefixArgs pa{};
    pa.raw = d_raw_in; pa.in_fmt = c->desc.in_format; pa.gain = c->desc.gain;
    pa.raw_aligned = raw_aligned();
    pa.c = c->dc_c; pa.logc = c->dc_logc; pa.geom = dg; pa.agg = (cf2 *)c->dc_agg.p;
    { KernelTimer kt(c, IQGPU_K_DC_PREFIX); HIP_TRY(launch_dc_prefix(pa, c->stream)); }
    DcScanArgs sa{};
    sa.agg = (const cf2 *)c->dc_agg.p; sa.carry = (cd2 *)c->dc_carry.p; sa.state = c->d_dc_state;
    sa.geom = dg; sa.logc = c->dc_logc;
    { KernelTimer kt(c, IQGPU_K_DC_SCAN); HIP_TRY(launch_dc_scan(sa, c->stream)); }
    return IQGPU_OK;
}

// the cf32 buffers between stages: [L-1 history][pending][new] in front of the filter, [ihist][new] in front of k_interp
int Call::prepare_buffers()
{
    if (filt) {
        const size_t need = (L1 + (size_t)c->fpending + (size_t)p.n_res + 1) * sizeof(cf2);
        int rc = c->fbuf[c->fcur].ensure_keep(need, (L1 + (size_t)c->fpending) * sizeof(cf2), c->stream);
        if (rc) return rc;
        fcur = (cf2 *)c->fbuf[c->fcur].p;
    }
    if (c->late) {
        int rc = c->ibuf[c->icur].ensure_keep(((size_t)c->ihist + (size_t)p.n_x + 1) * sizeof(cf2), (size_t)c->ihist * sizeof(cf2), c->stream);
        if (rc) return rc;
        icur = (cf2 *)c->ibuf[c->icur].p;
        rc = c->ibuf[c->icur ^ 1].ensure(((size_t)c->ihist + 1) * sizeof(cf2)); if (rc) return rc;
    }
    // the buffers the later stages write: sized here, before the first launch touches the stream state
    if (filt) {
        int rc = c->fbuf[c->fcur ^ 1].ensure((L1 + (size_t)p.fpending_next + 1) * sizeof(cf2)); if (rc) return rc;
    }
    if (c->dc) {
        const DcGeom dg = dc_geom();
        int rc = c->dc_agg.ensure((size_t)dg.n_seg * sizeof(cf2)); if (rc) return rc;
        rc = c->dc_carry.ensure((size_t)dg.n_seg * sizeof(cd2)); if (rc) return rc;
    }
    if (casc) {
        const int64_t n_mid = ((int64_t)rem_k + (int64_t)frames_in) >> casc_K;
        int rc = c->mid.ensure(((size_t)n_mid + 8) * sizeof(cf2)); if (rc) return rc;
    }
    if (c->agc && c->agc_rms_alpha > 0.0f) {
        int64_t chunk, warm; int32_t n_chunks;
        agc_rms_geometry(c->agc_rms_alpha, p.n_emit, &chunk, &warm, &n_chunks);
        int rc = c->agc_gain.ensure((size_t)(n_chunks > 0 ? n_chunks : 1) * 4 * sizeof(float)); if (rc) return rc;
    } else if (c->agc) {
        const AgcGeom g = agc_geom();
        if (agc_out_end(g, g.n_chunks - 1) != p.n_emit) return fail(IQGPU_EINVAL, "internal: AGC chunk map disagrees with the call plan");
        { const void *was = c->agc_peak.p;
          int rc0 = c->agc_peak.ensure((size_t)g.n_chunks * sizeof(unsigned long long)); if (rc0) return rc0;
          if (c->agc_peak.p != was) c->agc_peak_clean = false; }
        int rc = c->agc_gain.ensure((size_t)g.n_chunks * (sizeof(float) + sizeof(int32_t))); if (rc) return rc;
        if (agc_fused) {   // what the fallback launches need, should the verifier reject the fused pass
            rc = c->agc_peak_b.ensure((size_t)g.n_chunks * sizeof(unsigned long long)); if (rc) return rc;
            rc = c->abuf.ensure(((size_t)p.n_emit + 1) * sizeof(cf2)); if (rc) return rc;
        }
    }
    return IQGPU_OK;
}

int Call::stage_front()
{
    FrontArgs a{};
    a.dbg = c->dbg;
    a.raw = d_raw_in;
    a.hist_in = c->d_hist[c->hist_cur]; a.hist_out = c->d_hist[c->hist_cur ^ 1];
    a.frames_in = (int64_t)frames_in; a.hist_cap = c->hist_cap; a.rem0 = c->rem;
    a.in_fmt = c->desc.in_format; a.gain = c->desc.gain;
    a.raw_aligned = raw_aligned();
    a.dc_enable = c->dc ? 1 : 0;
    if (c->dc) {
        a.dc_c = c->dc_c; a.dc_a = 1.0f - c->dc_c; a.dc_logc = c->dc_logc;
        for (int k = 0; k < 6; ++k) a.dc_cpow[k] = (float)std::exp((double)(4 << k) * c->dc_logc);
        a.dc_cpow[6] = (float)std::exp(256.0 * c->dc_logc);
        a.dc_cpow[7] = (float)std::exp(1024.0 * c->dc_logc);
        a.dc_carry = (const cd2 *)c->dc_carry.p;
    }
    a.iq_enable = c->desc.iq_correct_enable ? 1 : 0;
    a.iq_magp1 = 1.0f + iq_mag; a.iq_phase = iq_phase;
    a.nco_mode = c->nco_mode;
    a.nco_dtheta = c->nco_dtheta;
    // phase of i_rel = 0, i.e. rem samples before the first new sample
    a.nco_theta0 = c->nco_theta - (uint32_t)c->rem * c->nco_dtheta;
    a.nco_tab = c->d_nco_tab;
    a.mode = c->decim ? 1 : 0;
    a.S = c->S;
    for (int i = 0; i < c->S; ++i) { a.m[i] = c->rp.stages[(size_t)i].m; a.tap_off[i] = c->tap_off[i]; }
    for (int i = 0; i <= c->S + 1; ++i) a.lvl_off[i] = c->lvl_off[i];
    a.n_hb_taps = c->n_hb_taps; a.hb_taps = c->d_hb; a.arb_table = c->d_arb;
    a.step = c->rp.step; a.n_est = c->n_est; a.phi0 = c->phi;
    a.n_groups = p.n_groups; a.n_out = p.n_res;
    a.total_tiles = total_tiles; a.tiles_per_block = tpb; a.warm_tiles = c->warm_tiles;
    a.pnco_theta0 = c->pnco_theta; a.pnco_dtheta = c->nco_dtheta;
    const bool nco_in_front = !filt && !c->late;     // otherwise the post NCO runs in the last stage
    a.pnco_mode = nco_in_front ? c->pnco_mode : 0;
    if (filt)         { a.out_fmt = IQGPU_FMT_CF32; a.out = fcur + L1 + c->fpending; }
    else if (c->late) { a.out_fmt = IQGPU_FMT_CF32; a.out = icur + c->ihist; }
    else              { a.out_fmt = fin_fmt; a.out = fin_out; }
    a.sink = c->d_sink;

    if (casc) {
        // ---- stages 0 .. S-2: raw -> mid (cf32 at rate / 2^K) ----
        const int K = casc_K;
        const int rem_1 = c->rem >> K;
        const int64_t n_mid = ((int64_t)rem_k + (int64_t)frames_in) >> K;
        FrontArgs a1 = a;
        a1.rem0 = rem_k;
        a1.nco_theta0 = c->nco_theta - (uint32_t)rem_k * c->nco_dtheta;
        a1.casc_K = K;
        for (int k = 0; k < K; ++k) {
            const std::vector<float> &br = c->rp.stages[(size_t)k].branch;
            for (size_t q = 0; q < 12; ++q) a1.casc_taps[k][q] = q < br.size() ? 0.5f * br[q] : 0.0f;
        }
        a1.casc_out = (cf2 *)c->mid.p; a1.casc_n_out = n_mid;
        a1.casc_wave_lds = (int)cascade_wave_lds(a1);
        a1.out_fmt = IQGPU_FMT_CF32; a1.pnco_mode = 0;
        copy_plan(a1);
        { KernelTimer kt(c, IQGPU_K_CASCADE); HIP_TRY(launch_cascade(a1, c->stream)); }
        // ---- last stage + polyphase: a one-stage chain on the intermediate stream ----
        if (n_mid > 0) {
            FrontArgs a2{};
            a2.dbg = c->dbg;
            a2.raw = c->mid.p; a2.hist_in = c->d_hist2[c->hist2_cur]; a2.hist_out = c->d_hist2[c->hist2_cur ^ 1];
            a2.frames_in = n_mid; a2.hist_cap = c->hist2_cap; a2.rem0 = rem_1;
            a2.in_fmt = IQGPU_FMT_CF32; a2.gain = 1.0f; a2.raw_aligned = 1;
            a2.nco_tab = c->d_nco_tab;
            a2.mode = 1; a2.S = 1; a2.m[0] = c->rp.stages[(size_t)K].m;
            a2.arb_table = c->d_arb; a2.step = c->rp.step; a2.phi0 = c->phi;
            a2.n_groups = p.n_groups; a2.n_out = p.n_res;
            a2.pnco_mode = a.pnco_mode; a2.pnco_theta0 = a.pnco_theta0; a2.pnco_dtheta = a.pnco_dtheta;
            a2.out_fmt = a.out_fmt; a2.out = a.out;
            a2.w_total_tiles = ((int64_t)rem_1 + n_mid + kWTile - 1) / kWTile;
            a2.agc_fused = agc_fused ? 1 : 0;
            plan_front_s1(a2, wave_slots(front_s1_waves(a2)), fixed_tpw(), 1, 1);
            for (int q = 0; q < 20; ++q) a2.hb0[q] = 0.5f * c->rp.stages[(size_t)K].branch[(size_t)q];
            a2.sink = c->d_sink;
            if (agc_fused) {
                a2.agc_fused = 1; a2.agc_state = c->d_agc_state; a2.agc_peak2 = (unsigned long long *)c->agc_peak.p;
                a2.agc_chunk_frames = c->agc_chunk; a2.agc_shift = c->S; a2.agc_rem = c->rem;
                HIP_TRY(clean_agc_peaks());
            }
            { KernelTimer kt(c, IQGPU_K_FRONT); HIP_TRY(launch_front_s1(a2, c->stream)); }
            if (agc_fused) { const int rc = stage_agc_verify_and_fallback(a2); if (rc) return rc; }
            c->hist2_cur ^= 1;
        }
    } else if (fast_s1) {
        // wave-autonomous kernel: one half-band stage (m = 10), or none
        copy_plan(a);
        if (!fast_s0) for (int q = 0; q < 20; ++q) a.hb0[q] = 0.5f * c->rp.stages[0].branch[(size_t)q];
        if (agc_fused) {
            a.agc_fused = 1; a.agc_state = c->d_agc_state; a.agc_peak2 = (unsigned long long *)c->agc_peak.p;
            a.agc_chunk_frames = c->agc_chunk; a.agc_shift = c->S; a.agc_rem = c->rem;
            HIP_TRY(clean_agc_peaks());
        }
        a.tap_fold = (uint32_t)(fat ? c->tap_fold8 : mid ? (front_mid_nl(a) == 8 ? c->tap_fold8 : c->tap_fold6) : 0);
        { KernelTimer kt(c, IQGPU_K_FRONT); HIP_TRY(fat ? launch_front_fat(a, c->stream) : mid ? launch_front_mid(a, c->stream) : launch_front_s1(a, c->stream)); }
        if (agc_fused) { const int rc = stage_agc_verify_and_fallback(a); if (rc) return rc; }
    } else {
        KernelTimer kt(c, IQGPU_K_FRONT);
        HIP_TRY(launch_front(a, n_blocks, c->stream));
    }
    if (c->decim) c->hist_cur ^= 1;
    return IQGPU_OK;
}

int Call::stage_filter()
{
    FirArgs fa{};
    fa.fbuf = fcur; fa.taps = c->d_ftaps; fa.ntaps = (int)c->fp.taps.size(); fa.is_complex = c->fp.is_complex ? 1 : 0;
    const int64_t n_filt = c->late ? p.n_x : p.n_emit;
    fa.n_emit = n_filt;
    fa.pnco_mode = c->late ? 0 : c->pnco_mode; fa.pnco_theta0 = c->pnco_theta; fa.pnco_dtheta = c->nco_dtheta; fa.nco_tab = c->d_nco_tab;
    if (c->late) { fa.out_fmt = IQGPU_FMT_CF32; fa.out = icur + c->ihist; }
    else         { fa.out_fmt = fin_fmt; fa.out = fin_out; }
    if (c->d_hfreq) {
        FftConvArgs ca{};
        ca.dbg = c->dbg;
        ca.fbuf = fcur; ca.fbuf_len = (int64_t)(L1 + (size_t)c->fpending + (size_t)p.n_res);
        ca.hfreq = c->d_hfreq; ca.twiddle = c->d_twiddle; ca.ntaps = fa.ntaps;
        ca.log2n = c->fft_log2n; ca.threads = c->fft_threads; ca.n_emit = n_filt;
        ca.pnco_mode = fa.pnco_mode; ca.pnco_theta0 = fa.pnco_theta0; ca.pnco_dtheta = fa.pnco_dtheta; ca.nco_tab = fa.nco_tab;
        ca.out_fmt = fa.out_fmt; ca.out = fa.out;
        KernelTimer kt(c, IQGPU_K_FILTER);
        HIP_TRY(launch_fftconv(ca, c->stream));
    } else {
        KernelTimer kt(c, IQGPU_K_FILTER);
        HIP_TRY(launch_fir(fa, c->stream));
    }
    // next call's buffer front: history (L-1) + still-pending samples
    const size_t keep = L1 + (size_t)p.fpending_next;
    { KernelTimer kt(c, IQGPU_K_MOVE);
      HIP_TRY(launch_copy_cf((cf2 *)c->fbuf[c->fcur ^ 1].p, fcur + n_filt, (int64_t)keep, c->stream)); }
    c->fcur ^= 1;
    c->fpending = p.fpending_next;
    return IQGPU_OK;
}

// r >= 1: the resampler behind the front stage / pre filter
int Call::stage_late_resampler()
{
    InterpArgs ia = c->ia;
    ia.xbuf = icur; ia.hist = c->ihist; ia.n_in = p.n_x;
    ia.phi0 = c->phi; ia.n_arb = p.n_arb; ia.n_emit = p.n_emit;
    ia.n_tiles = (p.n_emit + kInterpTile - 1) / kInterpTile;
    ia.hb_taps = c->d_ihb; ia.arb_table = c->d_arb;
    ia.pnco_mode = c->pnco_mode; ia.pnco_theta0 = c->pnco_theta; ia.pnco_dtheta = c->nco_dtheta; ia.nco_tab = c->d_nco_tab;
    ia.out_fmt = fin_fmt; ia.out = fin_out;
    { KernelTimer kt(c, IQGPU_K_FRONT); HIP_TRY(launch_interp(ia, c->n_cu, c->stream)); }
    { KernelTimer kt(c, IQGPU_K_MOVE);
      HIP_TRY(launch_copy_cf((cf2 *)c->ibuf[c->icur ^ 1].p, icur + p.n_x, (int64_t)c->ihist, c->stream)); }
    c->icur ^= 1;
    return IQGPU_OK;
}

// output AGC: agc_apply per reference chunk (src/post_processor.c:55-57)
AgcGeom Call::agc_geom() const
{
    AgcGeom g{};
    g.frames_in = (int64_t)frames_in; g.chunk_frames = c->agc_chunk;
    g.n_chunks = (int)(((int64_t)frames_in + c->agc_chunk - 1) / c->agc_chunk);
    g.mode = c->late ? 2 : (c->decim ? 1 : 0);
    g.rem = c->rem; g.S = c->late ? c->ia.S : c->S; g.phi = c->phi; g.step = c->rp.step;
    g.block = (filt && c->fp.block) ? c->fp.block : 0; g.fpending = fpending0;
    return g;
}

AgcArgs Call::agc_args() const
{
    AgcArgs ga{};
    ga.geom = agc_geom();
    const AgcGeom &g = ga.geom;
    ga.x = (const cf2 *)c->abuf.p; ga.n_out = p.n_emit;
    ga.peak2 = (unsigned long long *)c->agc_peak.p; ga.gain = (float *)c->agc_gain.p;
    ga.chunk_len = (int32_t *)((float *)c->agc_gain.p + g.n_chunks); ga.state = c->d_agc_state;
    ga.target = c->agc_target; ga.rate = c->target_rate;
    ga.clock_wall = c->desc.agc_clock == IQGPU_AGC_CLOCK_WALL ? 1 : 0;
    ga.t_wall = ga.clock_wall ? monotonic_sec() : 0.0;
    const int64_t avg = p.n_emit / g.n_chunks + 1;
    int64_t splits = (avg + 16383) / 16384; if (splits > 1024) splits = 1024;
    ga.splits = (int)splits;
    ga.out_fmt = c->desc.out_format; ga.out = d_out;
    return ga;
}

int Call::stage_agc()
{
    if (c->agc_rms_alpha > 0.0f) {
        AgcRmsArgs ra{};
        ra.x = (const cf2 *)c->abuf.p; ra.n = p.n_emit; ra.alpha = c->agc_rms_alpha; ra.state = c->d_agc_state;
        agc_rms_geometry(ra.alpha, ra.n, &ra.chunk, &ra.warm, &ra.n_chunks);
        int rc = c->agc_gain.ensure((size_t)(ra.n_chunks > 0 ? ra.n_chunks : 1) * 4 * sizeof(float)); if (rc) return rc;
        ra.st = (float *)c->agc_gain.p;
        ra.out_fmt = c->desc.out_format; ra.out = d_out;
        KernelTimer kt(c, IQGPU_K_AGC);
        HIP_TRY(launch_agc_rms(ra, c->stream));
        return IQGPU_OK;
    }
    const AgcArgs ga = agc_args();
    KernelTimer kt(c, IQGPU_K_AGC);
    c->agc_peak_clean = false;                       // (k_agc_peak leaves its maxima in agc_peak)
    HIP_TRY(launch_agc(ga, c->stream));
    return IQGPU_OK;
}

// behind a fused front launch: the verifier, then the unfused kernels as launches that do nothing unless the
// verifier raised its flag (same input, same history buffers, the untouched AGC state)
int Call::stage_agc_verify_and_fallback(const FrontArgs &spec)
{
    AgcArgs va = agc_args();
    va.verify_flag = c->d_agc_flag;
    va.peak_approx = mid ? 1 : 0;
    va.peak2_fallback = (unsigned long long *)c->agc_peak_b.p;
    KernelTimer kt(c, IQGPU_K_AGC);
    HIP_TRY(launch_agc_verify(va, c->stream));
    FrontArgs fb = spec;
    fb.agc_fused = 0; fb.agc_state = nullptr; fb.agc_peak2 = nullptr;
    fb.out_fmt = IQGPU_FMT_CF32; fb.out = c->abuf.p;
    fb.run_if = c->d_agc_flag;
    if (fat || mid) {
        // the fused launch ran on k_front_fat / k_front_mid with its own tile geometry: the fallback is k_front_s1's (512-frame tiles)
        fb.w_total_tiles = ((int64_t)fb.rem0 + fb.frames_in + kWTile - 1) / kWTile;
        int warm = (int)((c->rp.history_in + kWTile - 1) / kWTile);
        if (warm < 1) warm = 1;
        plan_front_s1(fb, wave_slots(front_s1_waves(fb)), fixed_tpw(), warm, 1, kWTile);
    }
    HIP_TRY(launch_front_s1(fb, c->stream));
    AgcArgs ga = va;
    ga.peak2_fallback = nullptr;
    ga.peak2 = (unsigned long long *)c->agc_peak_b.p;
    ga.run_if = c->d_agc_flag; ga.verify_flag = nullptr;
    HIP_TRY(launch_agc(ga, c->stream));
    return IQGPU_OK;
}

} // namespace

static int process_device_impl(iqgpu_chain *c, const void *d_raw_in, size_t frames_in,
                               void *d_out, size_t out_capacity_bytes, size_t *frames_out);

extern "C" int iqgpu_chain_process_device(iqgpu_chain *c, const void *d_raw_in, size_t frames_in,
                                          void *d_out, size_t out_capacity_bytes, size_t *frames_out)
{
    if (!c || !frames_out) return fail(IQGPU_EINVAL, "iqgpu_chain_process_device: NULL argument");
    int rc = pipe_advance(c, c->pipe_seq); if (rc) return rc;     // batches submitted earlier come first (same stream)
    return process_device_impl(c, d_raw_in, frames_in, d_out, out_capacity_bytes, frames_out);
}

// first frame count (a multiple of the AGC chunk, or the whole call) that must take the unfused AGC path: everything
// while the stream has not locked.  agc_apply locks on the first chunk that STARTS after AGC_DIGITAL_LOCK_TIME of
// output (src/agc.c:151-155: elapsed = samples_seen / rate before this chunk is counted), a closed form of the
// stream position; sets *locks when that chunk lies in this call.
static size_t agc_unfused_head(const iqgpu_chain *c, size_t frames_in, bool *locks)
{
    *locks = false;
    if (c->agc_locked_host) return 0;
    AgcGeom g{};
    g.frames_in = (int64_t)frames_in; g.chunk_frames = c->agc_chunk;
    g.n_chunks = (int)(((int64_t)frames_in + c->agc_chunk - 1) / c->agc_chunk);
    g.mode = 1; g.rem = c->rem; g.S = c->S; g.phi = c->phi; g.step = c->rp.step;
    int64_t lo = 0, hi = g.n_chunks;                        // first chunk whose start time exceeds the lock time
    while (lo < hi) {
        const int64_t mid = (lo + hi) / 2;
        const uint64_t seen = c->agc_seen_host + (uint64_t)agc_out_end(g, mid - 1);
        if ((double)seen / c->target_rate > (double)2.0f) hi = mid; else lo = mid + 1;
    }
    // (empty chunks never reach agc_apply; a decimating chain with chunks of at least a tile has none but a possible
    //  first one, which the search passes over because its successor starts at the same time)
    while (lo < g.n_chunks && agc_out_end(g, lo) == agc_out_end(g, lo - 1)) ++lo;
    if (lo >= g.n_chunks) return frames_in;
    *locks = true;
    const int64_t head = (lo + 1) * c->agc_chunk;
    return head < (int64_t)frames_in ? (size_t)head : frames_in;
}

static int process_one(iqgpu_chain *c, const void *d_raw_in, size_t frames_in,
                       void *d_out, size_t out_capacity_bytes, size_t *frames_out, bool agc_fused);

static int process_device_impl(iqgpu_chain *c, const void *d_raw_in, size_t frames_in,
                               void *d_out, size_t out_capacity_bytes, size_t *frames_out)
{
    if (!c || !frames_out) return fail(IQGPU_EINVAL, "iqgpu_chain_process_device: NULL argument");
    if (!c->agc_fusable || frames_in == 0) return process_one(c, d_raw_in, frames_in, d_out, out_capacity_bytes, frames_out, false);
    // output AGC on the specialised front kernel: the scanning phase (and the chunk that locks) through the unfused
    // kernels, everything behind it fused
    *frames_out = 0;
    bool locks = false;
    const size_t head = agc_unfused_head(c, frames_in, &locks);
    const size_t ibps = bytes_per_frame(c->desc.in_format), obps = bytes_per_frame(c->desc.out_format);
    if ((size_t)plan_call(c, frames_in).n_emit * obps > out_capacity_bytes)
        return fail(IQGPU_ECAPACITY, "output buffer too small: need %zu bytes, have %zu", (size_t)plan_call(c, frames_in).n_emit * obps, out_capacity_bytes);
    size_t n1 = 0, n2 = 0;
    if (head > 0) {
        const int rc = process_one(c, d_raw_in, head, d_out, out_capacity_bytes, &n1, false);
        if (rc) return rc;
        c->agc_seen_host += n1;
        if (locks) c->agc_locked_host = true;
    }
    if (head < frames_in) {
        const int rc = process_one(c, (const char *)d_raw_in + head * ibps, frames_in - head, (char *)d_out + n1 * obps,
                                   out_capacity_bytes - n1 * obps, &n2, true);
        if (rc) return rc;
        c->agc_seen_host += n2;
    }
    *frames_out = n1 + n2;
    return IQGPU_OK;
}

static int process_one(iqgpu_chain *c, const void *d_raw_in, size_t frames_in,
                       void *d_out, size_t out_capacity_bytes, size_t *frames_out, bool agc_fused)
{
    if (!c || !frames_out) return fail(IQGPU_EINVAL, "iqgpu_chain_process_device: NULL argument");
    *frames_out = 0;
    if (frames_in == 0) return IQGPU_OK;
    if (!d_raw_in || !d_out) return fail(IQGPU_EINVAL, "iqgpu_chain_process_device: NULL buffer");
    if (frames_in > ((size_t)1 << 40)) return fail(IQGPU_EINVAL, "frames_in too large");
    if (c->poisoned) return fail(IQGPU_EHIP, "an earlier call failed half way through: the stream state is undefined until iqgpu_chain_reset()");
    HIP_TRY(hipSetDevice(c->device));

    Call k{};
    k.c = c; k.d_raw_in = d_raw_in; k.frames_in = frames_in; k.d_out = d_out;
    k.p = plan_call(c, frames_in);
    const size_t obps = bytes_per_frame(c->desc.out_format);
    if ((size_t)k.p.n_emit * obps > out_capacity_bytes)
        return fail(IQGPU_ECAPACITY, "output buffer too small: need %zu bytes, have %zu", (size_t)k.p.n_emit * obps, out_capacity_bytes);
    k.filt = c->fp.enabled;
    k.L1 = k.filt ? c->fp.taps.size() - 1 : 0;
    k.fpending0 = c->fpending;
    // with the AGC on, the last stage leaves cf32 in abuf and k_agc_apply packs -- unless the call is past the lock
    // on a chain whose front kernel applies the gain itself (fused: packed output straight to the caller)
    k.fin_out = d_out; k.fin_fmt = c->desc.out_format;
    k.agc_fused = agc_fused;
    if (c->agc && !agc_fused) {
        int rc = c->abuf.ensure(((size_t)k.p.n_emit + 1) * sizeof(cf2)); if (rc) return rc;
        k.fin_out = c->abuf.p; k.fin_fmt = IQGPU_FMT_CF32;
    }
    k.plan_geometry();
    if (c->iq_pinned) { k.iq_mag = c->iq_pin_mag; k.iq_phase = c->iq_pin_phase; }                  // a pipelined batch: as of its submit()
    else { std::lock_guard<std::mutex> g(c->aux_mu); k.iq_mag = c->iq_mag; k.iq_phase = c->iq_phase; }   // read once per call

    // every buffer the stages need is sized before the first launch, so that an allocation failure leaves the
    // stream state untouched; a failure after that (a launch error) leaves the device state half advanced:
    // the handle is poisoned and every later call fails until iqgpu_chain_reset()
    int rc;
    if ((rc = k.prepare_buffers()) != IQGPU_OK) return rc;
    bool want_probe = false;
    {
        std::lock_guard<std::mutex> g(c->aux_mu);
        want_probe = c->probe_on && frames_in >= 1024 && !c->probe_pending;
    }
    if (want_probe) {
        // before k_dc_scan moves the dc state to the end of this call
        IqProbeArgs pa{};
        pa.raw = d_raw_in; pa.in_fmt = c->desc.in_format; pa.gain = c->desc.gain;
        pa.dc_enable = c->dc ? 1 : 0; pa.dc_c = c->dc_c; pa.dc_state = c->d_dc_state;
        pa.iq_enable = c->desc.iq_correct_enable ? 1 : 0; pa.iq_magp1 = 1.0f + k.iq_mag; pa.iq_phase = k.iq_phase;
        pa.nco_mode = c->nco_mode; pa.nco_theta0 = c->nco_theta; pa.nco_dtheta = c->nco_dtheta; pa.nco_tab = c->d_nco_tab;
        pa.out = c->d_probe;
        HIP_TRY(launch_iq_probe(pa, c->stream));
        HIP_TRY(hipMemcpyAsync(c->h_probe, c->d_probe, 1024 * sizeof(cf2), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipEventRecord(c->probe_done, c->stream));
        std::lock_guard<std::mutex> g(c->aux_mu);
        c->probe_pending = true;
    }
    if (c->dc && (rc = k.stage_dc_carries()) != IQGPU_OK) { c->poisoned = true; return rc; }
    if ((rc = k.stage_front()) != IQGPU_OK) { c->poisoned = true; return rc; }
    if (k.filt && (rc = k.stage_filter()) != IQGPU_OK) { c->poisoned = true; return rc; }
    if (c->late && (rc = k.stage_late_resampler()) != IQGPU_OK) { c->poisoned = true; return rc; }
    if (c->agc && !agc_fused && (rc = k.stage_agc()) != IQGPU_OK) { c->poisoned = true; return rc; }

    // ---- advance the stream position ----
    c->nco_theta += (uint32_t)frames_in * c->nco_dtheta;
    c->pnco_theta += (uint32_t)(uint64_t)k.p.n_emit * c->nco_dtheta;
    c->rem = k.p.rem_next;
    c->phi = k.p.phi_next;
    *frames_out = (size_t)k.p.n_emit;
    return IQGPU_OK;
}

extern "C" int iqgpu_chain_process(iqgpu_chain *c, const void *raw_in, size_t frames_in,
                                   void *out, size_t out_capacity_bytes, size_t *frames_out)
{
    if (!c || !frames_out) return fail(IQGPU_EINVAL, "iqgpu_chain_process: NULL argument");
    *frames_out = 0;
    if (frames_in == 0) return IQGPU_OK;
    if (!raw_in || !out) return fail(IQGPU_EINVAL, "iqgpu_chain_process: NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    int rc = pipe_advance(c, c->pipe_seq); if (rc) return rc;     // batches submitted earlier come first
    const size_t ibps = bytes_per_frame(c->desc.in_format), obps = bytes_per_frame(c->desc.out_format);
    const size_t n_emit = (size_t)plan_call(c, frames_in).n_emit;
    if (n_emit * obps > out_capacity_bytes)
        return fail(IQGPU_ECAPACITY, "output buffer too small: need %zu bytes, have %zu", n_emit * obps, out_capacity_bytes);
    rc = c->stage_in.ensure(frames_in * ibps); if (rc) return rc;
    rc = c->stage_out.ensure(n_emit * obps + 16); if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->stage_in.p, raw_in, frames_in * ibps, hipMemcpyHostToDevice, c->stream));
    size_t produced = 0;
    rc = process_device_impl(c, c->stage_in.p, frames_in, c->stage_out.p, c->stage_out.cap, &produced);
    if (rc) return rc;
    if (produced) HIP_TRY(hipMemcpyAsync(out, c->stage_out.p, produced * obps, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *frames_out = produced;
    return IQGPU_OK;
}

// ---- pipelined host entry point ----------------------------------------------------------------------
// Three stages -- H2D copies, kernels (the chain's stream), D2H copies -- and the HOST moves a batch from one to the
// next: submit(t) queues copy t, then the kernels of batch t-3 once hipEventSynchronize has seen its copy land, then
// the D2H copy of batch t-5 once its kernels are done (events that have normally fired before they are asked for).
// collect(t) pushes batch t through whatever stages it still lacks and waits for its bytes.  Nothing on the device
// ever waits on another stream and no stream switches between the copy engine and the compute queue.  That is the
// whole design: on this runtime a stream that waits on an event which has not fired yet -- or runs a copy behind a
// kernel -- loses ~20 us per hand-over.  Measured (round 2, tools/hostcall_bench.c, us per batch at
// 2^14 / 2^18 / 2^20 / 2^24 frames): one in-order stream per slot (H2D, kernels, D2H) with the kernels of consecutive
// tickets chained by events 33 / 33 / 77 / 1315; all kernels on the chain's stream behind one H2D and one D2H stream,
// chained by events 26 / 42 / 101 / 1212; per-slot copy streams 39 / 45 / 81 / 1313; host-ordered with the D2H copy on
// the kernels' stream 34 / 42 / 91 / 1197; the front kernel reading the batch straight from pinned host memory (no copy
// at all) 17 / 34 / 104 / - (tools/zc_probe.py); a shader copy instead of the copy engine 22 / 41 / 132 / -; this layout
// 22 / 25 / 80 / 1200.  What is left at 2^18 frames is the copy engine itself: rocprofv3 shows the 1 MiB H2D copies back
// to back at 25 us each (40 GB/s; 56 GB/s from 16 MiB up) whatever stream they are queued on, the kernel at 12 us.
static constexpr size_t kSmallCopy = (size_t)8 << 20;      // copies up to this size rotate over the copy streams

static int pipe_init(iqgpu_chain *c)
{
    if (c->pipe_ready) return IQGPU_OK;
    for (hipStream_t &st : c->pipe_h2d) HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (hipStream_t &st : c->pipe_d2h) HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (auto &ps : c->pipe) {
        HIP_TRY(hipEventCreateWithFlags(&ps.in_done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ps.k_done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ps.all_done, hipEventDisableTiming));
    }
    c->pipe_ready = true;
    return IQGPU_OK;
}

// kernels of every submitted batch up to ticket `upto`, in ticket order, on the chain's stream
static int pipe_advance(iqgpu_chain *c, uint64_t upto)
{
    if (!c->pipe_ready) return IQGPU_OK;
    while (c->pipe_launched < upto) {
        iqgpu_chain::PipeSlot &ps = c->pipe[c->pipe_launched % iqgpu_chain::kPipeSlots];
        HIP_TRY(hipSetDevice(c->device));
        int rc = IQGPU_OK;
        if (ps.frames_in) {
            HIP_TRY(hipEventSynchronize(ps.in_done));
            size_t produced = 0;
            c->iq_pinned = true; c->iq_pin_mag = ps.iq_mag; c->iq_pin_phase = ps.iq_phase;
            rc = process_device_impl(c, ps.d_in.p, ps.frames_in, ps.d_out.p, ps.d_out.cap, &produced);
            c->iq_pinned = false;
            if (!rc && produced != ps.n_emit) rc = fail(IQGPU_EHIP, "internal: batch produced %zu frames, planned %zu", produced, ps.n_emit);
            if (rc) ps.n_emit = 0;                               // nothing of a failed batch is copied back
        }
        ++c->pipe_launched;
        HIP_TRY(hipEventRecord(ps.k_done, c->stream));
        if (rc) return rc;
    }
    return IQGPU_OK;
}

// D2H copy of every launched batch up to ticket `upto`, on the D2H stream
static int pipe_drain(iqgpu_chain *c, uint64_t upto)
{
    if (!c->pipe_ready) return IQGPU_OK;
    if (upto > c->pipe_launched) upto = c->pipe_launched;
    while (c->pipe_copied < upto) {
        iqgpu_chain::PipeSlot &ps = c->pipe[c->pipe_copied % iqgpu_chain::kPipeSlots];
        hipStream_t d2h = c->pipe_d2h[ps.n_emit * bytes_per_frame(c->desc.out_format) <= kSmallCopy ? c->pipe_copied % (uint64_t)iqgpu_chain::kCopyStreams : 0];
        HIP_TRY(hipSetDevice(c->device));
        if (ps.n_emit) {
            HIP_TRY(hipEventSynchronize(ps.k_done));
            HIP_TRY(hipMemcpyAsync(ps.out, ps.d_out.p, ps.n_emit * bytes_per_frame(c->desc.out_format), hipMemcpyDeviceToHost, d2h));
        }
        ++c->pipe_copied;
        HIP_TRY(hipEventRecord(ps.all_done, ps.n_emit ? d2h : c->stream));
    }
    return IQGPU_OK;
}

extern "C" int iqgpu_chain_submit(iqgpu_chain *c, const void *raw_in, size_t frames_in,
                                  void *out, size_t out_capacity_bytes, size_t *frames_out, uint64_t *ticket)
{
    if (!c || !frames_out || !ticket) return fail(IQGPU_EINVAL, "iqgpu_chain_submit: NULL argument");
    *frames_out = 0; *ticket = 0;
    if (frames_in != 0 && (!raw_in || !out)) return fail(IQGPU_EINVAL, "iqgpu_chain_submit: NULL buffer");
    if (frames_in > ((size_t)1 << 40)) return fail(IQGPU_EINVAL, "frames_in too large");
    HIP_TRY(hipSetDevice(c->device));
    int rc = pipe_init(c); if (rc) return rc;
    iqgpu_chain::PipeSlot &ps = c->pipe[c->pipe_seq % iqgpu_chain::kPipeSlots];
    if (ps.busy) return fail(IQGPU_EINVAL, "iqgpu_chain_submit: %d batches are in flight; collect ticket %llu first",
                             iqgpu_chain::kPipeSlots, (unsigned long long)ps.ticket);
    const size_t ibps = bytes_per_frame(c->desc.in_format), obps = bytes_per_frame(c->desc.out_format);
    // Everything that can refuse the batch comes first and touches nothing: the exact output count (a closed form of the
    // stream position behind the tickets already handed out -- pipe_advance below never moves that position), the capacity
    // check, and both device buffers.  Only then is anything queued, and the look-ahead position moves together with the
    // ticket at the very end: a refused submit leaves the handle exactly as it was.
    StreamPos at;
    if (c->pipe_launched == c->pipe_seq) { at.rem = c->rem; at.phi = c->phi; at.fpending = c->fpending; }
    else { at.rem = c->pipe_rem; at.phi = c->pipe_phi; at.fpending = c->pipe_fpending; }
    const CallPlan plan = plan_call_at(c, at, frames_in);
    const size_t n_emit = (size_t)plan.n_emit;
    if (n_emit * obps > out_capacity_bytes)
        return fail(IQGPU_ECAPACITY, "output buffer too small: need %zu bytes, have %zu", n_emit * obps, out_capacity_bytes);
    if (frames_in) {
        rc = ps.d_in.ensure(frames_in * ibps); if (rc) return rc;
        rc = ps.d_out.ensure(n_emit * obps + 16); if (rc) return rc;
    }
    // this batch's copy next (it needs nothing but the slot), so that the copy stream never idles while the host
    // queues the previous batch's kernels
    if (frames_in) {
        hipStream_t h2d = c->pipe_h2d[frames_in * ibps <= kSmallCopy ? c->pipe_seq % (uint64_t)iqgpu_chain::kCopyStreams : 0];
        HIP_TRY(hipMemcpyAsync(ps.d_in.p, raw_in, frames_in * ibps, hipMemcpyHostToDevice, h2d));
        HIP_TRY(hipEventRecord(ps.in_done, h2d));
    }
    // ... then the kernels of the batch kLagK tickets back and the D2H copy of the batch kLagD behind that one: far
    // enough behind for their events to have fired (a 1 MiB copy takes ~40 us from hipMemcpyAsync to a visible event,
    // a kernel with its event ~25 us)
    const uint64_t t = c->pipe_seq + 1;
    constexpr uint64_t kLagK = 3, kLagD = 2;
    static_assert(kLagK + kLagD < (uint64_t)iqgpu_chain::kPipeSlots, "a batch must leave the pipeline before its slot comes round again");
    if (t > kLagK) { rc = pipe_advance(c, t - kLagK); if (rc) return rc; }
    if (t > kLagK + kLagD) { rc = pipe_drain(c, t - kLagK - kLagD); if (rc) return rc; }
    if (frames_in) { c->pipe_rem = plan.rem_next; c->pipe_phi = plan.phi_next; c->pipe_fpending = c->fp.enabled ? plan.fpending_next : at.fpending; }
    else { c->pipe_rem = at.rem; c->pipe_phi = at.phi; c->pipe_fpending = at.fpending; }
    ps.frames_in = frames_in; ps.n_emit = n_emit; ps.out = out;
    { std::lock_guard<std::mutex> g(c->aux_mu); ps.iq_mag = c->iq_mag; ps.iq_phase = c->iq_phase; }
    ps.ticket = ++c->pipe_seq; ps.busy = true;
    *ticket = ps.ticket; *frames_out = n_emit;
    return IQGPU_OK;
}

extern "C" int iqgpu_chain_collect(iqgpu_chain *c, uint64_t ticket)
{
    if (!c) return fail(IQGPU_EINVAL, "NULL chain");
    if (ticket == 0 || ticket > c->pipe_seq) return fail(IQGPU_EINVAL, "iqgpu_chain_collect: unknown ticket %llu", (unsigned long long)ticket);
    iqgpu_chain::PipeSlot &ps = c->pipe[(ticket - 1) % iqgpu_chain::kPipeSlots];
    if (!ps.busy || ps.ticket != ticket) return IQGPU_OK;        // collected before
    HIP_TRY(hipSetDevice(c->device));
    int rc = pipe_advance(c, ticket);
    if (rc && c->pipe_launched < ticket) return rc;              // an earlier batch failed; this one has not run
    { const int rc2 = pipe_drain(c, ticket); if (!rc) rc = rc2; }
    if (c->pipe_copied >= ticket) HIP_TRY(hipEventSynchronize(ps.all_done));
    ps.busy = false;
    return rc;
}

extern "C" int iqgpu_chain_pipeline_depth(void) { return iqgpu_chain::kPipeSlots; }

// ---- I/Q optimiser hand-off (src/pipeline.c:468-476, src/utility_threads.c:35-47) -------------------------
extern "C" int iqgpu_chain_enable_iq_probe(iqgpu_chain *c, int enable)
{
    if (!c) return fail(IQGPU_EINVAL, "NULL chain");
    HIP_TRY(hipSetDevice(c->device));
    if (enable && !(c->d_probe && c->h_probe && c->probe_done)) {
        // all three resources or none: a partial failure leaves nothing behind, so that a second enable() starts over
        cf2 *dp = nullptr, *hp = nullptr; hipEvent_t ev = nullptr;
        const bool ok = hipMalloc((void **)&dp, 1024 * sizeof(cf2)) == hipSuccess &&
                        hipHostMalloc((void **)&hp, 1024 * sizeof(cf2), hipHostMallocDefault) == hipSuccess &&
                        hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            if (ev) (void)hipEventDestroy(ev);
            if (hp) (void)hipHostFree(hp);
            if (dp) (void)hipFree(dp);
            (void)hipGetLastError();
            return fail(IQGPU_ENOMEM, "iqgpu_chain_enable_iq_probe: could not allocate the probe buffers");
        }
        std::lock_guard<std::mutex> g(c->aux_mu);
        c->d_probe = dp; c->h_probe = hp; c->probe_done = ev;
    }
    { std::lock_guard<std::mutex> g(c->aux_mu); c->probe_on = enable != 0; }     // process_one reads it under the same lock
    return IQGPU_OK;
}

extern "C" int iqgpu_chain_read_iq_probe(iqgpu_chain *c, float *block_re_im_1024, int *valid)
{
    if (!c || !block_re_im_1024 || !valid) return fail(IQGPU_EINVAL, "iqgpu_chain_read_iq_probe: NULL argument");
    *valid = 0;
    if (!c->h_probe) return fail(IQGPU_EINVAL, "the probe is not enabled (iqgpu_chain_enable_iq_probe)");
    bool pending;
    { std::lock_guard<std::mutex> g(c->aux_mu); pending = c->probe_pending; }
    if (pending) {
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipEventSynchronize(c->probe_done));          // recorded behind the copy into h_probe
        std::lock_guard<std::mutex> g(c->aux_mu);
        memcpy(c->probe_last, c->h_probe, 1024 * sizeof(cf2));
        c->probe_pending = false; c->probe_valid = true;       // the stage thread may stage the next block now
    }
    std::lock_guard<std::mutex> g(c->aux_mu);
    if (!c->probe_valid) return IQGPU_OK;
    memcpy(block_re_im_1024, c->probe_last, 1024 * sizeof(cf2));
    *valid = 1;
    return IQGPU_OK;
}

extern "C" int iqgpu_iq_optimizer_service(iqgpu_iq_optimizer *o, iqgpu_chain *c, double now_sec, int *updated)
{
    if (!o || !c) return fail(IQGPU_EINVAL, "iqgpu_iq_optimizer_service: NULL argument");
    if (updated) *updated = 0;
    static_assert(sizeof(cf2) == 2 * sizeof(float), "cf32 layout");
    float block[2048];
    int valid = 0, upd = 0;
    int rc = iqgpu_chain_read_iq_probe(c, block, &valid);
    if (rc != IQGPU_OK || !valid) return rc;
    rc = iqgpu_iq_optimizer_run(o, block, now_sec, &upd);
    if (rc != IQGPU_OK) return fail(rc, "iqgpu_iq_optimizer_run failed");
    if (upd) {
        float mag = 0.0f, phase = 0.0f;
        (void)iqgpu_iq_optimizer_get_factors(o, &mag, &phase);
        rc = iqgpu_chain_set_iq_factors(c, mag, phase);
    }
    if (updated) *updated = upd;
    return rc;
}

extern "C" int iqgpu_chain_reset(iqgpu_chain *c)
{
    if (!c) return fail(IQGPU_EINVAL, "NULL chain");
    HIP_TRY(hipSetDevice(c->device));
    // pre_processor_reset (dc state, NCO phase, filter), resampler_reset, post_processor_reset
    { const int rc = pipe_advance(c, c->pipe_seq); if (rc && !c->poisoned) return rc; }   // batches in flight come first (same stream)
    c->poisoned = false;
    c->rem = 0; c->phi = 0; c->nco_theta = 0; c->pnco_theta = 0;
    c->agc_locked_host = false; c->agc_seen_host = 0; c->agc_peak_clean = false;
    HIP_TRY(hipMemsetAsync(c->d_dc_state, 0, sizeof(cd2), c->stream));
    if (c->agc) { // agc_reset, src/agc.c:224-238
        c->agc_init.last_strong = c->desc.agc_clock == IQGPU_AGC_CLOCK_WALL ? monotonic_sec() : 0.0;
        HIP_TRY(hipMemcpyAsync(c->d_agc_state, &c->agc_init, sizeof(AgcState), hipMemcpyHostToDevice, c->stream));
    }
    if (c->late) HIP_TRY(hipMemsetAsync(c->ibuf[c->icur].p, 0, (size_t)c->ihist * sizeof(cf2), c->stream));
    if (c->decim)
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(hipMemsetAsync(c->d_hist[i], 0, (size_t)c->hist_cap * sizeof(cf2), c->stream));
            if (c->cascade) HIP_TRY(hipMemsetAsync(c->d_hist2[i], 0, (size_t)c->hist2_cap * sizeof(cf2), c->stream));
        }
    if (c->fp.enabled) {
        // the filter object's history is cleared; the FFT remainder is NOT (src/filter.c:417-436):
        // pending samples stay queued in front of the new stream
        const size_t L1 = c->fp.taps.size() - 1;
        HIP_TRY(hipMemsetAsync(c->fbuf[c->fcur].p, 0, L1 * sizeof(cf2), c->stream));
    }
    return IQGPU_OK;
}

extern "C" int iqgpu_chain_get_agc_state(iqgpu_chain *c, iqgpu_agc_state *st)
{
    if (!c || !st) return fail(IQGPU_EINVAL, "iqgpu_chain_get_agc_state: NULL argument");
    if (!c->agc) return fail(IQGPU_EINVAL, "the chain has no output AGC");
    static_assert(sizeof(iqgpu_agc_state) == sizeof(AgcState), "AGC state layout");
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = pipe_advance(c, c->pipe_seq); if (rc) return rc; }     // batches submitted and not yet collected
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(st, c->d_agc_state, sizeof(AgcState), hipMemcpyDeviceToHost));
    return IQGPU_OK;
}

extern "C" int iqgpu_chain_set_iq_factors(iqgpu_chain *c, float mag, float phase)
{
    if (!c) return fail(IQGPU_EINVAL, "NULL chain");
    std::lock_guard<std::mutex> g(c->aux_mu);                 // the reference's iq_factors_mutex (iq_correct.c:141-152)
    c->iq_mag = mag; c->iq_phase = phase;
    return IQGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// stream / profiling plumbing
// ------------------------------------------------------------------------------------------------
extern "C" int iqgpu_chain_set_stream(iqgpu_chain *c, void *hip_stream)
{
    if (!c) return fail(IQGPU_EINVAL, "NULL chain");
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = pipe_advance(c, c->pipe_seq); if (rc) return rc; }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return IQGPU_OK;
}
extern "C" void *iqgpu_chain_get_stream(const iqgpu_chain *c) { return c ? (void *)c->stream : nullptr; }
extern "C" int iqgpu_chain_synchronize(iqgpu_chain *c)
{
    if (!c) return fail(IQGPU_EINVAL, "NULL chain");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = pipe_advance(c, c->pipe_seq); if (!rc) rc = pipe_drain(c, c->pipe_seq); if (rc) return rc; }
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (hipStream_t st : c->pipe_d2h) if (st) HIP_TRY(hipStreamSynchronize(st));
    return IQGPU_OK;
}
extern "C" int iqgpu_chain_set_profiling(iqgpu_chain *c, int enable)
{
    if (!c) return fail(IQGPU_EINVAL, "NULL chain");
    c->profiling = enable != 0;
    return IQGPU_OK;
}
extern "C" int iqgpu_chain_get_profile(iqgpu_chain *c, iqgpu_profile *p)
{
    if (!c || !p) return fail(IQGPU_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    drain_events(c);
    *p = c->prof;
    memset(&c->prof, 0, sizeof(c->prof));
    return IQGPU_OK;
}

extern "C" int iqgpu_chain_debug_read_scratch(iqgpu_chain *c, void *host_64k)
{
    if (!c || !host_64k) return fail(IQGPU_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(host_64k, c->d_sink, 64 * 1024, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(c->d_sink, 0, 64 * 1024));
    return IQGPU_OK;
}

// ------------------------------------------------------------------------------------------------
// operator-level entry points
// ------------------------------------------------------------------------------------------------
static int convert_via_chain(const void *in, void *out, size_t frames, int in_fmt, int out_fmt, float gain, int device)
{
    iqgpu_chain_desc d;
    iqgpu_chain_desc_init(&d);
    d.in_format = in_fmt; d.out_format = out_fmt; d.gain = gain; d.no_resample = 1;
    d.input_rate_hz = 1.0; d.target_rate_hz = 1.0; d.device_ordinal = device;
    iqgpu_chain *c = nullptr;
    int rc = iqgpu_chain_create(&d, &c);
    if (rc) return rc;
    size_t n = 0;
    rc = iqgpu_chain_process(c, in, frames, out, frames * bytes_per_frame(out_fmt), &n);
    iqgpu_chain_destroy(c);
    if (rc == IQGPU_OK && n != frames) return fail(IQGPU_EHIP, "convert produced %zu of %zu frames", n, frames);
    return rc;
}

extern "C" int iqgpu_convert_block_to_cf32(const void *in, float *out_re_im, size_t frames, int in_format, float gain, int device)
{
    if (!bytes_per_frame(in_format)) return fail(IQGPU_EFORMAT, "Unhandled input format: %d", in_format);
    return convert_via_chain(in, out_re_im, frames, in_format, IQGPU_FMT_CF32, gain, device);
}

extern "C" int iqgpu_convert_cf32_to_block(const float *in_re_im, void *out, size_t frames, int out_format, int device)
{
    if (!bytes_per_frame(out_format)) return fail(IQGPU_EFORMAT, "Unhandled output format: %d", out_format);
    return convert_via_chain(in_re_im, out, frames, IQGPU_FMT_CF32, out_format, 1.0f, device);
}

// ------------------------------------------------------------------------------------------------
// device memory helpers
// ------------------------------------------------------------------------------------------------
extern "C" int iqgpu_device_malloc(int device, size_t bytes, void **d_ptr)
{
    if (!d_ptr) return fail(IQGPU_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(device));
    if (hipMalloc(d_ptr, bytes ? bytes : 1) != hipSuccess) return fail(IQGPU_ENOMEM, "hipMalloc(%zu) failed", bytes);
    return IQGPU_OK;
}
extern "C" int iqgpu_device_free(int device, void *d_ptr) { HIP_TRY(hipSetDevice(device)); HIP_TRY(hipFree(d_ptr)); return IQGPU_OK; }
extern "C" int iqgpu_host_malloc_pinned(size_t bytes, void **h_ptr)
{
    if (!h_ptr) return fail(IQGPU_EINVAL, "NULL argument");
    if (hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return fail(IQGPU_ENOMEM, "hipHostMalloc(%zu) failed", bytes);
    return IQGPU_OK;
}
extern "C" int iqgpu_host_free_pinned(void *h_ptr) { HIP_TRY(hipHostFree(h_ptr)); return IQGPU_OK; }
extern "C" int iqgpu_memcpy_h2d(int device, void *d_dst, const void *h_src, size_t bytes)
{
    HIP_TRY(hipSetDevice(device)); HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice)); return IQGPU_OK;
}
extern "C" int iqgpu_memcpy_d2h(int device, void *h_dst, const void *d_src, size_t bytes)
{
    HIP_TRY(hipSetDevice(device)); HIP_TRY(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost)); return IQGPU_OK;
}
extern "C" int iqgpu_memcpy_h2d_async(void *d_dst, const void *h_src, size_t bytes, void *s)
{
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)s)); return IQGPU_OK;
}
extern "C" int iqgpu_memcpy_d2h_async(void *h_dst, const void *d_src, size_t bytes, void *s)
{
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)s)); return IQGPU_OK;
}
extern "C" int iqgpu_stream_create(int device, void **s)
{
    if (!s) return fail(IQGPU_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *s = (void *)st;
    return IQGPU_OK;
}
extern "C" int iqgpu_stream_destroy(void *s) { HIP_TRY(hipStreamDestroy((hipStream_t)s)); return IQGPU_OK; }
extern "C" int iqgpu_stream_synchronize(void *s) { HIP_TRY(hipStreamSynchronize((hipStream_t)s)); return IQGPU_OK; }
extern "C" int iqgpu_event_create(void **e)
{
    if (!e) return fail(IQGPU_EINVAL, "NULL argument");
    hipEvent_t ev = nullptr;
    HIP_TRY(hipEventCreate(&ev));
    *e = (void *)ev;
    return IQGPU_OK;
}
extern "C" int iqgpu_event_destroy(void *e) { HIP_TRY(hipEventDestroy((hipEvent_t)e)); return IQGPU_OK; }
extern "C" int iqgpu_event_record(void *e, void *s) { HIP_TRY(hipEventRecord((hipEvent_t)e, (hipStream_t)s)); return IQGPU_OK; }
extern "C" int iqgpu_stream_wait_event(void *s, void *e) { HIP_TRY(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)e, 0)); return IQGPU_OK; }
extern "C" int iqgpu_event_elapsed_ms(void *a, void *b, float *ms)
{
    if (!ms) return fail(IQGPU_EINVAL, "NULL argument");
    HIP_TRY(hipEventSynchronize((hipEvent_t)b));
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return IQGPU_OK;
}
