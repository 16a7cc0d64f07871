// abi.cpp -- the C ABI of libiqgpu (include/iqgpu.h): library-level entry points, chain lifecycle with the create-time design,
// state access, the operator-level and device-memory helpers.  No CPU compute path exists here: every entry point that moves
// samples launches the gfx950 kernels or fails.
#include "chain.hpp"

// ------------------------------------------------------------------------------------------------
// error reporting
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
const char *last_error_text() { return g_err; }

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ------------------------------------------------------------------------------------------------
// diagnostic switches: ONE explicit entry point, iqgpu_debug_set(name, value).  The library never reads its switches from the
// environment (round 6, VERDICT r5 item 7: a production library whose kernel selection hangs on the caller's environment is a
// support problem); a chain takes the table as it stands when iqgpu_chain_create runs, and nothing on the launch path looks at it.
// ------------------------------------------------------------------------------------------------
static const char *const kDebugNames[] = {
    "force_generic", "no_fast", "agc_nofuse", "no_raw0", "no_kt", "fft_no_r16", "no_fat", "force_fat", "fat", "mid8", "no_s2",
    "no_fused_move", "no_p0", "no_casc2", "no_mid_8bit", "fuse_filter", "tap_fold", "steal", "steal_min", "steal_rounds",
    "steal_stride", "steal_lanes", "run_weights", "cus", "fft_log2n", "fft_threads", "fft_geometry", "casc2_min_run", "sysfs_root",
};
static std::mutex g_dbg_mu;
static std::map<std::string, std::string> &dbg_table() { static std::map<std::string, std::string> t; return t; }

// value of switch `name` ("" when unset); a copy: the table may change under another thread
std::string debug_value(const char *name)
{
    std::lock_guard<std::mutex> g(g_dbg_mu);
    const auto it = dbg_table().find(name);
    return it == dbg_table().end() ? std::string() : it->second;
}
static bool debug_on(const char *name) { const std::string v = debug_value(name); return !v.empty() && v[0] == '1'; }

extern "C" int iqgpu_debug_set(const char *name, const char *value)
{
    std::lock_guard<std::mutex> g(g_dbg_mu);
    if (!name) { dbg_table().clear(); return IQGPU_OK; }
    bool known = false;
    for (const char *k : kDebugNames) known = known || !strcmp(k, name);
    if (!known) return fail(IQGPU_EINVAL, "iqgpu_debug_set: unknown switch '%s'", name);
    if (!value || !value[0]) dbg_table().erase(name);
    else dbg_table()[name] = value;
    return IQGPU_OK;
}

extern "C" int iqgpu_debug_list(char *buf, size_t cap)
{
    if (!buf || cap == 0) return fail(IQGPU_EINVAL, "iqgpu_debug_list: NULL buffer");
    std::lock_guard<std::mutex> g(g_dbg_mu);
    std::string all;
    for (const auto &kv : dbg_table()) { if (!all.empty()) all += ';'; all += kv.first + "=" + kv.second; }
    if (all.size() + 1 > cap) return fail(IQGPU_ECAPACITY, "iqgpu_debug_list: %zu bytes needed", all.size() + 1);
    memcpy(buf, all.c_str(), all.size() + 1);
    return IQGPU_OK;
}

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

extern "C" int iqgpu_device_pci_bus_id(int ordinal, char *buf, size_t cap)
{
    if (!buf || cap < 16) return fail(IQGPU_EINVAL, "iqgpu_device_pci_bus_id: buffer of at least 16 bytes needed");
    buf[0] = 0;
    HIP_TRY(hipDeviceGetPCIBusId(buf, (int)cap, ordinal));
    return IQGPU_OK;
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
    c->abuf.release(); c->agc_peak.release(); c->agc_gain.release(); c->agc_peak_b.release(); c->agc_hist.release();
    if (c->d_agc_flag) (void)hipFree(c->d_agc_flag);
    if (c->h_agc_verdict) (void)hipHostFree((void *)c->h_agc_verdict);
    if (c->d_twiddle) (void)hipFree(c->d_twiddle);
    for (int i = 0; i < 2; ++i) if (c->d_hist[i]) (void)hipFree(c->d_hist[i]);
    for (int i = 0; i < 2; ++i) if (c->d_hist2[i]) (void)hipFree(c->d_hist2[i]);
    c->mid.release();
    if (c->d_dc_state) (void)hipFree(c->d_dc_state);
    if (c->d_sink) (void)hipFree(c->d_sink);
    c->steal_buf.release();
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

// the eight words behind d_agc_flag as a chain starts (and restarts, iqgpu_chain_reset) with them; static storage: the reset's
// asynchronous copy reads it after the call has returned
static const int32_t kAgcFlagInit[8] = {0, 0, 0, -1, 0, 0, 0, 0};

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
int design_chain(iqgpu_chain *c, const iqgpu_chain_desc *d)
{
    if (!bytes_per_frame(d->in_format)) return fail(IQGPU_EFORMAT, "Unhandled input format: %d", d->in_format);
    if (!bytes_per_frame(d->out_format)) return fail(IQGPU_EFORMAT, "Unhandled output format: %d", d->out_format);
    if (!(d->input_rate_hz > 0.0) && !(d->resample_ratio > 0.0f) && !d->no_resample)
        return fail(IQGPU_EINVAL, "input_rate_hz must be positive");
    c->desc = *d;
    c->device = d->device_ordinal;
    {   // every diagnostic switch is taken HERE, once per chain, from the table iqgpu_debug_set keeps (never from the environment):
        // process() and the launch functions look at the handle only
        c->force_generic = debug_on("force_generic");
        // (booleans: only "1..." switches one on -- "0" or an empty value leaves the default kernel selection alone)
        auto on = [](const char *name) { return debug_on(name); };
        c->dbg = (on("no_fast") ? kDbgNoFast : 0u) | (on("agc_nofuse") ? kDbgAgcNoFuse : 0u) |
                 (on("no_raw0") ? kDbgNoRaw0 : 0u) | (on("no_kt") ? kDbgNoKT : 0u) |
                 (on("fft_no_r16") ? kDbgFftNoR16 : 0u) | (on("no_fat") ? kDbgNoFat : 0u) |
                 (on("force_fat") ? kDbgForceFat : 0u) | (on("fat") ? kDbgUseFat : 0u) | (on("mid8") ? kDbgMid8 : 0u) |
                 (on("no_s2") ? kDbgNoS2 : 0u) | (on("no_fused_move") ? kDbgNoFusedMove : 0u) | (on("no_p0") ? kDbgNoP0 : 0u) |
                 (on("no_casc2") ? kDbgNoCasc2 : 0u) | (on("no_mid_8bit") ? kDbgNoMid8bit : 0u) |
                 (on("fuse_filter") ? kDbgFuseFilter : 0u);
        std::string v;
        if (!(v = debug_value("tap_fold")).empty()) c->tap_fold_env = atoi(v.c_str()) != 0 ? 1 : 0;
        if (!(v = debug_value("steal")).empty()) c->steal = v[0] == '1';
        if (!(v = debug_value("steal_min")).empty()) { const int x = atoi(v.c_str()); if (x >= 2 && x < 100000) c->steal_min = x; }
        if (!(v = debug_value("steal_rounds")).empty()) { const int x = atoi(v.c_str()); if (x >= 1 && x <= 64) c->steal_rounds = x; }
        if (!(v = debug_value("steal_stride")).empty()) { const int x = atoi(v.c_str()); if (x >= 1 && x <= 8192) c->steal_stride = x; }
        if (!(v = debug_value("steal_lanes")).empty()) { const int x = atoi(v.c_str()); if (x >= 1 && x <= 64) c->steal_lanes = x; }
        if (!(v = debug_value("run_weights")).empty()) { int x = 0, y = 0, z = 0; if (sscanf(v.c_str(), "%d,%d,%d", &x, &y, &z) == 3
            && x >= 0 && y >= 0 && z >= 0) { c->run_wt[0] = x; c->run_wt[1] = y; c->run_wt[2] = z; } }
        { const int x = atoi(debug_value("casc2_min_run").c_str()); cascade2_set_min_run(x > 0 ? x : 0); }
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
    if (!std::isfinite(c->ratio) || c->ratio < 0.001f || c->ratio > 1000.0f) return fail(IQGPU_ERATIO,
        "Calculated resampling ratio (%.6f) is invalid or outside acceptable range.", (double)c->ratio);
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
    if (d->shift_after_resample && std::fabs(d->shift_hz) < 1e-9) return fail(IQGPU_ESHIFT,
        "Option --shift-after-resample was used, but no effective frequency shift was requested or calculated.");
    if (std::fabs(d->shift_hz) >= 1e-9) {
        const double rate = d->shift_after_resample ? c->target_rate : in_rate;
        if (std::fabs(d->shift_hz) > 5.0 * rate) return fail(IQGPU_ESHIFT,
            "Requested frequency shift %.2f Hz exceeds sanity limit for the rate of %.1f Hz.", d->shift_hz, rate);
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
        if (d->agc_clock != IQGPU_AGC_CLOCK_SAMPLES && d->agc_clock != IQGPU_AGC_CLOCK_WALL) return fail(IQGPU_EINVAL,
            "agc_clock must be IQGPU_AGC_CLOCK_SAMPLES or IQGPU_AGC_CLOCK_WALL");
        c->agc = true;
        // dx / local: liquid agc_crcf with AGC_DX_BANDWIDTH / AGC_LOCAL_BANDWIDTH (src/agc.c:45-57, constants.h:169,175);
        // the target level does not reach the loop (agc_crcf_set_gain(1.0f) behind set_signal_level, agc.c:56-59)
        c->agc_rms_alpha = d->agc_profile == IQGPU_AGC_DX ? 1e-4f : d->agc_profile == IQGPU_AGC_LOCAL ? 1e-2f : 0.0f;
        c->agc_target = d->agc_target > 0.0f ? d->agc_target : 0.9f;      // AGC_DIGITAL_PEAK_TARGET
        c->agc_chunk = d->agc_chunk_frames ? (int64_t)d->agc_chunk_frames : 16384;   // PIPELINE_CHUNK_BASE_SAMPLES
        // k_agc_scan adds the output lengths of 64 chunks in 32 bits
        if ((double)c->agc_chunk * (double)(c->ratio > 1.0f ? c->ratio : 1.0f) * 64.0 >= 2147483648.0)
            return fail(IQGPU_EINVAL, "agc_chunk_frames %lld is too large for this ratio (64 chunks must stay below 2^31 output frames)",
                (long long)c->agc_chunk);
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
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { int rc = fail(IQGPU_ENODEV, "no HIP device available"); delete c;
        return rc; }
    if (c->device < 0 || c->device >= ndev) { int rc = fail(IQGPU_ENODEV, "device_ordinal %d out of range (%d devices)", c->device,
        ndev); delete c; return rc; }
    int rc = IQGPU_OK;
#define CREATE_TRY(expr)                                                                                         \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) { rc = fail(IQGPU_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); goto bad; } \
    } while (0)
#define CREATE_RC(expr) do { rc = (expr); if (rc != IQGPU_OK) goto bad; } while (0)
    {
        CREATE_TRY(hipSetDevice(c->device));
        {
            int ncu = 0;
            if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && ncu > 0) c->n_cu = ncu;
            // (diagnostic: cus=n plans every launch for n CUs -- how does a CU's throughput depend on how many of them work?)
            { const std::string v = debug_value("cus"); if (!v.empty()) { const int x = atoi(v.c_str()); if (x >= 8 && x <= c->n_cu) c->n_cu = x; } }
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
            c->agc_init = AgcState{0, c->agc_rms_alpha > 0.0f ? 1.0f : 0.05f, 1.0f, 0, c->desc.agc_clock == IQGPU_AGC_CLOCK_WALL
                ? monotonic_sec() : 0.0, 0};
            CREATE_TRY(hipMemcpy(c->d_agc_state, &c->agc_init, sizeof(AgcState), hipMemcpyHostToDevice));
            {   // [0] verdict of the verifier, [1] ratchet seen, [2] weak chunk seen, [3] last healthy chunk (agc.hip)
                // ... [4] the tickets of k_agc_classify's workgroups (kAgcFlagInit)
                CREATE_TRY(hipMalloc((void **)&c->d_agc_flag, sizeof(kAgcFlagInit)));
                CREATE_TRY(hipMemcpy(c->d_agc_flag, kAgcFlagInit, sizeof(kAgcFlagInit), hipMemcpyHostToDevice));
                // the verdict's second home: one word of pinned, device-visible host memory (chain.hpp, h_agc_verdict)
                void *hv = nullptr, *dv = nullptr;
                CREATE_TRY(hipHostMalloc(&hv, 64, hipHostMallocMapped | hipHostMallocCoherent));
                c->h_agc_verdict = (volatile int32_t *)hv; c->h_agc_verdict[0] = 0;
                CREATE_TRY(hipHostGetDevicePointer(&dv, hv, 0));
                c->d_agc_verdict = (int32_t *)dv;
            }
            // the fused path exists for the specialised front kernel: the shipped cs16 NRSC-5 preset shape
            FrontArgs fa{};
            fa.dbg = c->dbg;
            fa.S = c->S; fa.in_fmt = c->desc.in_format; fa.out_fmt = c->desc.out_format; fa.gain = c->desc.gain;
            fa.iq_enable = c->desc.iq_correct_enable ? 1 : 0; fa.dc_enable = c->dc ? 1 : 0;
            fa.nco_mode = c->nco_mode; fa.pnco_mode = c->pnco_mode; fa.agc_chunk_frames = c->agc_chunk; fa.agc_shift = c->S;
            if (c->cascade) { fa.S = 1; fa.in_fmt = IQGPU_FMT_CF32; }      // k_cascade in front: the last stage sees cf32, one half-band
            if (c->agc_rms_alpha > 0.0f) {
                int64_t chunk = 0; int32_t nch = 0;
                agc_rms_geometry(c->agc_rms_alpha, 0, 0, &chunk, &c->agc_rms_warm, &nch);
                CREATE_RC(c->agc_hist.ensure(((size_t)c->agc_rms_warm + 1) * sizeof(cf2)));
                CREATE_TRY(hipMemset(c->agc_hist.p, 0, c->agc_hist.cap));
            }
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
            // when the taps need it; measured with k_fftconv16 (round 4: first and last pass in registers) on config 3 (1025 taps):
            // N 2048 0.25 ms, 4096 0.15, 8192 0.16, 16384 0.21; on config 4 (4097 taps): N 8192 0.105 ms, 16384 0.114
            // (from 1024 points up: the radix-16 kernel.  Round 5: the 97-tap band-pass of the cs16 -usb / -lsb presets ran the
            //  radix-4 ping-pong kernel at N = 512 and took 1.49 ms for 83 M outputs; at N = 1024 it takes a fifth of that)
            int lg = (c->dbg & kDbgFftNoR16) ? 8 : 10;
            while ((size_t)(1 << lg) < 4 * (Lt - 1) && (1 << lg) < 4096) ++lg;
            while ((size_t)(1 << lg) < 2 * (Lt - 1)) ++lg;
            // round 6: a chain WITHOUT a half-band stage whose filter stands behind the resampler CAN run both in one kernel (k_p0fft16,
            // p0fft.hpp: no cf32 stream in HBM) -- built, byte-equal to the two kernels on the same windows, and slower than they are
            // (1.43 ms against 1.17 - 1.25 per 2^28 cu8 frames: the polyphase role's registers leave the transforms two waves per SIMD,
            // profiles/r06_fused_filter.md): opt-in, iqgpu_debug_set("fuse_filter", "1").  That kernel's workgroup carries 35 KB of tap
            // planes beside its transform buffer, which wants 4096 points (four waves per workgroup, two workgroups per CU) -- the
            // chain's transform is then that size on either path
            // (diagnostics: "fft_geometry" = "keep" runs the two kernels at the fused kernel's transform size and on its windows -- win
            //  stream samples and vout outputs per block -- so that the two paths can be compared byte for byte)
            c->fft_keep_geometry = debug_value("fft_geometry") == "keep";
            if (c->decim && c->S == 0 && !c->late && c->fp.post_resample && !(c->dbg & (kDbgFftNoR16 | kDbgNoP0 | kDbgNoFusedMove)) &&
                ((c->dbg & kDbgFuseFilter) || c->fft_keep_geometry)) {
                FrontArgs ff{};
                ff.dbg = c->dbg; ff.S = 0; ff.in_fmt = c->desc.in_format; ff.gain = c->desc.gain;
                ff.iq_enable = c->desc.iq_correct_enable ? 1 : 0; ff.dc_enable = c->dc ? 1 : 0; ff.nco_mode = c->nco_mode; ff.pnco_mode = 0;
                ff.step = c->rp.step;
                const int lgf = lg < 12 ? 12 : lg;
                if (p0fft_shape(ff, lgf, (int)Lt) && p0fft_geometry(lgf, (int)Lt, &c->fuse_win, &c->fuse_vout)) {
                    c->fuse_filter = (c->dbg & kDbgFuseFilter) != 0; lg = lgf;
                }
            }
            { const std::string e = debug_value("fft_log2n"); if (!e.empty()) { const int v = atoi(e.c_str()); if (v >= 1 && (1 << v) <= kMaxFftN
                && (size_t)(1 << v) >= 2 * (Lt - 1)) lg = v; } }
            { const std::string e = debug_value("fft_threads"); if (!e.empty()) c->fft_threads = atoi(e.c_str()); }
            if (c->fuse_win > 0 && !p0fft_geometry(lg, (int)Lt, &c->fuse_win, &c->fuse_vout)) { c->fuse_filter = false; c->fuse_win = c->fuse_vout = 0; }   // (a transform size forced by hand)
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
        // the digital AGC behind a post-resample filter on the overlap-save path: gain and per-chunk peaks in k_fftconv16's epilogue past
        // the lock.  A block of the filter must not hold more than one chunk boundary: a chunk's outputs (less the block quantisation's
        // slack) have to outnumber a workgroup's
        if (c->agc && c->agc_rms_alpha == 0.0f && c->decim && !c->late && !c->force_generic && c->fp.enabled && c->fp.post_resample && c->d_hfreq &&
            fftconv_agc_fusable(c->fft_log2n, (int)c->fp.taps.size(), c->dbg)) {
            const double chunk_out = (double)c->agc_chunk * (double)c->ratio - (double)c->fp.block - 2.0;
            c->agc_fusable_filter = chunk_out >= (double)(1 << c->fft_log2n);
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
    (void)agc_resolve_pending(c);
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
            if (n) memcpy(filter_taps_re_im, c->fp.taps.data(), n * sizeof(cfloat));     // (a chain without a filter: data() of an empty vector is null, and memcpy's arguments are declared non-null even for 0 bytes -- UBSan, round 6)
        }
        if (hb_taps) {
            size_t o = 0;
            for (int i = 0; i < c->rp.S; ++i)
                for (float v : c->rp.stages[(size_t)i].proto) { if (o < cap_hb) hb_taps[o] = v; ++o; }
        }
        if (arb_proto && c->resample) {
            const size_t n = c->rp.arb_proto.size() < cap_arb ? c->rp.arb_proto.size() : cap_arb;
            if (n) memcpy(arb_proto, c->rp.arb_proto.data(), n * sizeof(float));
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
    if (re_im && n && cap_taps) memcpy(re_im, c->fp.taps.data(), (n < cap_taps ? n : cap_taps) * sizeof(cfloat));
    return (int)n;
}
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
    { const int rc = agc_resolve_pending(c); if (rc && !c->poisoned) return rc; }        // ... and what their last fused launch owes
    c->pend.valid = false;
    c->poisoned = false;
    c->rem = 0; c->phi = 0; c->nco_theta = 0; c->pnco_theta = 0;
    c->agc_locked_host = false; c->agc_seen_host = 0; c->agc_peak_clean = false; c->agc_rms_pos = 0;
    HIP_TRY(hipMemsetAsync(c->d_dc_state, 0, sizeof(cd2), c->stream));
    if (c->agc) { // agc_reset, src/agc.c:224-238
        c->agc_init.last_strong = c->desc.agc_clock == IQGPU_AGC_CLOCK_WALL ? monotonic_sec() : 0.0;
        HIP_TRY(hipMemcpyAsync(c->d_agc_state, &c->agc_init, sizeof(AgcState), hipMemcpyHostToDevice, c->stream));
        // the verifier's words too: a classify launch lost to a device error would leave its ticket counter ([4]) non-zero, no
        // workgroup of the next launch would then be "the last", and the stale verdict in [0] would stand for ever
        HIP_TRY(hipMemcpyAsync(c->d_agc_flag, kAgcFlagInit, sizeof(kAgcFlagInit), hipMemcpyHostToDevice, c->stream));
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
    { const int rc = agc_resolve_pending(c); if (rc) return rc; }
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
    { const int rc = agc_resolve_pending(c); if (rc) return rc; }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return IQGPU_OK;
}
extern "C" void *iqgpu_chain_get_stream(const iqgpu_chain *c) { return c ? (void *)c->stream : nullptr; }
extern "C" int iqgpu_chain_synchronize(iqgpu_chain *c)
{
    if (!c) return fail(IQGPU_EINVAL, "NULL chain");
    HIP_TRY(hipSetDevice(c->device));
    { int rc = pipe_advance(c, c->pipe_seq); if (!rc) rc = pipe_drain(c, c->pipe_seq); if (!rc) rc = agc_resolve_pending(c); if (rc) return rc; }
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
    // a fused launch may still owe its fallback: launched (and timed) inside THIS window, as iqgpu_chain_synchronize does
    { const int rc = agc_resolve_pending(c); if (rc) return rc; }
    drain_events(c);
    *p = c->prof;
    memset(&c->prof, 0, sizeof(c->prof));
    return IQGPU_OK;
}

extern "C" const char *iqgpu_chain_front_kernel(const iqgpu_chain *c) { return c ? c->front_kernel : ""; }

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
    if (hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return fail(IQGPU_ENOMEM,
        "hipHostMalloc(%zu) failed", bytes);
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
