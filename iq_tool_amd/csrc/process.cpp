// process.cpp -- one process() call: every buffer sized first, then the stages in stream order
//   [dc carries] -> front (k_front | k_front_s1 | k_front_mid | k_cascade + k_front_s1) -> [filter] -> [k_interp] -> [agc]
#include "chain.hpp"

// ------------------------------------------------------------------------------------------------
// profiling helpers
// ------------------------------------------------------------------------------------------------
hipEvent_t get_event(iqgpu_chain *c)
{
    if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void drain_events(iqgpu_chain *c)
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
        // (k_p0fft16 computes the resampler's outputs inside the filter kernel: only the buffer front exists)
        const size_t need = (L1 + (size_t)c->fpending + (fusef ? 0 : (size_t)p.n_res) + 1) * sizeof(cf2);
        int rc = c->fbuf[c->fcur].ensure_keep(need, (L1 + (size_t)c->fpending) * sizeof(cf2), c->stream);
        if (rc) return rc;
        fcur = (cf2 *)c->fbuf[c->fcur].p;
    }
    if (c->late) {
        int rc = c->ibuf[c->icur].ensure_keep(((size_t)c->ihist + (size_t)p.n_x + 1) * sizeof(cf2), (size_t)c->ihist * sizeof(cf2),
            c->stream);
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
        // the intermediate stream between k_cascade and the last stage -- or, with both in k_front_s2, a few KB per edge wave
        const int64_t n_mid = s2 ? front_s2_mid_samples(cplan) : ((int64_t)rem_k + (int64_t)frames_in) >> casc_K;
        int rc = c->mid.ensure(((size_t)n_mid + 8) * sizeof(cf2)); if (rc) return rc;
    }
    // run descriptors of k_front_mid: a fresh array starts exhausted (all zero), a used one is left exhausted by every launch
    if (mid && (c->steal || fixed_tpw() != 0)) {
        const void *was = c->steal_buf.p;
        const size_t slots = (size_t)(fixed_tpw() != 0 ? wave_slots(front_mid_waves()) + cplan.w_n_edge : cplan.w_n_edge + cplan.w_n_stream) + 64;
        int rc = c->steal_buf.ensure(slots * (size_t)c->steal_stride * sizeof(unsigned long long)); if (rc) return rc;
        if (c->steal_buf.p != was && hipMemsetAsync(c->steal_buf.p, 0, c->steal_buf.cap, c->stream) != hipSuccess)
            return fail(IQGPU_EHIP, "hipMemsetAsync failed");
    }
    if (c->agc && c->agc_rms_alpha > 0.0f) {
        int64_t chunk, warm; int32_t n_chunks;
        agc_rms_geometry(c->agc_rms_alpha, (int64_t)c->agc_rms_pos, p.n_emit, &chunk, &warm, &n_chunks);
        int rc = c->agc_gain.ensure((size_t)(n_chunks > 0 ? n_chunks : 1) * 4 * sizeof(float)); if (rc) return rc;
    } else if (c->agc) {
        const AgcGeom g = agc_geom();
        if (agc_out_end(g, g.n_chunks - 1) != p.n_emit) return fail(IQGPU_EINVAL, "internal: AGC chunk map disagrees with the call plan");
        { const void *was = c->agc_peak.p;
          int rc0 = c->agc_peak.ensure((size_t)g.n_chunks * sizeof(unsigned long long)); if (rc0) return rc0;
          if (c->agc_peak.p != was) c->agc_peak_clean = false; }
        int rc = c->agc_gain.ensure((size_t)g.n_chunks * (sizeof(float) + sizeof(int32_t) + sizeof(int64_t)) + 2 * ((size_t)g.n_chunks / 256 + 2) * sizeof(int32_t)); if (rc) return rc;
        if (agc_fused) {   // what the fallback launches need, should the verifier reject the fused pass
            rc = c->agc_peak_b.ensure((size_t)g.n_chunks * sizeof(unsigned long long)); if (rc) return rc;
            rc = c->abuf.ensure(((size_t)p.n_emit + 1) * sizeof(cf2)); if (rc) return rc;
        }
    }
    return IQGPU_OK;
}

int Call::stage_front()
{
    if (fusef) {
        // resampler and filter run as ONE kernel, launched by stage_filter (which also turns the history buffers over)
        snprintf(c->front_kernel, sizeof(c->front_kernel), "k_p0fft16");
        return IQGPU_OK;
    }
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
    char mid_name[48] = "k_front_mid";
    bool casc2 = false;

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
        a1.casc_wave_lds = s2 ? (int)cascade_wave_lds(a1) : cplan.casc_wave_lds;      // (the plan's: sized for k_cascade2 where the call takes it)
        a1.out_fmt = IQGPU_FMT_CF32; a1.pnco_mode = 0;
        if (s2) a1.w_total_tiles = s2_in_tiles;          // (the fused kernel is planned in the last stage's tiles: cplan is a2's)
        else {
            copy_plan(a1);
            casc2 = cascade2_applies(a1);                // (launch_cascade's own test: raw cu8 frames, long runs -- cascade2.hip)
            KernelTimer kt(c, IQGPU_K_CASCADE); HIP_TRY(launch_cascade(a1, c->stream));
        }
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
            a2.agc_fused = front_fused() ? 1 : 0;
            if (s2) copy_plan(a2);
            else plan_front_s1(a2, wave_slots(front_s1_waves(a2)), fixed_tpw(), 1, 1);
            for (int q = 0; q < 20; ++q) a2.hb0[q] = 0.5f * c->rp.stages[(size_t)K].branch[(size_t)q];
            a2.sink = c->d_sink;
            if (front_fused()) {
                a2.agc_fused = 1; a2.agc_state = c->d_agc_state; a2.agc_peak2 = (unsigned long long *)c->agc_peak.p;
                a2.agc_chunk_frames = c->agc_chunk; a2.agc_shift = c->S; a2.agc_rem = c->rem;
                HIP_TRY(clean_agc_peaks());
            }
            { KernelTimer kt(c, IQGPU_K_FRONT); HIP_TRY(s2 ? launch_front_s2(a1, a2, c->stream) : launch_front_s1(a2, c->stream)); }
            if (front_fused()) { const int rc = stage_agc_verify_and_fallback(a2); if (rc) return rc; }
            c->hist2_cur ^= 1;
        }
    } else if (fast_s1) {
        // wave-autonomous kernel: one half-band stage (m = 10), or none
        copy_plan(a);
        if (!fast_s0) for (int q = 0; q < 20; ++q) a.hb0[q] = 0.5f * c->rp.stages[0].branch[(size_t)q];
        if (front_fused()) {
            a.agc_fused = 1; a.agc_state = c->d_agc_state; a.agc_peak2 = (unsigned long long *)c->agc_peak.p;
            a.agc_chunk_frames = c->agc_chunk; a.agc_shift = c->S; a.agc_rem = c->rem;
            HIP_TRY(clean_agc_peaks());
        }
        // k_front_mid, IQGPU_STEAL=1, one run per resident wave: every wave claims its tiles through its run descriptor and waves
        // that finish early split the runs of those that are behind (front_mid.hip; off by default, profiles/r04_steal.md)
        if (mid && c->steal && fixed_tpw() == 0 && a.w_n_stream >= 64 * (int64_t)front_mid_waves()) {
            a.w_steal = (unsigned long long *)c->steal_buf.p; a.w_steal_min = c->steal_min;
            a.w_steal_stride = c->steal_stride; a.w_steal_lanes = c->steal_lanes; a.w_steal_rounds = c->steal_rounds;
        } else if (mid && front_mid_nl(a) == 6 && fixed_tpw() != 0 && a.w_n_stream > wave_slots(front_mid_waves()) - a.w_n_edge) {
            // (the multi-run instantiation exists for six outputs per lane only: the 8-per-lane experiment, IQGPU_MID8=1, keeps one
            //  static run per wave over as many rounds of workgroups as the runs need -- ADVICE r4)
            // block_samples != 0 (BASELINE configs[1] as worded: "256 k-sample blocks"): more fixed-length runs than resident waves --
            // one round of workgroups, every streaming wave takes runs s, s + stride, ... through the multi-run instantiation
            a.w_steal = (unsigned long long *)c->steal_buf.p; a.w_steal_min = c->steal_min;
            a.w_steal_stride = c->steal_stride; a.w_steal_lanes = c->steal_lanes; a.w_steal_rounds = 0;
            a.w_run_stride = wave_slots(front_mid_waves()) - a.w_n_edge;
        }
        const int mid_nl = mid ? front_mid_nl(a) : 0;
        a.tap_fold = (uint32_t)(fat ? c->tap_fold8 : mid ? (mid_nl == 8 ? c->tap_fold8 : c->tap_fold6) : 0);
        // (k_front_p0: consecutive lanes are five outputs apart -- for the NRSC-5 step 16 arms -- so the lanes that re-read a slot in
        //  the same step would meet in two bank pairs of a linear plane: always the folded placement)
        if (p0) a.tap_fold = 1u;
        if (mid) {
            const bool b8 = a.in_fmt == IQGPU_FMT_CU8 || a.in_fmt == IQGPU_FMT_CS8 || a.out_fmt == IQGPU_FMT_CU8 || a.out_fmt == IQGPU_FMT_CS8;
            const bool n16 = (a.in_fmt == IQGPU_FMT_CS16 && a.gain != 1.0f) || a.in_fmt == IQGPU_FMT_SC16Q11;
            snprintf(mid_name, sizeof(mid_name), "k_front_mid<%d,%s%s%s%s>", mid_nl, c->nco_mode ? "nco" : "nonco", a.out_fmt == IQGPU_FMT_CF32 ? ",cf32" : "",
                     b8 ? ",8bit" : "", n16 ? ",gain" : "");
        }
        { KernelTimer kt(c, IQGPU_K_FRONT); HIP_TRY(fat ? launch_front_fat(a, c->stream) : mid ? launch_front_mid(a, c->stream)
            : p0 ? launch_front_p0(a, c->stream) : launch_front_s1(a, c->stream)); }
        if (front_fused()) { const int rc = stage_agc_verify_and_fallback(a); if (rc) return rc; }
    } else {
        KernelTimer kt(c, IQGPU_K_FRONT);
        HIP_TRY(launch_front(a, n_blocks, c->stream));
    }
    if (c->decim) c->hist_cur ^= 1;
    snprintf(c->front_kernel, sizeof(c->front_kernel), "%s",
             (casc && s2) ? "k_front_s2" : casc2 ? "k_cascade2+k_front_s1" : casc ? "k_cascade+k_front_s1" : fat ? "k_front_fat" : mid ? mid_name : p0 ? "k_front_p0"
             : fast_s1 ? "k_front_s1" : c->late ? "k_front+k_interp" : "k_front");
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
    // next call's buffer front: history (L-1) + still-pending samples, into the other buffer of the pair -- by the filter kernel's
    // last workgroup when one is launched (nobody reads that buffer meanwhile), else by a copy kernel
    const size_t keep = L1 + (size_t)p.fpending_next;
    const bool fused_move = n_filt > 0 && !(c->dbg & kDbgNoFusedMove);
    if (fused_move) { fa.move_dst = (cf2 *)c->fbuf[c->fcur ^ 1].p; fa.move_src = fcur + n_filt; fa.move_n = (int64_t)keep; }
    if (c->d_hfreq) {
        FftConvArgs ca{};
        ca.dbg = c->dbg;
        ca.fbuf = fcur; ca.fbuf_len = (int64_t)(L1 + (size_t)c->fpending + (size_t)p.n_res);
        ca.hfreq = c->d_hfreq; ca.twiddle = c->d_twiddle; ca.ntaps = fa.ntaps;
        ca.log2n = c->fft_log2n; ca.threads = c->fft_threads; ca.n_emit = n_filt;
        ca.pnco_mode = fa.pnco_mode; ca.pnco_theta0 = fa.pnco_theta0; ca.pnco_dtheta = fa.pnco_dtheta; ca.nco_tab = fa.nco_tab;
        ca.out_fmt = fa.out_fmt; ca.out = fa.out;
        ca.move_dst = fa.move_dst; ca.move_src = fa.move_src; ca.move_n = fa.move_n;
        if (fusef) {
            // k_p0fft16: the window fill needs what the front kernel would have been given
            if (!fused_move) return fail(IQGPU_EINVAL, "internal: the fused filter path needs the history move in the kernel");
            P0Feed &f = ca.feed;
            f.raw = d_raw_in; f.hist_in = c->d_hist[c->hist_cur]; f.hist_out = c->d_hist[c->hist_cur ^ 1];
            f.frames_in = (int64_t)frames_in; f.hist_cap = c->hist_cap; f.in_fmt = c->desc.in_format;
            f.arb_table = c->d_arb; f.step = c->rp.step; f.tap_fold = 1u; f.phi0 = c->phi;
            f.n_res = p.n_res; f.pre = (int64_t)(L1 + (size_t)c->fpending);
            // outputs whose 24-frame window load lies inside this call's input: q_k = (phi0 + k step) >> 24 in [13, frames_in - 11]
            auto first_k = [&](int64_t pos) {
                if (pos <= 0) return (int64_t)0;
                const uint64_t target = (uint64_t)pos << 24;
                return (int64_t)(target > c->phi ? (target - c->phi + (uint64_t)c->rp.step - 1) / (uint64_t)c->rp.step : 0);
            };
            f.k_a = first_k(13); f.k_b = first_k((int64_t)frames_in - 10);
            if (f.k_b > f.n_res) f.k_b = f.n_res;
            if (f.k_a > f.k_b) f.k_a = f.k_b;
            f.write_state = 1;
            f.grid = c->n_cu * (c->fft_log2n <= 12 ? 2 : 1);
            ca.win = c->fuse_win; ca.vout = c->fuse_vout;
        } else if (c->fuse_win > 0) {
            // the calls of such a chain that stay on the two kernels (short ones) run the fused kernel's windows: the
            // chain's bytes do not depend on which kernel a call took
            ca.win = c->fuse_win; ca.vout = c->fuse_vout;
        }
        if (agc_fused) {       // (past the lock, the filter between the resampler and the AGC: gain and per-chunk peaks in its epilogue)
            ca.agc_fused = 1; ca.agc_state = c->d_agc_state; ca.agc_peak2 = (unsigned long long *)c->agc_peak.p; ca.agc_geom = agc_geom();
            HIP_TRY(clean_agc_peaks());
        }
        { KernelTimer kt(c, IQGPU_K_FILTER); HIP_TRY(launch_fftconv(ca, c->stream)); }
        if (agc_fused) { const int rc = stage_agc_verify_and_fallback_filter(ca); if (rc) return rc; }
    } else {
        KernelTimer kt(c, IQGPU_K_FILTER);
        HIP_TRY(launch_fir(fa, c->stream));
    }
    if (!fused_move) {
        KernelTimer kt(c, IQGPU_K_MOVE);
        HIP_TRY(launch_copy_cf((cf2 *)c->fbuf[c->fcur ^ 1].p, fcur + n_filt, (int64_t)keep, c->stream));
    }
    c->fcur ^= 1;
    c->fpending = p.fpending_next;
    if (fusef && c->decim) c->hist_cur ^= 1;               // (what stage_front does behind its launch)
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
    // the next call's history (the last ihist resampler inputs) into the other buffer of the pair: by k_interp's last workgroup
    // when it is launched, else by a copy kernel
    const bool fused_move = p.n_emit > 0 && !(c->dbg & kDbgNoFusedMove);
    ia.move_dst = nullptr; ia.move_src = nullptr; ia.move_n = 0;
    if (fused_move) { ia.move_dst = (cf2 *)c->ibuf[c->icur ^ 1].p; ia.move_src = icur + p.n_x; ia.move_n = (int64_t)c->ihist; }
    { KernelTimer kt(c, IQGPU_K_FRONT); HIP_TRY(launch_interp(ia, c->n_cu, c->stream)); }
    if (!fused_move) {
        KernelTimer kt(c, IQGPU_K_MOVE);
        HIP_TRY(launch_copy_cf((cf2 *)c->ibuf[c->icur ^ 1].p, icur + p.n_x, (int64_t)c->ihist, c->stream));
    }
    c->icur ^= 1;
    return IQGPU_OK;
}
extern "C" int iqgpu_chain_process_device(iqgpu_chain *c, const void *d_raw_in, size_t frames_in,
                                          void *d_out, size_t out_capacity_bytes, size_t *frames_out)
{
    if (!c || !frames_out) return fail(IQGPU_EINVAL, "iqgpu_chain_process_device: NULL argument");
    int rc = pipe_advance(c, c->pipe_seq); if (rc) return rc;     // batches submitted earlier come first (same stream)
    rc = agc_resolve_pending(c); if (rc) return rc;               // ... with whatever their last fused launch still owes
    return process_device_impl(c, d_raw_in, frames_in, d_out, out_capacity_bytes, frames_out);
}
static int process_one(iqgpu_chain *c, const void *d_raw_in, size_t frames_in,
                       void *d_out, size_t out_capacity_bytes, size_t *frames_out, bool agc_fused);

int process_device_impl(iqgpu_chain *c, const void *d_raw_in, size_t frames_in,
                               void *d_out, size_t out_capacity_bytes, size_t *frames_out)
{
    if (!c || !frames_out) return fail(IQGPU_EINVAL, "iqgpu_chain_process_device: NULL argument");
    if (!(c->agc_fusable || c->agc_fusable_filter) || frames_in == 0) return process_one(c, d_raw_in, frames_in, d_out, out_capacity_bytes, frames_out, false);
    // output AGC on the specialised front kernel: the scanning phase (and the chunk that locks) through the unfused
    // kernels, everything behind it fused
    *frames_out = 0;
    bool locks = false;
    size_t head = agc_unfused_head(c, frames_in, &locks);
    // with a user filter between the resampler and the AGC the call that holds the lock stays whole: cutting it at the locking chunk
    // would move the filter's overlap-save windows on the stream (other roundings: the bytes of the unfused path would be missed by a
    // code here and there); the fused epilogue starts with the next call
    if (c->agc_fusable_filter && head > 0 && head < frames_in) head = frames_in;
    const size_t ibps = bytes_per_frame(c->desc.in_format), obps = bytes_per_frame(c->desc.out_format);
    if ((size_t)plan_call(c, frames_in).n_emit * obps > out_capacity_bytes)
        return fail(IQGPU_ECAPACITY, "output buffer too small: need %zu bytes, have %zu", (size_t)plan_call(c, frames_in).n_emit * obps,
            out_capacity_bytes);
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
    if (c->poisoned) return fail(IQGPU_EHIP,
        "an earlier call failed half way through: the stream state is undefined until iqgpu_chain_reset()");
    HIP_TRY(hipSetDevice(c->device));
    // a fused launch of an earlier call may still owe its fallback: before anything of this call is queued behind it
    { const int prc = agc_resolve_pending(c); if (prc) return prc; }

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
        // (dx / local: the AGC's input of earlier calls stands in front of this call's, see stage_agc)
        const size_t lead = c->agc_rms_alpha > 0.0f ? (size_t)c->agc_rms_warm : 0;
        int rc = c->abuf.ensure((lead + (size_t)k.p.n_emit + 1) * sizeof(cf2)); if (rc) return rc;
        k.fin_out = (cf2 *)c->abuf.p + lead; k.fin_fmt = IQGPU_FMT_CF32;
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
    // the host waits for this call's bytes anyway: the AGC verdict is read here and the fallback launched only when it is set
    c->defer_fallback = true;
    rc = process_device_impl(c, c->stage_in.p, frames_in, c->stage_out.p, c->stage_out.cap, &produced);
    c->defer_fallback = false;
    if (rc) return rc;
    rc = agc_resolve_pending(c); if (rc) return rc;
    if (produced) HIP_TRY(hipMemcpyAsync(out, c->stage_out.p, produced * obps, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *frames_out = produced;
    return IQGPU_OK;
}
