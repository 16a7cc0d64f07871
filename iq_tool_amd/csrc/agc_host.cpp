// agc_host.cpp -- host side of the output AGC (src/agc.c, src/post_processor.c:55-57): the chunk map of a call, which part of it
// runs fused in the front kernel, the verifier and the conditional fallback launches behind a fused launch.
#include "chain.hpp"

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
    ga.chunk_b = (int64_t *)((char *)c->agc_gain.p + (size_t)g.n_chunks * 8);                 // (behind gain[] and chunk_len[]: 8-byte aligned)
    ga.wg_last = (int32_t *)((char *)c->agc_gain.p + (size_t)g.n_chunks * 16);
    ga.wg_pend = ga.wg_last + ((size_t)g.n_chunks / 256 + 2);
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
        if (p.n_emit <= 0) return IQGPU_OK;
        // abuf = [the last agc_rms_warm samples of earlier calls][this call's n_emit]: chunks of the stream's grid that begin early
        // in this call warm up on the samples in front of it
        const int64_t W = c->agc_rms_warm;
        cf2 *const xin = (cf2 *)c->abuf.p + W;
        AgcRmsArgs ra{};
        ra.x = xin; ra.n = p.n_emit; ra.alpha = c->agc_rms_alpha; ra.state = c->d_agc_state;
        ra.pos0 = (int64_t)c->agc_rms_pos; ra.hist_valid = ra.pos0 < W ? ra.pos0 : W;
        agc_rms_geometry(ra.alpha, ra.pos0, ra.n, &ra.chunk, &ra.warm, &ra.n_chunks);
        int rc = c->agc_gain.ensure((size_t)(ra.n_chunks > 0 ? ra.n_chunks : 1) * 4 * sizeof(float)); if (rc) return rc;
        ra.st = (float *)c->agc_gain.p;
        ra.out_fmt = c->desc.out_format; ra.out = d_out;
        KernelTimer kt(c, IQGPU_K_AGC);
        HIP_TRY(launch_copy_cf((cf2 *)c->abuf.p, (const cf2 *)c->agc_hist.p, W, c->stream));
        HIP_TRY(launch_agc_rms(ra, c->stream));
        HIP_TRY(launch_copy_cf((cf2 *)c->agc_hist.p, xin + p.n_emit - W, W, c->stream));     // the last W of [history | new]
        c->agc_rms_pos += (uint64_t)p.n_emit;
        return IQGPU_OK;
    }
    const AgcArgs ga = agc_args();
    KernelTimer kt(c, IQGPU_K_AGC);
    c->agc_peak_clean = false;                       // (k_agc_peak leaves its maxima in agc_peak)
    HIP_TRY(launch_agc(ga, c->stream));
    return IQGPU_OK;
}

// behind a fused front launch: the verifier, then the unfused kernels -- same input, same history buffers, the untouched AGC
// state -- either queued right behind it as launches that do nothing unless the verifier raised its flag (iqgpu_chain_process_device:
// the caller owns the stream, nothing may be left for later), or kept here until the host has read the verdict from its pinned word
// (defer_fallback: iqgpu_chain_process and submit / collect, where the host waits for the call anyway; agc_resolve_pending)
static hipError_t launch_agc_fallback(iqgpu_chain *c, const FrontArgs &fb, const AgcArgs &ga)
{
    hipError_t e = launch_front_s1(fb, c->stream);
    if (e != hipSuccess) return e;
    return launch_agc(ga, c->stream);
}

int Call::stage_agc_verify_and_fallback(const FrontArgs &spec)
{
    const bool defer = c->defer_fallback && c->h_agc_verdict != nullptr;
    AgcArgs va = agc_args();
    va.verify_flag = c->d_agc_flag;
    va.peak_approx = (mid || p0) ? 1 : 0;
    va.peak2_fallback = (unsigned long long *)c->agc_peak_b.p;
    va.verdict_host = defer ? c->d_agc_verdict : nullptr;
    KernelTimer kt(c, IQGPU_K_AGC);
    if (defer) c->h_agc_verdict[0] = -1;               // (written before the launch is queued: the kernel's store comes later)
    HIP_TRY(launch_agc_verify(va, c->stream));
    FrontArgs fb = spec;
    fb.agc_fused = 0; fb.agc_state = nullptr; fb.agc_peak2 = nullptr; fb.w_steal = nullptr; fb.w_run_stride = 0;
    fb.out_fmt = IQGPU_FMT_CF32; fb.out = c->abuf.p;
    fb.run_if = c->d_agc_flag;
    if (fat || mid || p0) {
        // the fused launch ran on k_front_fat / k_front_mid / k_front_p0 with its own geometry: the fallback is k_front_s1's (512-frame
        // tiles; 256 without a half-band stage)
        const int ft = p0 ? 256 : kWTile;
        fb.w_total_tiles = ((int64_t)fb.rem0 + fb.frames_in + ft - 1) / ft;
        int warm = (int)((c->rp.history_in + ft - 1) / ft);
        if (warm < 1) warm = 1;
        plan_front_s1(fb, wave_slots(front_s1_waves(fb)), fixed_tpw(), warm, 1, ft);
    }
    AgcArgs ga = va;
    ga.peak2_fallback = nullptr; ga.verdict_host = nullptr;
    ga.peak2 = (unsigned long long *)c->agc_peak_b.p;
    ga.run_if = c->d_agc_flag; ga.verify_flag = nullptr;
    if (defer) {
        c->pend.valid = true; c->pend.filter = false; c->pend.fb = fb; c->pend.ga = ga;
        return IQGPU_OK;
    }
    HIP_TRY(launch_agc_fallback(c, fb, ga));
    return IQGPU_OK;
}

// ... the same behind a filter launch whose epilogue applied the gain: the fallback is that launch again with cf32 output into the
// AGC's buffer (no history move: the fused launch made it) and the unfused AGC kernels
int Call::stage_agc_verify_and_fallback_filter(const FftConvArgs &spec)
{
    const bool defer = c->defer_fallback && c->h_agc_verdict != nullptr;
    AgcArgs va = agc_args();
    va.verify_flag = c->d_agc_flag;
    va.peak_approx = 1;                                  // float peaks, as k_front_mid's
    va.peak2_fallback = (unsigned long long *)c->agc_peak_b.p;
    va.verdict_host = defer ? c->d_agc_verdict : nullptr;
    KernelTimer kt(c, IQGPU_K_AGC);
    if (defer) c->h_agc_verdict[0] = -1;
    HIP_TRY(launch_agc_verify(va, c->stream));
    FftConvArgs fc = spec;
    fc.agc_fused = 0; fc.agc_state = nullptr; fc.agc_peak2 = nullptr;
    fc.out_fmt = IQGPU_FMT_CF32; fc.out = c->abuf.p;
    fc.move_dst = nullptr; fc.move_src = nullptr; fc.move_n = 0;
    fc.feed.write_state = 0;                              // (k_p0fft16: the fused launch has left the next call's state)
    fc.run_if = c->d_agc_flag;
    AgcArgs ga = va;
    ga.peak2_fallback = nullptr; ga.verdict_host = nullptr;
    ga.peak2 = (unsigned long long *)c->agc_peak_b.p;
    ga.run_if = c->d_agc_flag; ga.verify_flag = nullptr;
    if (defer) {
        c->pend.valid = true; c->pend.filter = true; c->pend.fc = fc; c->pend.ga = ga;
        return IQGPU_OK;
    }
    HIP_TRY(launch_fftconv(fc, c->stream));
    HIP_TRY(launch_agc(ga, c->stream));
    return IQGPU_OK;
}

// The host's half of the deferred scheme: wait for k_agc_classify's verdict in the pinned word (the kernel is normally long done:
// the word is polled, no runtime call), and launch the fallback only when it is set.  Called by everything that needs the stream
// final or is about to queue work that touches the buffers the fallback reads: the next process call, the batch's D2H copy, reset,
// synchronize, get_agc_state, set_stream, destroy.
int agc_resolve_pending(iqgpu_chain *c, bool *ran)
{
    if (ran) *ran = false;
    if (!c->pend.valid) return IQGPU_OK;
    int32_t v = c->h_agc_verdict[0];
    if (v < 0) {
        const double t0 = monotonic_sec();
        while ((v = c->h_agc_verdict[0]) < 0) {
            if (monotonic_sec() - t0 > 0.25) {
                // a quarter of a second without an answer: let the runtime wait for the stream (and report a dead one)
                const hipError_t e = hipStreamSynchronize(c->stream);
                v = c->h_agc_verdict[0];
                if (e != hipSuccess || v < 0) {
                    c->pend.valid = false; c->poisoned = true;
                    return fail(IQGPU_EHIP, "the AGC verdict of a fused launch never arrived: %s", e != hipSuccess ? hipGetErrorString(e) : "stream idle, word unwritten");
                }
                break;
            }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
    }
    c->pend.valid = false;
    if (v == 0) return IQGPU_OK;
    KernelTimer kt(c, IQGPU_K_AGC);
    hipError_t e;
    if (c->pend.filter) { e = launch_fftconv(c->pend.fc, c->stream); if (e == hipSuccess) e = launch_agc(c->pend.ga, c->stream); }
    else e = launch_agc_fallback(c, c->pend.fb, c->pend.ga);
    if (e != hipSuccess) { c->poisoned = true; return fail(IQGPU_EHIP, "AGC fallback launch failed: %s", hipGetErrorString(e)); }
    // The fallback REWRITES the output of the launch it belongs to.  When that launch was a submitted batch whose D2H copy is still
    // to come (the copy runs on another stream and waits for the batch's "kernels done" event only), the event moves behind the
    // fallback HERE -- whoever asked for the verdict: the next batch's launch, process_device() or reset() running behind submitted
    // batches (those two left the old event standing and a later collect() copied bytes the fallback was still writing)
    if (c->pipe_ready && c->pipe_launched > c->pipe_copied) {
        const hipError_t er = hipEventRecord(c->pipe[(c->pipe_launched - 1) % iqgpu_chain::kPipeSlots].k_done, c->stream);
        if (er != hipSuccess) { c->poisoned = true; return fail(IQGPU_EHIP, "hipEventRecord failed: %s", hipGetErrorString(er)); }
    }
    if (ran) *ran = true;
    return IQGPU_OK;
}

// first frame count (a multiple of the AGC chunk, or the whole call) that must take the unfused AGC path: everything
// while the stream has not locked.  agc_apply locks on the first chunk that STARTS after AGC_DIGITAL_LOCK_TIME of
// output (src/agc.c:151-155: elapsed = samples_seen / rate before this chunk is counted), a closed form of the
// stream position; sets *locks when that chunk lies in this call.
size_t agc_unfused_head(const iqgpu_chain *c, size_t frames_in, bool *locks)
{
    *locks = false;
    if (c->agc_locked_host) return 0;
    AgcGeom g{};
    g.frames_in = (int64_t)frames_in; g.chunk_frames = c->agc_chunk;
    g.n_chunks = (int)(((int64_t)frames_in + c->agc_chunk - 1) / c->agc_chunk);
    g.mode = 1; g.rem = c->rem; g.S = c->S; g.phi = c->phi; g.step = c->rp.step;
    if (c->fp.enabled && c->fp.block) { g.block = c->fp.block; g.fpending = c->fpending; }     // (a filter behind the resampler emits whole blocks)
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
