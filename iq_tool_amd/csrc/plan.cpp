// plan.cpp -- where a call starts and ends in the stream (closed forms, SPEC B.6) and how its tiles are dealt out to the
// wave-autonomous kernels.
#include "chain.hpp"

CallPlan plan_call_at(const iqgpu_chain *c, const StreamPos &at, size_t frames_in)
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
        if (span > at.phi) { p.n_arb = (int64_t)((span - at.phi + step - 1) / step); p.phi_next
            = at.phi + (uint64_t)p.n_arb * step - span; }
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
        if (span > at.phi) { p.n_res = (int64_t)((span - at.phi + step - 1) / step); p.phi_next
            = at.phi + (uint64_t)p.n_res * step - span; }
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

CallPlan plan_call(const iqgpu_chain *c, size_t frames_in)
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
        cplan.agc_fused = front_fused() ? 1 : 0; cplan.agc_shift = c->S; cplan.agc_chunk_frames = c->agc_chunk;
        cplan.S = c->S; cplan.gain = c->desc.gain; cplan.iq_enable = c->desc.iq_correct_enable ? 1 : 0;
        cplan.dc_enable = c->dc ? 1 : 0; cplan.nco_mode = c->nco_mode;
        if (casc) {
            cplan.casc_K = casc_K;
            for (int k = 0; k < casc_K; ++k) cplan.m[k] = c->rp.stages[(size_t)k].m;
            cplan.casc_wave_lds = (int)cascade_wave_lds(cplan);      // (reads the pointwise switches above: cascade2_shape)
        }
        cplan.pnco_mode = (!filt && !c->late) ? c->pnco_mode : 0;
        cplan.step = c->rp.step;
        // the preset shape on a call long enough to give every one of the 8 x CUs fat waves a run of tiles: k_front_fat
        // (shorter calls keep k_front_s1's 16 x CUs waves of 512-frame tiles: what counts for them is latency; same bytes either way)
        const bool fat_ok = !casc && !fast_s0 && front_fat_shape(cplan) &&
              ((c->dbg & kDbgForceFat) || (int64_t)frames_in >= (int64_t)kFatMinTilesPerWave * kFatTile * wave_slots(front_fat_waves()));
        const int mid_nl = (!casc && !fast_s0) ? front_mid_nl(cplan) : 0;
        // (run descriptors hold tile indices in 32 bits)
        // (k_front_mid from 6 tiles per wave on: measured round 4 at 2^21 .. 2^25 frames, kernel ms k_front_s1 / k_front_mid:
        //  0.014 / 0.023, 0.017 / 0.023, 0.023 / 0.028, 0.039 / 0.036, 0.067 / 0.055 -- the crossover lies between 2^23 and 2^24 frames
        //  = 3.6 and 7.1 tiles per wave; the pipelined host path's 2^24-frame batches now run the headline kernel)
        constexpr int kMidMinTilesPerWave = 6;
        const bool mid_ok = mid_nl != 0 && (int64_t)frames_in < ((int64_t)1 << 40) &&
              ((c->dbg & kDbgForceFat)
                  || (int64_t)frames_in >= (int64_t)kMidMinTilesPerWave * front_mid_tile(mid_nl) * wave_slots(front_mid_waves()));
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
        plan_front_s1(cplan, wave_slots(casc ? cascade_waves(cplan) : fat ? front_fat_waves() : mid ? front_mid_waves()
            : front_s1_waves(cplan)),
                      ftpw, warm, mid_align, wtile, mid_align,
                      // (k_cascade2: a streaming run of an odd number of tiles starts one tile early -- that tile has to be loadable)
                      mid ? kMidLead : (casc && cascade2_shape(cplan)) ? kWTile : 0);
        // (a raw cascade whose call turns out too short for k_cascade2's two-tile trips: k_cascade's own slice -- more waves per CU -- and
        //  no lead)
        if (casc && cascade2_shape(cplan) && !cascade2_applies(cplan)) {
            cplan.casc_wave_lds = (int)cascade_wave_lds(cplan, false);
            plan_front_s1(cplan, wave_slots(cascade_waves(cplan)), ftpw, warm, mid_align, wtile, mid_align, 0);
        }
        // k_front_mid: the three waves of a SIMD get runs in proportion to the speed their age buys them (kernels.hpp, weight_runs)
        if (mid && ftpw == 0 && cplan.w_n_edge <= front_mid_max_edge_waves() && c->run_wt[0] > 0) weight_runs(cplan, front_mid_waves(),
            c->run_wt);
        if (mid && cplan.w_n_edge > front_mid_max_edge_waves()) {
            // (an unaligned buffer, a call that is all edges: k_front_mid keeps LDS for a handful of edge waves only)
            mid = false; wtile = kWTile;
            cplan.w_total_tiles = ((int64_t)rem_k + (int64_t)frames_in + wtile - 1) / wtile;
            warm = (int)((c->rp.history_in + wtile - 1) / wtile); if (warm < 1) warm = 1;
            plan_front_s1(cplan, wave_slots(front_s1_waves(cplan)), fixed_tpw(), warm, 1, wtile);
        }
        // S == 0 (the cu8-nrsc5 presets): k_front_p0 on calls long enough to give every one of its 8 x CUs waves a few steps of 320
        // outputs (shorter calls keep k_front_s1<S0>: same bytes).  Planned as k_front_s1's 256-frame tiles -- the edge runs are its
        // run_tiles -- with the streaming tiles' OUTPUTS dealt out as steps
        p0 = false;
        if (fast_s0 && !casc && !(c->dbg & kDbgNoP0)) {
            cplan.phi0 = c->phi;
            if (front_p0_shape(cplan) && ((int64_t)frames_in >= ((int64_t)1 << 22) || (c->dbg & kDbgForceFat))) {
                FrontArgs q = cplan;
                plan_front_s1(q, wave_slots(front_p0_waves()), 0, warm, front_p0_edge_tpw(), 256);
                if (q.w_n_edge <= front_p0_max_edge_waves() && q.w_edge_tb > q.w_edge_ta) {
                    plan_front_p0(q, wave_slots(front_p0_waves()));
                    if (q.w_n_stream > 0) { cplan = q; p0 = true; }
                }
            }
        }
        // ... and with a user filter behind the resampler on the overlap-save path: resampler AND filter in one kernel (k_p0fft16,
        // fftconv.hip, round 6) -- no front launch, no cf32 stream in HBM.  Long calls only (as k_front_p0); the next call's filter
        // front (history + pending samples) is recomputed by one workgroup, so it has to be short
        fusef = false;
        if (c->fuse_filter && filt && fast_s0 && !casc && p.n_emit > 0 && (int64_t)(L1 + p.fpending_next) <= kP0FftMaxKeep &&
            ((int64_t)frames_in >= ((int64_t)1 << 22) || (c->dbg & kDbgForceFat))) {
            fusef = true; p0 = false;
        }
                // S == 2: both stages in ONE kernel (k_front_s2, front_s2.hip), planned in tiles of the LAST stage -- 512 intermediate samples
        // = 1024 input frames -- on the intermediate stream's own geometry.  Its streaming waves read the input as whole 16-byte
        // words from the start of a decimation group; calls that do not start on one, or are shorter than the histories they have
        // to leave behind, keep the two kernels (same bytes either way).
        s2 = false;
        const int64_t n_mid = (int64_t)frames_in >> 1;
        if (casc && casc_K == 1 && c->rem == 0 && cplan.raw_aligned && !front_fused() && !(c->dbg & kDbgNoS2) && front_s2_shape(cplan) &&
            (int64_t)frames_in >= (int64_t)c->hist_cap && n_mid >= (int64_t)c->hist2_cap) {
            FrontArgs p2{};
            p2.frames_in = n_mid; p2.rem0 = 0; p2.hist_cap = c->hist2_cap; p2.in_fmt = IQGPU_FMT_CF32; p2.out_fmt = filt ? (int)IQGPU_FMT_CF32 : fin_fmt;
            p2.raw_aligned = 1;
            p2.w_total_tiles = (n_mid + kWTile - 1) / kWTile;
            plan_front_s1(p2, wave_slots(front_s2_waves()), fixed_tpw(), 1, 1);
            s2 = true;
            const FrontArgs keep = cplan;
            cplan = p2;                                    // the run geometry everything else reads (dc carries included)
            cplan.casc_K = keep.casc_K; cplan.m[0] = keep.m[0]; cplan.dbg = keep.dbg;
            s2_in_tiles = ((int64_t)frames_in + kWTile - 1) / kWTile;
            wtile = 2 * kWTile;                            // input frames per tile of the plan
            rem_k = 0;
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
        dg.edge_tpw = cplan.w_edge_tpw; dg.run_q = cplan.w_run_q; dg.run_r = cplan.w_run_r; dg.ta = cplan.w_edge_ta; dg.tb
            = cplan.w_edge_tb;
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
