// pipeline.cpp -- iqgpu_chain_submit / _collect: the chain fed from pinned host memory, kPipeSlots batches in flight.
#include "chain.hpp"

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
int pipe_advance(iqgpu_chain *c, uint64_t upto)
{
    if (!c->pipe_ready) return IQGPU_OK;
    while (c->pipe_launched < upto) {
        iqgpu_chain::PipeSlot &ps = c->pipe[c->pipe_launched % iqgpu_chain::kPipeSlots];
        HIP_TRY(hipSetDevice(c->device));
        int rc = IQGPU_OK;
        if (c->pend.valid && c->pipe_launched > 0) {
            // the batch launched last still owes its AGC verdict: read it before this one is queued behind it.  When the fallback
            // runs it rewrites that batch's output: agc_resolve_pending moves its "kernels done" event behind the fallback, so that
            // the D2H copy waits
            rc = agc_resolve_pending(c); if (rc) return rc;
        }
        if (ps.frames_in) {
            HIP_TRY(hipEventSynchronize(ps.in_done));
            size_t produced = 0;
            c->iq_pinned = true; c->iq_pin_mag = ps.iq_mag; c->iq_pin_phase = ps.iq_phase;
            // (the host orders every stage of a batch: the AGC verdict of a fused launch is read on the host -- by the next batch's
            //  launch or by this batch's D2H copy, whichever comes first -- and the fallback kernels are launched only when it is set)
            c->defer_fallback = true;
            rc = process_device_impl(c, ps.d_in.p, ps.frames_in, ps.d_out.p, ps.d_out.cap, &produced);
            c->defer_fallback = false;
            c->iq_pinned = false;
            if (!rc && produced != ps.n_emit) rc = fail(IQGPU_EHIP, "internal: batch produced %zu frames, planned %zu", produced,
                ps.n_emit);
            if (rc) ps.n_emit = 0;                               // nothing of a failed batch is copied back
        }
        ++c->pipe_launched;
        HIP_TRY(hipEventRecord(ps.k_done, c->stream));
        if (rc) return rc;
    }
    return IQGPU_OK;
}

// D2H copy of every launched batch up to ticket `upto`, on the D2H stream
int pipe_drain(iqgpu_chain *c, uint64_t upto)
{
    if (!c->pipe_ready) return IQGPU_OK;
    if (upto > c->pipe_launched) upto = c->pipe_launched;
    while (c->pipe_copied < upto) {
        iqgpu_chain::PipeSlot &ps = c->pipe[c->pipe_copied % iqgpu_chain::kPipeSlots];
        hipStream_t d2h = c->pipe_d2h[ps.n_emit * bytes_per_frame(c->desc.out_format) <= kSmallCopy
            ? c->pipe_copied % (uint64_t)iqgpu_chain::kCopyStreams : 0];
        HIP_TRY(hipSetDevice(c->device));
        if (ps.n_emit) {
            HIP_TRY(hipEventSynchronize(ps.k_done));
            if (c->pend.valid && c->pipe_copied + 1 == c->pipe_launched) {
                // the last batch launched is this one and its verdict is still out (no later launch has asked for it): read it now;
                // a fallback rewrites the batch's output, so the copy waits for it
                bool ran = false;
                const int vrc = agc_resolve_pending(c, &ran); if (vrc) return vrc;
                if (ran) HIP_TRY(hipStreamSynchronize(c->stream));
            }
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
    if (frames_in) { c->pipe_rem = plan.rem_next; c->pipe_phi = plan.phi_next; c->pipe_fpending = c->fp.enabled ? plan.fpending_next
        : at.fpending; }
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
    if (ticket == 0 || ticket > c->pipe_seq) return fail(IQGPU_EINVAL, "iqgpu_chain_collect: unknown ticket %llu",
        (unsigned long long)ticket);
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
