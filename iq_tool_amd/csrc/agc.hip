// agc.hip -- output AGC, "digital" profile: agc_apply of src/agc.c:85-222, which the reference runs
// once per chunk between the post NCO and convert_cf32_to_block (src/post_processor.c:55-57).
//
// The gain of a chunk depends on the chunk's OWN peak (look-ahead, agc.c:119-141), so the chain's
// last stage writes cf32 and three small kernels finish the job:
//   k_agc_peak   max |x|^2 per chunk (double: exact products, so sqrt gives cabsf's value),
//   k_agc_scan   one wavefront walks the chunks: scanning phase = prefix max (parallel), locked
//                phase = speculate "gain unchanged" over 64 chunks at a time and serialise only at the
//                chunks that ratchet or creep,
//   k_agc_apply  x * gain[chunk] -> pack.
// Chunk boundaries in the output come from agc_out_end() (kernels.hpp), the same closed form the
// host uses for frames_out.
#include <hip/hip_runtime.h>

#include "../../include/iqgpu.h"
#include "dsp_device.hpp"
#include "kernels.hpp"

namespace iqgpu {

// include/constants.h:184-192
constexpr float kAgcLockTime = 2.0f, kAgcHangTime = 4.0f, kAgcRecovery = 1.0005f, kAgcLower = 0.75f;

// (k_agc_peak / k_agc_apply walk their (chunk, split) pairs with a grid stride: as conditional fallback launches behind a fused
//  front kernel they are started with a small grid, so that the launch that finds nothing to do costs a couple of microseconds and
//  not the dispatch of 16384 x splits empty workgroups)
__global__ __launch_bounds__(kThreads) void k_agc_peak(const AgcArgs a)
{
    if (a.run_if && *a.run_if == 0) return;
    const int64_t n_items = (int64_t)a.geom.n_chunks * a.splits;
    for (int64_t w = blockIdx.x; w < n_items; w += gridDim.x) {
        const int c = (int)(w / a.splits), sy = (int)(w % a.splits);
        const int64_t b = agc_out_end(a.geom, (int64_t)c - 1), e = agc_out_end(a.geom, c);
        const int64_t len = e - b;
        // the scan walks the chunk lengths from this table instead of redoing the 64-bit divisions of the closed form
        if (sy == 0 && threadIdx.x == 0) a.chunk_len[c] = (int32_t)(len > 0 ? len : 0);
        if (len <= 0) continue;
        const int64_t per = (len + a.splits - 1) / a.splits;
        const int64_t lo = b + (int64_t)sy * per;
        int64_t hi = lo + per; if (hi > e) hi = e;
        double m = 0.0;
        for (int64_t i = lo + threadIdx.x; i < hi; i += kThreads) {
            const cf2 v = a.x[i];
            const double d = (double)v.x * (double)v.x + (double)v.y * (double)v.y;
            m = d > m ? d : m;
        }
#pragma unroll
        for (int k = 32; k >= 1; k >>= 1) { const double o = __shfl_xor(m, k); m = o > m ? o : m; }
        // non-negative doubles order like their bit patterns
        if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(a.peak2 + c, (unsigned long long)__double_as_longlong(m));
    }
}

__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src); }

// ---- wave-level helpers on the data-parallel-primitive path (a dependent chain of ds_bpermute shuffles
// costs ~250 cycles per step; the walk below is sequential, so that latency is its run time) ----
// value of lane `src` (wave-uniform index) as a scalar broadcast
__device__ __forceinline__ int rl_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ float rl_f(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
__device__ __forceinline__ double rl_d(double v, int src)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b & 0xffffffffll), src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// inclusive scans over the 64 lanes: row_shr 1, 2, 4, 8 inside each row of 16, then row_bcast 15 / 31
#define IQGPU_DPP(v, ctrl, rmask) __builtin_amdgcn_update_dpp(0, (v), (ctrl), (rmask), 0xf, true)
__device__ __forceinline__ int wave_scan_add(int v)
{
    v += IQGPU_DPP(v, 0x111, 0xf); v += IQGPU_DPP(v, 0x112, 0xf); v += IQGPU_DPP(v, 0x114, 0xf); v += IQGPU_DPP(v, 0x118, 0xf);
    v += IQGPU_DPP(v, 0x142, 0xa); v += IQGPU_DPP(v, 0x143, 0xc);
    return v;
}
__device__ __forceinline__ float wave_scan_max(float x)      // x >= 0
{
    int v = __float_as_int(x);                                // non-negative floats order like their bit patterns
    v = max(v, IQGPU_DPP(v, 0x111, 0xf)); v = max(v, IQGPU_DPP(v, 0x112, 0xf)); v = max(v, IQGPU_DPP(v, 0x114, 0xf));
    v = max(v, IQGPU_DPP(v, 0x118, 0xf)); v = max(v, IQGPU_DPP(v, 0x142, 0xa)); v = max(v, IQGPU_DPP(v, 0x143, 0xc));
    return __int_as_float(v);
}
#undef IQGPU_DPP

__global__ __launch_bounds__(64) void k_agc_scan(const AgcArgs a)
{
    if (a.run_if && *a.run_if == 0) return;
    const int lane = threadIdx.x;
    AgcState st = *a.state;
    const float target = a.target;
    int64_t end_prev = 0;
    // the batch's two table reads are issued one batch ahead: the walk is sequential, a global load per step
    // would put ~1 us of latency on each of its n_chunks / 64 steps
    int32_t len_nx = (lane < a.geom.n_chunks) ? a.chunk_len[lane] : 0;
    unsigned long long p2_nx = (lane < a.geom.n_chunks) ? a.peak2[lane] : 0ull;
    for (int c0 = 0; c0 < a.geom.n_chunks; c0 += 64) {
        const int c = c0 + lane;
        const bool valid = c < a.geom.n_chunks;
        const int32_t len_i = len_nx;
        const unsigned long long p2_i = p2_nx;
        {
            const int cn = c + 64;
            len_nx = (cn < a.geom.n_chunks) ? a.chunk_len[cn] : 0;
            p2_nx = (cn < a.geom.n_chunks) ? a.peak2[cn] : 0ull;
        }
        // chunk ends within the batch: inclusive prefix sum of the lengths k_agc_peak recorded
        const int len_v = valid ? len_i : 0;
        const int64_t e_i = end_prev + (int64_t)wave_scan_add(len_v);        // a batch holds < 2^31 samples
        const int64_t b_i = e_i - (int64_t)len_v;
        const bool active = valid && e_i > b_i;                   // empty chunks never reach agc_apply
        const uint64_t seen_i = st.seen + (uint64_t)(b_i - end_prev);
        const float pk = active ? (float)sqrt(__longlong_as_double((long long)p2_i)) : 0.0f;
        const double t_i = a.clock_wall ? a.t_wall : (double)seen_i / a.rate;
        float gain_i = st.gain;
        int cur = 0;

        if (!st.locked) {                                         // agc.c:117-160
            float run = fmaxf(wave_scan_max(pk), st.peak_memory);    // inclusive prefix max
            const bool lock_here = active && ((double)seen_i / a.rate > (double)kAgcLockTime);
            const unsigned long long lm = __ballot(lock_here);
            const int first = lm ? __ffsll((long long)lm) - 1 : 64;
            const float safe = run < 1e-4f ? 1e-4f : run;
            gain_i = target / safe;
            const int last = first < 64 ? first : 63;
            st.peak_memory = rl_f(run, last);
            if (first < 64) {
                st.locked = 1;
                st.gain = rl_f(gain_i, first);
                st.last_strong = rl_d(t_i, first);
            }
            cur = first + 1;
        }
        while (cur < 64) {                                        // agc.c:165-215, gain st.gain entering lane cur
            const float g = st.gain;
            const bool cand = active && lane >= cur;
            const float outp = pk * g;
            const bool ratchet = cand && outp > 1.0f;
            const bool healthy = cand && !ratchet && outp > target * kAgcLower;
            const unsigned long long hmask = __ballot(healthy);
            const bool weak = cand && !ratchet && !healthy;
            bool creep = false;
            if (__ballot(weak) != 0ull) {                         // (rare) some chunk is below the lower threshold
                const unsigned long long before = hmask & ((1ull << lane) - 1ull);
                double ls = st.last_strong;                       // last "strong" time seen by this chunk
                const int src = before ? 63 - __clzll((long long)before) : lane;
                const double t_src = shfl_d(t_i, src);
                if (before) ls = t_src;
                creep = weak && (t_i - ls > (double)kAgcHangTime);
            }
            const unsigned long long chg = __ballot(ratchet || creep);
            const int first = chg ? __ffsll((long long)chg) - 1 : 64;
            if (lane >= cur && lane < first) gain_i = g;
            const unsigned long long hm2 = first < 64 ? (hmask & ((1ull << first) - 1ull)) : hmask;
            if (hm2) st.last_strong = rl_d(t_i, 63 - __clzll((long long)hm2));
            if (first < 64) {
                const bool is_ratchet = (__ballot(ratchet) >> first) & 1ull;
                const float pk_f = rl_f(pk, first);
                if (is_ratchet) { st.gain = 0.99f / pk_f; st.last_strong = rl_d(t_i, first); }
                else st.gain = g * kAgcRecovery;
                if (lane == first) gain_i = st.gain;
            }
            cur = first + 1;
        }
        if (valid) a.gain[c] = gain_i;
        const int64_t e_last = end_prev + (int64_t)rl_i((int)(e_i - end_prev), 63);
        st.seen += (uint64_t)(e_last - end_prev);
        end_prev = e_last;
    }
    if (lane == 0) *a.state = st;
}

__global__ __launch_bounds__(kThreads) void k_agc_apply(const AgcArgs a)
{
    if (a.run_if && *a.run_if == 0) return;
    const int64_t n_items = (int64_t)a.geom.n_chunks * a.splits;
    for (int64_t w = blockIdx.x; w < n_items; w += gridDim.x) {
        const int c = (int)(w / a.splits), sy = (int)(w % a.splits);
        const int64_t b = agc_out_end(a.geom, (int64_t)c - 1), e = agc_out_end(a.geom, c);
        const int64_t len = e - b;
        if (len <= 0) continue;
        const int64_t per = (len + a.splits - 1) / a.splits;
        const int64_t lo = b + (int64_t)sy * per;
        int64_t hi = lo + per; if (hi > e) hi = e;
        const float g = a.gain[c];
        for (int64_t i = lo + threadIdx.x; i < hi; i += kThreads) {
            const cf2 v = a.x[i];
            pack_store(a.out, i, a.out_fmt, cf2{v.x * g, v.y * g});   // samples[i] *= gain (complex * real)
        }
    }
}

// ---- the fused path (front_wave.hip, k_front_s1<.., AGC>) -------------------------------------------------------
// In the locked phase agc_apply changes the gain only when a chunk ratchets (peak * g > 1) or has been weak (at or
// below the lower threshold) for more than the hang time since the last healthy chunk (src/agc.c:165-215); otherwise
// every chunk is multiplied by the same g.  The front kernel has therefore already multiplied by the g it found in the
// state and left max |y|^2 per chunk in peak2; this kernel confirms the assumption for every chunk of the call:
// locked, no ratchet, and no weak chunk further than the hang time from the last healthy one (a prefix maximum of the
// healthy chunks' times, only computed when a weak chunk exists).  Then the bytes the front kernel wrote are final and
// the state advances here: samples_seen, and the strong-peak time of the last healthy chunk.  Anything else sets
// *verify_flag: the unfused kernels queued behind redo the call from the untouched state.
// verify_flag[0] = the verdict (the fallback launches' run_if); [1] ratchet seen, [2] weak chunk seen, [3] last healthy chunk
// (reset by the verdict for the next call; zero / -1 at create; [4] = the tickets of the classification's workgroups).  k_agc_classify hands the peak array back zeroed (round 3:
// a fill in front of every fused launch did that; folding the classification into the one verdict workgroup was tried and
// is slower -- 16384 chunks of closed-form bookkeeping on one CU take 30 us).
__device__ void agc_verdict(const AgcArgs &a);
__global__ __launch_bounds__(256) void k_agc_classify(const AgcArgs a)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    const AgcState st = *a.state;
    int bad = 0, weak = 0, last_h = -1, cls_k = 0;
    int64_t b_c = 0;                                               // the call's outputs in front of this chunk
    if (c < a.geom.n_chunks) {
        int k = 0;
        if (st.locked) {
            const int64_t b = agc_out_end(a.geom, (int64_t)c - 1), e = agc_out_end(a.geom, c);
            a.chunk_b[c] = b; b_c = b;                             // (the verdict's chunk times: not worked out a second time by one workgroup)
            if (e > b) {                                           // empty chunks never reach agc_apply
                const float pk = (float)sqrt(__longlong_as_double((long long)a.peak2[c]));
                const float outp = pk * st.gain;
                const float lower = a.target * kAgcLower;
                // k_front_mid keeps max |y|^2 in float.  Error budget of `outp` against the exact kernels' (and the reference's cabsf):
                // m2 = fmaf(x, x, y y) is off by at most 1.5 ulp of a float (two roundings) = 1.8e-7 relative, the square root halves
                // that, the cast of the root and the product with the gain round once more each in BOTH paths (2 x 6e-8 apart at
                // worst): < 2.2e-7 relative in all.  A chunk within 8 FLT_EPSILON = 9.5e-7 (relative, at either threshold: more than
                // four times the budget) is not judged here -- the exact kernels redo the call
                const float tol = a.peak_approx ? 8.0f * 1.1920929e-7f : 0.0f;
                if (outp > 1.0f - tol) bad = 1;                    // ratchet (or too close to call)
                else if (a.peak_approx && fabsf(outp - lower) <= tol * lower) bad = 1;
                else if (outp > lower) { k = 1; last_h = c; }
                else { k = 2; weak = 1; }
            }
        }
        cls_k = k;
        a.chunk_len[c] = k;                                        // scratch: 0 empty, 1 healthy, 2 weak
        a.peak2[c] = 0ull;                                         // handed back zeroed: the next fused launch accumulates into it (no fill on the hot path)
        if (a.peak2_fallback) a.peak2_fallback[c] = 0ull;          // ... and the fallback's own array, which its k_agc_peak accumulates into
    }
    // The hang-time test (a weak chunk creeps iff its time is more than the hang time past the last healthy chunk in front of it, or
    // the state's last strong peak) where the chunks are: the last healthy chunk at or in front of every chunk inside this
    // workgroup's 256 by a prefix maximum of indices (chunk times grow with the index), a weak chunk that has one tests itself; of
    // those that have none -- they all share whatever came before the workgroup -- the LAST one decides, and only that index goes to
    // the verdict (wg_pend), with the workgroup's last healthy index (wg_last).  (Round 5, first form: ONE workgroup walked every weak
    // chunk -- 64 rounds of three dependent loads and four divisions on the cs16-am-nrsc5 preset, whose narrow output makes weak
    // chunks the rule: 82 us of a 0.44 ms step.)
    {
        __shared__ int s_wl[4], s_wp[4];
        __shared__ long long s_b[256];
        s_b[threadIdx.x] = b_c;
        int h = cls_k == 1 ? c : -1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(h, o); if ((int)(threadIdx.x & 63) >= o) h = v > h ? v : h; }
        if ((threadIdx.x & 63) == 63) s_wl[threadIdx.x >> 6] = h;
        __syncthreads();
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) h = s_wl[w] > h ? s_wl[w] : h;
        if (threadIdx.x == 255) a.wg_last[blockIdx.x] = h;
        auto t_of_b = [&](int64_t b) { return a.clock_wall ? a.t_wall : (double)(st.seen + (uint64_t)b) / a.rate; };
        int pend = -1;
        if (cls_k == 2) {
            if (h >= 0) {
                const double before = fmax(st.last_strong, t_of_b(s_b[h - (int)blockIdx.x * 256]));
                if (t_of_b(b_c) - before > (double)kAgcHangTime) bad = 1;
            } else pend = c;
        }
#pragma unroll
        for (int k2 = 32; k2 >= 1; k2 >>= 1) { const int o = __shfl_xor(pend, k2); pend = o > pend ? o : pend; }
        if ((threadIdx.x & 63) == 0) s_wp[threadIdx.x >> 6] = pend;
        __syncthreads();
        if (threadIdx.x == 0) {
            int m = s_wp[0];
            for (int w = 1; w < 4; ++w) m = s_wp[w] > m ? s_wp[w] : m;
            a.wg_pend[blockIdx.x] = m;
        }
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(a.verify_flag + 1, 1);
    if (__ballot(weak) != 0ull && (threadIdx.x & 63) == 0) atomicOr(a.verify_flag + 2, 1);
#pragma unroll
    for (int k2 = 32; k2 >= 1; k2 >>= 1) { const int o = __shfl_xor(last_h, k2); last_h = o > last_h ? o : last_h; }
    if ((threadIdx.x & 63) == 0 && last_h >= 0) atomicMax(a.verify_flag + 3, last_h);
    // the verdict, by whichever workgroup is the last to get here (one launch instead of two: n_chunks / 256 tickets on one word --
    // 64 for a 2^28-frame call -- cost less than the launch they save; a ticket per block of a LARGE grid does not)
    __shared__ int s_last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = __hip_atomic_fetch_add(a.verify_flag + 4, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    agc_verdict(a);
}

// (the body of the verdict: one workgroup of any size)
__device__ void agc_verdict(const AgcArgs &a)
{
    __shared__ int s_bad;
    const int tid = threadIdx.x, nthr = (int)blockDim.x;
    const AgcState st = *a.state;
    // (time of a chunk = outputs seen before it / rate; the counts come from the classification: 16384 chunks of two 64-bit
    //  divisions each took this one workgroup 0.1 ms on the cs16-am-nrsc5 preset, whose narrow output makes weak chunks the rule)
    auto t_of = [&](int c) { return a.clock_wall ? a.t_wall : (double)(st.seen + (uint64_t)a.chunk_b[c]) / a.rate; };
    const int any_bad = a.verify_flag[1], any_weak = a.verify_flag[2], last_healthy = a.verify_flag[3];
    if (tid == 0) s_bad = (!st.locked || any_bad) ? 1 : 0;
    __syncthreads();
    if (!s_bad && any_weak) {
        // what the workgroups of the classification could not settle: their last weak chunk with no healthy one in front of it inside
        // the workgroup, against the last healthy chunk of the workgroups before (or the state's last strong peak)
        __shared__ int s_wgp[1024];
        const int n_wg = (a.geom.n_chunks + 255) / 256;
        if (n_wg > 1024) { if (tid == 0) s_bad = 1; }              // (more than 2^18 chunks in one call: the exact kernels)
        else {
            for (int w = tid; w < n_wg; w += nthr) s_wgp[w] = a.wg_last[w];
            __syncthreads();
            if (tid == 0) { int m = -1; for (int w = 0; w < n_wg; ++w) { m = s_wgp[w] > m ? s_wgp[w] : m; s_wgp[w] = m; } }
            __syncthreads();
            for (int w = tid; w < n_wg; w += nthr) {
                const int c = a.wg_pend[w];
                if (c < 0) continue;
                const int last = w > 0 ? s_wgp[w - 1] : -1;
                const double before = last >= 0 ? fmax(st.last_strong, t_of(last)) : st.last_strong;
                if (t_of(c) - before > (double)kAgcHangTime) atomicOr(&s_bad, 1);
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        a.verify_flag[0] = s_bad;
        a.verify_flag[1] = 0; a.verify_flag[2] = 0; a.verify_flag[3] = -1; a.verify_flag[4] = 0;      // ready for the next call
        if (!s_bad) {
            AgcState nx = st;
            if (last_healthy >= 0) nx.last_strong = t_of(last_healthy);
            nx.seen = st.seen + (uint64_t)a.n_out;
            *a.state = nx;
        }
        // ... and to the host, where it waits for the verdict (the state above first: a host that has seen the word may read the state)
        if (a.verdict_host) {
            __threadfence_system();
            __hip_atomic_store(a.verdict_host, s_bad, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

hipError_t launch_agc_verify(const AgcArgs &a, hipStream_t s)
{
    const unsigned nb = a.geom.n_chunks > 0 ? (unsigned)((a.geom.n_chunks + 255) / 256) : 1u;
    hipLaunchKernelGGL(k_agc_classify, dim3(nb), dim3(256), 0, s, a);        // classification, then the verdict in its last workgroup
    return hipGetLastError();
}

hipError_t launch_agc(const AgcArgs &a, hipStream_t s)
{
    if (a.geom.n_chunks <= 0) return hipSuccess;
    const int64_t n_items = (int64_t)a.geom.n_chunks * a.splits;
    unsigned grid = (unsigned)n_items;
    if (a.run_if) {
        // conditional fallback behind a fused launch: k_agc_classify has zeroed peak2 already; small grids (see k_agc_peak)
        if (grid > 2048u) grid = 2048u;
    } else {
        const hipError_t e = hipMemsetAsync(a.peak2, 0, (size_t)a.geom.n_chunks * sizeof(unsigned long long), s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_agc_peak, dim3(grid), dim3(kThreads), 0, s, a);                     // also fills chunk_len[]
    hipLaunchKernelGGL(k_agc_scan, dim3(1), dim3(64), 0, s, a);
    if (a.n_out > 0) hipLaunchKernelGGL(k_agc_apply, dim3(grid), dim3(kThreads), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// RMS profiles "dx" / "local": liquid's agc_crcf (src/agc.c:39-62, 92-100, 227-229) -- per sample
//     y = x g;  y2 = |y|^2;  p = (1 - alpha) p + alpha y2  (double, stored as float);
//     if (p > 1e-6) g *= exp(-0.5 alpha ln p);  g = min(g, 1e6);   out = y
// a nonlinear recurrence with no closed form across samples.  What makes it parallel is that it FORGETS:
// linearised around its fixed point the state error decays like (1 - alpha / 2)^n, so a lane that starts
// `warm` = 26 / alpha samples early from a state within a few per cent arrives within 1e-7 of the true one.
//   The chunk grid belongs to the STREAM, not to the call (round 4): chunk k covers stream positions [k C, (k + 1) C), and
//   what a lane does depends on positions and samples only, never on where the calls were cut -- the output is the same
//   under ANY split of the stream into calls, byte for byte (test_agc_rms_is_split_invariant).
//   k_agc_rms_spec: one LANE per chunk that overlaps the call.  A chunk that began in an earlier call continues from the
//       carried state.  Every other chunk starts `warm` samples ahead of its first output -- in the history the chain
//       keeps of the AGC's input (the last `warm` samples of earlier calls) where that lies before the call -- from a
//       guess made of those samples alone (gain = 1 / rms of the first 256, unit energy); within `warm` samples of the
//       last reset it starts AT the reset, from the reset state (exact).  Each lane records the state it arrived with
//       and the state it left.
//   k_agc_rms_fix:  checks, in parallel, that every chunk arrived where its predecessor left (2e-6 relative
//       in gain and energy; the first chunk's predecessor is the carried state); if one did not -- a silent stretch
//       freezes the gain (p <= 1e-6) and with it the guess -- one lane re-runs the stream from there until the states
//       meet again.  Then stores the stream state.  (The sequential BITS are out of reach of any parallel scheme:
//       tools/agc_merge_probe.c)
// Unpinned like every liquid operator (DESIGN SPEC): parity with the oracle's restatement (glibc expf / logf) to 2e-5.
// ---------------------------------------------------------------------------------------------
struct RmsSt { float g, p; };

// ln x for a float-range x > 0, in double to ~1e-13 relative (of ln x): x = 2^e m with m in [sqrt(1/2), sqrt(2)),
// s = (m - 1) / (m + 1), ln m = 2 atanh s = 2 s (1 + s^2/3 + s^4/5 + ... + s^16/17), |s| <= 0.1716.
// Written out (v_frexp, v_rcp_f64 + two Newton steps, Horner) because a lane runs ONE dependent chain: the
// library log / exp are ~85 dependent instructions per sample together, these ~40.
__device__ __forceinline__ double rms_ln(double x)
{
    double m = __builtin_amdgcn_frexp_mant(x);              // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m; e = lo ? e - 1 : e;
    const double f = m - 1.0, d = m + 1.0;
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
    double sq = f * r;
    sq = __builtin_fma(r, __builtin_fma(-sq, d, f), sq);    // s = f / d to the last bit or so
    const double z = sq * sq;
    double q = 1.0 / 17.0;
    q = __builtin_fma(q, z, 1.0 / 15.0); q = __builtin_fma(q, z, 1.0 / 13.0); q = __builtin_fma(q, z, 1.0 / 11.0);
    q = __builtin_fma(q, z, 1.0 / 9.0);  q = __builtin_fma(q, z, 1.0 / 7.0);  q = __builtin_fma(q, z, 1.0 / 5.0);
    q = __builtin_fma(q, z, 1.0 / 3.0);
    const double lnm = __builtin_fma(sq + sq, q * z, sq + sq);
    return __builtin_fma((double)e, 0.69314718055994530942, lnm);
}

// e^t for |t| <= 0.14 (0.5 alpha |ln p| with p in (1e-6, 1e13), alpha <= 1e-2): Taylor to t^9 / 9!, < 1e-15
__device__ __forceinline__ double rms_exp_small(double t)
{
    double q = 1.0 / 362880.0;
    q = __builtin_fma(q, t, 1.0 / 40320.0); q = __builtin_fma(q, t, 1.0 / 5040.0); q = __builtin_fma(q, t, 1.0 / 720.0);
    q = __builtin_fma(q, t, 1.0 / 120.0);   q = __builtin_fma(q, t, 1.0 / 24.0);   q = __builtin_fma(q, t, 1.0 / 6.0);
    q = __builtin_fma(q, t, 0.5);           q = __builtin_fma(q, t, 1.0);
    return __builtin_fma(q, t, 1.0);
}

__device__ __forceinline__ cf2 rms_step(cf2 v, RmsSt &st, float alpha, float half_alpha_neg)
{
    const float yr = v.x * st.g, yi = v.y * st.g;
    const float y2 = yr * yr + yi * yi;
    st.p = (float)((1.0 - (double)alpha) * (double)st.p + (double)alpha * (double)y2);
    // logf / expf rounded the way glibc rounds them (to nearest, but for rare near-ties): through double.  It matters:
    // the update factor is 1 + O(alpha), so its float rounding is percent-level noise on the correction term, and a
    // libm that rounds differently by 1e-7 per step moves the settled gain by 1e-7 / alpha (1e-3 for dx).
    if (st.p > 1e-6f) st.g *= (float)rms_exp_small((double)(half_alpha_neg * (float)rms_ln((double)st.p)));
    st.g = fminf(st.g, 1e6f);
    return cf2{yr, yi};
}

// samples [i0, i1) of the stream through the loop; EMIT: pack and store them.  The loads run four samples ahead of
// the chain (a lane's stream is contiguous, but every load of a wave touches 64 different cache lines: waited for
// where it is used, its latency would be most of the step).
template <bool EMIT>
__device__ __forceinline__ void rms_run(const AgcRmsArgs &a, int64_t i0, int64_t i1, RmsSt &st, float han)
{
    int64_t i = i0;
    if (i + 4 <= i1) {
        cf2 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = a.x[i + k];
        for (; i + 8 <= i1; i += 4) {
            cf2 nx[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) nx[k] = a.x[i + 4 + k];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const cf2 y = rms_step(v[k], st, a.alpha, han);
                if (EMIT) pack_store(a.out, i + k, a.out_fmt, y);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = nx[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const cf2 y = rms_step(v[k], st, a.alpha, han);
            if (EMIT) pack_store(a.out, i + k, a.out_fmt, y);
        }
        i += 4;
    }
    for (; i < i1; ++i) {
        const cf2 y = rms_step(a.x[i], st, a.alpha, han);
        if (EMIT) pack_store(a.out, i, a.out_fmt, y);
    }
}

__global__ __launch_bounds__(64) void k_agc_rms_spec(const AgcRmsArgs a)
{
    const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (j >= a.n_chunks) return;
    const int64_t abs_s = (a.pos0 / a.chunk + j) * a.chunk;                              // the chunk's first stream position
    const int64_t s = (abs_s > a.pos0 ? abs_s : a.pos0) - a.pos0;                        // ... and its part in this call: x[s .. e)
    const int64_t e = (abs_s + a.chunk < a.pos0 + a.n ? abs_s + a.chunk : a.pos0 + a.n) - a.pos0;
    const float han = -0.5f * a.alpha;
    RmsSt st{a.state->gain, a.state->peak_memory};
    if (abs_s >= a.pos0) {                                   // the chunk begins in this call: a trajectory of its own
        const int64_t a0 = abs_s - a.warm;
        int64_t i;
        if (a0 <= 0) { i = -a.pos0; st.g = 1.0f; st.p = 1.0f; }      // from the reset, in the reset state (agc_reset: src/agc.c:227-229)
        else {
            i = a0 - a.pos0;                                 // >= -warm = -hist_valid here
            // (round 5: the power of the first 256 samples, four independent sums -- a guess within a few per cent instead of the
            //  ~18 % of 32 samples, which is what lets the warm-up end after 26 / alpha samples instead of 40 / alpha)
            float m4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int k = 0; k < 256; k += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { const cf2 v = a.x[i + k + u]; m4[u] = fmaf(v.x, v.x, fmaf(v.y, v.y, m4[u])); }
            }
            float m = ((m4[0] + m4[1]) + (m4[2] + m4[3])) * (1.0f / 256.0f);
            st.g = m > 1e-20f ? fminf(1.0f / sqrtf(m), 1e6f) : 1.0f;
            st.p = 1.0f;
        }
        rms_run<false>(a, i, s, st, han);
    }
    a.st[4 * j + 0] = st.g; a.st[4 * j + 1] = st.p;
    rms_run<true>(a, s, e, st, han);
    a.st[4 * j + 2] = st.g; a.st[4 * j + 3] = st.p;
}

__device__ __forceinline__ bool rms_close(float a, float b) { return fabsf(a - b) <= 2e-6f * fmaxf(fabsf(a), fabsf(b)); }

__global__ __launch_bounds__(1024) void k_agc_rms_fix(const AgcRmsArgs a)
{
    __shared__ int first_bad;
    if (threadIdx.x == 0) first_bad = 0x7fffffff;
    __syncthreads();
    const float g0 = a.state->gain, p0 = a.state->peak_memory;          // the predecessor of the call's first chunk
    for (int c = (int)threadIdx.x; c < a.n_chunks; c += 1024) {
        const float pg = c ? a.st[4 * c - 2] : g0, pp = c ? a.st[4 * c - 1] : p0;
        if (!rms_close(a.st[4 * c], pg) || !rms_close(a.st[4 * c + 1], pp)) atomicMin(&first_bad, c);
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const float han = -0.5f * a.alpha;
    const int64_t k0 = a.pos0 / a.chunk;
    for (int64_t c = first_bad; c < a.n_chunks; ++c) {
        RmsSt st{c ? a.st[4 * c - 2] : g0, c ? a.st[4 * c - 1] : p0};   // where the stream really is
        if (rms_close(a.st[4 * c], st.g) && rms_close(a.st[4 * c + 1], st.p)) continue;
        const int64_t abs_s = (k0 + c) * a.chunk;
        const int64_t s = (abs_s > a.pos0 ? abs_s : a.pos0) - a.pos0;
        const int64_t e = (abs_s + a.chunk < a.pos0 + a.n ? abs_s + a.chunk : a.pos0 + a.n) - a.pos0;
        rms_run<true>(a, s, e, st, han);
        a.st[4 * c + 2] = st.g; a.st[4 * c + 3] = st.p;
    }
    if (a.n_chunks > 0) {
        a.state->gain = a.st[4 * (int64_t)a.n_chunks - 2];
        a.state->peak_memory = a.st[4 * (int64_t)a.n_chunks - 1];
        a.state->seen += (uint64_t)a.n;
    }
}

void agc_rms_geometry(float alpha, int64_t pos0, int64_t n, int64_t *chunk, int64_t *warm, int32_t *n_chunks)
{
    // e^-13 = 2.3e-6 of the guess's error is left after 26 / alpha samples: with the guess within a few per cent (256 samples of
    // power) an order of magnitude inside the 2e-6 at which k_agc_rms_fix accepts a chunk's arrival (until round 4: 40 / alpha behind
    // a 32-sample guess).  Chunks of warm / 16, at least 256: a call costs warm + chunk dependent samples of ~0.18 us whatever its
    // length -- `local` (alpha 1e-2): 2600 + 256 = 0.5 ms (1.1 until round 4), `dx` (1e-4): 260 000 + 16 250 = 50 ms (77)
    int64_t w = (int64_t)(26.0 / (double)alpha + 0.5);
    w = (w + 1) & ~(int64_t)1;
    if (w < 512) w = 512;                                       // (the guess reads 256 samples of the window)
    int64_t ch = w / 16; if (ch < 256) ch = 256;
    ch &= ~(int64_t)1;
    *warm = w; *chunk = ch;
    *n_chunks = n > 0 ? (int32_t)((pos0 + n + ch - 1) / ch - pos0 / ch) : 0;     // chunks of the stream's grid that overlap [pos0, pos0 + n)
}

hipError_t launch_agc_rms(const AgcRmsArgs &a, hipStream_t s)
{
    if (a.n <= 0 || a.n_chunks <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_agc_rms_spec, dim3((unsigned)((a.n_chunks + 63) / 64)), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_agc_rms_fix, dim3(1), dim3(1024), 0, s, a);
    return hipGetLastError();
}

} // namespace iqgpu
