// front_s2.hip -- k_front_s2: a TWO-stage decimating chain (S = 2: one leading half-band of semi-length 3 or 5, the m = 10
// half-band, the 256-arm polyphase) in ONE wave-autonomous kernel:
//
//   raw -> unpack/gain -> [dc block] -> [iq correct] -> [pre NCO] -> half-band 0 -> half-band 1 (m = 10) -> polyphase -> [post NCO] -> out
//
// Until round 4 such chains (BASELINE configs[2]: cs16 10 MS/s -> 2.4 MS/s) ran as k_cascade (stage 0 -> cf32 in HBM) followed by
// k_front_s1 (last stage + polyphase on that stream): 16 bytes per 4 input frames written and read back, which made both kernels
// HBM-bound by their own intermediate (0.54 GB each way per 2^27 frames, counter traffic 4.4 x algorithmic).  The
// reference does all of it in one thread on one chunk (src/pipeline.c:492-537, src/resampler.c:49-53).
//
// Structure.  The tile is the one of k_front_s1's last-stage instantiation: 512 INTERMEDIATE samples = 1024 input frames, and the
// tile routine is that kernel's run_tiles (front_tiles.hpp) itself -- with a feeder in place of its vector loads: for each half
// of the tile the feeder unpacks 512 frames, runs the pointwise operators, hands them to the stage-0 rows of the wave's LDS slice
// and computes the lane's four stage-0 outputs, which ARE samples 256 c + 4 lane .. + 3 of the intermediate tile -- the very
// registers run_tiles would have loaded from the intermediate stream.  Stage 0 is casc_stage<M> of k_cascade on k_cascade's row
// layout, the operators are its statements: same products in the same order, the bytes equal the two-kernel path's
// (test_cascade_instantiations_equal_the_generic_kernel).  The intermediate stream never leaves the wave.
//
// Edge tiles (stream history, end of the call) are few and keep the two-kernel arithmetic literally: an edge wave runs
// casc_tiles<EDGE> over the input tiles of its run into a PRIVATE stretch of intermediate samples (a few KB per edge wave: the
// runs of neighbouring edge waves overlap in their warm-up tiles, whose first samples differ from wave to wave), fences, and
// runs run_tiles<EDGE> over them; the histories both leave for the next call are the ones the two kernels left.
#include "cascade_tiles.hpp"
#include "front_tiles.hpp"

namespace iqgpu {

constexpr int kS2Waves = kWaves;                  // 12: 3 per SIMD (the slices of both stages: 11 KB per wave)
constexpr int kS2Threads = kS2Waves * 64;
constexpr int kS2NcoLds = 1024 * 8;
constexpr int kS2ArbLds = 256 * 14 * 4;
__host__ __device__ constexpr int s2_stage0_bytes(int m0) { return 4 * plane_stride(casc_hist_rows(m0) + 64 + 1); }
__host__ __device__ constexpr int s2_wave_lds(int m0) { return kWaveLds + s2_stage0_bytes(m0); }
static_assert(kS2NcoLds + kS2ArbLds + kS2Waves * s2_wave_lds(5) <= 160 * 1024, "LDS");

struct S2Args { FrontArgs a1, a2; };              // a1: the chain as k_cascade sees it (K = 1), a2: the last stage as k_front_s1 sees it
// intermediate samples an edge wave keeps for itself: its run of edge_tpw last-stage tiles, the warm-up tile in front, the input
// tile behind the call's last group (a1.casc_out = the base of these stretches, one per edge wave)
__host__ __device__ constexpr int64_t s2_edge_slots(int64_t edge_tpw) { return 256 * (2 * edge_tpw + 4); }

// The feeder of a streaming run: input sub-tile u = 2 t + c of 512 frames -> the lane's four stage-0 outputs x[c][0 .. 3].
template <int BPS, int M0>
struct S2Feed {
    static constexpr int VB = BPS;
    static constexpr int H0 = casc_hist_rows(M0);
    static constexpr int PS0 = plane_stride(H0 + 64 + 1);
    const FrontArgs &a;                           // a1
    const cf2 *nco;
    char *XE0, *XO0;
    int lane, seg;
    RawChunk nxt[2];
    v2f cs_n[2][4];
    float dc_vr = 0.0f, dc_vi = 0.0f, se = 0.0f, so = 0.0f;
    DcLane lane_pow{1.0f, 1.0f, 1.0f};
    bool unit_gain, nco_on, dc_started = false;

    __device__ __forceinline__ S2Feed(const FrontArgs &a_, const cf2 *nco_, char *slice0, int lane_, int seg_)
        : a(a_), nco(nco_), XE0(slice0), XO0(slice0 + 2 * PS0), lane(lane_), seg(seg_)
    {
        unit_gain = a.gain == 1.0f;
        nco_on = a.nco_mode != 0;
        if (a.dc_enable) lane_pow = dc_lane_init(a, lane);
    }
    __device__ __forceinline__ void lookup(int64_t first_frame)
    {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            uint32_t th = a.nco_theta0 + ((uint32_t)first_frame + (uint32_t)(256 * c + 4 * lane)) * a.nco_dtheta;
#pragma unroll
            for (int s = 0; s < 4; ++s) { cs_n[c][s] = nco_phasor2(nco, th); th += a.nco_dtheta; }
        }
    }
    __device__ __forceinline__ void fetch(int64_t u)        // the frames of input sub-tile u (a1.rem0 = 0 on this path)
    {
        const char *src = (const char *)a.raw + u * (kWTile * VB) + 4 * VB * lane;
        load_chunk<VB, true>(src, nxt[0]);
        load_chunk<VB, true>(src + 256 * VB, nxt[1]);
    }
    __device__ __forceinline__ void start(int64_t t)        // in front of the run's first tile
    {
        fetch(2 * t);
        if (nco_on) lookup(2 * t * kWTile);
    }
    __device__ __forceinline__ void produce(int64_t t, cf2 (&out)[2][4])
    {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int64_t u = 2 * t + sub;
            // ---- pointwise: k_cascade's statements (casc_tiles, streaming branch)
            cf2 x[2][4];
            unpack_chunk<VB>(nxt[0], a.in_fmt, a.gain, unit_gain, x[0]);
            unpack_chunk<VB>(nxt[1], a.in_fmt, a.gain, unit_gain, x[1]);
            fetch(u + 1);                                    // (the plan keeps a tile behind every run readable)
            if (a.dc_enable) {
                if (!dc_started) { const cd2 cv = a.dc_carry[seg]; dc_vr = (float)cv.x; dc_vi = (float)cv.y; dc_started = true; }
                dc_chunk(a, lane, lane_pow, x[0], 0u, dc_vr, dc_vi);
                dc_chunk(a, lane, lane_pow, x[1], 0u, dc_vr, dc_vi);
            }
            if (a.iq_enable) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float re = x[c][s].x;
                        x[c][s].x = re * a.iq_magp1;
                        x[c][s].y = fmaf(a.iq_phase, re, x[c][s].y);
                    }
            }
            if (nco_on) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const v2f y = pk_cmul(v2f{x[c][s].x, x[c][s].y}, cs_n[c][s]);
                        x[c][s] = cf2{y.x, y.y};
                    }
                lookup((u + 1) * kWTile);
            }
            // ---- stage 0: history slides to the front of the rows, the sub-tile's samples behind it (k_cascade's layout)
            {
                const int so_ = (lane >= 4 * H0 ? PS0 - 16 * H0 : 0) + lane * 4;
                if (lane < 8 * H0) { *(float *)(XE0 + so_) = se; *(float *)(XO0 + so_) = so; }
                const int woff = (H0 + (lane >> 1)) * 16 + (lane & 1) * PS0;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int off = woff + 32 * c * 16;
                    st4a(XE0 + off, make_float4(x[c][0].x, x[c][0].y, x[c][2].x, x[c][2].y));
                    st4a(XO0 + off, make_float4(x[c][1].x, x[c][1].y, x[c][3].x, x[c][3].y));
                }
            }
            __builtin_amdgcn_wave_barrier();
            v2f y[4];
            {
                CascWin0<M0> wn;
                casc_stage_load<M0>(XE0, XO0, lane, wn);
                const int ls = lane < 8 * H0 ? lane : 8 * H0 - 1;
                const int sr = (ls >= 4 * H0 ? PS0 - 16 * H0 : 0) + ls * 4;
                se = *(const float *)(XE0 + 64 * 16 + sr); so = *(const float *)(XO0 + 64 * 16 + sr);
                casc_stage_fma<M0>(wn, a.casc_taps[0], y);
            }
            __builtin_amdgcn_wave_barrier();                 // (the next sub-tile writes the rows these reads came from)
#pragma unroll
            for (int i = 0; i < 4; ++i) out[sub][i] = cf2{y[i].x, y[i].y};
        }
    }
};

// SPEC (late round 5): the chain's switches as compile-time constants for the streaming waves of the common cs16 / cu8 chains -- the same
// statements with their branches and selects folded away (configs[2]: 168 -> 120 VGPRs, the kernel 0.248 -> 0.219 ms).  0 = whatever
// the arguments say; 1 .. 7 = unit gain, no mixer behind the resampler, and (dc blocker + iq correction, mixer in front, output):
//   1 = (both, none, cf32) -- BASELINE configs[2] in front of its filter --  2 = (none, none, cs16)   3 = (none, mixer, cs16)
//   4 = (none, none, cf32)   5 = (none, mixer, cf32)   6 = (none, none, cu8)   7 = (none, mixer, cu8)
struct S2Spec { int dcq, nco, out; };
__host__ __device__ constexpr S2Spec s2_spec(int k)
{
    return k == 1 ? S2Spec{1, 0, IQGPU_FMT_CF32} : k == 2 ? S2Spec{0, 0, IQGPU_FMT_CS16} : k == 3 ? S2Spec{0, 1, IQGPU_FMT_CS16}
         : k == 4 ? S2Spec{0, 0, IQGPU_FMT_CF32} : k == 5 ? S2Spec{0, 1, IQGPU_FMT_CF32} : k == 6 ? S2Spec{0, 0, IQGPU_FMT_CU8} : S2Spec{0, 1, IQGPU_FMT_CU8};
}
template <int BPS, int M0, int SPEC = 0>
__global__ __launch_bounds__(kS2Threads) void k_front_s2(const S2Args p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const FrontArgs &a1 = p.a1;
    FrontArgs a2 = p.a2;
    a2.gain = 1.0f; a2.iq_enable = 0; a2.dc_enable = 0; a2.nco_mode = 0; a2.in_fmt = IQGPU_FMT_CF32;       // (as k_front_s1<8, .., VAR = 4> sees it)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    cf2 *s_nco = (cf2 *)smem;
    float *s_arb = (float *)(smem + kS2NcoLds);
    char *slice = (char *)smem + kS2NcoLds + kS2ArbLds + wave * s2_wave_lds(M0);
    char *slice0 = slice + kWaveLds;                 // the stage-0 rows
    WaveLds w;
    w.XE = slice; w.XO = w.XE + kXRows * kRowB; w.HB = w.XO + kHBOff * kRowB;
    w.nco = s_nco; w.arb = s_arb;
    if (((unsigned)(size_t)(__attribute__((address_space(3))) const void *)s_nco & 8191u) != 0u) __builtin_trap();   // nco_phasor2 ORs the index into the base
    w.arb_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)s_arb;

    if (a1.nco_mode != 0 || a2.pnco_mode != 0) {     // (a chain shifts in front of the resampler or behind it, never both)
        const float sgn = (a1.nco_mode < 0 || a2.pnco_mode < 0) ? -1.0f : 1.0f;
        for (int i = tid; i < 1024; i += kS2Threads) { const cf2 v = a1.nco_tab[i]; s_nco[i] = cf2{v.x, sgn * v.y}; }
    }
    for (int i = tid; i < 256 * 14; i += kS2Threads) {
        const int arm = i / 14, k = i % 14;
        s_arb[(arm ^ (arm >> 5)) * 14 + k] = a2.arb_table[arm * 16 + k];
    }
    for (int i = lane; i < s2_wave_lds(M0) / 16; i += 64) ((float4 *)slice)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    const int64_t gw = (int64_t)blockIdx.x * kS2Waves + wave;
    if (gw < a2.w_n_edge) {
        // an edge run of the LAST stage's tiles [e0, e1): its input tiles through k_cascade's edge routine into the intermediate
        // buffer, then k_front_s1's edge routine over that -- the two-kernel path, confined to one wave
        int64_t e0, e1;
        if (gw < a2.w_n_edge1) { e0 = gw * a2.w_edge_tpw; e1 = e0 + a2.w_edge_tpw; if (e1 > a2.w_edge_ta) e1 = a2.w_edge_ta; }
        else { e0 = a2.w_edge_tb + (gw - a2.w_n_edge1) * a2.w_edge_tpw; e1 = e0 + a2.w_edge_tpw; if (e1 > a2.w_total_tiles) e1 = a2.w_total_tiles; }
        const int seg = (gw < a2.w_n_edge1) ? (int)gw : (int)(gw + a2.w_n_stream);   // DcGeom mode 1 order
        CascLds cw;
#pragma unroll
        for (int k = 0; k < kCascMaxK; ++k) { cw.XE[k] = slice0; cw.XO[k] = slice0; }
        cw.XO[0] = slice0 + 2 * plane_stride(casc_hist_rows(M0) + 64 + 1);
        cw.nco = s_nco;
        // input tiles of 512 frames: the run starts where its dc segment starts, (e0 - warm) * 1024 frames, or with one tile of
        // pure history at the start of the call; it ends with the call or with the last-stage tile e1 - 1
        const int64_t first = 2 * (e0 - a2.w_warm_tiles);
        // (the run that holds the call's last tile also takes the input tile behind it, if any: a call that ends inside a
        //  decimation group leaves frames there that complete no sample but belong to the history it hands on)
        int64_t u0 = first < 0 ? 0 : first, u1 = 2 * e1;
        if (u1 > a1.w_total_tiles || e1 == a2.w_total_tiles) u1 = a1.w_total_tiles;
        // intermediate sample 256 u0 (the first one this run computes) sits at the start of the wave's private stretch
        cf2 *const priv = p.a1.casc_out + gw * s2_edge_slots(a2.w_edge_tpw) - 256 * u0;
        FrontArgs a1e = p.a1;
        a1e.casc_out = priv;
        a2.raw = priv;
        casc_tiles<BPS, true, false, 0>(a1e, cw, lane, first < 0 ? u0 - 1 : u0, u0, u1, seg);
        // the wave's own stores, then its own loads of the same lines: out to memory and this CU's L1 refreshed
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        run_tiles<8, true, false, false, false, false>(a2, w, lane, e0 - a2.w_warm_tiles, e0, e1, seg);
    } else {
        const int64_t r = gw - a2.w_n_edge;
        if (r >= a2.w_n_stream) return;
        const int64_t t0 = w_run_start(a2, r), t1 = w_run_start(a2, r + 1);
        const int seg = (int)(a2.w_n_edge1 + r);
        FrontArgs a1c = a1;
        if constexpr (SPEC != 0) {
            // (what launch_front_s2 has checked the arguments to be, spelled out for the compiler; the mixer's direction stays a
            //  run-time value: it only picks the table's sign at the top of the kernel)
            constexpr S2Spec sp = s2_spec(SPEC);
            a1c.gain = 1.0f; a1c.in_fmt = BPS == 2 ? (int)IQGPU_FMT_CU8 : (int)IQGPU_FMT_CS16; a1c.dc_enable = sp.dcq; a1c.iq_enable = sp.dcq;
            if (!sp.nco) a1c.nco_mode = 0; else if (a1c.nco_mode == 0) a1c.nco_mode = 1;
            a2.pnco_mode = 0; a2.out_fmt = sp.out;
        }
        S2Feed<BPS, M0> feed(a1c, s_nco, slice0, lane, seg);
        feed.start(t0 - a2.w_warm_tiles);
        run_tiles<8, false, false, false, false, false, S2Feed<BPS, M0>>(a2, w, lane, t0 - a2.w_warm_tiles, t0, t1, seg, &feed);
    }
}

int front_s2_waves() { return kS2Waves; }
int64_t front_s2_mid_samples(const FrontArgs &a2) { return (a2.w_n_edge + 1) * s2_edge_slots(a2.w_edge_tpw); }

// which chains: two stages with liquid's 60 dB lengths, a vector-loadable input format, no fused AGC (that path keeps the two
// kernels), the call aligned on a decimation group (the streaming waves read whole 16-byte words)
bool front_s2_shape(const FrontArgs &a1)
{
    if (a1.casc_K != 1 || (a1.m[0] != 3 && a1.m[0] != 5)) return false;
    switch (a1.in_fmt) {
    case IQGPU_FMT_CS8: case IQGPU_FMT_CU8: case IQGPU_FMT_CS16: case IQGPU_FMT_CU16: case IQGPU_FMT_SC16Q11: case IQGPU_FMT_CF32: return true;
    default: return false;
    }
}

hipError_t launch_front_s2(const FrontArgs &a1, const FrontArgs &a2, hipStream_t s)
{
    if (!front_s2_shape(a1) || a1.rem0 != 0 || a2.rem0 != 0 || a2.agc_fused) return hipErrorInvalidValue;
    // S2Feed::produce fetches the 512 input frames behind a streaming run's last sub-tile ahead, and the tail of the call must stay
    // with the edge waves that leave stage 0's history: both hold for the plans plan_geometry makes (a2's streaming region ends
    // hist2_cap intermediate samples and a whole tile in front of the call's end) -- checked here, not assumed
    if (a2.w_n_stream > 0 && ((a2.w_edge_tb + 1) * 1024 > a1.frames_in || 1024 * (a2.w_total_tiles - a2.w_edge_tb) < (int64_t)a1.hist_cap))
        return hipErrorInvalidValue;
    const int64_t n_items = a2.w_n_edge + a2.w_n_stream;
    const unsigned grid = (unsigned)((n_items + kS2Waves - 1) / kS2Waves);
    if (grid == 0) return hipSuccess;
    S2Args p;
    p.a1 = a1; p.a2 = a2;
    const size_t lds = (size_t)kS2NcoLds + kS2ArbLds + (size_t)kS2Waves * s2_wave_lds(a1.m[0]);
    int cls;
    switch (a1.in_fmt) {
    case IQGPU_FMT_CS8: case IQGPU_FMT_CU8: cls = 2; break;
    case IQGPU_FMT_CF32: cls = 8; break;
    default: cls = 4; break;
    }
#define IQGPU_LAUNCH_S2X(BPS, M0, SPEC)                                                                              \
    do {                                                                                                              \
        static LdsAttrCache cache;                /* per instantiation */                                          \
        { const hipError_t e = cache.ensure((const void *)k_front_s2<BPS, M0, SPEC>, lds); if (e != hipSuccess) return e; } \
        hipLaunchKernelGGL((k_front_s2<BPS, M0, SPEC>), dim3(grid), dim3(kS2Threads), lds, s, p);                   \
    } while (0)
#define IQGPU_LAUNCH_S2(BPS, M0) IQGPU_LAUNCH_S2X(BPS, M0, 0)
    // the switch sets with an instantiation of their own (cs16, unit gain, no mixer behind the resampler): s2_spec
    int spec = 0;
    if ((a1.in_fmt == IQGPU_FMT_CS16 || a1.in_fmt == IQGPU_FMT_CU8) && a1.gain == 1.0f && a2.pnco_mode == 0 && !(a1.dbg & kDbgNoFast)) {
        for (int k = 1; k <= 7 && spec == 0; ++k) {
            const S2Spec sp = s2_spec(k);
            if ((a1.dc_enable != 0) == (sp.dcq != 0) && (a1.iq_enable != 0) == (sp.dcq != 0) && (a1.nco_mode != 0) == (sp.nco != 0) && a2.out_fmt == sp.out) spec = k;
        }
    }
    if (spec != 0) {
#define IQGPU_LAUNCH_S2S(BPS, M0)                                                                                    \
        do {                                                                                                          \
            if (spec == 1) IQGPU_LAUNCH_S2X(BPS, M0, 1); else if (spec == 2) IQGPU_LAUNCH_S2X(BPS, M0, 2);           \
            else if (spec == 3) IQGPU_LAUNCH_S2X(BPS, M0, 3); else if (spec == 4) IQGPU_LAUNCH_S2X(BPS, M0, 4);      \
            else if (spec == 5) IQGPU_LAUNCH_S2X(BPS, M0, 5); else if (spec == 6) IQGPU_LAUNCH_S2X(BPS, M0, 6);      \
            else IQGPU_LAUNCH_S2X(BPS, M0, 7);                                                                        \
        } while (0)
        if (cls == 2) { if (a1.m[0] == 5) IQGPU_LAUNCH_S2S(2, 5); else IQGPU_LAUNCH_S2S(2, 3); }
        else          { if (a1.m[0] == 5) IQGPU_LAUNCH_S2S(4, 5); else IQGPU_LAUNCH_S2S(4, 3); }
#undef IQGPU_LAUNCH_S2S
        return hipGetLastError();
    }
    if (a1.m[0] == 5) { if (cls == 2) IQGPU_LAUNCH_S2(2, 5); else if (cls == 4) IQGPU_LAUNCH_S2(4, 5); else IQGPU_LAUNCH_S2(8, 5); }
    else              { if (cls == 2) IQGPU_LAUNCH_S2(2, 3); else if (cls == 4) IQGPU_LAUNCH_S2(4, 3); else IQGPU_LAUNCH_S2(8, 3); }
#undef IQGPU_LAUNCH_S2
#undef IQGPU_LAUNCH_S2X
    return hipGetLastError();
}

} // namespace iqgpu
