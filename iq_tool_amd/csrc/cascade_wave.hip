// cascade_wave.hip -- k_cascade: the leading half-band stages of a multi-stage decimation
// (msresamp2, SPEC B.6) as a wave-autonomous kernel, for chains with S >= 2 stages:
//
//   raw -> unpack/gain -> [dc block] -> [iq correct] -> [pre NCO] -> half-band stages 0 .. K-1  -> cf32 (HBM)
//
// K = S - 1.  The stream it writes (rate / 2^K) is then consumed by k_front_s1 (front_wave.hip) as a
// one-stage chain: last half-band (m = 10) -> 256-arm polyphase -> [post NCO] -> pack.  The extra
// round trip through HBM is 16 bytes per 2^K input frames; what it buys is that every stage runs in
// the structure of the headline kernel instead of the workgroup-tiled k_front.
//
// Mapping (as in front_wave.hip): one wavefront owns a run of 512-frame tiles, frames arrive by
// register-prefetched 16-byte loads, every stage keeps its input split into even / odd streams in
// the wave's private LDS slice.  Stage 0 (512 samples per tile) uses rows of 4 cf32 (48-byte pitch)
// and a lane owns 4 consecutive outputs.  Stage k >= 1 sees 512 >> k samples per tile; a wave
// instruction costs the same with 8 active lanes as with 64, so the late stages are spread over the
// lanes instead of being run "narrow": 2 outputs per lane in stage 1, 1 output per lane from stage 2
// on, their inputs in plain linear even / odd arrays (8 bytes a sample: consecutive lanes read
// consecutive words, no bank conflicts, no padding).  Round 1 ran every stage 4 outputs per lane on
// 64 >> k lanes: 112 packed FMAs and 344 LDS cycles per tile at K = 4 against 52 and about 200 now.
// Semi-lengths 3 and 5 (what liquid designs for every stage but the last at 60 dB) are compiled in;
// anything else stays on k_front.
#include "cascade_tiles.hpp"

namespace iqgpu {

size_t cascade_wave_lds(const FrontArgs &a, bool two_tile_trips)
{
    size_t b = 0;
    for (int k = 0; k < a.casc_K; ++k) b += (size_t)casc_stage_bytes(k, a.m[k]);
    // (k_cascade2's streaming waves lay the same slice out for two tiles per trip: cascade2.hip)
    if (two_tile_trips && cascade2_shape(a) && (size_t)cascade2_wave_lds(a.casc_K, a.in_fmt) > b) b = (size_t)cascade2_wave_lds(a.casc_K, a.in_fmt);
    return b;
}

// the phasor table in front of the waves' slices: only chains with a mixer in front of stage 0 keep one
static inline int casc_nco_bytes(const FrontArgs &a) { return a.nco_mode != 0 ? 1024 * 8 : 0; }

template <int BPS, bool RAW0 = false, int KT = 0>
__global__ __launch_bounds__(kCascMaxWaves * 64) void k_cascade(const FrontArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    cf2 *s_nco = (cf2 *)smem;
    CascLds w;
    w.nco = s_nco;
    if (a.nco_mode != 0 && ((unsigned)(size_t)(__attribute__((address_space(3))) const void *)s_nco & 8191u) != 0u) __builtin_trap();   // nco_phasor2 ORs the index into the base
    {
        char *p = (char *)smem + (a.nco_mode != 0 ? 1024 * 8 : 0) + wave * a.casc_wave_lds;
        char *p0 = p;
#pragma unroll
        for (int k = 0; k < kCascMaxK; ++k) {
            w.XE[k] = p; w.XO[k] = p;
            if (k < a.casc_K) {
                if (k == 0) {
                    w.XE[0] = p; w.XO[0] = p + 2 * plane_stride(casc_hist_rows(a.m[0]) + 64 + 1);
                } else {
                    w.XE[k] = p; w.XO[k] = p + (((casc_lin_hs(a.m[k]) + (256 >> k)) * 8 + 15) & ~15);
                }
                p += casc_stage_bytes(k, a.m[k]);
            }
        }
        for (int i = lane; i < a.casc_wave_lds / 16; i += 64) ((float4 *)p0)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (a.nco_mode != 0) {
        const float sgn = a.nco_mode < 0 ? -1.0f : 1.0f;          // mix down: conj(phasor)
        for (int i = tid; i < 1024; i += (int)blockDim.x) { const cf2 v = a.nco_tab[i]; s_nco[i] = cf2{v.x, sgn * v.y}; }
    }
    __syncthreads();

    const int64_t gw = (int64_t)blockIdx.x * (int)(blockDim.x >> 6) + wave;
    if (gw == 0 && a.frames_in < (int64_t)a.hist_cap) {
        const int keep = a.hist_cap - (int)a.frames_in;
        for (int i = lane; i < keep; i += 64) a.hist_out[i] = a.hist_in[i + (int)a.frames_in];
    }
    if (gw < a.w_n_edge) {
        int64_t t0, t1;
        if (gw < a.w_n_edge1) { t0 = gw * a.w_edge_tpw; t1 = t0 + a.w_edge_tpw; if (t1 > a.w_edge_ta) t1 = a.w_edge_ta; }
        else { t0 = a.w_edge_tb + (gw - a.w_n_edge1) * a.w_edge_tpw; t1 = t0 + a.w_edge_tpw; if (t1 > a.w_total_tiles) t1 = a.w_total_tiles; }
        // position of this run among the dc-carry segments: edge runs of the first region, streaming runs,
        // edge runs of the second region (kernels.hpp, DcGeom mode 1)
        const int seg = (gw < a.w_n_edge1) ? (int)gw : (int)(gw + a.w_n_stream);
        casc_tiles<BPS, true, false, KT>(a, w, lane, t0 - a.w_warm_tiles, t0, t1, seg);
    } else {
        const int64_t r = gw - a.w_n_edge;
        if (r >= a.w_n_stream) return;
        const int64_t t0 = w_run_start(a, r), t1 = w_run_start(a, r + 1);
        const int seg = (int)(a.w_n_edge1 + r);
        if (BPS != 0) casc_tiles<BPS, false, RAW0, KT>(a, w, lane, t0 - a.w_warm_tiles, t0, t1, seg);
    }
}

// true when the chain's leading stages can run here (the caller still checks S >= 2 and no dc blocker)
bool cascade_supported(const int *m_run_order, int S)
{
    if (S < 2 || S - 1 > kCascMaxK) return false;
    for (int k = 0; k < S - 1; ++k) if (m_run_order[k] != 3 && m_run_order[k] != 5) return false;
    return m_run_order[S - 1] == 10;
}

// wavefronts per workgroup (one workgroup per CU): as many as the LDS slices allow, in whole waves per SIMD.
// A cascade of two or more stages is a chain of LDS round trips per tile and gains from the fourth wave per SIMD
// (config 4, K = 4: 0.787 ms with 12 waves, 0.699 ms with 16 -- which fit since stage 0's rows became planes);
// a single stage does not (config 3, K = 1: 0.223 against 0.221 ms) and keeps 12.
int cascade_waves(const FrontArgs &a)
{
    int w = (int)((160 * 1024 - casc_nco_bytes(a)) / (a.casc_wave_lds > 0 ? a.casc_wave_lds : 1));
    w &= ~3;
    const int cap = a.casc_K >= 2 ? kCascMaxWaves : kWaves;
    return w > cap ? cap : (w < 4 ? 4 : w);
}

hipError_t launch_cascade(const FrontArgs &a, hipStream_t s)
{
    if (cascade2_applies(a)) return launch_cascade2(a, s);
    const int waves = cascade_waves(a);
    const size_t lds = (size_t)casc_nco_bytes(a) + (size_t)waves * a.casc_wave_lds;
    const int64_t n_items = a.w_n_edge + a.w_n_stream;
    const unsigned grid = (unsigned)((n_items + waves - 1) / waves);
    if (grid == 0) return hipSuccess;
    int cls;
    switch (a.in_fmt) {
    case IQGPU_FMT_CS8: case IQGPU_FMT_CU8: cls = 2; break;
    case IQGPU_FMT_CS16: case IQGPU_FMT_CU16: case IQGPU_FMT_SC16Q11: cls = 4; break;
    case IQGPU_FMT_CF32: cls = 8; break;
    default: cls = 0; break;
    }
#define IQGPU_LAUNCH_CASC(...)                                                                                         \
    do {                                                                                                              \
        static LdsAttrCache cache;                /* per instantiation */                                          \
        { const hipError_t e = cache.ensure((const void *)k_cascade<__VA_ARGS__>, lds); if (e != hipSuccess) return e; } \
        hipLaunchKernelGGL((k_cascade<__VA_ARGS__>), dim3(grid), dim3(waves * 64), lds, s, a);                        \
    } while (0)
    // cu8 frames with nothing between the unpack and stage 0: the streaming waves keep them raw in LDS (casc_stage_raw8)
    const bool raw0 = a.in_fmt == IQGPU_FMT_CU8 && a.gain == 1.0f && !a.dc_enable && !a.iq_enable && a.nco_mode == 0 && !(a.dbg & kDbgNoRaw0);
    // liquid's 60 dB semi-lengths (3 .. 3 5): stage count and lengths resolved at compile time (casc_tiles, KT)
    int kt = a.casc_K;
    for (int k = 0; k < a.casc_K; ++k) if (a.m[k] != (k == a.casc_K - 1 ? 5 : 3)) kt = 0;
    if (a.dbg & kDbgNoKT) kt = 0;
#define IQGPU_LAUNCH_CASC_KT(BPS, RAW)                                                                                \
    do {                                                                                                              \
        if (kt == 1) IQGPU_LAUNCH_CASC(BPS, RAW, 1); else if (kt == 2) IQGPU_LAUNCH_CASC(BPS, RAW, 2);               \
        else if (kt == 3) IQGPU_LAUNCH_CASC(BPS, RAW, 3); else if (kt == 4) IQGPU_LAUNCH_CASC(BPS, RAW, 4);          \
        else IQGPU_LAUNCH_CASC(BPS, RAW, 0);                                                                          \
    } while (0)
    if (raw0) IQGPU_LAUNCH_CASC_KT(2, true);
    else if (cls == 2) IQGPU_LAUNCH_CASC_KT(2, false);
    else if (cls == 4) IQGPU_LAUNCH_CASC_KT(4, false);
    else if (cls == 8) IQGPU_LAUNCH_CASC_KT(8, false);
    else IQGPU_LAUNCH_CASC(0);
#undef IQGPU_LAUNCH_CASC_KT
#undef IQGPU_LAUNCH_CASC
    return hipGetLastError();
}

} // namespace iqgpu
