// cascade2.hip -- k_cascade2: k_cascade (cascade_wave.hip) for raw 8-bit / 16-bit frames with TWO 512-frame tiles per trip of a streaming wave.
//
// k_cascade hands every stage one 512-frame tile per trip, so at K = 4 its late stages run half empty: stage 2 produces one output per
// lane from 8-byte window reads, stage 3 keeps 32 lanes busy -- and an LDS instruction costs the pipe the same whatever the lanes do
// with it (profiles/r05_pmc_summary.txt, config 4: SQ_LDS_IDX_ACTIVE 184 of the 218 CU-cycles a tile takes).  With 1024 frames per
// trip every stage moves one step up the ladder of routines cascade_tiles.hpp already has:
//
//   stage 0   casc_stage_raw8 / casc_stage_raw16  twice (the two halves of the trip's raw frames: 2 / 4 KiB of LDS behind 64 / 128
//             bytes of history, 16-bit frames in two planes of alternate 16-byte blocks; 16-bit frames are unpacked WITHOUT their 2^-15, which the four outputs take at the end -- a power
//             of two commutes with every rounding on the way, so the bits are those of the scaled samples)
//   stage 1   casc_stage       (rows of four samples in two planes, four outputs per lane -- k_cascade's stage 0 on cf32 rows)
//   stage 2   casc_stage_lin<., 2>   (two outputs per lane from 16-byte reads -- k_cascade's stage 1)
//   stage 3   casc_stage_lin<., 1>   with all 64 lanes at work
//
// Same taps in the same order on the same samples as k_cascade: the bytes are equal (tests/test_gpu_parity.py).  The stages run skewed
// by one trip each (one LDS round trip per trip), K more trips drain the pipe.  Edge waves run k_cascade's own tile routine on its
// own layout inside the same slice.  A streaming run of an odd number of tiles starts one tile early (Call::plan_geometry leaves that
// tile loadable: `lead`); nothing is stored for tiles in front of the run.
#include "cascade_tiles.hpp"

namespace iqgpu {

// tiles per streaming run from which the two-tile trips are used: measured (tools/gpu/r5_casc2_min.py, cascade us per call, one- / two-tile
// trips): 2^22 frames = 2 tiles per run 28.0 / 26.3, 2^24 34.1 / 32.0, 2^25 48.7 / 41.0, 2^27 142 / 112 (cu8, K = 4); 16-bit frames alike
constexpr int kCasc2MinRun = 2;

// ---- stage 0 on raw 16-bit frames (cs16, sc16q11): 4 bytes a frame.  A lane's four outputs need its own eight frames (two 16-byte
// blocks) and the 4M - 2 in front of them (M blocks): even sample n = dword 2n, odd sample n = dword 2n + 1 of the lane's frames.
// The blocks live in TWO PLANES -- block g (frames 4g .. 4g + 3 of the trip) in plane g & 1 at position g >> 1 behind 64 bytes of
// history -- so that every window read and every write walks one plane at a 16-byte lane stride (linear, the lanes 32 bytes
// apart: SQ_LDS_BANK_CONFLICT 84 of 352 LDS cycles per trip).
constexpr int kRaw16Plane = 64 + 2048;                      // bytes of one plane: history, 128 blocks of the trip
template <int M> struct CascWinRaw16 { uint32_t W[4 * (M + 2)]; };
template <int M>
__device__ __forceinline__ void casc_stage_raw16_load(const char *RB, int lane, CascWinRaw16<M> &wn)
{
    // RB: plane 0 at the half's first block (position 0 or 64); block b of the window is block 2 lane - M + b of the half
#pragma unroll
    for (int b = 0; b < M + 2; ++b) {
        const int d = b - M;                                // block index relative to the lane's first
        const int pl = d & 1, q = (d - pl) / 2;            // plane, position relative to the lane's
        const uint4 v = *(const uint4 *)__builtin_assume_aligned(RB + pl * kRaw16Plane + 64 + (lane + q) * 16, 16);
        wn.W[4 * b + 0] = v.x; wn.W[4 * b + 1] = v.y; wn.W[4 * b + 2] = v.z; wn.W[4 * b + 3] = v.w;
    }
}
template <int M>
__device__ __forceinline__ void casc_stage_raw16_fma(const CascWinRaw16<M> &wn, const float *taps_sgpr, const float norm, v2f y[4])
{
    auto unpack = [&](uint32_t w) { return v2f{(float)(short)(w & 0xffffu), (float)(short)(w >> 16)}; };   // (x 2^15 or 2^11)
    constexpr int Z = 4 * M;                                // dword index of the lane's first frame
#pragma unroll
    for (int i = 0; i < 4 * (M + 2); ++i) keep(wn.W[i]);
    v2f E[2 * M + 3];                                       // E[k] = even sample n = k - (2M - 1)
#pragma unroll
    for (int k = 0; k < 2 * M + 3; ++k) E[k] = unpack(wn.W[Z + 2 * (k - (2 * M - 1))]);
    const v2f *hbp = (const v2f *)taps_sgpr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const v2f o = unpack(wn.W[Z + 2 * (i - M) + 1]);    // O[j - M], j = 4 lane + i
        y[i] = v2f{0.5f * o.x, 0.5f * o.y};
    }
#pragma unroll
    for (int q2 = 0; q2 < M; ++q2) {
        const v2f tp = hbp[q2];
#pragma unroll
        for (int i = 0; i < 4; ++i) pk_fma_lo_s(y[i], tp, E[2 * M - 1 + i - 2 * q2]);
#pragma unroll
        for (int i = 0; i < 4; ++i) pk_fma_hi_s(y[i], tp, E[2 * M - 1 + i - 2 * q2 - 1]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = v2f{y[i].x * norm, y[i].y * norm};
}

// BPF: bytes per frame, 2 (cu8) or 4 (cs16, sc16q11)
template <int KT, int BPF> struct Casc2 {
    static constexpr int M0 = 3, M1 = KT == 2 ? 5 : 3, M2 = KT == 3 ? 5 : 3, M3 = 5;     // liquid's 60 dB semi-lengths: 3 .. 3 5
    static constexpr int H1 = casc_hist_rows(M1);
    static constexpr int PS1 = plane_stride(H1 + 64 + 1);
    static constexpr int HIST = kRawHist;                                               // 64 bytes (per plane for 16-bit frames)
    static constexpr int HALF = 512 * BPF;                                              // bytes of one 512-frame half in memory
    static constexpr int HALF_LDS = 1024;                                               // ... and in LDS (in each plane for 16-bit frames)
    static constexpr int RAW = BPF == 2 ? HIST + 2048 : 2 * kRaw16Plane;
    static constexpr int ROWS = 4 * PS1;
    static constexpr int E2 = ((casc_lin_hs(M2) + 128) * 8 + 15) & ~15, O2 = ((casc_lin_ho(M2) + 128) * 8 + 15) & ~15;
    static constexpr int E3 = ((casc_lin_hs(M3) + 64) * 8 + 15) & ~15, O3 = ((casc_lin_ho(M3) + 64) * 8 + 15) & ~15;
    static constexpr int USED = RAW + ROWS + (KT > 2 ? E2 + O2 : 0) + (KT > 3 ? E3 + O3 : 0);
    // (16-bit frames: 20-dword windows twice and eight prefetched dwords more -- 134 registers, three waves per SIMD; the slice is
    //  padded to where cascade_waves() stops at twelve)
    static constexpr int BYTES = (BPF == 4 && USED < 10256) ? 10256 : USED;
    static constexpr int WAVES = (160 * 1024 / BYTES) >= 16 ? 16 : 12;                  // what cascade_waves() makes of BYTES
};

static inline int casc2_bpf(int fmt) { return (fmt == IQGPU_FMT_CU8 || fmt == IQGPU_FMT_CS8) ? 2 : (fmt == IQGPU_FMT_CS16 || fmt == IQGPU_FMT_SC16Q11) ? 4 : 0; }

int cascade2_wave_lds(int K, int in_fmt)
{
    const int b = casc2_bpf(in_fmt);
    if (b == 2) return K == 2 ? Casc2<2, 2>::BYTES : K == 3 ? Casc2<3, 2>::BYTES : K == 4 ? Casc2<4, 2>::BYTES : 0;
    if (b == 4) return K == 2 ? Casc2<2, 4>::BYTES : K == 3 ? Casc2<3, 4>::BYTES : K == 4 ? Casc2<4, 4>::BYTES : 0;
    return 0;
}

// the chain shape: cu8 / cs8 / cs16 / sc16q11 frames with nothing between the unpack and stage 0, two to four stages of liquid's lengths
bool cascade2_shape(const FrontArgs &a)
{
    if (casc2_bpf(a.in_fmt) == 0 || a.gain != 1.0f || a.dc_enable || a.iq_enable || a.nco_mode != 0) return false;
    if (a.dbg & (kDbgNoRaw0 | kDbgNoKT | kDbgNoCasc2)) return false;
    if (a.casc_K < 2 || a.casc_K > 4) return false;
    for (int k = 0; k < a.casc_K; ++k) if (a.m[k] != (k == a.casc_K - 1 ? 5 : 3)) return false;
    return true;
}

static int g_casc2_min_run = 0;                       // process-wide diagnostic override, set at iqgpu_chain_create
void cascade2_set_min_run(int n) { g_casc2_min_run = n; }

// ... and the call: streaming runs of two tiles and more, the tile in front of the first run's warm-up loadable, the slice sized for both layouts
bool cascade2_applies(const FrontArgs &a)
{
    // (iqgpu_debug_set("casc2_min_run", n): diagnostics -- where the two-tile trips start to pay, tools/gpu/r5_casc2_min.py)
    const int min_run = g_casc2_min_run > 0 ? g_casc2_min_run : kCasc2MinRun;
    if (!cascade2_shape(a) || a.w_n_stream <= 0 || a.w_run_q < min_run) return false;
    if ((a.w_edge_ta - a.w_warm_tiles - 1) * (int64_t)kWTile - a.rem0 < 0) return false;
    return a.casc_wave_lds >= cascade2_wave_lds(a.casc_K, a.in_fmt);
}

template <int KT, int BPF, bool S8>
__device__ __forceinline__ void casc_trips(const FrontArgs &a, char *slice, const int lane, int64_t t_begin, const int64_t t_emit0, const int64_t t_end)
{
    using G = Casc2<KT, BPF>;
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    char *RB = slice, *XE1 = RB + G::RAW, *XO1 = XE1 + 2 * G::PS1;
    char *XE2 = XE1 + G::ROWS, *XO2 = XE2 + G::E2;
    char *XE3 = XE2 + (G::E2 + G::O2), *XO3 = XE3 + G::E3;
    constexpr int HS2 = casc_lin_hs(G::M2), HO2 = casc_lin_ho(G::M2), HS3 = casc_lin_hs(G::M3), HO3 = casc_lin_ho(G::M3);
    auto ldg = [](const char *p) { return IQGPU_NT_CASC ? __builtin_nontemporal_load((const u4v *)p) : *(const u4v *)p; };

    if ((t_end - t_begin) & 1) --t_begin;                  // whole trips up to the run's end
    // lane l holds frames 8 l .. 8 l + 7 of each half of a trip (16 or 32 bytes)
    const char *src = (const char *)a.raw + (t_begin * kWTile - a.rem0) * BPF + 8 * BPF * lane;
    u4v nA = ldg(src), nB = ldg(src + G::HALF), nA2 = nA, nB2 = nB;
    if (BPF == 4) { nA2 = ldg(src + 16); nB2 = ldg(src + G::HALF + 16); }
    const float norm = a.in_fmt == IQGPU_FMT_SC16Q11 ? 1.0f / 2048.0f : 1.0f / 32768.0f;
    const int64_t o_first = (t_emit0 * kWTile) >> KT;      // first sample of the last stage this run stores
    const int woff1 = (G::H1 + (lane >> 1)) * 16 + (lane & 1) * G::PS1;       // the lane's write slot in stage 1's rows (plane lane & 1)
    const int ls1 = lane < 8 * G::H1 ? lane : 8 * G::H1 - 1;
    const int tail1 = (ls1 >= 4 * G::H1 ? G::PS1 - 16 * G::H1 : 0) + ls1 * 4;

    for (int64_t T = t_begin; T < t_end + 2 * KT; T += 2) {
        const u4v rA = nA, rB = nB, rA2 = nA2, rB2 = nB2;  // the frames of trip T: to LDS at the end of this iteration
        if (T + 2 < t_end) {
            src += 2 * G::HALF; nA = ldg(src); nB = ldg(src + G::HALF);
            if (BPF == 4) { nA2 = ldg(src + 16); nB2 = ldg(src + G::HALF + 16); }
        }

        // ------------------------------------------------------------ reads and FMAs: stage k works on trip T - 2 (k + 1)
        __builtin_amdgcn_s_setprio(1);
        CascWinRaw<G::M0> wA, wB;
        CascWinRaw16<G::M0> vA, vB;
        if (BPF == 2) {
            casc_stage_raw8_load<G::M0>(RB, lane, wA);
            casc_stage_raw8_load<G::M0>(RB + G::HALF_LDS, lane, wB);
        } else {
            casc_stage_raw16_load<G::M0>(RB, lane, vA);
            casc_stage_raw16_load<G::M0>(RB + G::HALF_LDS, lane, vB);
        }
        // the trip's last 32 frames (16-bit: the last four blocks of either plane, lanes 16 .. 31 in plane 1)
        const uint32_t hv = *(const uint32_t *)(RB + (BPF == 4 && (lane & 16) ? kRaw16Plane : 0) + 2048 + (lane & 15) * 4);
        CascWin0<G::M1> f1;
        casc_stage_load<G::M1>(XE1, XO1, lane, f1);
        const float se1 = *(const float *)(XE1 + 64 * 16 + tail1), so1 = *(const float *)(XO1 + 64 * 16 + tail1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
        v2f y0a[4], y0b[4], y1[4], y2[2] = {v2f{0.f, 0.f}, v2f{0.f, 0.f}}, y3[2] = {v2f{0.f, 0.f}, v2f{0.f, 0.f}};
        float se2 = 0.f, so2 = 0.f, se3 = 0.f, so3 = 0.f;
        if (BPF == 2) {
            casc_stage_raw8_fma<G::M0, !S8>(wA, a.casc_taps[0], y0a);
            casc_stage_raw8_fma<G::M0, !S8>(wB, a.casc_taps[0], y0b);
        } else {
            casc_stage_raw16_fma<G::M0>(vA, a.casc_taps[0], norm, y0a);
            casc_stage_raw16_fma<G::M0>(vB, a.casc_taps[0], norm, y0b);
        }
        CascWinLin<G::M2, 2> l2;
        CascWinLin<G::M3, 1> l3;
        if (KT > 2) {
            casc_stage_lin_load<G::M2, 2>(XE2, XO2, lane, l2);
            se2 = *(const float *)(XE2 + 128 * 8 + (lane < 2 * HS2 ? lane : 2 * HS2 - 1) * 4);
            so2 = *(const float *)(XO2 + 128 * 8 + (lane < 2 * HO2 ? lane : 2 * HO2 - 1) * 4);
        }
        casc_stage_fma<G::M1>(f1, a.casc_taps[1], y1);
        if (KT > 3) {
            casc_stage_lin_load<G::M3, 1>(XE3, XO3, lane, l3);
            se3 = *(const float *)(XE3 + 64 * 8 + (lane < 2 * HS3 ? lane : 2 * HS3 - 1) * 4);
            so3 = *(const float *)(XO3 + 64 * 8 + (lane < 2 * HO3 ? lane : 2 * HO3 - 1) * 4);
        }
        if (KT > 2) casc_stage_lin_fma<G::M2, 2>(l2, a.casc_taps[2], y2);
        if (KT > 3) casc_stage_lin_fma<G::M3, 1>(l3, a.casc_taps[3], y3);
        __builtin_amdgcn_wave_barrier();

        // ------------------------------------------------------------ the writes: histories slide, every stage hands its trip on
        __builtin_amdgcn_s_setprio(1);
        if (lane < (BPF == 4 ? 32 : 16)) *(uint32_t *)(RB + (BPF == 4 && (lane & 16) ? kRaw16Plane : 0) + (lane & 15) * 4) = hv;
        *(u4v *)__builtin_assume_aligned(RB + G::HIST + 16 * lane, 16) = rA;
        *(u4v *)__builtin_assume_aligned(RB + G::HIST + G::HALF_LDS + 16 * lane, 16) = rB;
        if (BPF == 4) {                                    // the lane's second block of either half: plane 1
            *(u4v *)__builtin_assume_aligned(RB + kRaw16Plane + G::HIST + 16 * lane, 16) = rA2;
            *(u4v *)__builtin_assume_aligned(RB + kRaw16Plane + G::HIST + G::HALF_LDS + 16 * lane, 16) = rB2;
        }
        {
            const int tw = (lane >= 4 * G::H1 ? G::PS1 - 16 * G::H1 : 0) + lane * 4;
            if (lane < 8 * G::H1) { *(float *)(XE1 + tw) = se1; *(float *)(XO1 + tw) = so1; }
        }
        st4a(XE1 + woff1, make_float4(y0a[0].x, y0a[0].y, y0a[2].x, y0a[2].y));
        st4a(XO1 + woff1, make_float4(y0a[1].x, y0a[1].y, y0a[3].x, y0a[3].y));
        st4a(XE1 + woff1 + 32 * 16, make_float4(y0b[0].x, y0b[0].y, y0b[2].x, y0b[2].y));
        st4a(XO1 + woff1 + 32 * 16, make_float4(y0b[1].x, y0b[1].y, y0b[3].x, y0b[3].y));
        const int64_t ob = ((T - 2 * KT) * kWTile) >> KT;  // first sample of the trip the last stage has just finished
        if (KT == 2) {
            const int64_t o = ob + 4 * lane;
            if (o >= o_first) {
                float4 *dst = (float4 *)(a.casc_out + o);
                dst[0] = make_float4(y1[0].x, y1[0].y, y1[1].x, y1[1].y);
                dst[1] = make_float4(y1[2].x, y1[2].y, y1[3].x, y1[3].y);
            }
        } else {
            if (lane < 2 * HS2) *(float *)(XE2 + lane * 4) = se2;
            if (lane < 2 * HO2) *(float *)(XO2 + lane * 4) = so2;
            st4a(XE2 + (HS2 + 2 * lane) * 8, make_float4(y1[0].x, y1[0].y, y1[2].x, y1[2].y));
            st4a(XO2 + (HO2 + 2 * lane) * 8, make_float4(y1[1].x, y1[1].y, y1[3].x, y1[3].y));
            if (KT == 3) {
                const int64_t o = ob + 2 * lane;
                if (o >= o_first) *(float4 *)(a.casc_out + o) = make_float4(y2[0].x, y2[0].y, y2[1].x, y2[1].y);
            } else {
                if (lane < 2 * HS3) *(float *)(XE3 + lane * 4) = se3;
                if (lane < 2 * HO3) *(float *)(XO3 + lane * 4) = so3;
                *(float2 *)(XE3 + (HS3 + lane) * 8) = make_float2(y2[0].x, y2[0].y);
                *(float2 *)(XO3 + (HO3 + lane) * 8) = make_float2(y2[1].x, y2[1].y);
                const int64_t o = ob + lane;
                if (o >= o_first) a.casc_out[o] = cf2{y3[0].x, y3[0].y};
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_setprio(0);
    }
}

// S8: signed 8-bit frames (cs8: a HackRF's) instead of cu8
template <int KT, int BPF, bool S8 = false>
__global__ __launch_bounds__((Casc2<KT, BPF>::WAVES * 64)) void k_cascade2(const FrontArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char *slice = (char *)smem + wave * a.casc_wave_lds;
    for (int i = lane; i < a.casc_wave_lds / 16; i += 64) ((float4 *)slice)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __builtin_amdgcn_wave_barrier();

    const int64_t gw = (int64_t)blockIdx.x * (int)(blockDim.x >> 6) + wave;
    if (gw < a.w_n_edge) {
        // k_cascade's edge runs on k_cascade's layout (no mixer in this shape: no phasor table)
        CascLds w;
        w.nco = nullptr;
        char *p = slice;
#pragma unroll
        for (int k = 0; k < kCascMaxK; ++k) {
            w.XE[k] = p; w.XO[k] = p;
            if (k < KT) {
                const int m = k == KT - 1 ? 5 : 3;
                if (k == 0) w.XO[0] = p + 2 * plane_stride(casc_hist_rows(m) + 64 + 1);
                else w.XO[k] = p + (((casc_lin_hs(m) + (256 >> k)) * 8 + 15) & ~15);
                p += casc_stage_bytes(k, m);
            }
        }
        int64_t t0, t1;
        if (gw < a.w_n_edge1) { t0 = gw * a.w_edge_tpw; t1 = t0 + a.w_edge_tpw; if (t1 > a.w_edge_ta) t1 = a.w_edge_ta; }
        else { t0 = a.w_edge_tb + (gw - a.w_n_edge1) * a.w_edge_tpw; t1 = t0 + a.w_edge_tpw; if (t1 > a.w_total_tiles) t1 = a.w_total_tiles; }
        const int seg = (gw < a.w_n_edge1) ? (int)gw : (int)(gw + a.w_n_stream);
        casc_tiles<BPF, true, false, KT>(a, w, lane, t0 - a.w_warm_tiles, t0, t1, seg);
    } else {
        const int64_t r = gw - a.w_n_edge;
        if (r >= a.w_n_stream) return;
        const int64_t t0 = w_run_start(a, r), t1 = w_run_start(a, r + 1);
        casc_trips<KT, BPF, S8>(a, slice, lane, t0 - a.w_warm_tiles, t0, t1);
    }
}

hipError_t launch_cascade2(const FrontArgs &a, hipStream_t s)
{
    if (!cascade2_applies(a)) return hipErrorInvalidValue;
    const int waves = cascade_waves(a);
    const size_t lds = (size_t)waves * a.casc_wave_lds;
    const int64_t n_items = a.w_n_edge + a.w_n_stream;
    const unsigned grid = (unsigned)((n_items + waves - 1) / waves);
    if (grid == 0) return hipSuccess;
#define IQGPU_LAUNCH_CASC2(KT, BPF, S8)                                                                                \
    do {                                                                                                              \
        static LdsAttrCache cache;                                                                                    \
        if (waves > Casc2<KT, BPF>::WAVES) return hipErrorInvalidValue;                                               \
        { const hipError_t e = cache.ensure((const void *)k_cascade2<KT, BPF, S8>, lds); if (e != hipSuccess) return e; } \
        hipLaunchKernelGGL((k_cascade2<KT, BPF, S8>), dim3(grid), dim3(waves * 64), lds, s, a);                       \
    } while (0)
    if (a.in_fmt == IQGPU_FMT_CS8) {
        if (a.casc_K == 2) IQGPU_LAUNCH_CASC2(2, 2, true);
        else if (a.casc_K == 3) IQGPU_LAUNCH_CASC2(3, 2, true);
        else IQGPU_LAUNCH_CASC2(4, 2, true);
    } else if (casc2_bpf(a.in_fmt) == 2) {
        if (a.casc_K == 2) IQGPU_LAUNCH_CASC2(2, 2, false);
        else if (a.casc_K == 3) IQGPU_LAUNCH_CASC2(3, 2, false);
        else IQGPU_LAUNCH_CASC2(4, 2, false);
    } else {
        if (a.casc_K == 2) IQGPU_LAUNCH_CASC2(2, 4, false);
        else if (a.casc_K == 3) IQGPU_LAUNCH_CASC2(3, 4, false);
        else IQGPU_LAUNCH_CASC2(4, 4, false);
    }
#undef IQGPU_LAUNCH_CASC2
    return hipGetLastError();
}

} // namespace iqgpu
