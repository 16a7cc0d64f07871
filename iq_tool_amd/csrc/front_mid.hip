// front_mid.hip -- k_front_mid: the NRSC-5 preset shape with SIX half-band outputs per lane, 12 waves per CU (3 per SIMD,
// <= 168 VGPRs), 768-frame tiles.
//
// Why a third shape.  On gfx950 one wave issues at most one instruction per four cycles, whatever the instruction, while a SIMD
// executes a plain VALU instruction in two and a packed one in four: k_front_fat's two waves per SIMD (front_fat.hip) do a
// third less work per frame than k_front_s1's four, but half of every SIMD's issue slots stay empty whenever one of the two
// waits -- 0.407 ms against 0.437.  k_front_s1 itself gains nothing from its last four waves (8 / 12 / 16 waves per CU: 0.507 /
// 0.452 / 0.441 ms), i.e. it is bound by the work, not by latency.  This kernel keeps k_front_fat's formulation (a lane owns a
// run of consecutive half-band outputs, polyphase slots with shifted zero-padded taps from the tap planes, LDS batches issued
// a phase ahead) at the widest run that still leaves three waves per SIMD:
//
//   * a lane owns 6 consecutive half-band outputs: 25 / 6 = 4.2 reads per even sample (8 per lane: 3.4, 4 per lane: 5.75),
//     FOUR polyphase slots per 6 samples (a lane's six hold 3 or 4 outputs for step / 2^24 >= 1.5): 63 FMAs per 6 samples
//     (8 per lane: 79 per 8, 4 per lane: 56 per 4);
//   * rows need no padding: 6 cf32 are 48 bytes = 3 slots of 16, an odd number, so the sample streams are plain linear arrays
//     (sample n at byte 8 n) and every window read is a conflict-free ds_read_b128 at an even sample; the writes of 8
//     consecutive lanes are 128 contiguous bytes;
//   * tile = 768 frames = 3 coalesced 16-byte loads per lane; 6.7 KB of LDS per wave.
//
// Order of a tile (every LDS batch one FMA run ahead of its use, the same window registers never live across two runs):
//     pointwise(T) -> X(T) in LDS | polyphase(T - 1) + pack + store | raw frames of T + 1, half-band window reads | NCO phasors of T + 1 |
//     half-band(T) | HB(T) -> LDS, polyphase window + first taps of T
// Arithmetic and summation order are k_front_s1's: bit-identical output (test_fat_kernel_equals_the_sixteen_wave_kernel).
// Edge tiles run on the scalar-load run_tiles (front_tiles.hpp): runs of two 768-frame tiles = three of its 512-frame ones.
#include "front_tiles.hpp"
#include "front_fat_common.hpp"

namespace iqgpu {

#ifndef IQGPU_MID_WAVES
#define IQGPU_MID_WAVES 12
#endif
constexpr int kMidN16 = 116;                  // INF of k_front_mid: 16-bit frames normalised at the unpack (a gain, sc16q11)
constexpr int kMidWaves = IQGPU_MID_WAVES;    // (16 is an experiment: only the shape without a mixer fits LDS and 128 VGPRs then)
constexpr int kMidThreads = kMidWaves * 64;
// NL = half-band outputs per lane: 6 (768-frame tiles; rows need no padding: 6 cf32 = 48 bytes = 3 slots of 16, an odd number, so a
// stream is a plain linear array, sample at row coordinate r at byte 8 r) or 8 (1024-frame tiles; rows of 8 cf32 padded to 80 bytes =
// 5 slots).  Either way a lane's windows are conflict-free ds_read_b128 at constant offsets from the lane's base (ROWB * lane).
template <int NL> struct MidGeom {
    static constexpr int NC = NL / 2;                       // 256-frame chunks per tile = 16-byte loads per lane and stream
    static constexpr int TILE = 128 * NL;                   // input frames per tile
    static constexpr int HB = 64 * NL;                      // half-band samples per tile
    static constexpr int XH = 24;                           // samples of history in front of a tile, even stream ...
    static constexpr int HH = NL == 6 ? 18 : 16;            // ... and half-band stream (whole rows)
    static constexpr int ROWB = NL == 6 ? 48 : 80;          // bytes from one lane's run to the next
    static constexpr int NS = NL == 6 ? 4 : 5;              // polyphase slots per lane
    static constexpr int CHUNKB = NL == 6 ? 1024 : 16 * 80; // bytes from chunk c's write slot to chunk c + 1's
    static constexpr int XBYTES = NL == 6 ? (XH + HB) * 8 : (XH + HB) / 8 * 80;   // the stream (the half-band stream lives on top of it; no XO stream)
    __host__ __device__ static constexpr int co(int k) { return NL == 6 ? 8 * k : (k / 8) * 80 + (k % 8) * 8; }   // byte offset of row coordinate k
};
constexpr int kMidWaveLds = MidGeom<8>::XBYTES;             // (slices sized for the larger shape)
constexpr int kMidEdgeMax = 4;                              // edge waves of a launch (they use k_front_s1's slice layout, in an arena of their own)
constexpr int kMidEdgeLds = kMidEdgeMax * kWaveLds;
constexpr int kMidNcoLds = 2 * 1024 * 8;
constexpr int kMidArbLds = 256 * 14 * 4;                    // the edge waves' table (layout of k_front_s1)
constexpr int kMidTabLds = kMidNcoLds + kMidArbLds + kFTapLds;
static_assert(kMidWaves > 12 || kMidTabLds + kMidWaves * kMidWaveLds + kMidEdgeLds + 16 <= 160 * 1024, "LDS");

int front_mid_waves() { return kMidWaves; }
int front_mid_max_edge_waves() { return kMidEdgeMax; }
static size_t mid_lds_bytes(int nl, bool nonco)
{
    // (+ 16: the workgroup's run counter of the fixed-run mode, behind everything else -- a static __shared__ word would move the
    //  dynamic block off the 8 KB boundary nco_phasor2 relies on)
    return (size_t)kMidTabLds - (nonco ? kMidNcoLds : 0) + (size_t)kMidWaves * (nl == 8 ? MidGeom<8>::XBYTES : MidGeom<6>::XBYTES) + kMidEdgeLds + 16;
}   // (no NCO tables without a mixer)

__device__ __forceinline__ float wave_max_f(float m)
{
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) { const float o = __shfl_xor(m, k); m = o > m ? o : m; }
    return m;
}

struct MidLds { char *XE; const cf2 *nco; unsigned tap_lds; };

// Tiles [T_begin, T_emit1) of 768 frames; those from T_emit0 on produce output.  Every tile, and the one behind the last
// (prefetch), lies inside the call's new, 16-byte aligned frames and outside the history the call leaves behind.
// L3 = floor(3 step / 2^24) of the step class (lo_0 .. lo_2 = 0, 1, 3).
// STEAL (run stealing, kernels.hpp): the run's end is whatever the wave's descriptor `desc` says when a tile is claimed -- one
// returning agent-scope add per tile, issued in front of the tile and read behind it (a tile is 3 us, the add comes back in 1).
// CF32OUT: the outputs leave as cf32 (a user filter behind the resampler: the -usb / -lsb presets) instead of packed cs16
// INF / OUT8 (late round 5): 8-bit frames in (IQGPU_FMT_CU8 / _CS8 instead of _CS16: unpacked to normalised floats, so nothing rides on
// the table or the taps), 16-bit frames with a gain or of the sc16q11 scale (kMidN16: normalised and gained by one product at the
// unpack) and 8-bit frames out (1 = cu8, 2 = cs8; 0 = cs16 or cf32 as CF32OUT says)
template <int NL, bool NONCO, int L3, int L4, bool AGC, bool STEAL, bool CF32OUT = false, int INF = IQGPU_FMT_CS16, int OUT8 = 0>
__device__ __forceinline__ void run_mid(const FrontArgs &a, const MidLds &w, const int lane,
                                        const int64_t T_begin, const int64_t T_emit0, int64_t T_emit1, unsigned long long *const desc)
{
    typedef MidGeom<NL> G;
    constexpr int NS = G::NS;
#ifdef IQGPU_DIAG_LEAN
    constexpr bool kLean = true;                            // DIAGNOSTIC: the order without anything fetched ahead at 12 waves per CU
#else
    constexpr bool kLean = kMidWaves > 12 || NL == 8;       // (8 per lane at 3 waves per SIMD: 168 VGPRs leave no room to fetch a phase ahead)
#endif
    constexpr int LO[5] = {0, 1, 3, L3, L4};
    auto addr_rt = [](int rc) { return NL == 6 ? 8 * rc : (rc >> 3) * 80 + (rc & 7) * 8; };
    char *XE = w.XE, *HB = w.XE;
    uint32_t step = a.step;
    if (STEAL) asm volatile("" : "+s"(step));         // (opaque per run: the reciprocals of the prologue's divisions are not held across the caller's loop)
    float hb[20];
#pragma unroll
    for (int k = 0; k < 20; ++k) hb[k] = a.hb0[k];

    // ---- output bookkeeping of the polyphase tile T_emit0 (wave-uniform), then per lane
    constexpr uint64_t SPAN = (uint64_t)G::HB << 24;
    uint64_t k_tile0 = first_k_at((uint64_t)T_emit0 * SPAN, a.phi0, step);
    uint32_t delta0 = (uint32_t)(a.phi0 + k_tile0 * (uint64_t)step - (uint64_t)T_emit0 * SPAN);            // < step
    const uint32_t n_est = (uint32_t)(SPAN / step);                      // outputs of a tile: n_est or n_est + 1
    const uint32_t span_rem = (uint32_t)(SPAN - (uint64_t)n_est * step);   // < step: a tile holds n_est + 1 outputs iff its first one starts below this
    uint32_t n0, Pl;                                   // the lane's first output of the tile: index in the tile, phase from sample 6 lane
    {
        const uint64_t tgt = (uint64_t)(NL * lane) << 24;
        const uint64_t nn = tgt > delta0 ? (tgt - delta0 + step - 1) / step : 0;
        n0 = (uint32_t)nn;
        Pl = (uint32_t)((uint64_t)delta0 + nn * step - tgt);
    }

    // fused output AGC of the locked phase (as in k_front_s1, front_tiles.hpp): the gain from the device state multiplies every
    // output before the pack, and max |y|^2 of the chunk the run is in and of the next one is kept per lane -- in FLOAT here
    // (k_front_s1 keeps it exactly, in double, at 11 % of its time): the verifier only compares the peak with two thresholds,
    // and a chunk within a few ulp of either goes to the exact kernels (k_agc_classify, peak_approx), so the bytes and the
    // state that come out are the same.  A chunk ends where input frame (c + 1) chunk - 1 completes a half-band sample: at most
    // one boundary per tile (chunk >= 768)
    float agc_g = 1.0f;
    float agc_m0 = 0.0f, agc_m1 = 0.0f;
    int64_t agc_c = 0, agc_B = 0, agc_T = T_emit0;
    const int AS = AGC ? a.agc_shift : 0;
    if (AGC) {
        agc_g = a.agc_state->gain;
        const int64_t F0 = (((int64_t)G::HB * T_emit0 + 1) << AS) - 1 - a.agc_rem;
        agc_c = F0 > 0 ? F0 / a.agc_chunk_frames : 0;
        agc_B = (agc_c + 1) * a.agc_chunk_frames;
    }

    // ---- per-lane LDS offsets (bytes; a sample at row coordinate r sits at 8 r)
    const int wq = addr_rt(G::XH + 2 * lane);                           // write slot of chunk 0: even samples 2 lane, 2 lane + 1 (chunk c: CHUNKB c on)
    const int sl_src = addr_rt(G::HB + (lane >> 1)) + 4 * (lane & 1);   // one dword per lane: the stream's tail becomes the next tile's history
    const int sl_dst = addr_rt(lane >> 1) + 4 * (lane & 1);
    const char *we = XE + G::ROWB * lane, *wh = HB + G::ROWB * lane;
    typedef __attribute__((address_space(3))) const v2f lds_v2f;

    // Raw frames: the coalesced stream (lane: frames 256 c + 4 lane .. + 3 of the tile) feeds the EVEN samples, which every lane's
    // window needs and which therefore go through LDS.  The ODD samples meet one tap only, the centre tap of one output -- output
    // i of the lane takes the odd sample 6 lane + i - 10, frame 12 lane - 19 + 2 i of the tile -- so the lane fetches its own six
    // straight from memory a second time (three more 16-byte loads at frame 12 lane - 20: the lines are in L2 / L1 from the
    // coalesced loads, HBM sees them once) instead of writing them to LDS for another lane to read: no XO stream at all.
    RawChunk nxt[G::NC], nxo[G::NC];
    v2f cs_n[G::NC][2], cs_o[NL];
    constexpr bool IN8 = INF == IQGPU_FMT_CU8 || INF == IQGPU_FMT_CS8;      // 2-byte frames: the same frames per lane from 8-byte loads
    constexpr bool N16 = INF == kMidN16;                 // 16-bit frames with a gain, or sc16q11: normalised (and gained) at the unpack
    constexpr bool NORM = IN8 || N16;                    // the samples are normalised floats: nothing rides on the table or the taps
    constexpr int VB = IN8 ? 2 : 4;
    // (s 2^-15) g as s (g 2^-15): the scale is exact, so the one product rounds to what the reference's two do
    const float sc16 = a.gain * (a.in_fmt == IQGPU_FMT_SC16Q11 ? 1.0f / 2048.0f : 1.0f / 32768.0f);
    const bool gained = a.gain != 1.0f;
    auto load_even = [&](int64_t T) {
        const char *src = (const char *)a.raw + (T * G::TILE - a.rem0) * VB + 4 * VB * lane;
#pragma unroll
        for (int c = 0; c < G::NC; ++c) load_chunk<VB>(src + 256 * VB * c, nxt[c]);
    };
    auto load_odd = [&](int64_t T) {
        const char *src = (const char *)a.raw + (T * G::TILE - a.rem0) * VB + 2 * VB * NL * lane - 20 * VB;
#pragma unroll
        for (int c = 0; c < G::NC; ++c) load_chunk<VB>(src + 4 * VB * c, nxo[c]);
    };
    // one frame (its 16 bits in the low half of h) as floats: 8-bit frames normalised the way unpack_chunk does it (cu8: one fused
    // multiply-add whose product and sum are exact), 16-bit frames as integers (their 2^-15 rides on the table / the taps)
    auto unp8 = [](uint32_t h) {
        if (INF == IQGPU_FMT_CU8)
            return v2f{__builtin_fmaf((float)(h & 0xffu), 1.0f / 128.0f, -127.5f / 128.0f), __builtin_fmaf((float)((h >> 8) & 0xffu), 1.0f / 128.0f, -127.5f / 128.0f)};
        return v2f{(float)(signed char)(h & 0xffu) * (1.0f / 128.0f), (float)(signed char)((h >> 8) & 0xffu) * (1.0f / 128.0f)};
    };
    // phase of a frame = theta0 + frame * dtheta (mod 2^32): the lane's share of the product once per run, the tile's share on the
    // scalar unit -- a v_mul_lo_u32 per chunk and tile is a quarter-rate instruction each (4 of the tile's ~340 VALU instructions,
    // the time of 16)
    const uint32_t th_lane = a.nco_theta0 + (uint32_t)(4 * lane) * a.nco_dtheta;
    const uint32_t tho_lane = a.nco_theta0 + (uint32_t)(2 * NL * lane - 19) * a.nco_dtheta;
#ifdef IQGPU_DIAG_NCOHOLD
    // DIAGNOSTIC build (timing only, wrong bytes): the phasors are looked up for the run's first tile and then HELD -- the most a
    // scheme that re-reads a lane's phasors only when their table index moves on could save (round 6, profiles/r06_headline.md)
    bool diag_nco_first = true;
#endif
    auto nco_lookup = [&](int64_t T) {
#ifdef IQGPU_DIAG_NCOHOLD
        if (!diag_nco_first) {
#pragma unroll
            for (int c = 0; c < G::NC; ++c) { asm volatile("" : "+v"(cs_n[c][0])); asm volatile("" : "+v"(cs_n[c][1])); }
#pragma unroll
            for (int i = 0; i < NL; ++i) asm volatile("" : "+v"(cs_o[i]));
            return;
        }
        diag_nco_first = false;
#endif
        const uint32_t tb = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(T * G::TILE) * a.nco_dtheta));
#pragma unroll
        for (int c = 0; c < G::NC; ++c) {
            const uint32_t th = th_lane + (tb + (uint32_t)(256 * c) * a.nco_dtheta);
            cs_n[c][0] = nco_phasor2(w.nco, th, 0);
            cs_n[c][1] = nco_phasor2(w.nco, th + 2u * a.nco_dtheta, 0);
        }
        uint32_t tho = tho_lane + tb;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            cs_o[i] = nco_phasor2(w.nco, tho, 1);               // the odd stream only meets the centre tap 0.5: half-scaled copy
            tho += 2u * a.nco_dtheta;
        }
    };
    load_even(T_begin); load_odd(T_begin);
    if (!NONCO) nco_lookup(T_begin);

    float sl_e = 0.f, sl_h = 0.f;
    v2f own[NL];                                       // the lane's own half-band outputs of the tile before
    v2f Hw[14];                                        // the 13 half-band samples in front of them (+ one unused)
    v2f tp[2][8], tq[2][8];                            // taps of slots 0, 1 (issued a phase ahead) and 2, 3
    unsigned trow[NS];
    v2f E[NL + 20], acc[NL], y[NS];
#pragma unroll
    for (int i = 0; i < NL; ++i) own[i] = v2f{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 14; ++i) Hw[i] = v2f{0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) tp[j][i] = v2f{0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NS; ++j) trow[j] = w.tap_lds;

    // (the empty asm with a memory clobber orders the LDS accesses around it already when the instruction stream is first laid
    //  out -- sched_barrier by itself only stops the machine scheduler, and the loads had floated above it before that)
#define FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#ifdef IQGPU_DIAG_NOGATHER
    // DIAGNOSTIC build (timing only, wrong bytes): the tap gather runs ONCE per run and the taps stay in their registers -- what a
    // polyphase stage that re-reads a slot only when its arm moves on could save at most (round 5, profiles/r05_headline.md)
    bool diag_first = true;
    auto taps = [&](v2f (&t)[8], unsigned row) {
        if (diag_first) {
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = *(lds_v2f *)(size_t)(row + tap_pair_off(i));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(t[i]));
    };
#else
    auto taps = [&](v2f (&t)[8], unsigned row) {
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = *(lds_v2f *)(size_t)(row + tap_pair_off(i));
    };
#endif
    // ---- pointwise, chunk by chunk: unpack, mix, -> XE / XO (on top of the half-band stream: its window reads are issued)
    auto VL_point = [&]() {
        if (lane < 48) *(float *)(XE + sl_dst) = sl_e;
#pragma unroll
        for (int c = 0; c < G::NC; ++c) {
            v2f x0, x2;
            if (IN8) {
                x0 = unp8(nxt[c].w[0]); x2 = unp8(nxt[c].w[1]);      // frames 0 and 2 of the lane's four: the low halves of its two words
                if (gained) { x0 = v2f{x0.x * a.gain, x0.y * a.gain}; x2 = v2f{x2.x * a.gain, x2.y * a.gain}; }      // (wave-uniform)
            } else {
            x0 = v2f{(float)(short)(nxt[c].w[0] & 0xffffu), (float)(short)(nxt[c].w[0] >> 16)};      // 2^-15: in the table (taps when NONCO)
            x2 = v2f{(float)(short)(nxt[c].w[2] & 0xffffu), (float)(short)(nxt[c].w[2] >> 16)};
            keep(nxt[c].w[1]); keep(nxt[c].w[3]);       // (whole 16-byte loads: unused words declared used, or hipcc narrows them to dword loads)
            if (N16) { x0 = v2f{x0.x * sc16, x0.y * sc16}; x2 = v2f{x2.x * sc16, x2.y * sc16}; }
            }
            if (!NONCO) { x0 = pk_cmul(x0, cs_n[c][0]); x2 = pk_cmul(x2, cs_n[c][1]); }
            stq(XE + wq + G::CHUNKB * c, make_float4(x0.x, x0.y, x2.x, x2.y));
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 48) sl_e = *(const float *)(XE + sl_src);
    };
    // ---- the half-band window: E[j] = even sample at row coordinate 6 lane + 4 + j, output i uses E[20 + i - k], k = 0 .. 19
    auto L_xr = [&]() {
#pragma unroll
        for (int q = 0; q < (NL + 20) / 2; ++q) {
            const float4 v = ldq(we + G::co(4 + 2 * q));
            E[2 * q] = v2f{v.x, v.y}; E[2 * q + 1] = v2f{v.z, v.w};
        }
        keep(E[0]);
    };
    // ---- centre taps: the lane's own six odd samples (words 1, 3, .. 11 of its second load), mixed with the half-scaled table
    auto V_centre = [&]() {
        if (!IN8) {
#pragma unroll
            for (int c = 0; c < G::NC; ++c) { keep(nxo[c].w[0]); keep(nxo[c].w[2]); }
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            v2f o;
            if (IN8) {
                o = unp8(nxo[i >> 1].w[i & 1] >> 16);           // frame 2 i + 1 of the lane's twelve: the high half of word i
                if (gained) o = v2f{o.x * a.gain, o.y * a.gain};
            } else {
                const uint32_t wd = nxo[(2 * i + 1) >> 2].w[(2 * i + 1) & 3];
                o = v2f{(float)(short)(wd & 0xffffu), (float)(short)(wd >> 16)};
                if (N16) o = v2f{o.x * sc16, o.y * sc16};
            }
            if (NONCO) { const float hc = NORM ? 0.5f : 0.5f / 32768.0f; acc[i] = v2f{hc * o.x, hc * o.y}; }
            else acc[i] = pk_cmul(o, cs_o[i]);
        }
    };
    // ---- pack + store of the polyphase tile, on to the next one, its tap rows
    auto V_emit = [&]() {
        if (AGC) {
            uint32_t agc_qb = (uint32_t)G::HB;              // half-band samples of this tile below it are in chunk agc_c
            const int64_t F0 = (((int64_t)G::HB * agc_T + 1) << AS) - 1 - a.agc_rem;    // last input frame that sample 0 of the tile needs
            if (F0 >= agc_B) {                              // the boundary fell between two tiles
                const double m = (double)wave_max_f(agc_m0);
                if (lane == 0 && m > 0.0) atomicMax(a.agc_peak2 + agc_c, (unsigned long long)__double_as_longlong(m));
                agc_m0 = 0.0f; agc_c += 1; agc_B += a.agc_chunk_frames;
            }
            const int64_t d = agc_B - F0;
            if (d < ((int64_t)G::HB << AS)) agc_qb = (uint32_t)((d + ((int64_t)1 << AS) - 1) >> AS);
            uint32_t P = Pl;
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const uint32_t pos = P >> 24;               // the output's half-band sample among the lane's own (last slot: may be past them)
                if (j < NS - 1 || pos < (uint32_t)NL) {
                    // agc_apply: peak of the chunk over the samples BEFORE the gain, then samples[i] *= g (src/agc.c:169-214)
                    const float m2 = fmaf(y[j].x, y[j].x, y[j].y * y[j].y);
                    if ((uint32_t)(NL * lane) + pos < agc_qb) agc_m0 = fmaxf(agc_m0, m2); else agc_m1 = fmaxf(agc_m1, m2);
                }
                y[j] = v2f{y[j].x * agc_g, y[j].y * agc_g};
                P += step;
            }
            if (agc_qb < (uint32_t)G::HB) {                 // the tile held a boundary: chunk agc_c is complete for this run
                const double m = (double)wave_max_f(agc_m0);
                if (lane == 0 && m > 0.0) atomicMax(a.agc_peak2 + agc_c, (unsigned long long)__double_as_longlong(m));
                agc_m0 = agc_m1; agc_m1 = 0.0f; agc_c += 1; agc_B += a.agc_chunk_frames;
            }
            agc_T += 1;
        }
        if constexpr (CF32OUT) {
            // the lane's NS - 1 or NS outputs are consecutive: 8-byte aligned cf32, NS - 1 of them always
            typedef float f32x2 __attribute__((ext_vector_type(2), aligned(8)));
            char *ob = (char *)a.out + ((int64_t)k_tile0 + n0) * 8;
            if constexpr (NS == 4) {
                // Round 6: two whole 16-byte pieces per lane instead of three 8-byte ones and a masked fourth.  A lane whose fourth slot
                // holds no output stores its UPPER NEIGHBOUR's first output there (two wave_shl:1 DPP moves): the very sample the
                // neighbour writes to the very same address in the very same instruction, so every lane but the wave's last can always
                // store 32 bytes and the tile leaves as one hole-free range in two instructions.  Same-box A/B, four rounds of 60 steps
                // on the cs16-fm-nrsc5-usb preset: 0.4116 -> 0.3945 ms (-4.2 %), bytes unchanged.  (The cs16 form of the same idea is
                // byte-identical too and gains nothing -- profiles/r06_headline.md (c) -- and is not in the shipped kernel.)
                typedef float f32x4 __attribute__((ext_vector_type(4), aligned(8)));
                const bool has_last = Pl + (uint32_t)(NS - 1) * step < ((uint32_t)NL << 24);
                const float ux = dpp_f<0x130, 0xf, false>(0.0f, y[0].x), uy = dpp_f<0x130, 0xf, false>(0.0f, y[0].y);      // wave_shl:1
                const v2f tail = has_last ? y[NS - 1] : v2f{ux, uy};
                if (lane < 63 || has_last) {
                    *(f32x4 *)ob = f32x4{y[0].x, y[0].y, y[1].x, y[1].y};
                    *(f32x4 *)(ob + 16) = f32x4{y[2].x, y[2].y, tail.x, tail.y};
                } else {
#pragma unroll
                    for (int j = 0; j < NS - 1; ++j) *(f32x2 *)(ob + 8 * j) = f32x2{y[j].x, y[j].y};
                }
            } else {
#pragma unroll
            for (int j = 0; j < NS - 1; ++j) *(f32x2 *)(ob + 8 * j) = f32x2{y[j].x, y[j].y};
            if (Pl + (uint32_t)(NS - 1) * step < ((uint32_t)NL << 24)) *(f32x2 *)(ob + 8 * (NS - 1)) = f32x2{y[NS - 1].x, y[NS - 1].y};
            }
        } else if constexpr (OUT8 != 0) {
            // 2-byte frames at a 2-byte-aligned address: the lane's NS - 1 or NS outputs as dwords and a short
            static_assert(NS == 4, "8-bit output: six outputs per lane only");
            typedef uint32_t u32a2 __attribute__((aligned(2)));
            uint32_t pk[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) pk[j] = pack_b8(cf2{y[j].x, y[j].y}, OUT8 == 1);
            char *ob = (char *)a.out + ((int64_t)k_tile0 + n0) * 2;
            *(u32a2 *)ob = pk[0] | (pk[1] << 16);
            if (Pl + (uint32_t)(NS - 1) * step < ((uint32_t)NL << 24)) *(u32a2 *)(ob + 4) = pk[2] | (pk[3] << 16);
            else *(uint16_t *)(ob + 4) = (uint16_t)pk[2];
        } else {
        uint32_t pk[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) pk[j] = pack_cs16(cf2{y[j].x, y[j].y});
        // the lane's NS - 1 or NS outputs are consecutive: one 12- or 16-byte store and at most one dword
        typedef uint32_t u32x3 __attribute__((ext_vector_type(3), aligned(4)));
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
        char *ob = (char *)a.out + ((int64_t)k_tile0 + n0) * 4;
#ifdef IQGPU_DIAG_NOSTORE
        // DIAGNOSTIC build (timing only, no output): what do the cs16 stores -- a 12-byte piece per lane and a masked dword in the holes,
        // i.e. short segments per wave-instruction -- cost this kernel?  8.7 % (0.337 against 0.369 ms); and the SHAPE is not what costs:
        // one whole 16-byte piece per lane (a lane without a fourth output repeating its upper neighbour's first code, by a DPP move) was
        // built, byte-identical, and timed the same (tools/gpu/r6_store_shape.patch; round 6, profiles/r06_headline.md)
#pragma unroll
        for (int j = 0; j < NS; ++j) asm volatile("" :: "v"(pk[j]));
        keep(ob);
#else
        if constexpr (NS == 4) *(u32x3 *)ob = u32x3{pk[0], pk[1], pk[2]};
        else *(u32x4 *)ob = u32x4{pk[0], pk[1], pk[2], pk[3]};
        if (Pl + (uint32_t)(NS - 1) * step < ((uint32_t)NL << 24)) *(uint32_t *)(ob + 4 * (NS - 1)) = pk[NS - 1];
#endif
        }
        // (all in 32 bits: delta0 + n_est step < SPAN <=> delta0 < span_rem, and nt step - SPAN is step - span_rem or -span_rem)
        const bool more = delta0 < span_rem;
        k_tile0 += n_est + (more ? 1u : 0u);
        const int32_t e = more ? (int32_t)(step - span_rem) : -(int32_t)span_rem;               // |e| < step
        delta0 = (uint32_t)((int32_t)delta0 + e);
        int32_t pl = (int32_t)Pl + e;
        if (pl < 0) { pl += (int32_t)step; n0 += 1u; }
        else if (pl >= (int32_t)step) { pl -= (int32_t)step; n0 -= 1u; }
        Pl = (uint32_t)pl;
    };
    auto V_taprows = [&]() {
        uint32_t P = Pl;
        if (a.tap_fold) {       // (wave-uniform: a scalar branch)
#pragma unroll
            for (int j = 0; j < NS; ++j) { trow[j] = tap_row<true>(w.tap_lds, P, LO[j]); P += step; }
        } else {
#pragma unroll
            for (int j = 0; j < NS; ++j) { trow[j] = tap_row<false>(w.tap_lds, P, LO[j]); P += step; }
        }
    };
    auto V_hb = [&]() {
#pragma unroll
        for (int q2 = 0; q2 < 10; ++q2) {
#pragma unroll
            for (int i = 0; i < NL; ++i) acc[i] = fma2(hb[2 * q2], E[20 + i - 2 * q2], acc[i]);
#pragma unroll
            for (int i = 0; i < NL; ++i) acc[i] = fma2(hb[2 * q2 + 1], E[19 + i - 2 * q2], acc[i]);
        }
    };
    // ---- half-band outputs -> LDS (on top of XE: its reads for this tile are long issued); the lane keeps its own six; then
    //      the polyphase window of this tile: Hw[i] = half-band sample at row coordinate 6 lane + 4 + i (m = i - 14), and the
    //      taps of its first two slots
    auto L_hb = [&](const bool with_pp_reads) {
        if (lane < 2 * G::HH) *(float *)(HB + sl_dst) = sl_h;
#pragma unroll
        for (int q = 0; q < NL / 2; ++q)
            stq(HB + G::ROWB * lane + G::co(G::HH + 2 * q), make_float4(acc[2 * q].x, acc[2 * q].y, acc[2 * q + 1].x, acc[2 * q + 1].y));
        __builtin_amdgcn_wave_barrier();
        if (lane < 2 * G::HH) sl_h = *(const float *)(HB + sl_src);
#pragma unroll
        for (int i = 0; i < NL; ++i) own[i] = acc[i];
        if (with_pp_reads) {
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                const float4 v = ldq(wh + G::co(G::HH - 14 + 2 * q));
                Hw[2 * q] = v2f{v.x, v.y}; Hw[2 * q + 1] = v2f{v.z, v.w};
            }
            keep(Hw[0]);
            if (!kLean) { taps(tp[0], trow[0]); taps(tp[1], trow[1]); }
        }
    };

    // the polyphase of a tile with its taps loaded where they are used, two slots at a time
    auto V_pp_lean = [&]() {
        taps(tp[0], trow[0]); taps(tp[1], trow[1]);
        pp_slots2<NL, 0, 1>(Hw, own, tp[0], tp[1], y[0], y[1]);
        FENCE();
#ifdef IQGPU_DIAG_NOGATHER
        if constexpr (NS == 4) {                      // (four register sets: every slot keeps its own)
#ifdef IQGPU_DIAG_NOGATHER2
            // ... or slots 0 and 1 only: slots 2 and 3 are gathered every tile as in the shipped kernel (fits 168 VGPRs without spills)
            { const bool keep_first = diag_first; diag_first = true; taps(tq[0], trow[2]); taps(tq[1], trow[3]); diag_first = keep_first; }
#else
            taps(tq[0], trow[2]); taps(tq[1], trow[3]);
#endif
            pp_slots2<NL, 3, L3>(Hw, own, tq[0], tq[1], y[2], y[3]);
            keep(y[NS - 1]);
            V_emit();
            diag_first = false;
            return;
        }
#endif
        taps(tp[0], trow[2]); taps(tp[1], trow[3]);
        if constexpr (NS == 5) taps(tq[0], trow[4]);
        pp_slots2<NL, 3, L3>(Hw, own, tp[0], tp[1], y[2], y[3]);
        if constexpr (NS == 5) pp_slot1<NL, L4>(Hw, own, tq[0], y[4]);
        keep(y[NS - 1]);                               // (computed by every lane beside the others, not as a chain of its own under the store's branch)
        V_emit();
    };
    // lean order (8 per lane; or 16 waves per CU, IQGPU_MID_WAVES=16): nothing is fetched more than one step ahead: the polyphase
    // first, then the pointwise phase, the windows, the half-band
    auto tile_lean = [&](const int64_t T, const bool PP, const bool next_pp) {
        if (PP) V_pp_lean();
        V_taprows(); FENCE();
        if (!NONCO) { nco_lookup(T); FENCE(); }
        VL_point();
        load_even(T + 1);
        FENCE();
        L_xr();
        V_centre();
        load_odd(T + 1);
        FENCE();
        V_hb(); FENCE();
        L_hb(next_pp); FENCE();
    };
    auto tile = [&](const int64_t T, const bool PP, const bool next_pp) {
        if (kLean) { tile_lean(T, PP, next_pp); return; }
        __builtin_amdgcn_s_setprio(1);
        VL_point();                                    // (before the polyphase, so that the mixed samples are not held across it)
        if (PP) { taps(tq[0], trow[2]); taps(tq[1], trow[3]); }
        FENCE();
        __builtin_amdgcn_s_setprio(0);
        if (PP) {
            pp_slots2<NL, 0, 1>(Hw, own, tp[0], tp[1], y[0], y[1]);
            pp_slots2<NL, 3, L3>(Hw, own, tq[0], tq[1], y[2], y[3]);
            keep(y[3]);                                // (computed by every lane beside the others, not as a chain of its own under the store's branch)
            V_emit();
        }
        V_taprows();
        load_even(T + 1);                              // (tile T_emit1 is readable too: the plan keeps one tile behind every run)
        FENCE();
        __builtin_amdgcn_s_setprio(1);                 // a wave about to feed the LDS pipe goes ahead of waves in their FMA runs
        L_xr();
        V_centre();
        load_odd(T + 1);
        FENCE();
        if (!NONCO) { nco_lookup(T + 1); FENCE(); }
        __builtin_amdgcn_s_setprio(0);
        V_hb(); FENCE();
        __builtin_amdgcn_s_setprio(1);
        L_hb(next_pp); FENCE();
        __builtin_amdgcn_s_setprio(0);
    };
    // claim(): next += 1 in the wave's descriptor (lane 0); claimed_end(): the `end` that add saw.  A thief only ever lowers `end`
    // to a tile behind the one being claimed (steal_run), so the tile in hand always stays this wave's.
    unsigned long long cv = 0;
    auto claim = [&]() { if (lane == 0) cv = __hip_atomic_fetch_add(desc, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto claimed_end = [&]() { return a.w_edge_ta + (int64_t)__builtin_amdgcn_readfirstlane((int)(cv >> 32)); };
    for (int64_t T = T_begin; T < T_emit0; ++T) tile(T, false, false);     // warm-up tiles
    if (STEAL) claim();
#ifdef IQGPU_DIAG_NOGATHER
    if (!kLean) { taps(tq[0], trow[2]); taps(tq[1], trow[3]); }
#endif
    tile(T_emit0, false, true);                                              // the first emitting tile: no polyphase in front of it yet
#ifdef IQGPU_DIAG_NOGATHER
    if (!kLean) diag_first = false;
#endif
    if (STEAL) T_emit1 = claimed_end();
    for (int64_t T = T_emit0 + 1; T < T_emit1; ++T) {                       // steady state
        if (STEAL) claim();
        tile(T, true, true);
        if (STEAL) T_emit1 = claimed_end();
    }
    // the last tile's polyphase
    if (kLean) V_pp_lean();
    else {
        taps(tq[0], trow[2]); taps(tq[1], trow[3]);
        pp_slots2<NL, 0, 1>(Hw, own, tp[0], tp[1], y[0], y[1]);
        pp_slots2<NL, 3, L3>(Hw, own, tq[0], tq[1], y[2], y[3]);
        keep(y[3]);
        V_emit();
    }
    if (AGC) {
        const double m = (double)wave_max_f(agc_m0);
        if (lane == 0 && m > 0.0) atomicMax(a.agc_peak2 + agc_c, (unsigned long long)__double_as_longlong(m));
    }
#undef FENCE
}

// A wave out of tiles looks for more: every lane loads one descriptor (64 of the launch's, spread evenly over its workgroups and so
// over the XCDs, a different set every round), the lane that saw the longest unclaimed run halves it -- compare-and-swap on the
// whole {end | next} word, retried on the value that comes back while the owner's claims keep moving `next` -- and the wave takes
// tiles [mid, end) with mid = next + ceil(rem / 2) >= next + 1: the owner is at most in tile next - 1 and about to claim `next`,
// which stays its own.  The taken run is published in the wave's own descriptor, so it can be split again.  Bounded: w_steal_rounds
// samples without a run of w_steal_min tiles anywhere and the wave retires.
__device__ __forceinline__ bool steal_run(const FrontArgs &a, const int64_t gw, const int lane, int64_t &t0, int64_t &t1)
{
    unsigned long long *const D = a.w_steal;
    const int n = (int)(a.w_n_edge + a.w_n_stream);
    // the victims of a round: one wave in each of `lanes` DIFFERENT workgroups, 37 workgroups apart -- workgroup b runs on XCD
    // b mod 8 and 37 = 5 (mod 8), so eight consecutive lanes look at all eight XCDs (round 4, first attempt: descriptors a fixed
    // stride of 4 .. 64 workgroups apart = always the thief's own XCD: runs only changed hands inside an XCD) -- and the wave
    // slot inside the workgroup varies with the lane too
    const int n_wg = n / kMidWaves;                                     // (n >= 64 * 12: the host leaves short launches static)
    const int wg = (int)(gw / kMidWaves), slot = (int)(gw % kMidWaves);
    for (int round = 0; round < a.w_steal_rounds; ++round) {
        const int vb = (wg + 1 + lane * 37 + round * (64 * 37 + 11)) % n_wg;
        const int v = vb * kMidWaves + (slot + lane * 5 + round) % kMidWaves;
        unsigned long long *const Dv = D + (size_t)v * (size_t)a.w_steal_stride;
        // (a read-modify-write, not a load: atomics execute at the memory side, while an sc1 load may be served by this XCD's L2
        //  with what the line held a launch ago -- the L2s of different XCDs are not coherent -- and an exhausted descriptor hides
        //  the run; adding a zero the compiler cannot see through: it turns an idempotent read-modify-write back into that load)
        unsigned long long cur = 0, zero = 0;
        asm volatile("" : "+v"(zero));
        if (lane < a.w_steal_lanes) cur = __hip_atomic_fetch_add(Dv, zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int32_t rem = (int32_t)(uint32_t)(cur >> 32) - (int32_t)(uint32_t)cur;
        uint32_t best = rem >= a.w_steal_min ? ((uint32_t)rem << 6) | (uint32_t)lane : 0u;
#pragma unroll
        for (int k = 32; k >= 1; k >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)best, k); best = o > best ? o : best; }
        best = (uint32_t)__builtin_amdgcn_readfirstlane((int)best);     // (every lane holds the maximum: tell the compiler it is uniform)
        if (best == 0u) continue;
        const int win = (int)(best & 63u);
        int32_t got_mid = -1, got_end = 0;
        if (lane == win) {
            for (int tries = 0; tries < 4; ++tries) {
                const int32_t e = (int32_t)(uint32_t)(cur >> 32), p = (int32_t)(uint32_t)cur;
                if (e - p < a.w_steal_min) break;
                const int32_t mid = p + ((e - p + 1) >> 1);
                const unsigned long long nd = ((unsigned long long)(uint32_t)mid << 32) | (unsigned long long)(uint32_t)p;
                if (__hip_atomic_compare_exchange_strong(Dv, &cur, nd, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    got_mid = mid; got_end = e;
                    break;
                }
            }
        }
        got_mid = __builtin_amdgcn_readfirstlane(__shfl(got_mid, win)); got_end = __builtin_amdgcn_readfirstlane(__shfl(got_end, win));
        if (got_mid < 0) continue;
        // (pinned to SGPRs: hipcc otherwise sinks their widening into both arms of the branch below, and the join of a divergent
        //  branch counts as divergent -- the whole tile loop then runs on vector compares and EXEC masks)
        asm volatile("" : "+s"(got_mid), "+s"(got_end));
        if (lane == 0) {
            __hip_atomic_store(D + (size_t)gw * (size_t)a.w_steal_stride, ((unsigned long long)(uint32_t)got_end << 32) | (unsigned long long)(uint32_t)got_mid,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd((unsigned long long *)((char *)a.sink + 32768 + 128) + 3, 1ull);      // runs taken, over the chain's life (iqgpu_chain_debug_read_scratch)
        }
        t0 = a.w_edge_ta + got_mid; t1 = a.w_edge_ta + got_end;
        return true;
    }
    return false;
}

// NONCO: the same shape without a shift (no mixer; the 2^-15 rides on the half-band taps, launch_front_mid scales hb0)
template <int NL, bool NONCO, int L3, int L4, bool AGC, bool STEAL, bool CF32OUT = false, int INF = IQGPU_FMT_CS16, int OUT8 = 0>
__global__ __launch_bounds__(kMidThreads) void k_front_mid(const FrontArgs a)
{
    typedef MidGeom<NL> G;
    if (a.run_if && *a.run_if == 0) return;         // (a fallback launch whose fused predecessor stood: never the case today, kept for symmetry)
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int kNco = NONCO ? 0 : kMidNcoLds;
    cf2   *s_nco = (cf2 *)smem, *s_nco_half = s_nco + 1024;
    float *s_arb = (float *)(smem + kNco);
    float *s_tap = (float *)(smem + kNco + kMidArbLds);
    char *slice = (char *)smem + kNco + kMidArbLds + kFTapLds + wave * G::XBYTES;
    char *arena = (char *)smem + kNco + kMidArbLds + kFTapLds + kMidWaves * G::XBYTES;
    if (!NONCO && ((unsigned)(size_t)(__attribute__((address_space(3))) const void *)s_nco & 8191u) != 0u) __builtin_trap();   // nco_phasor2 ORs the index into the base

    if (!NONCO) {
        const float sgn = a.nco_mode < 0 ? -1.0f : 1.0f;               // mix down: conj(phasor)
        const float scl = INF == IQGPU_FMT_CS16 ? 1.0f / 32768.0f : 1.0f;   // the cs16 normaliser, folded into the table (exact); every other INF: plain
        for (int i = tid; i < 1024; i += kMidThreads) {
            const cf2 v = a.nco_tab[i];
            s_nco[i] = cf2{v.x * scl, sgn * v.y * scl};
            s_nco_half[i] = cf2{v.x * (0.5f * scl), sgn * v.y * (0.5f * scl)};
        }
    }
    for (int i = tid; i < 256 * 14; i += kMidThreads) {                 // the edge waves' rows: arm a in row a ^ (a >> 5) of 56 B
        const int arm = i / 14, k = i % 14;
        s_arb[(arm ^ (arm >> 5)) * 14 + k] = a.arb_table[arm * 16 + k];
    }
    fill_tap_planes(s_tap, a.arb_table, tid, kMidThreads, a.tap_fold != 0);
    for (int i = lane; i < G::XBYTES / 16; i += 64) ((float4 *)slice)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < kMidEdgeLds / 16; i += kMidThreads) ((float4 *)arena)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned *const s_run_next = (unsigned *)(arena + kMidEdgeLds);   // fixed-length runs (w_run_stride): the workgroup's next run
    if (STEAL && tid == 0) *s_run_next = 0u;
    __syncthreads();

    const int64_t gw = (int64_t)blockIdx.x * kMidWaves + wave;
    if (gw == 0 && a.frames_in < (int64_t)a.hist_cap) {
        const int keep_n = a.hist_cap - (int)a.frames_in;
        for (int i = lane; i < keep_n; i += 64) a.hist_out[i] = a.hist_in[i + (int)a.frames_in];
    }
    constexpr bool EDGE = false;                      // (for CLOCK_END: the diagnostic -DIQGPU_CLOCKSTAMP build, tools/clock.py)
    CLOCK_BEGIN;
    int64_t t0 = 0, t1 = 0;
    bool have = false;
    if (gw < a.w_n_edge) {
        // edge work in tiles [0, w_edge_ta) and [w_edge_tb, w_total_tiles) of G::TILE frames, runs of w_edge_tpw of them (768-frame
        // tiles: two, from an even tile = three 512-frame tiles of run_tiles; 1024-frame tiles: one = two of them)
        int64_t e0, e1;
        if (gw < a.w_n_edge1) { e0 = gw * a.w_edge_tpw; e1 = e0 + a.w_edge_tpw; if (e1 > a.w_edge_ta) e1 = a.w_edge_ta; }
        else { e0 = a.w_edge_tb + (gw - a.w_n_edge1) * a.w_edge_tpw; e1 = e0 + a.w_edge_tpw; if (e1 > a.w_total_tiles) e1 = a.w_total_tiles; }
        if (gw >= kMidEdgeMax) __builtin_trap();     // (the host keeps launches with more edge runs on k_front_s1)
        WaveLds w;
        w.XE = arena + (int)gw * kWaveLds; w.XO = w.XE + kXRows * kRowB; w.HB = w.XO + kHBOff * kRowB;
        w.nco = s_nco; w.arb = s_arb;
        w.arb_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)s_arb;
        const int64_t o0 = e0 * G::TILE / 512, o1 = (e1 * G::TILE + 511) / 512;
        // (8-bit frames in: the run-time-switched tile routine -- the table above is the plain one then)
        if constexpr (INF == IQGPU_FMT_CS16) run_tiles<4, true, true, false, AGC, NONCO>(a, w, lane, o0 - 1, o0, o1, 0);
        else if constexpr (INF == kMidN16) run_tiles<4, true, false, false, AGC, false>(a, w, lane, o0 - 1, o0, o1, 0);
        else run_tiles<2, true, false, false, AGC, false>(a, w, lane, o0 - 1, o0, o1, 0);
    } else {
        const int64_t r = gw - a.w_n_edge;
        if (r >= a.w_n_stream || (a.w_run_stride > 0 && r >= a.w_run_stride)) return;
        t0 = w_run_start_weighted(a, r); t1 = w_run_start_weighted(a, r + 1);
        have = !(STEAL && a.w_run_stride > 0);       // (fixed-length runs: every run comes from the workgroup's queue, below)
        if (STEAL && have && lane == 0)              // the static run, open to thieves from here on
            __hip_atomic_store(a.w_steal + (size_t)gw * (size_t)a.w_steal_stride, ((unsigned long long)(uint32_t)(t1 - a.w_edge_ta) << 32) | (unsigned long long)(uint32_t)(t0 - a.w_edge_ta),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the wave's static run, then whatever it can take from the waves that are behind (an edge wave starts here); one copy of the
    // tile routine, the claim / steal logic all outside it
    MidLds w;
    w.XE = slice; w.nco = s_nco;
    w.tap_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)s_tap;
    unsigned n_stolen = 0;
    if constexpr (!STEAL) {
        if (have) run_mid<NL, NONCO, L3, L4, AGC, false, CF32OUT, INF, OUT8>(a, w, lane, t0 - a.w_warm_tiles, t0, t1, nullptr);
    } else {
        for (;;) {
            // (the lane index made opaque per run: nothing a run derives from it is then hoisted out of this loop and held --
            //  spilled -- across the tile loop of every run)
            int ln = (int)__lane_id();
            asm volatile("" : "+v"(ln));
            if (have) run_mid<NL, NONCO, L3, L4, AGC, true, CF32OUT, INF, OUT8>(a, w, ln, t0 - a.w_warm_tiles, t0, t1, a.w_steal + (size_t)gw * (size_t)a.w_steal_stride);
            if (a.w_run_stride > 0) {
                // fixed-length runs: the workgroup's next one (an LDS add: no memory traffic, nothing to reset between launches)
                unsigned k = 0u;
                if (ln == 0) k = __hip_atomic_fetch_add(s_run_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int64_t r_cur = (int64_t)blockIdx.x + (int64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)k) * (int64_t)gridDim.x;
                if (r_cur >= a.w_n_stream) break;
                t0 = w_run_start(a, r_cur); t1 = w_run_start(a, r_cur + 1);
                if (ln == 0)
                    __hip_atomic_store(a.w_steal + (size_t)gw * (size_t)a.w_steal_stride, ((unsigned long long)(uint32_t)(t1 - a.w_edge_ta) << 32) | (unsigned long long)(uint32_t)(t0 - a.w_edge_ta),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                have = true;
                continue;
            }
            if (!steal_run(a, gw, ln, t0, t1)) break;
            have = true; n_stolen += 1;
        }
    }
    CLOCK_END(a.sink);
#ifdef IQGPU_CLOCKSTAMP
    if (lane == 0 && gw < 4032) ((unsigned *)((char *)a.sink + 49408))[gw] = n_stolen;
#endif
    (void)EDGE; (void)n_stolen;
}

// Placement of the arms in the tap planes (front_fat_common.hpp).  A slot's 8-byte tap reads are served a half-wave at a time, 32
// lanes over 32 bank pairs, and take as many cycles as the fullest bank pair has DISTINCT entries.  Which entries meet is a
// function of the step alone: lane l's slot j reads entry slot(arm) + 257 d of the lane's phase.  Model both placements over a
// spread of tile phases and keep the linear one unless the fold wins by more than its two extra VALU instructions per slot are
// worth (8 reads per slot: 1.5 cycles per read).  NRSC-5's step at 6 per lane: 6.0 against 5.8 -> linear; the same step at 8 per
// lane walks the arms in strides of 16: 15.2 against 5.4 -> folded; s = 1.625 at 6 per lane 15.5 against 4.0.
static double tap_gather_cycles(const uint32_t step, const int nl, const bool fold)
{
    const int ns = nl == 6 ? 4 : 5;
    int lo[5];
    for (int j = 0; j < 5; ++j) lo[j] = (int)(((uint64_t)step * (uint64_t)j) >> 24);
    uint64_t total = 0, reads = 0;
    for (int t = 0; t < 48; ++t) {
        const uint64_t base = (uint64_t)64 * nl * t * 7;            // first half-band sample of the tile
        uint32_t P[64];
        for (int l = 0; l < 64; ++l) {
            const uint64_t g = (base + (uint64_t)nl * l) << 24;
            const uint64_t k = (g + step - 1) / step;               // first output at or behind the lane's first sample
            P[l] = (uint32_t)(k * step - g);
        }
        for (int j = 0; j < ns; ++j) {
            for (int h = 0; h < 2; ++h) {
                uint32_t ent[32]; int n_ent = 0, cnt[32] = {0};
                for (int l = 32 * h; l < 32 * h + 32; ++l) {
                    const uint32_t Pj = P[l] + (uint32_t)j * step;
                    uint32_t x = Pj >> 16;
                    if (fold) x ^= (Pj >> 21) & 7u;
                    const uint32_t e = x + (Pj >> 24) - 257u * (uint32_t)lo[j];
                    bool seen = false;
                    for (int i = 0; i < n_ent; ++i) if (ent[i] == e) { seen = true; break; }
                    if (!seen) { ent[n_ent++] = e; cnt[e & 31u] += 1; }
                }
                int mx = 0;
                for (int b = 0; b < 32; ++b) if (cnt[b] > mx) mx = cnt[b];
                total += (uint64_t)mx;
            }
            reads += 1;
        }
    }
    return (double)total / (double)reads;
}
int front_tap_fold(const uint32_t step, const int nl)
{
    if (nl != 6 && nl != 8) return 0;
    return tap_gather_cycles(step, nl, true) + 1.5 < tap_gather_cycles(step, nl, false) ? 1 : 0;
}

// step classes: lo_3 = floor(3 s), lo_4 = floor(4 s) with s = step / 2^24.  Six per lane (four slots) takes 1.5 <= s < 2, eight per
// lane (five slots) 1.6 <= s < 2
static bool mid_class(uint32_t step, int nl, int *l3, int *l4)
{
    const uint64_t one = (uint64_t)1 << 24;
    if ((uint64_t)step >= 2 * one) return false;
    if (nl == 6 ? (uint64_t)step * 2 < 3 * one : (uint64_t)step * 5 < 8 * one) return false;
    *l3 = (int)(((uint64_t)step * 3) >> 24); *l4 = (int)(((uint64_t)step * 4) >> 24);
    if (nl == 6) { *l4 = 0; return *l3 == 4 || *l3 == 5; }
    return (*l3 == 4 && *l4 == 6) || (*l3 == 5 && *l4 == 6) || (*l3 == 5 && *l4 == 7);
}

// outputs per lane of the k_front_mid instantiation for these arguments: 6 (8 on request); 0 = not this kernel's shape
int front_mid_nl(const FrontArgs &a)
{
    // (cf32 out: a user filter behind the resampler takes the samples -- the -usb / -lsb presets; six outputs per lane, no fused AGC)
    const bool cf32_out = a.out_fmt == IQGPU_FMT_CF32 && !a.agc_fused;
    // (late round 5: 8-bit frames on either side -- cu8 / cs8 in, cu8 / cs8 out, any mix with cs16 and cf32 -- six outputs per lane)
    const bool in8 = a.in_fmt == IQGPU_FMT_CU8 || a.in_fmt == IQGPU_FMT_CS8, out8 = a.out_fmt == IQGPU_FMT_CU8 || a.out_fmt == IQGPU_FMT_CS8;
    // ... and 16-bit frames with a gain or of the sc16q11 scale (normalised at the unpack: front_mid_inf)
    const bool n16 = (a.in_fmt == IQGPU_FMT_CS16 && a.gain != 1.0f) || a.in_fmt == IQGPU_FMT_SC16Q11;
    const bool any8 = (in8 || out8 || n16) && !(a.dbg & kDbgNoMid8bit);
    if (!(a.S == 1 && ((a.in_fmt == IQGPU_FMT_CS16 && !n16) || ((in8 || n16) && any8)) && (a.out_fmt == IQGPU_FMT_CS16 || cf32_out || (out8 && any8)) &&
          (a.gain == 1.0f || any8) && a.gain > 0.0f && a.gain < 1e30f &&
          !a.iq_enable && !a.dc_enable && a.pnco_mode == 0 && !(a.dbg & (kDbgNoFast | kDbgNoFat)))) return 0;
    int l3, l4;
    for (int nl : {8, 6}) {
        // 8 per lane is an experiment (IQGPU_MID8=1): in 168 VGPRs it has no room to fetch a phase ahead, and without that it runs
        // 0.440 ms against 0.384 for 6 per lane on the NRSC-5 chain; with the fused AGC it does not fit at all
        if (nl == 8 && (!(a.dbg & kDbgMid8) || a.agc_fused || cf32_out || any8)) continue;
        if (!mid_class(a.step, nl, &l3, &l4)) continue;
        if (a.agc_fused && !(a.agc_shift == 1 && a.agc_chunk_frames >= 128 * nl)) continue;
        return nl;
    }
    return 0;
}
int front_mid_tile(int nl) { return 128 * nl; }

hipError_t launch_front_mid(const FrontArgs &a_in, hipStream_t s)
{
    const bool nonco = a_in.nco_mode == 0;
    FrontArgs a = a_in;
    const bool n16 = (a.in_fmt == IQGPU_FMT_CS16 && a.gain != 1.0f) || a.in_fmt == IQGPU_FMT_SC16Q11;
    const bool in8 = a.in_fmt == IQGPU_FMT_CU8 || a.in_fmt == IQGPU_FMT_CS8 || n16, out8 = a.out_fmt == IQGPU_FMT_CU8 || a.out_fmt == IQGPU_FMT_CS8;
    // (in8: every input that is normalised at the unpack -- the 8-bit formats and kMidN16)
    if (nonco && !in8) for (float &h : a.hb0) h *= 1.0f / 32768.0f;  // the cs16 normaliser rides on the half-band taps (exact: a power of two)
    const int nl = front_mid_nl(a);
    int l3 = 0, l4 = 0;
    if (nl == 0 || !mid_class(a.step, nl, &l3, &l4)) return hipErrorInvalidValue;
    const size_t lds = mid_lds_bytes(nl, nonco);
    // (fixed-length runs dealt out inside a workgroup need the multi-run instantiation, which exists for six outputs per lane only:
    //  any other shape gets one static run per wave, however the caller filled w_run_stride)
    if (nl != 6 || a.w_steal == nullptr || a.out_fmt == IQGPU_FMT_CF32 || in8 || out8) { a.w_run_stride = 0; if (a.out_fmt == IQGPU_FMT_CF32 || in8 || out8) a.w_steal = nullptr; }
    const int64_t n_items = a.w_n_edge + (a.w_run_stride > 0 && a.w_run_stride < a.w_n_stream ? a.w_run_stride : a.w_n_stream);
    const unsigned grid = (unsigned)((n_items + kMidWaves - 1) / kMidWaves);
    if (grid == 0) return hipSuccess;
#define IQGPU_LAUNCH_MID1(NL, NONCO, L3, L4, AGC, STEAL, CF)                                                        \
    do {                                                                                                              \
        static LdsAttrCache cache;                /* per instantiation */                                          \
        { const hipError_t e = cache.ensure((const void *)k_front_mid<NL, NONCO, L3, L4, AGC, STEAL, CF>, lds); if (e != hipSuccess) return e; } \
        hipLaunchKernelGGL((k_front_mid<NL, NONCO, L3, L4, AGC, STEAL, CF>), dim3(grid), dim3(kMidThreads), lds, s, a); \
    } while (0)
    /* run stealing: six outputs per lane only, and only when the host provides descriptors and asks for it */
    const bool steal = a.w_steal != nullptr && (a.w_steal_rounds > 0 || a.w_run_stride > 0);
    /* cf32 out (a user filter behind the resampler): six per lane, static runs, no fused AGC */
    const bool cf = a.out_fmt == IQGPU_FMT_CF32;
    if (cf && (nl != 6 || a.agc_fused)) return hipErrorInvalidValue;
    if (cf) { a.w_steal = nullptr; a.w_run_stride = 0; }
#define IQGPU_LAUNCH_MID(NL, NONCO, L3, L4, AGC)                                                                    \
    do {                                                                                                              \
        if (NL == 6 && cf) IQGPU_LAUNCH_MID1(NL, NONCO, L3, L4, false, false, (NL == 6));                           \
        else if (NL == 6 && steal) IQGPU_LAUNCH_MID1(NL, NONCO, L3, L4, AGC, (NL == 6), false);                     \
        else IQGPU_LAUNCH_MID1(NL, NONCO, L3, L4, AGC, false, false);                                               \
    } while (0)
#define IQGPU_LAUNCH_MID2(NL, L3, L4)                                                                               \
    do {                                                                                                              \
        if (nonco && a.agc_fused) IQGPU_LAUNCH_MID(NL, true, L3, L4, (NL == 6));                                    \
        else if (nonco) IQGPU_LAUNCH_MID(NL, true, L3, L4, false);                                                  \
        else if (a.agc_fused) IQGPU_LAUNCH_MID(NL, false, L3, L4, (NL == 6));                                       \
        else IQGPU_LAUNCH_MID(NL, false, L3, L4, false);                                                            \
    } while (0)
    if (in8 || out8) {
        // 8-bit frames: six per lane, static runs; INF / OUT8 as the formats say, CF32OUT for a filter behind
        if (nl != 6 || (cf && a.agc_fused)) return hipErrorInvalidValue;
#define IQGPU_LAUNCH_MID8C(NONCO, L3, AGC, CF, INF, O8)                                                             \
        do {                                                                                                          \
            static LdsAttrCache cache;                                                                                \
            { const hipError_t e = cache.ensure((const void *)k_front_mid<6, NONCO, L3, 0, AGC, false, CF, INF, O8>, lds); if (e != hipSuccess) return e; } \
            hipLaunchKernelGGL((k_front_mid<6, NONCO, L3, 0, AGC, false, CF, INF, O8>), dim3(grid), dim3(kMidThreads), lds, s, a); \
        } while (0)
#define IQGPU_LAUNCH_MID8B(NONCO, L3, INF)                                                                          \
        do {                                                                                                          \
            if (cf) IQGPU_LAUNCH_MID8C(NONCO, L3, false, true, INF, 0);                                              \
            else if (a.out_fmt == IQGPU_FMT_CU8 && a.agc_fused) IQGPU_LAUNCH_MID8C(NONCO, L3, true, false, INF, 1);  \
            else if (a.out_fmt == IQGPU_FMT_CU8) IQGPU_LAUNCH_MID8C(NONCO, L3, false, false, INF, 1);                \
            else if (a.out_fmt == IQGPU_FMT_CS8 && a.agc_fused) IQGPU_LAUNCH_MID8C(NONCO, L3, true, false, INF, 2);  \
            else if (a.out_fmt == IQGPU_FMT_CS8) IQGPU_LAUNCH_MID8C(NONCO, L3, false, false, INF, 2);                \
            else if (INF != IQGPU_FMT_CS16 && a.agc_fused) IQGPU_LAUNCH_MID8C(NONCO, L3, true, false, INF, 0);       \
            else if (INF != IQGPU_FMT_CS16) IQGPU_LAUNCH_MID8C(NONCO, L3, false, false, INF, 0);                     \
            else return hipErrorInvalidValue;                                                                         \
        } while (0)
#define IQGPU_LAUNCH_MID8A(NONCO, L3)                                                                               \
        do {                                                                                                          \
            if (a.in_fmt == IQGPU_FMT_CU8) IQGPU_LAUNCH_MID8B(NONCO, L3, IQGPU_FMT_CU8);                             \
            else if (a.in_fmt == IQGPU_FMT_CS8) IQGPU_LAUNCH_MID8B(NONCO, L3, IQGPU_FMT_CS8);                        \
            else if (n16) IQGPU_LAUNCH_MID8B(NONCO, L3, kMidN16);                                                    \
            else IQGPU_LAUNCH_MID8B(NONCO, L3, IQGPU_FMT_CS16);                                                      \
        } while (0)
        if (l3 == 4) { if (nonco) IQGPU_LAUNCH_MID8A(true, 4); else IQGPU_LAUNCH_MID8A(false, 4); }
        else         { if (nonco) IQGPU_LAUNCH_MID8A(true, 5); else IQGPU_LAUNCH_MID8A(false, 5); }
#undef IQGPU_LAUNCH_MID8A
#undef IQGPU_LAUNCH_MID8B
#undef IQGPU_LAUNCH_MID8C
        return hipGetLastError();
    }
    if (nl == 6 && l3 == 4) IQGPU_LAUNCH_MID2(6, 4, 0);
    else if (nl == 6) IQGPU_LAUNCH_MID2(6, 5, 0);
    else if (l3 == 4) IQGPU_LAUNCH_MID2(8, 4, 6);
    else if (l4 == 6) IQGPU_LAUNCH_MID2(8, 5, 6);
    else IQGPU_LAUNCH_MID2(8, 5, 7);
#undef IQGPU_LAUNCH_MID2
#undef IQGPU_LAUNCH_MID
#undef IQGPU_LAUNCH_MID1
    return hipGetLastError();
}

} // namespace iqgpu
