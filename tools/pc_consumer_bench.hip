// pc_consumer_bench.hip -- round 6, VERDICT r5 item 2(b): the CONSUMER role of a producer / consumer split of k_front_mid, by itself.
//
// The proposal: half-band waves feed the HB stream the headline kernel already writes to LDS; polyphase waves keep the taps of their
// outputs in registers k_front_p0-style and re-read a slot only when its arm moves on; 16 waves per CU at <= 128 VGPRs each.  Before
// anybody ports that, this measures what the consumer role costs when it is fed for free: a wave owns an HB ring in LDS (pre-filled,
// 512 samples + a mirrored margin so that a lane's window is contiguous), and a step is 240 consecutive outputs -- 60 lanes x 4; for
// the NRSC-5 step 240 x 1.6125 = 387 samples, so a lane-slot's arm moves by -0.19 arms per step -- through the product's own slot
// routines (front_fat_common.hpp: pp_slots2 with written-out op_sel, the shifted zero-padded tap rows, tap_row / fill_tap_planes):
//   20 ds_read_b64 of window (immediate offsets from one address), the (position, arm) keys of the four slots compared with the ones
//   their registers were loaded for and the rows re-read under an EXEC mask where they moved, 64 v_pk_fma_f32, pack to cs16, one
//   16-byte store per lane.
// Reported: time per step and CU at WAVES waves per CU, and -- under rocprofv3 --pmc -- the role's LDS cycles and VALU instructions
// per step, to be put into the sum model of profiles/r05_headline.md beside the producer role's (profiles/r06_headline.md).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I iq_tool_amd/csrc tools/pc_consumer_bench.hip -o tools/pc_consumer_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "front_fat_common.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using namespace iqgpu;

constexpr int kRing = 512, kMargin = 32;                      // HB samples per wave's ring (+ the first kMargin mirrored behind it)
constexpr int kRingB = (kRing + kMargin) * 8;
constexpr int kStepOut = 240, kLanes = 60, kNS = 4, kNL = 7;
constexpr uint32_t kStep = 27053208u;                         // NRSC-5 (SPEC B.6): 1.6125 - 3.1e-6 samples per output
// slots of a lane's four outputs: LO_j = floor(j s) for s = 1.6125
constexpr int kLO1 = 1, kLO2 = 3, kLO3 = 4;

template <int WAVES, bool FOLD>
__global__ __launch_bounds__(WAVES * 64) void k_consumer(const float *arb_table, uint32_t *out, unsigned long long *cycles, int reps, int full_gather)
{
    extern __shared__ __align__(16) unsigned char smem[];
    typedef __attribute__((address_space(3))) const v2f lds_v2f;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *s_tap = (float *)smem;
    char *rings = (char *)smem + kFTapLds;
    fill_tap_planes(s_tap, arb_table, tid, WAVES * 64, FOLD);
    // the rings: any finite samples will do (the arithmetic does not depend on them)
    for (int i = tid; i < WAVES * (kRing + kMargin); i += WAVES * 64) ((v2f *)rings)[i] = v2f{0.001f * (float)(i & 255), -0.002f * (float)(i & 127)};
    __syncthreads();
    const unsigned tap_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)s_tap;
    const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)(rings + wave * kRingB);
    const bool active = lane < kLanes;

    // phase of the lane's first output, in ring samples << 24 (the position's integer part is taken modulo the ring)
    uint64_t P = (uint64_t)(4 * lane) * kStep + ((uint64_t)14 << 24);
    const uint64_t adv = (uint64_t)kStepOut * kStep;
    v2f t[kNS][8];
    uint32_t held[kNS];
#pragma unroll
    for (int j = 0; j < kNS; ++j) {
        held[j] = 0xffffffffu;
#pragma unroll
        for (int i = 0; i < 8; ++i) t[j][i] = v2f{0.f, 0.f};
    }
    constexpr int LO[kNS] = {0, kLO1, kLO2, kLO3};
    uint32_t *dst = out + ((size_t)blockIdx.x * WAVES + wave) * 256 + 4 * lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < reps; ++it) {
        const uint32_t F = (uint32_t)P & 0xffffffu;
        const uint32_t p0 = (uint32_t)(P >> 24);                        // position of the lane's first output
        // ---- the window: samples p0 - 13 .. p0 + 6, contiguous in the ring thanks to the mirrored margin
        const unsigned wa = ring_lds + (((p0 - 13u) & (kRing - 1)) << 3);
        v2f Hw[14], own[kNL];
        Hw[0] = v2f{0.f, 0.f};
        if (active) {
#pragma unroll
            for (int i = 1; i < 14; ++i) Hw[i] = *(lds_v2f *)(size_t)(wa + 8u * (i - 1));
#pragma unroll
            for (int m = 0; m < kNL; ++m) own[m] = *(lds_v2f *)(size_t)(wa + 8u * (13 + m));
        } else {
#pragma unroll
            for (int i = 1; i < 14; ++i) Hw[i] = v2f{0.f, 0.f};
#pragma unroll
            for (int m = 0; m < kNL; ++m) own[m] = v2f{0.f, 0.f};
        }
        // ---- tap rows: re-read under a mask where the slot's (position, arm) key moved (full_gather: always -- the general step)
#pragma unroll
        for (int j = 0; j < kNS; ++j) {
            const uint32_t pj = F + (uint32_t)j * kStep;
            const uint32_t key = pj >> 16;
            if (active && (full_gather || key != held[j])) {
                const unsigned row = tap_row<FOLD>(tap_lds, pj, LO[j]);
#pragma unroll
                for (int i = 0; i < 8; ++i) t[j][i] = *(lds_v2f *)(size_t)(row + tap_pair_off(i));
                held[j] = key;
            }
        }
        // ---- four outputs
        v2f y[kNS];
        pp_slots2<kNL, 0, kLO1, true>(Hw, own, t[0], t[1], y[0], y[1]);
        pp_slots2<kNL, kLO2, kLO3, true>(Hw, own, t[2], t[3], y[2], y[3]);
        // ---- pack + store
        if (active) {
            typedef uint32_t w4v __attribute__((ext_vector_type(4), aligned(16)));
            *(w4v *)dst = w4v{pack_cs16(cf2{y[0].x, y[0].y}), pack_cs16(cf2{y[1].x, y[1].y}), pack_cs16(cf2{y[2].x, y[2].y}), pack_cs16(cf2{y[3].x, y[3].y})};
        }
        P += adv;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int WAVES>
static void run(const float *d_tab, uint32_t *d_out, unsigned long long *d_cyc, int reps, int full_gather, int n_cu)
{
    const size_t lds = (size_t)kFTapLds + (size_t)WAVES * kRingB;
    CK(hipFuncSetAttribute((const void *)k_consumer<WAVES, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0, best = 1e9f;
    for (int w = 0; w < 4; ++w) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_consumer<WAVES, true>), dim3(n_cu), dim3(WAVES * 64), lds, 0, d_tab, d_out, d_cyc, reps, full_gather);
        CK(hipGetLastError());
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        if (w > 0 && ms < best) best = ms;
    }
    unsigned long long cyc; CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
    const double steps_cu = (double)WAVES * reps;                    // steps per CU
    printf("  %2d waves per CU, %s: %8.3f ms for %d steps per wave -> %7.1f ns = %7.1f CU-cycles per step of 240 outputs (clock %.2f GHz); "
           "a 2^28-frame call (349 525 tiles = steps) on %d CUs: %.3f ms of consumer role alone\n",
           WAVES, full_gather ? "full gather every step" : "taps held, masked re-reads", best, reps, best * 1e6 / steps_cu, (double)cyc / steps_cu,
           (double)cyc / (best * 1e6), n_cu, best * 1e6 / steps_cu * (349525.0 / n_cu) * 1e-6);
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 4000;
    int n_cu = 256;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); n_cu = prop.multiProcessorCount;
    // a plausible polyphase table: 256 arms x 16 floats (14 taps), values irrelevant to the timing
    std::vector<float> h_tab(256 * 16, 0.0f);
    for (int a = 0; a < 256; ++a) for (int k = 0; k < 14; ++k) h_tab[(size_t)a * 16 + k] = 0.07f * (float)((a * 31 + k * 7) % 13 - 6);
    float *d_tab; uint32_t *d_out; unsigned long long *d_cyc;
    CK(hipMalloc(&d_tab, h_tab.size() * 4)); CK(hipMalloc(&d_out, (size_t)n_cu * 16 * 256 * 4)); CK(hipMalloc(&d_cyc, (size_t)n_cu * 8));
    CK(hipMemcpy(d_tab, h_tab.data(), h_tab.size() * 4, hipMemcpyHostToDevice));
    printf("== consumer role of a producer / consumer split of k_front_mid: 240 outputs per step from an HB ring in LDS, %d CUs\n", n_cu);
    for (int fg = 0; fg < 2; ++fg) {
        run<16>(d_tab, d_out, d_cyc, reps, fg, n_cu);
        run<12>(d_tab, d_out, d_cyc, reps, fg, n_cu);
        run<8>(d_tab, d_out, d_cyc, reps, fg, n_cu);
        run<4>(d_tab, d_out, d_cyc, reps, fg, n_cu);
    }
    return 0;
}
