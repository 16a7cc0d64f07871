// lds_mask_bench.hip -- what does an LDS (or a vector-memory) gather cost when only a FEW lanes of the wave are active?
// Round 5, headline kernel: the polyphase tap gather is half of k_front_mid's LDS time.  An output-major polyphase stage whose
// step covers a whole number of arm periods (NRSC-5: 240 outputs = 3 x 80) keeps every slot's 14 taps in registers from step to step
// and re-reads a slot only when its arm moves on -- about one slot-lane in eight per step.  Whether that pays depends on what the
// hardware charges for a ds_read_b64 / ds_read_b128 under a sparse EXEC mask, which this measures: 12 waves per CU on every CU,
// each issuing `reads` gathers per iteration with k of 64 lanes active (random lanes, a fresh mask per instruction), rows random.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_mask_bench.hip -o tools/lds_mask_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kWaves = 12, kReads = 28, kMasks = 64;

// MODE 0: ds_read_b64, 1: ds_read_b128, 2: global_load_dwordx2 from a 16 KB table, 3: global_load_dwordx4
template <int MODE>
__global__ __launch_bounds__(kWaves * 64) void k(const unsigned long long *masks, const unsigned *rows, const float *table,
                                                 unsigned long long *cycles, float *sink, int reps)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) ((float *)smem)[i] = (float)i;
    __syncthreads();
    unsigned row[kReads];
    unsigned long long msk[kReads];
#pragma unroll
    for (int r = 0; r < kReads; ++r) {
        row[r] = rows[(r * 64 + lane + 64 * wave) & 4095] & 255u;
        msk[r] = masks[(r * 3 + wave) & (kMasks - 1)];                 // wave-uniform: lives in SGPRs
    }
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < reps; ++it) {
        v4f t[kReads];
#pragma unroll
        for (int r = 0; r < kReads; ++r) {                              // all gathers of the iteration in flight, ONE wait behind them
            t[r] = v4f{0.f, 0.f, 0.f, 0.f};
            if ((msk[r] >> lane) & 1ull) {
                if (MODE == 0) { v2f q; unsigned ad = row[r] * 64u + 8u * (r & 7); asm volatile("ds_read_b64 %0, %1" : "=v"(q) : "v"(ad)); t[r].x = q.x; t[r].y = q.y; }
                if (MODE == 1) { unsigned ad = row[r] * 64u + 16u * (r & 3); asm volatile("ds_read_b128 %0, %1" : "=v"(t[r]) : "v"(ad)); }
                if (MODE == 2) { const v2f q = *(const v2f *)(table + row[r] * 16 + 2 * (r & 7)); t[r].x = q.x; t[r].y = q.y; }
                if (MODE == 3) { t[r] = *(const v4f *)(table + row[r] * 16 + 4 * (r & 3)); }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < kReads; ++r) { asm volatile("" : "+v"(t[r])); acc += t[r].x; row[r] = (row[r] + 97u) & 255u; }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc == 1.2345f) sink[threadIdx.x] = acc;
}

int main()
{
    srand(7);
    std::vector<unsigned> h_rows(4096);
    for (auto &r : h_rows) r = (unsigned)rand();
    std::vector<float> h_tab(256 * 16, 1.0f);
    unsigned long long *d_masks, *d_cyc; unsigned *d_rows; float *d_tab, *d_sink;
    CK(hipMalloc(&d_masks, kMasks * 8)); CK(hipMalloc(&d_cyc, 256 * 8)); CK(hipMalloc(&d_rows, 4096 * 4));
    CK(hipMalloc(&d_tab, 256 * 16 * 4)); CK(hipMalloc(&d_sink, 4096));
    CK(hipMemcpy(d_rows, h_rows.data(), 4096 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tab, h_tab.data(), 256 * 16 * 4, hipMemcpyHostToDevice));
    const int reps = 400;
    const char *names[] = {"ds_read_b64", "ds_read_b128", "global_load_dwordx2 (16 KB table)", "global_load_dwordx4 (16 KB table)"};
    struct Pat { int k; int contiguous; } pats[] = {{64, 0}, {32, 0}, {16, 0}, {8, 0}, {8, 1}, {4, 0}, {2, 0}, {1, 0}, {0, 0}};
    for (int mode = 0; mode < 4; ++mode) {
        printf("== %s: %d waves per CU x 256 CUs, %d gathers per iteration, %d iterations\n", names[mode], kWaves, kReads, reps);
        for (const Pat &p : pats) {
            std::vector<unsigned long long> h_masks(kMasks, 0ull);
            for (auto &m : h_masks) {
                if (p.contiguous) { const int s = rand() % (65 - p.k); for (int i = 0; i < p.k; ++i) m |= 1ull << (s + i); }
                else { int n = 0; while (n < p.k) { const int b = rand() & 63; if (!((m >> b) & 1ull)) { m |= 1ull << b; ++n; } } }
            }
            CK(hipMemcpy(d_masks, h_masks.data(), kMasks * 8, hipMemcpyHostToDevice));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float ms = 0;
            for (int w = 0; w < 3; ++w) {
                CK(hipEventRecord(e0));
                switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(kWaves * 64), 16384, 0, d_masks, d_rows, d_tab, d_cyc, d_sink, reps); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(kWaves * 64), 16384, 0, d_masks, d_rows, d_tab, d_cyc, d_sink, reps); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(kWaves * 64), 16384, 0, d_masks, d_rows, d_tab, d_cyc, d_sink, reps); break;
                default: hipLaunchKernelGGL(k<3>, dim3(256), dim3(kWaves * 64), 16384, 0, d_masks, d_rows, d_tab, d_cyc, d_sink, reps); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            unsigned long long cyc; CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
            const double instr = (double)kWaves * reps * kReads;                      // gathers per CU
            // s_memtime counts shader clocks on this chip (tools/lds_gather_bench.hip): clock = ticks / time
            printf("  %2d lanes active%s: %8.3f ms, %6.2f ns = %6.2f cycles per gather per CU (clock %.2f GHz)\n",
                   p.k, p.contiguous ? " (contiguous)" : "", ms, ms * 1e6 / instr, (double)cyc / instr, (double)cyc / (ms * 1e6));
        }
    }
    return 0;
}
