// power_probe.hip -- what sets the shader clock the full chip holds?  (k_front_mid runs at 1.94 GHz on 256 CUs, 2.38 GHz
// on 128.)  Synthetic instruction mixes, 12 waves per workgroup, one workgroup per CU, launched back to back for a few seconds; every
// wave stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its loop: clock = d(memtime) / d(memrealtime) x 100 MHz.
//   mix 0  v_pk_fma_f32 only             mix 1  v_fma_f32 only (same flops per iteration)
//   mix 2  ds_read_b128, conflict-free   mix 3  ds_read_b64, lane-random addresses (the tap gather)
//   mix 4  global 16-byte loads, streaming (HBM)
//   mix 5  integer VALU only (v_add / v_xor)
//   mix 6  s_nop only (waves resident, nothing executed)
//   mix 7  k_front_mid's counts per tile: 212 pk_fma + 128 other VALU + 24 b128 + 32 b64 gathers + 12 b128 writes + 6 global loads + a 12-byte store
//   mix 8  mix 7 without the LDS part     mix 9  mix 7 without the FMAs     mix 10 mix 7 without the global loads
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/power_probe tools/power_probe.hip ; run: /tmp/power_probe [seconds per point] [first mix]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

constexpr int kWaves = 12;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void pkfma(v2f (&acc)[16], v2f a, v2f b)
{
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[i & 15]) : "v"(a), "v"(b));
}
template <int N> __device__ __forceinline__ void fma1(float (&acc)[32], float a, float b)
{
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i & 31]) : "v"(a), "v"(b));
}
template <int N> __device__ __forceinline__ void ialu(uint32_t (&r)[16], uint32_t k)
{
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (i & 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i & 15]) : "v"(k));
        else asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i & 15]) : "v"(k));
    }
}
template <int N> __device__ __forceinline__ void lds128(uint32_t addr, v4f (&d)[4])
{
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if ((i & 3) == 0) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(d[0]) : "v"(addr));
        if ((i & 3) == 1) asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(d[1]) : "v"(addr));
        if ((i & 3) == 2) asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(d[2]) : "v"(addr));
        if ((i & 3) == 3) { asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(d[3]) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
template <int N> __device__ __forceinline__ void lds64(uint32_t addr, v2f (&d)[8])
{
#pragma unroll
    for (int i = 0; i < N; ++i) {
        switch (i & 7) {
        case 0: asm volatile("ds_read_b64 %0, %1 offset:0" : "=v"(d[0]) : "v"(addr)); break;
        case 1: asm volatile("ds_read_b64 %0, %1 offset:4112" : "=v"(d[1]) : "v"(addr)); break;
        case 2: asm volatile("ds_read_b64 %0, %1 offset:8224" : "=v"(d[2]) : "v"(addr)); break;
        case 3: asm volatile("ds_read_b64 %0, %1 offset:12336" : "=v"(d[3]) : "v"(addr)); break;
        case 4: asm volatile("ds_read_b64 %0, %1 offset:16448" : "=v"(d[4]) : "v"(addr)); break;
        case 5: asm volatile("ds_read_b64 %0, %1 offset:20560" : "=v"(d[5]) : "v"(addr)); break;
        case 6: asm volatile("ds_read_b64 %0, %1 offset:24672" : "=v"(d[6]) : "v"(addr)); break;
        default: asm volatile("ds_read_b64 %0, %1 offset:28784" : "=v"(d[7]) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); break;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
template <int N> __device__ __forceinline__ void ldsw128(uint32_t addr, v4f v)
{
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("ds_write_b128 %0, %1 offset:0" :: "v"(addr), "v"(v) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

struct Stamp { uint64_t c0, c1, r0, r1; };

template <int MIX>
__global__ __launch_bounds__(kWaves * 64) void k_probe(Stamp *st, const char *in, char *out, int64_t tiles_per_wave, int iters, float *sink)
{
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t gw = (int64_t)blockIdx.x * kWaves + wave;
    // fill the LDS with something that is not all zeros
    for (int i = threadIdx.x; i < 40 * 1024 / 4; i += kWaves * 64) ((float *)lds)[i] = 1.0f + (float)(i * 2654435761u >> 9) * 1.0e-7f;
    __syncthreads();
    v2f acc[16]; float acc1[32]; uint32_t ir[16]; v4f d4[4] = {}; v2f d2[8] = {};
    const float fl = (float)(lane * 37 % 64) * (1.0f / 64.0f);
    const v2f a0 = {-0.984375f - fl * 0.0078125f, 0.9921875f - fl * 0.00390625f}, b0 = {0.37f + fl, -1.91f * fl - 0.11f}, b1 = {-3.3f * fl + 0.7f, 0.013f + fl * fl};
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = v2f{fl + i, fl - i}; ir[i] = lane * 2654435761u + i; }
#pragma unroll
    for (int i = 0; i < 32; ++i) acc1[i] = fl * i;
    const uint32_t a128 = (uint32_t)((wave & 7) * 4096 + lane * 16);                                        // 16 bytes per lane, conflict-free
    uint32_t a64 = (uint32_t)((lane * 2654435761u >> 13) % 256) * 8;                                        // lane-random arm, planes 4112 bytes apart
    const char *gp = in + gw * tiles_per_wave * 3072 + lane * 16;
    v4f g[6] = {};
    uint64_t c0, r0;
    c0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        const v2f b = (it & 1) ? b1 : b0;
        if (MIX == 0) pkfma<128>(acc, a0, b);
        if (MIX == 1) fma1<256>(acc1, a0.x, b.x);
        if (MIX == 2) lds128<32>(a128, d4);
        if (MIX == 3) { lds64<32>(a64, d2); a64 = (a64 + 8 * 103) % 2048; }
        if (MIX == 4) {
            const char *q = gp + (int64_t)(it % tiles_per_wave) * 3072;
#pragma unroll
            for (int c = 0; c < 3; ++c) g[c] = __builtin_nontemporal_load((const v4f *)(q + c * 1024));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc[0] += v2f{g[0].x + g[1].y, g[2].z};
        }
        if (MIX == 5) ialu<256>(ir, (uint32_t)it * 0x9e3779b9u + lane);
        if (MIX == 6) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("s_nop 7");
        }
        if (MIX >= 7) {
            constexpr bool L = MIX != 8, F = MIX != 9, G = MIX != 10;
            const char *q = gp + (int64_t)(it % tiles_per_wave) * 3072;
            if (G) {
#pragma unroll
                for (int c = 0; c < 3; ++c) g[c] = *(const v4f *)(q + c * 1024);
#pragma unroll
                for (int c = 0; c < 3; ++c) g[3 + c] = *(const v4f *)(q + c * 1024 + 16 * ((lane + 1) & 63) - 16 * lane);
            }
            if (F) pkfma<53>(acc, a0, b);
            ialu<32>(ir, (uint32_t)it + lane);
            if (L) ldsw128<6>(a128, d4[0]);
            if (L) lds128<12>(a128, d4);
            if (F) pkfma<53>(acc, a0, b0);
            ialu<32>(ir, 77u);
            if (L) { lds64<16>(a64, d2); a64 = (a64 + 8 * 103) % 2048; }
            if (F) pkfma<53>(acc, a0, b1);
            ialu<32>(ir, 99u);
            if (L) ldsw128<6>(a128, d4[1]);
            if (L) lds128<12>(a128, d4);
            if (L) { lds64<16>(a64, d2); a64 = (a64 + 8 * 103) % 2048; }
            if (F) pkfma<53>(acc, a0, b);
            ialu<32>(ir, 1234567u);
            if (G) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc[1] += v2f{g[0].x + g[1].y + g[3].x, g[2].z + g[4].y + g[5].w};
                typedef float v3f __attribute__((ext_vector_type(3), aligned(4)));
                *(v3f *)(out + (gw * tiles_per_wave + it % tiles_per_wave) * 768 + lane * 12) = v3f{acc[0].x, acc[1].y, acc[2].x};
            }
        }
    }
    const uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y + (float)ir[i];
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc1[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += d4[i].x + d4[i].w;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += d2[i].x + d2[i].y;
    if (s == 1.2345678f) sink[0] = s;
    if (lane == 0) st[gw] = Stamp{c0, c1, r0, r1};
}

typedef void (*probe_fn)(Stamp *, const char *, char *, int64_t, int, float *);
static const probe_fn kFns[] = {k_probe<0>, k_probe<1>, k_probe<2>, k_probe<3>, k_probe<4>, k_probe<5>, k_probe<6>, k_probe<7>, k_probe<8>, k_probe<9>, k_probe<10>};
static const char *kNames[] = {"pk_fma only", "v_fma only", "ds_read_b128", "ds_read_b64 gather", "global stream", "int VALU", "s_nop",
                               "front_mid mix", "mix - LDS", "mix - FMA", "mix - global"};
// iterations so that a launch lasts roughly 0.3 - 0.5 ms
static const int kIters[] = {300, 300, 900, 500, 114, 300, 900, 114, 114, 114, 114};

int main(int argc, char **argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 1.5;
    const int only = argc > 2 ? atoi(argv[2]) : -1;
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int n_cu = pr.multiProcessorCount;
    const int64_t tiles_per_wave = 114;
    const size_t in_bytes = (size_t)n_cu * kWaves * tiles_per_wave * 3072 + 4096;
    char *in, *out; Stamp *st; float *sink;
    CK(hipMalloc(&out, (size_t)n_cu * kWaves * tiles_per_wave * 768 + 4096));
    CK(hipMalloc(&in, in_bytes)); CK(hipMemset(in, 1, in_bytes));
    CK(hipMalloc(&st, sizeof(Stamp) * n_cu * kWaves)); CK(hipMalloc(&sink, 64));
    std::vector<Stamp> h((size_t)n_cu * kWaves);
    for (int m = 0; m < 11; ++m) CK(hipFuncSetAttribute((const void *)kFns[m], hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024));
    printf("# %s, %d CUs; %.1f s of back-to-back launches per point; clock from the LAST launch's stamps (median over waves)\n", pr.name, n_cu, secs);
    printf("# %-20s %5s %10s %10s %12s\n", "mix", "CUs", "launch ms", "clock GHz", "cycles/iter");
    for (int m = 0; m < 11; ++m) {
        if (m < only) continue;
        for (int cus : {n_cu, n_cu / 2, n_cu / 8}) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            // calibrate
            kFns[m]<<<cus, kWaves * 64, 40 * 1024>>>(st, in, out, tiles_per_wave, kIters[m], sink);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) kFns[m]<<<cus, kWaves * 64, 40 * 1024>>>(st, in, out, tiles_per_wave, kIters[m], sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const int n = std::max(20, (int)(secs * 1000.0 / (ms / 10.0)));
            CK(hipEventRecord(e0));
            for (int i = 0; i < n; ++i) kFns[m]<<<cus, kWaves * 64, 40 * 1024>>>(st, in, out, tiles_per_wave, kIters[m], sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), st, sizeof(Stamp) * cus * kWaves, hipMemcpyDeviceToHost));
            std::vector<double> ghz, cyc;
            for (int w = 0; w < cus * kWaves; ++w) {
                const double dc = (double)(h[w].c1 - h[w].c0), dr = (double)(h[w].r1 - h[w].r0);
                if (dr > 0) { ghz.push_back(dc / dr * 0.1); cyc.push_back(dc / kIters[m]); }
            }
            std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
            printf("  %-20s %5d %10.4f %10.3f %12.1f\n", kNames[m], cus, ms / n, ghz[ghz.size() / 2], cyc[cyc.size() / 2]);
            fflush(stdout);
            CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        }
    }
    return 0;
}
