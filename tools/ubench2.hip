// ubench2.hip -- round-2 micro-measurements (MI355X) behind the k_front_s1 restructuring:
//   clock    : in-kernel shader clock (s_memtime / s_memrealtime) of every test below
//   dpp      : issue rate of v_add_f32 with DPP operand (wave_shr:1, row_shr:1) against plain v_add_f32 / v_mov_dpp
//   overlap  : a wave that alternates F packed FMAs with R ds_read_b128 + W ds_write_b128 (16 waves per CU):
//              time of FMA only, LDS only, both -- how far VALU and LDS pipes overlap across the waves of a CU
//   bperm    : ds_bpermute_b32 rate
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench2.hip -o tools/ubench2 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Clk { unsigned long long cyc, rt; };

#define CLK_BEGIN const unsigned long long c0_ = __builtin_amdgcn_s_memtime(), r0_ = __builtin_amdgcn_s_memrealtime();
#define CLK_END(clk) do { const unsigned long long c1_ = __builtin_amdgcn_s_memtime(), r1_ = __builtin_amdgcn_s_memrealtime(); \
        if (threadIdx.x == 0) { clk[blockIdx.x].cyc = c1_ - c0_; clk[blockIdx.x].rt = r1_ - r0_; } } while (0)

// ---------------------------------------------------------------- DPP
template <int MODE>
__global__ __launch_bounds__(1024) void k_dpp(float *out, Clk *clk, int iters)
{
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    const float x = 1e-3f;
    CLK_BEGIN
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#define ONE(A)                                                                                                         \
            if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(A) : "v"(x));                                     \
            else if (MODE == 1) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(A) : "v"(x)); \
            else if (MODE == 2) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(A) : "v"(x));  \
            else if (MODE == 3) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(A));              \
            else if (MODE == 4) asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(A) : "v"(x)); \
            else if (MODE == 5) asm volatile("v_add_f32_dpp %0, %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(A) : "v"(x));
            ONE(a0) ONE(a1) ONE(a2) ONE(a3) ONE(a4) ONE(a5) ONE(a6) ONE(a7)
#undef ONE
        }
    }
    CLK_END(clk);
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// ---------------------------------------------------------------- ds_bpermute
__global__ __launch_bounds__(1024) void k_bperm(float *out, Clk *clk, int iters)
{
    int a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3;
    const int addr = ((threadIdx.x + 5) & 63) * 4;
    CLK_BEGIN
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a0) : "v"(addr));
            asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a1) : "v"(addr));
            asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a2) : "v"(addr));
            asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a3) : "v"(addr));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    CLK_END(clk);
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3);
}

// ---------------------------------------------------------------- VALU / LDS overlap
// per iteration and wave: F v_pk_fma_f32 (SGPR tap pair), R ds_read_b128 (conflict-free rows of 48 B),
// W ds_write_b128; the reads of an iteration are consumed at the top of the next one.
template <int F, int R, int W>
__global__ __launch_bounds__(1024) void k_overlap(float *out, Clk *clk, int iters, float t0, float t1)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)smem + wave * 8192 + lane * 48;
    v2f acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = v2f{(float)lane, (float)i};
    v4f rd[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rd[i] = v4f{0.f, 0.f, 0.f, 0.f};
    const v2f tt = {t0, t1};
    const v2f y = {1e-3f, 2e-3f};
    CLK_BEGIN
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rd[r & 7]) : "v"(base), "n"((r % 4) * 16 + (r / 4) * 3072));
#pragma unroll
        for (int w = 0; w < W; ++w) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(base), "v"(rd[(w + 4) & 7]), "n"(w * 3072 + 4096) : "memory");
#pragma unroll
        for (int f = 0; f < F; ++f) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(acc[f & 7]) : "s"(tt), "v"(y));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < (R < 8 ? R : 8); ++r) acc[r].x += rd[r].x;
    }
    CLK_END(clk);
    v2f s = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); return ms; }

static double clock_ghz(Clk *d_clk, int blocks)
{
    std::vector<Clk> h(blocks);
    CK(hipMemcpy(h.data(), d_clk, blocks * sizeof(Clk), hipMemcpyDeviceToHost));
    std::vector<double> g;
    for (auto &c : h) if (c.rt) g.push_back((double)c.cyc / (double)c.rt * 0.1);
    std::sort(g.begin(), g.end());
    return g.empty() ? 0.0 : g[g.size() / 2];
}

int main(int argc, char **argv)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256;                       // one 1024-thread workgroup (16 waves, 4 per SIMD) per CU
    float *d_out; CK(hipMalloc(&d_out, (size_t)blocks * 1024 * sizeof(float)));
    Clk *d_clk; CK(hipMalloc(&d_clk, blocks * sizeof(Clk)));
    const int iters = 8192;
    const char *names[] = {"v_add_f32", "v_add_f32_dpp wave_shr:1", "v_add_f32_dpp row_shr:1", "v_mov_b32_dpp wave_shr:1", "v_fmac_f32_dpp row_shr:1", "v_add_f32_dpp wave_ror:1"};
    for (int mode = 0; mode < 6; ++mode) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            switch (mode) {
            case 0: hipLaunchKernelGGL(k_dpp<0>, dim3(blocks), dim3(1024), 0, 0, d_out, d_clk, iters); break;
            case 1: hipLaunchKernelGGL(k_dpp<1>, dim3(blocks), dim3(1024), 0, 0, d_out, d_clk, iters); break;
            case 2: hipLaunchKernelGGL(k_dpp<2>, dim3(blocks), dim3(1024), 0, 0, d_out, d_clk, iters); break;
            case 3: hipLaunchKernelGGL(k_dpp<3>, dim3(blocks), dim3(1024), 0, 0, d_out, d_clk, iters); break;
            case 4: hipLaunchKernelGGL(k_dpp<4>, dim3(blocks), dim3(1024), 0, 0, d_out, d_clk, iters); break;
            default: hipLaunchKernelGGL(k_dpp<5>, dim3(blocks), dim3(1024), 0, 0, d_out, d_clk, iters); break;
            }
            CK(hipEventRecord(e1));
            ms = time_ms(e0, e1);
        }
        const double ghz = clock_ghz(d_clk, blocks);
        printf("%-28s 4 waves/SIMD  %.3f ms  clock %.2f GHz  %.2f cycles/wave-instr/SIMD\n", names[mode], ms, ghz,
               ms * 1e-3 * ghz * 1e9 / ((double)iters * 32 * 4));
    }
    {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_bperm, dim3(blocks), dim3(1024), 0, 0, d_out, d_clk, iters / 4);
            CK(hipEventRecord(e1));
            ms = time_ms(e0, e1);
        }
        const double ghz = clock_ghz(d_clk, blocks);
        printf("%-28s 16 waves/CU   %.3f ms  clock %.2f GHz  %.2f cycles/wave-instr/CU\n", "ds_bpermute_b32", ms, ghz,
               ms * 1e-3 * ghz * 1e9 / ((double)(iters / 4) * 32 * 16));
    }
    const size_t lds = 16 * 8192;
#define OV(F, R, W)                                                                                                     \
    do {                                                                                                                \
        CK(hipFuncSetAttribute((const void *)k_overlap<F, R, W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        float ms = 0;                                                                                                   \
        for (int rep = 0; rep < 2; ++rep) {                                                                             \
            CK(hipEventRecord(e0));                                                                                     \
            hipLaunchKernelGGL((k_overlap<F, R, W>), dim3(blocks), dim3(1024), lds, 0, d_out, d_clk, iters / 4, 0.999f, 0.998f); \
            CK(hipEventRecord(e1));                                                                                     \
            ms = time_ms(e0, e1);                                                                                       \
        }                                                                                                               \
        const double ghz = clock_ghz(d_clk, blocks);                                                                    \
        printf("overlap F=%3d pk_fma R=%2d rd128 W=%d wr128: %.3f ms  clock %.2f GHz  %.0f cycles/iteration/wave-slot (16 waves: x4 per SIMD)\n", \
               F, R, W, ms, ghz, ms * 1e-3 * ghz * 1e9 / (double)(iters / 4));                                          \
    } while (0)
    OV(80, 0, 0);
    OV(0, 14, 4);
    OV(80, 14, 4);
    OV(136, 0, 0);
    OV(0, 23, 6);
    OV(136, 23, 6);
    OV(136, 12, 2);
    OV(80, 7, 0);
    OV(80, 0, 4);
    return 0;
}
