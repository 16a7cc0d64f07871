// ubench3.hip -- round-2 (late) micro-measurements behind the half-band on the matrix pipe (MI355X):
//   mix      : a wave that issues F v_pk_fma_f32 and M v_mfma_f32_4x4x1_16b_f32 per iteration, interleaved
//              (16 waves per CU): what an MFMA costs on the issue port the VALU shares with it
//   exact    : the half-band as 23 rank-1 MFMA steps per component (each lane one column, the four outputs of the
//              lane the four rows, taps as the A operand by lane % 4) against the fmaf chain in the same tap order --
//              bit-identical or not
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench3.hip -o tools/ubench3 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Clk { unsigned long long cyc, rt; };
#define CLK_BEGIN const unsigned long long c0_ = __builtin_amdgcn_s_memtime(), r0_ = __builtin_amdgcn_s_memrealtime();
#define CLK_END(clk) do { const unsigned long long c1_ = __builtin_amdgcn_s_memtime(), r1_ = __builtin_amdgcn_s_memrealtime(); \
        if (threadIdx.x == 0) { clk[blockIdx.x].cyc = c1_ - c0_; clk[blockIdx.x].rt = r1_ - r0_; } } while (0)

// F packed FMAs and M MFMAs per iteration, spread evenly over each other
template <int F, int M>
__global__ __launch_bounds__(1024) void k_mix(float *out, Clk *clk, int iters, float t0, float t1)
{
    const int lane = threadIdx.x & 63;
    v2f acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = v2f{(float)lane, (float)i};
    v4f d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {1.f, 1.f, 1.f, 1.f};
    const v2f tt = {t0, t1};
    const v2f y = {1e-3f, 2e-3f};
    const float a = 1e-3f * (float)(lane & 3), b = 1e-3f * (float)lane;
    constexpr int N = F > M ? F : M;
    CLK_BEGIN
    for (int it = 0; it < iters; ++it) {
        int fi = 0, mi = 0;
#pragma unroll
        for (int n = 0; n < N; ++n) {
            if ((n + 1) * F / N > fi) { asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(acc[fi & 7]) : "s"(tt), "v"(y)); ++fi; }
            if ((n + 1) * M / N > mi) {
                if (mi & 1) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(d1) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(d0) : "v"(a), "v"(b));
                ++mi;
            }
        }
    }
    CLK_END(clk);
    v2f s = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + d0.x + d0.y + d0.z + d0.w + d1.x + d1.y + d1.z + d1.w;
}

// the same with v_mfma_f32_16x16x4_f32 (8 passes)
template <int F, int M>
__global__ __launch_bounds__(1024) void k_mix16(float *out, Clk *clk, int iters, float t0, float t1)
{
    const int lane = threadIdx.x & 63;
    v2f acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = v2f{(float)lane, (float)i};
    v4f d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {1.f, 1.f, 1.f, 1.f};
    const v2f tt = {t0, t1};
    const v2f y = {1e-3f, 2e-3f};
    const float a = 1e-3f * (float)(lane & 3), b = 1e-3f * (float)lane;
    constexpr int N = F > M ? F : M;
    CLK_BEGIN
    for (int it = 0; it < iters; ++it) {
        int fi = 0, mi = 0;
#pragma unroll
        for (int n = 0; n < N; ++n) {
            if ((n + 1) * M / N > mi) {
                if (mi & 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(d1) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(d0) : "v"(a), "v"(b));
                ++mi;
            }
            if ((n + 1) * F / N > fi) { asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(acc[fi & 7]) : "s"(tt), "v"(y)); ++fi; }
        }
    }
    CLK_END(clk);
    v2f s = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + d0.x + d0.y + d0.z + d0.w + d1.x + d1.y + d1.z + d1.w;
}

// D = A B + C with v_mfma_f32_16x16x4_f32: A[m][k] from lane 16 k + m, B[k][n] from lane 16 k + n, D[4 (l / 16) + i][l % 16] in register i
__global__ __launch_bounds__(64) void k_exact16(const float *A, const float *B, const float *C, float *D)
{
    const int l = threadIdx.x;
    v4f c = {C[(4 * (l / 16) + 0) * 16 + l % 16], C[(4 * (l / 16) + 1) * 16 + l % 16], C[(4 * (l / 16) + 2) * 16 + l % 16], C[(4 * (l / 16) + 3) * 16 + l % 16]};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l % 16) * 4 + l / 16], B[(l / 16) * 16 + l % 16], c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * (l / 16) + i) * 16 + l % 16] = c[i];
}

// one wave: lane l has a window E[0 .. 23] and a start value c[0 .. 3]; out[r] = c[r] + sum_i h[i] E[20 + r - i], i ascending
__global__ __launch_bounds__(64) void k_exact(const float *E_in, const float *c_in, const float *h, float *out_mfma, float *out_fma)
{
    const int lane = threadIdx.x;
    float E[24];
    for (int k = 0; k < 24; ++k) E[k] = E_in[lane * 24 + k];
    v4f d = {c_in[lane * 4 + 0], c_in[lane * 4 + 1], c_in[lane * 4 + 2], c_in[lane * 4 + 3]};
    float ref[4] = {d.x, d.y, d.z, d.w};
    for (int r = 0; r < 4; ++r)
        for (int i = 0; i < 20; ++i) ref[r] = __builtin_fmaf(h[i], E[20 + r - i], ref[r]);
#pragma unroll
    for (int k = 23; k >= 1; --k) {
        const int ti = 20 + (lane & 3) - k;
        const float a = (ti >= 0 && ti < 20) ? h[ti] : 0.0f;
        d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, E[k], d, 0, 0, 0);
    }
    out_mfma[lane * 4 + 0] = d.x; out_mfma[lane * 4 + 1] = d.y; out_mfma[lane * 4 + 2] = d.z; out_mfma[lane * 4 + 3] = d.w;
    for (int r = 0; r < 4; ++r) out_fma[lane * 4 + r] = ref[r];
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

int main()
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256;
    float *d_out; CK(hipMalloc(&d_out, (size_t)blocks * 1024 * sizeof(float)));
    Clk *d_clk; CK(hipMalloc(&d_clk, blocks * sizeof(Clk)));
    const int iters = 2048;
#define MIX(F, M)                                                                                                       \
    do {                                                                                                                \
        float ms = 0;                                                                                                   \
        for (int rep = 0; rep < 2; ++rep) {                                                                             \
            CK(hipEventRecord(e0));                                                                                     \
            hipLaunchKernelGGL((k_mix<F, M>), dim3(blocks), dim3(1024), 0, 0, d_out, d_clk, iters, 0.999f, 0.998f);      \
            CK(hipEventRecord(e1));                                                                                     \
            ms = time_ms(e0, e1);                                                                                       \
        }                                                                                                               \
        const double ghz = clock_ghz(d_clk, blocks);                                                                    \
        printf("mix F=%3d pk_fma M=%2d mfma_4x4x1: %.3f ms  clock %.2f GHz  %.0f cycles/iteration/wave-slot (16 waves: x4 per SIMD)\n", \
               F, M, ms, ghz, ms * 1e-3 * ghz * 1e9 / (double)iters);                                                   \
    } while (0)
    MIX(136, 0);
    MIX(80, 0);
    MIX(56, 0);
    MIX(0, 54);
    MIX(56, 54);
    MIX(80, 54);
    MIX(136, 54);

#define MIX16(F, M)                                                                                                     \
    do {                                                                                                                \
        float ms = 0;                                                                                                   \
        for (int rep = 0; rep < 2; ++rep) {                                                                             \
            CK(hipEventRecord(e0));                                                                                     \
            hipLaunchKernelGGL((k_mix16<F, M>), dim3(blocks), dim3(1024), 0, 0, d_out, d_clk, iters, 0.999f, 0.998f);    \
            CK(hipEventRecord(e1));                                                                                     \
            ms = time_ms(e0, e1);                                                                                       \
        }                                                                                                               \
        const double ghz = clock_ghz(d_clk, blocks);                                                                    \
        printf("mix F=%3d pk_fma M=%2d mfma_16x16x4: %.3f ms  clock %.2f GHz  %.0f cycles/iteration/wave-slot (16 waves: x4 per SIMD)\n", \
               F, M, ms, ghz, ms * 1e-3 * ghz * 1e9 / (double)iters);                                                   \
    } while (0)
    MIX16(0, 18);
    MIX16(56, 18);
    MIX16(136, 18);
    MIX16(136, 6);
    {
        std::vector<float> A(64), B(64), C(256), D(256);
        float *dA, *dB, *dC, *dD;
        CK(hipMalloc(&dA, 256)); CK(hipMalloc(&dB, 256)); CK(hipMalloc(&dC, 1024)); CK(hipMalloc(&dD, 1024));
        int n_asc = 0, n_desc = 0, n_tree = 0, n_unf = 0, total = 0;
        srand(777);
        for (int trial = 0; trial < 200; ++trial) {
            for (auto &v : A) v = (float)rand() / 2147483648.f * 2.f - 1.f;
            for (auto &v : B) v = (float)rand() / 2147483648.f * 2.f - 1.f;
            for (auto &v : C) v = (float)rand() / 2147483648.f * 2.f - 1.f;
            CK(hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice));
            CK(hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_exact16, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
            CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
            for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
                const float *a = &A[m * 4]; float b[4]; for (int k = 0; k < 4; ++k) b[k] = B[k * 16 + n];
                const float c = C[m * 16 + n], d = D[m * 16 + n];
                float asc = c; for (int k = 0; k < 4; ++k) asc = fmaf(a[k], b[k], asc);
                float desc = c; for (int k = 3; k >= 0; --k) desc = fmaf(a[k], b[k], desc);
                const double ex = (double)c + (double)a[0] * b[0] + (double)a[1] * b[1] + (double)a[2] * b[2] + (double)a[3] * b[3];
                const float tree = (float)ex;
                float unf = c; for (int k = 0; k < 4; ++k) { volatile float pr = a[k] * b[k]; unf = unf + pr; }
                n_asc += memcmp(&asc, &d, 4) == 0; n_desc += memcmp(&desc, &d, 4) == 0; n_tree += memcmp(&tree, &d, 4) == 0; n_unf += memcmp(&unf, &d, 4) == 0;
                ++total;
            }
        }
        printf("exact16: of %d outputs of v_mfma_f32_16x16x4_f32 equal to: fma chain k ascending %d, k descending %d, exact sum rounded once %d, unfused ascending %d\n",
               total, n_asc, n_desc, n_tree, n_unf);
    }
    // exactness
    {
        std::vector<float> E(64 * 24), c(64 * 4), h(20), om(256), of(256);
        srand(12345);
        int bad_total = 0;
        float *dE, *dc, *dh, *dm, *df;
        CK(hipMalloc(&dE, E.size() * 4)); CK(hipMalloc(&dc, c.size() * 4)); CK(hipMalloc(&dh, 80)); CK(hipMalloc(&dm, 1024)); CK(hipMalloc(&df, 1024));
        for (int trial = 0; trial < 200; ++trial) {
            for (auto &v : E) v = (float)rand() / RAND_MAX * 2.f - 1.f;
            for (auto &v : c) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.5f;
            for (int i = 0; i < 10; ++i) h[i] = h[19 - i] = ((float)rand() / RAND_MAX - 0.5f) * (0.6f / (1 + (9 - i)));
            if (trial % 4 == 3) for (auto &v : E) v *= 1e-38f;      // denormal products
            CK(hipMemcpy(dE, E.data(), E.size() * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(dh, h.data(), 80, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_exact, dim3(1), dim3(64), 0, 0, dE, dc, dh, dm, df);
            CK(hipMemcpy(om.data(), dm, 1024, hipMemcpyDeviceToHost));
            CK(hipMemcpy(of.data(), df, 1024, hipMemcpyDeviceToHost));
            int bad = 0;
            for (int i = 0; i < 256; ++i) if (memcmp(&om[i], &of[i], 4) != 0) { if (bad_total + bad < 5) printf("  trial %d lane %d r %d: mfma %.9g fma %.9g\n", trial, i / 4, i % 4, om[i], of[i]); ++bad; }
            bad_total += bad;
        }
        printf("exact: %d of %d outputs differ between the MFMA chain and the fmaf chain\n", bad_total, 200 * 256);
    }
    return 0;
}
