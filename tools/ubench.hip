// ubench.hip -- MI355X micro-measurements that the kernel design in DESIGN.md leans on:
//   fma     : v_fma_f32 issue rate (8 independent accumulators per lane)
//   pkfma   : v_pk_fma_f32 issue rate (same accumulators as 4 register pairs)
//   read    : streaming 16-byte-per-lane HBM read (cs16 frames), sum kept alive
//   rw      : the NRSC-5 traffic shape: read 4 B per frame, write 4 B per 3.225 frames
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_fma(float *out, int iters)
{
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    const float x = 0.999f, y = 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(x), "v"(y));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(x), "v"(y));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(x), "v"(y));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(x), "v"(y));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a4) : "v"(x), "v"(y));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a5) : "v"(x), "v"(y));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(x), "v"(y));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a7) : "v"(x), "v"(y));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ __launch_bounds__(256) void k_pkfma(float *out, int iters)
{
    v2f a0 = {(float)threadIdx.x, 1}, a1 = {2, 3}, a2 = {4, 5}, a3 = {6, 7}, a4 = {1, 2}, a5 = {3, 4}, a6 = {5, 6}, a7 = {7, 8};
    const v2f x = {0.999f, 0.998f}, y = {1e-3f, 2e-3f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(x), "v"(y));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(x), "v"(y));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(x), "v"(y));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(x), "v"(y));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a4) : "v"(x), "v"(y));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a5) : "v"(x), "v"(y));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(x), "v"(y));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a7) : "v"(x), "v"(y));
        }
    }
    v2f s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

// pkfma with the multiplier in an SGPR pair, as hipcc emits for the half-band taps
__global__ __launch_bounds__(256) void k_pkfma_s(float *out, int iters, float t0, float t1)
{
    v2f a0 = {(float)threadIdx.x, 1}, a1 = {2, 3}, a2 = {4, 5}, a3 = {6, 7}, a4 = {1, 2}, a5 = {3, 4}, a6 = {5, 6}, a7 = {7, 8};
    const v2f y = {1e-3f, 2e-3f};
    v2f tt = {t0, t1};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(a0) : "s"(tt), "v"(y));
            asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(a1) : "s"(tt), "v"(y));
            asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(a2) : "s"(tt), "v"(y));
            asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(a3) : "s"(tt), "v"(y));
            asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(a4) : "s"(tt), "v"(y));
            asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(a5) : "s"(tt), "v"(y));
            asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(a6) : "s"(tt), "v"(y));
            asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(a7) : "s"(tt), "v"(y));
        }
    }
    v2f s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

__global__ __launch_bounds__(256) void k_read(const uint4 *in, size_t n16, unsigned *out)
{
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint4 v = in[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// contiguous chunk per block (like the chain: each workgroup streams its own range)
__global__ __launch_bounds__(256) void k_rw(const uint4 *in, size_t n16, unsigned *out, size_t per_block16)
{
    const size_t b0 = (size_t)blockIdx.x * per_block16;
    size_t b1 = b0 + per_block16; if (b1 > n16) b1 = n16;
    for (size_t i = b0 + threadIdx.x; i < b1; i += 256) {
        const uint4 v = in[i];
        // 4 frames in -> ~1.24 frames out: lane writes one dword for 3 of every 10 loads... keep simple: 5 of 16 lanes
        const unsigned s = v.x ^ v.y ^ v.z ^ v.w;
        const size_t o = (i * 5) >> 2;     // 1.25 dwords per 16 B read
        out[o] = s;
        if ((i & 3) == 0) out[o + 1] = s;
    }
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main(int argc, char **argv)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float *d_out; CK(hipMalloc(&d_out, 256 * 64 * 256 * sizeof(float)));
    const int iters = 4096;
    for (int wps = 1; wps <= 8; wps *= 2) {           // waves per SIMD
        const int blocks = 256 * wps;                   // 256-thread blocks = 4 waves = 1 per SIMD
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, d_out, iters);
                else if (mode == 1) hipLaunchKernelGGL(k_pkfma, dim3(blocks), dim3(256), 0, 0, d_out, iters);
                else hipLaunchKernelGGL(k_pkfma_s, dim3(blocks), dim3(256), 0, 0, d_out, iters, 0.999f, 0.998f);
                CK(hipEventRecord(e1));
                const float ms = time_ms(e0, e1);
                if (rep == 1) {
                    const double instr = (double)iters * 32 * blocks * 4;        // wave-instructions
                    const double flop = instr * 64 * 2 * (mode ? 2 : 1);
                    printf("%-8s waves/SIMD=%d  %.3f ms  %.1f TFLOP/s  %.2f cycles/wave-instr/SIMD @2.4GHz\n",
                           mode == 0 ? "fma" : mode == 1 ? "pkfma" : "pkfma_s", wps, ms, flop / ms / 1e9,
                           ms * 1e-3 * 2.4e9 / ((double)iters * 32 * wps));
                }
            }
        }
    }
    // streaming
    const size_t bytes = (size_t)1 << 30;
    uint4 *d_in; unsigned *d_o; CK(hipMalloc(&d_in, bytes)); CK(hipMalloc(&d_o, bytes / 2));
    CK(hipMemset(d_in, 1, bytes));
    for (int blocks : {1024, 2048, 4096, 8192}) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, d_in, bytes / 16, d_o);
            CK(hipEventRecord(e1));
            const float ms = time_ms(e0, e1);
            if (rep == 2) printf("read     blocks=%d  %.3f ms  %.1f GB/s\n", blocks, ms, bytes / ms / 1e6);
        }
    }
    for (int blocks : {1024, 4096}) {
        const size_t per = (bytes / 16 + blocks - 1) / blocks;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_rw, dim3(blocks), dim3(256), 0, 0, d_in, bytes / 16, d_o, per);
            CK(hipEventRecord(e1));
            const float ms = time_ms(e0, e1);
            if (rep == 2) printf("rw       blocks=%d  %.3f ms  %.1f GB/s (read+write)\n", blocks, ms, (bytes * 1.3125) / ms / 1e6);
        }
    }
    return 0;
}
