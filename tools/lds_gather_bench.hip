// lds_gather_bench.hip -- cost of the polyphase tap gather (7 x ds_read_b64 per output) for several
// LDS layouts of the 256 x 14 tap table, with the arm sequence the NRSC-5 chain really produces.
// One wave per workgroup, one workgroup: cycles per gather instruction from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

// mode: 0 rows of 56 B, 1 rows of 56 B hashed (arm ^ (arm >> 5)), 2 tap-major [i][arm], 3 rows of 72 B,
//       4 rows of 64 B read as 4 x b128, 5 linear (conflict-free reference), 6 rows 56 B random arms
__global__ void k(int mode, const unsigned *arms_in, unsigned long long *out, float *sink, int reps)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) ((float *)smem)[i] = (float)i;
    __syncthreads();
    unsigned arm[4];
    for (int r = 0; r < 4; ++r) arm[r] = arms_in[r * 64 + lane];
    v2f acc = {0.f, 0.f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < reps; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned a = arm[r];
            v2f t[7];
            if (mode == 4) {
                unsigned row = a * 64u;
                float4 q0, q1, q2, q3;
                asm volatile("ds_read_b128 %0, %1" : "=v"(q0) : "v"(row));
                asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(q1) : "v"(row));
                asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(q2) : "v"(row));
                asm volatile("ds_read_b128 %0, %1 offset:48" : "=v"(q3) : "v"(row));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                acc.x += q0.x + q1.y + q2.z + q3.w;
            } else {
                unsigned row, st = 8;
                if (mode == 0 || mode == 6) row = a * 56u;
                else if (mode == 1) row = (a ^ (a >> 5)) * 56u;
                else if (mode == 2) { row = a * 8u; st = 2048; }
                else if (mode == 3) row = a * 72u;
                else row = lane * 8u;   // linear
                if (mode == 5) st = 512;
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    unsigned ad = row + st * i;
                    asm volatile("ds_read_b64 %0, %1" : "=v"(t[i]) : "v"(ad));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < 7; ++i) acc += t[i];
            }
            arm[r] = (arm[r] + (mode == 6 ? 97u : 0u)) & 255u;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (acc.x == 1.2345f) sink[lane] = acc.x + acc.y;
}

int main()
{
    // arms of the candidate scheme for NRSC-5: step = 27053208, lane l, candidate r
    const unsigned step = 27053208u;
    unsigned h_arms[256], h_rand[256];
    unsigned long long delta0 = 1234567;
    for (int l = 0; l < 64; ++l) {
        unsigned long long tgt = (unsigned long long)(4 * l) << 24;
        unsigned long long n0 = tgt > delta0 ? (tgt - delta0 + step - 1) / step : 0;
        unsigned Pl = (unsigned)(delta0 + n0 * step - tgt);
        for (int r = 0; r < 4; ++r) {
            h_arms[r * 64 + l] = (Pl >> 16) & 255u;
            if ((Pl >> 24) == (unsigned)r) Pl += step;
        }
    }
    srand(1);
    for (int i = 0; i < 256; ++i) h_rand[i] = rand() & 255;
    unsigned *d_arms; unsigned long long *d_out; float *d_sink;
    CK(hipMalloc(&d_arms, 1024)); CK(hipMalloc(&d_out, 64)); CK(hipMalloc(&d_sink, 1024));
    const char *names[] = {"rows 56B", "rows 56B hashed", "tap-major [i][arm]", "rows 72B", "rows 64B b128 x4", "linear (no conflict)", "rows 56B, random arms"};
    const int reps = 2000;
    for (int mode = 0; mode < 7; ++mode) {
        CK(hipMemcpy(d_arms, mode == 6 ? h_rand : h_arms, 1024, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms = 0;
        for (int w = 0; w < 2; ++w) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(256), dim3(1024), 40960, 0, mode, d_arms, d_out, d_sink, reps);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        unsigned long long cyc; CK(hipMemcpy(&cyc, d_out, 8, hipMemcpyDeviceToHost));
        // per CU: 16 waves x reps x 4 candidates x (7 or 4) read instructions
        const double instr = 16.0 * reps * 4 * (mode == 4 ? 4 : 7);
        printf("%-26s %7.3f ms  wave0 %llu cyc  -> %.2f cycles per read instruction per CU (clock from wave0: %.2f GHz)\n",
               names[mode], ms, cyc, (double)cyc / instr, cyc / (ms * 1e6));
    }
    return 0;
}
