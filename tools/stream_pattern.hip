// stream_pattern.hip -- how much of the HBM rate does the ACCESS PATTERN of the wave-autonomous front kernels leave?
// 12 waves x 256 CUs, every wave a "run" of 3 KB tiles (3 coalesced 16-byte loads per lane, next tile prefetched in registers),
// 12 bytes per lane written per tile (the headline chain's 0.31 output frames per input frame), nothing computed.
//   mode 0: runs = contiguous pieces of the buffer, one per wave            (what k_front_mid / k_front_s1 / k_cascade do)
//   mode 1: tiles dealt round-robin over all waves                           (adjacent waves touch adjacent tiles)
//   mode 2: runs contiguous per WORKGROUP, its 12 waves interleaved inside   (12 x fewer concurrent streams)
//   +4: read only (no stores)   +8: non-temporal stores   +16: non-temporal loads
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_pattern tools/stream_pattern.hip ; run: /tmp/stream_pattern [log2_frames]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

constexpr int kWaves = 12, kTileB = 3072;

__global__ __launch_bounds__(kWaves * 64) void k_stream(const char *in, char *out, int64_t n_tiles, int mode, int64_t total_waves)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t gw = (int64_t)blockIdx.x * kWaves + wave;
    const bool wr = !(mode & 4);
    const int m = mode & 3;
    int64_t t0, t1, stride;
    if (m == 0) { const int64_t q = n_tiles / total_waves; t0 = gw * q; t1 = t0 + q; stride = 1; }
    else if (m == 1) { t0 = gw; t1 = n_tiles / total_waves * total_waves; stride = total_waves; }
    else { const int64_t q = n_tiles / gridDim.x / kWaves * kWaves; t0 = (int64_t)blockIdx.x * q + wave; t1 = (int64_t)blockIdx.x * q + q; stride = kWaves; }
    uint4 a[3], acc = make_uint4(0, 0, 0, 0);
    const char *p = in + t0 * kTileB + lane * 16;
#pragma unroll
    for (int c = 0; c < 3; ++c) a[c] = *(const uint4 *)(p + c * 1024);
    for (int64_t t = t0; t < t1; t += stride) {
        uint4 b[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) b[c] = a[c];
        const int64_t tn = t + stride < t1 ? t + stride : t;
        p = in + tn * kTileB + lane * 16;
#pragma unroll
        for (int c = 0; c < 3; ++c) { typedef uint32_t u4 __attribute__((ext_vector_type(4))); if (mode & 16) { const u4 v = __builtin_nontemporal_load((const u4 *)(p + c * 1024)); a[c] = make_uint4(v.x, v.y, v.z, v.w); } else a[c] = *(const uint4 *)(p + c * 1024); }
#pragma unroll
        for (int c = 0; c < 3; ++c) { acc.x += b[c].x; acc.y ^= b[c].y; acc.z += b[c].z; acc.w ^= b[c].w; }
        if (wr) {
            typedef uint32_t u3 __attribute__((ext_vector_type(3), aligned(4)));
            if (mode & 8) __builtin_nontemporal_store(u3{acc.x, acc.y, acc.z}, (u3 *)(out + t * 768 + lane * 12));
            else *(u3 *)(out + t * 768 + lane * 12) = u3{acc.x, acc.y, acc.z};
        }
    }
    if (acc.w == 0x12345678u) out[0] = 1;
}

// the same with TWO tiles in flight per wave (6 KB): is mode 0 bound by bytes in flight or by the memory side?
__global__ __launch_bounds__(kWaves * 64) void k_stream2(const char *in, char *out, int64_t n_tiles, int mode, int64_t total_waves)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t gw = (int64_t)blockIdx.x * kWaves + wave;
    const bool wr = !(mode & 4);
    const int64_t q = n_tiles / total_waves, t0 = gw * q, t1 = t0 + q;
    uint4 a[3], b[3], acc = make_uint4(0, 0, 0, 0);
    const char *p = in + t0 * kTileB + lane * 16;
#pragma unroll
    for (int c = 0; c < 3; ++c) { a[c] = *(const uint4 *)(p + c * 1024); b[c] = *(const uint4 *)(p + kTileB + c * 1024); }
    for (int64_t t = t0; t < t1; t += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint4 (&cur)[3] = h ? b : a;
            uint4 v[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = cur[c];
            const int64_t tn = t + h + 2 < t1 ? t + h + 2 : t + h;
            p = in + tn * kTileB + lane * 16;
#pragma unroll
            for (int c = 0; c < 3; ++c) cur[c] = *(const uint4 *)(p + c * 1024);
#pragma unroll
            for (int c = 0; c < 3; ++c) { acc.x += v[c].x; acc.y ^= v[c].y; acc.z += v[c].z; acc.w ^= v[c].w; }
            if (wr) {
                typedef uint32_t u3 __attribute__((ext_vector_type(3), aligned(4)));
                *(u3 *)(out + (t + h) * 768 + lane * 12) = u3{acc.x, acc.y, acc.z};
            }
        }
    }
    if (acc.w == 0x12345678u) out[0] = 1;
}

int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 28;
    const int64_t frames = (int64_t)1 << lg, bytes = frames * 4, n_tiles = bytes / kTileB;
    char *in, *out;
    (void)hipMalloc(&in, bytes + 65536); (void)hipMalloc(&out, n_tiles * 768 + 65536);
    (void)hipMemset(in, 1, bytes); (void)hipMemset(out, 0, n_tiles * 768);
    int cus = 256;
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0); cus = pr.multiProcessorCount;
    const int64_t total_waves = (int64_t)cus * kWaves;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode : {0, 8, 16, 24, 1, 9, 25, 4, 20}) {
        for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(k_stream, dim3(cus), dim3(kWaves * 64), 0, 0, in, out, n_tiles, mode, total_waves);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        const int reps = 40;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_stream, dim3(cus), dim3(kWaves * 64), 0, 0, in, out, n_tiles, mode, total_waves);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        const double rd = (double)n_tiles * kTileB, wrb = (mode & 4) ? 0.0 : (double)n_tiles * 768;
        printf("mode %d: %.4f ms  read %.2f TB/s  read+write %.2f TB/s\n", mode, ms, rd / ms / 1e9, (rd + wrb) / ms / 1e9);
    }
    for (int mode : {0, 4}) {
        for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(k_stream2, dim3(cus), dim3(kWaves * 64), 0, 0, in, out, n_tiles, mode, total_waves);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        const int reps = 40;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_stream2, dim3(cus), dim3(kWaves * 64), 0, 0, in, out, n_tiles, mode, total_waves);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        const double rd = (double)n_tiles * kTileB, wrb = (mode & 4) ? 0.0 : (double)n_tiles * 768;
        printf("two tiles in flight, mode %d: %.4f ms  read %.2f TB/s  read+write %.2f TB/s\n", mode, ms, rd / ms / 1e9, (rd + wrb) / ms / 1e9);
    }
    return 0;
}
