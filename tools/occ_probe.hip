// occ_probe.hip -- how many 256-thread workgroups of a given dynamic LDS size does a gfx950 CU take?  (fftconv.hip: a fifth workgroup
// of N = 4096 needs its buffer in 32 768 bytes).  build: hipcc --offload-arch=gfx950 -O3 -o /tmp/occ_probe tools/occ_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(float *o) { extern __shared__ float s[]; s[threadIdx.x] = o[threadIdx.x]; __syncthreads(); o[threadIdx.x] = s[255 - threadIdx.x]; }
int main()
{
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int b : {16384, 20480, 27136, 27264, 27306, 32000, 32256, 32512, 32768, 33024, 33280, 33808, 40960, 54613, 81920, 163840}) {
        int n = -1; hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)k, 256, b);
        printf("lds %6d -> %d workgroups per CU (%s)\n", b, n, hipGetErrorString(e));
    }
    return 0;
}
