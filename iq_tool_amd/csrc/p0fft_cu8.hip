// p0fft_cu8.hip -- the cu8 instantiations of k_p0fft16 (p0fft.hpp), a translation unit of their own so that the three formats compile
// side by side
#include "p0fft.hpp"

namespace iqgpu {
size_t p0fft_tap_lds() { return (size_t)kFTapLds; }
hipError_t launch_p0fft_cu8(const FftConvArgs &a, size_t lds, hipStream_t s) { return launch_p0fft_fmt<IQGPU_FMT_CU8>(a, lds, s); }
} // namespace iqgpu
