// p0fft_cs8.hip -- the cs8 instantiations of k_p0fft16 (p0fft.hpp), a translation unit of their own so that the three formats compile
// side by side
#include "p0fft.hpp"

namespace iqgpu {
hipError_t launch_p0fft_cs8(const FftConvArgs &a, size_t lds, hipStream_t s) { return launch_p0fft_fmt<IQGPU_FMT_CS8>(a, lds, s); }
} // namespace iqgpu
