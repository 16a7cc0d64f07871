// design.hpp -- host-side (one-off, create-time) derivation of every constant the kernels use.
//
// This is the product's own statement of the design rules the reference gets from liquid-dsp
// and from src/filter.c; the SPEC it follows is DESIGN.md section "SPEC".  It shares no code
// with the parity oracle (an independent C restatement that only the tests use).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/iqgpu.h"

namespace iqgpu {

struct cfloat { float re, im; };

// ---- liquid-dsp design primitives (SPEC B.1) ----
float    kaiser_beta_As(float As);
double   bessel_i0(double z);
double   kaiser_window(unsigned i, unsigned n, double beta);
void     firdes_kaiser(unsigned n, float fc, float As, float mu, float *h);
unsigned estimate_req_filter_len(float df, float As);

// ---- nco_crcf / LIQUID_NCO (SPEC B.4) ----
uint32_t nco_constrain(float theta);
void     nco_fill_table(float *sin1024);                 // tab[i] = sinf((float)(2 pi i / 1024))
void     nco_fill_sincos(cfloat *tab1024);               // {cos, sin} pairs: .re = cos, .im = sin

// ---- msresamp_crcf (SPEC B.6) ----
constexpr int kMaxStages = 12;
constexpr int kArbM = 7;          // resamp_crcf semi-length
constexpr int kArbNpfb = 256;     // polyphase arms
constexpr int kArbTaps = 2 * kArbM;   // 14 taps per arm
constexpr int kArbStride = 16;    // floats per arm row in the device table (14 taps + 2 zero pad)

struct HalfbandStage {
    int m = 0;                     // semi-length; prototype 4m+1 taps
    std::vector<float> proto;      // 4m+1 prototype h[k]
    std::vector<float> branch;     // 2m filter-branch taps h[2j+1], j = 0..2m-1 (symmetric)
};

struct ResamplePlan {
    bool  enabled = false;
    float ratio = 1.0f;
    bool  interp = false;
    int   S = 0;                           // number of half-band stages
    float rate_arb = 1.0f;
    uint32_t step = 1u << 24;              // 24-bit fixed-point phase increment
    std::vector<HalfbandStage> stages;     // RUN order when decimating: [0] = highest rate
    std::vector<float> arb_table;          // [256][kArbStride]: arm a, tap n = proto[a + 256 n]
    std::vector<float> arb_proto;          // 2*7*256 scaled prototype taps
    // input history (at the input rate) needed so that every stage window is exact
    uint32_t history_in = 0;
};
// As = RESAMPLER_QUALITY_ATTENUATION_DB (include/constants.h:137) unless overridden
bool make_resample_plan(float ratio, float As, ResamplePlan &p, std::string &err);

// ---- user filter chain (src/filter.c:43-393) ----
struct FilterPlan {
    bool enabled = false;
    bool post_resample = false;      // apply_user_filter_post_resample
    bool is_complex = false;
    int  impl = IQGPU_FI_NONE;       // IQGPU_FI_*
    uint32_t block = 0;              // fftfilt block size
    std::vector<cfloat> taps;        // combined, normalised taps h[k]
};
// returns IQGPU_OK or IQGPU_EFILTER
int make_filter_plan(const iqgpu_chain_desc &d, double input_rate, double target_rate,
                     FilterPlan &f, std::string &err);

} // namespace iqgpu
