// design.cpp -- create-time constants for the MI355X I/Q chain (host only, runs once per chain).
//
// Every rule here is cited to the reference call site whose behaviour it must reproduce; the
// library arithmetic behind those call sites (liquid-dsp) is restated from DESIGN.md "SPEC".
#include "design.hpp"

#include <cmath>
#include <cstring>

namespace iqgpu {

static const double kPi = 3.14159265358979323846;

// ------------------------------------------------------------------ Kaiser prototype (SPEC B.1)
float kaiser_beta_As(float As)
{
    As = std::fabs(As);
    if (As > 50.0f) return 0.1102f * (As - 8.7f);
    if (As > 21.0f) return (float)(0.5842 * std::pow((double)As - 21.0, 0.4) + 0.07886 * ((double)As - 21.0));
    return 0.0f;
}

double bessel_i0(double z)
{
    // power series of I0; terms are squared ratios so convergence is fast for beta < 30
    double term = 1.0, sum = 1.0;
    const double half = 0.5 * z;
    for (int k = 1; k < 64; ++k) {
        term *= half / (double)k;
        const double t2 = term * term;
        sum += t2;
        if (t2 < 1e-20 * sum) break;
    }
    return sum;
}

double kaiser_window(unsigned i, unsigned n, double beta)
{
    const double t = (double)i - 0.5 * (double)(n - 1);
    const double r = 2.0 * t / (double)(n - 1);
    double arg = 1.0 - r * r;
    if (arg < 0.0) arg = 0.0;
    return bessel_i0(beta * std::sqrt(arg)) / bessel_i0(beta);
}

// liquid's sincf(): three-cosine product near zero, sin(pi x)/(pi x) elsewhere
static double sinc_liquid(double x)
{
    if (std::fabs(x) < 0.01)
        return std::cos(kPi * x * 0.5) * std::cos(kPi * x * 0.25) * std::cos(kPi * x * 0.125);
    return std::sin(kPi * x) / (kPi * x);
}

void firdes_kaiser(unsigned n, float fc, float As, float mu, float *h)
{
    const double beta = (double)kaiser_beta_As(As);
    for (unsigned i = 0; i < n; ++i) {
        const double t = (double)i - 0.5 * (double)(n - 1) + (double)mu;
        h[i] = (float)(sinc_liquid(2.0 * (double)fc * t) * kaiser_window(i, n, beta));
    }
}

unsigned estimate_req_filter_len(float df, float As)
{
    // float expression, truncated (call site: src/filter.c:192)
    const float len = (As - 7.95f) / (14.26f * df);
    return (unsigned)len;
}

// ------------------------------------------------------------------ NCO (SPEC B.4)
uint32_t nco_constrain(float theta)
{
    // theta / 2pi as a float, fractional part in [0,1), scaled by 2^32 (the int constant
    // 0xffffffff converts to float 2^32 in the original expression)
    const float p = (float)((double)theta * 0.159154943091895);
    float frac = p - (float)((long)p);
    if (frac < 0.0f) frac += 1.0f;
    const double scaled = (double)(frac * 4294967296.0f);
    if (scaled >= 4294967296.0) return 0u;
    return (uint32_t)scaled;
}

void nco_fill_table(float *t)
{
    for (unsigned i = 0; i < 1024; ++i) {
        const float arg = (float)(2.0 * kPi * (double)(float)i / 1024.0);
        t[i] = sinf(arg);
    }
}

void nco_fill_sincos(cfloat *t)
{
    float s[1024];
    nco_fill_table(s);
    for (unsigned i = 0; i < 1024; ++i) {
        t[i].im = s[i];
        t[i].re = s[(i + 256) & 1023];
    }
}

// ------------------------------------------------------------------ msresamp_crcf (SPEC B.6)
static void design_halfband(int m, float As, HalfbandStage &st)
{
    const unsigned len = 4u * (unsigned)m + 1u;
    const double beta = (double)kaiser_beta_As(As);
    st.m = m;
    st.proto.resize(len);
    for (unsigned i = 0; i < len; ++i) {
        const double t = (double)i - 0.5 * (double)(len - 1);
        st.proto[i] = (float)(sinc_liquid(0.5 * t) * kaiser_window(i, len, beta));
    }
    st.branch.resize(2u * (unsigned)m);
    for (unsigned j = 0; j < 2u * (unsigned)m; ++j) st.branch[j] = st.proto[2 * j + 1];
}

bool make_resample_plan(float ratio, float As, ResamplePlan &p, std::string &err)
{
    if (!(ratio > 0.0f) || !std::isfinite(ratio)) { err = "resampling ratio must be positive"; return false; }
    p = ResamplePlan();
    p.enabled = true;
    p.ratio = ratio;
    p.interp = ratio > 1.0f;
    p.rate_arb = ratio;
    p.S = 0;
    if (p.interp) { while (p.rate_arb > 2.0f) { ++p.S; p.rate_arb *= 0.5f; } }
    else          { while (p.rate_arb < 0.5f) { ++p.S; p.rate_arb *= 2.0f; } }
    if (p.S > kMaxStages) { err = "too many half-band stages"; return false; }

    // msresamp2 stage parameters, design index i = 0 is the LOWEST-rate stage
    std::vector<HalfbandStage> by_design((size_t)p.S);
    {
        float fc = 0.4f;
        const float as_stage = As + 5.0f;
        for (int i = 0; i < p.S; ++i) {
            fc = (i == 1) ? (float)((0.5 - (double)fc) / 2.0) : 0.5f * fc;
            const float ft = 2 * (0.25f - fc);
            const unsigned hl = estimate_req_filter_len(ft, as_stage);
            int m = (int)std::ceil((float)(hl - 1) / 4.0f);
            if (m < 3) m = 3;
            design_halfband(m, as_stage, by_design[(size_t)i]);
        }
    }
    // decimation runs the highest design index first (at the highest rate)
    p.stages.resize((size_t)p.S);
    for (int k = 0; k < p.S; ++k) p.stages[(size_t)k] = by_design[(size_t)(p.S - 1 - k)];

    // arbitrary resampler: resamp_crcf(rate_arb, 7, min(0.515 rate, 0.49), As, 256)
    {
        const unsigned n = 2u * kArbM * kArbNpfb + 1u;
        const float fc = std::fmin(0.515f * p.rate_arb, 0.49f);
        std::vector<float> hf(n);
        firdes_kaiser(n, fc / (float)kArbNpfb, As, 0.0f, hf.data());
        double sum = 0.0;
        for (unsigned i = 0; i < n; ++i) sum += (double)hf[i];
        const float g = (float)((double)kArbNpfb / sum);
        p.arb_proto.resize(n - 1);
        for (unsigned i = 0; i < n - 1; ++i) p.arb_proto[i] = hf[i] * g;
        p.arb_table.assign((size_t)kArbNpfb * kArbStride, 0.0f);
        for (int a = 0; a < kArbNpfb; ++a)
            for (int t = 0; t < kArbTaps; ++t)
                p.arb_table[(size_t)a * kArbStride + t] = p.arb_proto[(size_t)a + (size_t)kArbNpfb * t];
        const float quo = 16777216.0f / p.rate_arb;     // float quotient, as (1<<24)/rate
        p.step = (uint32_t)std::llround((double)quo);
    }

    // history at the input rate: 13 samples at the arbitrary stage, doubled and widened by each
    // half-band stage walking up towards the input
    if (!p.interp) {
        uint64_t h = kArbTaps - 1;
        for (int k = p.S - 1; k >= 0; --k) h = 2 * h + 4u * (unsigned)p.stages[(size_t)k].m;
        p.history_in = (uint32_t)h;
    }
    return true;
}

// ------------------------------------------------------------------ user filter (src/filter.c)
static void invert_spectrum(std::vector<float> &t) // src/filter.c:94-99
{
    for (float &v : t) v = -v;
    t[(t.size() - 1) / 2] += 1.0f;
}

int make_filter_plan(const iqgpu_chain_desc &d, double input_rate, double target_rate,
                     FilterPlan &f, std::string &err)
{
    f = FilterPlan();
    if (d.n_filters <= 0) return IQGPU_OK;
    if (d.n_filters > 5) { err = "at most 5 chained filters (MAX_FILTER_CHAIN)"; return IQGPU_EFILTER; }
    f.enabled = true;

    // placement rule, src/filter.c:43-92
    if (!d.no_resample && target_rate < input_rate) {
        float top = 0.0f;
        for (int i = 0; i < d.n_filters; ++i) {
            const iqgpu_filter_req &r = d.filters[i];
            float edge = 0.0f;
            if (r.type == IQGPU_FILTER_LOWPASS || r.type == IQGPU_FILTER_HIGHPASS) edge = std::fabs(r.f1_hz);
            else if (r.type == IQGPU_FILTER_PASSBAND || r.type == IQGPU_FILTER_STOPBAND) edge = std::fabs(r.f1_hz) + (r.f2_hz / 2.0f);
            if (edge > top) top = edge;
        }
        if ((double)top > target_rate / 2.0) {
            err = "filter chain extends beyond the output Nyquist frequency";
            return IQGPU_EFILTER;
        }
        f.post_resample = true;
    }
    const double fs = f.post_resample ? target_rate : input_rate;
    const float fsf = (float)fs;

    std::vector<cfloat> master(1, cfloat{1.0f, 0.0f});
    bool by_peak = false;

    for (int i = 0; i < d.n_filters; ++i) {
        const iqgpu_filter_req &r = d.filters[i];
        if (r.type != IQGPU_FILTER_LOWPASS) by_peak = true;
        const float As = (d.attenuation_db > 0.0f) ? d.attenuation_db : 60.0f;

        unsigned n;
        if (d.filter_taps > 0) n = (unsigned)d.filter_taps;
        else { // src/filter.c:182-195
            float tw;
            if (d.transition_width_hz > 0.0f) tw = d.transition_width_hz;
            else {
                const float ref = (r.type == IQGPU_FILTER_LOWPASS || r.type == IQGPU_FILTER_HIGHPASS) ? r.f1_hz : r.f2_hz;
                tw = std::fabs(ref) * 0.25f;
            }
            if (tw < 1.0f) tw = 1.0f;
            n = estimate_req_filter_len(tw / fsf, As);
            if (n % 2 == 0) ++n;
            if (n < 21) n = 21;
        }

        std::vector<float> real_taps(n);
        std::vector<cfloat> cur(n);
        const bool off_centre = (r.type == IQGPU_FILTER_PASSBAND && std::fabs(r.f1_hz) > 1e-9f);
        if (off_centre) { // src/filter.c:205-218: Kaiser low-pass modulated by the table NCO
            f.is_complex = true;
            const float half_bw = (r.f2_hz / 2.0f) / fsf;
            const float fcn = r.f1_hz / fsf;
            firdes_kaiser(n, half_bw, As, 0.0f, real_taps.data());
            float tab[1024];
            nco_fill_table(tab);
            uint32_t theta = 0;
            const uint32_t dth = nco_constrain((float)(2.0f * kPi * fcn));
            for (unsigned k = 0; k < n; ++k) {
                const unsigned idx = ((theta + (1u << 21)) >> 22) & 1023u;
                cur[k].re = tab[(idx + 256) & 1023u] * real_taps[k];
                cur[k].im = tab[idx] * real_taps[k];
                theta += dth;
            }
        } else { // src/filter.c:219-247
            switch (r.type) {
            case IQGPU_FILTER_LOWPASS:
                firdes_kaiser(n, r.f1_hz / fsf, As, 0.0f, real_taps.data());
                break;
            case IQGPU_FILTER_HIGHPASS:
                firdes_kaiser(n, r.f1_hz / fsf, As, 0.0f, real_taps.data());
                invert_spectrum(real_taps);
                break;
            case IQGPU_FILTER_PASSBAND:
                firdes_kaiser(n, (r.f2_hz / fsf) / 2.0f, As, 0.0f, real_taps.data());
                break;
            case IQGPU_FILTER_STOPBAND:
                firdes_kaiser(n, (r.f2_hz / fsf) / 2.0f, As, 0.0f, real_taps.data());
                invert_spectrum(real_taps);
                break;
            default:
                err = "unknown filter type";
                return IQGPU_EFILTER;
            }
            for (unsigned k = 0; k < n; ++k) cur[k] = cfloat{real_taps[k], 0.0f};
        }

        // chain by convolution, src/filter.c:114-136 / 249-255
        const size_t len1 = master.size(), len2 = cur.size(), lo = len1 + len2 - 1;
        std::vector<cfloat> next(lo);
        for (size_t a = 0; a < lo; ++a) {
            const size_t b0 = (a >= len1) ? a - len1 + 1 : 0;
            const size_t b1 = (a < len2 - 1) ? a : len2 - 1;
            double sr = 0.0, si = 0.0;
            for (size_t b = b0; b <= b1; ++b) {
                const cfloat &p = master[a - b], &q = cur[b];
                sr += (double)p.re * q.re - (double)p.im * q.im;
                si += (double)p.re * q.im + (double)p.im * q.re;
            }
            next[a] = cfloat{(float)sr, (float)si};
        }
        master.swap(next);
    }

    // gain normalisation, src/filter.c:272-299
    if (by_peak || f.is_complex) {
        float peak = 0.0f;
        for (int p = 0; p < 2048; ++p) {
            const float fr = ((float)p / 2048.0f) - 0.5f;
            double hr = 0.0, hi = 0.0;
            for (size_t k = 0; k < master.size(); ++k) {
                const double ph = -2.0 * kPi * (double)fr * (double)k;
                const double c = std::cos(ph), s = std::sin(ph);
                hr += master[k].re * c - master[k].im * s;
                hi += master[k].re * s + master[k].im * c;
            }
            const float mag = (float)std::sqrt(hr * hr + hi * hi);
            if (mag > peak) peak = mag;
        }
        if (peak > 1e-9f) for (cfloat &t : master) { t.re /= peak; t.im /= peak; }
    } else {
        double dc = 0.0;
        for (const cfloat &t : master) dc += (double)t.re;
        if (std::fabs(dc) > 1e-9f) { const float g = (float)dc; for (cfloat &t : master) { t.re /= g; t.im /= g; } }
    }

    // implementation choice, src/filter.c:301-354
    int choice = d.filter_impl;
    if (choice == IQGPU_FILTER_IMPL_AUTO) choice = f.is_complex ? IQGPU_FILTER_IMPL_FFT : IQGPU_FILTER_IMPL_FIR;
    const unsigned L = (unsigned)master.size();
    if (choice == IQGPU_FILTER_IMPL_FFT) {
        unsigned bs;
        if (d.fft_size > 0) {
            bs = (unsigned)d.fft_size / 2;
            if (bs < L - 1) { err = "--filter-fft-size too small for the combined filter"; return IQGPU_EFILTER; }
        } else {
            bs = 1;
            while (bs < L - 1) bs *= 2;
            if (bs < L * 2) bs *= 2;
        }
        if (bs > 1024u * 1024u) { err = "FFT block exceeds MAX_ALLOWED_FFT_BLOCK_SIZE"; return IQGPU_EFILTER; }
        f.block = bs;
        f.impl = f.is_complex ? IQGPU_FI_FFT_ASYMMETRIC : IQGPU_FI_FFT_SYMMETRIC;
    } else {
        f.impl = f.is_complex ? IQGPU_FI_FIR_ASYMMETRIC : IQGPU_FI_FIR_SYMMETRIC;
    }
    if (!f.is_complex) for (cfloat &t : master) t.im = 0.0f; // crcf objects are built from crealf()
    f.taps.swap(master);
    return IQGPU_OK;
}

} // namespace iqgpu
