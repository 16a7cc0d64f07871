// iq_optimizer.cpp -- the I/Q imbalance optimiser that produces the (mag, phase) pair iq_correct_apply consumes.
//
// Host code, as in the reference: a randomised hill climb on the asymmetry of a 1024-point spectrum, run by
// its own low-priority thread at most every 500 ms (src/iq_correct.c:154-219 iq_correct_run_optimization,
// 315-393 helpers; thread src/utility_threads.c:35-47; hand-off src/pipeline.c:468-476).  It never touches a
// device: the GPU chain hands it the first 1024 pre-processed samples of a call (iqgpu_chain_read_iq_probe)
// and takes the factors back through iqgpu_chain_set_iq_factors.
//
// Followed step by step, in float where the reference computes in float:
//   window      0.54f - 0.46f cosf(2 pi i / 1023)                                   iq_correct.c:122-124
//   spectrum    correct (re (1+mag), im + phase re) -> window -> forward FFT -> swap halves ->
//               20 log10f(|X| / 1024 + 1e-12f)                                      iq_correct.c:315-337
//   metric      sum over bins i in [25, 486) of (S[1023-i] - S[i])^2 where either side is above -80 dB   339-360
//   power gate  max - mean of the same bins, at least 20 dB                         362-389, 170-176
//   climb       25 candidates current +- 1e-4 (gain, then phase direction), keep a candidate that raises the metric   191-201
//   publish     factor <- 0.95 factor + 0.05 best                                   206-216
// Not replicated: the reference calls this through pre_stream_iq_correction BEFORE iq_correct_init has
// allocated fft_buffer (src/setup.c:291 vs src/pipeline.c:140), a NULL memcpy for files >= 1024 frames; here
// the optimiser owns its buffers from create().  The FFT is liquid's in the reference (any exact DFT to float
// rounding); here a radix-2 transform with double-precision twiddles rounded to float.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <new>

#include "../../include/iqgpu.h"

namespace {

constexpr int kN = 1024;                 // IQ_CORRECTION_FFT_SIZE, include/constants.h:157
constexpr double kIntervalMs = 500.0;    // IQ_CORRECTION_INTERVAL_MS
constexpr float kIncrement = 0.0001f;    // IQ_BASE_INCREMENT
constexpr int kPasses = 25;              // IQ_MAX_PASSES
constexpr float kPowerDb = 20.0f;        // IQ_CORRECTION_POWER_THRESHOLD_DB
constexpr float kSmooth = 0.05f;         // IQ_CORRECTION_SMOOTHING_FACTOR

struct cfl { float re, im; };

double monotonic_now()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

} // namespace

struct iqgpu_iq_optimizer {
    float window[kN];
    cfl tw[kN / 2];                      // exp(-2 pi i k / N)
    uint16_t rev[kN];
    cfl buf[kN];
    float spectrum[kN];
    // factors (the reference double-buffers them behind a mutex, iq_correct.c:141-152)
    std::mutex mu;
    float mag = 0.0f, phase = 0.0f;
    float average_power = 0.0f, power_range = 0.0f;
    double last_time = 0.0;
    // direction source: caller's function, the private generator (seeded), or libc rand() like the reference
    iqgpu_rand_dir_fn fn = nullptr; void *fn_user = nullptr;
    bool seeded = false; uint32_t lcg = 1;
    iqgpu_iq_optimizer_stats stats{};

    float direction()
    {
        if (fn) return fn(fn_user) > 0.0f ? 1.0f : -1.0f;
        if (seeded) {                     // minstd: x <- 48271 x mod (2^31 - 1); same threshold rule as below
            lcg = (uint32_t)(((uint64_t)lcg * 48271u) % 2147483647u);
            return lcg > 2147483647u / 2u ? 1.0f : -1.0f;
        }
        return (rand() > (RAND_MAX / 2)) ? 1.0f : -1.0f;            // iq_correct.c:391-393
    }

    void fft()
    {
        for (int i = 0; i < kN; ++i) { const int j = rev[i]; if (j > i) { const cfl t = buf[i]; buf[i] = buf[j]; buf[j] = t; } }
        for (int len = 2; len <= kN; len <<= 1) {
            const int half = len >> 1, stride = kN / len;
            for (int i = 0; i < kN; i += len)
                for (int k = 0; k < half; ++k) {
                    const cfl w = tw[k * stride];
                    const cfl a = buf[i + k], b = buf[i + k + half];
                    const float tr = b.re * w.re - b.im * w.im, ti = b.re * w.im + b.im * w.re;
                    buf[i + k] = cfl{a.re + tr, a.im + ti};
                    buf[i + k + half] = cfl{a.re - tr, a.im - ti};
                }
        }
    }

    // iq_correct.c:315-337
    void power_spectrum(const cfl *block, float gain_adj, float phase_adj)
    {
        const float magp1 = 1.0f + gain_adj;
        for (int i = 0; i < kN; ++i) {
            const float re = block[i].re, im = block[i].im;
            const float cr = re * magp1, ci = im + phase_adj * re;  // iq_correct.c:307-313
            buf[i] = cfl{cr * window[i], ci * window[i]};
        }
        fft();
        for (int i = 0; i < kN; ++i) {
            const cfl v = buf[(i + kN / 2) & (kN - 1)];             // swap the halves
            float m = hypotf(v.re, v.im);                           // cabsf
            m /= (float)kN;
            spectrum[i] = 20.0f * log10f(m + 1e-12f);
        }
    }

    // iq_correct.c:339-360
    float metric(const cfl *block, float gain_adj, float phase_adj)
    {
        power_spectrum(block, gain_adj, phase_adj);
        float total = 0.0f;
        const int lo = (int)(0.05f * (kN / 2)), hi = (int)(0.95f * (kN / 2));
        for (int i = lo; i < hi; ++i) {
            const float p_neg = spectrum[i], p_pos = spectrum[kN - 1 - i];
            if (p_pos > -80.0f || p_neg > -80.0f) { const float d = p_pos - p_neg; total += d * d; }
        }
        return total;
    }

    // iq_correct.c:362-389
    void estimate_power(const cfl *block)
    {
        power_spectrum(block, 0.0f, 0.0f);
        float max_power = -1000.0f;
        double sum = 0.0;
        int count = 0;
        const int lo = (int)(0.05f * (kN / 2)), hi = (int)(0.95f * (kN / 2));
        for (int i = lo; i < hi; ++i) {
            const float p_neg = spectrum[i], p_pos = spectrum[kN - 1 - i];
            if (p_pos > max_power) max_power = p_pos;
            if (p_neg > max_power) max_power = p_neg;
            sum += p_pos + p_neg;
            count += 2;
        }
        if (count > 0) { average_power = (float)(sum / count); power_range = max_power - average_power; }
        else { average_power = 0.0f; power_range = 0.0f; }
    }
};

extern "C" int iqgpu_iq_optimizer_create(iqgpu_iq_optimizer **out)
{
    if (!out) return IQGPU_EINVAL;
    iqgpu_iq_optimizer *o = new (std::nothrow) iqgpu_iq_optimizer();
    if (!o) return IQGPU_ENOMEM;
    for (int i = 0; i < kN; ++i)
        o->window[i] = 0.54f - 0.46f * cosf(2.0f * (float)3.14159265358979323846 * (float)i / (float)(kN - 1));
    for (int k = 0; k < kN / 2; ++k) {
        const double a = -2.0 * 3.14159265358979323846 * (double)k / (double)kN;
        o->tw[k] = cfl{(float)std::cos(a), (float)std::sin(a)};
    }
    for (int i = 0; i < kN; ++i) {
        int r = 0;
        for (int b = 0; b < 10; ++b) if (i & (1 << b)) r |= 1 << (9 - b);
        o->rev[i] = (uint16_t)r;
    }
    srand((unsigned int)time(nullptr));                              // iq_correct.c:92
    *out = o;
    return IQGPU_OK;
}

extern "C" void iqgpu_iq_optimizer_destroy(iqgpu_iq_optimizer *o) { delete o; }

extern "C" int iqgpu_iq_optimizer_seed(iqgpu_iq_optimizer *o, uint32_t seed)
{
    if (!o) return IQGPU_EINVAL;
    o->seeded = true; o->fn = nullptr;
    o->lcg = seed % 2147483647u; if (o->lcg == 0) o->lcg = 1;
    return IQGPU_OK;
}

extern "C" int iqgpu_iq_optimizer_set_rng(iqgpu_iq_optimizer *o, iqgpu_rand_dir_fn fn, void *user)
{
    if (!o) return IQGPU_EINVAL;
    o->fn = fn; o->fn_user = user;
    return IQGPU_OK;
}

extern "C" int iqgpu_iq_optimizer_set_factors(iqgpu_iq_optimizer *o, float mag, float phase)
{
    if (!o) return IQGPU_EINVAL;
    std::lock_guard<std::mutex> g(o->mu);
    o->mag = mag; o->phase = phase;
    return IQGPU_OK;
}

extern "C" int iqgpu_iq_optimizer_get_factors(iqgpu_iq_optimizer *o, float *mag, float *phase)
{
    if (!o || !mag || !phase) return IQGPU_EINVAL;
    std::lock_guard<std::mutex> g(o->mu);
    *mag = o->mag; *phase = o->phase;
    return IQGPU_OK;
}

extern "C" int iqgpu_iq_optimizer_get_stats(iqgpu_iq_optimizer *o, iqgpu_iq_optimizer_stats *st)
{
    if (!o || !st) return IQGPU_EINVAL;
    *st = o->stats;
    st->average_power_db = o->average_power; st->power_range_db = o->power_range;
    return IQGPU_OK;
}

extern "C" float iqgpu_iq_optimizer_metric(iqgpu_iq_optimizer *o, const float *block_re_im, float mag, float phase)
{
    if (!o || !block_re_im) return 0.0f;
    return o->metric((const cfl *)block_re_im, mag, phase);
}

// iq_correct_run_optimization, src/iq_correct.c:154-219.  now_sec < 0: read CLOCK_MONOTONIC like the reference.
extern "C" int iqgpu_iq_optimizer_run(iqgpu_iq_optimizer *o, const float *block_re_im, double now_sec, int *updated)
{
    if (!o || !block_re_im) return IQGPU_EINVAL;
    if (updated) *updated = 0;
    const cfl *block = (const cfl *)block_re_im;
    const double now = now_sec < 0.0 ? monotonic_now() : now_sec;
    o->stats.calls += 1;
    if ((now - o->last_time) * 1000.0 < kIntervalMs) { o->stats.skipped_interval += 1; return IQGPU_OK; }
    o->estimate_power(block);
    if (o->power_range < kPowerDb) { o->stats.skipped_power += 1; return IQGPU_OK; }
    o->last_time = now;

    float cur_gain, cur_phase, best;
    {
        std::lock_guard<std::mutex> g(o->mu);
        cur_gain = o->mag; cur_phase = o->phase;
        best = o->metric(block, cur_gain, cur_phase);
    }
    o->stats.initial_metric = best;
    for (int i = 0; i < kPasses; ++i) {
        const float cand_gain = cur_gain + kIncrement * o->direction();
        const float cand_phase = cur_phase + kIncrement * o->direction();
        const float m = o->metric(block, cand_gain, cand_phase);
        if (m > best) { best = m; cur_gain = cand_gain; cur_phase = cand_phase; o->stats.accepted += 1; }
    }
    o->stats.final_metric = best;
    {
        std::lock_guard<std::mutex> g(o->mu);
        o->mag = ((1.0f - kSmooth) * o->mag) + (kSmooth * cur_gain);
        o->phase = ((1.0f - kSmooth) * o->phase) + (kSmooth * cur_phase);
    }
    o->stats.runs += 1;
    if (updated) *updated = 1;
    return IQGPU_OK;
}

// iq_correct_run_initial_calibration's tail (iq_correct.c:294-297): after the synchronous first run the
// interval restarts from "now"
extern "C" int iqgpu_iq_optimizer_touch(iqgpu_iq_optimizer *o, double now_sec)
{
    if (!o) return IQGPU_EINVAL;
    o->last_time = now_sec < 0.0 ? monotonic_now() : now_sec;
    return IQGPU_OK;
}
