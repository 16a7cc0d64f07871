/*
 * iq_oracle.c -- CPU restatement of the iq_tool DSP hot path.  TEST INFRASTRUCTURE ONLY
 * (see iq_oracle.h for the rules and the parity-pinning status of every function).
 *
 * Citations "ref:" are into /root/reference.  Citations "liquid:" name the liquid-dsp
 * (github.com/jgaeddert/liquid-dsp, un-pinned by the reference; 1.4 - 1.6 semantics targeted)
 * source file whose published algorithm is restated; liquid-dsp is not present in this image,
 * so those parts are "parity unpinned".
 *
 * Build: strict IEEE, no contraction:  gcc -O2 -std=c99 -ffp-contract=off
 */
#include "iq_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef ORC_ACC
#define ORC_ACC double
#endif
typedef ORC_ACC acc_t;

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------
 * sample_convert                                    ref: src/sample_convert.c:102-309
 * ---------------------------------------------------------------------------------------- */

size_t orc_bytes_per_sample(int fmt) /* ref: sample_convert.c:102-122 */
{
    switch (fmt) {
    case 1: case 2: return 1;            /* U8, S8   (real scalar formats: sized, never converted) */
    case 3: case 4: return 2;            /* U16, S16 */
    case 5: case 6: case 7: return 4;    /* U32, S32, F32 */
    case ORC_FMT_CS8: case ORC_FMT_CU8: return 2;
    case ORC_FMT_CS16: case ORC_FMT_CU16: case ORC_FMT_SC16Q11: return 4;
    case ORC_FMT_CS24: return 6;
    case ORC_FMT_CS32: case ORC_FMT_CU32: case ORC_FMT_CF32: return 8;
    default: return 0;
    }
}

/* normaliser first, then gain: two separate float multiplies (ref: macros at 75-96) */
static inline float unpack_signed(float v, float norm, float gain) { float t = v * norm; return t * gain; }
static inline float unpack_unsigned(float v, float off, float norm, float gain)
{
    float t = v - off; t = t * norm; return t * gain;
}

int orc_convert_block_to_cf32(const void *in, orc_cf32 *out, size_t n, int fmt, float gain)
{
    size_t i;
    switch (fmt) {
    case ORC_FMT_CS8: { /* ref: 136-139 */
        const int8_t *p = (const int8_t *)in;
        for (i = 0; i < n; i++) {
            out[i].re = unpack_signed((float)p[2 * i], 1.0f / 128.0f, gain);
            out[i].im = unpack_signed((float)p[2 * i + 1], 1.0f / 128.0f, gain);
        }
        return 1;
    }
    case ORC_FMT_CU8: { /* ref: 140-143 */
        const uint8_t *p = (const uint8_t *)in;
        for (i = 0; i < n; i++) {
            out[i].re = unpack_unsigned((float)p[2 * i], 127.5f, 1.0f / 128.0f, gain);
            out[i].im = unpack_unsigned((float)p[2 * i + 1], 127.5f, 1.0f / 128.0f, gain);
        }
        return 1;
    }
    case ORC_FMT_CS16: case ORC_FMT_SC16Q11: { /* ref: 144-151 */
        const int16_t *p = (const int16_t *)in;
        const float norm = (fmt == ORC_FMT_CS16) ? 1.0f / 32768.0f : 1.0f / 2048.0f;
        for (i = 0; i < n; i++) {
            out[i].re = unpack_signed((float)p[2 * i], norm, gain);
            out[i].im = unpack_signed((float)p[2 * i + 1], norm, gain);
        }
        return 1;
    }
    case ORC_FMT_CU16: { /* ref: 168-170 */
        const uint16_t *p = (const uint16_t *)in;
        for (i = 0; i < n; i++) {
            out[i].re = unpack_unsigned((float)p[2 * i], 32767.5f, 1.0f / 32768.0f, gain);
            out[i].im = unpack_unsigned((float)p[2 * i + 1], 32767.5f, 1.0f / 32768.0f, gain);
        }
        return 1;
    }
    case ORC_FMT_CS24: { /* ref: 152-167: 3 LE bytes, sign-extended, /2^23 */
        const uint8_t *p = (const uint8_t *)in;
        for (i = 0; i < n; i++, p += 6) {
            int32_t a = (int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24) >> 8;
            int32_t b = (int32_t)((uint32_t)p[3] << 8 | (uint32_t)p[4] << 16 | (uint32_t)p[5] << 24) >> 8;
            out[i].re = unpack_signed((float)a, 1.0f / 8388608.0f, gain);
            out[i].im = unpack_signed((float)b, 1.0f / 8388608.0f, gain);
        }
        return 1;
    }
    case ORC_FMT_CS32: { /* ref: 171-182: double intermediate */
        const int32_t *p = (const int32_t *)in;
        for (i = 0; i < n; i++) {
            double a = (double)p[2 * i] * (1.0 / 2147483648.0);
            double b = (double)p[2 * i + 1] * (1.0 / 2147483648.0);
            out[i].re = (float)(a * (double)gain);
            out[i].im = (float)(b * (double)gain);
        }
        return 1;
    }
    case ORC_FMT_CU32: { /* ref: 183-194 */
        const uint32_t *p = (const uint32_t *)in;
        for (i = 0; i < n; i++) {
            double a = ((double)p[2 * i] - 2147483647.5) * (1.0 / 2147483648.0);
            double b = ((double)p[2 * i + 1] - 2147483647.5) * (1.0 / 2147483648.0);
            out[i].re = (float)(a * (double)gain);
            out[i].im = (float)(b * (double)gain);
        }
        return 1;
    }
    case ORC_FMT_CF32: { /* ref: 195-202 */
        const orc_cf32 *p = (const orc_cf32 *)in;
        for (i = 0; i < n; i++) { out[i].re = p[i].re * gain; out[i].im = p[i].im * gain; }
        return 1;
    }
    default: return 0; /* ref: 203-205 */
    }
}

/* signed: scale, +-0.5 by sign, clamp, truncate (ref: macro 40-57) */
static inline float pack_signed_f(float x, float scale, float lo, float hi)
{
    float v = x * scale;
    v = (v > 0.0f) ? v + 0.5f : v - 0.5f;
    if (v > hi) v = hi;
    if (v < lo) v = lo;
    return v;
}
/* unsigned: scale, offset, clamp, +0.5, truncate (ref: macro 59-73) */
static inline float pack_unsigned_f(float x, float scale, float off, float hi)
{
    float v = (x * scale) + off;
    if (v > hi) v = hi;
    if (v < 0.0f) v = 0.0f;
    return v + 0.5f;
}

int orc_convert_cf32_to_block(const orc_cf32 *in, void *out, size_t n, int fmt)
{
    size_t i;
    switch (fmt) {
    case ORC_FMT_CS8: { /* ref: 219-221 */
        int8_t *o = (int8_t *)out;
        for (i = 0; i < n; i++) {
            o[2 * i] = (int8_t)pack_signed_f(in[i].re, 127.0f, -128.0f, 127.0f);
            o[2 * i + 1] = (int8_t)pack_signed_f(in[i].im, 127.0f, -128.0f, 127.0f);
        }
        return 1;
    }
    case ORC_FMT_CU8: { /* ref: 222-224 */
        uint8_t *o = (uint8_t *)out;
        for (i = 0; i < n; i++) {
            o[2 * i] = (uint8_t)pack_unsigned_f(in[i].re, 127.0f, 127.5f, 255.0f);
            o[2 * i + 1] = (uint8_t)pack_unsigned_f(in[i].im, 127.0f, 127.5f, 255.0f);
        }
        return 1;
    }
    case ORC_FMT_CS16: case ORC_FMT_SC16Q11: { /* ref: 225-230 */
        int16_t *o = (int16_t *)out;
        const float s = (fmt == ORC_FMT_CS16) ? 32767.0f : 2048.0f;
        for (i = 0; i < n; i++) {
            o[2 * i] = (int16_t)pack_signed_f(in[i].re, s, -32768.0f, 32767.0f);
            o[2 * i + 1] = (int16_t)pack_signed_f(in[i].im, s, -32768.0f, 32767.0f);
        }
        return 1;
    }
    case ORC_FMT_CU16: { /* ref: 231-233 */
        uint16_t *o = (uint16_t *)out;
        for (i = 0; i < n; i++) {
            o[2 * i] = (uint16_t)pack_unsigned_f(in[i].re, 32767.0f, 32767.5f, 65535.0f);
            o[2 * i + 1] = (uint16_t)pack_unsigned_f(in[i].im, 32767.0f, 32767.5f, 65535.0f);
        }
        return 1;
    }
    case ORC_FMT_CS24: { /* ref: 234-262: round in float, convert, clamp as integers */
        uint8_t *o = (uint8_t *)out;
        for (i = 0; i < n; i++, o += 6) {
            float a = in[i].re * 8388607.0f, b = in[i].im * 8388607.0f;
            int32_t ia = (int32_t)((a > 0.0f) ? a + 0.5f : a - 0.5f);
            int32_t ib = (int32_t)((b > 0.0f) ? b + 0.5f : b - 0.5f);
            if (ia > 8388607) ia = 8388607;
            if (ia < -8388608) ia = -8388608;
            if (ib > 8388607) ib = 8388607;
            if (ib < -8388608) ib = -8388608;
            o[0] = (uint8_t)(ia & 0xFF); o[1] = (uint8_t)((ia >> 8) & 0xFF); o[2] = (uint8_t)((ia >> 16) & 0xFF);
            o[3] = (uint8_t)(ib & 0xFF); o[4] = (uint8_t)((ib >> 8) & 0xFF); o[5] = (uint8_t)((ib >> 16) & 0xFF);
        }
        return 1;
    }
    case ORC_FMT_CS32: { /* ref: 263-283: double */
        int32_t *o = (int32_t *)out;
        const double hi = (double)INT_MAX, lo = (double)INT_MIN;
        for (i = 0; i < n; i++) {
            double a = (double)in[i].re * hi, b = (double)in[i].im * hi;
            a = (a > 0.0) ? a + 0.5 : a - 0.5;
            b = (b > 0.0) ? b + 0.5 : b - 0.5;
            if (a > hi) a = hi;
            if (a < lo) a = lo;
            if (b > hi) b = hi;
            if (b < lo) b = lo;
            o[2 * i] = (int32_t)a; o[2 * i + 1] = (int32_t)b;
        }
        return 1;
    }
    case ORC_FMT_CU32: { /* ref: 284-300 */
        uint32_t *o = (uint32_t *)out;
        const double hi = (double)UINT_MAX;
        for (i = 0; i < n; i++) {
            double a = ((double)in[i].re * 2147483647.0) + 2147483647.5;
            double b = ((double)in[i].im * 2147483647.0) + 2147483647.5;
            if (a > hi) a = hi;
            if (a < 0.0) a = 0.0;
            if (b > hi) b = hi;
            if (b < 0.0) b = 0.0;
            o[2 * i] = (uint32_t)(a + 0.5); o[2 * i + 1] = (uint32_t)(b + 0.5);
        }
        return 1;
    }
    case ORC_FMT_CF32: /* ref: 301-303 */
        memcpy(out, in, n * sizeof(orc_cf32));
        return 1;
    default: return 0;
    }
}

/* ------------------------------------------------------------------------------------------
 * nco_crcf with LIQUID_NCO                liquid: src/nco/src/nco.proto.c  [parity unpinned]
 *   ref call sites: src/frequency_shift.c:54-60,70-77 (create/set_frequency), 92-94 (mix
 *   block up/down), 105 (set_phase 0); src/filter.c:211-218 (design-time modulation).
 *   uint32 phase accumulator, 1024-entry sine table, index = rounded top 10 bits.
 * ---------------------------------------------------------------------------------------- */
struct orc_nco { uint32_t theta, d_theta; float tab[1024]; };

uint32_t orc_nco_constrain(float theta)
{
    /* liquid: p = theta/(2 pi) (float result of a float*double product); fractional part in
     * [0,1); return (uint32)(fpart * 0xffffffff) -- the int constant converts to float 2^32 */
    float p = (float)((double)theta * 0.159154943091895);
    float fpart = p - (float)((long)p);
    if (fpart < 0.0f) fpart += 1.0f;
    double v = (double)(fpart * 4294967296.0f);
    if (v >= 4294967296.0) v = 0.0; /* fpart rounded up to 1.0: wraps */
    return (uint32_t)v;
}

orc_nco *orc_nco_create(void)
{
    orc_nco *q = (orc_nco *)calloc(1, sizeof(*q));
    unsigned i;
    for (i = 0; i < 1024; i++) {
        /* liquid: sintab[i] = sinf(2.0f*M_PI*(float)i/1024.0f): double argument rounded to float */
        float arg = (float)(2.0 * M_PI * (double)(float)i / 1024.0);
        q->tab[i] = sinf(arg);
    }
    return q;
}
void orc_nco_destroy(orc_nco *q) { free(q); }
void orc_nco_set_frequency(orc_nco *q, float dtheta) { q->d_theta = orc_nco_constrain(dtheta); }
void orc_nco_set_phase(orc_nco *q, float phi) { q->theta = orc_nco_constrain(phi); }
uint32_t orc_nco_get_dtheta_u32(const orc_nco *q) { return q->d_theta; }
uint32_t orc_nco_get_theta_u32(const orc_nco *q) { return q->theta; }
const float *orc_nco_table(const orc_nco *q) { return q->tab; }

static inline unsigned nco_index(uint32_t theta) { return ((theta + (1u << 21)) >> 22) & 0x3ff; }

void orc_nco_cexpf(const orc_nco *q, orc_cf32 *y)
{
    unsigned idx = nco_index(q->theta);
    y->im = q->tab[idx];
    y->re = q->tab[(idx + 256) & 0x3ff];
}
void orc_nco_step(orc_nco *q) { q->theta += q->d_theta; }

void orc_nco_mix_block(orc_nco *q, int up, const orc_cf32 *x, orc_cf32 *y, size_t n)
{
    size_t i;
    for (i = 0; i < n; i++) {
        orc_cf32 v; orc_nco_cexpf(q, &v);
        float s = up ? v.im : -v.im, c = v.re; /* mix_down multiplies by conj(v) */
        float xr = x[i].re, xi = x[i].im;
        y[i].re = (float)((acc_t)xr * c - (acc_t)xi * s);
        y[i].im = (float)((acc_t)xr * s + (acc_t)xi * c);
        q->theta += q->d_theta;
    }
}

/* ------------------------------------------------------------------------------------------
 * iirfilt_crcf_create_dc_blocker           liquid: src/filter/src/iirfilt.proto.c [unpinned]
 *   ref: src/dc_block.c:32 (alpha), 54 (create), 82 (execute_block), 73 (reset)
 *   b = {1,-1}, a = {1, -1+alpha}; direct form II: v0 = x - a1*v1; y = v0 - v1.
 * ---------------------------------------------------------------------------------------- */
struct orc_dcblock { float a1; int literal; double vr, vi; float fvr, fvi; };

orc_dcblock *orc_dcblock_create(float alpha, int f32_literal)
{
    orc_dcblock *q = (orc_dcblock *)calloc(1, sizeof(*q));
    q->a1 = -1.0f + alpha;
    q->literal = f32_literal;
    return q;
}
void orc_dcblock_destroy(orc_dcblock *q) { free(q); }
void orc_dcblock_reset(orc_dcblock *q) { q->vr = q->vi = 0.0; q->fvr = q->fvi = 0.0f; }

void orc_dcblock_apply(orc_dcblock *q, orc_cf32 *buf, size_t n)
{
    size_t i;
    if (q->literal) { /* float recurrence, operation for operation as liquid executes it */
        float a1 = q->a1, vr = q->fvr, vi = q->fvi;
        for (i = 0; i < n; i++) {
            float v0r = buf[i].re - a1 * vr, v0i = buf[i].im - a1 * vi;
            buf[i].re = v0r - vr; buf[i].im = v0i - vi;
            vr = v0r; vi = v0i;
        }
        q->fvr = vr; q->fvi = vi;
    } else { /* canonical oracle: same coefficients (float a1), state in double */
        double c = -(double)q->a1, vr = q->vr, vi = q->vi;
        for (i = 0; i < n; i++) {
            double v0r = (double)buf[i].re + c * vr, v0i = (double)buf[i].im + c * vi;
            buf[i].re = (float)(v0r - vr); buf[i].im = (float)(v0i - vi);
            vr = v0r; vi = v0i;
        }
        q->vr = vr; q->vi = vi;
    }
}

/* ------------------------------------------------------------------------------------------
 * iq_correct apply                          ref: src/iq_correct.c:307-313 (pure reference C)
 * ---------------------------------------------------------------------------------------- */
void orc_iq_correct_apply(orc_cf32 *buf, size_t n, float mag, float phase)
{
    const float magp1 = 1.0f + mag;
    size_t i;
    for (i = 0; i < n; i++) {
        float re = buf[i].re, im = buf[i].im;
        float t = phase * re;
        buf[i].re = re * magp1;
        buf[i].im = im + t;
    }
}

/* ------------------------------------------------------------------------------------------
 * Kaiser design primitives       liquid: src/filter/src/firdes.c, src/math/src/windows.c,
 *                                        src/math/src/math.bessel.c          [parity unpinned]
 * ---------------------------------------------------------------------------------------- */
float orc_kaiser_beta_As(float As)
{
    As = fabsf(As);
    if (As > 50.0f) return 0.1102f * (As - 8.7f);
    if (As > 21.0f) return (float)(0.5842 * pow((double)As - 21.0, 0.4) + 0.07886 * ((double)As - 21.0));
    return 0.0f;
}

double orc_besseli0(double z)
{
    /* I0(z) = sum_k ((z/2)^k / k!)^2 ; liquid sums 32 terms in float via lgamma */
    double t = 1.0, y = 1.0, h = 0.5 * z;
    int k;
    for (k = 1; k < 64; k++) { t *= h / (double)k; y += t * t; if (t * t < 1e-20 * y) break; }
    return y;
}

double orc_kaiser_window(unsigned i, unsigned n, double beta)
{
    /* liquid >= 1.4 liquid_kaiser(): t = i - (n-1)/2, r = 2t/(n-1) */
    double t = (double)i - (double)(n - 1) / 2.0;
    double r = 2.0 * t / (double)(n - 1);
    double a = 1.0 - r * r; if (a < 0.0) a = 0.0;
    return orc_besseli0(beta * sqrt(a)) / orc_besseli0(beta);
}

static double liquid_sinc(double x)
{
    /* liquid sincf(): product-of-cosines approximation below |x| < 0.01 */
    if (fabs(x) < 0.01) return cos(M_PI * x / 2.0) * cos(M_PI * x / 4.0) * cos(M_PI * x / 8.0);
    return sin(M_PI * x) / (M_PI * x);
}

void orc_firdes_kaiser(unsigned n, float fc, float As, float mu, float *h)
{
    double beta = (double)orc_kaiser_beta_As(As);
    unsigned i;
    for (i = 0; i < n; i++) {
        double t = (double)i - (double)(n - 1) / 2.0 + (double)mu;
        double h1 = liquid_sinc(2.0 * (double)fc * t);
        double h2 = orc_kaiser_window(i, n, beta);
        h[i] = (float)(h1 * h2);
    }
}

unsigned orc_estimate_req_filter_len(float df, float As)
{
    /* ref: src/filter.c:192 call site; liquid: (As - 7.95)/(14.26 df) in float, truncated */
    float v = (As - 7.95f) / (14.26f * df);
    return (unsigned)v;
}

/* ------------------------------------------------------------------------------------------
 * msresamp_crcf = msresamp2_crcf (half-band cascade) + resamp_crcf (arbitrary polyphase)
 *   liquid: src/filter/src/{msresamp,msresamp2,resamp2,resamp.fixed,firpfb}.proto.c [unpinned]
 *   ref: src/resampler.c:27 (create, As = 60 dB include/constants.h:137), 51, 45
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    unsigned m;        /* semi-length, prototype has 4m+1 taps */
    float *h;          /* prototype, 4m+1 */
    float *h1;         /* filter branch, 2m taps (odd prototype taps, reversed) */
    orc_cf32 *w0, *w1; /* 2m-sample windows, index 0 = oldest */
} orc_resamp2;

static void resamp2_init(orc_resamp2 *q, unsigned m, float As)
{
    unsigned hl = 4 * m + 1, i, j = 0;
    double beta = (double)orc_kaiser_beta_As(As);
    q->m = m;
    q->h = (float *)calloc(hl, sizeof(float));
    q->h1 = (float *)calloc(2 * m, sizeof(float));
    q->w0 = (orc_cf32 *)calloc(2 * m, sizeof(orc_cf32));
    q->w1 = (orc_cf32 *)calloc(2 * m, sizeof(orc_cf32));
    for (i = 0; i < hl; i++) {
        double t = (double)i - (double)(hl - 1) / 2.0;
        q->h[i] = (float)(liquid_sinc(t / 2.0) * orc_kaiser_window(i, hl, beta));
    }
    for (i = 1; i < hl; i += 2) q->h1[j++] = q->h[hl - i - 1];
}
static void resamp2_free(orc_resamp2 *q) { free(q->h); free(q->h1); free(q->w0); free(q->w1); }
static void resamp2_reset(orc_resamp2 *q)
{
    memset(q->w0, 0, 2 * q->m * sizeof(orc_cf32));
    memset(q->w1, 0, 2 * q->m * sizeof(orc_cf32));
}
static inline void win_push(orc_cf32 *w, unsigned len, orc_cf32 v)
{
    memmove(w, w + 1, (len - 1) * sizeof(orc_cf32));
    w[len - 1] = v;
}
/* x[0], x[1] -> one output; gain 2 per stage (normalised by zeta in the cascade) */
static inline orc_cf32 resamp2_decim(orc_resamp2 *q, const orc_cf32 *x)
{
    unsigned n = 2 * q->m, i;
    acc_t ar = 0, ai = 0;
    orc_cf32 y;
    win_push(q->w1, n, x[0]);
    for (i = 0; i < n; i++) { ar += (acc_t)q->h1[i] * q->w1[i].re; ai += (acc_t)q->h1[i] * q->w1[i].im; }
    win_push(q->w0, n, x[1]);
    ar += (acc_t)q->w0[q->m - 1].re; ai += (acc_t)q->w0[q->m - 1].im;
    y.re = (float)ar; y.im = (float)ai;
    return y;
}
/* one input -> y[0] (delay branch), y[1] (filter branch) */
static inline void resamp2_interp(orc_resamp2 *q, orc_cf32 x, orc_cf32 *y)
{
    unsigned n = 2 * q->m, i;
    acc_t ar = 0, ai = 0;
    win_push(q->w0, n, x);
    y[0] = q->w0[q->m - 1];
    win_push(q->w1, n, x);
    for (i = 0; i < n; i++) { ar += (acc_t)q->h1[i] * q->w1[i].re; ai += (acc_t)q->h1[i] * q->w1[i].im; }
    y[1].re = (float)ar; y[1].im = (float)ai;
}

#define ORC_ARB_M 7
#define ORC_ARB_NPFB 256
#define ORC_ARB_SUB (2 * ORC_ARB_M) /* 14 taps per arm */

struct orc_msresamp {
    float rate, As;
    int interp;
    unsigned S;
    float rate_arb;
    orc_resamp2 *st;   /* design order i = 0..S-1 (liquid msresamp2 index) */
    orc_cf32 *buf0, *buf1, *gbuf;
    unsigned gidx;
    /* arbitrary resampler */
    uint32_t step, phase;
    float *hp;         /* 2*m*npfb scaled prototype taps; arm a tap n = hp[a + 256 n] */
    orc_cf32 w[ORC_ARB_SUB]; /* index 0 = oldest */
};

orc_msresamp *orc_msresamp_create(float r, float As)
{
    orc_msresamp *q;
    unsigned i;
    if (!(r > 0.0f)) return NULL;
    q = (orc_msresamp *)calloc(1, sizeof(*q));
    q->rate = r; q->As = As;
    q->interp = (r > 1.0f);
    q->rate_arb = r; q->S = 0;
    if (q->interp) { while (q->rate_arb > 2.0f) { q->S++; q->rate_arb *= 0.5f; } }
    else           { while (q->rate_arb < 0.5f) { q->S++; q->rate_arb *= 2.0f; } }

    /* liquid msresamp2_create(type, S, fc = 0.4, f0 = 0, As): per-stage parameters */
    q->st = (orc_resamp2 *)calloc(q->S ? q->S : 1, sizeof(orc_resamp2));
    {
        float fc = 0.4f, as = As + 5.0f;
        for (i = 0; i < q->S; i++) {
            float ft; unsigned hl, m;
            fc = (i == 1) ? (float)((0.5 - (double)fc) / 2.0) : 0.5f * fc;
            ft = 2 * (0.25f - fc);
            hl = orc_estimate_req_filter_len(ft, as);
            m = (unsigned)ceilf((float)(hl - 1) / 4.0f);
            if (m < 3) m = 3;
            resamp2_init(&q->st[i], m, as);
        }
    }
    q->buf0 = (orc_cf32 *)calloc((size_t)1 << q->S, sizeof(orc_cf32));
    q->buf1 = (orc_cf32 *)calloc((size_t)1 << q->S, sizeof(orc_cf32));
    q->gbuf = (orc_cf32 *)calloc(((size_t)1 << q->S) + 4, sizeof(orc_cf32));

    /* liquid resamp_crcf_create(rate_arb, m = 7, fc = min(0.515 rate, 0.49), As, npfb = 256) */
    {
        unsigned n = 2 * ORC_ARB_M * ORC_ARB_NPFB + 1;
        float fc = fminf(0.515f * q->rate_arb, 0.49f);
        float *hf = (float *)calloc(n, sizeof(float));
        double sum = 0.0; float g;
        /* resamp.fixed set_rate(): step = round((1<<24)/rate) with the quotient formed in float */
        float quo = 16777216.0f / q->rate_arb;
        q->step = (uint32_t)round((double)quo);
        orc_firdes_kaiser(n, fc / (float)ORC_ARB_NPFB, As, 0.0f, hf);
        for (i = 0; i < n; i++) sum += (double)hf[i];
        g = (float)((double)ORC_ARB_NPFB / sum);
        q->hp = (float *)calloc(n - 1, sizeof(float));
        for (i = 0; i < n - 1; i++) q->hp[i] = hf[i] * g; /* firpfb_create(npfb, h, n-1) */
        free(hf);
    }
    orc_msresamp_reset(q);
    return q;
}

void orc_msresamp_destroy(orc_msresamp *q)
{
    unsigned i;
    if (!q) return;
    for (i = 0; i < q->S; i++) resamp2_free(&q->st[i]);
    free(q->st); free(q->buf0); free(q->buf1); free(q->gbuf); free(q->hp); free(q);
}

void orc_msresamp_reset(orc_msresamp *q)
{
    unsigned i;
    for (i = 0; i < q->S; i++) resamp2_reset(&q->st[i]);
    q->gidx = 0; q->phase = 0;
    memset(q->w, 0, sizeof(q->w));
}

/* resamp.fixed execute(): push, emit while phase < 2^24, arm = phase >> 16 */
static inline unsigned arb_execute(orc_msresamp *q, orc_cf32 x, orc_cf32 *y)
{
    unsigned n = 0, j;
    win_push(q->w, ORC_ARB_SUB, x);
    while (q->phase <= 0x00ffffffu) {
        unsigned arm = q->phase >> 16;
        acc_t ar = 0, ai = 0;
        /* firpfb arm taps are stored reversed against an oldest-first window:
         * y = sum_n hp[arm + 256 n] * x[newest - n] */
        for (j = 0; j < ORC_ARB_SUB; j++) {
            float t = q->hp[arm + ORC_ARB_NPFB * j];
            ar += (acc_t)t * q->w[ORC_ARB_SUB - 1 - j].re;
            ai += (acc_t)t * q->w[ORC_ARB_SUB - 1 - j].im;
        }
        y[n].re = (float)ar; y[n].im = (float)ai; n++;
        q->phase += q->step;
    }
    q->phase -= (1u << 24);
    return n;
}

/* msresamp2 decim_execute(): 2^S inputs -> 1 output, stages run high rate first using the
 * highest design index, output scaled by zeta = 2^-S */
static inline orc_cf32 cascade_decim(orc_msresamp *q, const orc_cf32 *x)
{
    const orc_cf32 *b0 = x;
    orc_cf32 *b1 = q->buf1;
    unsigned s, i;
    float zeta = 1.0f / (float)(1u << q->S);
    orc_cf32 y;
    for (s = 0; s < q->S; s++) {
        unsigned k = 1u << (q->S - s - 1), g = q->S - s - 1;
        for (i = 0; i < k; i++) b1[i] = resamp2_decim(&q->st[g], &b0[2 * i]);
        b0 = (s % 2) == 0 ? q->buf1 : q->buf0;
        b1 = (s % 2) == 0 ? q->buf0 : q->buf1;
    }
    y.re = b0[0].re * zeta; y.im = b0[0].im * zeta;
    return y;
}

static inline void cascade_interp(orc_msresamp *q, orc_cf32 x, orc_cf32 *y)
{
    orc_cf32 *b0 = q->buf0, *b1 = q->buf1;
    unsigned s, i;
    q->buf0[0] = x;
    for (s = 0; s < q->S; s++) {
        unsigned k = 1u << s;
        if (s == q->S - 1) b1 = y;
        for (i = 0; i < k; i++) resamp2_interp(&q->st[s], b0[i], &b1[2 * i]);
        b0 = (s % 2) == 0 ? q->buf1 : q->buf0;
        b1 = (s % 2) == 0 ? q->buf0 : q->buf1;
    }
}

void orc_msresamp_execute(orc_msresamp *q, const orc_cf32 *x, unsigned nx, orc_cf32 *y, unsigned *ny)
{
    unsigned i, n = 0, M = 1u << q->S;
    if (!q->interp) {
        for (i = 0; i < nx; i++) {
            q->gbuf[q->gidx++] = x[i];
            if (q->gidx == M) {
                orc_cf32 hb = (q->S == 0) ? q->gbuf[0] : cascade_decim(q, q->gbuf);
                n += arb_execute(q, hb, &y[n]);
                q->gidx = 0;
            }
        }
    } else {
        orc_cf32 tmp[4];
        for (i = 0; i < nx; i++) {
            unsigned nw = arb_execute(q, x[i], tmp), k;
            for (k = 0; k < nw; k++) {
                if (q->S == 0) y[n] = tmp[k]; else cascade_interp(q, tmp[k], &y[n]);
                n += M;
            }
        }
    }
    *ny = n;
}

int      orc_msresamp_is_interp(const orc_msresamp *q) { return q->interp; }
unsigned orc_msresamp_num_stages(const orc_msresamp *q) { return q->S; }
/* run_order_index 0 = first stage to run when decimating (highest rate) */
unsigned orc_msresamp_stage_m(const orc_msresamp *q, unsigned k) { return q->st[q->S - 1 - k].m; }
const float *orc_msresamp_stage_taps(const orc_msresamp *q, unsigned k) { return q->st[q->S - 1 - k].h; }
float    orc_msresamp_rate_arb(const orc_msresamp *q) { return q->rate_arb; }
uint32_t orc_msresamp_step(const orc_msresamp *q) { return q->step; }
const float *orc_msresamp_arb_proto(const orc_msresamp *q) { return q->hp; }

/* ------------------------------------------------------------------------------------------
 * filter.c: design, placement, FIR and FFT-block application      ref: src/filter.c:43-526
 *   liquid firfilt/fftfilt (src/filter/src/firfilt.proto.c, fftfilt.proto.c) [unpinned]: both are
 *   exact linear convolutions y[n] = sum_k h[k] x[n-k] with zero initial history; the FFT kind
 *   only adds the block-quantised output count of src/filter.c:491-526.
 * ---------------------------------------------------------------------------------------- */
struct orc_filter {
    int post, impl;
    unsigned block, ntaps;
    orc_cf32 *taps;
    orc_cf32 *hist;     /* last ntaps-1 inputs, oldest first */
    orc_cf32 *rem;      /* FFT remainder, < block samples */
    unsigned rem_len;
};

static void invert_spectrum(float *t, unsigned len) /* ref: filter.c:94-99 */
{
    unsigned k;
    for (k = 0; k < len; k++) t[k] = -t[k];
    t[(len - 1) / 2] += 1.0f;
}

orc_filter *orc_filter_create(const orc_filter_cfg *cfg, double input_rate, double target_rate,
                              int no_resample, int *err)
{
    orc_filter *q;
    orc_cf32 *master; int mlen = 1, i;
    int is_complex = 0, by_peak = 0, post = 0;
    double fs;
    *err = 0;
    if (cfg->n_req == 0) return NULL;
    if (no_resample) target_rate = input_rate; /* ref: setup.c:94-101 */

    /* ref: filter.c:43-92 _configure_filter_stage */
    if (!no_resample && target_rate < input_rate) {
        float maxf = 0.0f;
        for (i = 0; i < cfg->n_req; i++) {
            const orc_filter_req *r = &cfg->req[i]; float cur = 0.0f;
            if (r->type == ORC_FILT_LOWPASS || r->type == ORC_FILT_HIGHPASS) cur = fabsf(r->f1_hz);
            else if (r->type == ORC_FILT_PASSBAND || r->type == ORC_FILT_STOPBAND) cur = fabsf(r->f1_hz) + (r->f2_hz / 2.0f);
            if (cur > maxf) maxf = cur;
        }
        if ((double)maxf > target_rate / 2.0) { *err = -1; return NULL; }
        post = 1;
    }
    fs = post ? target_rate : input_rate;

    master = (orc_cf32 *)calloc(1, sizeof(orc_cf32));
    master[0].re = 1.0f;

    for (i = 0; i < cfg->n_req; i++) { /* ref: filter.c:169-256 */
        const orc_filter_req *r = &cfg->req[i];
        unsigned n, k;
        float As = (cfg->attenuation_db > 0.0f) ? cfg->attenuation_db : 60.0f;
        float *rt; orc_cf32 *cur, *nm; int nlen, a, b;
        if (r->type != ORC_FILT_LOWPASS) by_peak = 1;
        if (cfg->filter_taps > 0) n = (unsigned)cfg->filter_taps;
        else {
            float tw, ntw;
            if (cfg->transition_width_hz > 0.0f) tw = cfg->transition_width_hz;
            else {
                float reff = (r->type == ORC_FILT_LOWPASS || r->type == ORC_FILT_HIGHPASS) ? r->f1_hz : r->f2_hz;
                tw = fabsf(reff) * 0.25f;
            }
            if (tw < 1.0f) tw = 1.0f;
            ntw = tw / (float)fs;
            n = orc_estimate_req_filter_len(ntw, As);
            if (n % 2 == 0) n++;
            if (n < 21) n = 21;
        }
        rt = (float *)calloc(n, sizeof(float));
        cur = (orc_cf32 *)calloc(n, sizeof(orc_cf32));
        if (r->type == ORC_FILT_PASSBAND && fabsf(r->f1_hz) > 1e-9f) { /* ref: 205-218 */
            float half_bw = (r->f2_hz / 2.0f) / (float)fs;
            float fcn = r->f1_hz / (float)fs;
            orc_nco *sh = orc_nco_create();
            is_complex = 1;
            orc_firdes_kaiser(n, half_bw, As, 0.0f, rt);
            orc_nco_set_frequency(sh, (float)(2.0f * M_PI * fcn));
            for (k = 0; k < n; k++) {
                orc_cf32 v; orc_nco_cexpf(sh, &v);
                cur[k].re = v.re * rt[k]; cur[k].im = v.im * rt[k];
                orc_nco_step(sh);
            }
            orc_nco_destroy(sh);
        } else { /* ref: 219-247 */
            float fc, bw;
            switch (r->type) {
            case ORC_FILT_LOWPASS:  fc = r->f1_hz / (float)fs; orc_firdes_kaiser(n, fc, As, 0.0f, rt); break;
            case ORC_FILT_HIGHPASS: fc = r->f1_hz / (float)fs; orc_firdes_kaiser(n, fc, As, 0.0f, rt); invert_spectrum(rt, n); break;
            case ORC_FILT_PASSBAND: bw = r->f2_hz / (float)fs; orc_firdes_kaiser(n, bw / 2.0f, As, 0.0f, rt); break;
            case ORC_FILT_STOPBAND: bw = r->f2_hz / (float)fs; orc_firdes_kaiser(n, bw / 2.0f, As, 0.0f, rt); invert_spectrum(rt, n); break;
            default: break;
            }
            for (k = 0; k < n; k++) { cur[k].re = rt[k]; cur[k].im = 0.0f; }
        }
        /* ref: filter.c:114-136 convolve_complex_taps */
        nlen = mlen + (int)n - 1;
        nm = (orc_cf32 *)calloc((size_t)nlen, sizeof(orc_cf32));
        for (a = 0; a < nlen; a++) {
            int js = (a >= mlen) ? (a - mlen + 1) : 0, je = (a < (int)n - 1) ? a : ((int)n - 1);
            double sr = 0, si = 0;
            for (b = js; b <= je; b++) {
                double h1r = master[a - b].re, h1i = master[a - b].im, h2r = cur[b].re, h2i = cur[b].im;
                sr += h1r * h2r - h1i * h2i; si += h1r * h2i + h1i * h2r;
            }
            nm[a].re = (float)sr; nm[a].im = (float)si;
        }
        free(master); free(rt); free(cur);
        master = nm; mlen = nlen;
    }

    if (by_peak || is_complex) { /* ref: filter.c:272-289: peak |H| over 2048 points */
        float max_mag = 0.0f; int p;
        for (p = 0; p < 2048; p++) {
            float f = ((float)p / 2048.0f) - 0.5f;
            double hr = 0, hi = 0; int k;
            for (k = 0; k < mlen; k++) {
                double ph = -2.0 * M_PI * (double)f * (double)k, c = cos(ph), s = sin(ph);
                hr += master[k].re * c - master[k].im * s; hi += master[k].re * s + master[k].im * c;
            }
            { float mag = (float)sqrt(hr * hr + hi * hi); if (mag > max_mag) max_mag = mag; }
        }
        if (max_mag > 1e-9f) for (i = 0; i < mlen; i++) { master[i].re /= max_mag; master[i].im /= max_mag; }
    } else { /* ref: filter.c:290-299: DC gain, double accumulator, float divide */
        double g = 0.0;
        for (i = 0; i < mlen; i++) g += (double)master[i].re;
        if (fabs(g) > 1e-9f) for (i = 0; i < mlen; i++) { master[i].re /= (float)g; master[i].im /= (float)g; }
    }

    q = (orc_filter *)calloc(1, sizeof(*q));
    q->post = post; q->ntaps = (unsigned)mlen; q->taps = master;
    {
        int choice = cfg->impl_request;
        if (choice == ORC_IMPL_AUTO) choice = is_complex ? ORC_IMPL_FFT : ORC_IMPL_FIR; /* ref: 301-312 */
        if (choice == ORC_IMPL_FFT) { /* ref: 314-344 */
            unsigned bs;
            if (cfg->fft_size > 0) {
                bs = (unsigned)cfg->fft_size / 2;
                if (bs < (unsigned)mlen - 1) { *err = -2; orc_filter_destroy(q); return NULL; }
            } else {
                bs = 1;
                while (bs < (unsigned)mlen - 1) bs *= 2;
                if (bs < (unsigned)mlen * 2) bs *= 2;
            }
            q->block = bs;
            q->impl = is_complex ? ORC_FI_FFT_ASYM : ORC_FI_FFT_SYM;
            q->rem = (orc_cf32 *)calloc(bs, sizeof(orc_cf32));
        } else {
            q->impl = is_complex ? ORC_FI_FIR_ASYM : ORC_FI_FIR_SYM;
        }
    }
    if (!is_complex) for (i = 0; i < mlen; i++) q->taps[i].im = 0.0f; /* crcf objects take crealf() */
    q->hist = (orc_cf32 *)calloc(q->ntaps, sizeof(orc_cf32));
    return q;
}

void orc_filter_destroy(orc_filter *q)
{
    if (!q) return;
    free(q->taps); free(q->hist); free(q->rem); free(q);
}
/* ref: filter.c:417-436 -- resets the liquid object only; remainder_len is NOT cleared */
void orc_filter_reset(orc_filter *q) { if (q) memset(q->hist, 0, q->ntaps * sizeof(orc_cf32)); }
int      orc_filter_is_post(const orc_filter *q) { return q->post; }
int      orc_filter_impl(const orc_filter *q) { return q->impl; }
unsigned orc_filter_block_size(const orc_filter *q) { return q->block; }
unsigned orc_filter_ntaps(const orc_filter *q) { return q->ntaps; }
const orc_cf32 *orc_filter_taps(const orc_filter *q) { return q->taps; }

/* y[n] = sum_k h[k] x[n-k]; hist holds the L-1 samples before x[0] */
static void fir_run(orc_filter *q, const orc_cf32 *x, unsigned n, orc_cf32 *y)
{
    unsigned L = q->ntaps, H = L - 1, i, k;
    orc_cf32 *ext = (orc_cf32 *)malloc(((size_t)H + n) * sizeof(orc_cf32));
    memcpy(ext, q->hist, H * sizeof(orc_cf32));
    memcpy(ext + H, x, (size_t)n * sizeof(orc_cf32));
    for (i = 0; i < n; i++) {
        const orc_cf32 *p = ext + H + i; /* x[n] */
        acc_t ar = 0, ai = 0;
        for (k = 0; k < L; k++) {
            acc_t hr = q->taps[k].re, hi = q->taps[k].im, xr = p[-(long)k].re, xi = p[-(long)k].im;
            ar += hr * xr - hi * xi; ai += hr * xi + hi * xr;
        }
        y[i].re = (float)ar; y[i].im = (float)ai;
    }
    memcpy(q->hist, ext + n, H * sizeof(orc_cf32));
    free(ext);
}

unsigned orc_filter_apply(orc_filter *q, const orc_cf32 *in, unsigned n, orc_cf32 *out)
{
    if (n == 0) return 0;
    if (q->impl == ORC_FI_FIR_SYM || q->impl == ORC_FI_FIR_ASYM) { /* ref: filter.c:449-462 */
        fir_run(q, in, n, out);
        return n;
    } else { /* ref: filter.c:491-526: stream = remainder || input, whole blocks only */
        unsigned total = q->rem_len + n, nb = total / q->block, nout = nb * q->block, newrem = total - nout;
        orc_cf32 *s = (orc_cf32 *)malloc((size_t)total * sizeof(orc_cf32));
        memcpy(s, q->rem, q->rem_len * sizeof(orc_cf32));
        memcpy(s + q->rem_len, in, (size_t)n * sizeof(orc_cf32));
        if (nout) fir_run(q, s, nout, out);
        memcpy(q->rem, s + nout, newrem * sizeof(orc_cf32));
        q->rem_len = newrem;
        free(s);
        return nout;
    }
}

/* ------------------------------------------------------------------------------------------
 * Whole chain     ref: src/pre_processor.c:10-61, src/pipeline.c:492-537,
 *                      src/post_processor.c:9-76, src/setup.c:91-122, src/pipeline.c:138-157
 * ---------------------------------------------------------------------------------------- */
#define ORC_CHUNK 16384 /* ref: include/constants.h:123 */

/* ------------------------------------------------------------------------------------------
 * Output AGC, "digital" profile.  ref: src/agc.c (this part of the file is the reference's own
 * arithmetic, not liquid's; the dx / local profiles are liquid agc_crcf and are not restated).
 * Constants: include/constants.h:184-192.
 * ---------------------------------------------------------------------------------------- */
#define ORC_AGC_PEAK_TARGET      0.9f
#define ORC_AGC_LOCK_TIME        2.0f
#define ORC_AGC_HANG_TIME        4.0f
#define ORC_AGC_RECOVERY_RATE    1.0005f
#define ORC_AGC_LOWER_THRESHOLD  0.75f

struct orc_agc {
    float  target;
    double rate;
    int    clock_mode;
    double wall;
    int    profile;           /* ORC_AGC_PROFILE_* */
    float  alpha, y2_prime;   /* agc_crcf: bandwidth, smoothed output energy (its gain lives in `gain`) */
    /* AppResources state, include/app_context.h:227-231 */
    int      locked;
    float    gain, peak_memory;
    uint64_t seen;
    double   last_strong;
};

static double agc_now(const orc_agc *q)
{
    return q->clock_mode == ORC_AGC_CLOCK_WALL ? q->wall : (double)q->seen / q->rate;
}

orc_agc *orc_agc_create_profile(int profile, float target_level_arg, double sample_rate, int clock_mode)
{
    orc_agc *q = (orc_agc *)calloc(1, sizeof(*q));
    q->profile = profile ? profile : ORC_AGC_PROFILE_DIGITAL;
    q->target = (target_level_arg > 0) ? target_level_arg : ORC_AGC_PEAK_TARGET; /* ref: agc.c:111-113 */
    q->rate = sample_rate;      /* config->target_rate, agc.c:146 */
    q->clock_mode = clock_mode;
    /* ref: agc.c:45-57: AGC_LOCAL_BANDWIDTH, or AGC_DX_BANDWIDTH for the dx profile (constants.h:169,175) */
    q->alpha = (q->profile == ORC_AGC_PROFILE_DX) ? 1e-4f : 1e-2f;
    orc_agc_reset(q);
    return q;
}
orc_agc *orc_agc_create(float target_level_arg, double sample_rate, int clock_mode)
{
    return orc_agc_create_profile(ORC_AGC_PROFILE_DIGITAL, target_level_arg, sample_rate, clock_mode);
}
void orc_agc_destroy(orc_agc *q) { free(q); }
void orc_agc_set_wall_time(orc_agc *q, double now_sec) { q->wall = now_sec; }

void orc_agc_reset(orc_agc *q) /* ref: agc.c:27-33, 75, 231-237 */
{
    q->locked = 0; q->seen = 0; q->peak_memory = 0.05f; q->gain = 1.0f;
    q->last_strong = agc_now(q);
    q->y2_prime = 1.0f;         /* agc_crcf_reset + agc_crcf_set_gain(1.0f), ref: agc.c:227-229 */
}

void orc_agc_apply(orc_agc *q, orc_cf32 *x, unsigned n)
{
    unsigned i;
    float peak = 0.0f;
    if (n == 0) return;                                   /* ref: agc.c:86 */
    if (q->profile != ORC_AGC_PROFILE_DIGITAL) {          /* ref: agc.c:92-100 -> agc_crcf_execute_block [liquid-mem] */
        float g = q->gain, y2p = q->y2_prime;
        const float alpha = q->alpha;
        for (i = 0; i < n; i++) {
            const float yr = x[i].re * g, yi = x[i].im * g;
            const float y2 = yr * yr + yi * yi;
            y2p = (float)((1.0 - (double)alpha) * (double)y2p + (double)alpha * (double)y2);
            if (y2p > 1e-6f) g *= expf(-0.5f * alpha * logf(y2p));
            if (g > 1e6f) g = 1e6f;
            x[i].re = yr; x[i].im = yi;                   /* scale = 1 */
        }
        q->gain = g; q->y2_prime = y2p;
        q->seen += n;
        return;
    }
    for (i = 0; i < n; i++) {                             /* ref: agc.c:120-124, 166-170 */
        float mag = hypotf(x[i].re, x[i].im);             /* cabsf */
        if (mag > peak) peak = mag;
    }
    if (!q->locked) {                                     /* scanning, ref: agc.c:117-160 */
        float safe, g;
        double elapsed;
        if (peak > q->peak_memory) q->peak_memory = peak;
        safe = (q->peak_memory < 1e-4f) ? 1e-4f : q->peak_memory;
        g = q->target / safe;
        for (i = 0; i < n; i++) { x[i].re *= g; x[i].im *= g; }
        elapsed = (double)q->seen / q->rate;
        if (elapsed > ORC_AGC_LOCK_TIME) {
            q->locked = 1; q->gain = g;
            q->last_strong = agc_now(q);
        }
    } else {                                              /* locked, ref: agc.c:165-215 */
        float g = q->gain;
        float out_peak = peak * g;
        double now = agc_now(q);
        if (out_peak > 1.0f) {                            /* safety ratchet */
            g = 0.99f / peak;
            q->last_strong = now;
        } else if (out_peak > (q->target * ORC_AGC_LOWER_THRESHOLD)) {
            q->last_strong = now;
        } else if (now - q->last_strong > ORC_AGC_HANG_TIME) {
            g *= ORC_AGC_RECOVERY_RATE;
        }
        q->gain = g;
        for (i = 0; i < n; i++) { x[i].re *= g; x[i].im *= g; }
    }
    q->seen += n;                                         /* ref: agc.c:218 */
}
int      orc_agc_is_locked(const orc_agc *q) { return q->locked; }
float    orc_agc_gain(const orc_agc *q) { return q->gain; }
float    orc_agc_peak_memory(const orc_agc *q) { return q->peak_memory; }
float    orc_agc_y2_prime(const orc_agc *q) { return q->y2_prime; }
uint64_t orc_agc_samples_seen(const orc_agc *q) { return q->seen; }

struct orc_chain {
    orc_agc *agc;
    orc_chain_desc d;
    float ratio;
    orc_dcblock *dc;
    orc_nco *pre_nco, *post_nco;
    orc_msresamp *rs;
    orc_filter *filt;
    float iq_mag, iq_phase;
    size_t cap;
    orc_cf32 *A, *B;
};

orc_chain *orc_chain_create(const orc_chain_desc *d, int *err)
{
    orc_chain *c = (orc_chain *)calloc(1, sizeof(*c));
    double target = d->no_resample ? d->input_rate_hz : d->target_rate_hz;
    *err = 0;
    c->d = *d;
    c->ratio = (float)(target / d->input_rate_hz); /* ref: setup.c:107 */
    if (!isfinite(c->ratio) || c->ratio < 0.001f || c->ratio > 1000.0f) { *err = -10; free(c); return NULL; }
    if (d->in_format < ORC_FMT_CU8 || d->out_format < ORC_FMT_CU8 ||
        !orc_bytes_per_sample(d->in_format) || !orc_bytes_per_sample(d->out_format)) { *err = -11; free(c); return NULL; }
    if (d->dc_block_enable) { /* ref: dc_block.c:32 */
        float alpha = (float)(2.0 * M_PI * 10.0f / d->input_rate_hz);
        c->dc = orc_dcblock_create(alpha, d->dc_f32_literal);
    }
    c->iq_mag = d->iq_mag; c->iq_phase = d->iq_phase;
    if (d->shift_after_resample && fabs(d->shift_hz) < 1e-9) { *err = -12; orc_chain_destroy(c); return NULL; }
    if (fabs(d->shift_hz) >= 1e-9) { /* ref: frequency_shift.c:24-81 */
        double rate = d->shift_after_resample ? target : d->input_rate_hz;
        float w;
        if (fabs(d->shift_hz) > 5.0 * rate) { *err = -13; orc_chain_destroy(c); return NULL; }
        w = (float)(2.0 * M_PI * fabs(d->shift_hz) / rate);
        if (d->shift_after_resample) { c->post_nco = orc_nco_create(); orc_nco_set_frequency(c->post_nco, w); }
        else { c->pre_nco = orc_nco_create(); orc_nco_set_frequency(c->pre_nco, w); }
    }
    if (!d->no_resample) c->rs = orc_msresamp_create(c->ratio, 60.0f);
    if (d->filter.n_req > 0) {
        c->filt = orc_filter_create(&d->filter, d->input_rate_hz, target, d->no_resample, err);
        if (!c->filt) { orc_chain_destroy(c); return NULL; }
    }
    if (d->agc_enable) c->agc = orc_agc_create_profile(d->agc_profile, d->agc_target, target, d->agc_clock); /* ref: pipeline.c:145 */
    c->cap = orc_chain_max_out_frames(c, ORC_CHUNK);
    c->A = (orc_cf32 *)calloc(c->cap, sizeof(orc_cf32));
    c->B = (orc_cf32 *)calloc(c->cap, sizeof(orc_cf32));
    return c;
}

void orc_chain_destroy(orc_chain *c)
{
    if (!c) return;
    orc_dcblock_destroy(c->dc); orc_nco_destroy(c->pre_nco); orc_nco_destroy(c->post_nco);
    orc_msresamp_destroy(c->rs); orc_filter_destroy(c->filt); orc_agc_destroy(c->agc);
    free(c->A); free(c->B); free(c);
}

void orc_chain_reset(orc_chain *c) /* ref: pre_processor.c:57-61, resampler.c:43-47, post_processor.c:72-76 */
{
    if (c->dc) orc_dcblock_reset(c->dc);
    if (c->pre_nco) orc_nco_set_phase(c->pre_nco, 0.0f);
    if (c->post_nco) orc_nco_set_phase(c->post_nco, 0.0f);
    if (c->rs) orc_msresamp_reset(c->rs);
    orc_filter_reset(c->filt);
    if (c->agc) orc_agc_reset(c->agc); /* ref: post_processor.c:75 */
}

void orc_chain_set_iq_factors(orc_chain *c, float mag, float phase) { c->iq_mag = mag; c->iq_phase = phase; }
float orc_chain_ratio(const orc_chain *c) { return c->ratio; }
orc_agc *orc_chain_agc(orc_chain *c) { return c->agc; }

size_t orc_chain_max_out_frames(const orc_chain *c, size_t frames_in)
{
    /* ref: pipeline.c:232-263 capacity rule, generalised from 16384 to frames_in */
    double r = (double)c->ratio; if (r < 1.0) r = 1.0;
    size_t cap = (size_t)ceil((double)frames_in * r) + 128;
    if (cap < frames_in) cap = frames_in;
    if (c->filt && c->filt->block) cap += c->filt->block;
    return cap;
}

size_t orc_chain_process(orc_chain *c, const void *raw_in, size_t frames_in, void *out, orc_cf32 *tap)
{
    const orc_chain_desc *d = &c->d;
    size_t ibps = orc_bytes_per_sample(d->in_format), obps = orc_bytes_per_sample(d->out_format);
    size_t done = 0, total_out = 0;
    while (done < frames_in) {
        size_t n = frames_in - done; unsigned nf, nw;
        orc_cf32 *cur, *other;
        if (n > ORC_CHUNK) n = ORC_CHUNK;
        /* --- pre-processor, all in buffer A (ref: pre_processor.c:10-55) --- */
        orc_convert_block_to_cf32((const char *)raw_in + done * ibps, c->A, n, d->in_format, d->gain);
        if (c->dc) orc_dcblock_apply(c->dc, c->A, n);
        if (d->iq_correct_enable) orc_iq_correct_apply(c->A, n, c->iq_mag, c->iq_phase);
        if (c->pre_nco) orc_nco_mix_block(c->pre_nco, d->shift_hz >= 0, c->A, c->A, n);
        nf = (unsigned)n;
        if (c->filt && !c->filt->post) nf = orc_filter_apply(c->filt, c->A, nf, c->A);
        done += n;
        if (nf == 0) continue; /* ref: pipeline.c:479-486 */
        /* --- resampler: A -> B (ref: pipeline.c:511-528) --- */
        if (c->rs) orc_msresamp_execute(c->rs, c->A, nf, c->B, &nw);
        else { memcpy(c->B, c->A, (size_t)nf * sizeof(orc_cf32)); nw = nf; }
        cur = c->B; other = c->A;
        if (nw == 0) continue; /* ref: post_processor.c:12 */
        /* --- post-processor (ref: post_processor.c:9-70) --- */
        if (c->filt && c->filt->post) {
            int fft = (c->filt->impl == ORC_FI_FFT_SYM || c->filt->impl == ORC_FI_FFT_ASYM);
            if (fft) { nw = orc_filter_apply(c->filt, cur, nw, other); { orc_cf32 *t = cur; cur = other; other = t; } }
            else nw = orc_filter_apply(c->filt, cur, nw, cur);
        }
        if (c->post_nco && nw) { orc_nco_mix_block(c->post_nco, d->shift_hz >= 0, cur, other, nw); { orc_cf32 *t = cur; cur = other; other = t; } }
        if (nw && c->agc) orc_agc_apply(c->agc, cur, nw); /* ref: post_processor.c:55-57 */
        if (nw) {
            orc_convert_cf32_to_block(cur, (char *)out + total_out * obps, nw, d->out_format);
            if (tap) memcpy(tap + total_out, cur, (size_t)nw * sizeof(orc_cf32));
            total_out += nw;
        }
    }
    return total_out;
}

/* ------------------------------------------------------------------------------------------
 * Three stage threads over chunk trays, ref: src/pipeline.c:96-116 (thread creation), 436-490
 * (pre-processor loop), 492-537 (resampler loop), 539-595 (post-processor loop); queue.c.
 * ---------------------------------------------------------------------------------------- */
#include <pthread.h>

#define ORC_TRAYS 8
typedef struct {
    orc_cf32 *A, *B;
    const char *raw; size_t n;       /* input frames of this chunk */
    unsigned nf, nw;                 /* frames after pre / after resampler */
    int last;
} orc_tray;
typedef struct {
    orc_tray *slot[ORC_TRAYS + 1]; int head, tail;
    pthread_mutex_t mu; pthread_cond_t cv;
} orc_queue;
static void q_init(orc_queue *q) { q->head = q->tail = 0; pthread_mutex_init(&q->mu, NULL); pthread_cond_init(&q->cv, NULL); }
static void q_put(orc_queue *q, orc_tray *t)
{
    pthread_mutex_lock(&q->mu);
    q->slot[q->tail] = t; q->tail = (q->tail + 1) % (ORC_TRAYS + 1);
    pthread_cond_signal(&q->cv);
    pthread_mutex_unlock(&q->mu);
}
static orc_tray *q_get(orc_queue *q)
{
    orc_tray *t;
    pthread_mutex_lock(&q->mu);
    while (q->head == q->tail) pthread_cond_wait(&q->cv, &q->mu);
    t = q->slot[q->head]; q->head = (q->head + 1) % (ORC_TRAYS + 1);
    pthread_mutex_unlock(&q->mu);
    return t;
}
typedef struct {
    orc_chain *c; const char *raw; size_t frames; char *out; size_t total_out;
    orc_queue free_q, pre_q, rs_q;
} orc_pipe;

static void *pre_thread(void *arg)
{
    orc_pipe *p = (orc_pipe *)arg; orc_chain *c = p->c; const orc_chain_desc *d = &c->d;
    size_t ibps = orc_bytes_per_sample(d->in_format), done = 0;
    for (;;) {
        orc_tray *t = q_get(&p->free_q);
        size_t n = p->frames - done; if (n > ORC_CHUNK) n = ORC_CHUNK;
        t->last = (n == 0); t->n = n; t->nf = 0;
        if (n) {
            orc_convert_block_to_cf32(p->raw + done * ibps, t->A, n, d->in_format, d->gain);
            if (c->dc) orc_dcblock_apply(c->dc, t->A, n);
            if (d->iq_correct_enable) orc_iq_correct_apply(t->A, n, c->iq_mag, c->iq_phase);
            if (c->pre_nco) orc_nco_mix_block(c->pre_nco, d->shift_hz >= 0, t->A, t->A, n);
            t->nf = (unsigned)n;
            if (c->filt && !c->filt->post) t->nf = orc_filter_apply(c->filt, t->A, t->nf, t->A);
            done += n;
        }
        {   /* the tray may be recycled the moment it is handed on: decide first */
            const int last = t->last;
            q_put(&p->pre_q, t);
            if (last) return NULL;
        }
    }
}
static void *rs_thread(void *arg)
{
    orc_pipe *p = (orc_pipe *)arg; orc_chain *c = p->c;
    for (;;) {
        orc_tray *t = q_get(&p->pre_q);
        t->nw = 0;
        if (!t->last && t->nf) {
            if (c->rs) orc_msresamp_execute(c->rs, t->A, t->nf, t->B, &t->nw);
            else { memcpy(t->B, t->A, (size_t)t->nf * sizeof(orc_cf32)); t->nw = t->nf; }
        }
        {
            const int last = t->last;
            q_put(&p->rs_q, t);
            if (last) return NULL;
        }
    }
}
static void *post_thread(void *arg)
{
    orc_pipe *p = (orc_pipe *)arg; orc_chain *c = p->c; const orc_chain_desc *d = &c->d;
    size_t obps = orc_bytes_per_sample(d->out_format);
    for (;;) {
        orc_tray *t = q_get(&p->rs_q);
        int last = t->last;
        if (!last && t->nw) {
            orc_cf32 *cur = t->B, *other = t->A; unsigned nw = t->nw;
            if (c->filt && c->filt->post) {
                int fft = (c->filt->impl == ORC_FI_FFT_SYM || c->filt->impl == ORC_FI_FFT_ASYM);
                if (fft) { nw = orc_filter_apply(c->filt, cur, nw, other); { orc_cf32 *x = cur; cur = other; other = x; } }
                else nw = orc_filter_apply(c->filt, cur, nw, cur);
            }
            if (c->post_nco && nw) { orc_nco_mix_block(c->post_nco, d->shift_hz >= 0, cur, other, nw); { orc_cf32 *x = cur; cur = other; other = x; } }
            if (nw && c->agc) orc_agc_apply(c->agc, cur, nw);
            if (nw) { orc_convert_cf32_to_block(cur, p->out + p->total_out * obps, nw, d->out_format); p->total_out += nw; }
        }
        q_put(&p->free_q, t);
        if (last) return NULL;
    }
}

size_t orc_chain_process_pipelined(orc_chain *c, const void *raw_in, size_t frames_in, void *out)
{
    orc_pipe p; orc_tray trays[ORC_TRAYS]; pthread_t th[3]; int i;
    memset(&p, 0, sizeof(p));
    p.c = c; p.raw = (const char *)raw_in; p.frames = frames_in; p.out = (char *)out;
    q_init(&p.free_q); q_init(&p.pre_q); q_init(&p.rs_q);
    for (i = 0; i < ORC_TRAYS; i++) {
        trays[i].A = (orc_cf32 *)calloc(c->cap, sizeof(orc_cf32));
        trays[i].B = (orc_cf32 *)calloc(c->cap, sizeof(orc_cf32));
        q_put(&p.free_q, &trays[i]);
    }
    pthread_create(&th[0], NULL, pre_thread, &p);
    pthread_create(&th[1], NULL, rs_thread, &p);
    pthread_create(&th[2], NULL, post_thread, &p);
    for (i = 0; i < 3; i++) pthread_join(th[i], NULL);
    for (i = 0; i < ORC_TRAYS; i++) { free(trays[i].A); free(trays[i].B); }
    return p.total_out;
}


/* ================================================================================================
 * I/Q imbalance optimiser -- restates iq_correct_run_optimization and its helpers
 * (src/iq_correct.c:154-219, 307-393; constants include/constants.h:157-162).  The spectrum is an exact DFT
 * in double (the reference uses liquid's FFT plan, src/iq_correct.c:116, 326: any exact transform to float
 * rounding), everything the reference computes in float is float here.  The random direction source is the
 * minstd generator the product uses when seeded (the reference's rand() seeded by time() is irreproducible).
 * ================================================================================================ */
#define ORC_IQ_N 1024
struct orc_iqopt {
    float window[ORC_IQ_N];
    double ct[ORC_IQ_N], st[ORC_IQ_N];
    float spectrum[ORC_IQ_N];
    float mag, phase, average_power, power_range;
    double last_time;
    uint32_t lcg;
};

orc_iqopt *orc_iqopt_create(void)
{
    orc_iqopt *q = (orc_iqopt *)calloc(1, sizeof(*q));
    int i;
    if (!q) return NULL;
    for (i = 0; i < ORC_IQ_N; i++) {   /* src/iq_correct.c:122-124 */
        q->window[i] = 0.54f - 0.46f * cosf(2.0f * (float)M_PI * (float)i / (float)(ORC_IQ_N - 1));
        q->ct[i] = cos(2.0 * M_PI * (double)i / ORC_IQ_N);
        q->st[i] = sin(2.0 * M_PI * (double)i / ORC_IQ_N);
    }
    q->lcg = 1;
    return q;
}
void orc_iqopt_destroy(orc_iqopt *q) { free(q); }
void orc_iqopt_seed(orc_iqopt *q, uint32_t seed) { q->lcg = seed % 2147483647u; if (!q->lcg) q->lcg = 1; }
void orc_iqopt_set_factors(orc_iqopt *q, float mag, float phase) { q->mag = mag; q->phase = phase; }
void orc_iqopt_get_factors(const orc_iqopt *q, float *mag, float *phase) { *mag = q->mag; *phase = q->phase; }
float orc_iqopt_power_range(const orc_iqopt *q) { return q->power_range; }

static float iqopt_direction(orc_iqopt *q)   /* _get_random_direction, src/iq_correct.c:391-393, on minstd */
{
    q->lcg = (uint32_t)(((uint64_t)q->lcg * 48271u) % 2147483647u);
    return q->lcg > 2147483647u / 2u ? 1.0f : -1.0f;
}

/* _calculate_power_spectrum, src/iq_correct.c:315-337 */
static void iqopt_spectrum(orc_iqopt *q, const orc_cf32 *block, float gain_adj, float phase_adj)
{
    static float wr[ORC_IQ_N], wi[ORC_IQ_N];
    const float magp1 = 1.0f + gain_adj;
    int i, k;
    for (i = 0; i < ORC_IQ_N; i++) {
        const float re = block[i].re * magp1;                      /* _apply_correction_to_buffer, 307-313 */
        const float im = block[i].im + phase_adj * block[i].re;
        wr[i] = re * q->window[i]; wi[i] = im * q->window[i];
    }
    for (k = 0; k < ORC_IQ_N; k++) {
        /* forward transform, bin k; the shifted spectrum puts bin k at (k + N/2) mod N */
        double sr = 0.0, si = 0.0;
        unsigned idx = 0;
        float m;
        for (i = 0; i < ORC_IQ_N; i++) {
            const double c = q->ct[idx], s = -q->st[idx];
            sr += (double)wr[i] * c - (double)wi[i] * s;
            si += (double)wr[i] * s + (double)wi[i] * c;
            idx = (idx + (unsigned)k) & (ORC_IQ_N - 1);
        }
        m = hypotf((float)sr, (float)si);
        m /= (float)ORC_IQ_N;
        q->spectrum[(k + ORC_IQ_N / 2) & (ORC_IQ_N - 1)] = 20.0f * log10f(m + 1e-12f);
    }
}

/* _calculate_imbalance_metric, src/iq_correct.c:339-360 */
float orc_iqopt_metric(orc_iqopt *q, const orc_cf32 *block, float gain_adj, float phase_adj)
{
    const int half = ORC_IQ_N / 2;
    const int lo = (int)(0.05f * half), hi = (int)(0.95f * half);
    float total = 0.0f;
    int i;
    iqopt_spectrum(q, block, gain_adj, phase_adj);
    for (i = lo; i < hi; i++) {
        const float p_neg = q->spectrum[i], p_pos = q->spectrum[ORC_IQ_N - 1 - i];
        if (p_pos > -80.0f || p_neg > -80.0f) { const float d = p_pos - p_neg; total += d * d; }
    }
    return total;
}

/* iq_correct_run_optimization, src/iq_correct.c:154-219; returns 1 when the factors were updated */
int orc_iqopt_run(orc_iqopt *q, const orc_cf32 *block, double now_sec)
{
    const int half = ORC_IQ_N / 2;
    const int lo = (int)(0.05f * half), hi = (int)(0.95f * half);
    float max_power = -1000.0f, cur_gain, cur_phase, best;
    double sum = 0.0;
    int count = 0, i;
    if ((now_sec - q->last_time) * 1000.0 < 500.0) return 0;       /* IQ_CORRECTION_INTERVAL_MS */
    iqopt_spectrum(q, block, 0.0f, 0.0f);                          /* _estimate_power, 362-389 */
    for (i = lo; i < hi; i++) {
        const float p_neg = q->spectrum[i], p_pos = q->spectrum[ORC_IQ_N - 1 - i];
        if (p_pos > max_power) max_power = p_pos;
        if (p_neg > max_power) max_power = p_neg;
        sum += p_pos + p_neg; count += 2;
    }
    q->average_power = (float)(sum / count);
    q->power_range = max_power - q->average_power;
    if (q->power_range < 20.0f) return 0;                          /* IQ_CORRECTION_POWER_THRESHOLD_DB */
    q->last_time = now_sec;
    cur_gain = q->mag; cur_phase = q->phase;
    best = orc_iqopt_metric(q, block, cur_gain, cur_phase);
    for (i = 0; i < 25; i++) {                                     /* IQ_MAX_PASSES, IQ_BASE_INCREMENT */
        const float cand_gain = cur_gain + 0.0001f * iqopt_direction(q);
        const float cand_phase = cur_phase + 0.0001f * iqopt_direction(q);
        const float m = orc_iqopt_metric(q, block, cand_gain, cand_phase);
        if (m > best) { best = m; cur_gain = cand_gain; cur_phase = cand_phase; }
    }
    q->mag = ((1.0f - 0.05f) * q->mag) + (0.05f * cur_gain);       /* IQ_CORRECTION_SMOOTHING_FACTOR */
    q->phase = ((1.0f - 0.05f) * q->phase) + (0.05f * cur_phase);
    return 1;
}
