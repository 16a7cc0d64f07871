/*
 * iq_oracle.h -- CPU restatement of the iq_tool pre_processor -> resampler -> post_processor
 * hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is the parity oracle for the HIP kernels in iq_tool_amd/csrc.  Nothing in the
 * product path (libiqgpu and the iq_tool_amd python package) may include, link or call it; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY STATUS
 *   pinned    : orc_convert_block_to_cf32 / orc_convert_cf32_to_block / orc_bytes_per_sample are
 *               checked bit-for-bit against the reference's own src/sample_convert.c compiled
 *               into oracle/_ref/ (see oracle/Makefile) and against tests/golden/ vectors
 *               generated from that build.
 *   unpinned  : every operator that the reference delegates to liquid-dsp (nco_crcf, iirfilt,
 *               firfilt, fftfilt, msresamp_crcf, liquid_firdes_kaiser ...).  liquid-dsp is an
 *               un-vendored, un-pinned system dependency of the reference
 *               (/root/reference/CMakeLists.txt:184-240) that is absent from this image, and the
 *               reference has no tests or golden vectors.  Those operators restate liquid-dsp's
 *               published algorithm (target: liquid-dsp 1.4 .. 1.6 semantics, see DESIGN.md SPEC)
 *               and are cross-checked against independent numpy/scipy formulations in tests/.
 *               "parity unpinned" for them.
 *
 * Numerical contract: dot products and recurrences accumulate in ORC_ACC (double by default) and
 * round once to float per operator output.  Building with -DORC_ACC=float gives the
 * float-accumulator variant that bench.py times as the CPU baseline.
 */
#ifndef IQ_ORACLE_H_
#define IQ_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } orc_cf32;

/* format ids == reference format_t values (include/common_types.h:33-37) */
enum {
    ORC_FMT_CU8 = 8, ORC_FMT_CS8 = 9, ORC_FMT_CU16 = 10, ORC_FMT_CS16 = 11, ORC_FMT_CS24 = 12,
    ORC_FMT_CU32 = 13, ORC_FMT_CS32 = 14, ORC_FMT_CF32 = 15, ORC_FMT_SC16Q11 = 16
};
/* FilterType (include/common_types.h:45-51) */
enum { ORC_FILT_NONE = 0, ORC_FILT_LOWPASS = 1, ORC_FILT_HIGHPASS = 2, ORC_FILT_PASSBAND = 3, ORC_FILT_STOPBAND = 4 };
/* FilterTypeRequest (include/common_types.h:61-65) */
enum { ORC_IMPL_AUTO = 0, ORC_IMPL_FIR = 1, ORC_IMPL_FFT = 2 };
/* FilterImplementationType (include/common_types.h:53-59) */
enum { ORC_FI_NONE = 0, ORC_FI_FIR_SYM = 1, ORC_FI_FIR_ASYM = 2, ORC_FI_FFT_SYM = 3, ORC_FI_FFT_ASYM = 4 };

/* ---- sample_convert (src/sample_convert.c) ---- */
size_t orc_bytes_per_sample(int fmt);
int    orc_convert_block_to_cf32(const void *in, orc_cf32 *out, size_t n, int fmt, float gain);
int    orc_convert_cf32_to_block(const orc_cf32 *in, void *out, size_t n, int fmt);

/* ---- nco_crcf, LIQUID_NCO (src/frequency_shift.c:54-107) ---- */
typedef struct orc_nco orc_nco;
uint32_t orc_nco_constrain(float theta);
orc_nco *orc_nco_create(void);
void     orc_nco_destroy(orc_nco *q);
void     orc_nco_set_frequency(orc_nco *q, float dtheta);
void     orc_nco_set_phase(orc_nco *q, float phi);
uint32_t orc_nco_get_dtheta_u32(const orc_nco *q);
uint32_t orc_nco_get_theta_u32(const orc_nco *q);
const float *orc_nco_table(const orc_nco *q);            /* 1024 floats */
void     orc_nco_cexpf(const orc_nco *q, orc_cf32 *y);
void     orc_nco_step(orc_nco *q);
void     orc_nco_mix_block(orc_nco *q, int up, const orc_cf32 *x, orc_cf32 *y, size_t n);

/* ---- iirfilt_crcf dc blocker (src/dc_block.c) ---- */
typedef struct orc_dcblock orc_dcblock;
orc_dcblock *orc_dcblock_create(float alpha, int f32_literal);
void orc_dcblock_destroy(orc_dcblock *q);
void orc_dcblock_reset(orc_dcblock *q);
void orc_dcblock_apply(orc_dcblock *q, orc_cf32 *buf, size_t n);

/* ---- iq_correct apply (src/iq_correct.c:307-313) ---- */
void orc_iq_correct_apply(orc_cf32 *buf, size_t n, float mag, float phase);

/* ---- liquid filter design primitives ---- */
float    orc_kaiser_beta_As(float As);
double   orc_besseli0(double z);
double   orc_kaiser_window(unsigned i, unsigned n, double beta);
void     orc_firdes_kaiser(unsigned n, float fc, float As, float mu, float *h);
unsigned orc_estimate_req_filter_len(float df, float As);

/* ---- msresamp_crcf (src/resampler.c) ---- */
typedef struct orc_msresamp orc_msresamp;
orc_msresamp *orc_msresamp_create(float r, float As);
void orc_msresamp_destroy(orc_msresamp *q);
void orc_msresamp_reset(orc_msresamp *q);
void orc_msresamp_execute(orc_msresamp *q, const orc_cf32 *x, unsigned nx, orc_cf32 *y, unsigned *ny);
/* introspection for tests */
int      orc_msresamp_is_interp(const orc_msresamp *q);
unsigned orc_msresamp_num_stages(const orc_msresamp *q);
unsigned orc_msresamp_stage_m(const orc_msresamp *q, unsigned run_order_index);
const float *orc_msresamp_stage_taps(const orc_msresamp *q, unsigned run_order_index); /* 4m+1 prototype */
float    orc_msresamp_rate_arb(const orc_msresamp *q);
uint32_t orc_msresamp_step(const orc_msresamp *q);
const float *orc_msresamp_arb_proto(const orc_msresamp *q);  /* 2*7*256 used taps, scaled */

/* ---- filter.c ---- */
typedef struct { int type; float f1_hz, f2_hz; } orc_filter_req;
typedef struct {
    int n_req; orc_filter_req req[5];
    float transition_width_hz, attenuation_db;
    int filter_taps;      /* 0 = auto */
    int impl_request;     /* ORC_IMPL_* */
    int fft_size;         /* 0 = auto */
} orc_filter_cfg;
typedef struct orc_filter orc_filter;
/* err: 0 ok, <0 failure (reference would log_fatal) */
orc_filter *orc_filter_create(const orc_filter_cfg *cfg, double input_rate, double target_rate,
                              int no_resample, int *err);
void     orc_filter_destroy(orc_filter *q);
void     orc_filter_reset(orc_filter *q);
int      orc_filter_is_post(const orc_filter *q);
int      orc_filter_impl(const orc_filter *q);       /* ORC_FI_* */
unsigned orc_filter_block_size(const orc_filter *q);
unsigned orc_filter_ntaps(const orc_filter *q);
const orc_cf32 *orc_filter_taps(const orc_filter *q);
/* FIR: returns n; FFT: block-quantised count (src/filter.c:491-526). out may alias in. */
unsigned orc_filter_apply(orc_filter *q, const orc_cf32 *in, unsigned n, orc_cf32 *out);

/* ---- whole chain (pre_processor.c, pipeline.c:492-537, post_processor.c) ---- */
typedef struct {
    int    in_format, out_format;
    double input_rate_hz, target_rate_hz;
    float  gain;
    double shift_hz; int shift_after_resample;
    int    dc_block_enable;
    int    iq_correct_enable; float iq_mag, iq_phase;
    int    no_resample;
    orc_filter_cfg filter;
    int    dc_f32_literal;    /* 1: float recurrence exactly as liquid runs it */
    int    agc_enable;        /* config->output_agc.enable with profile "digital" (agc.c:105-222) */
    float  agc_target;        /* config->output_agc.target_level_arg; <= 0 -> AGC_DIGITAL_PEAK_TARGET */
    int    agc_clock;         /* ORC_AGC_CLOCK_* */
    int    agc_profile;       /* ORC_AGC_PROFILE_*; 0 = digital (fixtures written before the RMS profiles existed) */
} orc_chain_desc;
/* ---- output AGC, "digital" profile (ref: src/agc.c:21-83 create, 85-222 apply, 224-238 reset) ----
 * The reference reads get_monotonic_time_sec() for the hang / creep logic of its locked phase
 * (agc.c:176, 202); the clock is a parameter here: SAMPLES = time of the output stream
 * (samples_seen / sample_rate, deterministic, equal to the wall clock when the reference runs
 * in real time), WALL = the value handed to orc_agc_set_wall_time(). */
enum { ORC_AGC_CLOCK_SAMPLES = 0, ORC_AGC_CLOCK_WALL = 1 };
/* AgcProfile, include/common_types.h:77-82.  DX / LOCAL are liquid's agc_crcf (ref: src/agc.c:39-62 create,
 * 93-100 apply, 227-229 reset) with bandwidth AGC_DX_BANDWIDTH 1e-4 / AGC_LOCAL_BANDWIDTH 1e-2 (constants.h:169,175).
 * [liquid-mem] agc_crcf_execute per sample, liquid-dsp >= 1.3.2 (src/agc/src/agc.proto.c):
 *     y = x * g;  y2 = re(y)^2 + im(y)^2;  y2_prime = (1.0 - alpha) * y2_prime + alpha * y2   (in double, stored as float)
 *     if (y2_prime > 1e-6f) g *= expf(-0.5f * alpha * logf(y2_prime));  if (g > 1e6f) g = 1e6f;  y *= scale (= 1)
 * create / reset: g = 1, y2_prime = 1, alpha = bandwidth.  The reference calls agc_crcf_set_signal_level(target)
 * -- which sets g = 1 / target and y2_prime = 1 -- and then agc_crcf_set_gain(1.0f), so the target level never
 * reaches the loop: the output settles at unit mean power whatever --output-agc-target says (VERIFY). */
enum { ORC_AGC_PROFILE_DX = 1, ORC_AGC_PROFILE_LOCAL = 2, ORC_AGC_PROFILE_DIGITAL = 3 };
typedef struct orc_agc orc_agc;
orc_agc *orc_agc_create(float target_level_arg, double sample_rate, int clock_mode);   /* digital */
orc_agc *orc_agc_create_profile(int profile, float target_level_arg, double sample_rate, int clock_mode);
float    orc_agc_y2_prime(const orc_agc *q);    /* RMS profiles: the smoothed output energy */
void     orc_agc_destroy(orc_agc *q);
void     orc_agc_reset(orc_agc *q);
void     orc_agc_set_wall_time(orc_agc *q, double now_sec);
void     orc_agc_apply(orc_agc *q, orc_cf32 *x, unsigned n);   /* one reference chunk, in place */
int      orc_agc_is_locked(const orc_agc *q);
float    orc_agc_gain(const orc_agc *q);
float    orc_agc_peak_memory(const orc_agc *q);
uint64_t orc_agc_samples_seen(const orc_agc *q);

typedef struct orc_chain orc_chain;
orc_chain *orc_chain_create(const orc_chain_desc *d, int *err);
void   orc_chain_destroy(orc_chain *c);
void   orc_chain_reset(orc_chain *c);
void   orc_chain_set_iq_factors(orc_chain *c, float mag, float phase);
float  orc_chain_ratio(const orc_chain *c);
orc_agc *orc_chain_agc(orc_chain *c);        /* NULL when the AGC is off */
size_t orc_chain_max_out_frames(const orc_chain *c, size_t frames_in);
/* Processes frames_in frames in reference-sized chunks (16384).  Returns frames written.
 * If cf32_tap != NULL the cf32 samples entering convert_cf32_to_block are also stored there. */
size_t orc_chain_process(orc_chain *c, const void *raw_in, size_t frames_in, void *out,
                         orc_cf32 *cf32_tap);

/* The same stream through THREE concurrent stage threads (pre-processor, resampler, post-processor)
 * handing 16384-frame chunks over bounded queues, as the reference runs it (src/pipeline.c:96-116,
 * 436-595).  Same results as orc_chain_process; exists for bench.py's cpu_baseline ("cores": 3). */
size_t orc_chain_process_pipelined(orc_chain *c, const void *raw_in, size_t frames_in, void *out);

/* ---- I/Q imbalance optimiser (src/iq_correct.c:154-219, 307-393); restated from the reference's own source ---- */
typedef struct orc_iqopt orc_iqopt;
orc_iqopt *orc_iqopt_create(void);
void   orc_iqopt_destroy(orc_iqopt *q);
void   orc_iqopt_seed(orc_iqopt *q, uint32_t seed);
void   orc_iqopt_set_factors(orc_iqopt *q, float mag, float phase);
void   orc_iqopt_get_factors(const orc_iqopt *q, float *mag, float *phase);
float  orc_iqopt_power_range(const orc_iqopt *q);
float  orc_iqopt_metric(orc_iqopt *q, const orc_cf32 *block_1024, float gain_adj, float phase_adj);
int    orc_iqopt_run(orc_iqopt *q, const orc_cf32 *block_1024, double now_sec);

#ifdef __cplusplus
}
#endif
#endif
