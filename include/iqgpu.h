/*
 * iqgpu.h -- C ABI of libiqgpu: the MI355X (gfx950) replacement for iq_tool's
 * pre_processor -> resampler -> post_processor sample path.
 *
 * One `iqgpu_chain` handle stands in for the three DSP stage threads of the reference
 * (/root/reference/src/pipeline.c:436-595) for ONE stream.  It is what a single "GPU stage"
 * thread sitting between reader_output_queue and the writer would call; see INTEGRATION.md for
 * the binding a maintainer adds on the reference side.
 *
 * Conventions (they mirror the reference's, src/pipeline.c / include/resampler.h):
 *   - the caller owns every host buffer; handles are opaque; one thread per handle; handles are
 *     independent (one per GPU shard);
 *   - the stream is continuous across calls: any split of the input into calls produces the
 *     same output bytes (all in-scope operators are chunk-size invariant).  The
 *     one exception is the reference's own: the "digital" output AGC works chunk by chunk, so with
 *     that profile every call is cut into agc_chunk_frames-sized chunks from its first frame
 *     (the "dx" / "local" profiles are per-sample loops: any split gives the same bytes);
 *   - nothing is flushed at end of stream (resampler / FIR tails and the FFT-filter remainder
 *     are dropped exactly as the reference drops them, src/filter.c:521-525);
 *   - frames_out may be 0 (resampler group buffering, FFT block quantisation);
 *   - every entry point returns 0 on success or a negative IQGPU_E* code, with a message
 *     available from iqgpu_last_error().  There is no CPU fallback: without a usable HIP
 *     device every compute entry point fails with IQGPU_ENODEV.
 */
#ifndef IQGPU_H_
#define IQGPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IQGPU_ABI_VERSION 6

/* Sample formats: numerically equal to the reference's format_t (include/common_types.h:33-37) */
enum {
    IQGPU_FMT_CU8 = 8, IQGPU_FMT_CS8 = 9, IQGPU_FMT_CU16 = 10, IQGPU_FMT_CS16 = 11,
    IQGPU_FMT_CS24 = 12, IQGPU_FMT_CU32 = 13, IQGPU_FMT_CS32 = 14, IQGPU_FMT_CF32 = 15,
    IQGPU_FMT_SC16Q11 = 16
};
/* == FilterType (include/common_types.h:45-51) */
enum { IQGPU_FILTER_NONE = 0, IQGPU_FILTER_LOWPASS = 1, IQGPU_FILTER_HIGHPASS = 2,
       IQGPU_FILTER_PASSBAND = 3, IQGPU_FILTER_STOPBAND = 4 };
/* == FilterTypeRequest (include/common_types.h:61-65) */
enum { IQGPU_FILTER_IMPL_AUTO = 0, IQGPU_FILTER_IMPL_FIR = 1, IQGPU_FILTER_IMPL_FFT = 2 };
/* == FilterImplementationType (include/common_types.h:53-59) */
enum { IQGPU_FI_NONE = 0, IQGPU_FI_FIR_SYMMETRIC = 1, IQGPU_FI_FIR_ASYMMETRIC = 2,
       IQGPU_FI_FFT_SYMMETRIC = 3, IQGPU_FI_FFT_ASYMMETRIC = 4 };

enum {
    IQGPU_OK = 0,
    IQGPU_EINVAL = -1,      /* bad argument / NULL handle */
    IQGPU_ENODEV = -2,      /* no HIP device, or device_ordinal out of range */
    IQGPU_ENOMEM = -3,      /* device or host allocation failed */
    IQGPU_ERATIO = -4,      /* ratio not finite or outside [1e-3, 1e3]   (src/setup.c:109-112) */
    IQGPU_EFORMAT = -5,     /* unhandled sample format                    (src/sample_convert.c:203-206) */
    IQGPU_ESHIFT = -6,      /* shift beyond 5x rate, or shift_after_resample without a shift (src/frequency_shift.c:36-49) */
    IQGPU_EFILTER = -7,     /* filter band beyond output Nyquist, fft size too small, too many stages (src/filter.c:80-84, 321-325) */
    IQGPU_ECAPACITY = -8,   /* out_capacity_bytes too small for this call */
    IQGPU_EHIP = -9,        /* a HIP runtime call failed */
    IQGPU_EUNSUPPORTED = -10/* valid for the reference but not built (nothing on the path returns it since round 2) */
};

typedef struct iqgpu_chain iqgpu_chain; /* opaque, like resampler_t (include/resampler.h:25-26) */

typedef struct { int type; float f1_hz, f2_hz; } iqgpu_filter_req; /* == FilterRequest: f1 = cutoff or centre, f2 = bandwidth */

/* Mirrors the AppConfig / AppResources fields the path reads (include/app_context.h:66-138). */
typedef struct {
    int    in_format, out_format;       /* IQGPU_FMT_*                                                      */
    double input_rate_hz;               /* source_info.samplerate                                           */
    double target_rate_hz;              /* config->target_rate; r = (float)(target/input)  src/setup.c:107  */
    float  resample_ratio;              /* if > 0: use this float ratio as-is (create_resampler's argument,  */
                                        /*   src/resampler.c:20) instead of deriving it from the rates      */
    float  gain;                        /* config->gain, applied at unpack (src/pre_processor.c:21-22)      */
    double shift_hz;                    /* resources->nco_shift_hz (sign selects mix up / down)             */
    int    shift_after_resample;        /* config->shift_after_resample                                     */
    int    dc_block_enable;             /* config->dc_block.enable                                          */
    int    iq_correct_enable;           /* config->iq_correction.enable                                     */
    float  iq_mag, iq_phase;            /* initial correction factors (reference starts at 0,0)             */
    int    no_resample;                 /* config->no_resample                                              */
    int    n_filters;                   /* config->num_filter_requests (<= 5)                               */
    iqgpu_filter_req filters[5];        /* config->filter_requests                                          */
    float  transition_width_hz;         /* config->transition_width_hz_arg (0 = auto)                       */
    float  attenuation_db;              /* config->attenuation_db_arg (0 = 60 dB)                           */
    int    filter_taps;                 /* config->filter_taps_arg AFTER the odd bump of src/config.c:233-236 (0 = auto) */
    int    filter_impl;                 /* IQGPU_FILTER_IMPL_*  (config->filter_type_request)               */
    int    fft_size;                    /* config->filter_fft_size_arg (0 = auto)                           */
    /* GPU-only knobs */
    int    device_ordinal;              /* HIP device index                                                 */
    size_t block_samples;               /* input samples per workgroup block, multiple of 2048; 0 = auto (one run per resident wave) */
    /* output AGC, between the post NCO and the pack (src/post_processor.c:55-57, src/agc.c) */
    int    agc_enable;                  /* config->output_agc.enable                                        */
    int    agc_profile;                 /* IQGPU_AGC_*  (config->output_agc.profile): DIGITAL, or liquid's agc_crcf as DX / LOCAL */
    float  agc_target;                  /* config->output_agc.target_level_arg (0 = AGC_DIGITAL_PEAK_TARGET) */
    int    agc_clock;                   /* IQGPU_AGC_CLOCK_*: what stands in for get_monotonic_time_sec()    */
    uint32_t agc_chunk_frames;          /* input frames per reference chunk, 0 = 16384 (PIPELINE_CHUNK_BASE_SAMPLES): */
                                        /*   every process() call is cut into chunks of this many input frames, */
                                        /*   counted from the start of the call, and agc_apply sees one chunk at a time */
} iqgpu_chain_desc;

/* AgcProfile, include/common_types.h:77-82 */
enum { IQGPU_AGC_OFF = 0, IQGPU_AGC_DX = 1, IQGPU_AGC_LOCAL = 2, IQGPU_AGC_DIGITAL = 3 };
/* SAMPLES: time of the output stream (samples_seen / target_rate) -- deterministic, and equal to the
 * wall clock when the reference runs in real time.  WALL: CLOCK_MONOTONIC read once per process() call. */
enum { IQGPU_AGC_CLOCK_SAMPLES = 0, IQGPU_AGC_CLOCK_WALL = 1 };

/* AppResources' AGC fields (include/app_context.h:227-231) */
typedef struct {
    int      locked;
    float    peak_memory;               /* profiles dx / local: agc_crcf's y2_prime (smoothed output energy) */
    float    current_gain;              /* profiles dx / local: agc_crcf's gain                              */
    int      reserved;
    double   last_strong_peak_time;
    uint64_t samples_seen;
} iqgpu_agc_state;

/* What create() derived; for diagnostics and for parity tests of the design path. */
typedef struct {
    float    ratio;                     /* float32 resampling ratio                                         */
    int      interp;                    /* 1 if ratio > 1                                                   */
    int      num_halfband_stages;       /* S                                                                */
    int      stage_m[16];               /* half-band semi-lengths in run order (highest rate first)         */
    float    rate_arb;                  /* arbitrary-resampler rate                                         */
    uint32_t arb_step;                  /* 24-bit fixed-point phase step                                    */
    uint32_t nco_dtheta;                /* uint32 NCO phase increment                                       */
    float    dc_alpha;                  /* float32 alpha of the DC blocker                                  */
    int      filter_post_resample;      /* apply_user_filter_post_resample                                  */
    int      filter_impl;               /* IQGPU_FI_*                                                       */
    uint32_t filter_ntaps;
    uint32_t filter_block;              /* fftfilt block size (0 for FIR)                                   */
    uint32_t history_samples;           /* processed input samples kept between calls                       */
} iqgpu_chain_info;

/* Per-kernel device time accumulated by HIP events on the chain's stream (profiling mode only). */
enum { IQGPU_K_DC_PREFIX = 0, IQGPU_K_DC_SCAN = 1, IQGPU_K_FRONT = 2, IQGPU_K_FILTER = 3, IQGPU_K_MOVE = 4, IQGPU_K_AGC = 5,
    IQGPU_K_CASCADE = 6, IQGPU_K_COUNT = 8 };
typedef struct {
    uint64_t launches[IQGPU_K_COUNT];
    double   ms[IQGPU_K_COUNT];
} iqgpu_profile;

/* ---- library ---- */
int         iqgpu_abi_version(void);
const char *iqgpu_last_error(void);              /* thread-local message of the last failure        */
int         iqgpu_device_count(void);            /* number of HIP devices, 0 if none / no runtime   */
/* PCI bus id ("0000:05:00.0") of device `ordinal` (hipDeviceGetPCIBusId): what a multi-GPU launcher logs to show that its
 * ranks sit on distinct devices (bench.py `config.devices`; the reference has no counterpart: it runs on the host) */
int         iqgpu_device_pci_bus_id(int ordinal, char *buf, size_t cap);
/* NUMA placement of whoever feeds a GPU (ABI v5).  No counterpart in the reference: it is one host process whose stage threads
 * stay on the CPU (src/pipeline.c:96-116); here every bench rank / harness shard thread streams through pinned buffers into ONE
 * device and should sit on that device's socket.  Both read sysfs only (KFD topology -> PCI address -> numa_node /
 * local_cpulist) and make NO HIP call, so they can -- and should -- run before the first GPU call of the process or thread and
 * before its pinned buffers are allocated.  *node = -1 when the host does not say (single node, VM); IQGPU_EUNSUPPORTED when a
 * *_VISIBLE_DEVICES variable holds something other than indices (the device order is then not derived and nothing is bound).
 * iqgpu_bind_thread_to_device: sched_setaffinity of the CALLING THREAD to the node's CPUs (intersected with its current mask)
 * and a preferred-node memory policy for it; threads created afterwards inherit both. */
int         iqgpu_device_numa_node(int ordinal, int *node, char *pci_bus_id, size_t cap);
int         iqgpu_bind_thread_to_device(int ordinal, int *node);

/* ---- chain lifecycle: replaces _create_dsp_components/_destroy_dsp_components (src/pipeline.c:138-157) ---- */
void   iqgpu_chain_desc_init(iqgpu_chain_desc *d);                 /* reference defaults: gain 1, cs16->cs16, no ops */
int    iqgpu_chain_create(const iqgpu_chain_desc *d, iqgpu_chain **out);
void   iqgpu_chain_destroy(iqgpu_chain *c);
int    iqgpu_chain_get_info(const iqgpu_chain *c, iqgpu_chain_info *info);
/* copies min(cap, ntaps) complex taps (re,im interleaved) of the combined user filter; returns ntaps */
int    iqgpu_chain_get_filter_taps(const iqgpu_chain *c, float *re_im, size_t cap_taps);
/* The create-time design path alone (ratio, half-band plan, NCO increment, filter placement and
 * taps) without touching a device: same validation and error codes as iqgpu_chain_create.
 * filter_taps_re_im / hb_taps / arb_proto may be NULL.  hb_taps receives the 4m+1 prototype of
 * every half-band stage back to back in run order; arb_proto the 3584 scaled polyphase taps. */
int    iqgpu_design_probe(const iqgpu_chain_desc *d, iqgpu_chain_info *info,
                          float *filter_taps_re_im, size_t cap_taps,
                          float *hb_taps, size_t cap_hb, float *arb_proto, size_t cap_arb);

/* frames a FRESH chain of this description emits for a stream of frames_in frames (no device needed): the
 * closed form of the resampler law and the FFT-block quantisation on whichever side of the resampler the filter
 * sits.  What a stitching writer uses to place the outputs of independent shards (8 iq_tool runs + cat). */
int    iqgpu_design_out_frames(const iqgpu_chain_desc *d, size_t frames_in, size_t *frames_out);

/* ---- per-chunk: replaces pre_processor_apply_chain (src/pre_processor.c:10), resampler_execute
 *      (include/resampler.h:48) and post_processor_apply_chain (src/post_processor.c:9) in one call ---- */
int    iqgpu_chain_process(iqgpu_chain *c, const void *raw_in, size_t frames_in,
                           void *out, size_t out_capacity_bytes, size_t *frames_out);
/* Same, with both buffers already in device memory (HBM) of the chain's device.  Asynchronous on the
 * chain's stream; *frames_out is exact on return (it is a closed form of the stream position). */
int    iqgpu_chain_process_device(iqgpu_chain *c, const void *d_raw_in, size_t frames_in,
                                  void *d_out, size_t out_capacity_bytes, size_t *frames_out);
/* Pipelined form of iqgpu_chain_process for a stage thread that must not stall on PCIe: submit() queues the
 * batch's H2D copy and returns with the batch's exact *frames_out (a closed form of the stream position behind
 * the batches already submitted) and a ticket; the batch's kernels and its D2H copy are queued by the submit()
 * calls that follow (three and five batches later) or by collect(), which blocks until the batch's output
 * bytes are in `out`.  Up to iqgpu_chain_pipeline_depth() batches may be in flight; copies of one batch
 * overlap the kernels of its neighbours.  The pipeline makes progress inside submit() / collect() only, and a
 * failure of a batch's kernels (IQGPU_EHIP ...) is reported by the call that launches them -- a later submit()
 * or the batch's collect() -- after which the handle is poisoned until iqgpu_chain_reset(), as with process().
 * The I/Q factors a batch runs with are those in force at its submit().  raw_in / out should be pinned (iqgpu_host_malloc_pinned) --
 * pageable memory works but serialises -- and must stay untouched until the ticket is collected.  Batches are
 * processed in submit order: the stream is continuous across them exactly as across iqgpu_chain_process calls
 * (this is what replaces the reference's chunk hand-off between its three stage threads, src/pipeline.c:436-595).
 * How long submit() may block: normally the ~10 us of queueing.  On a chain with the digital output AGC fused into its last kernel
 * (agc_enable, profile `digital`, past the lock) the launch of batch N first reads the AGC verdict of batch N-1 from a pinned word,
 * i.e. it waits until batch N-1's kernels have FINISHED (polled; after 0.25 s it falls back to a stream synchronise): submit() then
 * blocks for at most one batch's kernel time (0.02 - 1.3 ms at 2^18 - 2^28 frames).  Chains without that AGC never wait for device work
 * in submit(). */
int    iqgpu_chain_submit(iqgpu_chain *c, const void *raw_in, size_t frames_in,
                          void *out, size_t out_capacity_bytes, size_t *frames_out, uint64_t *ticket);
int    iqgpu_chain_collect(iqgpu_chain *c, uint64_t ticket);
int    iqgpu_chain_pipeline_depth(void);
/* == pre_processor_reset + resampler_reset + post_processor_reset (stream discontinuity) */
int    iqgpu_chain_reset(iqgpu_chain *c);
/* what the I/Q optimiser thread publishes (src/iq_correct.c:141-152 reads them once per chunk) */
int    iqgpu_chain_set_iq_factors(iqgpu_chain *c, float mag, float phase);
/* synchronises the chain's stream and reports the AGC state (agc.c keeps it in AppResources) */
int    iqgpu_chain_get_agc_state(iqgpu_chain *c, iqgpu_agc_state *st);
/* upper bound on frames_out for a call with frames_in frames (>= ceil(n*max(1,r))+128 (+ FFT block),
 * the reference's buffer rule src/pipeline.c:246-258) */
size_t iqgpu_chain_max_out_frames(const iqgpu_chain *c, size_t frames_in);
/* exact number of frames the NEXT call with frames_in frames will produce */
size_t iqgpu_chain_next_out_frames(const iqgpu_chain *c, size_t frames_in);

/* ---- I/Q imbalance optimiser (host CPU, like the reference's): produces the factors iq_correct_apply consumes ----
 * iq_correct_init / iq_correct_run_optimization / helpers (src/iq_correct.c:85-139, 154-219, 315-393), the
 * optimiser thread's body (src/utility_threads.c:35-47) and the 1024-sample hand-off (src/pipeline.c:468-476).
 * A 1024-point Hamming-windowed spectrum of a pre-processed block, the squared dB asymmetry of its two halves over
 * the inner 90 % of the bins as utility, 25 random +-1e-4 steps per run, 5 % smoothing, at most one run per 500 ms
 * and only on blocks whose peak-to-average power is >= 20 dB. */
typedef struct iqgpu_iq_optimizer iqgpu_iq_optimizer;
typedef float (*iqgpu_rand_dir_fn)(void *user);   /* > 0: +1, else -1  (stands in for _get_random_direction, iq_correct.c:391) */
typedef struct {
    uint64_t calls, runs, skipped_interval, skipped_power, accepted;   /* accepted: candidates that raised the metric */
    float    initial_metric, final_metric;                            /* of the last run */
    float    average_power_db, power_range_db;                        /* of the last power estimate (iq_correct.c:362-389) */
} iqgpu_iq_optimizer_stats;
int   iqgpu_iq_optimizer_create(iqgpu_iq_optimizer **out);            /* factors (0, 0); srand(time) like iq_correct_init */
void  iqgpu_iq_optimizer_destroy(iqgpu_iq_optimizer *o);
/* direction source: by default libc rand() > RAND_MAX / 2 as in the reference (irreproducible by design);
 * seed() switches to a private minstd generator, set_rng() to the caller's function */
int   iqgpu_iq_optimizer_seed(iqgpu_iq_optimizer *o, uint32_t seed);
int   iqgpu_iq_optimizer_set_rng(iqgpu_iq_optimizer *o, iqgpu_rand_dir_fn fn, void *user);
int   iqgpu_iq_optimizer_set_factors(iqgpu_iq_optimizer *o, float mag, float phase);
int   iqgpu_iq_optimizer_get_factors(iqgpu_iq_optimizer *o, float *mag, float *phase);
int   iqgpu_iq_optimizer_get_stats(iqgpu_iq_optimizer *o, iqgpu_iq_optimizer_stats *st);
/* _calculate_imbalance_metric (iq_correct.c:339-360) of one 1024-sample cf32 block for candidate factors */
float iqgpu_iq_optimizer_metric(iqgpu_iq_optimizer *o, const float *block_re_im_1024, float mag, float phase);
/* iq_correct_run_optimization on one 1024-sample cf32 block; now_sec < 0 reads CLOCK_MONOTONIC as the reference
 * does, otherwise the caller's clock (e.g. stream time) gates the 500 ms interval.  *updated = 1 if factors changed. */
int   iqgpu_iq_optimizer_run(iqgpu_iq_optimizer *o, const float *block_re_im_1024, double now_sec, int *updated);
int   iqgpu_iq_optimizer_touch(iqgpu_iq_optimizer *o, double now_sec); /* restart the interval (iq_correct.c:294-297) */
/* The block the reference hands over: the first 1024 samples of a chunk AFTER unpack / dc block / iq correct / pre NCO
 * (src/pipeline.c:468-476).  With the probe enabled every process call of >= 1024 frames leaves that block of its
 * first chunk in a pinned host buffer; read() waits for it.  *valid = 0 until one exists. */
int   iqgpu_chain_enable_iq_probe(iqgpu_chain *c, int enable);
int   iqgpu_chain_read_iq_probe(iqgpu_chain *c, float *block_re_im_1024, int *valid);
/* the optimiser thread's loop body: read the probe, run, publish with iqgpu_chain_set_iq_factors */
int   iqgpu_iq_optimizer_service(iqgpu_iq_optimizer *o, iqgpu_chain *c, double now_sec, int *updated);

/* ---- WAV capture metadata -> frequency shift (host only): the step in front of the path for real captures ----
 * SdrMetadata and its parsers (src/input_wav.c:54-100, 146-438), the shift rule of wav_initialize (592-629). */
enum { IQGPU_SDR_UNKNOWN = 0, IQGPU_SDR_CONSOLE = 1, IQGPU_SDR_SHARP = 2, IQGPU_SDR_UNO = 3, IQGPU_SDR_CONNECT = 4 };
typedef struct {
    int     source_software;                    /* IQGPU_SDR_*                                                  */
    char    software_name[64], software_version[64], radio_model[128];
    int     software_name_present, software_version_present, radio_model_present;
    double  center_freq_hz;  int center_freq_hz_present;
    int64_t timestamp_unix;  int timestamp_unix_present;
    char    timestamp_str[64]; int timestamp_str_present;
    int     sdr_info_present;                   /* any metadata source produced something                      */
    /* what sf_open reports (SF_INFO) and the path needs */
    int32_t sample_rate, channels, bits_per_sample, format_tag;
    int     in_format;                          /* IQGPU_FMT_CS16 or IQGPU_FMT_CU8 (the two subtypes the reference accepts) */
    uint64_t data_offset, data_bytes, frames;
} iqgpu_wav_info;
void iqgpu_wav_info_init(iqgpu_wav_info *md);
/* one `auxi` chunk: SDR Console XML first, SDRuno / SDRconnect binary otherwise; 1 if anything was parsed */
int  iqgpu_wav_parse_auxi(const void *chunk, size_t size, iqgpu_wav_info *md);
/* SDR#-style base name: ..._<freq>Hz... and _YYYYMMDD_HHMMSSZ; fills only what the chunk left unset */
int  iqgpu_wav_parse_filename(const char *base_filename, iqgpu_wav_info *md);
/* header walk (RIFF / RF64: fmt, auxi, data) + both parsers; IQGPU_EFORMAT for != 2 channels or an unsupported subtype */
int  iqgpu_wav_probe(const char *path, iqgpu_wav_info *md);
/* resources->nco_shift_hz = center_freq_hz - (double)center_target (float option); IQGPU_ESHIFT when --freq-shift is also
 * given or the file has no centre frequency; 0 shift when the option is not used */
int  iqgpu_wav_shift_hz(const iqgpu_wav_info *md, float center_target_hz, float freq_shift_hz_arg, double *nco_shift_hz);

/* ---- stream / profiling plumbing ---- */
int    iqgpu_chain_set_stream(iqgpu_chain *c, void *hip_stream);   /* hipStream_t; NULL = chain's own stream */
void  *iqgpu_chain_get_stream(const iqgpu_chain *c);
int    iqgpu_chain_synchronize(iqgpu_chain *c);
int    iqgpu_chain_set_profiling(iqgpu_chain *c, int enable);      /* brackets every launch with HIP events  */
int    iqgpu_chain_get_profile(iqgpu_chain *c, iqgpu_profile *p);  /* synchronises, then reports and clears  */
/* name of the front kernel the LAST process call launched: "k_front_mid<6,nco>", "k_front_mid<6,nonco>", "k_front_mid<8,nco>" (the
 * outputs per lane of the instantiation and whether it mixes; ",cf32" = cf32 out to a filter, ",8bit" = 8-bit frames in or out,
 * ",gain" = 16-bit frames with a gain or of the sc16q11 scale), "k_front_s1", "k_front_fat", "k_front_s2", "k_cascade+k_front_s1",
 * "k_cascade2+k_front_s1", "k_front_p0", "k_p0fft16" (round 6, opt-in by iqgpu_debug_set("fuse_filter", "1"): resampler and post-resample
 * filter in one kernel, no front launch), "k_front", "k_front+k_interp"; "" before the first call.  Diagnostics, bench.py's
 * roofline.kernel */
const char *iqgpu_chain_front_kernel(const iqgpu_chain *c);

/* Diagnostic switches (ABI v6; no reference counterpart).  The library reads NO switch from the environment: kernel selection and
 * plan overrides used by the parity tests and the A/B tools go through this one entry point.  A chain takes the table as it
 * stands when iqgpu_chain_create runs; later changes do not touch existing chains.  name: "no_fast", "agc_nofuse", "force_generic",
 * "fft_log2n", ... (abi.cpp kDebugNames; an unknown name is IQGPU_EINVAL); value NULL or "" clears the switch, name NULL clears all.
 * iqgpu_debug_list writes "name=value;name=value" of what is set (bench.py records it in config.debug). */
int    iqgpu_debug_set(const char *name, const char *value);
int    iqgpu_debug_list(char *buf, size_t cap);

/* diagnostic hook: copies the chain's 64 KiB scratch (per-phase cycle counters in builds
 * made with -DIQGPU_STAMPS) to the host and clears it */
int    iqgpu_chain_debug_read_scratch(iqgpu_chain *c, void *host_64k);

/* ---- operator-level entry points (same kernels, one operator enabled) ----
 * Names follow the reference functions they replace. Host buffers. */
size_t iqgpu_get_bytes_per_sample(int format);                                  /* get_bytes_per_sample, src/sample_convert.c:102 */
int    iqgpu_convert_block_to_cf32(const void *in, float *out_re_im, size_t frames,
                                   int in_format, float gain, int device);       /* convert_block_to_cf32, src/sample_convert.c:127 */
int    iqgpu_convert_cf32_to_block(const float *in_re_im, void *out, size_t frames,
                                   int out_format, int device);                  /* convert_cf32_to_block, src/sample_convert.c:213 */

/* ---- device memory helpers for hosts that have no HIP binding of their own (harness, ctypes) ---- */
int    iqgpu_device_malloc(int device, size_t bytes, void **d_ptr);
int    iqgpu_device_free(int device, void *d_ptr);
int    iqgpu_host_malloc_pinned(size_t bytes, void **h_ptr);
int    iqgpu_host_free_pinned(void *h_ptr);
int    iqgpu_memcpy_h2d(int device, void *d_dst, const void *h_src, size_t bytes);
int    iqgpu_memcpy_d2h(int device, void *h_dst, const void *d_src, size_t bytes);
int    iqgpu_memcpy_h2d_async(void *d_dst, const void *h_src, size_t bytes, void *hip_stream);
int    iqgpu_memcpy_d2h_async(void *h_dst, const void *d_src, size_t bytes, void *hip_stream);
int    iqgpu_stream_create(int device, void **hip_stream);
int    iqgpu_stream_destroy(void *hip_stream);
int    iqgpu_stream_synchronize(void *hip_stream);
int    iqgpu_event_create(void **hip_event);
int    iqgpu_event_destroy(void *hip_event);
int    iqgpu_event_record(void *hip_event, void *hip_stream);
int    iqgpu_stream_wait_event(void *hip_stream, void *hip_event);
int    iqgpu_event_elapsed_ms(void *start_event, void *stop_event, float *ms); /* synchronises on stop */

#ifdef __cplusplus
}
#endif
#endif /* IQGPU_H_ */
