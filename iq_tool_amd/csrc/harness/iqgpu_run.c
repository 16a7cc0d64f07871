/*
 * iqgpu_run -- minimal raw-file harness around libiqgpu: reader -> GPU chain -> writer.
 *
 * It plays the role that iq_tool's raw-file reader (src/input_rawfile.c:168-252), stage threads and
 * raw-file writer (src/output_raw_file.c:146-184) play around the sample path, with the reference's
 * option names where they exist, so that BASELINE.json's configs can be run from a shell:
 *
 *   iqgpu_run -i in.cs16 --raw-file-input-rate 2.4e6 --raw-file-input-sample-format cs16 \
 *             -o out.cs16 --output-rate 744187.5 --output-sample-format cs16 --freq-shift 200e3
 *
 * Copies are double-buffered: pinned host buffers, hipMemcpyAsync on separate streams, events
 * between them (all through the C ABI, no HIP header here), so H2D of chunk i+1 and D2H of chunk
 * i-1 overlap the kernels of chunk i.  Nothing is flushed at end of stream (as in the reference).
 *
 * --shards N splits the input file into N equal frame ranges, one worker thread (and one GPU,
 * round-robin over --devices) per range, each with fresh state, and stitches the outputs in shard
 * order with pwrite at offsets known up front (the output count of a shard is a closed form):
 * exactly what N iq_tool runs + cat would produce (BASELINE configs[4]).  No collective.  Every shard thread binds itself to
 * the NUMA node of its GPU before its first GPU call and before it allocates its pinned buffers (iqgpu_bind_thread_to_device),
 * and fails when the frames its chain produced differ from the count its output offset was planned with.
 *
 * --synthetic FRAMES [--synthetic-hash SEED]: no input file.  Without a seed one constant pinned buffer is sent again and again
 * (the PCIe-inclusive rate of the path, nothing else); with one, shard s is the stream frame n -> splitmix64((SEED + s) * K + n)
 * of its own (iq_tool_amd/synth.py hash_stream restates it): configs[4] at its real size -- 8 x 2.5 G frames -- without 80 GB of
 * files, every range of every shard reproducible by the checker.
 */
#define _GNU_SOURCE
#define _FILE_OFFSET_BITS 64
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "iqgpu.h"

#define NBUF 2

typedef struct {
    const char *in_path, *out_path;
    iqgpu_chain_desc desc;
    size_t chunk_frames;
    int shards, devices, device0;
    long long synthetic_frames;       /* > 0: no input file, reuse one pinned buffer (PCIe-inclusive rate) */
    int have_hash; unsigned long long hash_seed;   /* ... or generate shard s as the counter-hash stream of seed hash_seed + s */
    int no_bind;                      /* --no-numa-bind */
    int dry;                          /* --dry-placement: plan, bind, size the buffers, report -- no GPU call */
    int quiet;
} Options;

typedef struct {
    const Options *opt;
    int shard;
    long long first_frame, frames;    /* input range */
    long long out_offset_bytes;       /* where this shard's output starts in the output file */
    long long frames_out;             /* result */
    long long planned_out;            /* iqgpu_design_out_frames(frames): what out_offset_bytes of the NEXT shard was computed from */
    int device, numa_node;            /* where it ran; -1 = the host does not say / not bound */
    char bus_id[64]; int cpus_allowed; long long pinned_bytes, hbm_bytes;   /* --dry-placement's report */
    double seconds, stream_seconds;   /* whole shard incl. set-up / copy-process-copy loop only */
    int rc;
    char err[256];
} Shard;

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* frame n of the stream with this seed: splitmix64 of a counter (any range of it can be regenerated, iq_tool_amd/synth.py) */
static inline uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
/* cs16: I, Q = the two low 16-bit words as signed values, arithmetic >> 2 (quarter scale: nothing on the path clips); every other
 * integer format: the low bytes of the hash as they are (full-scale noise) */
static void hash_fill(void *dst, int fmt, size_t bps, uint64_t seed, long long first, size_t n)
{
    const uint64_t base = seed * 0xD1342543DE82EF95ull;
    if (fmt == IQGPU_FMT_CS16) {
        int16_t *d = (int16_t *)dst;
        for (size_t i = 0; i < n; i++) {
            const uint64_t x = mix64(base + (uint64_t)(first + (long long)i));
            d[2 * i] = (int16_t)((int16_t)(x & 0xffffu) >> 2);
            d[2 * i + 1] = (int16_t)((int16_t)((x >> 16) & 0xffffu) >> 2);
        }
    } else {
        unsigned char *d = (unsigned char *)dst;
        for (size_t i = 0; i < n; i++) {
            const uint64_t x = mix64(base + (uint64_t)(first + (long long)i));
            memcpy(d + i * bps, &x, bps);              /* (little-endian hosts only, like the raw formats themselves) */
        }
    }
}

static int fmt_from_name(const char *s)
{
    static const struct { const char *n; int v; } t[] = {
        {"cu8", IQGPU_FMT_CU8}, {"cs8", IQGPU_FMT_CS8}, {"cu16", IQGPU_FMT_CU16}, {"cs16", IQGPU_FMT_CS16},
        {"cs24", IQGPU_FMT_CS24}, {"cu32", IQGPU_FMT_CU32}, {"cs32", IQGPU_FMT_CS32}, {"cf32", IQGPU_FMT_CF32},
        {"sc16q11", IQGPU_FMT_SC16Q11}};
    for (size_t i = 0; i < sizeof(t) / sizeof(t[0]); i++) if (!strcmp(s, t[i].n)) return t[i].v;
    return -1;
}

#define CK(call) do { int rc_ = (call); if (rc_ != IQGPU_OK) { snprintf(sh->err, sizeof(sh->err), "%s: %s", #call, iqgpu_last_error()); sh->rc = rc_; goto done; } } while (0)

static void *run_shard(void *arg)
{
    Shard *sh = (Shard *)arg;
    const Options *o = sh->opt;
    iqgpu_chain_desc d = o->desc;
    d.device_ordinal = o->device0 + (sh->shard % o->devices);
    const size_t ibps = iqgpu_get_bytes_per_sample(d.in_format), obps = iqgpu_get_bytes_per_sample(d.out_format);
    iqgpu_chain *chain = NULL;
    void *s_in = NULL, *s_k = NULL, *s_out = NULL;
    void *h_in[NBUF] = {0}, *h_out[NBUF] = {0}, *d_in[NBUF] = {0}, *d_out[NBUF] = {0};
    void *e_in[NBUF] = {0}, *e_k[NBUF] = {0}, *e_out[NBUF] = {0};
    size_t out_frames[NBUF] = {0};
    int in_fd = -1, out_fd = -1;
    const double t0 = now_s();

    /* this thread -- and with it the pages it touches first and the buffers it pins -- onto the socket of its GPU, before anything
     * touches that GPU.  Best effort: a host that hides its topology leaves the thread where it is (numa_node stays -1) */
    sh->device = d.device_ordinal; sh->numa_node = -1;
    if (!o->no_bind) { int node = -1; if (iqgpu_bind_thread_to_device(d.device_ordinal, &node) == IQGPU_OK) sh->numa_node = node; }

    if (o->dry) {
        /* --dry-placement: everything the first multi-GPU run does BEFORE its first GPU call -- the device of
         * this shard, its PCI address and NUMA node from sysfs, the binding, the sizes of the buffers it would pin and allocate --
         * and nothing after it.  The capacity rule without a handle: src/pipeline.c:246-258 on the designed ratio */
        int node = -1;
        sh->bus_id[0] = 0;
        (void)iqgpu_device_numa_node(d.device_ordinal, &node, sh->bus_id, sizeof(sh->bus_id));
        if (sh->numa_node < 0 && o->no_bind) sh->numa_node = node;
        { cpu_set_t m; CPU_ZERO(&m); sh->cpus_allowed = pthread_getaffinity_np(pthread_self(), sizeof(m), &m) == 0 ? CPU_COUNT(&m) : -1; }
        size_t cap_frames = 0;
        CK(iqgpu_design_out_frames(&d, o->chunk_frames, &cap_frames));
        cap_frames += 64;
        sh->pinned_bytes = (long long)NBUF * (long long)(o->chunk_frames * ibps + cap_frames * obps);
        sh->hbm_bytes = sh->pinned_bytes;
        sh->frames_out = sh->planned_out;
        goto done;
    }

    CK(iqgpu_chain_create(&d, &chain));
    const size_t chunk = o->chunk_frames;
    const size_t out_cap = iqgpu_chain_max_out_frames(chain, chunk) * obps;
    CK(iqgpu_stream_create(d.device_ordinal, &s_in));
    CK(iqgpu_stream_create(d.device_ordinal, &s_k));
    CK(iqgpu_stream_create(d.device_ordinal, &s_out));
    CK(iqgpu_chain_set_stream(chain, s_k));
    for (int b = 0; b < NBUF; b++) {
        CK(iqgpu_host_malloc_pinned(chunk * ibps, &h_in[b]));
        CK(iqgpu_host_malloc_pinned(out_cap, &h_out[b]));
        CK(iqgpu_device_malloc(d.device_ordinal, chunk * ibps, &d_in[b]));
        CK(iqgpu_device_malloc(d.device_ordinal, out_cap, &d_out[b]));
        CK(iqgpu_event_create(&e_in[b])); CK(iqgpu_event_create(&e_k[b])); CK(iqgpu_event_create(&e_out[b]));
    }
    if (o->synthetic_frames <= 0) {
        in_fd = open(o->in_path, O_RDONLY);
        if (in_fd < 0) { snprintf(sh->err, sizeof(sh->err), "open %s: %s", o->in_path, strerror(errno)); sh->rc = -1; goto done; }
    } else if (!o->have_hash) {
        for (int b = 0; b < NBUF; b++) memset(h_in[b], 0x11 * (b + 1), chunk * ibps);
    }
    if (o->out_path) {
        out_fd = open(o->out_path, O_WRONLY | O_CREAT, 0644);
        if (out_fd < 0) { snprintf(sh->err, sizeof(sh->err), "open %s: %s", o->out_path, strerror(errno)); sh->rc = -1; goto done; }
    }

    long long done_in = 0, written = 0;
    const double t_loop = now_s();
    long long n_chunks = (sh->frames + (long long)chunk - 1) / (long long)chunk;
    /* software pipeline: iteration i first issues chunk i (H2D, kernels, D2H, all asynchronous), then
     * retires chunk i-1 (waits for its D2H, writes it): chunk i's copies and kernels overlap the
     * write-out of chunk i-1, and buffer i % 2 is free again because chunk i-2 retired last time */
    for (long long i = 0; i <= n_chunks; i++) {
        if (i < n_chunks) {
            const int b = (int)(i % NBUF);
            size_t n = chunk;
            if ((long long)n > sh->frames - done_in) n = (size_t)(sh->frames - done_in);
            if (in_fd >= 0) {
                const size_t nb = n * ibps;
                size_t got = 0;
                while (got < nb) {
                    ssize_t r = pread(in_fd, (char *)h_in[b] + got, nb - got, (sh->first_frame + done_in) * (long long)ibps + (long long)got);
                    if (r <= 0) { snprintf(sh->err, sizeof(sh->err), "pread: %s", r < 0 ? strerror(errno) : "short file"); sh->rc = -1; goto done; }
                    got += (size_t)r;
                }
            } else if (o->have_hash) {
                /* (buffer b is free: chunk i - 2, its last user, retired in the iteration before this one) */
                hash_fill(h_in[b], d.in_format, ibps, o->hash_seed + (uint64_t)sh->shard, done_in, n);
            }
            CK(iqgpu_memcpy_h2d_async(d_in[b], h_in[b], n * ibps, s_in));
            CK(iqgpu_event_record(e_in[b], s_in));
            CK(iqgpu_stream_wait_event(s_k, e_in[b]));
            CK(iqgpu_chain_process_device(chain, d_in[b], n, d_out[b], out_cap, &out_frames[b]));
            CK(iqgpu_event_record(e_k[b], s_k));
            CK(iqgpu_stream_wait_event(s_out, e_k[b]));
            if (out_frames[b]) CK(iqgpu_memcpy_d2h_async(h_out[b], d_out[b], out_frames[b] * obps, s_out));
            CK(iqgpu_event_record(e_out[b], s_out));
            done_in += (long long)n;
        }
        if (i >= 1) {
            const int b = (int)((i - 1) % NBUF);
            float ms;
            CK(iqgpu_event_elapsed_ms(e_k[b], e_out[b], &ms));    /* synchronises on e_out */
            if (out_fd >= 0 && out_frames[b]) {
                const size_t nb = out_frames[b] * obps;
                if (pwrite(out_fd, h_out[b], nb, sh->out_offset_bytes + written * (long long)obps) != (ssize_t)nb) {
                    snprintf(sh->err, sizeof(sh->err), "pwrite: %s", strerror(errno)); sh->rc = -1; goto done;
                }
            }
            written += (long long)out_frames[b];
        }
    }
    sh->frames_out = written;
    sh->stream_seconds = now_s() - t_loop;
    if (written != sh->planned_out) {
        snprintf(sh->err, sizeof(sh->err), "shard %d produced %lld frames, its place in the output was planned for %lld", sh->shard, written, sh->planned_out);
        sh->rc = -1;
    }
done:
    if (chain) iqgpu_chain_synchronize(chain);
    for (int b = 0; b < NBUF; b++) {
        if (e_in[b]) iqgpu_event_destroy(e_in[b]);
        if (e_k[b]) iqgpu_event_destroy(e_k[b]);
        if (e_out[b]) iqgpu_event_destroy(e_out[b]);
        if (h_in[b]) iqgpu_host_free_pinned(h_in[b]);
        if (h_out[b]) iqgpu_host_free_pinned(h_out[b]);
        if (d_in[b]) iqgpu_device_free(d.device_ordinal, d_in[b]);
        if (d_out[b]) iqgpu_device_free(d.device_ordinal, d_out[b]);
    }
    if (s_in) iqgpu_stream_destroy(s_in);
    if (s_k) { if (chain) iqgpu_chain_set_stream(chain, NULL); iqgpu_stream_destroy(s_k); }
    if (s_out) iqgpu_stream_destroy(s_out);
    if (chain) iqgpu_chain_destroy(chain);
    if (in_fd >= 0) close(in_fd);
    if (out_fd >= 0) close(out_fd);
    sh->seconds = now_s() - t0;
    return NULL;
}

static void usage(void)
{
    fprintf(stderr,
            "iqgpu_run -i IN --raw-file-input-rate HZ --raw-file-input-sample-format FMT -o OUT --output-rate HZ\n"
            "          [--output-sample-format FMT] [--freq-shift HZ] [--shift-after-resample] [--gain G] [--dc-block]\n"
            "          [--iq-factors MAG:PHASE] [--no-resample] [--lowpass HZ] [--highpass HZ] [--pass-range A:B] [--stopband A:B]\n"
            "          [--transition-width HZ] [--attenuation DB] [--filter-taps N] [--filter-type fir|fft] [--filter-fft-size N]\n"
            "          [--chunk-frames N (default 4194304)] [--shards N] [--devices N] [--device D] [--synthetic FRAMES [--synthetic-hash SEED]]\n"
            "          [--no-numa-bind] [--quiet] [--debug NAME=VALUE (iqgpu_debug_set)]\n"
            "          [--dry-placement (plan the shards, bind every shard thread, size its buffers, report as JSON: no GPU call)]\n");
}

int main(int argc, char **argv)
{
    Options o;
    memset(&o, 0, sizeof(o));
    iqgpu_chain_desc_init(&o.desc);
    o.chunk_frames = 1u << 22; o.shards = 1; o.devices = 1;
    double in_rate = 0, out_rate = 0;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
#define NEXT (i + 1 < argc ? argv[++i] : (usage(), exit(2), ""))
        if (!strcmp(a, "-i") || !strcmp(a, "--input")) { const char *v = NEXT; if (!strcmp(v, "raw-file")) v = NEXT; o.in_path = v; }
        else if (!strcmp(a, "-o") || !strcmp(a, "--output")) { const char *v = NEXT; if (!strcmp(v, "raw-file")) v = NEXT; o.out_path = v; }
        else if (!strcmp(a, "--raw-file-input-rate")) in_rate = atof(NEXT);
        else if (!strcmp(a, "--raw-file-input-sample-format")) o.desc.in_format = fmt_from_name(NEXT);
        else if (!strcmp(a, "--output-rate")) out_rate = atof(NEXT);
        else if (!strcmp(a, "--output-sample-format")) o.desc.out_format = fmt_from_name(NEXT);
        else if (!strcmp(a, "--freq-shift")) o.desc.shift_hz = atof(NEXT);
        else if (!strcmp(a, "--shift-after-resample")) o.desc.shift_after_resample = 1;
        else if (!strcmp(a, "--gain")) o.desc.gain = (float)atof(NEXT);
        else if (!strcmp(a, "--dc-block")) o.desc.dc_block_enable = 1;
        else if (!strcmp(a, "--iq-factors")) { o.desc.iq_correct_enable = 1; sscanf(NEXT, "%f:%f", &o.desc.iq_mag, &o.desc.iq_phase); }
        else if (!strcmp(a, "--no-resample")) o.desc.no_resample = 1;
        else if (!strcmp(a, "--lowpass") || !strcmp(a, "--highpass")) {
            if (o.desc.n_filters >= 5) { fprintf(stderr, "at most 5 filters\n"); return 2; }
            iqgpu_filter_req *r = &o.desc.filters[o.desc.n_filters++];
            r->type = !strcmp(a, "--lowpass") ? IQGPU_FILTER_LOWPASS : IQGPU_FILTER_HIGHPASS; r->f1_hz = (float)atof(NEXT); r->f2_hz = 0;
        } else if (!strcmp(a, "--pass-range") || !strcmp(a, "--stopband")) {
            if (o.desc.n_filters >= 5) { fprintf(stderr, "at most 5 filters\n"); return 2; }
            float s = 0, e = 0; sscanf(NEXT, "%f:%f", &s, &e);
            iqgpu_filter_req *r = &o.desc.filters[o.desc.n_filters++];
            r->type = !strcmp(a, "--pass-range") ? IQGPU_FILTER_PASSBAND : IQGPU_FILTER_STOPBAND;
            const float bw = e - s; r->f1_hz = s + (bw / 2.0f); r->f2_hz = bw;     /* src/config.c:203-214 */
        }
        else if (!strcmp(a, "--transition-width")) o.desc.transition_width_hz = (float)atof(NEXT);
        else if (!strcmp(a, "--attenuation")) o.desc.attenuation_db = (float)atof(NEXT);
        else if (!strcmp(a, "--filter-taps")) { int t = atoi(NEXT); if (t && t % 2 == 0) t++; o.desc.filter_taps = t; }   /* src/config.c:233-236 */
        else if (!strcmp(a, "--filter-type")) { const char *v = NEXT; o.desc.filter_impl = !strcmp(v, "fft") ? IQGPU_FILTER_IMPL_FFT : IQGPU_FILTER_IMPL_FIR; }
        else if (!strcmp(a, "--filter-fft-size")) o.desc.fft_size = atoi(NEXT);
        else if (!strcmp(a, "--output-agc")) { o.desc.agc_enable = 1; if (!o.desc.agc_profile) o.desc.agc_profile = IQGPU_AGC_LOCAL; }   /* src/config.c:306-310 */
        else if (!strcmp(a, "--agc-profile")) { const char *v = NEXT; o.desc.agc_enable = 1;
            o.desc.agc_profile = !strcasecmp(v, "dx") ? IQGPU_AGC_DX : !strcasecmp(v, "local") ? IQGPU_AGC_LOCAL : !strcasecmp(v, "digital") ? IQGPU_AGC_DIGITAL : -1; }
        else if (!strcmp(a, "--agc-target")) o.desc.agc_target = (float)atof(NEXT);
        else if (!strcmp(a, "--chunk-frames")) o.chunk_frames = (size_t)atoll(NEXT);
        else if (!strcmp(a, "--shards")) o.shards = atoi(NEXT);
        else if (!strcmp(a, "--devices")) o.devices = atoi(NEXT);
        else if (!strcmp(a, "--device")) o.device0 = atoi(NEXT);
        else if (!strcmp(a, "--synthetic")) o.synthetic_frames = atoll(NEXT);
        else if (!strcmp(a, "--synthetic-hash")) { o.have_hash = 1; o.hash_seed = strtoull(NEXT, NULL, 0); }
        else if (!strcmp(a, "--no-numa-bind")) o.no_bind = 1;
        else if (!strcmp(a, "--dry-placement")) o.dry = 1;
        else if (!strcmp(a, "--debug")) {              /* --debug name=value -> iqgpu_debug_set (the library reads no environment) */
            char kv[512]; snprintf(kv, sizeof(kv), "%s", NEXT);
            char *eq = strchr(kv, '=');
            if (!eq) { fprintf(stderr, "--debug wants name=value\n"); return 2; }
            *eq = 0;
            if (iqgpu_debug_set(kv, eq + 1) != IQGPU_OK) { fprintf(stderr, "%s\n", iqgpu_last_error()); return 2; }
        }
        else if (!strcmp(a, "--quiet")) o.quiet = 1;
        else { usage(); return 2; }
    }
    if (o.desc.in_format < 0 || o.desc.out_format < 0) { fprintf(stderr, "unknown sample format\n"); return 2; }
    if (o.have_hash && (o.synthetic_frames <= 0 || o.desc.in_format == IQGPU_FMT_CF32)) { fprintf(stderr, "--synthetic-hash needs --synthetic FRAMES and an integer input format\n"); return 2; }
    if (!(in_rate > 0) || (!o.desc.no_resample && !(out_rate > 0)) || (!o.in_path && o.synthetic_frames <= 0)) { usage(); return 2; }
    o.desc.input_rate_hz = in_rate; o.desc.target_rate_hz = o.desc.no_resample ? in_rate : out_rate;
    if (o.shards < 1) o.shards = 1;
    if (o.devices < 1) o.devices = 1;

    const size_t ibps = iqgpu_get_bytes_per_sample(o.desc.in_format), obps = iqgpu_get_bytes_per_sample(o.desc.out_format);
    long long total_frames = o.synthetic_frames;
    if (total_frames <= 0) {
        struct stat st;
        if (stat(o.in_path, &st) != 0) { fprintf(stderr, "stat %s: %s\n", o.in_path, strerror(errno)); return 1; }
        total_frames = (long long)(st.st_size / (off_t)ibps);
    }
    /* shard plan: equal frame ranges (the last takes the rest), output offsets from the closed form */
    Shard *sh = (Shard *)calloc((size_t)o.shards, sizeof(Shard));
    iqgpu_chain_info info;
    {
        int rc = iqgpu_design_probe(&o.desc, &info, NULL, 0, NULL, 0, NULL, 0);
        if (rc != IQGPU_OK) { fprintf(stderr, "%s\n", iqgpu_last_error()); return 1; }
    }
    long long per = total_frames / o.shards, off = 0;
    for (int s = 0; s < o.shards; s++) {
        sh[s].opt = &o; sh[s].shard = s;
        sh[s].first_frame = (long long)s * per;
        sh[s].frames = (s == o.shards - 1) ? total_frames - sh[s].first_frame : per;
        sh[s].out_offset_bytes = off;
        /* frames a fresh chain will produce for sh[s].frames input frames: the library's own closed form (decimating
         * or interpolating resampler, FFT-block quantisation in front of or behind it) */
        size_t nout = 0;
        if (iqgpu_design_out_frames(&o.desc, (size_t)sh[s].frames, &nout) != IQGPU_OK) { fprintf(stderr, "%s\n", iqgpu_last_error()); return 1; }
        sh[s].planned_out = (long long)nout;
        off += (long long)nout * (long long)obps;
    }
    if (o.out_path && !o.dry) { int fd = open(o.out_path, O_WRONLY | O_CREAT | O_TRUNC, 0644); if (fd >= 0) close(fd); }

    const double t0 = now_s();
    pthread_t *th = (pthread_t *)calloc((size_t)o.shards, sizeof(pthread_t));
    for (int s = 0; s < o.shards; s++) pthread_create(&th[s], NULL, run_shard, &sh[s]);
    int rc = 0;
    long long frames_out = 0;
    double stream_s = 0;
    for (int s = 0; s < o.shards; s++) {
        pthread_join(th[s], NULL);
        if (sh[s].rc) { fprintf(stderr, "shard %d failed: %s\n", s, sh[s].err); rc = 1; }
        frames_out += sh[s].frames_out;
        if (sh[s].stream_seconds > stream_s) stream_s = sh[s].stream_seconds;
    }
    const double dt = now_s() - t0;
    if (o.dry) {
        /* one JSON line: the placement as planned; `distinct_devices` is what an N-shard job on N GPUs must show */
        int distinct = 0;
        for (int s = 0; s < o.shards; s++) { int seen = 0; for (int q = 0; q < s; q++) if (!strcmp(sh[q].bus_id, sh[s].bus_id) && sh[q].device == sh[s].device) seen = 1; if (!seen) distinct++; }
        printf("{\"dry_placement\": true, \"frames_in\": %lld, \"frames_out\": %lld, \"shards\": %d, \"devices\": %d, \"distinct_devices\": %d, \"per_shard\": [",
               total_frames, frames_out, o.shards, o.devices, distinct);
        for (int s = 0; s < o.shards; s++)
            printf("%s{\"shard\": %d, \"device\": %d, \"pci_bus_id\": \"%s\", \"numa_node\": %d, \"cpus_allowed\": %d, \"first_frame\": %lld, \"frames_in\": %lld, \"planned_out\": %lld, "
                   "\"out_offset_bytes\": %lld, \"pinned_bytes\": %lld, \"hbm_bytes\": %lld}", s ? ", " : "", s, sh[s].device, sh[s].bus_id, sh[s].numa_node, sh[s].cpus_allowed,
                   sh[s].first_frame, sh[s].frames, sh[s].planned_out, sh[s].out_offset_bytes, sh[s].pinned_bytes, sh[s].hbm_bytes);
        printf("]}\n");
        free(th); free(sh);
        return rc;
    }
    if (!o.quiet) {
        printf("{\"frames_in\": %lld, \"frames_out\": %lld, \"shards\": %d, \"devices\": %d, \"seconds\": %.6f, \"msps_end_to_end\": %.3f, \"stream_seconds\": %.6f, \"msps_streaming\": %.3f, "
               "\"h2d_GBs\": %.3f, \"d2h_GBs\": %.3f, \"in_bytes_per_frame\": %zu, \"out_bytes_per_frame\": %zu, \"input\": \"%s\", \"per_shard\": [",
               total_frames, frames_out, o.shards, o.devices, dt, total_frames / dt / 1e6, stream_s, stream_s > 0 ? total_frames / stream_s / 1e6 : 0.0,
               stream_s > 0 ? (double)total_frames * (double)ibps / stream_s / 1e9 : 0.0, stream_s > 0 ? (double)frames_out * (double)obps / stream_s / 1e9 : 0.0, ibps, obps,
               o.synthetic_frames <= 0 ? "file" : o.have_hash ? "synthetic-hash" : "synthetic-constant");
        for (int s = 0; s < o.shards; s++)
            printf("%s{\"shard\": %d, \"device\": %d, \"numa_node\": %d, \"first_frame\": %lld, \"frames_in\": %lld, \"frames_out\": %lld, \"planned_out\": %lld, \"out_offset_bytes\": %lld, \"seconds\": %.6f}",
                   s ? ", " : "", s, sh[s].device, sh[s].numa_node, sh[s].first_frame, sh[s].frames, sh[s].frames_out, sh[s].planned_out, sh[s].out_offset_bytes, sh[s].stream_seconds);
        printf("]}\n");
    }
    free(th); free(sh);
    return rc;
}
