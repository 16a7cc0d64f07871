/* hostcall_bench.c -- the pipelined host entry point from C, as the INTEGRATION.md stub calls it (no ctypes in the way):
 * iqgpu_chain_submit / _collect with pinned buffers, iqgpu_chain_pipeline_depth() batches in flight, NRSC-5 chain.
 *   gcc -O2 -I include tools/hostcall_bench.c -o tools/hostcall_bench -L iq_tool_amd/lib -liqgpu -Wl,-rpath,$PWD/iq_tool_amd/lib
 *   tools/hostcall_bench [log2 frames per batch ...]                                                                     */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "iqgpu.h"

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
    int sizes[16], ns = 0;
    for (int i = 1; i < argc && ns < 16; i++) sizes[ns++] = atoi(argv[i]);
    if (!ns) { sizes[0] = 14; sizes[1] = 18; sizes[2] = 20; sizes[3] = 22; ns = 4; }
    for (int si = 0; si < ns; si++) {
        const size_t n = (size_t)1 << sizes[si];
        iqgpu_chain_desc d;
        iqgpu_chain_desc_init(&d);
        d.input_rate_hz = 2.4e6; d.target_rate_hz = 744187.5; d.shift_hz = 200e3;
        iqgpu_chain *c = NULL;
        if (iqgpu_chain_create(&d, &c) != IQGPU_OK) { fprintf(stderr, "%s\n", iqgpu_last_error()); return 1; }
        const int depth = iqgpu_chain_pipeline_depth();
        const size_t cap = iqgpu_chain_max_out_frames(c, n) * 4;
        void *in[8], *out[8];
        uint64_t ticket[8];
        for (int s = 0; s < depth; s++) {
            if (iqgpu_host_malloc_pinned(n * 4, &in[s]) || iqgpu_host_malloc_pinned(cap, &out[s])) { fprintf(stderr, "%s\n", iqgpu_last_error()); return 1; }
            short *p = (short *)in[s];
            for (size_t i = 0; i < 2 * n; i++) p[i] = (short)((i * 2654435761u >> 18) & 0x3fff) - 8192;
        }
        long total = (long)(((size_t)1 << 31) >> sizes[si]);
        if (total > 20000) total = 20000;
        if (total < 4 * depth) total = 4 * depth;
        double t0 = 0, t_submit = 0, t_collect = 0;
        for (int pass = 0; pass < 2; pass++) {                 /* pass 0 warms (buffers grow, pages touched) */
            const long cnt = pass ? total : 2 * depth;
            if (pass) t0 = now_s();
            int inflight = 0; long head = 0, tail = 0;
            for (long i = 0; i < cnt; i++) {
                double ta = now_s();
                if (inflight == depth) { if (iqgpu_chain_collect(c, ticket[tail % depth])) return 1; tail++; inflight--; }
                double tb = now_s();
                size_t got = 0;
                if (iqgpu_chain_submit(c, in[head % depth], n, out[head % depth], cap, &got, &ticket[head % depth])) { fprintf(stderr, "%s\n", iqgpu_last_error()); return 1; }
                if (pass) { t_collect += tb - ta; t_submit += now_s() - tb; }
                head++; inflight++;
            }
            while (inflight) { if (iqgpu_chain_collect(c, ticket[tail % depth])) return 1; tail++; inflight--; }
        }
        const double dt = (now_s() - t0) / (double)total;
        printf("C: iqgpu_chain_submit/_collect (pinned, %d in flight), %9zu frames per batch: %8.1f us per batch, %6.2f GS/s sustained  (host time inside submit %.1f us, waiting in collect %.1f us per batch)\n",
               depth, n, dt * 1e6, (double)n / dt / 1e9, t_submit / total * 1e6, t_collect / total * 1e6);
        for (int s = 0; s < depth; s++) { iqgpu_host_free_pinned(in[s]); iqgpu_host_free_pinned(out[s]); }
        iqgpu_chain_destroy(c);
    }
    return 0;
}
