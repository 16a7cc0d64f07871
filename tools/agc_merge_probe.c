/* tools/agc_merge_probe.c -- can a parallel scheme reproduce the BITS of the sequential RMS AGC (profiles dx / local)?
 * The recurrence of liquid's agc_crcf as the oracle restates it (oracle/iq_oracle.c, ref: src/agc.c:92-100), run twice on the
 * same input from different starting states: how many samples until gain and smoothed energy are bit-equal for good?  A chunk-
 * parallel scheme (agc.hip: every chunk warms up from a guess) is exact only where that happens inside its warm-up.
 *   gcc -O2 -o /tmp/agc_merge_probe tools/agc_merge_probe.c -lm && /tmp/agc_merge_probe      (results: profiles/r04_agc_merge.txt) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void step(float xr, float xi, float alpha, float *g, float *p) {
    float yr = xr * *g, yi = xi * *g;
    float y2 = yr * yr + yi * yi;
    *p = (1.0f - alpha) * *p + alpha * y2;
    if (*p > 1e-6f) *g *= expf(-0.5f * alpha * logf(*p));
    if (*g > 1e6f) *g = 1e6f;
}
int main(void) {
    srand(5);
    for (int ai = 0; ai < 2; ai++) {
        float alpha = ai ? 1e-2f : 1e-4f;
        long maxn = (long)(400.0 / alpha);
        for (int trial = 0; trial < 12; trial++) {
            float amp = 0.05f + 0.3f * (trial % 4);
            float g1 = 1.0f + trial, p1 = 0.3f, g2 = (trial & 1) ? 0.2f : 30.0f, p2 = 1.0f;
            if (trial >= 8) { g2 = g1 * 1.0001f; p2 = p1 * 0.9999f; }   /* a good guess */
            long merged = -1; int run = 0;
            for (long n = 0; n < maxn; n++) {
                float ph = 0.37f * n;
                float xr = amp * cosf(ph) + 0.02f * ((rand() & 1023) / 512.0f - 1.0f), xi = amp * sinf(ph) + 0.02f * ((rand() & 1023) / 512.0f - 1.0f);
                if (trial % 3 == 2 && (n / 5000) % 2) { xr *= 0.05f; xi *= 0.05f; }   /* fades */
                step(xr, xi, alpha, &g1, &p1); step(xr, xi, alpha, &g2, &p2);
                if (memcmp(&g1, &g2, 4) == 0 && memcmp(&p1, &p2, 4) == 0) { if (++run == 1) merged = n; } else { run = 0; merged = -1; }
            }
            printf("alpha %g trial %2d: merged for good at n = %ld = %.1f / alpha\n", alpha, trial, merged, merged >= 0 ? merged * alpha : -1.0);
        }
    }
    return 0;
}
