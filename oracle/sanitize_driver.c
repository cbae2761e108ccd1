/* oracle/sanitize_driver.c -- TEST INFRASTRUCTURE: runs the C restatement (celerite_ref.c) on a case
 * read from a binary file and prints the log-probabilities, so that tests/test_oracle.py can run it
 * under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (the
 * sanitizers cannot be preloaded into the Python process that loads the plain library):
 *   gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -fopenmp celerite_ref.c sanitize_driver.c -lm
 *   ./a.out case.bin   ->  one "%.17g %d" line per evaluation (two-sweep), then the same for the fused sweep
 * File layout (native endianness): int64 N, L, B, nterms, PF, mean_kind, add_prior, nthreads;
 * int32 kinds[nterms]; double extra[nterms], t[N], y[L][N], dy[L][N], bounds[PF][2], params[B][PF];
 * int32 lc_index[B]. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int oracle_logprob_batch(long N, long L, const double *t, const double *y, const double *dy, int nterms,
                         const int *kinds, const double *extra, int mean_kind, int PF, const double *bounds, long B,
                         const double *params, const int *lc_index, int add_prior, int nthreads, double *out,
                         int *status);
int oracle_logprob_batch_fused(long N, long L, const double *t, const double *y, const double *dy, int nterms,
                               const int *kinds, const double *extra, int mean_kind, int PF, const double *bounds,
                               long B, const double *params, const int *lc_index, int add_prior, int nthreads,
                               double *out, int *status);

static void *take(FILE *f, size_t bytes)
{
    void *p = malloc(bytes ? bytes : 1);
    if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(2); }
    return p;
}

int main(int argc, char **argv)
{
    if (argc != 2) { fprintf(stderr, "usage: %s case.bin\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int64_t h[8];
    if (fread(h, sizeof h[0], 8, f) != 8) { fprintf(stderr, "short header\n"); return 2; }
    const long N = h[0], L = h[1], B = h[2];
    const int nterms = (int)h[3], PF = (int)h[4], mean_kind = (int)h[5], add_prior = (int)h[6], nthreads = (int)h[7];
    int *kinds = take(f, sizeof(int) * nterms);
    double *extra = take(f, sizeof(double) * nterms);
    double *t = take(f, sizeof(double) * N);
    double *y = take(f, sizeof(double) * L * N);
    double *dy = take(f, sizeof(double) * L * N);
    double *bounds = take(f, sizeof(double) * 2 * PF);
    double *params = take(f, sizeof(double) * B * PF);
    int *lc = take(f, sizeof(int) * B);
    fclose(f);
    double *out = malloc(sizeof(double) * B);
    int *status = malloc(sizeof(int) * B);
    for (int fused = 0; fused < 2; ++fused) {
        int rc = (fused ? oracle_logprob_batch_fused : oracle_logprob_batch)(N, L, t, y, dy, nterms, kinds, extra, mean_kind, PF,
                                                                             bounds, B, params, lc, add_prior, nthreads, out,
                                                                             status);
        if (rc != 0) { fprintf(stderr, "oracle failed (%d)\n", rc); return 3; }
        for (long b = 0; b < B; ++b) printf("%.17g %d\n", out[b], status[b]);
    }
    free(kinds); free(extra); free(t); free(y); free(dy); free(bounds); free(params); free(lc); free(out); free(status);
    return 0;
}
