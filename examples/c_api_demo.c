/* c_api_demo.c -- the C-ABI of include/mtg.h driven from plain C: no Python, no torch.
 *
 *   gcc -std=c99 -O2 -I include examples/c_api_demo.c -o examples/c_api_demo \
 *       -L mind_the_gaps_amd -lmtg_hip -Wl,-rpath,$PWD/mind_the_gaps_amd -lm
 *   ./examples/c_api_demo            -> one line per evaluation: "lnL status"
 *
 * A damped random walk + SHO (the tutorials' null model) on a small irregular light curve with a
 * deterministic pattern, so that tests/test_capi_example_gpu.py can rebuild the same inputs and
 * compare the printed log-likelihoods with the oracle.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "mtg.h"

#define N 400
#define B 6
#define PI 3.14159265358979323846

int main(void)
{
    static double t[N], y[N], dy[N];
    double tt = 0.0;
    int n, b, rc = 1;
    for (n = 0; n < N; ++n) {                       /* irregular sampling with two long gaps */
        tt += 0.3 + 0.7 * fabs(sin(1.7 * n)) + ((n % 150) == 149 ? 40.0 : 0.0);
        t[n] = tt;
        y[n] = 100.0 + 8.0 * sin(0.9 * tt) + 3.0 * cos(0.13 * n * n);
        dy[n] = 1.0 + 0.5 * fabs(cos(2.3 * n)) + 1e-12;   /* yerr = dy + 1e-12 (gpmodelling.py:54) */
    }
    const int32_t kinds[2] = {MTG_TERM_DRW, MTG_TERM_SHO};
    /* full vector: DRW (log_S0, log_omega0), SHO (log_S0, log_Q, log_omega0), mean value */
    const double full[6] = {log(100.0), log(2 * PI / 20.0), log(50.0), log(3.0), log(2 * PI / 7.0), 100.0};
    const int32_t free_index[5] = {0, 1, 2, 3, 4};
    const double bounds[12] = {-10, 50, -10, 10, -10, 50, -10, 10, -10, 10, -INFINITY, INFINITY};
    double theta[B][5], out[B];
    int32_t status[B];
    for (b = 0; b < B; ++b) {
        int p;
        for (p = 0; p < 5; ++p) theta[b][p] = full[p] * (1.0 + 0.04 * (b - 2) * (p % 2 ? 1 : -1));
    }
    theta[5][1] = 11.0;                              /* outside the box: status 1, -inf */

    mtg_ctx *ctx = mtg_create(0);
    if (!ctx) { fprintf(stderr, "mtg_create: %s\n", mtg_last_error(NULL)); return 2; }
    if (mtg_set_lightcurves(ctx, N, 1, t, 0, y, dy, NULL) != MTG_OK) goto fail;
    if (mtg_set_model(ctx, 2, kinds, NULL, MTG_MEAN_CONSTANT, 6, full, 5, free_index, bounds) != MTG_OK) goto fail;
    if (mtg_loglike_batch(ctx, B, &theta[0][0], NULL, 1, out, status) != MTG_OK) goto fail;
    for (b = 0; b < B; ++b) printf("%.17g %d\n", out[b], (int)status[b]);
    rc = 0;
fail:
    if (rc) fprintf(stderr, "mtg: %s\n", mtg_last_error(ctx));
    mtg_destroy(ctx);
    return rc;
}
