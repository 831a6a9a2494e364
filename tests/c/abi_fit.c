/* Plain C99 caller of the GPU part of the C-ABI: fits an aligned panel read from raw binary
 * files and predicts, with no Python and no torch in the process -- the boundary the
 * reference's FFI would bind (include/tsf.h).
 * Usage: abi_fit N T H ds.i64 y.f64 fut.i64 out.f64   (linear growth, additive weekly order 3)
 * out: N*stride theta, then N*H yhat, then N (status, n_iter, n_eval as doubles). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tsf.h"

static void *slurp(const char *path, size_t bytes)
{
    FILE *f = fopen(path, "rb");
    void *p = malloc(bytes ? bytes : 1);
    if (!f || !p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot read %s\n", path); exit(10); }
    fclose(f);
    return p;
}

int main(int argc, char **argv)
{
    if (argc != 8) return 2;
    const int64_t N = atoll(argv[1]);
    const int32_t T = atoi(argv[2]), H = atoi(argv[3]);
    int64_t *ds = slurp(argv[4], sizeof(int64_t) * (size_t)T);
    double *y = slurp(argv[5], sizeof(double) * (size_t)(N * T));
    int64_t *fut = slurp(argv[6], sizeof(int64_t) * (size_t)H);

    tsf_spec spec;
    tsf_spec_default(&spec);
    spec.growth = TSF_GROWTH_LINEAR;
    spec.n_seas = 1;
    spec.seas_period[0] = 7.0;
    spec.seas_order[0] = 3;
    spec.seas_prior_scale[0] = 10.0;
    spec.seas_mode[0] = TSF_MODE_ADDITIVE;
    const int stride = tsf_theta_stride(&spec);

    tsf_ctx *ctx = NULL;
    if (tsf_create(0, &ctx) != 0) { fprintf(stderr, "tsf_create failed: no GPU\n"); return 3; }
    tsf_fit_out out;
    tsf_grid_info grid;
    out.theta = calloc((size_t)(N * stride), sizeof(double));
    out.y_scale = calloc((size_t)N, sizeof(double));
    out.fval = calloc((size_t)N, sizeof(double));
    out.status = calloc((size_t)N, sizeof(int32_t));
    out.n_iter = calloc((size_t)N, sizeof(int32_t));
    out.n_eval = calloc((size_t)N, sizeof(int32_t));
    out.grid = &grid;
    int rc = tsf_fit_aligned(ctx, &spec, N, T, ds, y, TSF_Y_F64, NULL, NULL, NULL, &out);
    if (rc != 0) { fprintf(stderr, "fit rc=%d: %s\n", rc, tsf_last_error(ctx)); return 4; }
    /* the same fit again with the evaluation counts of the first one as scheduling hints (tsf_set_cost_hints):
     * the launch starts its longest fits first, and not a bit of the result may change */
    {
        double *theta0 = malloc((size_t)(N * stride) * sizeof(double));
        int32_t *hints = malloc((size_t)N * sizeof(int32_t));
        memcpy(theta0, out.theta, (size_t)(N * stride) * sizeof(double));
        memcpy(hints, out.n_eval, (size_t)N * sizeof(int32_t));
        if (tsf_set_cost_hints(ctx, hints, N) != 0) { fprintf(stderr, "hints: %s\n", tsf_last_error(ctx)); return 7; }
        rc = tsf_fit_aligned(ctx, &spec, N, T, ds, y, TSF_Y_F64, NULL, NULL, NULL, &out);
        if (rc != 0) { fprintf(stderr, "hinted fit rc=%d: %s\n", rc, tsf_last_error(ctx)); return 8; }
        if (memcmp(theta0, out.theta, (size_t)(N * stride) * sizeof(double)) != 0) { fprintf(stderr, "hints changed a result\n"); return 9; }
        free(theta0); free(hints);
    }
    double *yhat = calloc((size_t)(N * H), sizeof(double));
    rc = tsf_predict(ctx, &spec, N, H, out.theta, out.y_scale, &grid, 1, fut, 1, NULL, NULL, NULL, yhat, NULL);
    if (rc != 0) { fprintf(stderr, "predict rc=%d: %s\n", rc, tsf_last_error(ctx)); return 5; }
    tsf_destroy(ctx);

    FILE *f = fopen(argv[7], "wb");
    if (!f) return 6;
    fwrite(out.theta, sizeof(double), (size_t)(N * stride), f);
    fwrite(yhat, sizeof(double), (size_t)(N * H), f);
    for (int64_t n = 0; n < N; ++n) {
        double t[3];
        t[0] = out.status[n]; t[1] = out.n_iter[n]; t[2] = out.n_eval[n];
        fwrite(t, sizeof(double), 3, f);
    }
    fclose(f);
    printf("stride=%d S=%d T=%d\n", stride, grid.S, grid.T);
    return 0;
}
