/* Plain C99 caller of the C-ABI (include/tsf.h): proves the header is C, not C++, and that the
 * host stages (spec defaults, reader, packer, sink) work without Python or a GPU.
 * Usage: abi_host_stages <input.csv> <output.csv>   (input: header-less "dim_id,ts,qty" rows of
 * series 7).  Prints one summary line; exit code 0 on success. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tsf.h"

int main(int argc, char **argv)
{
    if (argc != 3) return 2;
    tsf_spec spec;
    tsf_spec_default(&spec);
    if (tsf_spec_size() != (int)sizeof(tsf_spec)) return 3;
    if (tsf_theta_stride(&spec) != 3 + spec.n_changepoints + tsf_spec_K(&spec)) return 4;

    const char *paths[1];
    int64_t part_sid[1] = {7};
    tsf_csv *tab = NULL;
    int64_t n_rows = 0, bad_line = 0;
    int32_t bad_file = -1;
    paths[0] = argv[1];
    int rc = tsf_csv_read(1, paths, part_sid, "dtq", 2, &tab, &n_rows, &bad_file, &bad_line);
    if (rc != 0) { fprintf(stderr, "read rc=%d file=%d line=%lld\n", rc, bad_file, (long long)bad_line); return 5; }
    int64_t *sid = malloc(sizeof(int64_t) * (size_t)(n_rows + 1));
    int64_t *did = malloc(sizeof(int64_t) * (size_t)(n_rows + 1));
    int64_t *ds = malloc(sizeof(int64_t) * (size_t)(n_rows + 1));
    double *y = malloc(sizeof(double) * (size_t)(n_rows + 1));
    if (tsf_csv_fetch(tab, sid, did, ds, y) != 0) return 6;
    tsf_csv_free(tab);

    tsf_pack *plan = NULL;
    int64_t rows = 0, n_series = 0;
    int32_t identity = 0;
    if (tsf_pack_rows(n_rows, sid, did, ds, y, 2, &plan, &rows, &n_series, &identity) != 0) return 7;
    int64_t *ksid = malloc(sizeof(int64_t) * (size_t)(n_series + 1));
    int64_t *kdid = malloc(sizeof(int64_t) * (size_t)(n_series + 1));
    int64_t *off = malloc(sizeof(int64_t) * (size_t)(n_series + 2));
    int64_t *dso = malloc(sizeof(int64_t) * (size_t)(rows + 1));
    double *yo = malloc(sizeof(double) * (size_t)(rows + 1));
    int64_t *span = malloc(sizeof(int64_t) * (size_t)(n_series + 1));
    int64_t *mdt = malloc(sizeof(int64_t) * (size_t)(n_series + 1));
    double *ymax = malloc(sizeof(double) * (size_t)(n_series + 1));
    if (tsf_pack_fetch(plan, ksid, kdid, off, dso, yo, span, mdt, ymax) != 0) return 8;
    tsf_pack_free(plan);

    /* the packed rows written back through the sink (quantity truncated to an integer) */
    int64_t *q = malloc(sizeof(int64_t) * (size_t)(rows + 1));
    int64_t *osid = malloc(sizeof(int64_t) * (size_t)(rows + 1));
    int64_t *odid = malloc(sizeof(int64_t) * (size_t)(rows + 1));
    for (int64_t s = 0; s < n_series; ++s)
        for (int64_t r = off[s]; r < off[s + 1]; ++r) { osid[r] = ksid[s]; odid[r] = kdid[s]; q[r] = (int64_t)yo[r]; }
    if (tsf_csv_write_forecasts(argv[2], "2020-01-01T00:00:00+00:00", rows, osid, odid, dso, q, 2) != 0) return 9;
    printf("rows_in=%lld rows=%lld series=%lld identity=%d first_key=%lld/%lld span0=%lld min_dt0=%lld ymax0=%.1f\n",
           (long long)n_rows, (long long)rows, (long long)n_series, identity, (long long)ksid[0],
           (long long)kdid[0], (long long)span[0], (long long)mdt[0], ymax[0]);
    return 0;
}
