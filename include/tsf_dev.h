/*
 * tsf_dev.h -- what tests and measurements need from libtsf_amd.so beyond the drop-in surface of tsf.h:
 * per-evaluation hooks (log-posterior / gradient, design matrix, device arithmetic self test), the route switches
 * of a context, the fit kernel's HIP-event timers.  Nothing here has a counterpart in the reference
 * (mageky/time-series-spark calls Prophet.fit / Prophet.predict and nothing finer), and a drop-in caller never
 * includes this file.  Same conventions as tsf.h (C99, plain pointers, 0 / < 0 return codes).
 */
#ifndef TSF_DEV_H
#define TSF_DEV_H

#include "tsf.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Route switches of ONE context (round 5; until round 4 these were process-wide environment variables read inside
 * the library).  No result depends on a route -- the GPU tests compare the routes bit for bit, which is what the
 * switches exist for, beside measurements.  value -1 (the state after tsf_create) = the library's own choice; what
 * the other values mean is stated per option.  Not part of what the reference's path needs: a drop-in caller never
 * calls this. */
enum {
    TSF_OPT_HARM = 0,        /* 0: the residual-form kernel streams every design column from the table instead of
                                expanding the Fourier columns from the rows' base pairs; 1 / 2: never / always the variant of
                                that kernel that requests a row one step ahead (default: where the rows come from HBM) */
    TSF_OPT_LATTICE,         /* 0 / 1: never / always the shared lattice table of a ragged call on regular timestamps */
    TSF_OPT_SPARSE_EXTRA,    /* 0: holiday columns of a wide model as dense columns; 2: sparse fit kernel, but its stragglers on the
                                64-column cooperative kernel instead of the sparse one (A/B runs, tests) */
    TSF_OPT_FIT_GROUPED,     /* 0: wide models on the workgroup kernel from the first evaluation */
    TSF_OPT_GRAM_SHARE,      /* 0: a ragged quadratic-form call builds Z^T Z per series even where calendars are shared */
    TSF_OPT_GRID_ORDER,      /* 0: a ragged call does not start its series grouped by calendar */
    TSF_OPT_GRID_SHARE,      /* 0: a ragged call keeps one set of grid tables per series */
    TSF_OPT_RAGGED_SPLIT,    /* 0: tsf_fit_ragged never cuts a call into length classes; 1: whenever its padding exceeds its rows,
                                also for calls whose tables are small anyway (tests) */
    TSF_OPT_QUAD_REG,        /* quadratic-form kernel variant: 0 Z^T Z in LDS, 1 in registers */
    TSF_OPT_QUAD_M2_LDS,     /* 0: the two-slot quadratic-form kernel reads Z^T Z from global memory */
    TSF_OPT_QUAD_W4,         /* waves per workgroup of the aligned quadratic-form kernel: 8, 12 or 16 */
    TSF_OPT_QUAD_RREG,       /* 0: residual-pass weights staged through memory */
    TSF_OPT_NEWTON_BATCH,    /* series per resident wave from which Newton runs several series per wave (0: never) */
    TSF_OPT_NEWTON_FLAGS, TSF_OPT_NEWTON_NS, TSF_OPT_NEWTON_LCAP, TSF_OPT_NEWTON_FILL,   /* dev knobs of that kernel */
    TSF_OPT_QUAD_RAW_Y,      /* 0: the quadratic-form fit reads a scaled step-major copy of y (written by the set-up kernel) instead
                                of the caller's rows */
    TSF_OPT_DEBUG_ASYNC_SCRATCH, /* dev (tools/dev/nb_debug.py): bit 0 the slot records of that kernel from hipMallocAsync /
                                hipFreeAsync as in round 3 instead of the context's cached block; bit 1 synchronise the
                                stream before the free; bit 2 canary pages either side of the records, checked after the
                                kernel (count on stderr); bit 3 the default pool never releases memory */
    TSF_OPT_COOP_TAIL,       /* residual-form launches: the fits still running are handed to the cooperative kernel once no
                                more of them are left than this many per hundred compute units (default 200) */
    TSF_OPT_MAP_DIRECT,      /* 0: converge = MAP always as a continuation of the Stan-rule fit (map_kernel), also where the model is
                                linear / additive on an aligned panel and the estimate can be computed directly (map_quad_kernel) */
    TSF_OPT_COUNT
};
int tsf_set_option(tsf_ctx *ctx, int option, int value);
int tsf_get_option(const tsf_ctx *ctx, int option);       /* -1: default (or a bad argument) */

/* ---- test / diagnostics hooks (host pointers) -------------------------------------------
 * tsf_eval: f = -log posterior and gradient [N][stride] at theta [N][stride] for an aligned
 * panel.  tsf_design: X [T][K] (row-major, original column order), scaled t [T]. */
int tsf_eval(tsf_ctx *ctx, const tsf_spec *spec, int64_t N, int32_t T, const int64_t *ds,
             const void *y, int32_t y_dtype, const double *floor, const double *cap,
             const double *extra, const double *theta, double *f_out, double *grad_out);
/* tsf_eval_quadratic: the QUADRATIC evaluation form (eval_form; what fit_quad_kernel evaluates at every
 * trial point of its line searches) at theta [N][stride], built around the reference point
 * theta_ref [N][stride]: s0 = |y - Z ref|^2 and c = Z^T (y - Z ref) from one residual-form pass at
 * theta_ref, M = Z^T Z once per call, then f and the gradient from (s0, c, M, theta - theta_ref).  Aligned
 * panel, linear growth, additive columns only, 3 + n_changepoints + K <= 64 (else an error).  The
 * per-evaluation check of the headline kernel's arithmetic against the literal model
 * (tests/test_gpu_literal.py); reference-side counterpart: none (Stan evaluates in residual form). */
int tsf_eval_quadratic(tsf_ctx *ctx, const tsf_spec *spec, int64_t N, int32_t T, const int64_t *ds,
                       const void *y, int32_t y_dtype, const double *extra, const double *theta_ref,
                       const double *theta, double *f_out, double *grad_out);
int tsf_design(tsf_ctx *ctx, const tsf_spec *spec, int32_t T, const int64_t *ds,
               const double *extra, double *X_out, double *t_out, tsf_grid_info *grid_out);
/* IEEE self test of the device arithmetic the canonical order relies on: fills out[n] with
 * op(a[n], b[n]) for op in {0:div, 1:sqrt(a), 2:det_exp(a), 3:det_log(a), 4:det_sin(a),
 * 5:det_cos(a), 6:fma(a,b,a)} computed on the GPU. */
int tsf_selftest_math(tsf_ctx *ctx, int32_t op, int64_t n, const double *a, const double *b,
                      double *out);

/* ---- measurement hooks ------------------------------------------------------------------
 * With profiling enabled every tsf_fit_*_dev call records a pair of HIP events on ITS stream
 * right before and after the fit kernel (the dominant kernel of the path); up to
 * TSF_PROFILE_RING calls are kept.  tsf_profile_read waits for the recorded events and
 * returns the kernel durations (milliseconds, oldest first) of the calls made since
 * profiling was last (re-)enabled; tsf_last_fit_kernel_ms returns the newest one. */
#define TSF_PROFILE_RING 64
int tsf_set_profiling(tsf_ctx *ctx, int32_t enable);
int tsf_profile_read(tsf_ctx *ctx, float *ms_out, int32_t max_n, int32_t *n_out);
int tsf_last_fit_kernel_ms(tsf_ctx *ctx, float *ms_out);
/* Which route the last fit call of this context took where the library decides on the device (tests, measurements):
 * *sparse_columns = 1 if a wide model's indicator columns ran in sparse form (every grid qualified), 0 if the dense
 * kernels ran or the call was not a candidate.  Waits for the device; valid until the next fit call. */
int tsf_last_fit_route(tsf_ctx *ctx, int32_t *sparse_columns);

#ifdef __cplusplus
}
#endif
#endif /* TSF_DEV_H */
