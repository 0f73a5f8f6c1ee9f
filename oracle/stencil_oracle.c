/*
 * stencil_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See stencil_oracle.h for scope, pinning status and the reference lines
 * followed.  Build with -ffp-contract=off so every float operation rounds
 * once, like the reference's cpu backend built with g++ on x86-64.
 */
#include "stencil_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* Window mode (tests of grids too large to sweep on the CPU): the buffers hold the window
 * [row0, row0+H) x [col0, col0+W) of a larger grid; transition functions see the global coordinates and the
 * global grid range, neighbours outside the WINDOW read the halo value.  Cells further than
 * n_iterations * n_subiterations * radius from a window border that is not a border of the global grid are
 * exact; the caller discards the rest.  Off (a whole grid) unless set. */
static size_t window_row0 = 0, window_col0 = 0, window_grid_h = 0, window_grid_w = 0;
void oracle_set_window(size_t row0, size_t col0, size_t grid_h, size_t grid_w) {
    window_row0 = row0;
    window_col0 = col0;
    window_grid_h = grid_h;
    window_grid_w = grid_w;
}

/* One full-grid sweep: StencilStream/cpu/StencilUpdate.hpp:185-223. */
static void sweep(const oracle_function *f, const unsigned char *src, unsigned char *dst, size_t H,
                  size_t W, const unsigned char *halo, size_t iteration, size_t subiteration,
                  const void *tdv, int n_threads) {
    const size_t R = f->radius, D = 2 * R + 1, cs = f->cell_size;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 1 ? n_threads : 1)
#endif
    for (long long rr = 0; rr < (long long)H; rr++) {
        size_t r = (size_t)rr;
        unsigned char *nb = (unsigned char *)malloc(D * D * cs);
        for (size_t c = 0; c < W; c++) {
            /* :202-216 -- neighbours inside the grid come from the source, the rest is halo */
            for (size_t rel_r = 0; rel_r < D; rel_r++) {
                for (size_t rel_c = 0; rel_c < D; rel_c++) {
                    const unsigned char *cell;
                    if (r + rel_r >= R && c + rel_c >= R && r + rel_r < H + R &&
                        c + rel_c < W + R) {
                        cell = src + ((r + rel_r - R) * W + (c + rel_c - R)) * cs;
                    } else {
                        cell = halo;
                    }
                    memcpy(nb + (rel_r * D + rel_c) * cs, cell, cs);
                }
            }
            oracle_stencil st;
            st.row = r + window_row0;
            st.col = c + window_col0;
            st.grid_h = window_grid_h ? window_grid_h : H;
            st.grid_w = window_grid_w ? window_grid_w : W;
            st.iteration = iteration;
            st.subiteration = subiteration;
            st.tdv = tdv;
            st.radius = R;
            st.cell_size = cs;
            st.nb = nb;
            f->transition(f->ctx, &st, dst + (r * W + c) * cs); /* :219 */
        }
        free(nb);
    }
}

/* Driver: StencilStream/cpu/StencilUpdate.hpp:109-142. */
int oracle_run(const oracle_function *f, const void *in, void *out, size_t H, size_t W,
               const void *halo_value, size_t iteration_offset, size_t n_iterations,
               int n_threads) {
    const size_t bytes = H * W * f->cell_size;
    if (bytes == 0)
        return 0;
    unsigned char *swap_a = (unsigned char *)malloc(bytes);
    unsigned char *swap_b = (unsigned char *)malloc(bytes);
    unsigned char *tdv = (unsigned char *)malloc(f->tdv_size ? f->tdv_size : 1);
    if (!swap_a || !swap_b || !tdv) {
        free(swap_a);
        free(swap_b);
        free(tdv);
        return -1;
    }
    /* touch the pages from one thread: first-touch faults taken by all OpenMP threads at once serialise on
     * the address-space lock (seconds for a few hundred MiB inside a VM) */
    memset(swap_a, 0, bytes);
    memset(swap_b, 0, bytes);
    const unsigned char *pass_source = (const unsigned char *)in; /* :112 */
    unsigned char *pass_target = swap_b;                           /* :113 */

    for (size_t i_iter = 0; i_iter < n_iterations; i_iter++) {
        /* the TDV is evaluated once per iteration and shared by its sub-iterations (:197) */
        if (f->tdv_size && f->tdv)
            f->tdv(f->ctx, iteration_offset + i_iter, tdv);
        for (size_t i_sub = 0; i_sub < f->n_subiterations; i_sub++) {
            sweep(f, pass_source, pass_target, H, W, (const unsigned char *)halo_value,
                  iteration_offset + i_iter, i_sub, f->tdv_size ? tdv : NULL, n_threads);
            if (i_iter == 0 && i_sub == 0) { /* :122-127 */
                pass_source = swap_b;
                pass_target = swap_a;
            } else {
                const unsigned char *t = pass_source;
                pass_source = pass_target;
                pass_target = (unsigned char *)t;
            }
        }
    }
    memcpy(out, pass_source, bytes); /* :141 -- the result aliases the last written grid */
    free(swap_a);
    free(swap_b);
    free(tdv);
    return 0;
}

/* stencil[dr][dc], origin = centre (Stencil.hpp:120-122). */
#define NB(st, T, dr, dc)                                                                          \
    (*(const T *)((st)->nb + ((size_t)((long)(st)->radius + (dr)) * (2 * (st)->radius + 1) +      \
                              (size_t)((long)(st)->radius + (dc))) *                              \
                                 (st)->cell_size))

/* ------------------------------------------------------------------ Jacobi */
typedef struct {
    int variant;
    float coef[9];
} jacobi_ctx;

static void jacobi_fn(const void *vctx, const oracle_stencil *st, void *out) {
    const jacobi_ctx *k = (const jacobi_ctx *)vctx;
    const float *c = k->coef;
    float v;
#define S(dr, dc) NB(st, float, dr, dc)
    switch (k->variant) {
    case ORACLE_JACOBI1_GENERAL:
        v = c[0] * S(0, 0);
        break;
    case ORACLE_JACOBI2_CONSTANT:
        v = (S(-1, 0) + S(1, 0)) * 0.5f;
        break;
    case ORACLE_JACOBI3_CONSTANT:
        v = (S(-1, 0) + S(0, 0) + S(1, 0)) * 0.33333334f;
        break;
    case ORACLE_JACOBI4_CONSTANT:
        v = (S(-1, 0) + S(0, -1) + S(1, 0) + S(0, 1)) * 0.25f;
        break;
    case ORACLE_JACOBI5_CONSTANT:
        v = (S(-1, 0) + S(0, -1) + S(1, 0) + S(0, 1) + S(0, 0)) * 0.2f;
        break;
    case ORACLE_JACOBI4_GENERAL:
        v = c[0] * S(-1, 0) + c[1] * S(0, -1) + c[2] * S(1, 0) + c[3] * S(0, 1);
        break;
    case ORACLE_JACOBI5_GENERAL:
        v = c[0] * S(-1, 0) + c[1] * S(0, -1) + c[2] * S(1, 0) + c[3] * S(0, 1) +
            c[4] * S(0, 0);
        break;
    default: { /* ORACLE_JACOBI9_GENERAL: kernels.hpp:309-317, coef[r+1][c+1] row-major */
        float sum = 0.0f;
        for (int r = -1; r <= 1; r++)
            for (int cc = -1; cc <= 1; cc++)
                sum += c[(r + 1) * 3 + (cc + 1)] * S(r, cc);
        v = sum;
    }
    }
#undef S
    *(float *)out = v;
}

int oracle_jacobi(int variant, const float *coef, const float *in, float *out, size_t H,
                  size_t W, float halo, size_t iteration_offset, size_t n_iterations,
                  int n_threads) {
    static const int n_coef[8] = {1, 0, 0, 0, 0, 4, 5, 9};
    if (variant < 0 || variant > 7)
        return -2;
    jacobi_ctx k;
    memset(&k, 0, sizeof k);
    k.variant = variant;
    for (int i = 0; i < n_coef[variant]; i++)
        k.coef[i] = coef[i];
    oracle_function f = {sizeof(float), 1, 1, 0, jacobi_fn, NULL, &k};
    return oracle_run(&f, in, out, H, W, &halo, iteration_offset, n_iterations, n_threads);
}

/* A dense 5 x 5 Jacobi (radius 2): the loop of Jacobi9General (examples/jacobi/kernels.hpp:307-318: rows, then
 * columns, sum starting at 0.0f) over a radius-2 stencil.  NOT a function of the reference -- SURVEY section 8(f)4
 * asks for a tuned radius > 1 kernel and the reference ships no such application; the reference code it exercises
 * is the radius-2 Stencil indexing (Stencil.hpp:120-146, tests/Stencil.cpp:35-50) and the sweep with a 2-cell halo. */
static void jacobi25_fn(const void *vctx, const oracle_stencil *st, void *out) {
    const float *coef = (const float *)vctx;
    float sum = 0.0f;
    for (int r = -2; r <= 2; r++)
        for (int c = -2; c <= 2; c++)
            sum += coef[(r + 2) * 5 + (c + 2)] * NB(st, float, r, c);
    *(float *)out = sum;
}

int oracle_jacobi25(const float *coef, const float *in, float *out, size_t H, size_t W, float halo,
                    size_t n_iterations, int n_threads) {
    float k[25];
    memcpy(k, coef, sizeof k);
    oracle_function f = {sizeof(float), 2, 1, 0, jacobi25_fn, NULL, k};
    return oracle_run(&f, in, out, H, W, &halo, 0, n_iterations, n_threads);
}

void oracle_jacobi_init(float *grid, size_t H, size_t W) {
    /* examples/jacobi/jacobi.cpp:114-122 -- the comparisons are done in double */
    for (size_t r = 0; r < H; r++)
        for (size_t c = 0; c < W; c++)
            grid[r * W + c] =
                (r >= H * 0.25 && r < H * 0.75 && c >= W * 0.25 && c < W * 0.75) ? 1.0f : 0.0f;
}

/* ----------------------------------------------------------------- HotSpot */
void oracle_hotspot_params_for_grid(size_t n_rows, size_t n_columns, oracle_hotspot_params *p) {
    /* examples/hotspot/hotspot.cpp:40-55 and :281-295, same types per expression */
    const float t_chip = 0.0005f, chip_height = 0.016f, chip_width = 0.016f;
    const double MAX_PD = 3.0e6, PRECISION = 0.001, SPEC_HEAT_SI = 1.75e6, FACTOR_CHIP = 0.5;
    const int K_SI = 100;
    float grid_height = chip_height / n_rows;
    float grid_width = chip_width / n_columns;
    float Cap = FACTOR_CHIP * SPEC_HEAT_SI * t_chip * grid_height * grid_width;
    float Rx = grid_width / (2.0 * K_SI * t_chip * grid_height);
    float Ry = grid_height / (2.0 * K_SI * t_chip * grid_width);
    float Rz = t_chip / (K_SI * grid_height * grid_width);
    float max_slope = MAX_PD / (FACTOR_CHIP * t_chip * SPEC_HEAT_SI);
    float step = PRECISION / max_slope / 1000.0;
    p->Rx_1 = 1.f / Rx;
    p->Ry_1 = 1.f / Ry;
    p->Rz_1 = 1.f / Rz;
    p->Cap_1 = step / Cap;
}

static void hotspot_fn(const void *vctx, const oracle_stencil *st, void *out) {
    /* examples/hotspot/hotspot.cpp:69-96 */
    const oracle_hotspot_params *k = (const oracle_hotspot_params *)vctx;
    const float amb_temp = 80.0f;
    float power = NB(st, oracle_hotspot_cell, 0, 0).power;
    float old = NB(st, oracle_hotspot_cell, 0, 0).temp;
    float top = NB(st, oracle_hotspot_cell, -1, 0).temp;
    float bottom = NB(st, oracle_hotspot_cell, 1, 0).temp;
    float left = NB(st, oracle_hotspot_cell, 0, -1).temp;
    float right = NB(st, oracle_hotspot_cell, 0, 1).temp;

    if (st->row == 0) {
        top = old;
    } else if (st->row == st->grid_h - 1) {
        bottom = old;
    }
    if (st->col == 0) {
        left = old;
    } else if (st->col == st->grid_w - 1) {
        right = old;
    }

    float new_temp = old + k->Cap_1 * (power + (bottom + top - 2.f * old) * k->Ry_1 +
                                       (right + left - 2.f * old) * k->Rx_1 +
                                       (amb_temp - old) * k->Rz_1);
    oracle_hotspot_cell nc = {new_temp, power};
    *(oracle_hotspot_cell *)out = nc;
}

int oracle_hotspot(const oracle_hotspot_params *p, const oracle_hotspot_cell *in,
                   oracle_hotspot_cell *out, size_t H, size_t W, size_t iteration_offset,
                   size_t n_iterations, int n_threads) {
    const oracle_hotspot_cell halo = {0.0f, 0.0f}; /* hotspot.cpp:300 */
    oracle_function f = {sizeof(oracle_hotspot_cell), 1, 1, 0, hotspot_fn, NULL, p};
    return oracle_run(&f, in, out, H, W, &halo, iteration_offset, n_iterations, n_threads);
}

static void hotspot_f64_fn(const void *vctx, const oracle_stencil *st, void *out) {
    /* examples/hotspot/hotspot.cpp:69-96 with FLOAT = double */
    const oracle_hotspot_params_f64 *k = (const oracle_hotspot_params_f64 *)vctx;
    const double amb_temp = 80.0;
    double power = NB(st, oracle_hotspot_cell_f64, 0, 0).power;
    double old = NB(st, oracle_hotspot_cell_f64, 0, 0).temp;
    double top = NB(st, oracle_hotspot_cell_f64, -1, 0).temp;
    double bottom = NB(st, oracle_hotspot_cell_f64, 1, 0).temp;
    double left = NB(st, oracle_hotspot_cell_f64, 0, -1).temp;
    double right = NB(st, oracle_hotspot_cell_f64, 0, 1).temp;
    if (st->row == 0) {
        top = old;
    } else if (st->row == st->grid_h - 1) {
        bottom = old;
    }
    if (st->col == 0) {
        left = old;
    } else if (st->col == st->grid_w - 1) {
        right = old;
    }
    double new_temp = old + k->Cap_1 * (power + (bottom + top - 2.0 * old) * k->Ry_1 +
                                        (right + left - 2.0 * old) * k->Rx_1 + (amb_temp - old) * k->Rz_1);
    oracle_hotspot_cell_f64 nc = {new_temp, power};
    *(oracle_hotspot_cell_f64 *)out = nc;
}

int oracle_hotspot_f64(const oracle_hotspot_params_f64 *p, const oracle_hotspot_cell_f64 *in,
                       oracle_hotspot_cell_f64 *out, size_t H, size_t W, size_t iteration_offset,
                       size_t n_iterations, int n_threads) {
    const oracle_hotspot_cell_f64 halo = {0.0, 0.0};
    oracle_function f = {sizeof(oracle_hotspot_cell_f64), 1, 1, 0, hotspot_f64_fn, NULL, p};
    return oracle_run(&f, in, out, H, W, &halo, iteration_offset, n_iterations, n_threads);
}

/* ------------------------------------------------------------------ Conway */
static void conway_fn(const void *vctx, const oracle_stencil *st, void *out) {
    /* examples/conway/conway.cpp:38-55 */
    (void)vctx;
    int alive = 0;
    for (int r = -1; r <= 1; r++)
        for (int c = -1; c <= 1; c++)
            if (NB(st, uint8_t, r, c) && !(r == 0 && c == 0))
                alive += 1;
    uint8_t v;
    if (NB(st, uint8_t, 0, 0))
        v = (alive == 2 || alive == 3);
    else
        v = (alive == 3);
    *(uint8_t *)out = v;
}

int oracle_conway(const uint8_t *in, uint8_t *out, size_t H, size_t W, size_t n_iterations,
                  int n_threads) {
    const uint8_t halo = 0; /* Params default Cell() = false, conway.cpp:102-105 */
    oracle_function f = {1, 1, 1, 0, conway_fn, NULL, NULL};
    return oracle_run(&f, in, out, H, W, &halo, 0, n_iterations, n_threads);
}

/* --------------------------------------------------------------- selfcheck */
static const oracle_selfcheck_cell selfcheck_halo = {0, 0, 0, 0, 2}; /* TransFuncs.hpp:43 */

static void selfcheck_tdv(const void *ctx, size_t iteration, void *out) {
    (void)ctx;
    *(size_t *)out = iteration; /* TransFuncs.hpp:65 */
}

static void selfcheck_fn(const void *vctx, const oracle_stencil *st, void *out) {
    /* tests/TransFuncs.hpp:67-103, n_subiterations = 2 */
    (void)vctx;
    const int radius = (int)st->radius;
    oracle_selfcheck_cell nc = NB(st, oracle_selfcheck_cell, 0, 0);
    int is_valid = 1;
    for (int r = -radius; r <= radius; r++) {
        for (int c = -radius; c <= radius; c++) {
            oracle_selfcheck_cell old = NB(st, oracle_selfcheck_cell, r, c);
            int cell_r = (int)(st->row + (size_t)(long)r);
            int cell_c = (int)(st->col + (size_t)(long)c);
            if (cell_r >= 0 && cell_c >= 0 && (size_t)cell_r < st->grid_h &&
                (size_t)cell_c < st->grid_w) {
                is_valid &= old.r == cell_r;
                is_valid &= old.c == cell_c;
                is_valid &= (size_t)(long)old.i_iteration == st->iteration;
                is_valid &= (size_t)(long)old.i_subiteration == st->subiteration;
                is_valid &= old.status == 0;
            } else {
                is_valid &= old.r == selfcheck_halo.r;
                is_valid &= old.c == selfcheck_halo.c;
                is_valid &= old.i_iteration == selfcheck_halo.i_iteration;
                is_valid &= old.i_subiteration == selfcheck_halo.i_subiteration;
                is_valid &= old.status == selfcheck_halo.status;
            }
        }
    }
    is_valid &= *(const size_t *)st->tdv == st->iteration;

    nc.status = is_valid ? 0 : 1;
    if (nc.i_subiteration == 2 - 1) {
        nc.i_iteration += 1;
        nc.i_subiteration = 0;
    } else {
        nc.i_subiteration++;
    }
    *(oracle_selfcheck_cell *)out = nc;
}

int oracle_selfcheck(size_t radius, const oracle_selfcheck_cell *in, oracle_selfcheck_cell *out,
                     size_t H, size_t W, size_t iteration_offset, size_t n_iterations,
                     int n_threads) {
    oracle_function f = {sizeof(oracle_selfcheck_cell), radius,        2, sizeof(size_t),
                         selfcheck_fn,                  selfcheck_tdv, NULL};
    return oracle_run(&f, in, out, H, W, &selfcheck_halo, iteration_offset, n_iterations,
                      n_threads);
}

/* -------------------------------------------------------------------- FDTD */
float oracle_fdtd_tdv(const oracle_fdtd_params *p, size_t iteration) {
    /* examples/fdtd/src/Kernel.hpp:80-84 -- host libm cosf/expf in float */
    float current_time = iteration * p->dt;
    float wave_progress = (current_time - p->t_0) / p->tau;
    return cosf(p->omega * current_time) * expf(-1 * wave_progress * wave_progress);
}

static void fdtd_tdv(const void *ctx, size_t iteration, void *out) {
    *(float *)out = oracle_fdtd_tdv((const oracle_fdtd_params *)ctx, iteration);
}

static void fdtd_fn(const void *vctx, const oracle_stencil *st, void *out) {
    /* examples/fdtd/src/Kernel.hpp:86-128 with CoefResolver (material/CoefResolver.hpp:59-66) */
    const oracle_fdtd_params *k = (const oracle_fdtd_params *)vctx;
    oracle_fdtd_cell cell = NB(st, oracle_fdtd_cell, 0, 0);

    float r = st->row;
    float c = st->col;
    float source_distance_score = r * (r - 2 * k->source_r) + c * (c - 2 * k->source_c);

    float ca = cell.ca, cb = cell.cb, da = cell.da, db = cell.db;
    if (st->subiteration == 0) {
        cell.ex *= ca;
        cell.ex += cb * (NB(st, oracle_fdtd_cell, 0, 0).hz - NB(st, oracle_fdtd_cell, 0, -1).hz);

        cell.ey *= ca;
        cell.ey += cb * (NB(st, oracle_fdtd_cell, -1, 0).hz - NB(st, oracle_fdtd_cell, 0, 0).hz);
    } else {
        cell.hz *= da;
        cell.hz += db * (NB(st, oracle_fdtd_cell, 0, 1).ex - NB(st, oracle_fdtd_cell, 0, 0).ex +
                         NB(st, oracle_fdtd_cell, 0, 0).ey - NB(st, oracle_fdtd_cell, 1, 0).ey);

        if (source_distance_score <= k->source_distance_bound &&
            st->iteration <= k->cutoff_iteration) {
            float interp_factor;
            if (k->source_radius_squared != 0) {
                float cell_distance_squared =
                    source_distance_score + k->source_c * k->source_c + k->source_r * k->source_r;
                /* 1.0 is a double literal: the subtraction happens in double (:113) */
                interp_factor = 1.0 - (float)cell_distance_squared / k->source_radius_squared;
            } else {
                interp_factor = 1.0;
            }
            float source_amplitude = *(const float *)st->tdv;
            cell.hz += interp_factor * source_amplitude;
        }

        if (st->iteration > k->detect_iteration) {
            cell.hz_sum += cell.hz * cell.hz;
        }
    }
    *(oracle_fdtd_cell *)out = cell;
}

int oracle_fdtd(const oracle_fdtd_params *p, const oracle_fdtd_cell *in, oracle_fdtd_cell *out,
                size_t H, size_t W, size_t iteration_offset, size_t n_iterations, int n_threads) {
    const oracle_fdtd_cell halo = {0, 0, 0, 0, 0, 0, 0, 0}; /* CoefResolver.hpp:31 */
    oracle_function f = {sizeof(oracle_fdtd_cell), 1, 2, sizeof(float), fdtd_fn, fdtd_tdv, p};
    return oracle_run(&f, in, out, H, W, &halo, iteration_offset, n_iterations, n_threads);
}

/* ------------------------------------------------------------------ convection (examples/convection/convection.cpp) */
#define CC(dr, dc) NB(st, oracle_convection_cell, dr, dc)

static void pseudo_transient_fn(const void *vctx, const oracle_stencil *st, void *out) {
    /* convection.cpp:95-176; the macros of :67-74 written out: D_XA(F) = [1][0].F - [0][0].F, D_YA(F) = [0][1].F -
     * [0][0].F, D_XI(F) = [1][1].F - [0][1].F, D_YI(F) = [1][1].F - [1][0].F */
    const oracle_pseudo_transient_params *k = (const oracle_pseudo_transient_params *)vctx;
    oracle_convection_cell n = CC(0, 0);
    const size_t x = st->row, y = st->col, nx = k->nx, ny = k->ny;
    if (st->subiteration == 0) {
        if (x < nx && y < ny + 1)
            n.ErrV = CC(0, 0).Vy;
        if (x < nx && y < ny)
            n.ErrP = CC(0, 0).Pt;
        if (x < nx && y < ny) {
            double delta_V = (CC(1, 0).Vx - CC(0, 0).Vx) / k->dx + (CC(0, 1).Vy - CC(0, 0).Vy) / k->dy;
            double eta = k->eta0 * (1.0 - k->delta_eta_delta_T * (CC(0, 0).T + k->deltaT / 2.0));
            n.Pt = CC(0, 0).Pt - k->delta_tau_iter / k->beta * delta_V;
            n.tau_xx = 2.0 * eta * ((CC(1, 0).Vx - CC(0, 0).Vx) / k->dx - (1.0 / 3.0) * delta_V);
            n.tau_yy = 2.0 * eta * ((CC(0, 1).Vy - CC(0, 0).Vy) / k->dy - (1.0 / 3.0) * delta_V);
            if (x < nx - 1 && y < ny - 1)
                n.sigma_xy = eta * ((CC(1, 1).Vx - CC(1, 0).Vx) / k->dy + (CC(1, 1).Vy - CC(0, 1).Vy) / k->dx);
        }
    } else if (st->subiteration == 1) {
        if (x >= 1 && y >= 1) {
            if (x < (nx + 1) - 1 && y < ny - 1) {
                double Rx = 1.0 / k->rho *
                            ((CC(0, 0).tau_xx - CC(-1, 0).tau_xx) / k->dx +
                             (CC(-1, 0).sigma_xy - CC(-1, -1).sigma_xy) / k->dy -
                             (CC(0, 0).Pt - CC(-1, 0).Pt) / k->dx);
                n.dVxd_tau = k->dampX * CC(0, 0).dVxd_tau + Rx * k->delta_tau_iter;
                n.Vx = CC(0, 0).Vx + n.dVxd_tau * k->delta_tau_iter;
            }
            if (x < nx - 1 && y < (ny + 1) - 1) {
                double Ry = 1.0 / k->rho *
                            ((CC(0, 0).tau_yy - CC(0, -1).tau_yy) / k->dy +
                             (CC(0, -1).sigma_xy - CC(-1, -1).sigma_xy) / k->dx -
                             (CC(0, 0).Pt - CC(0, -1).Pt) / k->dy +
                             k->roh0_g_alpha * ((CC(0, -1).T + CC(0, 0).T) * 0.5));
                n.dVyd_tau = k->dampY * CC(0, 0).dVyd_tau + Ry * k->delta_tau_iter;
                n.Vy = CC(0, 0).Vy + n.dVyd_tau * k->delta_tau_iter;
            }
        }
    } else if (st->subiteration == 2) {
        if (x < nx + 1 && y < ny) {
            if (y == 0)
                n.Vx = CC(0, 1).Vx;
            if (y == ny - 1)
                n.Vx = CC(0, -1).Vx;
        }
        if (x < nx && y < ny + 1) {
            if (x == 0)
                n.Vy = CC(1, 0).Vy;
            if (x == nx - 1)
                n.Vy = CC(-1, 0).Vy;
        }
        if (x < nx && y < ny + 1)
            n.ErrV = CC(0, 0).ErrV - n.Vy;
        if (x < nx && y < ny)
            n.ErrP = CC(0, 0).ErrP - CC(0, 0).Pt;
    }
    *(oracle_convection_cell *)out = n;
}

int oracle_pseudo_transient(const oracle_pseudo_transient_params *p, const oracle_convection_cell *in,
                            oracle_convection_cell *out, size_t H, size_t W, size_t iteration_offset,
                            size_t n_iterations, int n_threads) {
    const oracle_convection_cell halo = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; /* convection.cpp:42-56 */
    oracle_function f = {sizeof(oracle_convection_cell), 1, 3, 0, pseudo_transient_fn, NULL, p};
    return oracle_run(&f, in, out, H, W, &halo, iteration_offset, n_iterations, n_threads);
}

static void thermal_solver_fn(const void *vctx, const oracle_stencil *st, void *out) {
    /* convection.cpp:190-240 */
    const oracle_thermal_solver_params *k = (const oracle_thermal_solver_params *)vctx;
    oracle_convection_cell n = CC(0, 0);
    const size_t x = st->row, y = st->col, nx = k->nx, ny = k->ny;
    if (st->subiteration == 0) {
        if (x > 0 && y > 0 && x < nx - 1 && y < ny - 1) {
            double qTx_top_left = -k->DcT * (CC(0, 0).T - CC(-1, 0).T) / k->dx;
            double qTx_top = -k->DcT * (CC(1, 0).T - CC(0, 0).T) / k->dx;
            double qTy_top_left = -k->DcT * (CC(0, 0).T - CC(0, -1).T) / k->dy;
            double qTy_left = -k->DcT * (CC(0, 1).T - CC(0, 0).T) / k->dy;
            double dT_dt = -((qTx_top - qTx_top_left) / k->dx + (qTy_left - qTy_top_left) / k->dy);
            if (CC(0, 0).Vx > 0)
                dT_dt -= CC(0, 0).Vx * (CC(0, 0).T - CC(-1, 0).T) / k->dx;
            if (CC(1, 0).Vx < 0)
                dT_dt -= CC(1, 0).Vx * (CC(1, 0).T - CC(0, 0).T) / k->dx;
            if (CC(0, 0).Vy > 0)
                dT_dt -= CC(0, 0).Vy * (CC(0, 0).T - CC(0, -1).T) / k->dy;
            if (CC(0, 1).Vy < 0)
                dT_dt -= CC(0, 1).Vy * (CC(0, 1).T - CC(0, 0).T) / k->dy;
            n.T = CC(0, 0).T + dT_dt * k->dt;
        }
    } else if (st->subiteration == 1) {
        if (x == nx - 1 && y < ny)
            n.T = CC(-1, 0).T;
        if (x == 0 && y < ny)
            n.T = CC(1, 0).T;
    }
    *(oracle_convection_cell *)out = n;
}

int oracle_thermal_solver(const oracle_thermal_solver_params *p, const oracle_convection_cell *in,
                          oracle_convection_cell *out, size_t H, size_t W, size_t iteration_offset,
                          size_t n_iterations, int n_threads) {
    const oracle_convection_cell halo = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    oracle_function f = {sizeof(oracle_convection_cell), 1, 2, 0, thermal_solver_fn, NULL, p};
    return oracle_run(&f, in, out, H, W, &halo, iteration_offset, n_iterations, n_threads);
}
