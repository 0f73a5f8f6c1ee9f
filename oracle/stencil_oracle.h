/*
 * stencil_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the reference's generation sweep, used only as the
 * checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * Nothing under stencilstream_amd/ or include/ may include, link or call it.
 *
 * Pinning status: the sweep semantics (halo, iteration/sub-iteration/TDV
 * bookkeeping, non-square grids) are pinned by the reference's own
 * self-checking known-answer test (tests/TransFuncs.hpp:55-104 driven by
 * tests/StencilUpdateTest.hpp:30-63 with the four cases of
 * tests/cpu/StencilUpdate.cpp:35-41), restated here as oracle_selfcheck_*.
 * The numerical results of Jacobi/HotSpot/FDTD are pinned by NO test of the
 * reference ("parity unpinned" at application level); the
 * known answers recorded in SURVEY.md section 8c (reference cpu backend run by
 * the survey) are the numerical anchors for Jacobi, HotSpot and Conway, and
 * the frames written by the reference's unchanged examples/fdtd sources
 * (tests/golden/fdtd, six printed digits) anchor FDTD; all are checked in
 * tests/test_oracle_golden.py.  The reference itself needs a SYCL
 * implementation that this image lacks, so it is unbuildable here; only the
 * third-party Rodinia file examples/hotspot/hotspot_openmp.cpp builds from its
 * own source (oracle/Makefile target _ref/hotspot_openmp).
 *
 * What is restated, with the reference lines each piece follows:
 *   - driver loop + double buffering   StencilStream/cpu/StencilUpdate.hpp:109-142
 *   - one sweep                        StencilStream/cpu/StencilUpdate.hpp:185-223
 *   - stencil indexing (row, column)   StencilStream/Stencil.hpp:120-122,141-162
 */
#ifndef STENCIL_ORACLE_H
#define STENCIL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Everything a transition function sees for one cell (Stencil.hpp:165-180). */
typedef struct {
    size_t row, col;          /* stencil.id                                   */
    size_t grid_h, grid_w;    /* stencil.grid_range                           */
    size_t iteration;         /* iteration_offset + i                         */
    size_t subiteration;      /* 0 .. n_subiterations-1                       */
    const void *tdv;          /* stencil.time_dependent_value (may be NULL)   */
    size_t radius;            /* stencil radius                               */
    size_t cell_size;         /* bytes per cell                               */
    const unsigned char *nb;  /* (2r+1)x(2r+1) cells, row-major, NW origin    */
} oracle_stencil;

/* new_cell = f(ctx, stencil) */
typedef void (*oracle_transition_fn)(const void *ctx, const oracle_stencil *st, void *new_cell);
/* tdv = get_time_dependent_value(iteration); evaluated once per iteration */
typedef void (*oracle_tdv_fn)(const void *ctx, size_t iteration, void *tdv_out);

typedef struct {
    size_t cell_size;
    size_t radius;
    size_t n_subiterations;
    size_t tdv_size;               /* 0 = monostate */
    oracle_transition_fn transition;
    oracle_tdv_fn tdv;             /* may be NULL when tdv_size == 0 */
    const void *ctx;
} oracle_function;

/*
 * Run n_iterations generations (each n_subiterations sweeps) on an H x W
 * row-major AoS grid.  `in` is never written; `out` receives the final grid.
 * With n_iterations == 0, out is a copy of in.  Returns 0, or -1 on
 * allocation failure.  n_threads <= 1 runs scalar; otherwise OpenMP over rows
 * when compiled with -fopenmp.
 */
/* Window mode for the next runs: buffers are the window at (row0, col0) of a grid_h x grid_w grid (see
 * stencil_oracle.c); (0, 0, 0, 0) switches it off.  Process-wide, not thread-safe: tests only. */
void oracle_set_window(size_t row0, size_t col0, size_t grid_h, size_t grid_w);

int oracle_run(const oracle_function *f, const void *in, void *out, size_t H, size_t W,
               const void *halo_value, size_t iteration_offset, size_t n_iterations,
               int n_threads);

/* ---- Jacobi family: examples/jacobi/kernels.hpp:34-319 (Cell = float, r = 1) ---- */
enum {
    ORACLE_JACOBI1_GENERAL = 0,  /* :63-66   */
    ORACLE_JACOBI2_CONSTANT = 1, /* :95-98   */
    ORACLE_JACOBI3_CONSTANT = 2, /* :127-130 */
    ORACLE_JACOBI4_CONSTANT = 3, /* :159-162 */
    ORACLE_JACOBI5_CONSTANT = 4, /* :191-195 */
    ORACLE_JACOBI4_GENERAL = 5,  /* :229-233 */
    ORACLE_JACOBI5_GENERAL = 6,  /* :267-271 */
    ORACLE_JACOBI9_GENERAL = 7   /* :307-318 */
};
int oracle_jacobi(int variant, const float *coef, const float *in, float *out, size_t H,
                  size_t W, float halo, size_t iteration_offset, size_t n_iterations,
                  int n_threads);
/* dense 5 x 5 Jacobi, radius 2 (an extra, see stencil_oracle.c); coef[(dr+2)*5 + (dc+2)] */
int oracle_jacobi25(const float *coef, const float *in, float *out, size_t H, size_t W, float halo,
                    size_t n_iterations, int n_threads);
/* grid init of examples/jacobi/jacobi.cpp:111-124 */
void oracle_jacobi_init(float *grid, size_t H, size_t W);

/* ---- HotSpot: examples/hotspot/hotspot.cpp:57-97, constants :281-295 ---- */
typedef struct {
    float temp, power;
} oracle_hotspot_cell;
typedef struct {
    float Rx_1, Ry_1, Rz_1, Cap_1;
} oracle_hotspot_params;
void oracle_hotspot_params_for_grid(size_t n_rows, size_t n_columns, oracle_hotspot_params *p);
int oracle_hotspot(const oracle_hotspot_params *p, const oracle_hotspot_cell *in,
                   oracle_hotspot_cell *out, size_t H, size_t W, size_t iteration_offset,
                   size_t n_iterations, int n_threads);

/* the same formula in fp64 (an extra: the reference computes in fp32) */
typedef struct {
    double temp, power;
} oracle_hotspot_cell_f64;
typedef struct {
    double Rx_1, Ry_1, Rz_1, Cap_1;
} oracle_hotspot_params_f64;
int oracle_hotspot_f64(const oracle_hotspot_params_f64 *p, const oracle_hotspot_cell_f64 *in,
                       oracle_hotspot_cell_f64 *out, size_t H, size_t W, size_t iteration_offset,
                       size_t n_iterations, int n_threads);

/* ---- Conway: examples/conway/conway.cpp:35-56 (Cell = bool, halo = false) ---- */
int oracle_conway(const uint8_t *in, uint8_t *out, size_t H, size_t W, size_t n_iterations,
                  int n_threads);

/* ---- self-checking test function: tests/TransFuncs.hpp:33-104 ---- */
typedef struct {
    int32_t r, c, i_iteration, i_subiteration, status; /* status: 0 Normal, 1 Invalid, 2 Halo */
} oracle_selfcheck_cell;
int oracle_selfcheck(size_t radius, const oracle_selfcheck_cell *in, oracle_selfcheck_cell *out,
                     size_t H, size_t W, size_t iteration_offset, size_t n_iterations,
                     int n_threads);

/* ---- FDTD, coefficient resolver: examples/fdtd/src/Kernel.hpp:52-141,
 *      material/CoefResolver.hpp:24-68 ---- */
typedef struct {
    float ex, ey, hz, hz_sum, ca, cb, da, db;
} oracle_fdtd_cell;
typedef struct {
    float dt, t_0, tau, omega;
    uint64_t cutoff_iteration, detect_iteration;
    float source_radius_squared;
    float source_r, source_c, source_distance_bound;
    float double_center_rc;
} oracle_fdtd_params;
float oracle_fdtd_tdv(const oracle_fdtd_params *p, size_t iteration);
int oracle_fdtd(const oracle_fdtd_params *p, const oracle_fdtd_cell *in, oracle_fdtd_cell *out,
                size_t H, size_t W, size_t iteration_offset, size_t n_iterations, int n_threads);

/* ---- Convection: examples/convection/convection.cpp:36-242 (fp64, 11-field cell) ----
 * PseudoTransientKernel (three sub-iterations, :76-177) and ThermalSolverKernel (two, :179-242), restated with the
 * reference's expressions in the reference's order.  dimension 0 is "x" (rows), dimension 1 "y" (columns).  The
 * reference's tests hold no vector for them ("parity unpinned"); tests/test_oracle_golden.py checks this restatement
 * against the reference's own functor source on this repository's stencil::cpu (examples/convection_oracle_test.cpp). */
typedef struct {
    double T, Pt, Vx, Vy, tau_xx, tau_yy, sigma_xy, dVxd_tau, dVyd_tau, ErrV, ErrP;
} oracle_convection_cell;
typedef struct {
    size_t nx, ny;
    double roh0_g_alpha, delta_eta_delta_T, eta0, deltaT, dx, dy, delta_tau_iter, beta, rho, dampX, dampY, DcT;
} oracle_pseudo_transient_params;
typedef struct {
    size_t nx, ny;
    double dx, dy, dt, DcT;
} oracle_thermal_solver_params;
int oracle_pseudo_transient(const oracle_pseudo_transient_params *p, const oracle_convection_cell *in,
                            oracle_convection_cell *out, size_t H, size_t W, size_t iteration_offset,
                            size_t n_iterations, int n_threads);
int oracle_thermal_solver(const oracle_thermal_solver_params *p, const oracle_convection_cell *in,
                          oracle_convection_cell *out, size_t H, size_t W, size_t iteration_offset,
                          size_t n_iterations, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
