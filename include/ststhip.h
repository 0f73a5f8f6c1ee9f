/*
 * ststhip.h -- C ABI of the MI355X StencilUpdate backend (libststhip.so).
 *
 * Two layers, both `extern "C"`, plain pointers and sizes, int status codes
 * (0 = success, otherwise a ststhip_status; ststhip_last_error() describes the
 * last failure of the calling thread), no exceptions across the boundary.
 *
 * Layer 0 -- runtime services underneath the C++ templates.  In the reference
 * these are the implicit services of the SYCL runtime behind
 * cuda::StencilUpdate / cuda::Grid (queue creation and wait
 * StencilStream/cuda/StencilUpdate.hpp:124-135, buffer allocation :204-205 and
 * cuda/internal/Helpers.hpp:37-45, implicit host<->device copies behind
 * host_accessor cuda/Grid.hpp:145-153, event profiling :184-198).
 *
 * Layer 1 -- the generation sweep itself for the precompiled transition
 * functions (the five example applications and the reference's self-checking
 * test function).  One call of ststhip_app_sweep() replaces the body of
 * cuda::StencilUpdate::run_simulation's loop (StencilStream/cuda/
 * StencilUpdate.hpp:212-273 AoS, :324-405 SoA) for `n_generations`
 * generations at once (temporal blocking); ststhip_app_run() replaces
 * cuda::StencilUpdate::operator() (:123-144).  This is the surface a foreign
 * language binds (ctypes stub in INTEGRATION.md; stencilstream_amd/capi.py).
 *
 * Transition functions that only exist as C++ templates in the user's
 * translation unit are compiled there (StencilStream/hip/StencilUpdate.hpp)
 * and use layer 0 only.
 */
#ifndef STSTHIP_H
#define STSTHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STSTHIP_ABI_VERSION 6

typedef enum {
    STSTHIP_OK = 0,
    STSTHIP_ERR_HIP = 1,          /* a HIP runtime call failed               */
    STSTHIP_ERR_INVALID = 2,      /* bad argument                            */
    STSTHIP_ERR_UNKNOWN_APP = 3,  /* no such precompiled transition function */
    STSTHIP_ERR_NO_DEVICE = 4,    /* no usable GPU                           */
    STSTHIP_ERR_COMM = 5          /* RCCL failure                            */
} ststhip_status;

typedef void *ststhip_stream; /* a hipStream_t; NULL = the runtime's own stream */
typedef void *ststhip_event;  /* a hipEvent_t */

/* ------------------------------------------------------------- layer 0 */
int ststhip_abi_version(void);
const char *ststhip_last_error(void);
/* Lets code on the caller's side of a callback (ststhip_sweep_fn) report why it failed. */
void ststhip_set_last_error(const char *message);

/* Select a GPU for the calling process (device < 0: keep the current HIP device) and create the
 * runtime's stream.  Idempotent.  Fails with STSTHIP_ERR_NO_DEVICE when no GPU is visible. */
int ststhip_init(int device);
int ststhip_shutdown(void);
int ststhip_device_count(int *count);
int ststhip_device_name(char *buf, size_t buf_size);
int ststhip_compute_units(int *count);

/* Device memory from a size-bucketed pool (freed blocks are reused, not returned to HIP).
 * Releases are stream-ordered: ststhip_free_async(ptr, stream) may be called while work that uses the
 * block is still queued on `stream`; the block is handed out again to the same stream at once, to any
 * other stream after a wait on the release event, and to ststhip_malloc (host order) once that event has
 * completed.  ststhip_free(ptr) = ststhip_free_async(ptr, NULL): release ordered after the runtime's own
 * stream.  A block that other streams still use must be released on a stream that was joined with them. */
int ststhip_malloc(void **ptr, size_t bytes);
int ststhip_free(void *ptr);
int ststhip_malloc_async(void **ptr, size_t bytes, ststhip_stream stream);
int ststhip_free_async(void *ptr, ststhip_stream stream);
int ststhip_pool_trim(void);
/* Pinned host memory for grids' host mirrors.  Freed blocks are kept for reuse (pinning is slow) up
 * to STSTHIP_HOST_CACHE_MIB (default 4096) MiB; ststhip_pool_trim releases them. */
int ststhip_host_malloc(void **ptr, size_t bytes);
int ststhip_host_free(void *ptr);

/* Asynchronous on `stream`; the host buffer must stay alive until the stream is synchronised. */
int ststhip_memcpy_h2d(void *dst, const void *src, size_t bytes, ststhip_stream stream);
int ststhip_memcpy_d2h(void *dst, const void *src, size_t bytes, ststhip_stream stream);
int ststhip_memcpy_d2d(void *dst, const void *src, size_t bytes, ststhip_stream stream);
int ststhip_memset(void *dst, int value, size_t bytes, ststhip_stream stream);

int ststhip_default_stream(ststhip_stream *stream);
int ststhip_stream_create(ststhip_stream *stream);
int ststhip_stream_destroy(ststhip_stream stream);
int ststhip_stream_synchronize(ststhip_stream stream);
int ststhip_stream_wait_event(ststhip_stream stream, ststhip_event event);

int ststhip_event_create(ststhip_event *event);
int ststhip_event_destroy(ststhip_event event);
int ststhip_event_record(ststhip_event event, ststhip_stream stream);
int ststhip_event_synchronize(ststhip_event event);
int ststhip_event_elapsed_ms(ststhip_event start, ststhip_event stop, float *ms);

/* Launch a kernel that lives in the caller's code object (`function` = host-side kernel symbol).
 * Used by the C++ templates so that launches share the runtime's error handling and stream. */
int ststhip_launch(const void *function, unsigned grid_x, unsigned grid_y, unsigned grid_z,
                   unsigned block_x, unsigned block_y, unsigned block_z, void **args,
                   size_t shared_bytes, ststhip_stream stream);

/* Workgroups of `block_threads` threads of kernel `function` that one CU keeps resident (HIP
 * occupancy query); the sweep launcher sizes its wave grid in whole residency rounds with it. */
int ststhip_occupancy(const void *function, unsigned block_threads, size_t shared_bytes,
                      int *blocks_per_cu);

/* Bytes of scratch (private memory: register spills, dynamically indexed locals) per work-item of kernel
 * `function`, from the code object.  The C++ templates use it to leave out temporal-blocking depths whose kernel the
 * compiler could only build with spills (hip/StencilUpdate.hpp: a user's transition function may need far more
 * registers than its cell size suggests). */
int ststhip_kernel_scratch_bytes(const void *function, size_t *bytes_per_work_item);

/* How many sweep launches the pass driver currently keeps in flight side by side (1 outside of
 * ststhip_run_passes).  The launcher sizes its row chunks with it: concurrent launches share the
 * chip and hide each other's tails, so longer chunks (fewer warm-up rows) pay. */
int ststhip_launch_concurrency(void);
/* 1 while the pass driver is in a pass whose target planes were already written, row for row, by the pass
 * before the previous one of the same call (the third pass of a run and every later one); 0 otherwise.  The
 * sweep launcher then leaves out the stores of the per-field planes a transition function declares constant
 * (F::constant_fields): they already hold those values. */
int ststhip_target_holds_constants(void);
/* The device table of time-dependent values the pass driver has built for the call it is in (ststhip_sweep_desc:
 * tdv_size / fill_tdv / tdv_device_table), for the calling thread: `base` = value of generation `first_iteration`,
 * `n_values` of `value_size` bytes; base = NULL outside of such a call.  The sweep launcher points the kernel at it
 * instead of evaluating the launch's values on the host (tdv::single_pass strategies, hip/StencilUpdate.hpp). */
int ststhip_current_tdv_table(const void **base, uint64_t *first_iteration, uint64_t *n_values,
                              uint64_t *value_size);
/* A host that runs row-range sweeps side by side itself (the multi-GPU strip driver,
 * stencilstream_amd/dist.py) states their number here for the calling thread; 1 resets it. */
int ststhip_set_launch_concurrency(int n_launches_side_by_side);
/* A hole in the row range of the calling thread's next sweep launches: rows [*begin, *end) are left out, the
 * launch produces the rows on both sides of them (begin == end: no hole, the state outside of a driver).  The
 * strip driver sweeps the two boundary bands of a strip as one launch this way -- one dependent chain of band
 * latency per pass instead of two (ststhip_strip_advance).  The setter is for hosts that drive ststhip_app_sweep
 * themselves: set it, launch over [top of the upper band, end of the lower band), reset it with (0, 0). */
int ststhip_launch_row_hole(uint64_t *begin, uint64_t *end);
int ststhip_set_launch_row_hole(uint64_t begin, uint64_t end);
/* The global columns [begin, end) the calling thread's next sweep launches produce (begin == end: all columns the
 * buffers hold inside the grid, the state outside of a driver).  The block driver (ststhip_block_*) names the owned
 * columns of a block, widened by what the rest of a launch group still needs, this way; the row range stays an
 * argument of the launch.  Every launch through hip/internal/Sweep.hpp's launch_sweep honours it. */
int ststhip_launch_columns(uint64_t *begin, uint64_t *end);
int ststhip_set_launch_columns(uint64_t begin, uint64_t end);
/* Into how many row strips (1 or 2) a caller that advances `rows` x `width` cells of `app` for
 * `n_passes` launches should split them, each strip on its own stream and coupled to its
 * neighbours through halo-deep boundary bands only: the rule ststhip_run_passes applies to a whole
 * grid, for callers that drive ststhip_app_sweep themselves.  Returns 1 for unknown apps. */
int ststhip_suggest_row_strips(const char *app, uint64_t rows, uint64_t width, uint64_t n_passes);

/* Tuning knobs of the launcher and the drivers.  They come from the environment (STSTHIP_<NAME>, upper case) ONCE,
 * when the runtime first needs them (ststhip_init), never per launch; a host that changes its environment afterwards
 * (tools, tests) calls ststhip_reload_options().  0 / -1 = "not set, use the built-in rule" where noted.  Every knob
 * changes speed only, never results. */
typedef struct {
    int32_t chunk_rows;            /* rows of output per wave; 0 = the launcher's rule                    */
    int32_t tail_permille;         /* weight of a launch's ragged tail in that rule; 0 = built-in         */
    int32_t n_taper;               /* entries of STSTHIP_TAPER="permille:split,..."; -1 = built-in        */
    int32_t taper_permille[3], taper_split[3];
    int32_t narrow_form_kcells;    /* grids up to this many thousand cells run the one-cell-per-lane form */
    int32_t narrow_band_rows;      /* so do launches of at most this many rows (0 = never)                */
    int32_t skip_constant_stores;  /* leave out the stores of F::constant_fields where the target holds them */
    int32_t xcd_remap, last_chunk_early;
    int32_t max_generations;       /* cap of the temporal-blocking depth; 0 = the compiled maximum        */
    int32_t allow_spilling_depths; /* C++ templates: also use depths whose kernel spills registers        */
    int32_t virtual_strips;        /* row strips of the pass driver; 0 = its rule                         */
    int32_t two_strips_permille, two_strips_permille_outer; /* thresholds of that rule                    */
    int32_t strip_skew_permille;   /* where the pass driver cuts two strips; 0 = built-in                 */
    int32_t bands_beside_interior, band_stream_priority, bands_apart, bands_one_launch, comm_stream_priority;
    int32_t jacobi_fastpath, conway_fastpath;
    int32_t prepare_streams;       /* create and first-use the pass driver's streams at ststhip_init      */
    int32_t host_cache_mib;        /* free pinned host blocks kept for reuse                              */
    int32_t tune_depth;            /* pass driver, families with two candidate depths (ststhip_sweep_desc::alt_generations):
                                      1 (default) = time both on the first long call for a grid shape and keep the faster;
                                      0 = always the trusted depth; N >= 2 = depth N outright                         */
    int32_t exchange_every;        /* strip driver: exchange m*g ghost rows every m-th launch; 0/1 = every launch */
    int32_t stream_upload;         /* 1 (default) = a grid that is not in HBM yet is uploaded in row blocks and the pass
                                      driver starts on the blocks that have arrived (ststhip_set_source_arrival); 0 = one
                                      copy in front of the first pass                                                   */
    int32_t upload_block_mib;      /* size of those blocks; 0 = the rule of ststhip_suggest_upload_blocks              */
    int32_t skewed_strips;         /* pass driver, two row strips: 1 (default) = their common boundary moves up by a launch's
                                      ghost rows from pass to pass (no boundary bands: one launch per strip and pass, the
                                      upper strip never waits for the lower one); 0 = fixed strips with boundary bands on
                                      streams of their own (rounds 1-3)                                                   */
    int32_t strip_substrips;       /* strip driver: a rank's strip as two sub-strips with a moving boundary (two launches in
                                      flight); -1 (default) = the rule of ststhip_suggest_row_strips on the strip's cells,
                                      1 = never, 2 = wherever the strip is tall enough                                  */
    int32_t reserved[3];
} ststhip_options;
const ststhip_options *ststhip_get_options(void);
int ststhip_reload_options(void);

/* AoS <-> per-field planes by byte geometry (the reference's scatter/gather kernels,
 * StencilStream/cuda/StencilUpdate.hpp:294-321 and :408-438).  Field f of cell i is the
 * `field_size[f]` bytes at aos + i*cell_size + field_offset[f]; plane f holds them densely.
 * Staged through LDS so that both the AoS side and the plane side move whole cache lines. */
int ststhip_scatter_fields(const void *aos, size_t cell_size, size_t n_cells, int n_fields,
                           const size_t *field_offset, const size_t *field_size,
                           void *const *planes, ststhip_stream stream);
int ststhip_gather_fields(void *aos, size_t cell_size, size_t n_cells, int n_fields,
                          const size_t *field_offset, const size_t *field_size,
                          const void *const *planes, ststhip_stream stream);

/* Device-side maximum of |field| over a sub-rectangle of an AoS grid, for up to 8 fields in ONE pass over the
 * cells: what an application's convergence check needs from a grid without bringing the cells to the host
 * (replaces the host scan of examples/convection/convection.cpp:412-438; an extension -- the reference's API has
 * no reduction).  Field i is the f32 / f64 at byte `offset` of every cell; only cells with row < row_limit and
 * column < col_limit count.  result[i] = max |value| (NaNs are skipped, as `std::abs(v) > max` skips them), or
 * -infinity when no cell counts.  Wave-level DPP reduction, one atomic per wave and field.  Synchronises
 * `stream` and writes `result` (host memory) before returning. */
#define STSTHIP_F32 0u
#define STSTHIP_F64 1u
typedef struct {
    uint32_t offset;
    uint32_t type; /* STSTHIP_F32 or STSTHIP_F64 */
    uint64_t row_limit;
    uint64_t col_limit;
} ststhip_reduce_field;
int ststhip_reduce_max_abs(const void *cells, size_t cell_size, uint64_t height, uint64_t width,
                           uint64_t pitch, int n_fields, const ststhip_reduce_field *fields,
                           double *result, ststhip_stream stream);

/* ------------------------------------------------------------- layer 1 */

/* Where a buffer sits inside the global grid.  A single-GPU grid has row_origin = 0 and
 * local_rows = global_height.  A row strip of a domain-decomposed grid holds its owned rows plus
 * ghost rows: buffer row 0 is global row `row_origin` (may be "negative" for the first strip:
 * rows before global row 0 are never read). */
typedef struct {
    uint64_t global_height; /* stencil.grid_range[0]                              */
    uint64_t global_width;  /* stencil.grid_range[1]                              */
    int64_t row_origin;     /* global row index of buffer row 0                   */
    uint64_t local_rows;    /* rows held by the buffers                           */
    uint64_t pitch;         /* elements between consecutive rows (>= the columns held) */
    /* ABI 5: a block of a 2-D decomposition holds a column range too.  local_cols = 0: whole rows (columns
     * 0 .. global_width, the case of everything above); else buffer column 0 is global column `col_origin` (may be
     * negative for the first block of a row: columns before global column 0 are never read) and the buffers hold
     * `local_cols` columns: the owned ones plus ghost columns on both sides. */
    int64_t col_origin;
    uint64_t local_cols;
} ststhip_domain;

/* Static description of a precompiled transition function. */
typedef struct {
    const char *name;
    uint32_t cell_size;          /* sizeof(Cell) of the AoS cell                         */
    uint32_t params_size;        /* sizeof of the transition function's parameter block  */
    uint32_t stencil_radius;
    uint32_t n_subiterations;
    uint32_t n_planes;           /* 1 = AoS sweep; >1 = one plane per field (SoA sweep)  */
    uint32_t plane_elem_size[16];
    uint32_t field_offset[16];   /* byte offset of each plane's field inside the cell    */
    uint32_t max_generations;    /* deepest temporal blocking compiled in                */
    uint32_t tdv_size;           /* 0 = no time-dependent value                          */
    uint32_t halo_depth_per_generation; /* ghost rows one generation consumes per side   */
    uint32_t strip_width;        /* columns one wavefront produces at max_generations (staged sweeps: its share of the workgroup's strip) */
    uint32_t cells_per_lane;     /* adjacent cells a lane holds per row (K)              */
    uint32_t prefetch_rows;      /* rows loaded ahead of the pipeline (P)                */
    uint32_t stages;             /* waves of a workgroup that share one column strip as a pipeline over the levels (1 = independent waves) */
    uint32_t default_generations; /* depth of launches nobody has measured (= max_generations unless the function's
                                     tuning compiles a deeper family than it trusts: ststhip_sweep_desc::alt_generations) */
} ststhip_app_info;

/* Largest scratch (private memory: register spills, dynamically indexed locals) per work-item over the kernels a
 * launch of `app` at depth `n_generations` may start (default shape, narrow form, constant-plane variant), from the
 * code object.  The shipped library has none at any depth (tests/test_parity_gpu.py asserts it). */
int ststhip_app_scratch_bytes(const char *app, uint32_t n_generations, size_t *bytes_per_work_item);

int ststhip_app_count(void);
int ststhip_app_info_at(int index, ststhip_app_info *info);
int ststhip_app_find(const char *name, ststhip_app_info *info);

/* Advance global rows [out_row_begin, out_row_end) by `n_generations` (max_generations or one of its
 * repeated halvings) generations in ONE kernel launch, reading `src` and writing `dst` (arrays of
 * n_planes device pointers with identical geometry `dom`).  Rows of `dst` outside the range are
 * not touched.  Input rows [out_row_begin - g, out_row_end + g) that lie inside the global grid
 * must be present in `src`, g = n_generations * halo_depth_per_generation.  `tf_params` is the
 * host-side parameter block of the transition function, `halo_cell` one AoS cell.  The
 * time-dependent values of generations `iteration .. iteration+n-1` are evaluated on the host
 * inside this call, once each (reference: StencilStream/cuda/StencilUpdate.hpp:224). */
int ststhip_app_sweep(const char *app, const void *tf_params, const void *halo_cell,
                      const ststhip_domain *dom, const void *const *src, void *const *dst,
                      uint64_t out_row_begin, uint64_t out_row_end, uint64_t iteration,
                      uint32_t n_generations, ststhip_stream stream);

/* Timing of one ststhip_app_run call. */
typedef struct {
    double walltime_s;        /* host wall clock around the whole call (incl. final sync)      */
    double kernel_time_s;     /* sum of sweep-kernel durations from HIP events (if profiling)  */
    uint64_t n_launches;      /* sweep kernels launched                                         */
    uint64_t n_processed_cells; /* n_iterations * H * W (sub-iterations not counted)           */
    uint64_t n_streamed_passes; /* ABI 6: passes that ran as row tiles behind a source still arriving
                                   (ststhip_set_source_arrival); 0 otherwise                              */
} ststhip_run_info;

/* cuda::StencilUpdate::operator() for a precompiled transition function: advance the whole
 * grid by n_iterations generations.  `src` is only read; `dst` receives the result; both are
 * arrays of n_planes device pointers of geometry `dom` (row_origin 0, local_rows = height).
 * Scratch planes come from the runtime's pool.  blocking != 0 synchronises the stream before
 * returning; profiling != 0 brackets every sweep kernel with HIP events.  n_iterations == 0
 * copies src to dst.
 * Two applications have a second, bit-identical form the call switches to on its own:
 * "jacobi5general" with five equal positive coefficients and a +0 halo (product-carrying form),
 * and "conway" with halo = false, width and pitch multiples of four and 4-byte aligned planes
 * (four cells per 32-bit word).  Cells of "conway" are C++ bools: bytes 0 or 1, nothing else. */
int ststhip_app_run(const char *app, const void *tf_params, const void *halo_cell,
                    const ststhip_domain *dom, const void *const *src, void *const *dst,
                    uint64_t iteration_offset, uint64_t n_iterations, int blocking,
                    int profiling, ststhip_stream stream, ststhip_run_info *info);

/* The pass driver behind ststhip_app_run and stencil::hip::StencilUpdate: splits n_iterations into
 * launches of the compiled blocking depths, ping-pongs between `dst` and pooled scratch planes so
 * that the last pass lands in `dst` (`src` is never written), and -- for tall grids -- advances
 * two row strips on separate streams, so that the tail of one launch overlaps with the next
 * launches: strips whose common boundary moves up by a launch's ghost rows from pass to pass
 * (ststhip_options::skewed_strips; the upper strip then never waits for the lower one), or fixed
 * strips coupled through boundary bands.  `sweep` performs one launch (for C++ transition
 * functions it is instantiated in the user's translation unit). */
typedef int (*ststhip_sweep_fn)(void *ctx, const ststhip_domain *dom, const void *const *src,
                                void *const *dst, uint64_t out_row_begin, uint64_t out_row_end,
                                uint64_t iteration, uint32_t n_generations, ststhip_stream stream);
typedef struct {
    uint32_t n_planes;
    uint32_t max_generations;           /* deepest blocking `sweep` accepts (powers of two up to it) */
    uint32_t halo_depth_per_generation; /* radius * n_subiterations */
    uint32_t strip_width;               /* columns one wave produces at max_generations (0 = unknown) */
    uint64_t plane_elem_size[16];
    /* Time-dependent values (0 / NULL: the sweep callback provides them per launch).  The driver builds ONE device
     * table per call -- element i = value of generation iteration_offset + i -- either by calling fill_tdv (host
     * evaluation, once per generation as the reference's backends do, cuda/StencilUpdate.hpp:224) and uploading
     * it, or by taking `tdv_device_table` (n_iterations values the caller has already computed on the device). */
    uint64_t tdv_size;
    void (*fill_tdv)(void *ctx, uint64_t iteration_offset, uint64_t n_iterations, void *values);
    const void *tdv_device_table;
    /* Depth by measurement (ABI 5; 0 / 0: none).  The deepest compiled depth is not the fastest for every function and
     * grid: what a transition function costs per cell is opaque to the rule that sizes the pipeline from the cell (the
     * same five-point Jacobi source is nine operations per cell compiled with -ffp-contract=off and five with fused
     * multiply-adds: the first is fastest at 8 generations per launch, the second at 16), and a small grid pays more
     * for deep halos than a large one.  `alt_generations` names a second depth `sweep` accepts (a repeated halving of
     * max_generations) and is the depth of every launch nobody has measured; on the first call for a grid shape that is
     * long enough (>= 6 launches of max_generations) the pass driver times two launches of max_generations against the
     * same generations at alt_generations -- these are the call's own first passes, no work is repeated --, keeps the
     * faster for the rest of the call and, per (tune_key, height, width), for the process
     * (ststhip_tuned_depth reads the choice).  Results do not depend on it.  `tune_key` identifies the kernel family
     * (e.g. the address of the launch callback). */
    uint32_t alt_generations;
    uint32_t reserved;
    uint64_t tune_key;
} ststhip_sweep_desc;
int ststhip_run_passes(ststhip_sweep_fn sweep, void *ctx, const ststhip_sweep_desc *desc,
                       const ststhip_domain *dom, const void *const *src, void *const *dst,
                       uint64_t iteration_offset, uint64_t n_iterations, int blocking, int profiling,
                       ststhip_stream stream, ststhip_run_info *info);
/* A source that is still on its way (ABI 6).  A grid's first update pays for its upload: 1 GiB takes 18.7 ms over PCIe,
 * a quarter of a 1000-generation run of the Jacobi example, and in front of the first pass the chip idles for all of
 * it.  A host that uploads in ROW BLOCKS (in order, first rows first) names them here for the calling thread's NEXT
 * ststhip_run_passes / ststhip_app_run call: rows [blocks[i-1].row_end, blocks[i].row_end) of every source plane are in
 * HBM once event blocks[i].ready has completed (the last row_end >= the grid's height; the call consumes the list
 * whatever it returns).  The driver then starts on what has arrived: it runs its first passes as row tiles skewed in
 * time -- pass p of the rows above block boundary b reaches g*(p+1) rows less far down than the boundary, g = ghost
 * rows of one launch, so that every tile depends only on tiles of rows that arrived earlier --, as many passes deep
 * as it takes to keep the chip busy between two arrivals (measured per call: the first tile column is timed against
 * the gap between the arrival events), completes those passes over the whole grid when the last block is there and
 * continues with whole-grid passes.  The targets alternate exactly as without the list, the result is the same bit for
 * bit.  The host thread waits for the arrival events inside the call (also with blocking = 0).  `stream` must NOT be
 * made to wait for the upload by the caller; when the driver cannot use the list (profiling runs, grids of few rows per
 * block, options.stream_upload = 0) it makes `stream` wait for every block itself. */
typedef struct {
    uint64_t row_end;
    ststhip_event ready;
} ststhip_source_block;
int ststhip_set_source_arrival(const ststhip_source_block *blocks, uint32_t n_blocks);
/* The rule for such an upload: into how many row blocks (1 = do not split) a host should divide `rows` rows of
 * `row_bytes` bytes (all planes together), and the streams the runtime keeps for it: `copies` for the transfers
 * (ordered against nothing else) and `work` for kernels a host runs per block behind its copy (the scatter into
 * per-field planes; an event per block carries the order: a kernel queued on the copies' own stream holds up the next
 * transfer until it has found room on the chip).  Both are streams of the pass driver that idle while a source
 * arrives -- its tiles run on the caller's stream and the driver's side stream. */
int ststhip_suggest_upload_blocks(uint64_t rows, uint64_t row_bytes, uint32_t *n_blocks);
int ststhip_upload_streams(ststhip_stream *copies, ststhip_stream *work);

/* The depth the pass driver has measured to be the faster one for (tune_key, height, width) in this process; 0 = not
 * measured yet. */
int ststhip_tuned_depth(uint64_t tune_key, uint64_t height, uint64_t width, uint32_t *depth);
/* The same for a precompiled transition function (its kernel family's key is internal to the registry). */
int ststhip_app_tuned_depth(const char *app, uint64_t height, uint64_t width, uint32_t *depth);

/* Parameter blocks of the precompiled transition functions (plain data, host side). */
typedef struct {
    float coef[9]; /* as many as the variant takes; Jacobi9General: coef[r*3+c] */
} ststhip_jacobi_params;
typedef struct {
    float coef[25]; /* "jacobi25general": dense 5 x 5, radius 2, coef[(dr+2)*5 + (dc+2)] */
} ststhip_jacobi25_params;
typedef struct {
    float Rx_1, Ry_1, Rz_1, Cap_1;
} ststhip_hotspot_params;
typedef struct {
    double Rx_1, Ry_1, Rz_1, Cap_1;
} ststhip_hotspot_params_f64; /* "hotspot_f64" / "hotspot_f64_aos": the same formula in fp64 (an extra) */
typedef struct {
    float dt, t_0, tau, omega;
    uint64_t cutoff_iteration, detect_iteration;
    float source_radius_squared;
    float source_r, source_c, source_distance_bound;
    float double_center_rc;
    uint32_t reserved;
} ststhip_fdtd_params;

/* ------------------------------------------------- multi-GPU ghost exchange */
/* One process per GPU.  The unique id is created on rank 0 and distributed by the host program
 * (torch.distributed / MPI / a file); every rank then joins.  Exchange = grouped ncclSend/ncclRecv
 * with the upper (rank-1) and lower (rank+1) neighbour over the direct xGMI links. */
#define STSTHIP_COMM_ID_BYTES 128
typedef void *ststhip_comm;
int ststhip_comm_unique_id(unsigned char id[STSTHIP_COMM_ID_BYTES]);
int ststhip_comm_create(const unsigned char id[STSTHIP_COMM_ID_BYTES], int rank, int n_ranks,
                        ststhip_comm *comm);
int ststhip_comm_destroy(ststhip_comm comm);
/* The ranks this rank exchanges ghost rows with: `up` holds the rows above its own, `down` the rows below; -1 = no
 * neighbour on that side.  Default after ststhip_comm_create: a chain, rank-1 above and rank+1 below.  Other wirings
 * are for hosts whose strips are not numbered like their ranks (meshes), for rings -- both sides may name the same
 * rank, the messages are posted so that its upper rows arrive below and its lower rows above -- and for a rank that
 * names ITSELF on both sides: a self send/receive inside one group is legal in RCCL, which lets one GPU run the whole
 * exchange path (dlopen'ed symbols, byte counts, pointer arithmetic, stream order) without a second device. */
int ststhip_comm_set_neighbours(ststhip_comm comm, int up, int down);
int ststhip_comm_neighbours(ststhip_comm comm, int *up, int *down);
/* The same for the ghost COLUMNS of a 2-D block decomposition (ststhip_block_*): `left` holds the columns before this
 * rank's, `right` those after; default -1 / -1.  ststhip_comm_set_mesh wires all four sides for ranks laid out
 * row-major on a mesh_rows x mesh_cols mesh (rank = mesh_row * mesh_cols + mesh_col; mesh_rows * mesh_cols = n_ranks). */
int ststhip_comm_set_column_neighbours(ststhip_comm comm, int left, int right);
int ststhip_comm_set_mesh(ststhip_comm comm, int mesh_rows, int mesh_cols);
/* For every plane p: send `n_rows` rows starting at send_up[p] to the upper neighbour (rank-1 unless set otherwise)
 * and at send_down[p] to the lower one (rank+1), receive into recv_up[p] (from the upper neighbour) and recv_down[p]
 * (from the lower one).  row_bytes[p] = bytes of one row of plane p.  Ranks without a neighbour on a side skip it. */
int ststhip_comm_exchange_rows(ststhip_comm comm, int n_planes, const void *const *send_up,
                               const void *const *send_down, void *const *recv_up,
                               void *const *recv_down, const size_t *row_bytes, size_t n_rows,
                               ststhip_stream stream);

/* For every plane p: send the contiguous block of block_bytes[p] bytes at send_left[p] to the left neighbour and the
 * one at send_right[p] to the right one, receive into recv_left[p] / recv_right[p] (the packed ghost columns of a 2-D
 * block: the block driver packs and unpacks them with small copy kernels). */
int ststhip_comm_exchange_columns(ststhip_comm comm, int n_planes, const void *const *send_left,
                                  const void *const *send_right, void *const *recv_left, void *const *recv_right,
                                  const size_t *block_bytes, ststhip_stream stream);

/* ------------------------------------------------- row-strip driver (one strip per process / GPU)
 * The grid is cut into n_ranks strips of consecutive rows; every process owns one strip on its GPU and keeps it,
 * with ghost rows, in two buffer sets inside the library.  ststhip_strip_advance() is cuda::StencilUpdate::operator()
 * for the whole distributed grid.  Its launches go in groups of m = STSTHIP_EXCHANGE_EVERY (default: 4 for strips
 * of up to 4096 rows, else 2) with ONE exchange of m*T*radius*n_subiterations ghost rows per group (RCCL
 * send/recv with the two neighbours over xGMI on a second stream, no collective): inside a group every launch
 * produces the owned rows widened by the ghost depth the rest of the group still needs; the last launch of a group
 * sweeps the rows next to the neighbours first (one band launch), hands them to the exchange for the next group, and
 * sweeps its interior beside both.  No host code between the launches of a call: a C++ (or any FFI)
 * host calls create / upload / advance / download.  The semantics are those of one StencilUpdate on the whole
 * grid (the reference has no spatial decomposition; role model for "same interface, several devices":
 * StencilStream/monotile/StencilUpdate.hpp:166-227).
 *
 * `comm`: a communicator of exactly the n_ranks strips (ststhip_comm_create), rank r = strip r.  Hosts whose ranks
 * cannot be joined by RCCL (tests: several ranks on one GPU) pass comm = NULL and an `exchange` callback with the
 * contract of ststhip_comm_exchange_rows: it must have filled recv_up / recv_down in stream order of `stream`
 * (a callback that stages through host memory synchronises `stream` itself). */
typedef void *ststhip_strip;
typedef int (*ststhip_exchange_fn)(void *ctx, int n_planes, const void *const *send_up,
                                   const void *const *send_down, void *const *recv_up,
                                   void *const *recv_down, const size_t *row_bytes, size_t n_rows,
                                   ststhip_stream stream);
int ststhip_strip_create(const char *app, const void *tf_params, const void *halo_cell, uint64_t total_rows,
                         uint64_t width, int rank, int n_ranks, ststhip_comm comm, ststhip_exchange_fn exchange,
                         void *exchange_ctx, ststhip_strip *strip);
/* The same strip for a sweep that is not in the registry: the launch callback and description a caller would hand to
 * ststhip_run_passes (the C++ templates instantiate the kernel for a user's transition function in their own
 * translation unit: StencilStream/hip/StripUpdate.hpp).  The buffers are the planes `desc` describes. */
/* The `sweep` callback of a custom strip MUST honour the calling thread's row hole (ststhip_launch_row_hole): the
 * driver sweeps both boundary bands of a strip as one launch over [first owned row, last owned row) with the interior
 * left out as the hole, on a highest-priority stream, while another launch sweeps the interior.  Every launch that
 * goes through the C++ templates' launcher (hip/internal/Sweep.hpp: launch_sweep) does; a callback that ignored the
 * hole would sweep the interior twice (same values, doubled work, no band / exchange overlap). */
int ststhip_strip_create_custom(ststhip_sweep_fn sweep, void *ctx, const ststhip_sweep_desc *desc, uint64_t total_rows,
                                uint64_t width, int rank, int n_ranks, ststhip_comm comm,
                                ststhip_exchange_fn exchange, void *exchange_ctx, ststhip_strip *strip);
int ststhip_strip_destroy(ststhip_strip strip);
/* global rows [row_begin, row_end) this strip owns */
int ststhip_strip_rows(ststhip_strip strip, uint64_t *row_begin, uint64_t *row_end);
/* Device pointer to the first owned row of plane `plane` of the CURRENT buffer set (it changes with every
 * advance) and the bytes per row; rows are dense.  Upload / download with ststhip_memcpy_* on the strip's stream. */
int ststhip_strip_plane(ststhip_strip strip, unsigned plane, void **owned_rows, size_t *row_bytes);
int ststhip_strip_stream(ststhip_strip strip, ststhip_stream *stream);
int ststhip_strip_synchronize(ststhip_strip strip);
/* One ghost exchange outside any timed region (RCCL sets up its channels on first use). */
int ststhip_strip_warm_up(ststhip_strip strip);
/* Advance the whole distributed grid by n_generations generations (every rank calls it with the same arguments). */
int ststhip_strip_advance(ststhip_strip strip, uint64_t iteration_offset, uint64_t n_generations, int blocking);
int ststhip_strip_counters(ststhip_strip strip, uint64_t *n_launches, uint64_t *n_exchanges);

/* ------------------------------------------------- 2-D block decomposition (one block per process / GPU)
 * The grid is cut into mesh_rows x mesh_cols blocks, rank = mesh_row * mesh_cols + mesh_col.  Every process keeps its
 * block with ghost rows AND ghost columns in two buffer sets; the handle is an ststhip_strip: ststhip_strip_advance /
 * _synchronize / _warm_up / _counters / _destroy / _stream work on it.  Per group of launches (STSTHIP_EXCHANGE_EVERY)
 * the ghost cells travel in two phases on the comm stream -- columns first (strided in memory: packed into contiguous
 * staging buffers by a copy kernel, sent, unpacked), then rows over the full buffer width, ghost columns included, which
 * carries the corners --, then every launch of the group sweeps the owned block widened by what the rest of the group
 * still needs.  The minimal form of the reference's tile geometry (StencilStream/tiling/Grid.hpp:305-450): no overlap of
 * exchange and interior (a block's boundary is a frame, not two bands).  mesh_cols = 1 gives the row strips above
 * without their overlap; use ststhip_strip_create for those.
 * `comm`: a communicator of the mesh's ranks whose neighbours are wired for the mesh (ststhip_comm_set_mesh), or
 * NULL and the two callbacks (contract of ststhip_comm_exchange_rows; the column callback gets the packed blocks as
 * "one row" of block bytes: send_up = send_left, send_down = send_right, ...). */
int ststhip_block_create(const char *app, const void *tf_params, const void *halo_cell, uint64_t total_rows,
                         uint64_t total_cols, int rank, int mesh_rows, int mesh_cols, ststhip_comm comm,
                         ststhip_exchange_fn exchange_rows, void *exchange_rows_ctx,
                         ststhip_exchange_fn exchange_cols, void *exchange_cols_ctx, ststhip_strip *block);
/* The same for a sweep that is not in the registry (ABI 6): the launch callback + description of ststhip_run_passes, as
 * ststhip_strip_create_custom takes them -- what stencil::hip::BlockUpdate<F> (a user's transition function on a mesh of
 * GPUs) is built on.  The callback must honour the column range of the calling thread's launches
 * (ststhip_launch_columns) and the column geometry of the domain it is handed (col_origin / local_cols); a callback that
 * goes through the C++ templates' launcher (hip/internal/Sweep.hpp: launch_sweep) does. */
int ststhip_block_create_custom(ststhip_sweep_fn sweep, void *ctx, const ststhip_sweep_desc *desc, uint64_t total_rows,
                                uint64_t total_cols, int rank, int mesh_rows, int mesh_cols, ststhip_comm comm,
                                ststhip_exchange_fn exchange_rows, void *exchange_rows_ctx,
                                ststhip_exchange_fn exchange_cols, void *exchange_cols_ctx, ststhip_strip *block);
/* global rows and columns this block owns */
int ststhip_block_geometry(ststhip_strip block, uint64_t *row_begin, uint64_t *row_end, uint64_t *col_begin,
                           uint64_t *col_end);
/* Copy the owned cells of plane `plane` of the current buffer set from / to host memory whose rows are
 * `host_pitch_bytes` apart (2-D copies on the block's stream; both synchronise it before returning). */
int ststhip_block_upload(ststhip_strip block, unsigned plane, const void *host_cells, size_t host_pitch_bytes);
int ststhip_block_download(ststhip_strip block, unsigned plane, void *host_cells, size_t host_pitch_bytes);

#ifdef __cplusplus
}
#endif
#endif /* STSTHIP_H */
