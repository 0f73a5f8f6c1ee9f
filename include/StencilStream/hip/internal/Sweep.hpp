// The generation sweep of the MI355X backend: a register-resident temporal pipeline per wavefront.
//
// What it replaces: the one-work-item-per-cell SYCL kernels of the reference,
// StencilStream/cuda/StencilUpdate.hpp:227-263 (AoS) and :346-396 (SoA), launched once per
// (iteration, sub-iteration).  Semantics reproduced exactly: every
// out-of-grid neighbour reads Params::halo_value in every sub-step, stencil.id / iteration /
// subiteration / grid_range / time_dependent_value are exact per sub-step, the source grid is
// never written.
//
// How it maps to CDNA4 instead:
//  * One 64-lane wavefront owns a column strip of 64*K cells and streams down the rows of a row
//    chunk.  Lane i holds K adjacent cells of the current row of every pipeline level in VGPRs.
//  * T generations x n_subiterations sub-steps = S pipeline levels live in registers at once
//    (the FPGA design's chain of processing elements, folded into one wave).  Feeding input row y
//    makes level l emit row y - l*r; each level keeps its last 2r rows as a rotating register
//    window, so a cell travels HBM -> registers once and comes back S sub-steps later.
//  * North/south neighbours are the lane's own registers; west/east neighbours of a lane's edge
//    cells come from the adjacent lanes by DPP wave shifts (v_mov_b32_dpp wave_shr/wave_shl), no LDS.
//  * Strips overlap by the halo depth G = r*S on each side (redundant compute instead of
//    inter-wave synchronisation); chunks overlap by G rows for pipeline warm-up.
//  * HBM rows are read as one K-wide vector per lane (1 KiB per wave and row for fp32, K = 4),
//    software-prefetched P rows ahead, and written the same way.
//  * Cooperative strips (SweepTuning::cooperative, for transition functions that read no west / east
//    neighbour in the southernmost stencil row): the four waves of a workgroup take four ADJACENT strips
//    without overlap and hand each other the edge columns of every pipeline level through LDS, one
//    workgroup barrier per row.  Only the workgroup as a whole overlaps its neighbours by G columns, so a
//    wave produces (256*K - 2G)/4 instead of 64*K - 2G columns of the 64*K it loads and computes (FDTD:
//    58 instead of 40 of 64).  This is the LDS halo exchange north_star asks for, placed where the
//    redundancy was: replaces the per-work-item neighbour loads of cuda/StencilUpdate.hpp:346-396.
#pragma once
#include "../../Concepts.hpp"
#include "../../Stencil.hpp"
#include "Runtime.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <initializer_list>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>

namespace stencil {
namespace hip {
namespace internal {

using stencil::internal::round_up;
using stencil::internal::static_for;

constexpr int wave_size = 64;

// ------------------------------------------------------------------ cell layout
// A cell type opts into per-field planes with `static constexpr auto fields = std::make_tuple(
// &Cell::a, ...)` (protocol of StencilStream/cuda/internal/Helpers.hpp:37-45).
template <typename Cell>
concept SplittableCell = requires {
    std::tuple_size<std::remove_cvref_t<decltype(Cell::fields)>>::value;
};

template <typename Cell> constexpr int field_count() {
    if constexpr (SplittableCell<Cell>)
        return int(std::tuple_size_v<std::remove_cvref_t<decltype(Cell::fields)>>);
    else
        return 1;
}

template <typename Cell, int I>
using FieldType =
    std::remove_reference_t<decltype(std::declval<Cell &>().*std::get<I>(Cell::fields))>;

// Member pointer of field I as a compile-time value (never a load from the tuple object, so the
// accesses below stay static and the cells stay in registers).
template <typename Cell, int I> constexpr auto field_pointer() {
    constexpr auto pointer = std::get<I>(Cell::fields);
    return pointer;
}

template <typename Cell, int I> inline std::size_t field_offset() {
    Cell probe{};
    return std::size_t(reinterpret_cast<const char *>(&(probe.*field_pointer<Cell, I>())) -
                       reinterpret_cast<const char *>(&probe));
}

constexpr int max_planes = 16;

// Results are not read again before the next pass (a whole grid later).  Storing them with the non-temporal
// hint leaves L2 to the input rows neighbouring waves share; whether that pays is a property of the kernel
// (streaming_stores_for below).
template <bool STREAMING, typename T, int N> STST_DEVICE inline void store_cells(T *to, T const (&from)[N]) {
    if constexpr (STREAMING && (N * sizeof(T)) % 4 == 0) {
        std::uint32_t words[N * sizeof(T) / 4];
        __builtin_memcpy(words, from, sizeof words);
#pragma unroll
        for (unsigned i = 0; i < N * sizeof(T) / 4; i++)
            __builtin_nontemporal_store(words[i], reinterpret_cast<std::uint32_t *>(to) + i);
    } else {
        __builtin_memcpy(to, from, N * sizeof(T)); // typed pointers: the alignment of T decides the store width
    }
}

// Device pointers of one grid: a single AoS plane of cells, or one dense plane per field.
template <typename Cell, bool SOA> struct PlaneSet;

template <typename Cell> struct PlaneSet<Cell, false> {
    static constexpr int n_planes = 1;
    void *plane[1];

    static std::size_t elem_size(int) { return sizeof(Cell); }
    static std::size_t elem_offset(int) { return 0; }

    template <int K> STST_DEVICE void load(std::size_t first, Cell (&cells)[K]) const {
        __builtin_memcpy(cells, static_cast<const Cell *>(plane[0]) + first, K * sizeof(Cell));
    }
    STST_DEVICE void load_one(std::size_t at, Cell &cell) const {
        cell = static_cast<const Cell *>(plane[0])[at];
    }
    template <int K, bool STREAMING = false, std::uint32_t = 0>
    STST_DEVICE void store(std::size_t first, Cell const (&cells)[K]) const {
        store_cells<STREAMING>(static_cast<Cell *>(plane[0]) + first, cells);
    }
    template <std::uint32_t = 0> STST_DEVICE void store_one(std::size_t at, Cell const &cell) const {
        static_cast<Cell *>(plane[0])[at] = cell;
    }
};

template <typename Cell> struct PlaneSet<Cell, true> {
    static_assert(SplittableCell<Cell>, "split_cell_structure needs Cell::fields");
    static constexpr int n_planes = field_count<Cell>();
    static_assert(n_planes <= max_planes);
    void *plane[n_planes];

    static std::size_t elem_size(int i) {
        std::size_t sizes[n_planes];
        static_for<0, n_planes>([&](auto f) { sizes[f] = sizeof(FieldType<Cell, f>); });
        return sizes[i];
    }
    static std::size_t elem_offset(int i) {
        std::size_t offsets[n_planes];
        static_for<0, n_planes>([&](auto f) { offsets[f] = field_offset<Cell, f>(); });
        return offsets[i];
    }

    template <int K> STST_DEVICE void load(std::size_t first, Cell (&cells)[K]) const {
        static_for<0, n_planes>([&](auto f) __attribute__((always_inline)) {
            using E = FieldType<Cell, f>;
            constexpr auto member = field_pointer<Cell, f>();
            E values[K];
            __builtin_memcpy(values, static_cast<const E *>(plane[f]) + first, K * sizeof(E));
#pragma unroll
            for (int k = 0; k < K; k++)
                cells[k].*member = values[k];
        });
    }
    STST_DEVICE void load_one(std::size_t at, Cell &cell) const {
        static_for<0, n_planes>([&](auto f) __attribute__((always_inline)) {
            using E = FieldType<Cell, f>;
            constexpr auto member = field_pointer<Cell, f>();
            cell.*member = static_cast<const E *>(plane[f])[at];
        });
    }
    // SKIP_MASK: planes that already hold these values and are not stored (bit f = plane f)
    template <int K, bool STREAMING = false, std::uint32_t SKIP_MASK = 0>
    STST_DEVICE void store(std::size_t first, Cell const (&cells)[K]) const {
        static_for<0, n_planes>([&](auto f) __attribute__((always_inline)) {
            using E = FieldType<Cell, f>;
            constexpr auto member = field_pointer<Cell, f>();
            if constexpr ((SKIP_MASK >> int(f)) & 1u)
                return;
            E values[K];
#pragma unroll
            for (int k = 0; k < K; k++)
                values[k] = static_cast<E>(cells[k].*member);
            store_cells<STREAMING>(static_cast<E *>(plane[f]) + first, values);
        });
    }
    template <std::uint32_t SKIP_MASK = 0> STST_DEVICE void store_one(std::size_t at, Cell const &cell) const {
        static_for<0, n_planes>([&](auto f) __attribute__((always_inline)) {
            using E = FieldType<Cell, f>;
            constexpr auto member = field_pointer<Cell, f>();
            if constexpr ((SKIP_MASK >> int(f)) & 1u)
                return;
            static_cast<E *>(plane[f])[at] = static_cast<E>(cell.*member);
        });
    }
};

// ------------------------------------------------------------------ lane exchange
// Value held by the neighbouring lane, moved with DPP wave shifts (full 64-lane shift on gfx9-family
// ISAs).  Works on the 32-bit words of any trivially copyable T.
template <int DppCtrl, typename T> STST_DEVICE inline T lane_shift(T const &value) {
    static_assert(std::is_trivially_copyable_v<T>);
    constexpr int n_words = int((sizeof(T) + 3) / 4);
    struct Words {
        int w[n_words];
    } words = {};
    __builtin_memcpy(&words, &value, sizeof(T));
#pragma unroll
    for (int i = 0; i < n_words; i++)
        words.w[i] = __builtin_amdgcn_update_dpp(0, words.w[i], DppCtrl, 0xf, 0xf, true);
    T shifted;
    __builtin_memcpy(&shifted, &words, sizeof(T));
    return shifted;
}
// The same shift, but the lane without a source lane (lane 0 for wave_shr, lane 63 for wave_shl) receives
// `edge` instead (DPP with bound_ctrl = 0 leaves the destination, preloaded with `edge`, untouched there).
template <int DppCtrl, typename T> STST_DEVICE inline T lane_shift_or(T const &value, T const &edge) {
    static_assert(std::is_trivially_copyable_v<T>);
    constexpr int n_words = int((sizeof(T) + 3) / 4);
    struct Words {
        int w[n_words];
    } words = {}, old = {};
    __builtin_memcpy(&words, &value, sizeof(T));
    __builtin_memcpy(&old, &edge, sizeof(T));
#pragma unroll
    for (int i = 0; i < n_words; i++)
        words.w[i] = __builtin_amdgcn_update_dpp(old.w[i], words.w[i], DppCtrl, 0xf, 0xf, false);
    T shifted;
    __builtin_memcpy(&shifted, &words, sizeof(T));
    return shifted;
}
template <typename T> STST_DEVICE inline T from_west_lane_or(T const &v, T const &edge) {
    return lane_shift_or<0x138>(v, edge);
}
template <typename T> STST_DEVICE inline T from_east_lane_or(T const &v, T const &edge) {
    return lane_shift_or<0x130>(v, edge);
}
// value of lane-1 (data moves towards higher lanes): DPP wave_shr:1
template <typename T> STST_DEVICE inline T from_west_lane(T const &v) { return lane_shift<0x138>(v); }
// value of lane+1: DPP wave_shl:1
template <typename T> STST_DEVICE inline T from_east_lane(T const &v) { return lane_shift<0x130>(v); }

// ------------------------------------------------------------------ tuning
constexpr int ceil_pow2(int v) {
    int p = 1;
    while (p < v)
        p *= 2;
    return p;
}

template <typename Cell, bool SOA> constexpr int cell_words() {
    if constexpr (SOA) {
        int words = 0;
        static_for<0, field_count<Cell>()>(
            [&](auto f) { words += int((sizeof(FieldType<Cell, f>) + 3) / 4); });
        return words;
    } else {
        return int((sizeof(Cell) + 3) / 4);
    }
}

} // namespace internal

// Shape of the wave pipeline for a transition function.  Specialise for a concrete F to override:
//   cells_per_lane (K)   adjacent cells per lane and row (vector width of the HBM accesses)
//   max_generations (T)  deepest temporal blocking compiled (it and its repeated halvings are built)
//   prefetch_rows (P)    rows loaded ahead; must be a multiple of 2*radius
//   interior_variant     also build the check-free code path for waves away from the grid edge
//   min_waves_per_simd   occupancy the register allocator must allow (second __launch_bounds__ argument)
//   streaming_stores     (optional member) store results with the non-temporal hint (streaming_stores_for below)
//   cooperative          (optional member, default false) the four waves of a workgroup take adjacent strips and
//                        exchange their edge columns through LDS (see the head of Sweep.hpp).  ONLY for transition
//                        functions that never read stencil[radius][dc] with dc != 0 ... more precisely: no cell of
//                        the southernmost stencil row other than the columns of the lane's own cells, i.e. for
//                        radius 1 neither stencil[1][-1] nor stencil[1][1] (5-point and "north-heavy" stencils).
//                        The southernmost row is the one a level receives in the same step, before the
//                        neighbour wave's copy can have crossed the barrier.
//   trapezoid_fill       (optional member) skip the levels that are not due yet while a wave's pipeline fills;
//                        default: cells of up to four words per generation (measured: Jacobi +3 %, HotSpot
//                        +5 %, Conway +4 %, FDTD -2..-8 %: profiles/r01_ab_trapezoid_fill.txt)
template <typename F, bool SOA> struct SweepTuning {
  private:
    static constexpr int R = int(F::stencil_radius);
    static constexpr int NS = int(F::n_subiterations);
    static constexpr int W = internal::cell_words<typename F::Cell, SOA>();
    // Measured on MI355X (profiles/r01_tune_shapes_apps.txt): the kernel is bound by VALU issue and
    // occupancy, not by HBM, so the register window T*NS*2R*K*W decides.  One-word cells want K = 4
    // (fewest DPP moves and halo columns per cell); fatter cells want the narrowest lane that still
    // covers the radius: K = 2 for two-word AoS cells, K = 1 otherwise.
    static constexpr int pick_k() {
        int k = (W == 1) ? 4 : ((W == 2 && !SOA) ? 2 : 1);
        return std::max(k, internal::ceil_pow2(R));
    }
    // Deepest power of two up to 8 generations whose window stays within 128 words per lane -- and 6 for
    // functions with sub-iterations: they touch only part of the cell (and of the neighbourhood) per sub-step,
    // so part of the nominal window is dead after inlining.  Measured on FDTD (8 words, 2 sub-iterations): the
    // best depth is 6, a nominal 192 words (profiles/r01_tune_shapes_apps.txt: T = 4: 270, 6: 347, 7: 234
    // Gcell/s; the unchanged example: 1.96 -> 1.46 s, profiles/r01_ab_examples.txt).  Counting two thirds of
    // the nominal window at depth 6 reproduces that; it is only trusted for up to 8 words per lane, and a
    // window that fits as it is keeps the powers of two (FDTD with the LUT resolver, 5 words: T = 4 beats 6 and 8).
    static constexpr int nominal_window(int t, int k) { return t * NS * 2 * R * k * W; }
    static constexpr bool geometry_ok(int t, int k) { return 64 * k > 2 * internal::round_up(R * t * NS, k); }
    static constexpr bool relaxed(int t, int k) {
        return t == 6 && NS >= 2 && k * W <= 8 && nominal_window(t, k) > 128 && nominal_window(t, k) * 2 / 3 <= 128 &&
               geometry_ok(t, k);
    }
    // Radius 2 and more: a level reads (2R+1)^2 cells and keeps 2R rows, so the arithmetic per cell grows with R^2
    // while a deeper launch saves the same bytes -- the measured optimum is shallow (dense 5 x 5 Jacobi, 16384^2:
    // T = 8: 540, 4: 650, 2: 760 Gcell/s, profiles/r02_tune_radius.txt): the window is capped at 32 words there.
    static constexpr int window_limit = R >= 2 ? 32 : 128;
    static constexpr int pick_t(int k) {
        for (int t : {8, 6, 4, 2})
            if (t == 6 ? relaxed(t, k) : (nominal_window(t, k) <= window_limit && geometry_ok(t, k)))
                return t;
        return 1;
    }
    // rows in flight: at least four, except where the window was admitted on the relaxed count
    static constexpr int pick_p(int t, int k) {
        int p = 2 * R;
        while (p < 4 && !relaxed(t, k))
            p += 2 * R;
        return p;
    }

  public:
    static constexpr int cells_per_lane = pick_k();
    static constexpr int max_generations = pick_t(cells_per_lane);
    static constexpr int prefetch_rows = pick_p(max_generations, cells_per_lane);
    static constexpr bool interior_variant = (W * NS <= 16);
    static constexpr int min_waves_per_simd = 1;
};

namespace internal {
// The same transition function swept with the narrowest lanes (one cell per lane for radius 1): four times as many
// waves for a thin-cell function whose default shape holds four cells per lane.  The launcher switches to it for
// grids too small to fill the chip with the default shape (narrow_form_cells below): a launch is then the latency
// of one wave through its warm-up rows, and narrower lanes mean more waves sharing it.  Measured, Jacobi5General
// (profiles/r02_small_grids.txt): 256^2 26.7 -> 46.7, 512^2 92 -> 158, 1024^2 310 -> 462, 2048^2 854 -> 1025,
// 4096^2 1830 -> 1865 Gcell-updates/s; the default shape wins from about 4600^2 on.
template <typename F> struct NarrowForm : public F {
    NarrowForm(F const &f) : F(f) {}
};
} // namespace internal

template <typename F, bool SOA> struct SweepTuning<internal::NarrowForm<F>, SOA> {
    static constexpr int cells_per_lane = internal::ceil_pow2(int(F::stencil_radius));
    static constexpr int max_generations = SweepTuning<F, SOA>::max_generations;
    static constexpr int prefetch_rows = SweepTuning<F, SOA>::prefetch_rows;
    static constexpr bool interior_variant = SweepTuning<F, SOA>::interior_variant;
    static constexpr int min_waves_per_simd = SweepTuning<F, SOA>::min_waves_per_simd;
    static constexpr bool cooperative = false;
    static constexpr bool trapezoid_fill = [] {
        if constexpr (requires { SweepTuning<F, SOA>::trapezoid_fill; })
            return SweepTuning<F, SOA>::trapezoid_fill;
        else
            return internal::cell_words<typename F::Cell, SOA>() * int(F::n_subiterations) <= 4;
    }();
    static constexpr bool streaming_stores = [] {
        if constexpr (requires { SweepTuning<F, SOA>::streaming_stores; })
            return SweepTuning<F, SOA>::streaming_stores;
        else
            return sizeof(typename F::Cell) >= 4 &&
                   internal::cell_words<typename F::Cell, SOA>() * int(F::n_subiterations) <= 2;
    }();
};

namespace internal {

template <typename T> struct is_narrow_form : std::false_type {};
template <typename F> struct is_narrow_form<NarrowForm<F>> : std::true_type {};

// Does F have a narrower shape worth compiling?  Only functions whose default shape holds several cells per lane
// and that are not swept cooperatively; the wave of the narrow shape must still produce columns at full depth.
template <typename F, bool SOA> constexpr bool has_narrow_form() {
    if constexpr (is_narrow_form<F>::value) {
        return false;
    } else {
        constexpr int k = ceil_pow2(int(F::stencil_radius));
        constexpr int g = int(F::stencil_radius) * int(F::n_subiterations) * SweepTuning<F, SOA>::max_generations;
        bool coop = false;
        if constexpr (requires { SweepTuning<F, SOA>::cooperative; })
            coop = SweepTuning<F, SOA>::cooperative;
        return !coop && SweepTuning<F, SOA>::cells_per_lane > k && wave_size * k - 2 * round_up(g, k) >= 16 * k;
    }
}

// Geometry of one sweep launch, in global grid coordinates.
struct SweepGeometry {
    std::int32_t grid_h, grid_w;        // stencil.grid_range
    std::int32_t row_origin;            // global row of buffer row 0
    std::int32_t load_lo, load_hi;      // global rows present in the source buffers
    std::int32_t out_begin, out_end;    // global rows to produce
    std::int32_t chunk_rows;            // rows of output per wave
    std::uint32_t n_strips, n_chunks;   // wave grid
    // The last waves of the grid (dispatched last) take shorter chunks, so that the ragged end of a
    // launch -- SIMDs left with one or two waves -- is short.  Tier t covers chunks
    // [tier_first[t], tier_first[t+1]) with tier_rows[t] rows each; tier 0 starts at out_begin.
    static constexpr int max_tiers = 4;
    std::uint32_t n_tiers;
    std::uint32_t tier_first[max_tiers + 1];
    std::int32_t tier_rows[max_tiers];
    std::int32_t tier_begin[max_tiers]; // first output row of the tier
    std::uint64_t pitch;                // elements between rows
    std::uint32_t pitch32;              // the same (below 2^31, checked at launch)
    std::uint64_t iteration;            // generation index of the first level
    std::uint32_t xcd_remap;            // 1: give every XCD a contiguous range of the wave grid
    std::uint32_t last_chunk_early;     // 1: the last row chunk is dispatched second instead of last
    // Persistent waves (sweep_kernel<..., PERSISTENT>): the rows are cut into fine chunks of `fine_rows`; a wave
    // claims a chunk, and when it has finished it claims the chunk BELOW and keeps streaming -- its pipeline holds
    // exactly the state that chunk needs, so only the first chunk of a run pays the 2G warm-up rows -- until it
    // meets a chunk somebody else has claimed; then it takes the next unclaimed chunk from a shared cursor.
    std::uint32_t fine_rows, n_fine;    // rows per fine chunk, chunks per strip
    std::uint32_t starts_per_strip;     // runs that start side by side in one strip (cursor phase 0)
    std::uint32_t start_stride;         // distance of those starts in chunks
    std::uint32_t phase_step;           // chunks between the starts of consecutive ticket phases
    std::uint32_t cursor_end;           // tickets: phases * starts_per_strip * n_strips
    std::uint32_t *claims;              // one word per (chunk, strip), zeroed before the launch
    std::uint32_t *cursor;              // one word, zeroed before the launch
};

template <typename F, bool SOA> constexpr bool cooperative_for() {
    if constexpr (requires { SweepTuning<F, SOA>::cooperative; })
        return SweepTuning<F, SOA>::cooperative;
    else
        return false;
}

// SweepTuning<F, SOA>::persistent (optional member, default false): launch as many waves as stay resident and let
// each claim fine row chunks, continuing into the chunk below without re-warming its pipeline (SweepGeometry).
template <typename F, bool SOA> constexpr bool persistent_for() {
    if constexpr (requires { SweepTuning<F, SOA>::persistent; })
        return SweepTuning<F, SOA>::persistent;
    else
        return false;
}

template <typename F, bool SOA> constexpr int cooperative_debug_for() {
    if constexpr (requires { SweepTuning<F, SOA>::cooperative_debug; })
        return SweepTuning<F, SOA>::cooperative_debug;
    else
        return 0;
}

template <typename F, bool SOA> constexpr bool trapezoid_fill_for() {
    if constexpr (requires { SweepTuning<F, SOA>::trapezoid_fill; })
        return SweepTuning<F, SOA>::trapezoid_fill;
    else
        return cell_words<typename F::Cell, SOA>() * int(F::n_subiterations) <= 4;
}

// A transition function may declare fields it only copies from the centre cell,
//     static constexpr auto constant_fields = std::make_tuple(&Cell::power);
// (an extension: the reference's API has no such hint, cuda/StencilUpdate.hpp:387-392 stores every field in
// every sweep).  With per-field planes the pass driver's targets alternate between two buffers, so from the
// third pass of a run on the target planes of those fields already hold their values and the stores are left
// out: a quarter of the HBM bytes of HotSpot and FDTD, which run at the practical HBM rate.
template <typename F> constexpr std::uint32_t constant_plane_mask() {
    std::uint32_t mask = 0;
    if constexpr (requires { F::constant_fields; } && SplittableCell<typename F::Cell>) {
        using Cell = typename F::Cell;
        constexpr int n_constant = int(std::tuple_size_v<std::remove_cvref_t<decltype(F::constant_fields)>>);
        static_for<0, field_count<Cell>()>([&](auto f) {
            static_for<0, n_constant>([&](auto c) {
                using A = std::remove_cvref_t<decltype(std::get<f>(Cell::fields))>;
                using B = std::remove_cvref_t<decltype(std::get<c>(F::constant_fields))>;
                if constexpr (std::is_same_v<A, B>)
                    if (std::get<f>(Cell::fields) == std::get<c>(F::constant_fields))
                        mask |= 1u << int(f);
            });
        });
    }
    return mask;
}

// Non-temporal stores of the results (SweepTuning<F, SOA>::streaming_stores, optional member).  Measured
// (profiles/r01_ab_nt_stores.txt): +1.2..1.6 % Jacobi, +1..5 % HotSpot fp32, +2..4 % HotSpot fp64 on planes; but
// -8 % FDTD on planes, -1 % FDTD / HotSpot fp64 as AoS and -3 % for the packed Game of Life -- the kernels that are
// bound by HBM or store narrow rows pay for partial lines that L2 no longer merges.  Default: cells of up to two
// 32-bit words per generation that are at least one word wide.
template <typename F, bool SOA> constexpr bool streaming_stores_for() {
    if constexpr (requires { SweepTuning<F, SOA>::streaming_stores; })
        return SweepTuning<F, SOA>::streaming_stores;
    else
        return sizeof(typename F::Cell) >= 4 &&
               cell_words<typename F::Cell, SOA>() * int(F::n_subiterations) <= 2;
}

constexpr int waves_per_block = 4; // 4 waves per workgroup: 2 or 1 are 1-3 % slower (profiles/r01_tune_taper.txt)

// COOP_DEBUG (timing experiments only, results are wrong): 1 = no barrier, 2 = no LDS traffic, 3 = neither
template <typename F, bool SOA, int T, int K, int P, bool INTERIOR_VARIANT, bool COOP = false, int COOP_DEBUG = 0,
          bool INLINE_TDV = false>
struct Sweep {
    using Cell = typename F::Cell;
    using TDV = typename F::TimeDependentValue;
    using Planes = PlaneSet<Cell, SOA>;
    using StencilImpl = Stencil<Cell, F::stencil_radius, TDV>;

    static constexpr int R = int(F::stencil_radius);
    static constexpr int NS = int(F::n_subiterations);
    static constexpr int S = T * NS;            // pipeline levels
    static constexpr int G = R * S;             // halo depth in cells (rows and columns)
    static constexpr int GX = round_up(G, K);   // column halo rounded to whole lanes
    static constexpr int LW = wave_size * K;    // columns a wave loads
    // columns one unit of the wave grid loads / produces: a wave, or (COOP) the waves of a workgroup
    static constexpr int UW = COOP ? waves_per_block * LW : LW;
    static constexpr int OW = UW - 2 * GX;
    static constexpr int OW_PER_WAVE = COOP ? OW / waves_per_block : OW; // what the launch heuristics count waves with
    static constexpr int NWIN = 2 * R;          // rows each level keeps
    static constexpr int D = 2 * R + 1;
    static constexpr int prefetch_depth = P;

    // COOP: edge columns in LDS.  Slot (step mod N_SLOTS) holds, per level and wave, the R westernmost cells of
    // lane 0 and the R easternmost cells of lane 63 of the row that entered that level's window in that step.
    static constexpr int CW = int((sizeof(Cell) + 3) / 4);          // 32-bit words per cell
    static constexpr int N_SLOTS = 2 * R + 1;                       // rows alive in a window + the one being written
    static constexpr int EDGE_WORDS = R * CW;                       // one edge of one wave at one level
    static constexpr int LEVEL_WORDS = waves_per_block * 2 * EDGE_WORDS;
    static constexpr int SLOT_WORDS = S * LEVEL_WORDS;
    static constexpr int LDS_WORDS = COOP ? N_SLOTS * SLOT_WORDS : 1;

    static_assert(K >= R, "a lane must hold at least `radius` cells so neighbours are one lane away");
    static_assert(OW >= K, "halo consumes the whole strip: lower max_generations or raise K");
    static_assert(!COOP || LDS_WORDS * 4 <= 64 * 1024, "edge exchange buffers exceed the LDS budget of a workgroup");
    static_assert(P % NWIN == 0, "prefetch depth must be a multiple of the window length");
    static_assert(std::is_trivially_copyable_v<F> && std::is_trivially_copyable_v<Cell> &&
                  std::is_trivially_copyable_v<TDV>);

    struct Args {
        F f;
        Cell halo;
        // Time-dependent values of the launch's T generations, one of three sources (tdv/SinglePassStrategies.hpp):
        // `tdv_table` != nullptr: device array, element i = generation i of this launch (the pass driver's per-call
        // table: precomputed on the host or on the device); else `tdv`: evaluated on the host for this launch
        // and shipped as kernel arguments; INLINE_TDV: evaluated by the kernel itself, neither is read.
        TDV tdv[T];
        TDV const *tdv_table;
        Planes src, dst;
        SweepGeometry geo;
    };

    // SKIP_CONSTANTS: the target planes of F::constant_fields already hold their values (see
    // constant_plane_mask); their stores are left out
    // `strip`: index of the unit (wave, or workgroup if COOP) along the columns; `wib`: wave in the workgroup
    // `more_rows(yb)`: persistent waves only -- called when the rows up to yb are done, returns the new end of the
    // wave's rows if it could claim the chunk below (then the wave keeps streaming), or yb.
    // Returns the row the wave's output ends at.
    template <bool EDGE, bool SKIP_CONSTANTS, typename MoreRows>
    STST_DEVICE static int run(Args const &a, const int lane, const int strip, const int ya,
                               const int yb_first, const int wib, std::uint32_t *lds, MoreRows more_rows) {
        constexpr bool CONTINUES = !std::is_same_v<MoreRows, std::nullptr_t>;
        int yb = yb_first;
        constexpr std::uint32_t skip_mask = SKIP_CONSTANTS ? constant_plane_mask<F>() : 0u;
        SweepGeometry const &g = a.geo;
        const int unit_x = COOP ? wib * LW + lane * K : lane * K; // column of the lane inside its unit
        const int x0 = strip * OW - GX + unit_x; // global column of the lane's first cell
        const int ystart = ya - G;
        // a wave that may continue below its chunk prefetches real rows there: up to what the launch may read
        const int y_load_end = CONTINUES ? (g.out_end + G < g.load_hi ? g.out_end + G : g.load_hi)
                                         : (yb + G < g.load_hi ? yb + G : g.load_hi);

        bool col_in[K];
#pragma unroll
        for (int k = 0; k < K; k++)
            col_in[k] = unsigned(x0 + k) < unsigned(g.grid_w);
        const bool vec_in = x0 >= 0 && x0 + K <= g.grid_w;
        const bool lane_stores = unit_x >= GX && unit_x + K <= UW - GX;

        // COOP: LDS word offsets.  A lane reads the neighbour waves' edges (every lane the same address; only the
        // value of lane 0 / 63 is used) and the two edge lanes write their own.
        int lds_west = 0, lds_east = 0, lds_mine = 0; // words inside a (slot, level) block
        if constexpr (COOP) {
            const int w_west = wib > 0 ? wib - 1 : 0, w_east = wib + 1 < waves_per_block ? wib + 1 : wib;
            lds_west = (w_west * 2 + 1) * EDGE_WORDS; // east edge of the wave to the west
            lds_east = (w_east * 2 + 0) * EDGE_WORDS; // west edge of the wave to the east
            lds_mine = (wib * 2 + (lane == 0 ? 0 : 1)) * EDGE_WORDS;
        }
        int slot_now = 0; // slot written in this step; the row of m steps ago is in slot (slot_now - m) mod N_SLOTS
        auto lds_cell = [&](int slot, int level_index, int edge_offset, int e) __attribute__((always_inline)) {
            Cell cell;
            std::uint32_t words[CW] = {};
            const std::uint32_t *from = lds + slot * SLOT_WORDS + level_index * LEVEL_WORDS + edge_offset + e * CW;
#pragma unroll
            for (int i = 0; i < CW; i++)
                words[i] = from[i];
            __builtin_memcpy(&cell, words, sizeof(Cell));
            return cell;
        };

        // The launch's time-dependent values: the call's device table, or the kernel arguments.  Both are read
        // through the constant address space -- nothing writes them while the kernel runs --, so the compiler may
        // load a value again wherever it needs it instead of holding T scalar registers over the row loop (a plain
        // global load could not be moved over the loop's stores at all).
        using ConstantTDV = const TDV __attribute__((address_space(4)));
        ConstantTDV *launch_tdv = a.tdv_table ? (ConstantTDV *)(a.tdv_table) : (ConstantTDV *)(a.tdv);

        Cell win[S][NWIN][K]; // level l-1's older rows, rotating
        Cell pre[P][K];       // rows in flight from HBM
        static_for<0, S>([&](auto l) __attribute__((always_inline)) {
            static_for<0, NWIN>([&](auto s) __attribute__((always_inline)) {
#pragma unroll
                for (int k = 0; k < K; k++)
                    win[l][s][k] = a.halo;
            });
        });
        static_for<0, P>([&](auto u) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < K; k++)
                pre[u][k] = a.halo;
        });

        // Rows outside [load_lo, y_load_end) are clamped into that range instead of being skipped: the
        // load stays unconditional, so the compiler can count the loads in flight (s_waitcnt vmcnt(P-1)
        // instead of vmcnt(0)).  A clamped row only feeds cells that lie outside the grid (replaced by
        // the halo value below) or below the last row this wave has to produce.
        auto load_row = [&](const int y, Cell(&into)[K]) __attribute__((always_inline)) {
            int yc = y < g.load_lo ? g.load_lo : y;
            yc = yc < y_load_end ? yc : y_load_end - 1;
            const std::size_t first =
                row_offset(g, yc) + std::size_t(std::int64_t(x0));
            if constexpr (!EDGE) {
                a.src.template load<K>(first, into);
            } else {
                if (vec_in) {
                    a.src.template load<K>(first, into);
                } else {
#pragma unroll
                    for (int k = 0; k < K; k++)
                        if (col_in[k])
                            a.src.load_one(first + k, into[k]);
                }
            }
        };

        static_for<0, P>([&](auto u) __attribute__((always_inline)) { load_row(ystart + u, pre[u]); });

        // One input row through the pipeline.  While the pipeline fills (FILLING: the first 2G rows of the
        // wave) level l only has to produce rows from input row 2*l*R of the wave on -- earlier outputs
        // cannot reach a stored row -- so the deeper levels are skipped by wave-uniform branches: a
        // trapezoid of level-steps instead of a parallelogram, G*(S+1) fewer of them per wave.
        int n_rows_in = yb - ya + 2 * G;
        auto row_step = [&](auto u, const int it, auto filling) __attribute__((always_inline)) {
                constexpr bool FILLING = decltype(filling)::value;
                const int step = it + u; // index of the input row inside the wave
                const int y = ystart + it + u;
                Cell cur[K];
#pragma unroll
                for (int k = 0; k < K; k++)
                    cur[k] = pre[u][k];
                load_row(y + P, pre[u]);

                if constexpr (EDGE) {
                    const bool row_in = unsigned(y) < unsigned(g.grid_h);
#pragma unroll
                    for (int k = 0; k < K; k++)
                        if (!(row_in && col_in[k]))
                            cur[k] = a.halo;
                }

                // COOP: the neighbour waves' edge cells of the window rows of every level, fetched before this
                // step writes anything to LDS (so the reads are not ordered behind those writes and their
                // latency is paid once per row, not once per level); what the transition function does not
                // use is never loaded.  Window row rr entered D-1-rr steps ago and has crossed a barrier since.
                Cell edge_west[COOP ? S : 1][D - 1][R], edge_east[COOP ? S : 1][D - 1][R];
                if constexpr (COOP && !(COOP_DEBUG & 2)) {
                    static_for<0, S>([&](auto lc) __attribute__((always_inline)) {
                        static_for<0, D - 1>([&](auto rr) __attribute__((always_inline)) {
                            constexpr int age = D - 1 - int(rr);
                            int slot = slot_now - age;
                            slot = slot < 0 ? slot + N_SLOTS : slot;
#pragma unroll
                            for (int e = 0; e < R; e++) {
                                edge_west[lc][rr][e] = lds_cell(slot, lc, lds_west, e);
                                edge_east[lc][rr][e] = lds_cell(slot, lc, lds_east, e);
                            }
                        });
                    });
                }

                bool live = true; // FILLING: the levels up to here are due at this row
                static_for<0, S>([&](auto lc) __attribute__((always_inline)) {
                    constexpr int level = lc + 1;           // level being computed
                    constexpr int oldest = u % NWIN;        // window slot holding the oldest row
                    if constexpr (FILLING) {
                        if (!live)
                            return;
                        if (step < 2 * level * R) {
                            // not due yet, but its window takes the row the level before it just emitted
#pragma unroll
                            for (int k = 0; k < K; k++)
                                win[lc][oldest][k] = cur[k];
                            live = false;
                            return;
                        }
                    }
                    const int j = y - level * R;            // global row this level emits now
                    const std::size_t iteration = g.iteration + std::size_t((level - 1) / NS);
                    const std::size_t subiteration = std::size_t((level - 1) % NS);
                    const TDV tdv = [&]() -> TDV {
                        if constexpr (std::is_empty_v<TDV>)
                            return TDV{};
                        else if constexpr (INLINE_TDV)
                            return a.f.get_time_dependent_value(iteration);
                        else
                            return launch_tdv[(level - 1) / NS];
                    }();

                    // rows j-R .. j+R of the previous level, widened by R cells from both neighbour lanes
                    Cell ext[D][K + 2 * R];
                    static_for<0, D>([&](auto rr) __attribute__((always_inline)) {
                        Cell const(&row)[K] = [&]() -> Cell const(&)[K] {
                            if constexpr (rr < NWIN)
                                return win[lc][(oldest + rr) % NWIN];
                            else
                                return cur;
                        }();
#pragma unroll
                        for (int k = 0; k < K; k++)
                            ext[rr][R + k] = row[k];
                        if constexpr (COOP && !(COOP_DEBUG & 2) && rr < D - 1) {
#pragma unroll
                            for (int d = 1; d <= R; d++) {
                                ext[rr][R - d] = from_west_lane_or(row[K - d], edge_west[lc][rr][R - d]);
                                ext[rr][R + K - 1 + d] = from_east_lane_or(row[d - 1], edge_east[lc][rr][d - 1]);
                            }
                        } else {
#pragma unroll
                            for (int d = 1; d <= R; d++) {
                                ext[rr][R - d] = from_west_lane(row[K - d]);
                                ext[rr][R + K - 1 + d] = from_east_lane(row[d - 1]);
                            }
                        }
                    });

                    Cell next[K];
                    bool row_in = true;
                    if constexpr (EDGE)
                        row_in = unsigned(j) < unsigned(g.grid_h);
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        // stencil.id and grid_range are size_t in the API; their values fit 31 bits (checked at
                        // launch).  Inside the grid -- always, in waves whose footprint lies inside it -- the
                        // coordinates are not negative either: saying so lets a transition function's index
                        // arithmetic (comparisons, conversions to float) stay 32 bits wide
                        if constexpr (!EDGE) {
                            __builtin_assume(j >= 0);
                            __builtin_assume(x0 + k >= 0);
                        }
                        __builtin_assume(g.grid_h >= 0);
                        __builtin_assume(g.grid_w >= 0);
                        StencilImpl st(sycl::id<2>(std::size_t(std::int64_t(j)),
                                                   std::size_t(std::int64_t(x0 + k))),
                                       sycl::range<2>(std::size_t(g.grid_h), std::size_t(g.grid_w)),
                                       iteration, subiteration, tdv);
#pragma unroll
                        for (int rr = 0; rr < D; rr++)
#pragma unroll
                            for (int cc = 0; cc < D; cc++)
                                st[sycl::id<2>(rr, cc)] = ext[rr][k + cc];
                        // a transition function may provide a form that knows its level inside the
                        // launch at compile time (used by fused forms whose first / last level differ)
                        if constexpr (requires { a.f.template at_level<0, 1>(st); })
                            next[k] = a.f.template at_level<decltype(lc)::value, S>(st);
                        // ... or a form for cells that are not on the rim of the grid: in a wave whose
                        // whole footprint lies inside the grid every cell that can reach the output has
                        // 0 < row < height-1 and 0 < column < width-1 at every level
                        else if constexpr (!EDGE && requires { a.f.interior(st); })
                            next[k] = a.f.interior(st);
                        else
                            next[k] = a.f(st);
                        if constexpr (EDGE)
                            if (!(row_in && col_in[k]))
                                next[k] = a.halo; // out-of-grid cells never evolve
                    }

                    // the newest row of the previous level replaces the oldest one in its window
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        win[lc][oldest][k] = cur[k];
                        cur[k] = next[k];
                    }
                });

                const int j = y - G; // row leaving the last level
                if ((!FILLING || live) && j >= ya && j < yb && lane_stores) {
                    const std::size_t first =
                        row_offset(g, j) + std::size_t(std::int64_t(x0));
                    if (!EDGE || vec_in) {
                        a.dst.template store<K, streaming_stores_for<F, SOA>(), skip_mask>(first, cur);
                    } else {
#pragma unroll
                        for (int k = 0; k < K; k++)
                            if (col_in[k])
                                a.dst.template store_one<skip_mask>(first + k, cur[k]);
                    }
                }
                if constexpr (COOP) {
                    // The rows that entered the windows in this step now sit in the windows' `oldest` slots (they
                    // replaced the oldest rows): their edge cells go to the neighbour waves.  One masked block per
                    // step -- a branch per level would cut the step into basic blocks too small for the scheduler
                    // to interleave the levels.
                    if ((lane == 0 || lane == wave_size - 1) && !(COOP_DEBUG & 2)) {
                        static_for<0, S>([&](auto lc) __attribute__((always_inline)) {
                            constexpr int entered = u % NWIN;
                            std::uint32_t *to = lds + slot_now * SLOT_WORDS + lc * LEVEL_WORDS + lds_mine;
#pragma unroll
                            for (int e = 0; e < R; e++) {
                                Cell const &cell = lane == 0 ? win[lc][entered][e] : win[lc][entered][K - R + e];
                                std::uint32_t words[CW] = {};
                                __builtin_memcpy(words, &cell, sizeof(Cell));
#pragma unroll
                                for (int i = 0; i < CW; i++)
                                    to[e * CW + i] = words[i];
                            }
                        });
                    }
                    // edges of this step become visible to the other waves; LDS only -- the rows in flight from
                    // HBM (pre[]) must not be waited for here
                    if constexpr (!(COOP_DEBUG & 1)) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
                    }
                    slot_now = slot_now + 1 == N_SLOTS ? 0 : slot_now + 1;
                }
        };

        // Waves at the grid edge are few: they keep one loop (every level at every row) and small code.
        int it = 0;
        if constexpr (!EDGE && trapezoid_fill_for<F, SOA>()) {
            for (; it < 2 * G && it < n_rows_in; it += P)
                static_for<0, P>([&](auto u) __attribute__((always_inline)) { row_step(u, it, std::true_type{}); });
        }
        for (; it < n_rows_in; it += P)
            static_for<0, P>([&](auto u) __attribute__((always_inline)) { row_step(u, it, std::false_type{}); });
        if constexpr (CONTINUES) {
            // exactly n_rows_in rows have been fed (the launcher keeps chunk lengths and 2G multiples of P): the
            // pipeline is in the state the chunk below starts from
            while (it == n_rows_in) {
                const int further = more_rows(yb);
                if (further == yb)
                    break;
                yb = further;
                n_rows_in = yb - ya + 2 * G;
                for (; it < n_rows_in; it += P)
                    static_for<0, P>([&](auto u) __attribute__((always_inline)) { row_step(u, it, std::false_type{}); });
            }
        }
        return yb;
    }

    template <bool SKIP_CONSTANTS = false> STST_DEVICE static void entry(Args const &a, std::uint32_t *lds) {
        SweepGeometry const &g = a.geo;
        const int lane = int(threadIdx.x) & (wave_size - 1);
        const int wib = int(__builtin_amdgcn_readfirstlane(threadIdx.x / wave_size));
        // Workgroups are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8, each with its own
        // L2).  Renumber them so that the blocks of one XCD cover a contiguous range of (strip, chunk)
        // tiles: neighbouring tiles share halo columns/rows, which then hit in that XCD's L2.  Only the
        // speed depends on the placement assumption, never the result.
        unsigned block = blockIdx.x;
        if (g.xcd_remap) {
            constexpr unsigned n_xcd = 8;
            const unsigned q = gridDim.x / n_xcd, r = gridDim.x % n_xcd, x = block % n_xcd;
            block = x * q + (x < r ? x : r) + block / n_xcd;
        }
        // unit of the wave grid: a wave, or (COOP) the whole workgroup -- its waves then share strip and chunk,
        // run the same number of rows and meet at one barrier per row
        const unsigned unit = COOP ? block : __builtin_amdgcn_readfirstlane(block * unsigned(waves_per_block) + unsigned(wib));
        if (unit >= g.n_strips * g.n_chunks)
            return;
        const int strip = int(unit % g.n_strips);
        // The bottom chunk of a launch that reaches the grid's last rows runs the slower edge code; in
        // dispatch order it would come last and stretch the end of the launch, so it is moved to the front.
        int chunk = int(unit / g.n_strips);
        if (g.last_chunk_early && g.n_chunks >= 3)
            chunk = chunk == 0 ? 0 : (chunk == 1 ? int(g.n_chunks) - 1 : chunk - 1);
        int tier = 0;
#pragma unroll
        for (int t = 1; t < SweepGeometry::max_tiers; t++)
            if (t < int(g.n_tiers) && unsigned(chunk) >= g.tier_first[t])
                tier = t;
        const int ya = g.tier_begin[tier] + (chunk - int(g.tier_first[tier])) * g.tier_rows[tier];
        int yb = ya + g.tier_rows[tier];
        yb = yb < g.out_end ? yb : g.out_end;

        if constexpr (INTERIOR_VARIANT) {
            const int xw0 = strip * OW - GX; // footprint of the unit: decided per workgroup when COOP
            const bool interior =
                xw0 >= 0 && xw0 + UW <= g.grid_w && ya - G >= 0 && yb + G <= g.grid_h;
            if (interior)
                run<false, SKIP_CONSTANTS>(a, lane, strip, ya, yb, wib, lds, nullptr);
            else
                run<true, SKIP_CONSTANTS>(a, lane, strip, ya, yb, wib, lds, nullptr);
        } else {
            run<true, SKIP_CONSTANTS>(a, lane, strip, ya, yb, wib, lds, nullptr);
        }
    }

    // Element offset of buffer row `y` (a global row the buffers hold): both factors fit 31 bits (checked at launch),
    // so the 64-bit product is a 32 x 32 multiply -- two scalar instructions instead of the five of a 64 x 64 one,
    // once per loaded and per stored row.
    STST_DEVICE static std::size_t row_offset(SweepGeometry const &g, int y) {
        return std::size_t(std::uint32_t(y - g.row_origin)) * std::size_t(g.pitch32);
    }

    // ---- persistent waves (see SweepGeometry) ----
    static constexpr bool can_continue = !COOP && (2 * G) % P == 0;

    STST_DEVICE static bool claim(SweepGeometry const &g, std::uint32_t item, int lane) {
        std::uint32_t got = 0;
        if (lane == 0)
            got = atomicCAS(g.claims + item, 0u, 1u) == 0u ? 1u : 0u;
        return __builtin_amdgcn_readfirstlane(got) != 0;
    }
    STST_DEVICE static bool chunk_is_interior(SweepGeometry const &g, int strip, int ya, int yb) {
        const int xw0 = strip * OW - GX;
        return INTERIOR_VARIANT && xw0 >= 0 && xw0 + UW <= g.grid_w && ya - G >= 0 && yb + G <= g.grid_h;
    }

    template <bool SKIP_CONSTANTS = false> STST_DEVICE static void entry_persistent(Args const &a, std::uint32_t *lds) {
        SweepGeometry const &g = a.geo;
        const int lane = int(threadIdx.x) & (wave_size - 1);
        const int wib = int(__builtin_amdgcn_readfirstlane(threadIdx.x / wave_size));
        // One ticket per wave, in dispatch order.  The first starts_per_strip * n_strips tickets (one residency
        // round) start runs evenly spaced in every strip; a run goes down its strip chunk by chunk until the chunk
        // below belongs to somebody else -- who by the same rule takes care of everything below it -- so every chunk
        // is swept exactly once.  The later tickets (phases 1, 2, ...) point at chunks inside those runs' ranges:
        // their waves only start when a slot frees up, find their chunk taken if the run above has got there (then
        // they end at once), and otherwise split what a slow or late run has left.  No wave waits for another.
        const std::uint32_t ticket = __builtin_amdgcn_readfirstlane(blockIdx.x * std::uint32_t(waves_per_block) + std::uint32_t(wib));
        if (ticket >= g.cursor_end)
            return;
        const std::uint32_t per_phase = g.starts_per_strip * g.n_strips;
        const std::uint32_t phase = ticket / per_phase, rest = ticket % per_phase;
        const int strip = int(rest % g.n_strips);
        std::uint32_t chunk = (rest / g.n_strips) * g.start_stride + phase * g.phase_step;
        if (chunk >= g.n_fine || !claim(g, chunk * g.n_strips + std::uint32_t(strip), lane))
            return;
        for (;;) {
            const int ya = g.out_begin + int(chunk * g.fine_rows);
            int yb = ya + int(g.fine_rows);
            yb = yb < g.out_end ? yb : g.out_end;
            const bool interior = chunk_is_interior(g, strip, ya, yb);
            // the chunk below, if it is free and runs the same code path (interior / edge): keep streaming
            auto more_rows = [&](int done_to) __attribute__((always_inline)) -> int {
                if (done_to >= g.out_end)
                    return done_to;
                const std::uint32_t next = std::uint32_t(done_to - g.out_begin) / g.fine_rows;
                int next_end = done_to + int(g.fine_rows);
                next_end = next_end < g.out_end ? next_end : g.out_end;
                if (chunk_is_interior(g, strip, done_to, next_end) != interior)
                    return done_to;
                return claim(g, next * g.n_strips + std::uint32_t(strip), lane) ? next_end : done_to;
            };
            int done_to;
            if constexpr (INTERIOR_VARIANT) {
                if (interior)
                    done_to = run<false, SKIP_CONSTANTS>(a, lane, strip, ya, yb, wib, lds, more_rows);
                else
                    done_to = run<true, SKIP_CONSTANTS>(a, lane, strip, ya, yb, wib, lds, more_rows);
            } else {
                done_to = run<true, SKIP_CONSTANTS>(a, lane, strip, ya, yb, wib, lds, more_rows);
            }
            // the code path changes at the chunk below (or the feed did not end on a chunk boundary): a fresh run
            if (done_to >= g.out_end)
                return;
            chunk = std::uint32_t(done_to - g.out_begin) / g.fine_rows;
            if (!claim(g, chunk * g.n_strips + std::uint32_t(strip), lane))
                return;
        }
    }
};

// The sweep SweepTuning<F, SOA> asks for, at blocking depth T.
template <typename F, bool SOA, int T = SweepTuning<F, SOA>::max_generations, bool INLINE_TDV = false>
using SweepOf = Sweep<F, SOA, T, SweepTuning<F, SOA>::cells_per_lane, SweepTuning<F, SOA>::prefetch_rows,
                      SweepTuning<F, SOA>::interior_variant, cooperative_for<F, SOA>(), cooperative_debug_for<F, SOA>(),
                      INLINE_TDV>;

// MIN_WAVES = waves per SIMD the register allocator must leave room for (launch-bounds semantics).
template <typename SW, int MIN_WAVES = 1, bool SKIP_CONSTANTS = false, bool PERSISTENT = false>
__global__ void __launch_bounds__(256, MIN_WAVES) sweep_kernel(const typename SW::Args args) {
    __shared__ std::uint32_t edge_columns[SW::LDS_WORDS]; // cooperative strips only (one word otherwise)
    if constexpr (PERSISTENT)
        SW::template entry_persistent<SKIP_CONSTANTS>(args, edge_columns);
    else
        SW::template entry<SKIP_CONSTANTS>(args, edge_columns);
}

// ------------------------------------------------------------------ host side
inline int env_int(const char *name, int fallback) {
    const char *v = std::getenv(name);
    return (v && *v) ? std::atoi(v) : fallback;
}

// Rows of output per wave.  A wave costs (rows + 2*G warm-up rows + prologue); the chip keeps
// S = CUs * resident workgroups * 4 waves in flight and back-fills as waves retire, so
//   time ~ waves * cost / S  +  alpha * cost      (alpha ~ 0.5: the ragged tail of the last waves)
// with waves = strips * out_rows / rows.  Minimising over rows gives the closed form below: long
// chunks waste the tail, short chunks waste warm-up rows.  (Measured optimum for Jacobi 16384^2,
// T = 8: ~135 rows; the formula gives 133.)
inline int pick_chunk_rows(int out_rows, unsigned n_strips, int halo_rows, int resident_blocks,
                           int waves_per_block) {
    int forced = env_int("STSTHIP_CHUNK_ROWS", 0);
    if (forced > 0)
        return std::min(forced, std::max(out_rows, 1));
    int cus = 256;
    ststhip_compute_units(&cus);
    // launches running side by side (row strips of the pass driver) fill each other's tails: the tail
    // weight shrinks by their number (fitted to chunk sweeps of Jacobi 16384^2 and HotSpot 8192^2)
    const int side_by_side = std::max(1, ststhip_launch_concurrency());
    const double slots = double(cus) * std::max(resident_blocks, 1) * waves_per_block;
    // tail weight: 0.5 for a launch that has the chip to itself; launches that run side by side (their boundary
    // bands on streams of their own) want slightly longer chunks still (profiles/r02_ab_bands_beside.txt)
    const double alpha = env_int("STSTHIP_TAIL_PERMILLE", side_by_side > 1 ? 350 : 500) / 1000.0 / side_by_side;
    const double overhead = 2.0 * halo_rows + 8.0;
    double rows = std::sqrt(double(out_rows) * double(n_strips) * overhead / (alpha * slots));
    rows = std::max(rows, 1.0);
    long chunks = std::max<long>(1, long(double(out_rows) / rows + 0.5));
    // snap to a whole number of residency rounds (just below it) when that is a small change:
    // a launch of k*S + a few waves pays for a nearly empty extra round
    const double rounds = double(chunks) * n_strips / slots;
    if (rounds >= 0.75) {
        const long k = std::max<long>(1, long(rounds + 0.5));
        const long snapped = long(k * slots) / long(std::max(1u, n_strips));
        if (snapped >= 1 && snapped * 4 >= chunks * 3 && snapped * 4 <= chunks * 5)
            chunks = snapped;
    }
    chunks = std::min<long>(chunks, std::max(out_rows, 1));
    return int((out_rows + chunks - 1) / chunks);
}

// Chunk lengths of a launch.  All chunks have g.chunk_rows rows, except that the tail of the wave grid is
// cut finer: STSTHIP_TAPER = "permille:split[,permille:split...]" makes the last `permille` of the rows
// chunks of chunk_rows/split rows (later entries refine the end further; they must shrink).
inline void plan_tiers(SweepGeometry &g, int out_rows) {
    g.n_tiers = 1;
    g.tier_first[0] = 0;
    g.tier_rows[0] = g.chunk_rows;
    g.tier_begin[0] = g.out_begin;
    // Default (profiles/r01_tune_taper.txt): the last 12 % of the rows in quarter-length chunks when the
    // launch has the chip to itself (+7 % for a full-grid Jacobi launch, +2..4 % HotSpot / FDTD); launches
    // that run side by side already fill each other's ends and lose 1-6 % with shorter chunks.
    const char *spec = std::getenv("STSTHIP_TAPER");
    std::string text = spec ? spec : (ststhip_launch_concurrency() == 1 ? "120:4" : "150:2");
    int done_rows = 0; // rows covered by the tiers closed so far
    unsigned done_chunks = 0;
    std::size_t at = 0;
    int previous_start = 0;
    while (at < text.size() && g.n_tiers < unsigned(SweepGeometry::max_tiers)) {
        const int permille = std::atoi(text.c_str() + at);
        const std::size_t colon = text.find(':', at);
        if (colon == std::string::npos)
            break;
        const int split = std::atoi(text.c_str() + colon + 1);
        const std::size_t comma = text.find(',', colon);
        at = comma == std::string::npos ? text.size() : comma + 1;
        const int rows = (g.chunk_rows + std::max(split, 1) - 1) / std::max(split, 1);
        // the tier starts at a chunk boundary of the tier before it
        const int current = g.tier_rows[g.n_tiers - 1];
        int start = out_rows - int(std::int64_t(out_rows) * std::clamp(permille, 0, 1000) / 1000);
        start = done_rows + (std::max(start - done_rows, 0) / current) * current;
        if (split <= 1 || rows < 4 || rows >= current || start <= previous_start || start >= out_rows)
            continue;
        done_chunks += unsigned((start - done_rows) / current);
        done_rows = start;
        previous_start = start;
        g.tier_first[g.n_tiers] = done_chunks;
        g.tier_rows[g.n_tiers] = rows;
        g.tier_begin[g.n_tiers] = g.out_begin + start;
        g.n_tiers++;
    }
    const int last = g.tier_rows[g.n_tiers - 1];
    g.n_chunks = done_chunks + unsigned((out_rows - done_rows + last - 1) / last);
    g.tier_first[g.n_tiers] = g.n_chunks;
}

template <typename SW> constexpr int P_rows() { return SW::prefetch_depth; }

// One kernel launch = T generations over global rows [out_begin, out_end).  `tdv`: the T time-dependent values
// evaluated on the host, or nullptr when the kernel takes them from the pass driver's device table
// (ststhip_current_tdv_table) or evaluates them itself (INLINE_TDV).
template <typename F, bool SOA, int T, bool INLINE_TDV = false>
void launch_sweep(F const &f, typename F::Cell const &halo, typename F::TimeDependentValue const *tdv,
                  ststhip_domain const &dom, PlaneSet<typename F::Cell, SOA> const &src,
                  PlaneSet<typename F::Cell, SOA> const &dst, std::uint64_t out_begin,
                  std::uint64_t out_end, std::uint64_t iteration, ststhip_stream stream) {
    using Tuning = SweepTuning<F, SOA>;
    using SW = SweepOf<F, SOA, T, INLINE_TDV>;
    constexpr bool coop = cooperative_for<F, SOA>();
    if (out_end <= out_begin || dom.global_width == 0)
        return;
    // a driver may leave a hole in the row range (ststhip_launch_row_hole): the two boundary bands of a row strip as
    // ONE launch -- rows [out_begin, hole) and [hole end, out_end), the interior in between is another launch's
    std::uint64_t hole_begin = 0, hole_end = 0;
    ststhip_launch_row_hole(&hole_begin, &hole_end);
    if constexpr (has_narrow_form<F, SOA>()) {
        // grids that cannot fill the chip with this shape: the same function on the narrowest lanes.  So are launches
        // of a few rows -- the boundary bands of row strips: a band is a dependent chain of 3g row steps per wave that
        // runs beside a busy interior, and with one cell per lane instead of several a step is that much shorter
        // (the next pass of two strips and the ghost-row exchange wait for it)
        static const std::uint64_t narrow_form_cells = std::uint64_t(env_int("STSTHIP_NARROW_FORM_KCELLS", 20000)) * 1000;
        static const std::uint64_t narrow_band_rows = std::uint64_t(env_int("STSTHIP_NARROW_BAND_ROWS", 0));
        const std::uint64_t launch_rows = (out_end - out_begin) - (hole_end - hole_begin);
        if (dom.global_height * dom.global_width <= narrow_form_cells || launch_rows <= narrow_band_rows) {
            launch_sweep<NarrowForm<F>, SOA, T, INLINE_TDV>(NarrowForm<F>(f), halo, tdv, dom, src, dst, out_begin, out_end,
                                                            iteration, stream);
            return;
        }
    }
    if (dom.global_height >= (1ull << 31) || dom.global_width >= (1ull << 31))
        throw std::range_error("grid extents must be below 2^31 per dimension");

    SweepGeometry g;
    g.grid_h = std::int32_t(dom.global_height);
    g.grid_w = std::int32_t(dom.global_width);
    g.row_origin = std::int32_t(dom.row_origin);
    g.load_lo = std::int32_t(std::max<std::int64_t>(0, dom.row_origin));
    g.load_hi = std::int32_t(std::min<std::int64_t>(std::int64_t(dom.global_height),
                                                    dom.row_origin + std::int64_t(dom.local_rows)));
    g.out_begin = std::int32_t(out_begin);
    g.out_end = std::int32_t(out_end);
    const void *kernel = reinterpret_cast<const void *>(&sweep_kernel<SW, Tuning::min_waves_per_simd>);
    // per kernel instantiation; a property of the code object and the architecture, so racing host
    // threads would store the same number
    static std::atomic<int> resident_blocks_cache{0};
    int resident_blocks = resident_blocks_cache.load(std::memory_order_relaxed);
    if (resident_blocks == 0) {
        check(ststhip_occupancy(kernel, waves_per_block * wave_size, 0, &resident_blocks), "occupancy query");
        resident_blocks_cache.store(resident_blocks, std::memory_order_relaxed);
    }
    // per-field planes of fields F only copies: from the third pass of a run on the target holds them already
    if constexpr (SOA && constant_plane_mask<F>() != 0)
        if (ststhip_target_holds_constants() && env_int("STSTHIP_SKIP_CONSTANT_STORES", 1))
            kernel = reinterpret_cast<const void *>(&sweep_kernel<SW, Tuning::min_waves_per_simd, true>);
    g.n_strips = unsigned((dom.global_width + SW::OW - 1) / SW::OW); // units: waves, or workgroups (coop)
    if (hole_begin < hole_end) {
        if (hole_begin <= out_begin || hole_end >= out_end)
            throw std::invalid_argument("the row hole must lie strictly inside the launch's row range");
        const int part[2] = {int(hole_begin - out_begin), int(out_end - hole_end)};
        const std::uint64_t part_begin[2] = {out_begin, hole_end};
        const int wanted = pick_chunk_rows(std::max(part[0], part[1]), coop ? g.n_strips * waves_per_block : g.n_strips,
                                           SW::G, resident_blocks, int(waves_per_block));
        g.n_tiers = 2;
        unsigned first = 0;
        for (int t = 0; t < 2; t++) {
            // equal chunks that tile the part exactly (a chunk must not reach into the hole)
            int n = std::max(1, (part[t] + wanted / 2) / std::max(wanted, 1));
            while (part[t] % n != 0)
                n--;
            g.tier_first[t] = first;
            g.tier_rows[t] = part[t] / n;
            g.tier_begin[t] = std::int32_t(part_begin[t]);
            first += unsigned(n);
        }
        g.tier_first[2] = first;
        g.n_chunks = first;
        g.chunk_rows = std::max(g.tier_rows[0], g.tier_rows[1]);
    } else {
        g.chunk_rows = pick_chunk_rows(int(out_end - out_begin), coop ? g.n_strips * waves_per_block : g.n_strips,
                                       SW::G, resident_blocks, int(waves_per_block));
        g.n_chunks = unsigned((out_end - out_begin + g.chunk_rows - 1) / g.chunk_rows);
        plan_tiers(g, int(out_end - out_begin));
    }
    if (dom.pitch >= (1ull << 31))
        throw std::range_error("the pitch must be below 2^31 elements");
    g.pitch = dom.pitch;
    g.pitch32 = std::uint32_t(dom.pitch);
    g.iteration = iteration;
    // measured (profiles/r01_xcd_remap.txt): 4 % fewer HBM reads, but no gain in time for these
    // VALU-bound kernels, so the remap is off unless asked for
    g.xcd_remap = env_int("STSTHIP_XCD_REMAP", 0) ? 1u : 0u;
    g.last_chunk_early = (out_end + SW::G > dom.global_height && env_int("STSTHIP_LAST_CHUNK_EARLY", 1)) ? 1u : 0u;

    using TDV = typename F::TimeDependentValue;
    TDV const *table = nullptr;
    if constexpr (!std::is_empty_v<TDV> && !INLINE_TDV) {
        const void *base = nullptr;
        std::uint64_t first_iteration = 0, n_values = 0, value_size = 0;
        ststhip_current_tdv_table(&base, &first_iteration, &n_values, &value_size);
        if (base && value_size == sizeof(TDV) && iteration >= first_iteration &&
            iteration + std::uint64_t(T) <= first_iteration + n_values)
            table = static_cast<TDV const *>(base) + (iteration - first_iteration);
        else if (!tdv)
            throw std::invalid_argument("no time-dependent values for this launch");
    }
    // persistent waves: one residency round of waves that claim fine row chunks and continue downwards
    void *claim_words = nullptr;
    bool persistent = false;
    g.fine_rows = g.n_fine = g.starts_per_strip = g.start_stride = g.phase_step = g.cursor_end = 0;
    g.claims = g.cursor = nullptr;
    if constexpr (persistent_for<F, SOA>() && SW::can_continue) {
        const int rows = int(out_end - out_begin);
        int fine = env_int("STSTHIP_FINE_ROWS", 32);
        fine = round_up(std::max(fine, P_rows<SW>()), P_rows<SW>());
        if (env_int("STSTHIP_PERSISTENT", 1) && rows >= 8 * fine && hole_begin == hole_end) {
            int cus = 256;
            ststhip_compute_units(&cus);
            std::uint32_t slots = std::uint32_t(cus) * std::uint32_t(std::max(resident_blocks, 1)) * waves_per_block /
                                  std::uint32_t(std::max(1, ststhip_launch_concurrency()));
            slots = std::uint32_t(std::uint64_t(slots) * std::uint64_t(env_int("STSTHIP_PERSISTENT_FILL_PERMILLE", 1000)) / 1000);
            g.fine_rows = std::uint32_t(fine);
            g.n_fine = std::uint32_t((rows + fine - 1) / fine);
            g.starts_per_strip = std::min(std::max(slots / g.n_strips, 1u), g.n_fine);
            g.start_stride = (g.n_fine + g.starts_per_strip - 1) / g.starts_per_strip;
            g.phase_step = std::max(1u, g.start_stride / std::uint32_t(std::max(1, env_int("STSTHIP_PERSISTENT_PHASES", 4))));
            g.cursor_end = ((g.start_stride + g.phase_step - 1) / g.phase_step) * g.starts_per_strip * g.n_strips;
            const std::size_t n_words = std::size_t(g.n_fine) * g.n_strips;
            check(ststhip_malloc_async(&claim_words, n_words * 4, stream), "claim table");
            check(ststhip_memset(claim_words, 0, n_words * 4, stream), "claim table");
            g.claims = static_cast<std::uint32_t *>(claim_words);
            g.cursor = nullptr;
            kernel = reinterpret_cast<const void *>(&sweep_kernel<SW, Tuning::min_waves_per_simd, false, true>);
            if constexpr (SOA && constant_plane_mask<F>() != 0)
                if (ststhip_target_holds_constants() && env_int("STSTHIP_SKIP_CONSTANT_STORES", 1))
                    kernel = reinterpret_cast<const void *>(&sweep_kernel<SW, Tuning::min_waves_per_simd, true, true>);
            persistent = true;
        }
    }
    // transition functions need not be default-constructible: build the argument block in one go
    typename SW::Args args = [&]<std::size_t... Is>(std::index_sequence<Is...>) {
        return typename SW::Args{f, halo, {((table || !tdv) ? TDV{} : tdv[Is])...}, table, src, dst, g};
    }(std::make_index_sequence<std::size_t(T)>{});

    const unsigned units = g.n_strips * g.n_chunks;
    unsigned blocks = coop ? units : (units + waves_per_block - 1) / waves_per_block;
    if (persistent) {
        int cus = 256;
        ststhip_compute_units(&cus);
        const unsigned resident = unsigned(cus) * unsigned(std::max(resident_blocks, 1)) /
                                  unsigned(std::max(1, ststhip_launch_concurrency()));
        (void)resident;
        blocks = std::max(1u, (g.cursor_end + waves_per_block - 1) / waves_per_block); // one wave per ticket
    }
    void *kernel_args[] = {&args};
    const int launched = ststhip_launch(kernel, blocks, 1, 1, waves_per_block * wave_size, 1, 1, kernel_args, 0, stream);
    if (claim_words)
        ststhip_free_async(claim_words, stream); // released behind the launch
    check(launched, "sweep launch");
}

// Runtime n_generations -> compiled T (powers of two up to the tuning's maximum).
template <typename F, bool SOA, int T = SweepTuning<F, SOA>::max_generations, bool INLINE_TDV = false>
void dispatch_sweep(int n_generations, F const &f, typename F::Cell const &halo,
                    typename F::TimeDependentValue const *tdv, ststhip_domain const &dom,
                    PlaneSet<typename F::Cell, SOA> const &src,
                    PlaneSet<typename F::Cell, SOA> const &dst, std::uint64_t out_begin,
                    std::uint64_t out_end, std::uint64_t iteration, ststhip_stream stream) {
    if (n_generations == T) {
        launch_sweep<F, SOA, T, INLINE_TDV>(f, halo, tdv, dom, src, dst, out_begin, out_end, iteration, stream);
    } else if constexpr (T > 1) {
        dispatch_sweep<F, SOA, T / 2, INLINE_TDV>(n_generations, f, halo, tdv, dom, src, dst, out_begin,
                                      out_end, iteration, stream);
    } else {
        throw std::invalid_argument("n_generations is not a compiled temporal-blocking depth");
    }
}

// Largest compiled depth that fits into `remaining` generations.
template <typename F, bool SOA> inline int next_pass_depth(std::uint64_t remaining) {
    int t = SweepTuning<F, SOA>::max_generations;
    int cap = env_int("STSTHIP_MAX_GENERATIONS", t);
    while (t > 1 && (std::uint64_t(t) > remaining || t > cap))
        t /= 2;
    return t;
}

} // namespace internal
} // namespace hip
} // namespace stencil
