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
//  * Staged sweeps (SweepTuning::stages = W > 1): the W waves of a workgroup share ONE column strip as a software
//    pipeline over the levels -- stage s runs levels [s*S/W, (s+1)*S/W), takes its input rows from stage s-1 and
//    hands its output rows to stage s+1 through an LDS ring (batches of P rows, one workgroup barrier per batch).
//    A wave then keeps the register window of S/W levels only, a row chunk is W times longer for the same number of
//    waves (the 2G warm-up rows are paid once per W waves), and a launch of few rows -- the boundary band of a row
//    strip -- is a dependent chain of S/W instead of S levels per row.  This is where north_star's "LDS tiles" sit
//    in this design: between the stages of the temporal pipeline, not under the column halo.
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
//   stages               (optional member, default 1) waves of a workgroup that share ONE column strip as a software
//                        pipeline over the levels: stage s runs levels [s*S/stages, (s+1)*S/stages) and hands every row
//                        it emits to stage s+1 through an LDS ring (see the head of Sweep.hpp).  Depths whose level
//                        count is not a multiple of `stages` run with the largest divisor that is (12 generations in
//                        4 stages; its halvings 6 and 3 in 3, 1 in 1).
//   streaming_stores     (optional member) store results with the non-temporal hint (streaming_stores_for below)
//   trapezoid_fill       (optional member) skip the levels that are not due yet while a wave's pipeline fills;
//                        default: cells of up to four words per generation (measured: Jacobi +3 %, HotSpot
//                        +5 %, Conway +4 %, FDTD -2..-8 %: profiles/r01_ab_trapezoid_fill.txt)
//   narrow_form          (optional member, default true) grids too small to fill the chip may be swept with the
//                        one-cell-per-lane form of the same function (NarrowForm below); explicit shapes say false
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
        // (two-word cells on planes: K = 1 in rounds 1-2; on four stages two cells per lane win there too --
        // HotSpot 8192^2 planes K = 1: 1750, K = 2: 1820 at T = 8, 2090 at T = 12, profiles/r03_tune_staged.txt)
        int k = (W == 1) ? 4 : (W == 2 ? 2 : 1);
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
    // One-word cells of radius 1 without sub-iterations (the Jacobi family, the Game of Life): sixteen generations on
    // four stages of four levels are COMPILED as well, because what such a function costs per cell decides between 8
    // and 16 and the rule cannot see it -- the reference's five-point Jacobi source is nine operations per cell with
    // -ffp-contract=off (fastest at 8) and five with fused multiply-adds (fastest at 16, like the library's
    // uniform-coefficient form).  The pass driver measures (ststhip_sweep_desc::alt_generations); unmeasured launches
    // run `default_generations` = 8, round 3's depth.
    static constexpr bool thin_deep = (W == 1 && NS == 1 && R == 1);
    static constexpr int pick_t(int k) {
        if (thin_deep && nominal_window(16, k) <= window_limit && geometry_ok(16, k))
            return 16;
        // two-word cells with one sub-iteration (HotSpot): twelve generations on four stages of three levels
        if (W == 2 && NS == 1 && R == 1 && nominal_window(12, k) <= window_limit && geometry_ok(12, k))
            return 12;
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

    // Stages (Sweep.hpp head): four waves per column strip where the levels divide by four, else two, as long as
    // the row rings between the stages stay within 48 KiB of LDS per workgroup (three workgroups per CU).  Measured on
    // MI355X (profiles/r03_tune_staged.txt), full grids / 2048-row strips: Jacobi5General +3 % / +14 %, the packed
    // Game of Life +29 %, HotSpot +18 %, FDTD +13...23 %; launches of a few rows (boundary bands) run a third as long.
    static constexpr int ring_bytes(int w, int k, int p) {
        return (w - 1) * 2 * p * 64 * k * int((sizeof(typename F::Cell) + 3) / 4) * 4;
    }
    static constexpr int lds_limit = 48 * 1024; // three workgroups per CU at least
    struct Shape {
        int t, p, w;
    };
    static constexpr int pick_w(int t, int k, int p) {
        for (int w : {4, 2})
            if ((t * NS) % w == 0 && ring_bytes(w, k, p) <= lds_limit)
                return w;
        return 1;
    }
    // Fat cells with sub-iterations (eight words and more per generation: FDTD, convection) are limited by the
    // register window, not by anything else; a stage keeps the window of its own levels only, so with stages the
    // launch can be deeper than one wave could hold: the deepest T <= 8 whose window PER STAGE stays within what the
    // unstaged shape was admitted with (128 words, or one generation's window where even that is more), the rings
    // within the LDS budget (batches of 2R rows).  FDTD: T = 6 -> 8 on four stages (the precompiled form's
    // measured optimum, profiles/r03_tune_staged.txt); convection's 88-byte fp64 cell with three sub-iterations: T = 1 ->
    // 2 on three stages, 134.6 -> 88.4 us per iteration at res = 1024 (T = 2 on two stages spills: 132.8; T = 1 on
    // three: 148.3; profiles/r03_convection.txt).
    static constexpr Shape pick_shape(int k) {
        const int t0 = pick_t(k);
        const int p0 = pick_p(t0, k);
        if (R == 1 && W * NS >= 8) {
            const int admitted = std::max(128, nominal_window(1, k));
            for (int t : {8, 4, 2})
                for (int w : {4, 3, 2})
                    if (t > t0 && (t * NS) % w == 0 && nominal_window(t, k) / w <= admitted && geometry_ok(t, k) &&
                        ring_bytes(w, k, 2 * R) <= lds_limit)
                        return Shape{t, 2 * R, w}; // batches of 2R rows: the smallest rings
        }
        return Shape{t0, p0, pick_w(t0, k, p0)};
    }

  public:
    static constexpr int cells_per_lane = pick_k();
    static constexpr int max_generations = pick_shape(cells_per_lane).t;
    static constexpr int prefetch_rows = pick_shape(cells_per_lane).p;
    static constexpr bool interior_variant = (W * NS <= 16);
    static constexpr int min_waves_per_simd = 1;
    static constexpr int stages = pick_shape(cells_per_lane).w;
    // the depth unmeasured launches run with (0: max_generations); see thin_deep above
    static constexpr int default_generations = (max_generations == 16 && thin_deep) ? 8 : 0;
};

namespace internal {
// The same transition function swept with the narrowest lanes (one cell per lane for radius 1): four times as many
// waves for a thin-cell function whose default shape holds four cells per lane.  The launcher switches to it for
// grids too small to fill the chip with the default shape (narrow_form_cells below): a launch is then the latency
// of one wave through its warm-up rows, and narrower lanes mean more waves sharing it.  Measured, Jacobi5General
// (profiles/r02_small_grids.txt, independent waves): 256^2 26.7 -> 46.7, 512^2 92 -> 158, 1024^2 310 -> 462,
// 2048^2 854 -> 1025; with four stages (profiles/r03_small_grids.txt) 256^2 37 -> 62, 1024^2 432 -> 551, 2048^2
// 1088 -> 1243, 3072^2 1727 -> 1422: the default shape wins from about 2500^2 (6 M cells) on.
template <typename F> struct NarrowForm : public F {
    NarrowForm(F const &f) : F(f) {}
};

template <typename F, bool SOA> constexpr int stages_for() {
    if constexpr (requires { SweepTuning<F, SOA>::stages; })
        return SweepTuning<F, SOA>::stages;
    else
        return 1;
}

// SweepTuning<F, SOA>::pinned_loads (optional member; default: cells of up to four words; staged sweeps): keep stage 0's HBM load of row
// y + P where the source has it -- right after row y has been taken out of its registers -- instead of letting the
// scheduler hoist the batch's loads to its top into fresh registers, which it then has to wait for at the loop's end
// (a stage's batch is a quarter as long as an independent wave's, and so was the time a load had to arrive).
// Measured (profiles/r03_tune_staged.txt): Jacobi5Uniform single launches 5640 -> 5770, 2048-row strip 3670 -> 3890,
// with two strips 5760 -> 5730; HotSpot planes 2050 -> 2100 (same box, explicit shapes: 1981 / 2031 -> 2035 / 2069).  A fat
// cell's row is several loads per lane already and the scheduler's own placement is the better one: FDTD two planes
// 452 -> 459, AoS 415 -> 425 with the loads left free.
template <typename F, bool SOA> constexpr bool pinned_loads_for() {
    if constexpr (requires { SweepTuning<F, SOA>::pinned_loads; })
        return SweepTuning<F, SOA>::pinned_loads;
    else
        return cell_words<typename F::Cell, SOA>() <= 4;
}

// SweepTuning<F, SOA>::tail_permille_beside / taper_beside (optional members, defaults 350 / true): the chunk model's
// tail weight and whether the last chunks of a launch are cut shorter, for launches that run side by side with
// others (the pass driver's row strips).  Fitted per kernel: Jacobi5Uniform 16384^2 on two strips 6375 -> 6470 with
// 200 / false (16 chunks of ~500 rows per strip instead of 25 of 328 with a tapered end), HotSpot 8192^2 loses 14 % with
// the same setting (profiles/r03_stage_experiments.txt).
template <typename F, bool SOA> constexpr int tail_permille_beside_for() {
    if constexpr (requires { SweepTuning<F, SOA>::tail_permille_beside; })
        return SweepTuning<F, SOA>::tail_permille_beside;
    else
        return 350;
}
template <typename F, bool SOA> constexpr bool taper_beside_for() {
    if constexpr (requires { SweepTuning<F, SOA>::taper_beside; })
        return SweepTuning<F, SOA>::taper_beside;
    else
        return true;
}

// SweepTuning<F, SOA>::default_generations (optional member, default 0 = max_generations): the blocking depth of
// launches nobody has measured; a value below max_generations makes the pass driver time both on the first long call
// for a grid shape and keep the faster (ststhip_sweep_desc::alt_generations).
template <typename F, bool SOA> constexpr int default_generations_for() {
    if constexpr (requires { SweepTuning<F, SOA>::default_generations; })
        return SweepTuning<F, SOA>::default_generations > 0 ? SweepTuning<F, SOA>::default_generations
                                                             : SweepTuning<F, SOA>::max_generations;
    else
        return SweepTuning<F, SOA>::max_generations;
}

// SweepTuning<F, SOA>::pinned_stores (optional member; default: cells of up to four words; staged sweeps): keep the store
// of a finished row -- into the LDS ring, or to HBM in the last stage -- right behind the row instead of letting the
// scheduler collect a batch's stores at its end, where their latency sits in front of the barrier.  Measured, same box, bit
// for bit the same results (profiles/r04_micro_variants.txt): uniform Jacobi two strips +2.3 %, single launches +2.3 %,
// 2048-row strip +5.9 %; general Jacobi +4.9 %; HotSpot +2.4 %; packed Game of Life +2.5 %; FDTD (eight words) -2.5 %.
template <typename F, bool SOA> constexpr bool pinned_stores_for() {
    if constexpr (requires { SweepTuning<F, SOA>::pinned_stores; })
        return SweepTuning<F, SOA>::pinned_stores;
    else
        return cell_words<typename F::Cell, SOA>() <= 4;
}

// SweepTuning<F, SOA>::scalar_function (optional member; default: transition functions of up to nine 32-bit words): the
// function object lives in scalar registers for the whole kernel instead of being read from the kernel-argument segment
// wherever it is used.  The compiler re-loaded a coefficient inside the row loop, and a scalar load in flight makes the
// wait for the FIRST row from LDS a wait for all of them (scalar loads return out of order: s_waitcnt lgkmcnt(0) instead
// of lgkmcnt(3)).  Uniform Jacobi +1.6 %, general Jacobi +1.8 %, HotSpot +0.4 %; FDTD's fourteen words -12 % (scalar
// register pressure), hence the size limit.
template <typename F, bool SOA> constexpr bool scalar_function_for() {
    if constexpr (requires { SweepTuning<F, SOA>::scalar_function; })
        return SweepTuning<F, SOA>::scalar_function;
    else
        return sizeof(F) % 4 == 0 && sizeof(F) <= 36 && !std::is_empty_v<F>;
}

template <typename F, bool SOA> constexpr bool trapezoid_fill_for() {
    if constexpr (requires { SweepTuning<F, SOA>::trapezoid_fill; })
        return SweepTuning<F, SOA>::trapezoid_fill;
    else
        return cell_words<typename F::Cell, SOA>() * int(F::n_subiterations) <= 4;
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
} // namespace internal

template <typename F, bool SOA> struct SweepTuning<internal::NarrowForm<F>, SOA> {
    static constexpr int cells_per_lane = internal::ceil_pow2(int(F::stencil_radius));
    static constexpr int max_generations = SweepTuning<F, SOA>::max_generations;
    static constexpr int prefetch_rows = SweepTuning<F, SOA>::prefetch_rows;
    static constexpr bool interior_variant = SweepTuning<F, SOA>::interior_variant;
    static constexpr int min_waves_per_simd = SweepTuning<F, SOA>::min_waves_per_simd;
    static constexpr int stages = internal::stages_for<F, SOA>();
    static constexpr int default_generations = internal::default_generations_for<F, SOA>();
    static constexpr bool narrow_form = false;
    static constexpr bool trapezoid_fill = internal::trapezoid_fill_for<F, SOA>();
    static constexpr bool streaming_stores = internal::streaming_stores_for<F, SOA>();
};

namespace internal {

template <typename T> struct is_narrow_form : std::false_type {};
template <typename F> struct is_narrow_form<NarrowForm<F>> : std::true_type {};

// Does F have a narrower shape worth compiling?  Only functions whose default shape holds several cells per lane
// (and whose tuning does not say `narrow_form = false`: an explicit shape is launched as it is); the wave of the
// narrow shape must still produce columns at full depth.
template <typename F, bool SOA> constexpr bool has_narrow_form() {
    if constexpr (is_narrow_form<F>::value) {
        return false;
    } else {
        constexpr int k = ceil_pow2(int(F::stencil_radius));
        constexpr int g = int(F::stencil_radius) * int(F::n_subiterations) * SweepTuning<F, SOA>::max_generations;
        bool allowed = true;
        if constexpr (requires { SweepTuning<F, SOA>::narrow_form; })
            allowed = SweepTuning<F, SOA>::narrow_form;
        return allowed && SweepTuning<F, SOA>::cells_per_lane > k && wave_size * k - 2 * round_up(g, k) >= 16 * k;
    }
}

// Geometry of one sweep launch, in global grid coordinates.
struct SweepGeometry {
    std::int32_t grid_h, grid_w;        // stencil.grid_range
    std::int32_t row_origin;            // global row of buffer row 0
    std::int32_t load_lo, load_hi;      // global rows present in the source buffers
    std::int32_t out_begin, out_end;    // global rows to produce
    // columns (a buffer of a 2-D block decomposition holds a column range of the grid; whole rows otherwise:
    // col_origin = 0, col_lo = 0, col_hi = grid_w, out columns = all)
    std::int32_t col_origin;            // global column of buffer column 0
    std::int32_t col_lo, col_hi;        // global columns present in the source buffers (inside the grid)
    std::int32_t out_col_begin, out_col_end; // global columns to produce
    std::int32_t chunk_rows;            // rows of output per unit of the wave grid
    std::uint32_t n_strips, n_chunks;   // wave grid, in units (a wave, or the workgroup of a staged sweep)
    // The last units of the grid (dispatched last) take shorter chunks, so that the ragged end of a
    // launch -- SIMDs left with one or two waves -- is short.  Tier t covers chunks
    // [tier_first[t], tier_first[t+1]) with tier_rows[t] rows each; tier 0 starts at out_begin.
    static constexpr int max_tiers = 4;
    std::uint32_t n_tiers;
    std::uint32_t tier_first[max_tiers + 1];
    std::int32_t tier_rows[max_tiers];
    std::int32_t tier_begin[max_tiers]; // first output row of the tier
    std::uint32_t pitch32;              // elements between rows (below 2^31, checked at launch)
    std::uint64_t iteration;            // generation index of the first level
    std::uint32_t xcd_remap;            // 1: give every XCD a contiguous range of the wave grid
    std::uint32_t last_chunk_early;     // 1: the last row chunk is dispatched second instead of last
};

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

// Independent waves (stages = 1): 4 waves per workgroup, 2 or 1 are 1-3 % slower (profiles/r01_tune_taper.txt).
constexpr int independent_waves_per_block = 4;

// Largest divisor of `levels` that is at most `wanted`: the stage count a launch of `levels` levels runs with.
constexpr int stages_dividing(int levels, int wanted) {
    int w = wanted < 1 ? 1 : wanted;
    while (levels % w != 0)
        w--;
    return w;
}

template <typename F, bool SOA, int T, int K, int P, bool INTERIOR_VARIANT, int STAGES = 1, bool INLINE_TDV = false>
struct Sweep {
    using Cell = typename F::Cell;
    using TDV = typename F::TimeDependentValue;
    using Planes = PlaneSet<Cell, SOA>;
    using StencilImpl = Stencil<Cell, F::stencil_radius, TDV>;

    static constexpr int R = int(F::stencil_radius);
    static constexpr int NS = int(F::n_subiterations);
    static constexpr int S = T * NS;            // pipeline levels
    static constexpr int G = R * S;             // halo depth in cells (rows and columns)
    // Columns a generation's dependency cone grows by per side: R per sub-iteration -- unless the function says less
    // (`static constexpr std::size_t halo_columns_per_generation`, an extension hint like constant_fields).  FDTD's first
    // sub-step reads its west and north neighbours only, its second the east and south ones: over a generation the cone
    // grows by ONE cell per side, not two, so a strip of 64 one-cell lanes produces 48 columns at eight generations per
    // launch instead of 32 -- a third fewer bytes and instructions per useful cell: 435 -> 556 Gcell-updates/s, bit for
    // bit the same results (profiles/r04_micro_variants.txt).  A wrong hint gives wrong results; the parity tests of the
    // functions that carry one run at full size.  (Rows keep the generic depth: the row pipeline's lag is structural.)
    static constexpr int GC = [] {
        if constexpr (requires { F::halo_columns_per_generation; })
            return T * int(F::halo_columns_per_generation);
        else
            return G;
    }();
    static constexpr int GX = round_up(GC, K);  // column halo rounded to whole lanes
    static constexpr int LW = wave_size * K;    // columns a wave loads
    static constexpr int OW = LW - 2 * GX;      // columns a unit of the wave grid produces
    static constexpr int NWIN = 2 * R;          // rows each level keeps
    static constexpr int D = 2 * R + 1;
    static constexpr int prefetch_depth = P;

    // Staged sweep: the W waves of a workgroup share one column strip.  Stage s keeps the windows of its own L levels
    // only, takes its input rows from stage s-1 and passes its output rows to stage s+1 through LDS, in batches of P
    // rows: interface i (between stages i and i+1) is a ring of two batches, one workgroup barrier per batch.  Stage
    // s works on batch b during super-step b + s, so a batch written in one super-step is read in the next.
    static constexpr int W = stages_dividing(S, STAGES);
    static constexpr int L = S / W;             // levels per stage
    static constexpr int block_waves = W > 1 ? W : independent_waves_per_block;
    static constexpr int units_per_block = W > 1 ? 1 : independent_waves_per_block;
    // what the launch heuristics count waves with: columns produced per wave
    static constexpr int OW_PER_WAVE = W > 1 ? (OW / W > 0 ? OW / W : 1) : OW;
    static constexpr int CW = int((sizeof(Cell) + 3) / 4);  // 32-bit words per cell in LDS
    static constexpr int LANE_WORDS = K * CW;               // one lane's cells of one row
    static constexpr int ROW_WORDS = wave_size * LANE_WORDS;
    static constexpr int IFACE_WORDS = 2 * P * ROW_WORDS;   // two batches of P rows
    static constexpr int LDS_WORDS = W > 1 ? (W - 1) * IFACE_WORDS : 1;
    static constexpr int lane_align = (LANE_WORDS % 4 == 0) ? 16 : ((LANE_WORDS % 2 == 0) ? 8 : 4);

    static_assert(K >= R, "a lane must hold at least `radius` cells so neighbours are one lane away");
    static_assert(OW >= K, "halo consumes the whole strip: lower max_generations or raise K");
    static_assert(LDS_WORDS * 4 <= 64 * 1024, "the row rings of a staged sweep exceed the LDS budget of a workgroup");
    static_assert(P % NWIN == 0, "prefetch depth must be a multiple of the window length");
    static_assert(std::is_trivially_copyable_v<F> && std::is_trivially_copyable_v<Cell> &&
                  std::is_trivially_copyable_v<TDV>);

    struct Args {
        // Time-dependent values of the launch's T generations, one of three sources (tdv/SinglePassStrategies.hpp):
        // `tdv_table` != nullptr: device array, element i = generation i of this launch (the pass driver's per-call
        // table: precomputed on the host or on the device); else `tdv`: evaluated on the host for this launch
        // and shipped as kernel arguments (FIRST member: the kernel reads them at offset 0 of its argument
        // segment); INLINE_TDV: evaluated by the kernel itself, neither is read.
        TDV tdv[T];
        TDV const *tdv_table;
        F f;
        Cell halo;
        Planes src, dst;
        SweepGeometry geo;
    };

    struct alignas(lane_align) LanePacket {
        std::uint32_t w[LANE_WORDS];
    };
    // The rings are addressed as LDS (address space 3) in the source, not through generic pointers: besides sparing
    // the address-space inference, it keeps the optimiser from merging a store into a ring with a store into a
    // register window at the end of two branches ("sink common code": one store through a phi of the two pointers),
    // which made the window escape and live in scratch (HotSpot, two levels per stage: 80 bytes per lane).
    using LdsWord = __attribute__((address_space(3))) std::uint32_t;
    // a lane's packet moves in the widest vectors its size allows (16, 8 or 4 bytes: one ds_write_b128 / ds_read_b128 per
    // lane and row for a 16-byte lane); builtin vector types, because a class type cannot be assigned across address spaces
    static constexpr int CHUNK_WORDS = lane_align / 4;
    typedef std::uint32_t LdsChunk __attribute__((ext_vector_type(CHUNK_WORDS)));
    using LdsChunkPtr = __attribute__((address_space(3))) LdsChunk *;
    STST_DEVICE static void lds_store_row(LdsWord *lds, int iface, int slot, int lane, Cell const (&cells)[K]) {
        LanePacket packet = {};
#pragma unroll
        for (int k = 0; k < K; k++)
            __builtin_memcpy(&packet.w[k * CW], &cells[k], sizeof(Cell));
        LdsWord *at = lds + iface * IFACE_WORDS + slot * ROW_WORDS + lane * LANE_WORDS;
#pragma unroll
        for (int c = 0; c < LANE_WORDS / CHUNK_WORDS; c++) {
            LdsChunk chunk;
            __builtin_memcpy(&chunk, &packet.w[c * CHUNK_WORDS], sizeof chunk);
            *(LdsChunkPtr)(at + c * CHUNK_WORDS) = chunk;
        }
    }
    STST_DEVICE static void lds_load_row(const LdsWord *lds, int iface, int slot, int lane, Cell (&cells)[K]) {
        LanePacket packet;
        const LdsWord *at = lds + iface * IFACE_WORDS + slot * ROW_WORDS + lane * LANE_WORDS;
#pragma unroll
        for (int c = 0; c < LANE_WORDS / CHUNK_WORDS; c++) {
            const LdsChunk chunk = *(const __attribute__((address_space(3))) LdsChunk *)(at + c * CHUNK_WORDS);
            __builtin_memcpy(&packet.w[c * CHUNK_WORDS], &chunk, sizeof chunk);
        }
#pragma unroll
        for (int k = 0; k < K; k++)
            __builtin_memcpy(&cells[k], &packet.w[k * CW], sizeof(Cell));
    }

    // One wave = stage SG of the unit (strip, rows [ya, yb)).
    // SKIP_CONSTANTS: the target planes of F::constant_fields already hold their values (see
    // constant_plane_mask); their stores are left out
    // (always inlined: as a called function a stage would receive the kernel's arguments through a private copy in
    // scratch memory -- seen for large transition functions at depths of 4 and more)
    template <bool EDGE, bool SKIP_CONSTANTS, int SG>
    STST_DEVICE __attribute__((always_inline)) static void run(Args const &a, const int lane, const int strip, const int ya, const int yb,
                                LdsWord *lds) {
        constexpr std::uint32_t skip_mask = SKIP_CONSTANTS ? constant_plane_mask<F>() : 0u;
        constexpr int L0 = SG * L; // levels below this stage
        // (read where the kernel's arguments lie: indexing the tier tables at run time must not cost a private copy)
        SweepGeometry const &g = *(SweepGeometry const *)(const SweepGeometry __attribute__((address_space(4))) *)(
            (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(Args, geo));
        const int x0 = g.out_col_begin + strip * OW - GX + lane * K; // global column of the lane's first cell
        const int ystart = ya - G;
        const int y_load_end = yb + G < g.load_hi ? yb + G : g.load_hi;

        bool col_in[K];
#pragma unroll
        for (int k = 0; k < K; k++)
            col_in[k] = unsigned(x0 + k - g.col_lo) < unsigned(g.col_hi - g.col_lo);
        // (columns of the grid the buffers do not hold -- beyond the ghost columns of a block -- are treated like
        // columns outside the grid: never loaded; they cannot reach a column this launch stores)
        const bool vec_in = x0 >= g.col_lo && x0 + K <= g.col_hi;
        const bool lane_stores = lane * K >= GX && lane * K + K <= LW - GX;
        const bool vec_out = vec_in && x0 + K <= g.out_col_end; // the last strip of a column range may be ragged

        // The launch's time-dependent values: the call's device table, or the kernel arguments (Args::tdv sits at
        // offset 0 of the kernel's argument segment).  Both are read through the constant address space -- nothing
        // writes them while the kernel runs --, so the compiler may load a value again wherever it needs it instead
        // of holding T scalar registers over the row loop (a plain global load could not be moved over the loop's
        // stores at all).
        // The transition function is read where the kernel's arguments lie (constant address space), not through the
        // by-value copy: a function that indexes tables of its own at run time (FDTD's render resolver: sixteen
        // ring bounds and coefficient sets) otherwise gets a private copy of itself in scratch memory in every
        // stage's code -- 550 bytes per lane at every depth, which the spill-free depth rule then refuses.
        using ConstantF = const F __attribute__((address_space(4)));
        F const &fn_in_arguments = *(F const *)(ConstantF *)((const char __attribute__((address_space(4))) *)
                                                    __builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(Args, f));
        // ... unless it is small: then it is read ONCE and kept in scalar registers (passed through readfirstlane, so
        // that the compiler cannot turn it back into loads inside the row loop: scalar_function_for)
        constexpr bool SCALAR_F = W > 1 && scalar_function_for<F, SOA>();
        F fn_in_registers = fn_in_arguments;
        if constexpr (SCALAR_F) {
            std::uint32_t words[sizeof(F) / 4];
            __builtin_memcpy(words, &fn_in_arguments, sizeof(F));
#pragma unroll
            for (unsigned i = 0; i < sizeof(F) / 4; i++)
                words[i] = std::uint32_t(__builtin_amdgcn_readfirstlane(int(words[i])));
            __builtin_memcpy(static_cast<void *>(&fn_in_registers), words, sizeof(F));
        }
        F const &fn = SCALAR_F ? fn_in_registers : fn_in_arguments;
        using ConstantTDV = const TDV __attribute__((address_space(4)));
        ConstantTDV *launch_tdv = a.tdv_table ? (ConstantTDV *)(a.tdv_table)
                                              : (ConstantTDV *)(__builtin_amdgcn_kernarg_segment_ptr());

        Cell win[L][NWIN][K]; // level l-1's older rows, rotating
        Cell pre[P][K];       // rows in flight from HBM (stage 0) or fetched from the LDS ring
        static_for<0, L>([&](auto l) __attribute__((always_inline)) {
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
                row_offset(g, yc) + std::size_t(std::int64_t(x0 - g.col_origin));
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

        if constexpr (SG == 0)
            static_for<0, P>([&](auto u) __attribute__((always_inline)) { load_row(ystart + u, pre[u]); });

        // the levels that are not due yet are skipped while the pipeline fills (check-free waves only)
        constexpr bool TRAPEZOID = !EDGE && trapezoid_fill_for<F, SOA>();

        // One input row through this stage's levels.  `step` counts the rows fed to level 1 of the unit; the row this
        // stage receives at `step` is what level L0 emitted for it.  While the pipeline fills (FILLING) level l only
        // has to produce rows from step 2*l*R on -- earlier outputs cannot reach a stored row -- so the deeper levels
        // are skipped by wave-uniform branches: a trapezoid of level-steps instead of a parallelogram.
        auto row_step = [&](auto u, const int it, auto filling, const int ring) __attribute__((always_inline)) {
                constexpr bool FILLING = decltype(filling)::value;
                const int step = it + u;
                const int y = ystart + step; // global row entering level 1 at this step
                Cell cur[K];
#pragma unroll
                for (int k = 0; k < K; k++)
                    cur[k] = pre[u][k];
                if constexpr (SG == 0) {
                    if constexpr (W > 1 && pinned_loads_for<F, SOA>())
                        __builtin_amdgcn_sched_barrier(0);
                    load_row(y + P, pre[u]);
                    if constexpr (W > 1 && pinned_loads_for<F, SOA>())
                        __builtin_amdgcn_sched_barrier(0);
                    if constexpr (EDGE) {
                        const bool row_in = unsigned(y) < unsigned(g.grid_h);
#pragma unroll
                        for (int k = 0; k < K; k++)
                            if (!(row_in && col_in[k]))
                                cur[k] = a.halo;
                    }
                }

                bool live = true; // FILLING: the levels up to here are due at this row
                static_for<0, L>([&](auto lc) __attribute__((always_inline)) {
                    constexpr int level = L0 + lc + 1;      // level being computed, 1 .. S
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
                            return fn.get_time_dependent_value(iteration);
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
#pragma unroll
                        for (int d = 1; d <= R; d++) {
                            ext[rr][R - d] = from_west_lane(row[K - d]);
                            ext[rr][R + K - 1 + d] = from_east_lane(row[d - 1]);
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
                        // -- and positive: a cell that can reach a stored row has 0 < row < height-1 and 0 < column <
                        // width-1 at every level there (the footprint lies inside the grid and a level-l cell that
                        // matters lies l*R cells inside the footprint), so a transition function's own tests for the
                        // first row and column fold away.  (The upper bounds are true as well and fold the tests for the last
                        // row and column -- the unchanged HotSpot example 0.061 -> 0.056 s with all four --, but as
                        // relations between two variables they cost the optimiser dearly: the self-checking function
                        // of the reference's tests compiles in 150 s without them and not within 15 minutes with them.)
                        // Both statements are TRUE wherever they are made: x0 + k >= the footprint's first column,
                        // which is positive in a check-free wave (entry()); the row is positive at every level a
                        // trapezoid fill computes (a level is due from step 2*level*R on, so j >= level*R), and a
                        // kernel without the trapezoid hands never-read warm-up cells row max(j, 1) instead of a
                        // negative one (one scalar instruction per level and row).
                        int row_id = j;
                        if constexpr (!EDGE) {
                            if constexpr (!TRAPEZOID)
                                row_id = j > 1 ? j : 1;
                            __builtin_assume(row_id > 0);
                            __builtin_assume(x0 + k > 0);
                        }
                        __builtin_assume(g.grid_h >= 0);
                        __builtin_assume(g.grid_w >= 0);
                        StencilImpl st(sycl::id<2>(std::size_t(std::int64_t(row_id)),
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
                        if constexpr (requires { fn.template at_level<0, 1>(st); })
                            next[k] = fn.template at_level<level - 1, S>(st);
                        // ... or a form for cells that are not on the rim of the grid: in a wave whose
                        // whole footprint lies inside the grid every cell that can reach the output has
                        // 0 < row < height-1 and 0 < column < width-1 at every level
                        else if constexpr (!EDGE && requires { fn.interior(st); })
                            next[k] = fn.interior(st);
                        else
                            next[k] = fn(st);
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

                if constexpr (SG == W - 1) {
                    const int j = y - G; // row leaving the last level
                    if ((!FILLING || live) && j >= ya && j < yb && lane_stores) {
                        const std::size_t first =
                            row_offset(g, j) + std::size_t(std::int64_t(x0 - g.col_origin));
                        if (!EDGE || vec_out) {
                            if constexpr (W > 1 && pinned_stores_for<F, SOA>())
                                __builtin_amdgcn_sched_barrier(0);
                            a.dst.template store<K, streaming_stores_for<F, SOA>(), skip_mask>(first, cur);
                            if constexpr (W > 1 && pinned_stores_for<F, SOA>())
                                __builtin_amdgcn_sched_barrier(0);
                        } else {
#pragma unroll
                            for (int k = 0; k < K; k++)
                                if (col_in[k] && x0 + k < g.out_col_end)
                                    a.dst.template store_one<skip_mask>(first + k, cur[k]);
                        }
                    }
                } else {
                    // the row this stage emits: input of the next stage, one super-step later
                    if (!FILLING || live) {
                        if constexpr (pinned_stores_for<F, SOA>())
                            __builtin_amdgcn_sched_barrier(0);
                        lds_store_row(lds, SG, ring + u, lane, cur);
                        if constexpr (pinned_stores_for<F, SOA>())
                            __builtin_amdgcn_sched_barrier(0);
                    }
                }
        };

        // Super-step t: stage SG works on batch t - SG (rows [b*P, (b+1)*P) of the unit's feed).  Every wave of the
        // workgroup passes the same number of barriers, whether it has a batch in this super-step or not.
        // batches before this one hold nothing this stage needs: its first level is due from step 2*(L0+1)*R on and
        // reads window rows from 2R steps before that
        constexpr int first_batch = TRAPEZOID ? (2 * L0 * R) / P : 0;
        const int n_rows_in = yb - ya + 2 * G;
        const int n_batches = (n_rows_in + P - 1) / P;
        const int n_super = n_batches + (W - 1);
        auto super_step = [&](const int t, auto filling) __attribute__((always_inline)) {
            if constexpr (W > 1) {
                // rows of the previous super-step become visible to the next stage; LDS only -- the rows in flight
                // from HBM (pre[]) must not be waited for here
                // (the explicit wait is not redundant: the compiler may leave out the wait a release fence implies when it
                // has seen an earlier wait of its own, and a row then reaches the next stage late -- measured: one wrong
                // launch in a few hundred with the fences alone)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            }
            const int b = t - SG;
            if (b < first_batch || b >= n_batches)
                return;
            const int ring = (b & 1) * P;
            if constexpr (SG > 0)
                static_for<0, P>([&](auto u) __attribute__((always_inline)) {
                    lds_load_row(lds, SG - 1, ring + u, lane, pre[u]);
                });
            static_for<0, P>([&](auto u) __attribute__((always_inline)) { row_step(u, b * P, filling, ring); });
        };
        int t = 0;
        if constexpr (TRAPEZOID) {
            // super-steps in which some level of this stage is not due yet
            constexpr int fill_batches = (2 * (L0 + L) * R + P - 1) / P;
            const int t_fill = SG + fill_batches < n_super ? SG + fill_batches : n_super;
            for (; t < t_fill; t++)
                super_step(t, std::true_type{});
        }
        for (; t < n_super; t++)
            super_step(t, std::false_type{});
    }

    // The kernel's argument block where it lies (the kernel-argument segment, constant address space).  The device code
    // reads everything from here and never touches the by-value parameter: where the optimiser could not take that
    // copy apart it lived in private memory (64-80 bytes of scratch per lane in the L = 1 Jacobi kernels and two HotSpot
    // depths, every plane pointer re-loaded from scratch in the edge code: tools/kernel_resources.sh).
    STST_DEVICE __attribute__((always_inline)) static Args const &kernel_arguments() {
        return *(Args const *)(const Args __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    }

    template <bool SKIP_CONSTANTS = false>
    STST_DEVICE __attribute__((always_inline)) static void entry(Args const &a, LdsWord *lds) {
        // (read where the kernel's arguments lie: indexing the tier tables at run time must not cost a private copy)
        SweepGeometry const &g = *(SweepGeometry const *)(const SweepGeometry __attribute__((address_space(4))) *)(
            (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(Args, geo));
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
        // unit of the wave grid: a wave, or (staged) the whole workgroup -- its waves then share strip and chunk,
        // run the same number of super-steps and meet at one barrier per batch of rows
        const unsigned unit = W > 1 ? block
                                    : __builtin_amdgcn_readfirstlane(block * unsigned(units_per_block) + unsigned(wib));
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

        const int xw0 = g.out_col_begin + strip * OW - GX; // footprint of the unit
        // (xw0 > 0, not >= 0: the check-free code tells the transition function that its column is positive; col_lo and
        // col_hi lie inside the grid; the strip's columns must all be columns to produce)
        const bool interior = INTERIOR_VARIANT && xw0 > 0 && xw0 >= g.col_lo && xw0 + LW <= g.col_hi &&
                              xw0 + LW - GX <= g.out_col_end && ya - G >= 0 && yb + G <= g.grid_h;
        // the stage of a wave is a compile-time property of the code it runs (levels, iteration and sub-iteration
        // indices, fused forms): one instantiation per stage, selected by the wave's index in the workgroup
        static_for<0, W>([&](auto sg) __attribute__((always_inline)) {
            if (W > 1 && wib != int(sg))
                return;
            if constexpr (INTERIOR_VARIANT) {
                if (interior)
                    run<false, SKIP_CONSTANTS, int(sg)>(a, lane, strip, ya, yb, lds);
                else
                    run<true, SKIP_CONSTANTS, int(sg)>(a, lane, strip, ya, yb, lds);
            } else {
                run<true, SKIP_CONSTANTS, int(sg)>(a, lane, strip, ya, yb, lds);
            }
        });
    }

    // Element offset of buffer row `y` (a global row the buffers hold): both factors fit 31 bits (checked at launch),
    // so the 64-bit product is a 32 x 32 multiply -- two scalar instructions instead of the five of a 64 x 64 one,
    // once per loaded and per stored row.
    STST_DEVICE static std::size_t row_offset(SweepGeometry const &g, int y) {
        return std::size_t(std::uint32_t(y - g.row_origin)) * std::size_t(g.pitch32);
    }
};

// The sweep SweepTuning<F, SOA> asks for, at blocking depth T.
template <typename F, bool SOA, int T = SweepTuning<F, SOA>::max_generations, bool INLINE_TDV = false>
using SweepOf = Sweep<F, SOA, T, SweepTuning<F, SOA>::cells_per_lane, SweepTuning<F, SOA>::prefetch_rows,
                      SweepTuning<F, SOA>::interior_variant, stages_for<F, SOA>(), INLINE_TDV>;

// MIN_WAVES = waves per SIMD the register allocator must leave room for (launch-bounds semantics).
template <typename SW, int MIN_WAVES = 1, bool SKIP_CONSTANTS = false>
__global__ void __launch_bounds__(SW::block_waves * wave_size, MIN_WAVES) sweep_kernel(const typename SW::Args args) {
    // the row rings between the stages of a staged sweep (one word otherwise)
    __shared__ __attribute__((aligned(16))) std::uint32_t rings[SW::LDS_WORDS];
    SW::template entry<SKIP_CONSTANTS>(SW::kernel_arguments(), (typename SW::LdsWord *)rings); // `args` itself is not touched (see there)
}

// ------------------------------------------------------------------ host side
// Tuning knobs come from the environment ONCE (ststhip_options: read at ststhip_init, again on
// ststhip_reload_options), not per launch.
inline ststhip_options const &options() { return *ststhip_get_options(); }

// Rows of output per unit of the wave grid.  A unit costs (rows + 2*G warm-up rows + prologue); the chip keeps
// `slots` units in flight and back-fills as they retire, so
//   time ~ units * cost / slots  +  alpha * cost      (alpha ~ 0.5: the ragged tail of the last units)
// with units = strips * out_rows / rows.  Minimising over rows gives the closed form below: long
// chunks waste the tail, short chunks waste warm-up rows.  (Measured optimum for Jacobi 16384^2,
// T = 8: ~135 rows; the formula gives 133.)
// `tail_permille_beside` (a kernel's own weight for launches that run side by side, SweepTuning) replaces the general
// 350 only where the general rule already gives the launch well over one residency round of units: with fewer, longer
// chunks leave part of the chip without a unit (uniform Jacobi form, 200 against 350: 16384^2 +1.4 %, 24576^2 +1.2 %,
// but 8192^2 -11 %, 6144^2 -12 %: profiles/r03_stage_experiments.txt).  *kernel_rule says which one was taken.
inline int pick_chunk_rows(int out_rows, unsigned n_strips, int overhead_rows, int resident_blocks,
                           int units_per_block, int tail_permille_beside = 350, bool *kernel_rule = nullptr) {
    ststhip_options const &opt = options();
    if (kernel_rule)
        *kernel_rule = false;
    if (opt.chunk_rows > 0)
        return std::min(opt.chunk_rows, std::max(out_rows, 1));
    int cus = 256;
    ststhip_compute_units(&cus);
    // launches running side by side (row strips of the pass driver) fill each other's tails: the tail
    // weight shrinks by their number (fitted to chunk sweeps of Jacobi 16384^2 and HotSpot 8192^2)
    const int side_by_side = std::max(1, ststhip_launch_concurrency());
    const double slots = double(cus) * std::max(resident_blocks, 1) * units_per_block;
    // tail weight: 0.5 for a launch that has the chip to itself; launches that run side by side (their boundary
    // bands on streams of their own) want slightly longer chunks still (profiles/r02_ab_bands_beside.txt)
    double alpha = (opt.tail_permille > 0 ? opt.tail_permille : (side_by_side > 1 ? 350 : 500)) / 1000.0 / side_by_side;
    const double overhead = double(overhead_rows) + 8.0;
    double rows = std::sqrt(double(out_rows) * double(n_strips) * overhead / (alpha * slots));
    rows = std::max(rows, 1.0);
    long chunks = std::max<long>(1, long(double(out_rows) / rows + 0.5));
    if (opt.tail_permille <= 0 && side_by_side > 1 && tail_permille_beside != 350 &&
        double(chunks) * n_strips / slots >= 1.2) {
        alpha = tail_permille_beside / 1000.0 / side_by_side;
        rows = std::max(std::sqrt(double(out_rows) * double(n_strips) * overhead / (alpha * slots)), 1.0);
        chunks = std::max<long>(1, long(double(out_rows) / rows + 0.5));
        if (kernel_rule)
            *kernel_rule = true;
    }
    // A launch beside another one that would fill less than two thirds of a residency round: the model above (units queue
    // for slots) does not hold -- every unit runs at once, the launch takes as long as one unit, so shorter chunks are
    // faster until the units fill the slots.  Measured (profiles/r04_thin_strips.txt, r04_skewed_strips.txt): a rank's
    // 2048-row strip as two sub-strips 4264 -> 4412 (116 rows by the model, 64 by this), the pass driver's two strips at
    // 6144^2 4664 -> 4871, general coefficients at 4096^2 2086 -> 2221; from 0.72 of a round on (8192^2) it loses 2 %.
    if (side_by_side > 1 && double(chunks) * n_strips < 0.65 * slots) {
        const long filled = long(0.93 * slots / double(std::max(1u, n_strips)));
        const long least_rows = std::max<long>(16, long(overhead) * 3 / 2); // (not below one and a half warm-ups of useful rows)
        chunks = std::max<long>(chunks, std::min<long>(filled, std::max<long>(1, long(out_rows) / least_rows)));
    }
    // snap to a whole number of residency rounds (just below it) when that is a small change:
    // a launch of k*S + a few units pays for a nearly empty extra round
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
// cut finer: the taper "permille:split[,permille:split...]" makes the last `permille` of the rows
// chunks of chunk_rows/split rows (later entries refine the end further; they must shrink).
// Default (profiles/r01_tune_taper.txt): the last 12 % of the rows in quarter-length chunks when the
// launch has the chip to itself (+7 % for a full-grid Jacobi launch, +2..4 % HotSpot / FDTD); launches
// that run side by side already fill each other's ends and lose 1-6 % with shorter chunks.
inline void plan_tiers(SweepGeometry &g, int out_rows, bool taper_beside = true) {
    ststhip_options const &opt = options();
    g.n_tiers = 1;
    g.tier_first[0] = 0;
    g.tier_rows[0] = g.chunk_rows;
    g.tier_begin[0] = g.out_begin;
    int n_entries = opt.n_taper;
    int permilles[3] = {opt.taper_permille[0], opt.taper_permille[1], opt.taper_permille[2]};
    int splits[3] = {opt.taper_split[0], opt.taper_split[1], opt.taper_split[2]};
    if (n_entries < 0) { // not set in the environment
        n_entries = (ststhip_launch_concurrency() == 1 || taper_beside) ? 1 : 0;
        permilles[0] = ststhip_launch_concurrency() == 1 ? 120 : 150;
        splits[0] = ststhip_launch_concurrency() == 1 ? 4 : 2;
    }
    int done_rows = 0; // rows covered by the tiers closed so far
    unsigned done_chunks = 0;
    int previous_start = 0;
    for (int e = 0; e < n_entries && g.n_tiers < unsigned(SweepGeometry::max_tiers); e++) {
        const int permille = permilles[e], split = splits[e];
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
    if (out_end <= out_begin || dom.global_width == 0)
        return;
    ststhip_options const &opt = options();
    // a driver may leave a hole in the row range (ststhip_launch_row_hole): the two boundary bands of a row strip as
    // ONE launch -- rows [out_begin, hole) and [hole end, out_end), the interior in between is another launch's
    std::uint64_t hole_begin = 0, hole_end = 0;
    ststhip_launch_row_hole(&hole_begin, &hole_end);
    if constexpr (has_narrow_form<F, SOA>()) {
        // grids that cannot fill the chip with this shape: the same function on the narrowest lanes.  So are launches
        // of a few rows when asked for (the boundary bands of row strips)
        const std::uint64_t launch_rows = (out_end - out_begin) - (hole_end - hole_begin);
        if (dom.global_height * dom.global_width <= std::uint64_t(opt.narrow_form_kcells) * 1000 ||
            launch_rows <= std::uint64_t(opt.narrow_band_rows)) {
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
    // columns: whole rows unless the buffers are a block of a 2-D decomposition (ststhip_domain::local_cols) and / or
    // the driver has named a column range for this thread's launches (ststhip_set_launch_columns)
    const std::int64_t local_cols = dom.local_cols ? std::int64_t(dom.local_cols) : std::int64_t(dom.global_width);
    g.col_origin = std::int32_t(dom.local_cols ? dom.col_origin : 0);
    g.col_lo = std::int32_t(std::max<std::int64_t>(0, g.col_origin));
    g.col_hi = std::int32_t(std::min<std::int64_t>(std::int64_t(dom.global_width), std::int64_t(g.col_origin) + local_cols));
    std::uint64_t col_begin = 0, col_end = 0;
    ststhip_launch_columns(&col_begin, &col_end);
    if (col_begin == col_end) {
        col_begin = std::uint64_t(g.col_lo);
        col_end = std::uint64_t(g.col_hi);
    }
    if (col_end <= col_begin || std::int64_t(col_begin) < g.col_lo || std::int64_t(col_end) > g.col_hi)
        throw std::invalid_argument("the launch's column range must lie inside the columns the buffers hold");
    g.out_col_begin = std::int32_t(col_begin);
    g.out_col_end = std::int32_t(col_end);
    const void *kernel = reinterpret_cast<const void *>(&sweep_kernel<SW, Tuning::min_waves_per_simd>);
    // per kernel instantiation; a property of the code object and the architecture, so racing host
    // threads would store the same number
    static std::atomic<int> resident_blocks_cache{0};
    int resident_blocks = resident_blocks_cache.load(std::memory_order_relaxed);
    if (resident_blocks == 0) {
        check(ststhip_occupancy(kernel, SW::block_waves * wave_size, 0, &resident_blocks), "occupancy query");
        resident_blocks_cache.store(resident_blocks, std::memory_order_relaxed);
    }
    // per-field planes of fields F only copies: from the third pass of a run on the target holds them already
    if constexpr (SOA && constant_plane_mask<F>() != 0)
        if (ststhip_target_holds_constants() && opt.skip_constant_stores)
            kernel = reinterpret_cast<const void *>(&sweep_kernel<SW, Tuning::min_waves_per_simd, true>);
    g.n_strips = unsigned((col_end - col_begin + SW::OW - 1) / SW::OW); // units: waves, or workgroups (staged)
    // rows a unit spends besides its output: 2G warm-up rows, and the skew of a staged pipeline (stage s starts s
    // batches late)
    const int overhead_rows = 2 * SW::G + (SW::W - 1) * SW::prefetch_depth;
    if (hole_begin < hole_end) {
        if (hole_begin <= out_begin || hole_end >= out_end)
            throw std::invalid_argument("the row hole must lie strictly inside the launch's row range");
        const int part[2] = {int(hole_begin - out_begin), int(out_end - hole_end)};
        const std::uint64_t part_begin[2] = {out_begin, hole_end};
        const int wanted = pick_chunk_rows(std::max(part[0], part[1]), g.n_strips, overhead_rows, resident_blocks,
                                           SW::units_per_block);
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
        bool kernel_rule = false;
        g.chunk_rows = pick_chunk_rows(int(out_end - out_begin), g.n_strips, overhead_rows, resident_blocks,
                                       SW::units_per_block, tail_permille_beside_for<F, SOA>(), &kernel_rule);
        g.n_chunks = unsigned((out_end - out_begin + g.chunk_rows - 1) / g.chunk_rows);
        plan_tiers(g, int(out_end - out_begin), !kernel_rule || taper_beside_for<F, SOA>());
    }
    if (dom.pitch >= (1ull << 31))
        throw std::range_error("the pitch must be below 2^31 elements");
    g.pitch32 = std::uint32_t(dom.pitch);
    g.iteration = iteration;
    // measured (profiles/r01_xcd_remap.txt): 4 % fewer HBM reads, but no gain in time for the independent-wave
    // kernels; with the staged kernels 0.50 -> 0.70 ms per launch (the taper's short chunks no longer come last), and
    // column bands per XCD in chunk-major order -1 % (profiles/r03_multi_pass.txt): off unless asked for
    g.xcd_remap = opt.xcd_remap ? 1u : 0u;
    g.last_chunk_early = (out_end + SW::G > dom.global_height && opt.last_chunk_early) ? 1u : 0u;

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
    // transition functions need not be default-constructible: build the argument block in one go
    typename SW::Args args = [&]<std::size_t... Is>(std::index_sequence<Is...>) {
        return typename SW::Args{{((table || !tdv) ? TDV{} : tdv[Is])...}, table, f, halo, src, dst, g};
    }(std::make_index_sequence<std::size_t(T)>{});

    const unsigned units = g.n_strips * g.n_chunks;
    const unsigned blocks = (units + SW::units_per_block - 1) / SW::units_per_block;
    void *kernel_args[] = {&args};
    check(ststhip_launch(kernel, blocks, 1, 1, SW::block_waves * wave_size, 1, 1, kernel_args, 0, stream), "sweep launch");
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
    int t = default_generations_for<F, SOA>();
    const int cap = options().max_generations > 0 ? options().max_generations : t;
    while (t > 1 && (std::uint64_t(t) > remaining || t > cap))
        t /= 2;
    return t;
}

} // namespace internal
} // namespace hip
} // namespace stencil
