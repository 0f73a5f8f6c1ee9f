// API tests of the MI355X backend through the C++ templates (hipcc, needs a GPU to run).
// Also compiled: an *unannotated* user functor (no __device__), which is how applications written
// for the reference look; it is device-callable because this file is built with --hipstdpar.
#include "api_tests.hpp"
#include <StencilStream/BaseTransitionFunction.hpp>
#include <StencilStream/cpu/StencilUpdate.hpp>
#include <StencilStream/cuda/StencilUpdate.hpp>
#include <StencilStream/cuda/internal/Helpers.hpp>
#include <apps/conway.hpp>
#include <cstdio>
#include <cstring>

using namespace stencil;

// pinned: the tests below want the per-field planes exercised through the templates whatever the policy says
namespace stencil {
namespace hip {
template <> struct SplitCellPolicy<apps::SelfCheck<1>> {
    static constexpr bool sweep_on_planes = true;
};
} // namespace hip
} // namespace stencil

static_assert(concepts::Grid<hip::Grid<bool>, bool>);
static_assert(concepts::StencilUpdate<cuda::StencilUpdate<apps::Conway>, apps::Conway, cuda::Grid<bool>>);
static_assert(std::is_same_v<cuda::Grid<float>, hip::Grid<float>>);

// exactly what user code looks like: plain C++, no HIP annotations
struct UserHeat : public BaseTransitionFunction {
    using Cell = float;
    float alpha;
    float operator()(Stencil<float, 1> const &s) const {
        return s[0][0] + alpha * (s[-1][0] + s[1][0] + s[0][-1] + s[0][1] - 4.0f * s[0][0]);
    }
};

static void test_user_functor() {
    const std::size_t h = 123, w = 517, n = 21;
    using SU = cuda::StencilUpdate<UserHeat>;
    SU::GridImpl grid(h, w);
    std::vector<float> host(h * w);
    {
        SU::GridImpl::GridAccessor<sycl::access::mode::read_write> ac(grid);
        for (std::size_t r = 0; r < h; r++)
            for (std::size_t c = 0; c < w; c++)
                host[r * w + c] = ac[r][c] = float((r * 31 + c * 17) % 101) / 101.0f;
    }
    SU update({.transition_function = UserHeat{.alpha = 0.1f},
               .halo_value = 0.5f,
               .n_iterations = n,
               .blocking = true,
               .profiling = true});
    SU::GridImpl out = update(grid);
    // host restatement of the same sweep
    std::vector<float> next(h * w);
    auto at = [&](std::vector<float> const &g, long r, long c) {
        return (r < 0 || c < 0 || r >= long(h) || c >= long(w)) ? 0.5f : g[r * w + c];
    };
    for (std::size_t it = 0; it < n; it++) {
        for (long r = 0; r < long(h); r++)
            for (long c = 0; c < long(w); c++)
                next[r * w + c] = at(host, r, c) + 0.1f * (at(host, r - 1, c) + at(host, r + 1, c) +
                                                          at(host, r, c - 1) + at(host, r, c + 1) -
                                                          4.0f * at(host, r, c));
        host.swap(next);
    }
    SU::GridImpl::GridAccessor<sycl::access::mode::read> ac(out);
    bool same = true;
    for (std::size_t r = 0; r < h; r++)
        for (std::size_t c = 0; c < w; c++)
            same = same && ac[r][c] == host[r * w + c];
    REQUIRE(same);
    REQUIRE(update.get_kernel_runtime() > 0.0 && update.get_kernel_runtime() <= update.get_walltime());
}

static void test_zero_iterations_alias() {
    hip::Grid<bool> g(4, 4);
    hip::StencilUpdate<apps::Conway> update({.transition_function = apps::Conway(), .n_iterations = 0});
    hip::Grid<bool> out = update(g);
    {
        hip::Grid<bool>::GridAccessor<sycl::access::mode::read_write> ac(out);
        ac[1][1] = true;
    }
    hip::Grid<bool>::GridAccessor<sycl::access::mode::read> in(g);
    REQUIRE(in[1][1] == true);
}

// split_cell_structure = true is a request for per-field planes; the backend honours it except for cells of 16
// or 32 bytes, which it sweeps as AoS with identical results (hip::SplitCellPolicy)
struct FatCell {
    float a, b, c, d;
    static constexpr auto fields = std::make_tuple(&FatCell::a, &FatCell::b, &FatCell::c, &FatCell::d);
};
struct FatShift : public BaseTransitionFunction {
    using Cell = FatCell;
    FatCell operator()(Stencil<FatCell, 1> const &s) const {
        return FatCell{s[0][-1].a, s[-1][0].b + 1.0f, s[0][0].c * 0.5f, s[0][0].d + s[0][1].a};
    }
};
static_assert(!hip::SplitCellPolicy<FatShift>::sweep_on_planes);
static_assert(hip::SplitCellPolicy<apps::SelfCheck<2>>::sweep_on_planes); // 20 bytes: planes, as asked

static void test_split_request_on_fat_cells() {
    const std::size_t h = 70, w = 131;
    hip::Grid<FatCell> grid(h, w);
    {
        hip::Grid<FatCell>::GridAccessor<sycl::access::mode::read_write> ac(grid);
        for (std::size_t r = 0; r < h; r++)
            for (std::size_t c = 0; c < w; c++)
                ac[r][c] = FatCell{float(r), float(c), float(r + c), 1.0f};
    }
    auto run = [&](auto &update) { return update(grid); };
    hip::StencilUpdate<FatShift, true> split({.transition_function = FatShift(),
                                              .halo_value = FatCell{-1.0f, -2.0f, -3.0f, -4.0f},
                                              .n_iterations = 11,
                                              .blocking = true});
    hip::StencilUpdate<FatShift, false> plain({.transition_function = FatShift(),
                                               .halo_value = FatCell{-1.0f, -2.0f, -3.0f, -4.0f},
                                               .n_iterations = 11,
                                               .blocking = true});
    hip::Grid<FatCell> a = run(split), b = run(plain);
    bool same = true;
    {
        hip::Grid<FatCell>::GridAccessor<sycl::access::mode::read> x(a), y(b);
        for (std::size_t r = 0; r < h; r++)
            for (std::size_t c = 0; c < w; c++)
                same = same && std::memcmp(&x[r][c], &y[r][c], sizeof(FatCell)) == 0;
    }
    REQUIRE(same);
    // zero iterations: the split path returns a fresh grid (cuda/StencilUpdate.hpp:285,440), never an alias
    split.get_params().n_iterations = 0;
    hip::Grid<FatCell> copy = split(grid);
    {
        hip::Grid<FatCell>::GridAccessor<sycl::access::mode::read_write> ac(copy);
        REQUIRE(ac[3][4].a == 3.0f && ac[3][4].b == 4.0f);
        ac[3][4].a = 99.0f;
    }
    hip::Grid<FatCell>::GridAccessor<sycl::access::mode::read> in(grid);
    REQUIRE(in[3][4].a == 3.0f);
}

// A user function that declares the field it only copies: on per-field planes the stores of that plane are
// left out from the third pass of a call on (hip/internal/Sweep.hpp, constant_plane_mask)
struct Conductor {
    float temperature, conductivity;
    static constexpr auto fields = std::make_tuple(&Conductor::temperature, &Conductor::conductivity);
};
struct Conduction : public BaseTransitionFunction {
    using Cell = Conductor;
    static constexpr auto constant_fields = std::make_tuple(&Conductor::conductivity);
    Conductor operator()(Stencil<Conductor, 1> const &s) const {
        const float k = s[0][0].conductivity;
        const float t = s[0][0].temperature;
        return Conductor{t + k * (s[-1][0].temperature + s[1][0].temperature + s[0][-1].temperature +
                                  s[0][1].temperature - 4.0f * t),
                         k};
    }
};
static_assert(hip::internal::constant_plane_mask<Conduction>() == 2u);
static_assert(hip::internal::constant_plane_mask<UserHeat>() == 0u);
static_assert(hip::SplitCellPolicy<Conduction>::sweep_on_planes);

static void test_constant_fields_hint() {
    const std::size_t h = 300, w = 217;
    hip::Grid<Conductor> grid(h, w);
    {
        hip::Grid<Conductor>::GridAccessor<sycl::access::mode::read_write> ac(grid);
        for (std::size_t r = 0; r < h; r++)
            for (std::size_t c = 0; c < w; c++)
                ac[r][c] = Conductor{float((r * 7 + c * 13) % 50), 0.05f + 0.001f * float((r + 3 * c) % 40)};
    }
    for (std::size_t n : {std::size_t(8), std::size_t(17), std::size_t(45)}) { // 1, 3 and 6 passes
        hip::StencilUpdate<Conduction, true> planes({.transition_function = Conduction(),
                                                     .halo_value = Conductor{20.0f, 0.0f},
                                                     .n_iterations = n,
                                                     .blocking = true});
        hip::StencilUpdate<Conduction, false> cells({.transition_function = Conduction(),
                                                     .halo_value = Conductor{20.0f, 0.0f},
                                                     .n_iterations = n,
                                                     .blocking = true});
        hip::Grid<Conductor> a = planes(grid), b = cells(grid);
        hip::Grid<Conductor>::GridAccessor<sycl::access::mode::read> x(a), y(b), in(grid);
        bool same = true, kept = true;
        for (std::size_t r = 0; r < h; r++)
            for (std::size_t c = 0; c < w; c++) {
                same = same && std::memcmp(&x[r][c], &y[r][c], sizeof(Conductor)) == 0;
                kept = kept && x[r][c].conductivity == in[r][c].conductivity;
            }
        REQUIRE(same);
        REQUIRE(kept);
    }
}

static void test_field_buffers() {
    // cuda/internal/Helpers.hpp:37-67: one typed plane per field, zipped iteration
    auto buffers = cuda::internal::alloc_field_buffers<apps::SelfCheckCell>(1000);
    REQUIRE(buffers.size() == 1000);
    int visited = 0;
    cuda::internal::for_each_in_two_tuples(buffers.pointers(), apps::SelfCheckCell::fields,
                                           [&](auto *plane, auto member) {
                                               REQUIRE(plane != nullptr);
                                               apps::SelfCheckCell probe{};
                                               static_assert(sizeof(*plane) == sizeof(probe.*member));
                                               visited++;
                                           });
    REQUIRE(visited == 5);
    auto set = buffers.plane_set();
    REQUIRE(set.plane[0] == std::get<0>(buffers.pointers()) && set.plane[4] == std::get<4>(buffers.pointers()));
}

// A function with a time-dependent value and two sub-iterations, exact in fp32 whatever evaluates it (host libm,
// device, inline): the three tdv::single_pass strategies must give the cpu backend's result bit for bit.
struct Ramp {
    using Cell = float;
    using TimeDependentValue = float;
    static constexpr std::size_t stencil_radius = 1;
    static constexpr std::size_t n_subiterations = 2;
    float gain;
    float get_time_dependent_value(std::size_t i) const { return float(i % 7) * gain; }
    float operator()(Stencil<float, 1, float> const &s) const {
        if (s.subiteration == 0)
            return 0.5f * (s[0][-1] + s[0][1]) + s.time_dependent_value;
        return 0.5f * (s[-1][0] + s[1][0]) - 0.25f * s.time_dependent_value + float(s.iteration % 3);
    }
};

template <typename Strategy> static void test_tdv_strategy(const char *name) {
    const std::size_t h = 200, w = 333, offset = 5;
    using SU = hip::StencilUpdate<Ramp, false, Strategy>;
    using Reference = cpu::StencilUpdate<Ramp>;
    static_assert(tdv::single_pass::Strategy<Strategy, Ramp, 8>);
    hip::Grid<float> grid(h, w);
    cpu::Grid<float> host_grid(h, w);
    {
        hip::Grid<float>::GridAccessor<sycl::access::mode::read_write> ac(grid);
        cpu::Grid<float>::GridAccessor<sycl::access::mode::read_write> hc(host_grid);
        for (std::size_t r = 0; r < h; r++)
            for (std::size_t c = 0; c < w; c++)
                hc[r][c] = ac[r][c] = float((r * 13 + c * 7) % 64) * 0.125f;
    }
    // 29 generations = launches of every compiled depth; a second call resumes at the new offset
    SU update({.transition_function = Ramp{0.5f}, .halo_value = 1.0f, .iteration_offset = offset, .n_iterations = 29,
               .blocking = true});
    Reference reference({.transition_function = Ramp{0.5f}, .halo_value = 1.0f, .iteration_offset = offset,
                         .n_iterations = 29, .blocking = true});
    hip::Grid<float> out = update(grid);
    cpu::Grid<float> want = reference(host_grid);
    update.get_params().iteration_offset = reference.get_params().iteration_offset = offset + 29;
    update.get_params().n_iterations = reference.get_params().n_iterations = 4;
    out = update(out);
    want = reference(want);
    hip::Grid<float>::GridAccessor<sycl::access::mode::read> ac(out);
    cpu::Grid<float>::GridAccessor<sycl::access::mode::read> wc(want);
    bool same = true;
    for (std::size_t r = 0; r < h; r++)
        for (std::size_t c = 0; c < w; c++)
            same = same && std::memcmp(&ac[r][c], &wc[r][c], sizeof(float)) == 0;
    if (!same)
        std::fprintf(stderr, "strategy %s differs from the cpu backend\n", name);
    REQUIRE(same);
}

// A transition function that carries a lot of state of its own (the shape of FDTD's RenderResolver: a ladder of bounds,
// each with a set of coefficients, compared for every cell in both sub-iterations).  SweepTuning's generic rule would
// sweep its 16-byte cell eight generations deep; the compiler can only build that kernel with spills, so the update
// leaves that depth out (StencilUpdate::spill_free_depth) -- whatever depth runs, the result is the cpu backend's.
struct LadderCell {
    float a, b, c, d;
};
struct Ladder {
    using Cell = LadderCell;
    using TimeDependentValue = std::monostate;
    static constexpr std::size_t stencil_radius = 1;
    static constexpr std::size_t n_subiterations = 2;
    static constexpr int n_steps = 20;
    float bound[n_steps];
    float gain[n_steps][4];
    std::monostate get_time_dependent_value(std::size_t) const { return {}; }
    LadderCell operator()(Stencil<LadderCell, 1> const &s) const {
        LadderCell me = s[0][0];
        const float dr = float(s.id[0]) - float(s.grid_range[0] / 2), dc = float(s.id[1]) - float(s.grid_range[1] / 2);
        const float score = dr * dr + dc * dc;
        int step = n_steps - 1;
        for (int i = n_steps - 1; i >= 0; i--)
            if (score <= bound[i])
                step = i;
        if (s.subiteration == 0) {
            me.a = gain[step][0] * me.a + gain[step][1] * (s[0][-1].c - s[0][0].c);
            me.b = gain[step][0] * me.b + gain[step][1] * (s[-1][0].c - s[0][0].c);
        } else {
            me.c = gain[step][2] * me.c + gain[step][3] * ((s[0][1].a - s[0][0].a) + (s[1][0].b - s[0][0].b));
            me.d += me.c * me.c;
        }
        return me;
    }
};

static void test_state_heavy_functor() {
    const std::size_t h = 190, w = 275;
    Ladder f;
    for (int i = 0; i < Ladder::n_steps; i++) {
        f.bound[i] = float((i + 1) * (i + 1) * 40);
        for (int k = 0; k < 4; k++)
            f.gain[i][k] = 0.25f + 0.03125f * float((i * 4 + k) % 13);
    }
    hip::Grid<LadderCell> grid(h, w);
    cpu::Grid<LadderCell> host_grid(h, w);
    {
        hip::Grid<LadderCell>::GridAccessor<sycl::access::mode::read_write> ac(grid);
        cpu::Grid<LadderCell>::GridAccessor<sycl::access::mode::read_write> hc(host_grid);
        for (std::size_t r = 0; r < h; r++)
            for (std::size_t c = 0; c < w; c++)
                hc[r][c] = ac[r][c] = LadderCell{float((r * 7 + c) % 32) * 0.03125f, float((r + c * 5) % 16) * 0.0625f,
                                                 float((r * 3 + c * 11) % 64) * 0.015625f, 0.0f};
    }
    hip::StencilUpdate<Ladder> update({.transition_function = f, .halo_value = LadderCell{0.5f, 0.25f, 0.125f, 0.0f},
                                       .n_iterations = 23, .blocking = true});
    cpu::StencilUpdate<Ladder> reference({.transition_function = f, .halo_value = LadderCell{0.5f, 0.25f, 0.125f, 0.0f},
                                          .n_iterations = 23, .blocking = true});
    hip::Grid<LadderCell> out = update(grid);
    cpu::Grid<LadderCell> want = reference(host_grid);
    hip::Grid<LadderCell>::GridAccessor<sycl::access::mode::read> ac(out);
    cpu::Grid<LadderCell>::GridAccessor<sycl::access::mode::read> wc(want);
    bool same = true;
    for (std::size_t r = 0; r < h; r++)
        for (std::size_t c = 0; c < w; c++)
            same = same && std::memcmp(&ac[r][c], &wc[r][c], sizeof(LadderCell)) == 0;
    REQUIRE(same);
}

static void test_tdv_strategies() {
    test_tdv_strategy<tdv::single_pass::PrecomputeOnHostStrategy>("PrecomputeOnHost");
    test_tdv_strategy<tdv::single_pass::PrecomputeOnDeviceStrategy>("PrecomputeOnDevice");
    test_tdv_strategy<tdv::single_pass::InlineStrategy>("Inline");
}

int main() {
    test_tdv_strategies();
    test_field_buffers();
    api_tests::test_stencil_indexing();
    api_tests::test_grid<hip::Grid<sycl::id<2>>>(128, 128);
    api_tests::test_grid<hip::Grid<sycl::id<2>>>(3, 17);
    // AoS and split cell structure, as tests/cuda/StencilUpdate.cpp:30-50
    api_tests::test_stencil_update_cases<hip::Grid<apps::SelfCheckCell>,
                                         hip::StencilUpdate<apps::SelfCheck<1>, false>>();
    api_tests::test_stencil_update_cases<hip::Grid<apps::SelfCheckCell>,
                                         hip::StencilUpdate<apps::SelfCheck<1>, true>>();
    test_user_functor();
    test_zero_iterations_alias();
    test_split_request_on_fat_cells();
    test_constant_fields_hint();
    test_state_heavy_functor();
    return finish("hip_api_test");
}
