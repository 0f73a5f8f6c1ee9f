// Host-only API tests (g++): value types, concepts, the explicitly selected cpu backend.
#include "api_tests.hpp"
#include <StencilStream/BaseTransitionFunction.hpp>
#include <StencilStream/cpu/StencilUpdate.hpp>
#include <StencilStream/tdv/SinglePassStrategies.hpp>
#include <algorithm>
#include <apps/conway.hpp>
#include <nlohmann/json.hpp>
#include <sstream>
#include <sycl/ext/intel/ac_types/ac_int.hpp>

using namespace stencil;

static_assert(concepts::TransitionFunction<apps::SelfCheck<1>>);
static_assert(concepts::TransitionFunction<apps::Conway>);
static_assert(concepts::Grid<cpu::Grid<bool>, bool>);
static_assert(concepts::StencilUpdate<cpu::StencilUpdate<apps::Conway>, apps::Conway, cpu::Grid<bool>>);

struct NotATransitionFunction {
    using Cell = int;
};
static_assert(!concepts::TransitionFunction<NotATransitionFunction>);

static void test_zero_iterations_alias() {
    cpu::Grid<bool> g(4, 4);
    cpu::StencilUpdate<apps::Conway> update({.transition_function = apps::Conway(), .n_iterations = 0});
    cpu::Grid<bool> out = update(g);
    {
        cpu::Grid<bool>::GridAccessor<sycl::access::mode::read_write> ac(out);
        ac[1][1] = true;
    }
    cpu::Grid<bool>::GridAccessor<sycl::access::mode::read> in(g);
    REQUIRE(in[1][1] == true); // n_iterations == 0 returns a handle onto the input
}

static void test_blinker() {
    cpu::Grid<bool> g(5, 5);
    {
        cpu::Grid<bool>::GridAccessor<sycl::access::mode::read_write> ac(g);
        ac[2][1] = ac[2][2] = ac[2][3] = true;
    }
    cpu::StencilUpdate<apps::Conway> update({.transition_function = apps::Conway(), .n_iterations = 1});
    cpu::Grid<bool> out = update(g);
    cpu::Grid<bool>::GridAccessor<sycl::access::mode::read> ac(out);
    REQUIRE(ac[1][2] && ac[2][2] && ac[3][2] && !ac[2][1] && !ac[2][3]);
}

static void test_json() {
    std::istringstream in(R"({"tau": 100e-15, "n": 1024, "time": {"t_max": 15.0}, "rings": [{"radius": 800e-9}, {"radius": 1}], "s": "x\ty"})");
    nlohmann::json j = nlohmann::json::parse(in);
    REQUIRE(j.contains("tau") && !j.contains("nope") && j["tau"].is_number() && j["time"].is_object());
    REQUIRE(j["tau"].get<float>() == float(100e-15));
    std::size_t n = j.at("n");
    double t = j["time"]["t_max"];
    REQUIRE(n == 1024 && t == 15.0 && j["rings"].is_array() && j["rings"].size() == 2);
    int count = 0;
    for (auto ring : j["rings"]) {
        REQUIRE(ring["radius"].is_number());
        count++;
    }
    REQUIRE(count == 2 && std::string(j["rings"].type_name()) == "array");
    REQUIRE(j["s"].get<std::string>() == "x\ty");
    bool threw = false;
    try {
        nlohmann::json::parse(std::string("{\"a\": }"));
    } catch (nlohmann::detail::parse_error const &e) {
        threw = std::string(e.what()).find("parse error") != std::string::npos;
    }
    REQUIRE(threw);
}

static void test_ac_int() {
    ac_int<5, false> i = 0;
    int loops = 0;
    for (; i < 16; i++)
        loops++;
    REQUIRE(loops == 16);
    ac_int<4, false> wrap = 15;
    wrap++;
    REQUIRE(int(wrap) == 0);
    ac_int<4, true> s = 7;
    s++;
    REQUIRE(int(s) == -8);
    float table[4] = {1, 2, 3, 4};
    ac_int<5, false> idx = 2;
    REQUIRE(table[idx] == 3);
}

// The strategy protocol of the reference (tdv/SinglePassStrategies.hpp:44-112: GlobalState -> KernelArgument ->
// LocalState) walked on the host for all three strategies: the value of the pass's i-th iteration.
struct Wave {
    using Cell = int;
    using TimeDependentValue = long;
    static constexpr std::size_t stencil_radius = 1, n_subiterations = 1;
    long scale;
    long get_time_dependent_value(std::size_t i) const { return scale * long(i) + 3; }
    int operator()(Stencil<int, 1, long> const &s) const { return s[0][0]; }
};

template <typename S> static void walk_strategy() {
    using namespace stencil::tdv::single_pass;
    static_assert(Strategy<S, Wave, 4>);
    using Global = typename S::template GlobalState<Wave, 4>;
    Global global(Wave{10}, /*iteration_offset=*/100, /*n_iterations=*/10);
    for (std::size_t pass = 100; pass < 110; pass += 4) {
        const std::size_t n = std::min<std::size_t>(4, 110 - pass);
        typename Global::KernelArgument argument = [&] {
            if constexpr (S::kind == Kind::PrecomputeOnHost)
                return typename Global::KernelArgument(global, pass, n);
            else
                return typename Global::KernelArgument(global, pass);
        }();
        typename Global::KernelArgument::LocalState local(argument);
        for (std::size_t i = 0; i < n; i++)
            REQUIRE(local.get_time_dependent_value(i) == 10 * long(pass + i) + 3);
    }
}

static void test_tdv_strategy_protocol() {
    walk_strategy<stencil::tdv::single_pass::InlineStrategy>();
    walk_strategy<stencil::tdv::single_pass::PrecomputeOnDeviceStrategy>();
    walk_strategy<stencil::tdv::single_pass::PrecomputeOnHostStrategy>();
}

int main() {
    test_tdv_strategy_protocol();
    api_tests::test_stencil_indexing();
    api_tests::test_grid<cpu::Grid<sycl::id<2>>>(128, 128);
    api_tests::test_grid<cpu::Grid<sycl::id<2>>>(3, 17);
    api_tests::test_stencil_update_cases<cpu::Grid<apps::SelfCheckCell>, cpu::StencilUpdate<apps::SelfCheck<1>>>();
    test_zero_iterations_alias();
    test_blinker();
    test_json();
    test_ac_int();
    return finish("host_api_test");
}
