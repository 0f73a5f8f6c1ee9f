// Backend-independent API tests, instantiated for stencil::cpu (host, g++) and stencil::hip
// (MI355X, hipcc).  They follow the reference's own unit tests: tests/Stencil.cpp:28-50,
// tests/GridTest.hpp:24-124 and tests/StencilUpdateTest.hpp:30-63 with the self-checking
// transition function of tests/TransFuncs.hpp:55-104 (here: stencilstream_amd/csrc/apps/selfcheck.hpp).
#pragma once
#include "mini_test.hpp"
#include <StencilStream/Concepts.hpp>
#include <apps/selfcheck.hpp>

namespace api_tests {

using stencil::apps::SelfCheck;
using stencil::apps::SelfCheckCell;
using stencil::apps::SelfCheckStatus;

inline void test_stencil_indexing() {
    constexpr std::size_t radius = 2;
    using S = stencil::Stencil<int, radius>;
    S st(sycl::id<2>(0, 0), sycl::range<2>(42, 42), 0, 0, std::monostate());
    REQUIRE(S::diameter == 2 * radius + 1);
    for (std::size_t r = 0; r < S::diameter; r++)
        for (std::size_t c = 0; c < S::diameter; c++)
            st[sycl::id<2>(r, c)] = int(r) + int(c) - int(2 * radius);
    for (int r = -int(radius); r <= int(radius); r++)
        for (int c = -int(radius); c <= int(radius); c++)
            REQUIRE(st[r][c] == r + c);
    int raw[S::diameter][S::diameter];
    for (std::size_t r = 0; r < S::diameter; r++)
        for (std::size_t c = 0; c < S::diameter; c++)
            raw[r][c] = int(10 * r + c);
    S from_raw(sycl::id<2>(3, 4), sycl::range<2>(9, 9), 5, 1, std::monostate(), raw);
    REQUIRE(from_raw[-2][-2] == 0 && from_raw[0][0] == 22 && from_raw[2][1] == 43);
    REQUIRE(from_raw.id[0] == 3 && from_raw.id[1] == 4 && from_raw.iteration == 5 &&
            from_raw.subiteration == 1 && from_raw.grid_range[1] == 9);
}

template <typename G> void test_grid(std::size_t h, std::size_t w) {
    static_assert(stencil::concepts::Grid<G, sycl::id<2>>);
    G grid(1, 1);
    REQUIRE(grid.get_grid_height() == 1 && grid.get_grid_width() == 1);
    REQUIRE(grid.get_grid_range() == sycl::range<2>(1, 1));
    grid = G(h / 2, w / 2);
    REQUIRE(grid.get_grid_range() == sycl::range<2>(h / 2, w / 2));
    grid = G(sycl::range<2>(h, w));
    REQUIRE(grid.get_grid_height() == h && grid.get_grid_width() == w);

    sycl::buffer<sycl::id<2>, 2> in_buffer = sycl::range<2>(h, w);
    {
        sycl::host_accessor ac(in_buffer, sycl::read_write);
        for (std::size_t r = 0; r < h; r++)
            for (std::size_t c = 0; c < w; c++)
                ac[r][c] = sycl::id<2>(r, c);
    }
    grid = in_buffer; // construct from a buffer
    {
        typename G::template GridAccessor<sycl::access::mode::read> ac(grid);
        bool ok = true;
        for (std::size_t r = 0; r < h; r++)
            for (std::size_t c = 0; c < w; c++)
                ok = ok && ac[r][c] == sycl::id<2>(r, c) && ac[sycl::id<2>(r, c)] == sycl::id<2>(r, c);
        REQUIRE(ok);
    }
    G other(h, w);
    other.copy_from_buffer(in_buffer);
    sycl::buffer<sycl::id<2>, 2> out_buffer = sycl::range<2>(h, w);
    other.copy_to_buffer(out_buffer);
    {
        sycl::host_accessor ac(out_buffer, sycl::read_only);
        bool ok = true;
        for (std::size_t r = 0; r < h; r++)
            for (std::size_t c = 0; c < w; c++)
                ok = ok && ac[r][c] == sycl::id<2>(r, c);
        REQUIRE(ok);
    }
    // size mismatch must throw std::range_error
    sycl::buffer<sycl::id<2>, 2> wrong = sycl::range<2>(h + 1, w);
    bool threw = false;
    try {
        other.copy_from_buffer(wrong);
    } catch (std::range_error const &) {
        threw = true;
    }
    REQUIRE(threw);
    threw = false;
    try {
        other.copy_to_buffer(wrong);
    } catch (std::range_error const &) {
        threw = true;
    }
    REQUIRE(threw);
    // copies share the cells
    G alias(other);
    {
        typename G::template GridAccessor<sycl::access::mode::read_write> ac(alias);
        ac[0][0] = sycl::id<2>(77, 88);
    }
    {
        typename G::template GridAccessor<sycl::access::mode::read> ac(other);
        REQUIRE(ac[0][0] == sycl::id<2>(77, 88));
    }
    G similar = other.make_similar();
    REQUIRE(similar.get_grid_range() == sycl::range<2>(h, w));
}

template <typename G, typename SU>
void test_stencil_update(std::size_t h, std::size_t w, std::size_t offset, std::size_t n) {
    static_assert(stencil::concepts::StencilUpdate<SU, SelfCheck<1>, G>);
    using Accessor = typename G::template GridAccessor<sycl::access::mode::read_write>;
    G input(h, w);
    {
        Accessor ac(input);
        for (std::size_t r = 0; r < h; r++)
            for (std::size_t c = 0; c < w; c++)
                ac[r][c] = SelfCheckCell{int(r), int(c), int(offset), 0, SelfCheckStatus::Normal};
    }
    SU update({.transition_function = SelfCheck<1>(),
               .halo_value = SelfCheckCell::halo(),
               .iteration_offset = offset,
               .n_iterations = n});
    G output = update(input);
    REQUIRE(update.get_n_processed_cells() == n * h * w);
    REQUIRE(update.get_params().n_iterations == n);
    bool ok = true, untouched = true;
    {
        Accessor ac(output);
        for (std::size_t r = 0; r < h; r++)
            for (std::size_t c = 0; c < w; c++)
                ok = ok && ac[r][c].r == int(r) && ac[r][c].c == int(c) &&
                     ac[r][c].i_iteration == int(offset + n) && ac[r][c].i_subiteration == 0 &&
                     ac[r][c].status == SelfCheckStatus::Normal;
    }
    {
        Accessor ac(input); // the source grid is never written
        for (std::size_t r = 0; r < h; r++)
            for (std::size_t c = 0; c < w; c++)
                untouched = untouched && ac[r][c].i_iteration == int(offset) && ac[r][c].r == int(r);
    }
    REQUIRE(ok);
    REQUIRE(untouched);
    // resume: get_params() is live; a second call continues from the new offset
    update.get_params().iteration_offset = offset + n;
    update.get_params().n_iterations = 3;
    G more = update(output);
    {
        Accessor ac(more);
        bool ok2 = true;
        for (std::size_t r = 0; r < h; r++)
            for (std::size_t c = 0; c < w; c++)
                ok2 = ok2 && ac[r][c].i_iteration == int(offset + n + 3) &&
                      ac[r][c].status == SelfCheckStatus::Normal;
        REQUIRE(ok2);
    }
    REQUIRE(update.get_n_processed_cells() == (n + 3) * h * w);
    REQUIRE(update.get_walltime() >= 0.0);
}

template <typename G, typename SU> void test_stencil_update_cases() {
    // the reference's cases (tests/cpu/StencilUpdate.cpp:35-41) plus ragged ones
    test_stencil_update<G, SU>(64, 64, 0, 1);
    test_stencil_update<G, SU>(64, 64, 32, 64);
    test_stencil_update<G, SU>(32, 64, 0, 1);
    test_stencil_update<G, SU>(64, 32, 0, 1);
    test_stencil_update<G, SU>(37, 301, 5, 9);
    test_stencil_update<G, SU>(1, 1, 0, 2);
}

} // namespace api_tests
