// Shape sweep for the reference's PseudoTransientKernel (88-byte fp64 cell, three sub-iterations): the kernel compiled
// with an explicit SweepTuning (-DSHAPE_T / SHAPE_W / SHAPE_P / SHAPE_INTERIOR), timed at res = 1024 and compared bit
// for bit with the shape the generic rule picks (run with SHAPE_DEFAULT).  tools/experiments/convection/run.sh.
#define main reference_convection_main
#include <convection.cpp>
#undef main
#include <chrono>
#include <cstdio>
#include <cstring>
#ifndef SHAPE_DEFAULT
namespace stencil { namespace hip {
template <> struct SweepTuning<PseudoTransientKernel, true> {
    static constexpr int cells_per_lane = 1, max_generations = SHAPE_T, prefetch_rows = SHAPE_P, min_waves_per_simd = 1, stages = SHAPE_W;
    static constexpr bool interior_variant = SHAPE_INTERIOR;
};
}}
#endif
int main(int argc, char **argv) {
    const std::size_t res = 1024, iterations = 200;
    using Cell = ThermalConvectionCell;
    const double lx = 3.0, ly = 1.0, px = 1.5, py = 0.5, eta0 = 1.0, DcT = 1.0, deltaT = 1.0, Ra = 1e7, Pra = 1e3, dmp = 2;
    const std::size_t nx = res * lx - 1, ny = res * ly - 1;
    const double w = 1e-2 * ly, dx = lx / (nx - 1), dy = ly / (ny - 1), rho = 1.0 / Pra * eta0 / DcT;
    const double delta_tau_iter = 1.0 / 6.1 * std::min(dx, dy) / std::sqrt(eta0 / rho);
    PseudoTransientKernel pt{.nx = nx, .ny = ny, .roh0_g_alpha = Ra * eta0 * DcT / deltaT / std::pow(ly, 3),
                             .delta_eta_delta_T = 1e-10 / deltaT, .eta0 = eta0, .deltaT = deltaT, .dx = dx, .dy = dy,
                             .delta_tau_iter = delta_tau_iter,
                             .beta = 6.1 * std::pow(delta_tau_iter, 2) / std::pow(std::min(dx, dy), 2) / rho, .rho = rho,
                             .dampX = 1.0 - dmp / nx, .dampY = 1.0 - dmp / ny, .DcT = DcT};
    Grid grid(nx + 1, ny + 1);
    {
        Grid::GridAccessor<sycl::access::mode::read_write> a(grid);
        for (std::size_t x = 0; x < nx + 1; x++)
            for (std::size_t y = 0; y < ny + 1; y++) {
                Cell cell = Cell::halo_value();
                if (y == 0) cell.T = deltaT / 2.0;
                else if (y == ny - 1) cell.T = -deltaT / 2.0;
                else if (x < nx && y < ny) cell.T = deltaT * std::exp(-std::pow((x * dx - px) / w, 2) - std::pow((y * dy - py) / w, 2));
                cell.Vx = 1e-3 * std::sin(0.37 * x + 0.11 * y);
                cell.Vy = 1e-3 * std::cos(0.23 * x - 0.19 * y);
                a[x][y] = cell;
            }
    }
    PseudoTransientUpdate update({.transition_function = pt, .halo_value = Cell::halo_value(), .n_iterations = iterations, .blocking = true});
    Grid out = update(grid); // warm-up (upload, allocation)
    double best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        auto t0 = std::chrono::high_resolution_clock::now();
        out = update(grid);
        best = std::min(best, std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count());
    }
    // checksum of all bits
    unsigned long long sum = 0;
    {
        Grid::GridAccessor<sycl::access::mode::read> a(out);
        for (std::size_t x = 0; x < nx + 1; x++)
            for (std::size_t y = 0; y < ny + 1; y++) {
                Cell c = a[x][y];
                unsigned long long words[11];
                std::memcpy(words, &c, sizeof c);
                for (auto wd : words) sum = sum * 1099511628211ull + wd;
            }
    }
    std::printf("%s: %zu iterations of %zu x %zu cells: %.4f s = %.1f us per iteration, checksum %016llx\n", argv[0], iterations,
                nx + 1, ny + 1, best, best / iterations * 1e6, sum);
    return 0;
}
