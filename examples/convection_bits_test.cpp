// Bit-level parity of the reference's convection kernels on the MI355X backend against the same kernels on this
// repository's stencil::cpu backend: both in one binary, the same grid, every field of every cell compared as bits
// (fp64; both sides are built with -ffp-contract=off, IEEE division and no libm call inside the kernels).  The kernels
// and the cell type are the reference's own (examples/convection/convection.cpp:36-242, included with its main()
// renamed); the text comparison of the example's CSV files (tests/test_examples.py) only sees six digits.
//
// usage: convection_bits_test <res> <pseudo-transient iterations> ; exit code 0 = identical
#include <StencilStream/cpu/StencilUpdate.hpp>
#define main reference_convection_main
#include <convection.cpp> // STENCILSTREAM_BACKEND_CUDA: Grid / PseudoTransientUpdate / ThermalSolverUpdate = stencil::hip
#undef main

#include <cstdio>
#include <cstdlib>
#include <cstring>

int main(int argc, char **argv) {
    const std::size_t res = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 64;
    const std::size_t iterations = argc > 2 ? std::strtoul(argv[2], nullptr, 10) : 30;
    using Cell = ThermalConvectionCell;
    // the default experiment's physics (examples/convection/experiments/default.json) at resolution `res`
    const double lx = 3.0, ly = 1.0, px = 1.5, py = 0.5, eta0 = 1.0, DcT = 1.0, deltaT = 1.0, Ra = 1e7, Pra = 1e3, dmp = 2;
    const std::size_t nx = res * lx - 1, ny = res * ly - 1;
    const double w = 1e-2 * ly, dx = lx / (nx - 1), dy = ly / (ny - 1), rho = 1.0 / Pra * eta0 / DcT;
    const double delta_tau_iter = 1.0 / 6.1 * std::min(dx, dy) / std::sqrt(eta0 / rho);
    PseudoTransientKernel pt{.nx = nx, .ny = ny, .roh0_g_alpha = Ra * eta0 * DcT / deltaT / std::pow(ly, 3),
                             .delta_eta_delta_T = 1e-10 / deltaT, .eta0 = eta0, .deltaT = deltaT, .dx = dx, .dy = dy,
                             .delta_tau_iter = delta_tau_iter,
                             .beta = 6.1 * std::pow(delta_tau_iter, 2) / std::pow(std::min(dx, dy), 2) / rho, .rho = rho,
                             .dampX = 1.0 - dmp / nx, .dampY = 1.0 - dmp / ny, .DcT = DcT};
    ThermalSolverKernel ts{.nx = nx, .ny = ny, .dx = dx, .dy = dy, .dt = 1.0 / 4.1 * std::pow(std::min(dx, dy), 2) / DcT,
                           .DcT = DcT};

    Grid device_grid(nx + 1, ny + 1);
    stencil::cpu::Grid<Cell> host_grid(nx + 1, ny + 1);
    {
        Grid::GridAccessor<sycl::access::mode::read_write> a(device_grid);
        stencil::cpu::Grid<Cell>::GridAccessor<sycl::access::mode::read_write> b(host_grid);
        for (std::size_t x = 0; x < nx + 1; x++)
            for (std::size_t y = 0; y < ny + 1; y++) {
                Cell cell = Cell::halo_value();
                if (y == 0)
                    cell.T = deltaT / 2.0;
                else if (y == ny - 1)
                    cell.T = -deltaT / 2.0;
                else if (x < nx && y < ny)
                    cell.T = deltaT * std::exp(-std::pow((x * dx - px) / w, 2) - std::pow((y * dy - py) / w, 2));
                // velocities that are not zero, so that every term of the kernels is exercised from the first step
                cell.Vx = 1e-3 * std::sin(0.37 * x + 0.11 * y);
                cell.Vy = 1e-3 * std::cos(0.23 * x - 0.19 * y);
                a[x][y] = cell;
                b[x][y] = cell;
            }
    }
    PseudoTransientUpdate device_pt({.transition_function = pt, .halo_value = Cell::halo_value(), .n_iterations = iterations,
                                     .blocking = true});
    stencil::cpu::StencilUpdate<PseudoTransientKernel> host_pt(
        {.transition_function = pt, .halo_value = Cell::halo_value(), .n_iterations = iterations, .blocking = true});
    ThermalSolverUpdate device_ts({.transition_function = ts, .halo_value = Cell::halo_value(), .n_iterations = 1, .blocking = true});
    stencil::cpu::StencilUpdate<ThermalSolverKernel> host_ts(
        {.transition_function = ts, .halo_value = Cell::halo_value(), .n_iterations = 1, .blocking = true});
    std::size_t differing = 0, compared = 0;
    for (int round = 0; round < 2; round++) { // two time steps: pseudo-transient block, then the thermal solver
        device_grid = device_pt(device_grid);
        host_grid = host_pt(host_grid);
        device_grid = device_ts(device_grid);
        host_grid = host_ts(host_grid);
        Grid::GridAccessor<sycl::access::mode::read> a(device_grid);
        stencil::cpu::Grid<Cell>::GridAccessor<sycl::access::mode::read> b(host_grid);
        for (std::size_t x = 0; x < nx + 1; x++)
            for (std::size_t y = 0; y < ny + 1; y++) {
                const Cell ca = a[x][y], cb = b[x][y];
                compared++;
                if (std::memcmp(&ca, &cb, sizeof(Cell)) != 0 && differing++ < 5)
                    std::printf("round %d cell (%zu, %zu): T %.17g vs %.17g, Vx %.17g vs %.17g, Pt %.17g vs %.17g\n", round,
                                x, y, ca.T, cb.T, ca.Vx, cb.Vx, ca.Pt, cb.Pt);
            }
    }
    std::printf("convection_bits_test: res %zu, %zu x %zu cells of %zu bytes, 2 x (%zu pseudo-transient iterations + 1 "
                "thermal step): %zu of %zu cell comparisons differ\n",
                res, nx + 1, ny + 1, sizeof(Cell), iterations, differing, compared);
    return differing == 0 ? 0 : 1;
}
