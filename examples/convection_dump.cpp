// The reference's convection kernels (examples/convection/convection.cpp:36-242, included with its main() renamed) on
// the backend the build selects -- STENCILSTREAM_BACKEND_CPU: this repository's stencil::cpu (g++),
// STENCILSTREAM_BACKEND_CUDA: stencil::hip (MI355X) -- on a deterministic grid, with the input, the kernels'
// parameters and every field of every cell after each phase written as raw bytes.  tests/test_oracle_golden.py and
// tests/test_examples.py run oracle/stencil_oracle.c's restatement of the two kernels on the same input and compare
// all eleven fp64 fields as bits: the oracle is checked against the reference's functor source (cpu build), the
// MI355X backend against the oracle (hip build).
//
// usage: convection_dump_{cpu,hip} <res> <pseudo-transient iterations> <output directory>
//   writes params.bin (2 x u64 nx ny, then 14 + 4 doubles: the PseudoTransientKernel and ThermalSolverKernel members in
//   declaration order), init.bin, and round<r>_{pt,ts}.bin for r = 0, 1 (cells after the pseudo-transient block and
//   after the thermal step of time step r), (nx+1) x (ny+1) cells of 88 bytes, row-major
#if defined(STENCILSTREAM_BACKEND_CPU)
    #include <StencilStream/cpu/StencilUpdate.hpp>
#endif
#define main reference_convection_main
#include <convection.cpp>
#undef main

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

static void dump(Grid &grid, std::size_t rows, std::size_t cols, std::string const &path) {
    Grid::GridAccessor<sycl::access::mode::read> ac(grid);
    std::ofstream out(path, std::ios::binary);
    for (std::size_t x = 0; x < rows; x++)
        for (std::size_t y = 0; y < cols; y++) {
            const ThermalConvectionCell cell = ac[x][y];
            out.write(reinterpret_cast<const char *>(&cell), sizeof cell);
        }
}

int main(int argc, char **argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s <res> <pseudo-transient iterations> <output directory>\n", argv[0]);
        return 2;
    }
    const std::size_t res = std::strtoul(argv[1], nullptr, 10), iterations = std::strtoul(argv[2], nullptr, 10);
    const std::string dir = argv[3];
    using Cell = ThermalConvectionCell;
    static_assert(sizeof(Cell) == 88);
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
    {
        std::ofstream out(dir + "/params.bin", std::ios::binary);
        const std::uint64_t dims[2] = {nx, ny};
        const double values[18] = {pt.roh0_g_alpha, pt.delta_eta_delta_T, pt.eta0, pt.deltaT, pt.dx, pt.dy, pt.delta_tau_iter,
                                   pt.beta, pt.rho, pt.dampX, pt.dampY, pt.DcT, 0.0, 0.0, ts.dx, ts.dy, ts.dt, ts.DcT};
        out.write(reinterpret_cast<const char *>(dims), sizeof dims);
        out.write(reinterpret_cast<const char *>(values), sizeof values);
    }
    Grid grid(nx + 1, ny + 1);
    {
        Grid::GridAccessor<sycl::access::mode::read_write> a(grid);
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
            }
    }
    dump(grid, nx + 1, ny + 1, dir + "/init.bin");
    PseudoTransientUpdate pt_update({.transition_function = pt, .halo_value = Cell::halo_value(), .n_iterations = iterations,
                                     .blocking = true});
    ThermalSolverUpdate ts_update({.transition_function = ts, .halo_value = Cell::halo_value(), .n_iterations = 1, .blocking = true});
    for (int round = 0; round < 2; round++) {
        grid = pt_update(grid);
        dump(grid, nx + 1, ny + 1, dir + "/round" + std::to_string(round) + "_pt.bin");
        grid = ts_update(grid);
        dump(grid, nx + 1, ny + 1, dir + "/round" + std::to_string(round) + "_ts.bin");
    }
    std::printf("convection_dump: res %zu, %zu x %zu cells, 2 x (%zu pseudo-transient iterations + 1 thermal step)\n", res,
                nx + 1, ny + 1, iterations);
    return 0;
}
