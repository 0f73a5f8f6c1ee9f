// Jacobi on this repository's stencil::cpu backend: the CPU baseline of bench.py ("the build's own
// stencil::cpu backend ... on all host cores").  The reference's examples/jacobi/jacobi.cpp has no cpu branch (it
// selects monotile / tiling / cuda, jacobi.cpp:27-58), so this is its main() for the backend it lacks: same command
// line, same centred-square set-up (jacobi.cpp:107-137), same `Walltime:` line; the transition functions are the
// reference's own (examples/jacobi/kernels.hpp, compiled from where it lies, JACOBI_KERNEL as in its build).
#include <StencilStream/cpu/StencilUpdate.hpp>

#include <kernels.hpp> // -I<reference>/examples/jacobi

#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

using namespace stencil;

#ifndef JACOBI_KERNEL
    #define JACOBI_KERNEL Jacobi5General
#endif
using JacobiKernel = JACOBI_KERNEL;
using StencilUpdate = cpu::StencilUpdate<JacobiKernel>;
using Grid = StencilUpdate::GridImpl;

// kernels.hpp calls this when the coefficient count is wrong (defined by jacobi.cpp:62-76 in the reference's example)
void print_usage(int argc, char **argv) {
    std::cerr << "Usage: " << argv[0] << " <grid_rows> <grid_cols> <no. of iterations> <output_file> <coef...>" << std::endl;
    std::exit(1);
}

int main(int argc, char **argv) {
    if (argc < 5)
        print_usage(argc, argv);
    sycl::range<2> range(std::atoi(argv[1]), std::atoi(argv[2]));
    const std::size_t n_iterations = std::atoi(argv[3]);
    const std::string out_path(argv[4]);

    Grid grid(range);
    {
        Grid::GridAccessor<sycl::access::mode::read_write> grid_ac(grid);
        for (std::size_t r = 0; r < range[0]; r++)
            for (std::size_t c = 0; c < range[1]; c++)
                grid_ac[r][c] = (r >= range[0] * 0.25 && r < range[0] * 0.75 && c >= range[1] * 0.25 && c < range[1] * 0.75)
                                    ? 1.0f
                                    : 0.0f;
    }
    StencilUpdate update({
        .transition_function = JacobiKernel(argc, argv),
        .halo_value = 0.0,
        .n_iterations = n_iterations,
        .blocking = true,
    });
    std::cout << "Starting simulation" << std::endl;
    grid = update(grid);
    std::cout << "Simulation complete!" << std::endl;
    std::cout << "Walltime: " << update.get_walltime() << " s" << std::endl;
    if (out_path != "/dev/null") {
        Grid::GridAccessor<sycl::access::mode::read> grid_ac(grid);
        std::fstream out(out_path, out.out | out.trunc | out.binary);
        if (!out.is_open())
            throw std::runtime_error("The output file can't be opened!\n");
        for (std::size_t r = 0; r < range[0]; r++)
            for (std::size_t c = 0; c < range[1]; c++)
                out.write((char *)&grid_ac[r][c], sizeof(float));
    }
    return 0;
}
