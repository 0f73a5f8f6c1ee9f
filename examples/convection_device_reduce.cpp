// Thermal convection with the convergence check on the device.
//
// The reference's example (examples/convection/convection.cpp) scans the whole grid on the host after every
// `nerr` pseudo-transient iterations (:412-438): five maxima of |field| over slightly different index ranges.
// With the MI355X backend that scan -- a download of every 88-byte cell plus a serial loop -- costs as much as
// the sweeps themselves.  This driver is the same program with that one block replaced by
// stencil::hip::max_abs (StencilStream/hip/Reduce.hpp, an extension of the API): the cells stay in HBM.
//
// Nothing of the physics is restated here: the transition functions, the cell type and the backend aliases are
// the reference's own, compiled from where they lie (the file is included with its main() renamed); only the
// driver loop is this file's.  Command line, stdout lines and CSV output follow convection.cpp:270-487 so the
// two binaries can be compared file for file (tests/test_examples.py).
#define main reference_convection_main
#include <convection.cpp> // -I<reference>/examples/convection
#undef main

#include <StencilStream/hip/Reduce.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <iostream>
#include <limits>
#include <string>

namespace {

// The experiment file and what convection.cpp:299-343 derives from it (same expressions, same order).
struct Setup {
    double lx, ly, px, py, eta0, DcT, deltaT, Ra, Pra;
    std::size_t res, iterMax, nt, nout, nerr;
    double epsilon, dmp;

    explicit Setup(json const &e)
        : lx(e.at("lx")), ly(e.at("ly")), px(e.at("px")), py(e.at("py")), eta0(e.at("eta0")), DcT(e.at("DcT")),
          deltaT(e.at("deltaT")), Ra(e.at("Ra")), Pra(e.at("Pra")), res(e.at("res")), iterMax(e.at("iterMax")),
          nt(e.at("nt")), nout(e.at("nout")), nerr(e.at("nerr")), epsilon(e.at("epsilon")), dmp(e.at("dmp")) {}

    std::size_t nx() const { return res * lx - 1; }
    std::size_t ny() const { return res * ly - 1; }
    double w() const { return 1e-2 * ly; }
    double roh0_g_alpha() const { return Ra * eta0 * DcT / deltaT / std::pow(ly, 3); }
    double delta_eta_delta_T() const { return 1e-10 / deltaT; }
    double dx() const { return lx / (nx() - 1); }
    double dy() const { return ly / (ny() - 1); }
    double rho() const { return 1.0 / Pra * eta0 / DcT; }
    double dt_diff() const { return 1.0 / 4.1 * std::pow(std::min(dx(), dy()), 2) / DcT; }
    double delta_tau_iter() const { return 1.0 / 6.1 * std::min(dx(), dy()) / std::sqrt(eta0 / rho()); }
    double beta() const { return 6.1 * std::pow(delta_tau_iter(), 2) / std::pow(std::min(dx(), dy()), 2) / rho(); }
    double dampX() const { return 1.0 - dmp / nx(); }
    double dampY() const { return 1.0 - dmp / ny(); }
};

int usage(const char *argv0) {
    std::cerr << "Usage: " << argv0 << " <path to experiment>.json <path to output directory>" << std::endl;
    return 1;
}

} // namespace

int main(int argc, char **argv) {
    if (argc != 3)
        return usage(argv[0]);
    std::filesystem::path experiment_path(argv[1]), output_dir(argv[2]);
    if (!std::filesystem::is_regular_file(experiment_path)) {
        std::cerr << "The experiment file does not exist or is not a regular file." << std::endl;
        return 1;
    }
    if (!std::filesystem::is_directory(output_dir)) {
        std::cerr << "The output directory does not exist or is not a directory." << std::endl;
        return 1;
    }
    std::ifstream experiment_file(experiment_path);
    if (!experiment_file.is_open()) {
        std::cerr << "Could not open experiment file!" << std::endl;
        return 1;
    }
    json experiment;
    try {
        experiment = json::parse(experiment_file);
    } catch (json::parse_error e) {
        std::cerr << "Could not parse experiment file:" << std::endl << e.what() << std::endl;
        return 1;
    }
    const Setup s(experiment);
    const std::size_t nx = s.nx(), ny = s.ny();
    const double dx = s.dx(), dy = s.dy();
    using Cell = ThermalConvectionCell;

    PseudoTransientUpdate pseudo_transient_update({
        .transition_function = PseudoTransientKernel{.nx = nx, .ny = ny, .roh0_g_alpha = s.roh0_g_alpha(),
                                                     .delta_eta_delta_T = s.delta_eta_delta_T(), .eta0 = s.eta0,
                                                     .deltaT = s.deltaT, .dx = dx, .dy = dy,
                                                     .delta_tau_iter = s.delta_tau_iter(), .beta = s.beta(),
                                                     .rho = s.rho(), .dampX = s.dampX(), .dampY = s.dampY(),
                                                     .DcT = s.DcT},
        .halo_value = Cell::halo_value(),
        .n_iterations = s.nerr,
        .blocking = true,
    });

    // initial temperature field (convection.cpp:379-397)
    Grid grid(nx + 1, ny + 1);
    {
        Grid::GridAccessor<sycl::access::mode::read_write> ac(grid);
        for (std::size_t x = 0; x < nx + 1; x++)
            for (std::size_t y = 0; y < ny + 1; y++) {
                Cell cell = Cell::halo_value();
                if (y == 0)
                    cell.T = s.deltaT / 2.0;
                else if (y == ny - 1)
                    cell.T = -s.deltaT / 2.0;
                else if (x < nx && y < ny)
                    cell.T = s.deltaT * std::exp(-std::pow((x * dx - s.px) / s.w(), 2) - std::pow((y * dy - s.py) / s.w(), 2));
                ac[x][y] = cell;
            }
    }

    auto computation_start = std::chrono::system_clock::now();
    for (std::size_t it = 1; it <= s.nt; it++) {
        double errV = 2 * s.epsilon, errP = 2 * s.epsilon;
        double max_Vx = 0, max_Vy = 0;
        std::size_t iter;
        auto transients_start = std::chrono::high_resolution_clock::now();
        for (iter = 0; iter < s.iterMax && (errV > s.epsilon || errP > s.epsilon); iter += s.nerr) {
            grid = pseudo_transient_update(grid);
            // the five maxima of convection.cpp:412-438, index ranges as there, computed where the cells are
            const std::vector<double> m = stencil::hip::max_abs(
                grid, {stencil::hip::over(&Cell::ErrV, nx, ny + 1), stencil::hip::over(&Cell::ErrP, nx, ny),
                       stencil::hip::over(&Cell::Vx, nx + 1, ny), stencil::hip::over(&Cell::Vy, nx, ny),
                       stencil::hip::over(&Cell::Pt, nx, ny)});
            max_Vx = m[2];
            max_Vy = m[3];
            errV = m[0] / (1e-12 + max_Vy);
            errP = m[1] / (1e-12 + m[4]);
        }
        std::chrono::duration<double> transients_time = std::chrono::high_resolution_clock::now() - transients_start;
        printf("it = %zu (iter = %zu, time = %e), errV=%1.3e, errP=%1.3e \n", it, iter, transients_time.count(), errV,
               errP);

        const double dt = std::min(s.dt_diff(), std::min(dx / max_Vx, dy / max_Vy) / 2.1);
        ThermalSolverUpdate thermal_solver_update({
            .transition_function = ThermalSolverKernel{.nx = nx, .ny = ny, .dx = dx, .dy = dy, .dt = dt, .DcT = s.DcT},
            .halo_value = Cell::halo_value(),
            .n_iterations = 1,
        });
        grid = thermal_solver_update(grid);

        if (it % s.nout == 0) {
            std::ofstream out_file(output_dir / std::filesystem::path(std::to_string(it) + ".csv"));
            Grid::GridAccessor<sycl::access::mode::read> ac(grid);
            for (std::size_t x = 0; x < nx; x++) {
                for (std::size_t y = 0; y < ny; y++) {
                    out_file << ac[x][y].T;
                    if (y != ny - 1)
                        out_file << ",";
                }
                out_file << "\n";
            }
        }
    }
    std::chrono::duration<double> computation_time = std::chrono::system_clock::now() - computation_start;
    std::cout << "Total time = " << computation_time.count() << std::endl;
    std::cout << "Of which transient computation time: " << pseudo_transient_update.get_walltime() << " s" << std::endl;
    return 0;
}
