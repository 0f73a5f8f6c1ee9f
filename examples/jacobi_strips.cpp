// Jacobi on a grid that is cut into row strips over several MI355X, one process per GPU -- the counterpart, for this
// backend, of the reference example's multi-device mode (examples/jacobi/jacobi.cpp:83-92,148,173: MPI ranks, one
// FPGA each, behind the same StencilUpdate interface).
//
// Nothing of the numerics is restated: the transition functions are the reference's own (examples/jacobi/kernels.hpp,
// compiled from where it lies, JACOBI_KERNEL chosen on the command line of the compiler as in the reference's build).
// Command line as jacobi.cpp:62-76:   jacobi_strips <rows> <cols> <iterations> <output file> <coefficients...>
// Ranks: RANK / WORLD_SIZE / LOCAL_RANK from the environment (torchrun, mpirun wrappers and Slurm set them; default
// one rank).  The RCCL communicator's id travels through a file: STST_ID_FILE (default /tmp/jacobi_strips.id.<MASTER_PORT>),
// written by rank 0.  Rank 0 prints "Walltime:" as the reference does; every rank writes its own rows of the output file
// at their offset (raw row-major fp32, jacobi.cpp:150-154), so the file equals the single-GPU example's.
#include <StencilStream/hip/StripUpdate.hpp>

#include <kernels.hpp> // -I<reference>/examples/jacobi

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

using namespace stencil;

#ifndef JACOBI_KERNEL
    #define JACOBI_KERNEL Jacobi5General
#endif
using JacobiKernel = JACOBI_KERNEL;

// kernels.hpp calls this when the coefficient count is wrong (defined by jacobi.cpp:62-76 in the reference's example)
void print_usage(int argc, char **argv) {
    std::cerr << "Usage: " << argv[0] << " <grid_rows> <grid_cols> <no. of iterations> <output_file> <coef...>" << std::endl;
    std::exit(1);
}

namespace {
int env_int(const char *name, int fallback) {
    const char *v = std::getenv(name);
    return (v && *v) ? std::atoi(v) : fallback;
}

void check(int rc, const char *what) {
    if (rc != STSTHIP_OK) {
        std::cerr << what << ": " << ststhip_last_error() << std::endl;
        std::exit(1);
    }
}

// the unique id of the communicator: rank 0 creates it, the others read it from the file
ststhip_comm join(int rank, int n_ranks) {
    if (n_ranks == 1)
        return nullptr;
    const char *named = std::getenv("STST_ID_FILE");
    const char *port = std::getenv("MASTER_PORT");
    const std::string path = named ? named : std::string("/tmp/jacobi_strips.id.") + (port ? port : "0");
    unsigned char id[STSTHIP_COMM_ID_BYTES];
    if (rank == 0) {
        check(ststhip_comm_unique_id(id), "ststhip_comm_unique_id");
        const std::string tmp = path + ".tmp";
        std::ofstream(tmp, std::ios::binary).write(reinterpret_cast<const char *>(id), sizeof id);
        std::rename(tmp.c_str(), path.c_str());
    } else {
        for (int tries = 0;; tries++) {
            std::ifstream in(path, std::ios::binary);
            if (in && in.read(reinterpret_cast<char *>(id), sizeof id))
                break;
            if (tries > 6000) {
                std::cerr << "no communicator id in " << path << std::endl;
                std::exit(1);
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
    }
    ststhip_comm comm = nullptr;
    check(ststhip_comm_create(id, rank, n_ranks, &comm), "ststhip_comm_create");
    if (rank == 0)
        std::remove(path.c_str());
    return comm;
}
} // namespace

int main(int argc, char **argv) {
    const int rank = env_int("RANK", 0), n_ranks = env_int("WORLD_SIZE", 1), device = env_int("LOCAL_RANK", rank);
    if (argc < n_main_arguments + int(JacobiKernel::n_coefficients))
        print_usage(argc, argv);
    const std::size_t rows = std::atoi(argv[1]), cols = std::atoi(argv[2]), n_iterations = std::atoi(argv[3]);
    const std::string out_path(argv[4]);

    check(ststhip_init(device), "ststhip_init");
    ststhip_comm comm = join(rank, n_ranks);

    using Strip = hip::StripUpdate<JacobiKernel>;
    Strip strip({.transition_function = JacobiKernel(argc, argv), .halo_value = 0.0, .n_iterations = n_iterations, .blocking = true},
                rows, cols, rank, n_ranks, comm);
    // the initial grid of jacobi.cpp:111-123, this strip's rows of it
    std::vector<float> mine(strip.n_cells());
    for (std::size_t r = strip.first_row(); r < strip.end_row(); r++)
        for (std::size_t c = 0; c < cols; c++)
            mine[(r - strip.first_row()) * cols + c] =
                (r >= rows * 0.25 && r < rows * 0.75 && c >= cols * 0.25 && c < cols * 0.75) ? 1.0f : 0.0f;
    strip.upload(mine.data());
    strip.warm_up();

    if (rank == 0)
        std::cout << "Starting simulation" << std::endl;
    const auto started = std::chrono::high_resolution_clock::now();
    strip();
    const std::chrono::duration<double> walltime = std::chrono::high_resolution_clock::now() - started;
    if (rank == 0) {
        std::cout << "Simulation complete!" << std::endl;
        std::cout << "Walltime: " << walltime.count() << " s" << std::endl;
        std::cout << "Strips: " << n_ranks << ", rows per strip: " << strip.end_row() - strip.first_row() << std::endl;
    }

    strip.download(mine.data());
    if (rank == 0) // the file at its full size first, then everybody writes their rows in place
        std::ofstream(out_path, std::ios::binary | std::ios::trunc);
    if (out_path != "/dev/null") {
        // (ranks other than 0 may arrive before the file exists on a shared file system: retry briefly)
        std::fstream out;
        for (int tries = 0; tries < 1000 && !out.is_open(); tries++) {
            out.open(out_path, std::ios::in | std::ios::out | std::ios::binary);
            if (!out.is_open())
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        if (!out.is_open())
            throw std::runtime_error("The output file can't be opened!\n");
        out.seekp(std::streamoff(strip.first_row() * cols * sizeof(float)));
        out.write(reinterpret_cast<const char *>(mine.data()), std::streamsize(mine.size() * sizeof(float)));
    }
    if (comm)
        ststhip_comm_destroy(comm);
    return 0;
}
