// stencil::hip::BlockUpdate (the template-level block driver) with USER transition functions: the blocks of a mesh as
// THREADS of one process on one GPU, ghost columns and rows through an in-process mailbox (RCCL cannot join several ranks
// on one device; on a box with a GPU per rank the same code passes a communicator instead of the callbacks).  Every mesh's
// assembled result must equal hip::StencilUpdate on the whole grid, bit for bit.
//
// usage: block_template_test            (meshes 1 x 1, 2 x 2, 1 x 3, 3 x 1)
#include <StencilStream/BaseTransitionFunction.hpp>
#include <StencilStream/hip/BlockUpdate.hpp>

#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <tuple>
#include <vector>

using namespace stencil;

namespace {
constexpr std::size_t H = 700, W = 530;

// a time-dependent value, two sub-iterations and the cell's own coordinates (a block's columns do not start at zero),
// exact in fp32 wherever it is evaluated
struct Ramp {
    using Cell = float;
    using TimeDependentValue = float;
    static constexpr std::size_t stencil_radius = 1;
    static constexpr std::size_t n_subiterations = 2;
    float gain;
    float get_time_dependent_value(std::size_t i) const { return float(i % 7) * gain; }
    float operator()(Stencil<float, 1, float> const &s) const {
        if (s.subiteration == 0)
            return 0.5f * (s[0][-1] + s[0][1]) + s.time_dependent_value + float((s.id[1] + 2 * s.id[0]) % 5) * 0.25f;
        return 0.5f * (s[-1][0] + s[1][0]) - 0.25f * s.time_dependent_value + float(s.iteration % 3);
    }
};

// two fields, one of them only copied: swept on per-field planes when the split cell structure is requested
struct Plate {
    float temperature, conductivity;
    static constexpr auto fields = std::make_tuple(&Plate::temperature, &Plate::conductivity);
};
struct Conduction : public BaseTransitionFunction {
    using Cell = Plate;
    Plate operator()(Stencil<Plate, 1> const &s) const {
        Plate me = s[0][0];
        const float k = me.conductivity;
        me.temperature = me.temperature + k * (s[-1][0].temperature + s[1][0].temperature + s[0][-1].temperature +
                                               s[0][1].temperature - 4.0f * me.temperature);
        return me;
    }
};

float ramp_cell(std::size_t r, std::size_t c) { return float((r * 13 + c * 7) % 64) * 0.125f; }
Plate plate_cell(std::size_t r, std::size_t c) {
    return Plate{float((r * 5 + c * 3) % 97) * 0.25f, 0.0625f + 0.015625f * float((r + 2 * c) % 8)};
}

#define CHECK(call)                                                                                 \
    do {                                                                                            \
        int rc_ = (call);                                                                           \
        if (rc_ != STSTHIP_OK) {                                                                    \
            std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ststhip_last_error());         \
            std::exit(2);                                                                           \
        }                                                                                           \
    } while (0)

// one directed edge of the mesh: a buffer of one message
struct Edge {
    std::mutex m;
    std::condition_variable cv;
    bool full = false;
    std::vector<std::vector<unsigned char>> planes;
    void put(int n_planes, const void *const *device, const size_t *row_bytes, size_t n_rows, ststhip_stream stream) {
        std::vector<std::vector<unsigned char>> message(n_planes);
        for (int p = 0; p < n_planes; p++) {
            message[p].resize(row_bytes[p] * n_rows);
            CHECK(ststhip_memcpy_d2h(message[p].data(), device[p], message[p].size(), stream));
        }
        CHECK(ststhip_stream_synchronize(stream));
        std::unique_lock<std::mutex> lock(m);
        cv.wait(lock, [&] { return !full; });
        planes.swap(message);
        full = true;
        cv.notify_all();
    }
    void take(int n_planes, void *const *device, const size_t *row_bytes, size_t n_rows, ststhip_stream stream) {
        std::vector<std::vector<unsigned char>> message;
        {
            std::unique_lock<std::mutex> lock(m);
            cv.wait(lock, [&] { return full; });
            message.swap(planes);
            full = false;
            cv.notify_all();
        }
        for (int p = 0; p < n_planes; p++) {
            if (message[p].size() != row_bytes[p] * n_rows) {
                std::fprintf(stderr, "mailbox: a message of %zu bytes where %zu were expected\n", message[p].size(),
                             row_bytes[p] * n_rows);
                std::exit(2);
            }
            CHECK(ststhip_memcpy_h2d(device[p], message[p].data(), message[p].size(), stream));
        }
        CHECK(ststhip_stream_synchronize(stream));
    }
};
struct Mesh {
    int rows, cols;
    std::vector<Edge> to_up, to_down, to_left, to_right; // indexed by the SENDING rank
    Mesh(int r, int c) : rows(r), cols(c), to_up(r * c), to_down(r * c), to_left(r * c), to_right(r * c) {}
};
struct Side {
    Mesh *mesh;
    int rank;
};
// the contract of ststhip_comm_exchange_rows; "first" / "second" neighbour = up / down for rows, left / right for columns
int exchange_rows(void *ctx, int n_planes, const void *const *send_up, const void *const *send_down, void *const *recv_up,
                  void *const *recv_down, const size_t *row_bytes, size_t n_rows, ststhip_stream stream) {
    Side *s = static_cast<Side *>(ctx);
    Mesh &m = *s->mesh;
    const int r = s->rank / m.cols;
    CHECK(ststhip_stream_synchronize(stream));
    if (r > 0)
        m.to_up[s->rank].put(n_planes, send_up, row_bytes, n_rows, stream);
    if (r + 1 < m.rows)
        m.to_down[s->rank].put(n_planes, send_down, row_bytes, n_rows, stream);
    if (r > 0)
        m.to_down[s->rank - m.cols].take(n_planes, recv_up, row_bytes, n_rows, stream);
    if (r + 1 < m.rows)
        m.to_up[s->rank + m.cols].take(n_planes, recv_down, row_bytes, n_rows, stream);
    return STSTHIP_OK;
}
int exchange_cols(void *ctx, int n_planes, const void *const *send_left, const void *const *send_right,
                  void *const *recv_left, void *const *recv_right, const size_t *row_bytes, size_t n_rows,
                  ststhip_stream stream) {
    Side *s = static_cast<Side *>(ctx);
    Mesh &m = *s->mesh;
    const int c = s->rank % m.cols;
    CHECK(ststhip_stream_synchronize(stream));
    if (c > 0)
        m.to_left[s->rank].put(n_planes, send_left, row_bytes, n_rows, stream);
    if (c + 1 < m.cols)
        m.to_right[s->rank].put(n_planes, send_right, row_bytes, n_rows, stream);
    if (c > 0)
        m.to_right[s->rank - 1].take(n_planes, recv_left, row_bytes, n_rows, stream);
    if (c + 1 < m.cols)
        m.to_left[s->rank + 1].take(n_planes, recv_right, row_bytes, n_rows, stream);
    return STSTHIP_OK;
}

// one rank of one mesh: both functions, its cells into the assembled results
void run_rank(Mesh *mesh, int rank, float *ramp_out, Plate *plate_out) {
    Side side{mesh, rank};
    const bool alone = mesh->rows * mesh->cols == 1;
    ststhip_exchange_fn rows = alone ? nullptr : &exchange_rows, cols = alone ? nullptr : &exchange_cols;
    {
        hip::BlockUpdate<Ramp> block({.transition_function = Ramp{0.5f}, .halo_value = 1.0f, .iteration_offset = 5,
                                      .n_iterations = 29, .blocking = true},
                                     H, W, rank, mesh->rows, mesh->cols, nullptr, rows, &side, cols, &side);
        const std::size_t n_cols = block.end_col() - block.first_col();
        std::vector<float> mine(block.n_cells());
        for (std::size_t r = block.first_row(); r < block.end_row(); r++)
            for (std::size_t c = block.first_col(); c < block.end_col(); c++)
                mine[(r - block.first_row()) * n_cols + (c - block.first_col())] = ramp_cell(r, c);
        block.upload(mine.data());
        block.warm_up();
        block();
        block.get_params().iteration_offset = 34;
        block.get_params().n_iterations = 4;
        block();
        block.download(mine.data());
        for (std::size_t r = block.first_row(); r < block.end_row(); r++)
            std::memcpy(ramp_out + r * W + block.first_col(), mine.data() + (r - block.first_row()) * n_cols,
                        n_cols * sizeof(float));
    }
    {
        hip::BlockUpdate<Conduction, true> block({.transition_function = Conduction{}, .halo_value = Plate{2.0f, 0.125f},
                                                  .n_iterations = 37, .blocking = true},
                                                 H, W, rank, mesh->rows, mesh->cols, nullptr, rows, &side, cols, &side);
        const std::size_t n_cols = block.end_col() - block.first_col();
        std::vector<Plate> mine(block.n_cells());
        for (std::size_t r = block.first_row(); r < block.end_row(); r++)
            for (std::size_t c = block.first_col(); c < block.end_col(); c++)
                mine[(r - block.first_row()) * n_cols + (c - block.first_col())] = plate_cell(r, c);
        block.upload(mine.data());
        block.warm_up();
        block();
        block.download(mine.data());
        for (std::size_t r = block.first_row(); r < block.end_row(); r++)
            std::memcpy(plate_out + r * W + block.first_col(), mine.data() + (r - block.first_row()) * n_cols,
                        n_cols * sizeof(Plate));
    }
}
} // namespace

int main() {
    // what every mesh must reproduce: hip::StencilUpdate on the whole grid
    std::vector<float> ramp_want(H * W);
    std::vector<Plate> plate_want(H * W);
    {
        hip::Grid<float> grid(H, W);
        {
            hip::Grid<float>::GridAccessor<sycl::access::mode::read_write> ac(grid);
            for (std::size_t r = 0; r < H; r++)
                for (std::size_t c = 0; c < W; c++)
                    ac[r][c] = ramp_cell(r, c);
        }
        hip::StencilUpdate<Ramp> update({.transition_function = Ramp{0.5f}, .halo_value = 1.0f, .iteration_offset = 5,
                                         .n_iterations = 29, .blocking = true});
        hip::Grid<float> out = update(grid);
        update.get_params().iteration_offset = 34;
        update.get_params().n_iterations = 4;
        out = update(out);
        hip::Grid<float>::GridAccessor<sycl::access::mode::read> ac(out);
        std::memcpy(ramp_want.data(), ac.get_pointer(), H * W * sizeof(float));
    }
    {
        hip::Grid<Plate> grid(H, W);
        {
            hip::Grid<Plate>::GridAccessor<sycl::access::mode::read_write> ac(grid);
            for (std::size_t r = 0; r < H; r++)
                for (std::size_t c = 0; c < W; c++)
                    ac[r][c] = plate_cell(r, c);
        }
        hip::StencilUpdate<Conduction, true> update({.transition_function = Conduction{}, .halo_value = Plate{2.0f, 0.125f},
                                                     .n_iterations = 37, .blocking = true});
        hip::Grid<Plate> out = update(grid);
        hip::Grid<Plate>::GridAccessor<sycl::access::mode::read> ac(out);
        std::memcpy(plate_want.data(), ac.get_pointer(), H * W * sizeof(Plate));
    }
    const int meshes[4][2] = {{1, 1}, {2, 2}, {1, 3}, {3, 1}};
    for (auto const &shape : meshes) {
        Mesh mesh(shape[0], shape[1]);
        std::vector<float> ramp_got(H * W, -1.0f);
        std::vector<Plate> plate_got(H * W, Plate{-1.0f, -1.0f});
        std::vector<std::thread> ranks;
        for (int rank = 0; rank < shape[0] * shape[1]; rank++)
            ranks.emplace_back(run_rank, &mesh, rank, ramp_got.data(), plate_got.data());
        for (auto &t : ranks)
            t.join();
        if (std::memcmp(ramp_got.data(), ramp_want.data(), H * W * sizeof(float)) != 0) {
            std::fprintf(stderr, "Ramp: the %d x %d mesh differs from hip::StencilUpdate\n", shape[0], shape[1]);
            return 1;
        }
        if (std::memcmp(plate_got.data(), plate_want.data(), H * W * sizeof(Plate)) != 0) {
            std::fprintf(stderr, "Conduction: the %d x %d mesh differs from hip::StencilUpdate\n", shape[0], shape[1]);
            return 1;
        }
        std::printf("block_template_test: mesh %d x %d equals hip::StencilUpdate\n", shape[0], shape[1]);
    }
    return 0;
}
