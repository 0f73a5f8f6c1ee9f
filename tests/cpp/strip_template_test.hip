// stencil::hip::StripUpdate (the template-level strip driver) with USER transition functions: one process per strip
// on one GPU, the ghost rows through a mailbox in a shared file (RCCL cannot join two ranks on one device; on a box
// with a GPU per rank the same code passes a communicator instead of the callback).
//
// usage: strip_template_test <rank> <n_ranks> <mailbox file>      (tests/test_cpp_api.py starts the ranks)
//   every rank writes its rows of both results into the file; n_ranks = 1 additionally checks that the single strip
//   equals hip::StencilUpdate on the whole grid, bit for bit.
#include <StencilStream/BaseTransitionFunction.hpp>
#include <StencilStream/hip/StripUpdate.hpp>
#include <StencilStream/tdv/SinglePassStrategies.hpp>

#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <tuple>
#include <unistd.h>
#include <vector>

using namespace stencil;

namespace {
constexpr std::size_t H = 700, W = 530, max_rows = 128, max_planes = 2; // ghost rows of one exchange: up to four launches

// a time-dependent value and two sub-iterations, exact in fp32 wherever it is evaluated
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

struct Slot {
    std::atomic<std::uint64_t> written, read;
    unsigned char rows[max_planes][max_rows * W * sizeof(float)];
};
struct Mailbox {
    Slot down[8], up[8]; // down[b]: rank b -> rank b + 1; up[b]: rank b + 1 -> rank b
    float ramp_result[H * W];
    Plate plate_result[H * W];
};
struct ExchangeContext {
    Mailbox *box;
    int rank, n_ranks;
    std::uint64_t round;
};

#define CHECK(call)                                                                                 \
    do {                                                                                            \
        int rc_ = (call);                                                                           \
        if (rc_ != STSTHIP_OK) {                                                                    \
            std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ststhip_last_error());         \
            std::exit(2);                                                                           \
        }                                                                                           \
    } while (0)

void put(Slot &slot, std::uint64_t round, int n_planes, const void *const *device_rows, const size_t *row_bytes,
         size_t n_rows, ststhip_stream stream) {
    while (slot.read.load(std::memory_order_acquire) != round)
        usleep(50);
    for (int p = 0; p < n_planes; p++)
        CHECK(ststhip_memcpy_d2h(slot.rows[p], device_rows[p], row_bytes[p] * n_rows, stream));
    CHECK(ststhip_stream_synchronize(stream));
    slot.written.store(round + 1, std::memory_order_release);
}
void take(Slot &slot, std::uint64_t round, int n_planes, void *const *device_rows, const size_t *row_bytes, size_t n_rows,
          ststhip_stream stream) {
    while (slot.written.load(std::memory_order_acquire) != round + 1)
        usleep(50);
    for (int p = 0; p < n_planes; p++)
        CHECK(ststhip_memcpy_h2d(device_rows[p], slot.rows[p], row_bytes[p] * n_rows, stream));
    CHECK(ststhip_stream_synchronize(stream));
    slot.read.store(round + 1, std::memory_order_release);
}
// the contract of ststhip_comm_exchange_rows, staged through the mailbox
int exchange(void *ctx, int n_planes, const void *const *send_up, const void *const *send_down, void *const *recv_up,
             void *const *recv_down, const size_t *row_bytes, size_t n_rows, ststhip_stream stream) {
    ExchangeContext *c = static_cast<ExchangeContext *>(ctx);
    if (n_planes > int(max_planes) || n_rows > max_rows)
        return STSTHIP_ERR_INVALID;
    CHECK(ststhip_stream_synchronize(stream));
    if (c->rank > 0)
        put(c->box->up[c->rank - 1], c->round, n_planes, send_up, row_bytes, n_rows, stream);
    if (c->rank + 1 < c->n_ranks)
        put(c->box->down[c->rank], c->round, n_planes, send_down, row_bytes, n_rows, stream);
    if (c->rank > 0)
        take(c->box->down[c->rank - 1], c->round, n_planes, recv_up, row_bytes, n_rows, stream);
    if (c->rank + 1 < c->n_ranks)
        take(c->box->up[c->rank], c->round, n_planes, recv_down, row_bytes, n_rows, stream);
    c->round++;
    return STSTHIP_OK;
}

float ramp_cell(std::size_t r, std::size_t c) { return float((r * 13 + c * 7) % 64) * 0.125f; }
Plate plate_cell(std::size_t r, std::size_t c) {
    return Plate{float((r * 5 + c * 3) % 97) * 0.25f, 0.0625f + 0.015625f * float((r + 2 * c) % 8)};
}
} // namespace

int main(int argc, char **argv) {
    if (argc == 2 && std::strcmp(argv[1], "layout") == 0) { // for the test driver: size of the file and where the results are
        std::printf("%zu %zu %zu %zu %zu\n", sizeof(Mailbox), offsetof(Mailbox, ramp_result), offsetof(Mailbox, plate_result),
                    H, W);
        return 0;
    }
    if (argc != 4) {
        std::fprintf(stderr, "usage: strip_template_test <rank> <n_ranks> <mailbox file>\n");
        return 2;
    }
    const int rank = std::atoi(argv[1]), n_ranks = std::atoi(argv[2]);
    const int fd = open(argv[3], O_RDWR);
    if (fd < 0 || rank < 0 || rank >= n_ranks || n_ranks > 8)
        return 2;
    Mailbox *box = static_cast<Mailbox *>(mmap(nullptr, sizeof(Mailbox), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
    if (box == MAP_FAILED)
        return 2;
    ExchangeContext ctx{box, rank, n_ranks, 0};
    ststhip_exchange_fn callback = n_ranks > 1 ? &exchange : nullptr;

    { // Ramp: time-dependent values, sub-iterations, two calls with an iteration offset
        using Strip = hip::StripUpdate<Ramp>;
        Strip strip({.transition_function = Ramp{0.5f}, .halo_value = 1.0f, .iteration_offset = 5, .n_iterations = 29,
                     .blocking = true},
                    H, W, rank, n_ranks, nullptr, callback, &ctx);
        std::vector<float> mine(strip.n_cells());
        for (std::size_t r = strip.first_row(); r < strip.end_row(); r++)
            for (std::size_t c = 0; c < W; c++)
                mine[(r - strip.first_row()) * W + c] = ramp_cell(r, c);
        strip.upload(mine.data());
        strip.warm_up();
        strip();
        strip.get_params().iteration_offset = 34;
        strip.get_params().n_iterations = 4;
        strip();
        strip.download(box->ramp_result + strip.first_row() * W);
        if (n_ranks == 1) { // the single strip is hip::StencilUpdate on the whole grid
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
            if (std::memcmp(ac.get_pointer(), box->ramp_result, H * W * sizeof(float)) != 0) {
                std::fprintf(stderr, "Ramp: the single strip differs from hip::StencilUpdate\n");
                return 1;
            }
        }
    }
    { // Conduction on per-field planes (split cell structure): scatter / gather at the strip's edge, two planes exchanged
        using Strip = hip::StripUpdate<Conduction, true>;
        Strip strip({.transition_function = Conduction{}, .halo_value = Plate{2.0f, 0.125f}, .n_iterations = 37,
                     .blocking = true},
                    H, W, rank, n_ranks, nullptr, callback, &ctx);
        std::vector<Plate> mine(strip.n_cells());
        for (std::size_t r = strip.first_row(); r < strip.end_row(); r++)
            for (std::size_t c = 0; c < W; c++)
                mine[(r - strip.first_row()) * W + c] = plate_cell(r, c);
        strip.upload(mine.data());
        strip.warm_up();
        strip();
        strip.download(box->plate_result + strip.first_row() * W);
        if (n_ranks == 1) {
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
            if (std::memcmp(ac.get_pointer(), box->plate_result, H * W * sizeof(Plate)) != 0) {
                std::fprintf(stderr, "Conduction: the single strip differs from hip::StencilUpdate\n");
                return 1;
            }
        }
    }
    std::printf("strip_template_test: rank %d of %d done\n", rank, n_ranks);
    return 0;
}
