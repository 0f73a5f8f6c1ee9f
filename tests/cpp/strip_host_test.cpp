// A plain C++ host for the native row-strip driver (ststhip_strip_*): no Python, no torch.
//
// Two (or three) processes are forked BEFORE anything touches the GPU; each owns one strip of a Jacobi5General grid
// on cuda:0 and advances it with ststhip_strip_advance.  RCCL cannot join two ranks on one device, so the ghost rows
// go through the exchange callback of ststhip_strip_create: a mailbox in shared memory (mmap'ed before the fork) with
// sequence counters.  The parent then runs the whole grid as a single strip and compares the pieces bit for bit.
// On a box with one GPU per rank the same host passes a communicator from ststhip_comm_create instead of the callback.
//
// usage: strip_host_test [ranks = 2] ; exit code 0 = identical.  Built by tests/cpp/Makefile (g++, links libststhip).
#include <ststhip.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>

namespace {

constexpr std::size_t H = 1500, W = 1100, max_rows = 128; // ghost rows of one exchange: up to four launches of 16 rows
constexpr std::uint64_t generations_a = 29, generations_b = 11;

#define CHECK(call)                                                                                 \
    do {                                                                                            \
        int rc_ = (call);                                                                           \
        if (rc_ != STSTHIP_OK) {                                                                    \
            std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ststhip_last_error());         \
            std::exit(2);                                                                           \
        }                                                                                           \
    } while (0)

// one direction of one boundary: rows from the rank above to the rank below, or back
struct Slot {
    std::atomic<std::uint64_t> written, read;
    float rows[max_rows * W];
};
struct Mailbox {
    Slot down[8], up[8]; // down[b]: rank b -> rank b+1; up[b]: rank b+1 -> rank b
    float result[H * W];
};

struct ExchangeContext {
    Mailbox *box;
    int rank, n_ranks;
    std::uint64_t round;
};

void put(Slot &slot, std::uint64_t round, const void *device_rows, std::size_t bytes, ststhip_stream stream) {
    while (slot.read.load(std::memory_order_acquire) != round) // the previous message has been taken
        usleep(50);
    CHECK(ststhip_memcpy_d2h(slot.rows, device_rows, bytes, stream));
    CHECK(ststhip_stream_synchronize(stream));
    slot.written.store(round + 1, std::memory_order_release);
}
void take(Slot &slot, std::uint64_t round, void *device_rows, std::size_t bytes, ststhip_stream stream) {
    while (slot.written.load(std::memory_order_acquire) != round + 1)
        usleep(50);
    CHECK(ststhip_memcpy_h2d(device_rows, slot.rows, bytes, stream));
    CHECK(ststhip_stream_synchronize(stream));
    slot.read.store(round + 1, std::memory_order_release);
}

// the contract of ststhip_comm_exchange_rows, staged through the mailbox
int exchange(void *ctx, int n_planes, const void *const *send_up, const void *const *send_down, void *const *recv_up,
             void *const *recv_down, const size_t *row_bytes, size_t n_rows, ststhip_stream stream) {
    ExchangeContext *c = static_cast<ExchangeContext *>(ctx);
    if (n_planes != 1 || n_rows > max_rows)
        return STSTHIP_ERR_INVALID;
    const std::size_t bytes = row_bytes[0] * n_rows;
    CHECK(ststhip_stream_synchronize(stream)); // the rows to send are complete
    if (c->rank > 0)
        put(c->box->up[c->rank - 1], c->round, send_up[0], bytes, stream);
    if (c->rank + 1 < c->n_ranks)
        put(c->box->down[c->rank], c->round, send_down[0], bytes, stream);
    if (c->rank > 0)
        take(c->box->down[c->rank - 1], c->round, recv_up[0], bytes, stream);
    if (c->rank + 1 < c->n_ranks)
        take(c->box->up[c->rank], c->round, recv_down[0], bytes, stream);
    c->round++;
    return STSTHIP_OK;
}

float cell(std::size_t r, std::size_t c) { return float((r * 131 + c * 71) % 257) / 257.0f; }

void run_rank(Mailbox *box, int rank, int n_ranks, float *out_rows) {
    CHECK(ststhip_init(0));
    ststhip_jacobi_params params = {};
    const float coef[5] = {0.2f, 0.21f, 0.19f, 0.22f, 0.18f};
    std::memcpy(params.coef, coef, sizeof coef);
    const float halo = 0.25f;
    ExchangeContext ctx{box, rank, n_ranks, 0};
    ststhip_strip strip = nullptr;
    CHECK(ststhip_strip_create("jacobi5general", &params, &halo, H, W, rank, n_ranks, nullptr,
                               n_ranks > 1 ? &exchange : nullptr, &ctx, &strip));
    std::uint64_t a = 0, b = 0;
    CHECK(ststhip_strip_rows(strip, &a, &b));
    std::vector<float> mine((b - a) * W);
    for (std::uint64_t r = a; r < b; r++)
        for (std::size_t c = 0; c < W; c++)
            mine[(r - a) * W + c] = cell(r, c);
    void *owned = nullptr;
    std::size_t row_bytes = 0;
    ststhip_stream stream = nullptr;
    CHECK(ststhip_strip_stream(strip, &stream));
    CHECK(ststhip_strip_plane(strip, 0, &owned, &row_bytes));
    CHECK(ststhip_memcpy_h2d(owned, mine.data(), mine.size() * 4, stream));
    CHECK(ststhip_strip_synchronize(strip));
    CHECK(ststhip_strip_warm_up(strip));
    CHECK(ststhip_strip_advance(strip, 0, generations_a, 0));
    CHECK(ststhip_strip_advance(strip, generations_a, generations_b, 1));
    CHECK(ststhip_strip_plane(strip, 0, &owned, &row_bytes)); // the current buffer set has changed
    CHECK(ststhip_memcpy_d2h(out_rows + a * W, owned, (b - a) * W * 4, stream));
    CHECK(ststhip_strip_synchronize(strip));
    CHECK(ststhip_strip_destroy(strip));
}

} // namespace

int main(int argc, char **argv) {
    const int n_ranks = argc > 1 ? std::atoi(argv[1]) : 2;
    if (n_ranks < 2 || n_ranks > 8) {
        std::fprintf(stderr, "usage: strip_host_test [ranks 2..8]\n");
        return 2;
    }
    Mailbox *box = static_cast<Mailbox *>(
        mmap(nullptr, sizeof(Mailbox), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0));
    if (box == MAP_FAILED)
        return 2;
    std::memset(static_cast<void *>(box), 0, sizeof(Mailbox));
    std::vector<pid_t> children;
    for (int rank = 0; rank < n_ranks; rank++) { // fork first: no process may carry an initialised GPU over a fork
        pid_t pid = fork();
        if (pid == 0) {
            run_rank(box, rank, n_ranks, box->result);
            _exit(0);
        }
        children.push_back(pid);
    }
    bool ok = true;
    for (pid_t pid : children) {
        int status = 0;
        waitpid(pid, &status, 0);
        ok = ok && WIFEXITED(status) && WEXITSTATUS(status) == 0;
    }
    if (!ok) {
        std::fprintf(stderr, "a rank failed\n");
        return 2;
    }
    // the same grid as one strip, in a further child (the parent never initialises the GPU)
    Mailbox *reference_box = static_cast<Mailbox *>(
        mmap(nullptr, sizeof(Mailbox), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0));
    pid_t pid = fork();
    if (pid == 0) {
        run_rank(reference_box, 0, 1, reference_box->result);
        _exit(0);
    }
    int status = 0;
    waitpid(pid, &status, 0);
    if (!(WIFEXITED(status) && WEXITSTATUS(status) == 0))
        return 2;
    std::size_t differing = 0;
    for (std::size_t i = 0; i < H * W; i++)
        differing += std::memcmp(&box->result[i], &reference_box->result[i], 4) != 0;
    std::printf("strip_host_test: %d strips of a %zu x %zu grid, %llu + %llu generations: %zu cells differ from the "
                "single-strip run\n",
                n_ranks, H, W, (unsigned long long)generations_a, (unsigned long long)generations_b, differing);
    return differing == 0 ? 0 : 1;
}
