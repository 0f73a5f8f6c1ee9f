// libststhip.so -- C-ABI runtime of the MI355X StencilUpdate backend (include/ststhip.h).
//
// Layer 0: device selection, one runtime stream, a size-bucketed HBM pool, pinned host memory,
// async copies, events, kernel launch, LDS-staged AoS<->planes transforms, RCCL ghost-row exchange.
// Layer 1: registry and drivers for the precompiled transition functions (app_*.hip).
#include "app_registry.hpp"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

namespace ststhip_detail {

static thread_local std::string g_last_error;

void set_error(const char *message) { g_last_error = message ? message : ""; }
int fail(int status, const char *message) {
    set_error(message);
    return status;
}

static int hip_fail(hipError_t err, const char *what) {
    std::string msg = std::string(what) + ": " + hipGetErrorString(err);
    g_last_error = msg;
    return STSTHIP_ERR_HIP;
}

// ------------------------------------------------------------------ options (environment, read once)
// Has everything recorded into `event` finished?  (hipEventQuery's "not ready" is an ERROR of the calling thread as far
// as hipGetLastError is concerned: left in place, the next kernel launch that checks hipGetLastError reports it as its own
// -- seen with several blocks as threads of one process, where a pooled buffer's release event is still pending.)
static bool event_done(hipEvent_t event) {
    if (hipEventQuery(event) == hipSuccess)
        return true;
    (void)hipGetLastError();
    return false;
}

static int env_int(const char *name, int fallback) {
    const char *v = std::getenv(name);
    return (v && *v) ? std::atoi(v) : fallback;
}
static ststhip_options read_options() {
    ststhip_options o;
    std::memset(&o, 0, sizeof o);
    o.chunk_rows = env_int("STSTHIP_CHUNK_ROWS", 0);
    o.tail_permille = env_int("STSTHIP_TAIL_PERMILLE", 0);
    o.n_taper = -1;
    if (const char *spec = std::getenv("STSTHIP_TAPER")) {
        // "permille:split[,permille:split...]" (Sweep.hpp, plan_tiers); an empty string switches the taper off
        std::string text = spec;
        o.n_taper = 0;
        std::size_t at = 0;
        while (at < text.size() && o.n_taper < 3) {
            const std::size_t colon = text.find(':', at);
            if (colon == std::string::npos)
                break;
            o.taper_permille[o.n_taper] = std::atoi(text.c_str() + at);
            o.taper_split[o.n_taper] = std::atoi(text.c_str() + colon + 1);
            o.n_taper++;
            const std::size_t comma = text.find(',', colon);
            at = comma == std::string::npos ? text.size() : comma + 1;
        }
    }
    o.narrow_form_kcells = env_int("STSTHIP_NARROW_FORM_KCELLS", 6000);
    o.narrow_band_rows = env_int("STSTHIP_NARROW_BAND_ROWS", 0);
    o.skip_constant_stores = env_int("STSTHIP_SKIP_CONSTANT_STORES", 1);
    o.xcd_remap = env_int("STSTHIP_XCD_REMAP", 0);
    o.last_chunk_early = env_int("STSTHIP_LAST_CHUNK_EARLY", 1);
    o.max_generations = env_int("STSTHIP_MAX_GENERATIONS", 0);
    o.allow_spilling_depths = env_int("STSTHIP_ALLOW_SPILLING_DEPTHS", 0);
    o.virtual_strips = env_int("STSTHIP_VIRTUAL_STRIPS", 0);
    o.two_strips_permille = env_int("STSTHIP_TWO_STRIPS_PERMILLE", 1220);
    o.two_strips_permille_outer = env_int("STSTHIP_TWO_STRIPS_PERMILLE", 1500);
    o.strip_skew_permille = env_int("STSTHIP_STRIP_SKEW_PERMILLE", 0);
    o.bands_beside_interior = env_int("STSTHIP_BANDS_BESIDE_INTERIOR", 1);
    o.band_stream_priority = env_int("STSTHIP_BAND_STREAM_PRIORITY", 1);
    o.bands_apart = env_int("STSTHIP_BANDS_APART", 0);
    o.bands_one_launch = env_int("STSTHIP_BANDS_ONE_LAUNCH", 1);
    o.comm_stream_priority = env_int("STSTHIP_COMM_STREAM_PRIORITY", 0);
    o.jacobi_fastpath = env_int("STSTHIP_JACOBI_FASTPATH", 1);
    o.conway_fastpath = env_int("STSTHIP_CONWAY_FASTPATH", 1);
    o.prepare_streams = env_int("STSTHIP_PREPARE_STREAMS", 1);
    o.host_cache_mib = env_int("STSTHIP_HOST_CACHE_MIB", 4096);
    o.exchange_every = env_int("STSTHIP_EXCHANGE_EVERY", 0);
    o.tune_depth = env_int("STSTHIP_TUNE_DEPTH", 1);
    o.stream_upload = env_int("STSTHIP_STREAM_UPLOAD", 1);
    o.skewed_strips = env_int("STSTHIP_SKEWED_STRIPS", 1);
    o.strip_substrips = env_int("STSTHIP_STRIP_SUBSTRIPS", -1);
    o.upload_block_mib = env_int("STSTHIP_UPLOAD_BLOCK_MIB", 0);
    return o;
}
static ststhip_options &options_storage() {
    static ststhip_options o = read_options();
    return o;
}
static const ststhip_options &opt() { return options_storage(); }

#define HIP_TRY(call)                                                                              \
    do {                                                                                           \
        hipError_t err_ = (call);                                                                  \
        if (err_ != hipSuccess)                                                                    \
            return hip_fail(err_, #call);                                                          \
    } while (0)

struct Runtime {
    std::mutex lock;
    bool up = false;
    int device = -1;
    int compute_units = 256;
    hipStream_t stream = nullptr;
    // pool: bucket size -> free blocks; live: ptr -> bucket size.  A free block remembers the stream it was
    // released on and an event recorded there: work queued on that stream before the release may still use
    // the block, so another stream (or the host) may only have it after the event.
    struct FreeBlock {
        void *ptr;
        hipStream_t released_on;
        hipEvent_t released; // nullptr: nothing in flight
    };
    std::multimap<std::size_t, FreeBlock> free_blocks;
    std::map<void *, std::size_t> live_blocks;
    // the same for pinned host memory (pinning hundreds of MiB costs tens of ms per allocation, and every
    // grid a StencilUpdate returns gets a host mirror as soon as the application looks at it)
    std::multimap<std::size_t, void *> free_host_blocks;
    std::map<void *, std::size_t> live_host_blocks;
    std::size_t cached_host_bytes = 0;
};
// free pinned blocks kept for reuse, at most (pinned memory is taken from the host's RAM)
static std::size_t host_cache_limit() {
    return std::size_t(opt().host_cache_mib) << 20;
}
static Runtime &rt() {
    static Runtime r;
    return r;
}

// Events of the drivers (hipEventDisableTiming), recycled: a 1000-generation call records about 500, and creating and
// destroying each one showed in the launch gaps of small grids.  A driver call takes events from the process-wide free
// list and hands them all back when it returns: by then every wait on them has been ENQUEUED (hipStreamWaitEvent
// captures the record it waits for at enqueue time), so recording them again in a later call is safe.
struct EventPool {
    std::vector<hipEvent_t> taken;
    static std::vector<hipEvent_t> &free_list() {
        static std::vector<hipEvent_t> list;
        return list;
    }
    hipEvent_t take() {
        hipEvent_t ev = nullptr;
        {
            std::lock_guard<std::mutex> guard(rt().lock);
            if (!free_list().empty()) {
                ev = free_list().back();
                free_list().pop_back();
            }
        }
        if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
            return nullptr;
        taken.push_back(ev);
        return ev;
    }
    ~EventPool() {
        std::lock_guard<std::mutex> guard(rt().lock);
        for (hipEvent_t ev : taken)
            free_list().push_back(ev);
    }
};

// Streams of the pass driver beside the caller's own: side streams (further row strips advancing concurrently) and
// highest-priority band streams (a band's few waves then get the wave slots that free up first instead of queueing
// behind the pending workgroups of an interior launch).  ONE set for the whole process, shared by every caller stream
// and host thread: the orderings a call needs are events, so calls that overlap merely serialise on these streams.
// HIP deals streams onto a few hardware queues in creation order, idle ones included, and launches that share a
// queue serialise -- a set per caller stream (round 2) multiplied idle streams and made a host that brings its own
// stream collide with the set prepared for the runtime's stream (bench 6500 -> 4280, profiles/r02_short_runs.txt).
// With one set there is nothing to multiply, so it is created AND first used at ststhip_init (a stream's first use
// costs milliseconds, 15-20 ms for the three, which would otherwise land in the first update call).
static std::vector<hipStream_t> &shared_side_streams() {
    static std::vector<hipStream_t> streams;
    return streams;
}
static std::vector<hipStream_t> &shared_band_streams() {
    static std::vector<hipStream_t> streams;
    return streams;
}
static hipError_t create_band_stream(hipStream_t *stream) {
    if (!opt().band_stream_priority) // A/B: bands on normal-priority streams
        return hipStreamCreateWithFlags(stream, hipStreamNonBlocking);
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess)
        greatest = 0;
    return hipStreamCreateWithPriority(stream, hipStreamNonBlocking, greatest);
}
// caller holds the runtime lock
static bool grow_shared_streams(int n_side, int n_band) {
    auto &side = shared_side_streams();
    auto &band = shared_band_streams();
    while (int(side.size()) < n_side) {
        hipStream_t extra;
        if (hipStreamCreateWithFlags(&extra, hipStreamNonBlocking) != hipSuccess)
            return false;
        side.push_back(extra);
    }
    while (int(band.size()) < n_band) {
        hipStream_t extra;
        if (create_band_stream(&extra) != hipSuccess)
            return false;
        band.push_back(extra);
    }
    return true;
}
static bool side_streams_for(hipStream_t, int n, std::vector<hipStream_t> &out) {
    Runtime &r = rt();
    std::lock_guard<std::mutex> guard(r.lock);
    if (!grow_shared_streams(n, 0))
        return false;
    out.assign(shared_side_streams().begin(), shared_side_streams().begin() + n);
    return true;
}
static bool band_streams_for(hipStream_t, int n, std::vector<hipStream_t> &out) {
    Runtime &r = rt();
    std::lock_guard<std::mutex> guard(r.lock);
    if (!grow_shared_streams(0, n))
        return false;
    out.assign(shared_band_streams().begin(), shared_band_streams().begin() + n);
    return true;
}

// Creates the shared streams (one side stream, two band streams) and uses each once: a stream's first use costs
// milliseconds -- 25 ms for the three, which would otherwise land in the first update call (a one-shot 1000-generation
// run of the unchanged jacobi example: Walltime 0.123 -> 0.097 s, hotspot 0.098 -> 0.068 s; profiles/r03_short_runs.txt).
// Called when the host first asks for the runtime's own stream or for pinned host memory, i.e. when it is a host
// that runs on the runtime's stream (the C++ templates: a Grid's host mirror is allocated before the first update).
// NOT at ststhip_init: for a host that creates its own stream afterwards (torch) the three would sit in front of
// it in the hardware queues' dealing order, and its sweeps measured 2-11 % slower (bench.py 6900 -> 6200).
static void prepare_shared_streams_once() {
    static bool done = false;
    Runtime &r = rt();
    std::lock_guard<std::mutex> guard(r.lock);
    if (done || !r.up || !opt().prepare_streams)
        return;
    done = true;
    if (!grow_shared_streams(1, 2))
        return;
    std::vector<hipStream_t> made(shared_side_streams());
    made.insert(made.end(), shared_band_streams().begin(), shared_band_streams().end());
    made.push_back(r.stream);
    void *word = nullptr;
    if (hipMalloc(&word, 256) == hipSuccess) {
        for (hipStream_t st : made)
            (void)hipMemsetAsync(word, 0, 4, st);
        for (hipStream_t st : made)
            (void)hipStreamSynchronize(st);
        (void)hipFree(word);
    }
    // ... and the copy engines: the first copy of a megabyte or more between pinned host memory and the device takes
    // 7.4 ms longer than any later one, whatever its size and buffers (tools/debug/malloc_time.py: 1 GiB 25.9 ms, then
    // 18.7 ms; a 4 KiB copy does not take that path).  One megabyte each way on the runtime's stream, here, instead of
    // inside the first grid's upload and the first result's download.
    constexpr std::size_t probe = 1 << 20;
    void *host = nullptr, *device = nullptr;
    if (hipHostMalloc(&host, probe, hipHostMallocDefault) == hipSuccess && hipMalloc(&device, probe) == hipSuccess) {
        std::memset(host, 0, probe);
        (void)hipMemcpyAsync(device, host, probe, hipMemcpyHostToDevice, r.stream);
        (void)hipMemcpyAsync(host, device, probe, hipMemcpyDeviceToHost, r.stream);
        (void)hipStreamSynchronize(r.stream);
    }
    if (device)
        (void)hipFree(device);
    if (host)
        (void)hipHostFree(host);
}

static std::vector<AppEntry> &apps() {
    static std::vector<AppEntry> registry;
    return registry;
}
void register_app(AppEntry const &entry) { apps().push_back(entry); }
const AppEntry *find_app(const char *name) {
    if (!name)
        return nullptr;
    for (auto const &e : apps())
        if (std::strcmp(e.info.name, name) == 0)
            return &e;
    return nullptr;
}

// hipFree of every pooled block, each after the work that was queued before its release
static void release_free_blocks(Runtime &r) {
    for (auto &kv : r.free_blocks) {
        if (kv.second.released) {
            (void)hipEventSynchronize(kv.second.released);
            (void)hipEventDestroy(kv.second.released);
        }
        (void)hipFree(kv.second.ptr);
    }
    r.free_blocks.clear();
}

static hipStream_t resolve(ststhip_stream s) { return s ? static_cast<hipStream_t>(s) : rt().stream; }

static std::size_t bucket_of(std::size_t bytes) {
    // powers of two up to 1 MiB, then multiples of 2 MiB (HBM is plentiful: 288 GB)
    if (bytes <= (1u << 20)) {
        std::size_t b = 256;
        while (b < bytes)
            b <<= 1;
        return b;
    }
    const std::size_t step = std::size_t(2) << 20;
    return (bytes + step - 1) / step * step;
}

// ------------------------------------------------------------------ AoS <-> planes kernels
struct FieldTable {
    int n_fields;
    unsigned offset[16];
    unsigned size[16];
    void *plane[16];
};

constexpr int transform_tile = 256; // cells per workgroup

template <typename W>
__device__ inline void copy_field(unsigned char *dst, const unsigned char *src) {
    *reinterpret_cast<W *>(dst) = *reinterpret_cast<const W *>(src);
}

__device__ inline void copy_bytes(unsigned char *dst, const unsigned char *src, unsigned size) {
    if ((size & 3u) == 0 && ((reinterpret_cast<std::uintptr_t>(dst) |
                              reinterpret_cast<std::uintptr_t>(src)) & 3u) == 0) {
        for (unsigned b = 0; b < size; b += 4)
            copy_field<std::uint32_t>(dst + b, src + b);
    } else {
        for (unsigned b = 0; b < size; b++)
            dst[b] = src[b];
    }
}

// AoS tile -> LDS with whole-line loads, then each lane peels one cell's fields into the planes.
__global__ void __launch_bounds__(transform_tile)
    scatter_kernel(const unsigned char *aos, unsigned cell_size, std::size_t n_cells, FieldTable table) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tile[];
    const std::size_t first = std::size_t(blockIdx.x) * transform_tile;
    const std::size_t cells_here = n_cells - first < transform_tile ? n_cells - first : transform_tile;
    const std::size_t tile_bytes = cells_here * cell_size;
    const unsigned char *src = aos + first * cell_size;
    if ((cell_size & 3u) == 0) {
        for (std::size_t w = threadIdx.x; w < tile_bytes / 4; w += transform_tile)
            reinterpret_cast<std::uint32_t *>(tile)[w] = reinterpret_cast<const std::uint32_t *>(src)[w];
    } else {
        for (std::size_t b = threadIdx.x; b < tile_bytes; b += transform_tile)
            tile[b] = src[b];
    }
    __syncthreads();
    if (threadIdx.x < cells_here) {
        const unsigned char *cell = tile + std::size_t(threadIdx.x) * cell_size;
        for (int f = 0; f < table.n_fields; f++) {
            unsigned char *out = static_cast<unsigned char *>(table.plane[f]) +
                                 (first + threadIdx.x) * table.size[f];
            copy_bytes(out, cell + table.offset[f], table.size[f]);
        }
    }
}

__global__ void __launch_bounds__(transform_tile)
    gather_kernel(unsigned char *aos, unsigned cell_size, std::size_t n_cells, FieldTable table) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tile[];
    const std::size_t first = std::size_t(blockIdx.x) * transform_tile;
    const std::size_t cells_here = n_cells - first < transform_tile ? n_cells - first : transform_tile;
    const std::size_t tile_bytes = cells_here * cell_size;
    // bytes of the cell that no field covers stay zero (the reference gathers into a
    // default-constructed cell, cuda/StencilUpdate.hpp:424)
    for (std::size_t b = threadIdx.x; b < tile_bytes; b += transform_tile)
        tile[b] = 0;
    __syncthreads();
    if (threadIdx.x < cells_here) {
        unsigned char *cell = tile + std::size_t(threadIdx.x) * cell_size;
        for (int f = 0; f < table.n_fields; f++) {
            const unsigned char *in = static_cast<const unsigned char *>(table.plane[f]) +
                                      (first + threadIdx.x) * table.size[f];
            copy_bytes(cell + table.offset[f], in, table.size[f]);
        }
    }
    __syncthreads();
    unsigned char *dst = aos + first * cell_size;
    if ((cell_size & 3u) == 0) {
        for (std::size_t w = threadIdx.x; w < tile_bytes / 4; w += transform_tile)
            reinterpret_cast<std::uint32_t *>(dst)[w] = reinterpret_cast<const std::uint32_t *>(tile)[w];
    } else {
        for (std::size_t b = threadIdx.x; b < tile_bytes; b += transform_tile)
            dst[b] = tile[b];
    }
}

// ------------------------------------------------------------------ max |field| reduction
struct ReduceTable {
    int n_fields;
    unsigned offset[8];
    unsigned type[8];
    unsigned long long row_limit[8], col_limit[8];
};

// A wave reduces its 64 cells per field with DPP-based shuffles and adds one atomic max; |x| >= 0, so the bit
// patterns of the doubles order like the values and an unsigned 64-bit max is exact.
__global__ void __launch_bounds__(256)
    reduce_max_abs_kernel(const unsigned char *cells, unsigned cell_size, unsigned long long height,
                          unsigned long long width, unsigned long long pitch, ReduceTable table,
                          unsigned long long *acc) {
    double best[8];
    for (int f = 0; f < 8; f++)
        best[f] = 0.0;
    const unsigned long long n = height * width;
    for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < n; i += 256ull * gridDim.x) {
        const unsigned long long r = i / width, c = i - r * width;
        const unsigned char *cell = cells + (r * pitch + c) * cell_size;
        for (int f = 0; f < table.n_fields; f++) {
            if (r < table.row_limit[f] && c < table.col_limit[f]) {
                double v;
                if (table.type[f] == STSTHIP_F64)
                    v = fabs(*reinterpret_cast<const double *>(cell + table.offset[f]));
                else
                    v = double(fabsf(*reinterpret_cast<const float *>(cell + table.offset[f])));
                if (v > best[f]) // false for NaN
                    best[f] = v;
            }
        }
    }
    for (int f = 0; f < table.n_fields; f++) {
        double v = best[f];
        for (int delta = 32; delta >= 1; delta >>= 1) {
            const double other = __shfl_xor(v, delta, 64);
            v = other > v ? other : v;
        }
        if ((threadIdx.x & 63) == 0 && v > 0.0)
            atomicMax(acc + f, static_cast<unsigned long long>(__double_as_longlong(v)));
    }
}

static int fill_table(FieldTable &t, std::size_t cell_size, int n_fields, const size_t *offset,
                      const size_t *size, void *const *planes) {
    if (n_fields < 1 || n_fields > 16 || cell_size == 0 || cell_size > 256)
        return fail(STSTHIP_ERR_INVALID, "scatter/gather: need 1..16 fields and cells of 1..256 bytes");
    t.n_fields = n_fields;
    for (int f = 0; f < n_fields; f++) {
        if (offset[f] + size[f] > cell_size || size[f] == 0)
            return fail(STSTHIP_ERR_INVALID, "scatter/gather: field outside the cell");
        t.offset[f] = unsigned(offset[f]);
        t.size[f] = unsigned(size[f]);
        t.plane[f] = planes[f];
    }
    return STSTHIP_OK;
}

// ------------------------------------------------------------------ RCCL, loaded on first use
struct Id128 {
    char bytes[128];
};
struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Id128 /* ncclUniqueId, passed by value */, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
static Rccl &rccl() {
    static Rccl r;
    return r;
}
static int load_rccl() {
    Rccl &r = rccl();
    if (r.handle)
        return STSTHIP_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (r.handle)
            break;
    }
    if (!r.handle)
        return fail(STSTHIP_ERR_COMM, "cannot load librccl.so");
    bool ok = true;
    auto sym = [&](const char *name) {
        void *p = dlsym(r.handle, name);
        ok = ok && p;
        return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok)
        return fail(STSTHIP_ERR_COMM, "librccl.so lacks a required symbol");
    return STSTHIP_OK;
}
static int nccl_fail(int code, const char *what) {
    std::string msg = std::string(what) + ": " +
                      (rccl().GetErrorString ? rccl().GetErrorString(code) : "rccl error");
    g_last_error = msg;
    return STSTHIP_ERR_COMM;
}
#define NCCL_TRY(call)                                                                             \
    do {                                                                                           \
        int rc_ = (call);                                                                          \
        if (rc_ != 0)                                                                              \
            return nccl_fail(rc_, #call);                                                          \
    } while (0)

struct Comm {
    void *nccl = nullptr;
    int rank = 0, n_ranks = 1;
    int up = -1, down = -1; // ranks the ghost rows are exchanged with (-1: none); a chain of strips by default
    int left = -1, right = -1; // ranks the ghost columns of a 2-D block are exchanged with (-1: none)
};

} // namespace ststhip_detail

using namespace ststhip_detail;

extern "C" {

int ststhip_abi_version(void) { return STSTHIP_ABI_VERSION; }
const char *ststhip_last_error(void) { return g_last_error.c_str(); }
void ststhip_set_last_error(const char *message) { set_error(message); }

const ststhip_options *ststhip_get_options(void) { return &options_storage(); }
int ststhip_reload_options(void) {
    options_storage() = read_options();
    return STSTHIP_OK;
}

int ststhip_init(int device) {
    Runtime &r = rt();
    std::lock_guard<std::mutex> guard(r.lock);
    if (r.up && (device < 0 || device == r.device))
        return STSTHIP_OK;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return fail(STSTHIP_ERR_NO_DEVICE, "no HIP device visible: the MI355X backend has no CPU fallback");
    if (device >= count)
        return fail(STSTHIP_ERR_INVALID, "device index out of range");
    if (r.up && device != r.device)
        return fail(STSTHIP_ERR_INVALID, "runtime already bound to another device of this process");
    if (device >= 0)
        HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipGetDevice(&r.device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, r.device));
    r.compute_units = prop.multiProcessorCount;
    HIP_TRY(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
    r.up = true;
    return STSTHIP_OK;
}

int ststhip_shutdown(void) {
    Runtime &r = rt();
    std::lock_guard<std::mutex> guard(r.lock);
    if (!r.up)
        return STSTHIP_OK;
    (void)hipStreamSynchronize(r.stream);
    for (auto *pool : {&shared_side_streams(), &shared_band_streams()}) {
        for (hipStream_t extra : *pool) {
            (void)hipStreamSynchronize(extra);
            (void)hipStreamDestroy(extra);
        }
        pool->clear();
    }
    for (hipEvent_t ev : EventPool::free_list())
        (void)hipEventDestroy(ev);
    EventPool::free_list().clear();
    release_free_blocks(r);
    for (auto &kv : r.free_host_blocks)
        (void)hipHostFree(kv.second);
    r.free_host_blocks.clear();
    r.cached_host_bytes = 0;
    (void)hipStreamDestroy(r.stream);
    r.stream = nullptr;
    r.up = false;
    return STSTHIP_OK;
}

int ststhip_device_count(int *count) {
    if (!count)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    if (hipGetDeviceCount(count) != hipSuccess)
        *count = 0;
    return STSTHIP_OK;
}

int ststhip_device_name(char *buf, size_t buf_size) {
    if (!buf || buf_size == 0)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    std::snprintf(buf, buf_size, "%s (%s)", prop.name, prop.gcnArchName);
    return STSTHIP_OK;
}

int ststhip_compute_units(int *count) {
    if (!count)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    *count = rt().compute_units;
    return STSTHIP_OK;
}

// Takes a free block of the bucket for use on `stream` (or by the host when for_host).  A block released on
// the same stream is reusable at once (stream order); one released elsewhere makes `stream` wait for the
// release event first (the host: blocks on it), unless the event has completed already.
static int take_free_block(std::size_t bucket, hipStream_t stream, bool for_host, void **ptr) {
    Runtime &r = rt();
    Runtime::FreeBlock block{nullptr, nullptr, nullptr};
    {
        std::lock_guard<std::mutex> guard(r.lock);
        auto range = r.free_blocks.equal_range(bucket);
        auto pick = r.free_blocks.end();
        for (auto it = range.first; it != range.second; ++it) {
            Runtime::FreeBlock &b = it->second;
            if (b.released && event_done(b.released)) {
                (void)hipEventDestroy(b.released);
                b.released = nullptr;
            }
            if (!b.released || (!for_host && b.released_on == stream)) {
                pick = it;
                break;
            }
            if (pick == r.free_blocks.end())
                pick = it;
        }
        if (pick == r.free_blocks.end())
            return -1;
        block = pick->second;
        r.free_blocks.erase(pick);
        r.live_blocks[block.ptr] = bucket;
    }
    if (block.released) {
        hipError_t err = hipSuccess;
        if (for_host)
            err = hipEventSynchronize(block.released);
        else if (block.released_on != stream)
            err = hipStreamWaitEvent(stream, block.released, 0);
        (void)hipEventDestroy(block.released);
        if (err != hipSuccess)
            return hip_fail(err, "waiting for a pooled block's release");
    }
    *ptr = block.ptr;
    return STSTHIP_OK;
}

static int pool_malloc(void **ptr, size_t bytes, hipStream_t stream, bool for_host) {
    if (!ptr)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    if (int rc = ststhip_init(-1))
        return rc;
    Runtime &r = rt();
    const std::size_t bucket = bucket_of(bytes ? bytes : 1);
    const int taken = take_free_block(bucket, for_host ? nullptr : (stream ? stream : r.stream), for_host, ptr);
    if (taken >= 0)
        return taken;
    void *p = nullptr;
    hipError_t err = hipMalloc(&p, bucket);
    if (err != hipSuccess) {
        ststhip_pool_trim();
        err = hipMalloc(&p, bucket);
    }
    if (err != hipSuccess)
        return hip_fail(err, "hipMalloc");
    std::lock_guard<std::mutex> guard(r.lock);
    r.live_blocks[p] = bucket;
    *ptr = p;
    return STSTHIP_OK;
}

static int pool_free(void *ptr, hipStream_t stream) {
    if (!ptr)
        return STSTHIP_OK;
    Runtime &r = rt();
    std::lock_guard<std::mutex> guard(r.lock);
    auto it = r.live_blocks.find(ptr);
    if (it == r.live_blocks.end())
        return fail(STSTHIP_ERR_INVALID, "ststhip_free: pointer not from ststhip_malloc");
    if (r.up) {
        Runtime::FreeBlock block{ptr, stream ? stream : r.stream, nullptr};
        // if the event cannot be made the block is simply kept out of circulation until the stream is idle
        if (hipEventCreateWithFlags(&block.released, hipEventDisableTiming) == hipSuccess) {
            if (hipEventRecord(block.released, block.released_on) != hipSuccess) {
                (void)hipEventDestroy(block.released);
                block.released = nullptr;
                (void)hipStreamSynchronize(block.released_on);
            }
        } else {
            block.released = nullptr;
            (void)hipStreamSynchronize(block.released_on);
        }
        r.free_blocks.emplace(it->second, block);
    }
    r.live_blocks.erase(it);
    return STSTHIP_OK;
}

int ststhip_malloc(void **ptr, size_t bytes) { return pool_malloc(ptr, bytes, nullptr, /*for_host=*/true); }
int ststhip_malloc_async(void **ptr, size_t bytes, ststhip_stream stream) {
    return pool_malloc(ptr, bytes, static_cast<hipStream_t>(stream), /*for_host=*/false);
}
int ststhip_free(void *ptr) { return pool_free(ptr, nullptr); }
int ststhip_free_async(void *ptr, ststhip_stream stream) { return pool_free(ptr, static_cast<hipStream_t>(stream)); }

int ststhip_pool_trim(void) {
    Runtime &r = rt();
    std::lock_guard<std::mutex> guard(r.lock);
    release_free_blocks(r);
    for (auto &kv : r.free_host_blocks)
        (void)hipHostFree(kv.second);
    r.free_host_blocks.clear();
    r.cached_host_bytes = 0;
    return STSTHIP_OK;
}

int ststhip_host_malloc(void **ptr, size_t bytes) {
    if (!ptr)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    if (int rc = ststhip_init(-1))
        return rc;
    prepare_shared_streams_once();
    Runtime &r = rt();
    const std::size_t bucket = bucket_of(bytes ? bytes : 1);
    {
        std::lock_guard<std::mutex> guard(r.lock);
        auto it = r.free_host_blocks.find(bucket);
        if (it != r.free_host_blocks.end()) {
            *ptr = it->second;
            r.free_host_blocks.erase(it);
            r.cached_host_bytes -= bucket;
            r.live_host_blocks[*ptr] = bucket;
            return STSTHIP_OK;
        }
    }
    void *p = nullptr;
    hipError_t err = hipHostMalloc(&p, bucket, hipHostMallocDefault);
    if (err != hipSuccess) {
        ststhip_pool_trim();
        err = hipHostMalloc(&p, bucket, hipHostMallocDefault);
    }
    if (err != hipSuccess)
        return hip_fail(err, "hipHostMalloc");
    std::lock_guard<std::mutex> guard(r.lock);
    r.live_host_blocks[p] = bucket;
    *ptr = p;
    return STSTHIP_OK;
}

int ststhip_host_free(void *ptr) {
    if (!ptr)
        return STSTHIP_OK;
    Runtime &r = rt();
    std::size_t bucket = 0;
    {
        std::lock_guard<std::mutex> guard(r.lock);
        auto it = r.live_host_blocks.find(ptr);
        if (it == r.live_host_blocks.end())
            return fail(STSTHIP_ERR_INVALID, "ststhip_host_free: pointer not from ststhip_host_malloc");
        bucket = it->second;
        r.live_host_blocks.erase(it);
        if (!r.up)
            return STSTHIP_OK; // the HIP runtime is gone (static destruction order): nothing to release
        if (r.cached_host_bytes + bucket <= host_cache_limit()) {
            // Transfers are stream-ordered and the owners (hip::Grid) synchronise before the host reads or
            // releases a mirror, so a cached block has no copy in flight.
            r.free_host_blocks.emplace(bucket, ptr);
            r.cached_host_bytes += bucket;
            return STSTHIP_OK;
        }
    }
    HIP_TRY(hipHostFree(ptr));
    return STSTHIP_OK;
}

int ststhip_memcpy_h2d(void *dst, const void *src, size_t bytes, ststhip_stream stream) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, resolve(stream)));
    return STSTHIP_OK;
}
int ststhip_memcpy_d2h(void *dst, const void *src, size_t bytes, ststhip_stream stream) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, resolve(stream)));
    return STSTHIP_OK;
}
int ststhip_memcpy_d2d(void *dst, const void *src, size_t bytes, ststhip_stream stream) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, resolve(stream)));
    return STSTHIP_OK;
}
int ststhip_memset(void *dst, int value, size_t bytes, ststhip_stream stream) {
    HIP_TRY(hipMemsetAsync(dst, value, bytes, resolve(stream)));
    return STSTHIP_OK;
}

int ststhip_default_stream(ststhip_stream *stream) {
    if (!stream)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    if (int rc = ststhip_init(-1))
        return rc;
    prepare_shared_streams_once();
    *stream = rt().stream;
    return STSTHIP_OK;
}
int ststhip_stream_create(ststhip_stream *stream) {
    if (!stream)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    if (int rc = ststhip_init(-1))
        return rc;
    hipStream_t s;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return STSTHIP_OK;
}
int ststhip_stream_destroy(ststhip_stream stream) {
    if (stream)
        HIP_TRY(hipStreamDestroy(static_cast<hipStream_t>(stream)));
    return STSTHIP_OK;
}
int ststhip_stream_synchronize(ststhip_stream stream) {
    if (!rt().up && !stream)
        return STSTHIP_OK;
    HIP_TRY(hipStreamSynchronize(resolve(stream)));
    return STSTHIP_OK;
}
int ststhip_stream_wait_event(ststhip_stream stream, ststhip_event event) {
    HIP_TRY(hipStreamWaitEvent(resolve(stream), static_cast<hipEvent_t>(event), 0));
    return STSTHIP_OK;
}

int ststhip_event_create(ststhip_event *event) {
    if (!event)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    *event = e;
    return STSTHIP_OK;
}
int ststhip_event_destroy(ststhip_event event) {
    if (event)
        HIP_TRY(hipEventDestroy(static_cast<hipEvent_t>(event)));
    return STSTHIP_OK;
}
int ststhip_event_record(ststhip_event event, ststhip_stream stream) {
    HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(event), resolve(stream)));
    return STSTHIP_OK;
}
int ststhip_event_synchronize(ststhip_event event) {
    HIP_TRY(hipEventSynchronize(static_cast<hipEvent_t>(event)));
    return STSTHIP_OK;
}
int ststhip_event_elapsed_ms(ststhip_event start, ststhip_event stop, float *ms) {
    if (!ms)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    HIP_TRY(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
    return STSTHIP_OK;
}

int ststhip_launch(const void *function, unsigned grid_x, unsigned grid_y, unsigned grid_z,
                   unsigned block_x, unsigned block_y, unsigned block_z, void **args,
                   size_t shared_bytes, ststhip_stream stream) {
    if (!function || grid_x == 0 || grid_y == 0 || grid_z == 0)
        return fail(STSTHIP_ERR_INVALID, "ststhip_launch: empty grid or null kernel");
    if (int rc = ststhip_init(-1))
        return rc;
    HIP_TRY(hipLaunchKernel(function, dim3(grid_x, grid_y, grid_z), dim3(block_x, block_y, block_z),
                            args, shared_bytes, resolve(stream)));
    return STSTHIP_OK;
}

int ststhip_kernel_scratch_bytes(const void *function, size_t *bytes_per_work_item) {
    if (!function || !bytes_per_work_item)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    if (int rc = ststhip_init(-1))
        return rc;
    hipFuncAttributes attr;
    HIP_TRY(hipFuncGetAttributes(&attr, function));
    *bytes_per_work_item = attr.localSizeBytes;
    return STSTHIP_OK;
}

int ststhip_occupancy(const void *function, unsigned block_threads, size_t shared_bytes,
                      int *blocks_per_cu) {
    if (!function || !blocks_per_cu || block_threads == 0)
        return fail(STSTHIP_ERR_INVALID, "ststhip_occupancy: bad argument");
    if (int rc = ststhip_init(-1))
        return rc;
    int blocks = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, function, int(block_threads),
                                                         shared_bytes));
    *blocks_per_cu = blocks > 0 ? blocks : 1;
    return STSTHIP_OK;
}

static thread_local const void *g_tdv_table = nullptr;
static thread_local std::uint64_t g_tdv_first = 0, g_tdv_count = 0, g_tdv_size = 0;
int ststhip_current_tdv_table(const void **base, uint64_t *first_iteration, uint64_t *n_values,
                              uint64_t *value_size) {
    if (base)
        *base = g_tdv_table;
    if (first_iteration)
        *first_iteration = g_tdv_first;
    if (n_values)
        *n_values = g_tdv_count;
    if (value_size)
        *value_size = g_tdv_size;
    return STSTHIP_OK;
}

static thread_local int g_launch_concurrency = 1;
static thread_local int g_target_holds_constants = 0;
int ststhip_launch_concurrency(void) { return g_launch_concurrency; }
static thread_local std::uint64_t g_row_hole_begin = 0, g_row_hole_end = 0;
int ststhip_launch_row_hole(uint64_t *begin, uint64_t *end) {
    if (begin)
        *begin = g_row_hole_begin;
    if (end)
        *end = g_row_hole_end;
    return STSTHIP_OK;
}
int ststhip_set_launch_row_hole(uint64_t begin, uint64_t end) {
    if (begin > end)
        return fail(STSTHIP_ERR_INVALID, "row hole: begin after end");
    g_row_hole_begin = begin;
    g_row_hole_end = end;
    return STSTHIP_OK;
}
static thread_local std::uint64_t g_col_begin = 0, g_col_end = 0;
int ststhip_launch_columns(uint64_t *begin, uint64_t *end) {
    if (begin)
        *begin = g_col_begin;
    if (end)
        *end = g_col_end;
    return STSTHIP_OK;
}
int ststhip_set_launch_columns(uint64_t begin, uint64_t end) {
    if (end < begin)
        return fail(STSTHIP_ERR_INVALID, "bad column range");
    g_col_begin = begin;
    g_col_end = end;
    return STSTHIP_OK;
}
int ststhip_target_holds_constants(void) { return g_target_holds_constants; }
int ststhip_set_launch_concurrency(int n) {
    g_launch_concurrency = std::min(std::max(n, 1), 8);
    return STSTHIP_OK;
}

// Two row strips (two launches side by side, their boundary bands beside them on streams of their own) against one
// launch per pass: measured over Jacobi (both forms), HotSpot fp32 / fp64, FDTD and the Game of Life at 1024 ... 16384
// rows (profiles/r02_tune_strip_rule.txt), two strips win (+3 ... +18 %) when one launch of the whole grid would be
// 1.22 or more times the wave slots of the chip in the launcher's own chunk model, and lose below (-3 ... -18 %).
// A strip of a multi-GPU run that has neighbours sweeps two more bands per pass: there the second sub-strip pays
// only from 1.5 on (Jacobi 4096 x 16384, 1.24: one sub-strip 3990, two 4020; 8192 x 16384, 1.75: 4990 / 5350).
static int suggest_row_strips(std::uint64_t rows, std::uint64_t width, std::uint32_t strip_width,
                              std::uint64_t g_max, std::uint64_t n_passes, bool outer_bands = false) {
    int strips = opt().virtual_strips;
    if (strips <= 0) {
        strips = 1;
        if (n_passes >= 2 && strip_width > 0) {
            const double threshold = (outer_bands ? opt().two_strips_permille_outer : opt().two_strips_permille) / 1000.0;
            const double n_cols = std::ceil(double(width) / strip_width);
            const double slots = double(rt().compute_units) * 16.0; // ~4 workgroups of 4 waves per CU
            const double chunk = std::sqrt(double(rows) * n_cols * (2.0 * double(g_max) + 8.0) / (0.5 * slots));
            const double waves = n_cols * double(rows) / std::max(chunk, 1.0);
            if (waves >= threshold * slots)
                strips = 2;
        } else if (n_passes >= 2 && rows >= 12288 && width >= 4096) {
            strips = 2;
        }
    }
    if (rows < std::uint64_t(strips) * 8 * std::max<std::uint64_t>(g_max, 1))
        strips = 1;
    return std::min(strips, 8);
}

int ststhip_scatter_fields(const void *aos, size_t cell_size, size_t n_cells, int n_fields,
                           const size_t *field_offset, const size_t *field_size,
                           void *const *planes, ststhip_stream stream) {
    if (!aos || !field_offset || !field_size || !planes)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    if (int rc = ststhip_init(-1))
        return rc;
    FieldTable table;
    if (int rc = fill_table(table, cell_size, n_fields, field_offset, field_size, planes))
        return rc;
    if (n_cells == 0)
        return STSTHIP_OK;
    const unsigned blocks = unsigned((n_cells + transform_tile - 1) / transform_tile);
    hipLaunchKernelGGL(scatter_kernel, dim3(blocks), dim3(transform_tile),
                       transform_tile * cell_size, resolve(stream),
                       static_cast<const unsigned char *>(aos), unsigned(cell_size), n_cells, table);
    HIP_TRY(hipGetLastError());
    return STSTHIP_OK;
}

int ststhip_gather_fields(void *aos, size_t cell_size, size_t n_cells, int n_fields,
                          const size_t *field_offset, const size_t *field_size,
                          const void *const *planes, ststhip_stream stream) {
    if (!aos || !field_offset || !field_size || !planes)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    if (int rc = ststhip_init(-1))
        return rc;
    FieldTable table;
    if (int rc = fill_table(table, cell_size, n_fields, field_offset, field_size,
                            const_cast<void *const *>(planes)))
        return rc;
    if (n_cells == 0)
        return STSTHIP_OK;
    const unsigned blocks = unsigned((n_cells + transform_tile - 1) / transform_tile);
    hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(transform_tile), transform_tile * cell_size,
                       resolve(stream), static_cast<unsigned char *>(aos), unsigned(cell_size),
                       n_cells, table);
    HIP_TRY(hipGetLastError());
    return STSTHIP_OK;
}

int ststhip_reduce_max_abs(const void *cells, size_t cell_size, uint64_t height, uint64_t width,
                           uint64_t pitch, int n_fields, const ststhip_reduce_field *fields,
                           double *result, ststhip_stream stream) {
    if (!fields || !result || n_fields < 1 || n_fields > 8 || cell_size == 0 || pitch < width)
        return fail(STSTHIP_ERR_INVALID, "reduce_max_abs: need 1..8 fields, a result buffer and pitch >= width");
    if (int rc = ststhip_init(-1))
        return rc;
    ReduceTable table;
    table.n_fields = n_fields;
    bool any = false;
    for (int f = 0; f < n_fields; f++) {
        const size_t size = fields[f].type == STSTHIP_F64 ? 8 : 4;
        if (fields[f].type > STSTHIP_F64 || fields[f].offset + size > cell_size || fields[f].offset % size != 0 ||
            cell_size % size != 0)
            return fail(STSTHIP_ERR_INVALID, "reduce_max_abs: field outside the cell, misaligned or of unknown type");
        table.offset[f] = fields[f].offset;
        table.type[f] = fields[f].type;
        table.row_limit[f] = fields[f].row_limit < height ? fields[f].row_limit : height;
        table.col_limit[f] = fields[f].col_limit < width ? fields[f].col_limit : width;
        any = any || (table.row_limit[f] > 0 && table.col_limit[f] > 0);
        result[f] = -HUGE_VAL;
    }
    if (!any)
        return STSTHIP_OK;
    if (!cells)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    hipStream_t s = resolve(stream);
    void *acc = nullptr;
    if (int rc = ststhip_malloc_async(&acc, 8 * sizeof(unsigned long long), stream))
        return rc;
    unsigned long long host[8] = {0};
    int rc = STSTHIP_OK;
    hipError_t err = hipMemsetAsync(acc, 0, sizeof host, s);
    if (err == hipSuccess) {
        const unsigned long long n = height * width;
        const unsigned blocks = unsigned(std::min<unsigned long long>((n + 255) / 256, 16ull * rt().compute_units));
        hipLaunchKernelGGL(reduce_max_abs_kernel, dim3(blocks), dim3(256), 0, s,
                           static_cast<const unsigned char *>(cells), unsigned(cell_size),
                           (unsigned long long)height, (unsigned long long)width, (unsigned long long)pitch, table,
                           static_cast<unsigned long long *>(acc));
        err = hipGetLastError();
    }
    if (err == hipSuccess)
        err = hipMemcpyAsync(host, acc, sizeof host, hipMemcpyDeviceToHost, s);
    if (err == hipSuccess)
        err = hipStreamSynchronize(s);
    if (err != hipSuccess)
        rc = hip_fail(err, "reduce_max_abs");
    ststhip_free_async(acc, stream);
    if (rc != STSTHIP_OK)
        return rc;
    for (int f = 0; f < n_fields; f++)
        if (table.row_limit[f] > 0 && table.col_limit[f] > 0)
            std::memcpy(&result[f], &host[f], sizeof(double)); // >= +0: at least one cell counted
    return STSTHIP_OK;
}

// ------------------------------------------------------------------ layer 1
int ststhip_app_count(void) { return int(apps().size()); }

int ststhip_app_info_at(int index, ststhip_app_info *info) {
    if (!info || index < 0 || index >= int(apps().size()))
        return fail(STSTHIP_ERR_INVALID, "app index out of range");
    *info = apps()[index].info;
    return STSTHIP_OK;
}

int ststhip_app_find(const char *name, ststhip_app_info *info) {
    const AppEntry *e = find_app(name);
    if (!e)
        return fail(STSTHIP_ERR_UNKNOWN_APP, "unknown transition function");
    if (info)
        *info = e->info;
    return STSTHIP_OK;
}

static int check_domain(const AppEntry *e, const ststhip_domain *dom, std::uint64_t out_begin,
                        std::uint64_t out_end, std::uint32_t n_generations) {
    if (!dom)
        return fail(STSTHIP_ERR_INVALID, "null domain");
    if (dom->pitch < (dom->local_cols ? dom->local_cols : dom->global_width))
        return fail(STSTHIP_ERR_INVALID, "pitch smaller than the columns the buffers hold");
    if (out_end > dom->global_height || out_begin > out_end)
        return fail(STSTHIP_ERR_INVALID, "output rows outside the grid");
    // compiled depths: max_generations and its repeated halvings
    bool compiled = false;
    for (std::uint32_t t = e->info.max_generations; t >= 1 && !compiled; t /= 2)
        compiled = (t == n_generations);
    if (!compiled)
        return fail(STSTHIP_ERR_INVALID,
                    "n_generations must be the app's max_generations or one of its repeated halvings");
    // every input row the sweep reads must be inside the buffers: this is what keeps a
    // hand-written kernel from touching memory it does not own
    const std::int64_t g = std::int64_t(n_generations) * e->info.halo_depth_per_generation;
    const std::int64_t need_lo = std::max<std::int64_t>(0, std::int64_t(out_begin) - g);
    const std::int64_t need_hi =
        std::min<std::int64_t>(std::int64_t(dom->global_height), std::int64_t(out_end) + g);
    if (out_end > out_begin &&
        (need_lo < dom->row_origin || need_hi > dom->row_origin + std::int64_t(dom->local_rows)))
        return fail(STSTHIP_ERR_INVALID, "buffers do not hold the ghost rows this sweep reads");
    // the same for the columns of a block (whole rows: nothing to check)
    if (dom->local_cols) {
        const std::int64_t held_lo = std::max<std::int64_t>(0, dom->col_origin);
        const std::int64_t held_hi = std::min<std::int64_t>(std::int64_t(dom->global_width),
                                                            dom->col_origin + std::int64_t(dom->local_cols));
        std::int64_t cb = std::int64_t(g_col_begin), ce = std::int64_t(g_col_end);
        if (cb == ce) {
            cb = held_lo;
            ce = held_hi;
        }
        const std::int64_t want_lo = std::max<std::int64_t>(0, cb - g);
        const std::int64_t want_hi = std::min<std::int64_t>(std::int64_t(dom->global_width), ce + g);
        if (cb < held_lo || ce > held_hi || ((cb != held_lo || ce != held_hi) && (want_lo < held_lo || want_hi > held_hi)))
            return fail(STSTHIP_ERR_INVALID, "buffers do not hold the ghost columns this sweep reads");
    }
    return STSTHIP_OK;
}

int ststhip_app_sweep(const char *app, const void *tf_params, const void *halo_cell,
                      const ststhip_domain *dom, const void *const *src, void *const *dst,
                      uint64_t out_row_begin, uint64_t out_row_end, uint64_t iteration,
                      uint32_t n_generations, ststhip_stream stream) {
    const AppEntry *e = find_app(app);
    if (!e)
        return fail(STSTHIP_ERR_UNKNOWN_APP, "unknown transition function");
    if (!tf_params || !halo_cell || !src || !dst)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    for (unsigned p = 0; p < e->info.n_planes; p++)
        if (!src[p] || !dst[p] || src[p] == dst[p])
            return fail(STSTHIP_ERR_INVALID, "source and target planes must be distinct non-null buffers");
    if (int rc = check_domain(e, dom, out_row_begin, out_row_end, n_generations))
        return rc;
    if (int rc = ststhip_init(-1))
        return rc;
    return e->sweep(tf_params, halo_cell, dom, src, dst, out_row_begin, out_row_end, iteration,
                    n_generations, resolve(stream));
}

// ------------------------------------------------------------------ a source that is still arriving
// (ststhip.h, ststhip_set_source_arrival; consumed by the next pass-driver call of the thread)
extern "C++" {
namespace {
std::vector<ststhip_source_block> &source_arrival() {
    static thread_local std::vector<ststhip_source_block> blocks;
    return blocks;
}
} // namespace
} // extern "C++"

int ststhip_set_source_arrival(const ststhip_source_block *blocks, uint32_t n_blocks) {
    source_arrival().clear();
    if (n_blocks == 0)
        return STSTHIP_OK;
    if (!blocks)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    std::uint64_t before = 0;
    for (std::uint32_t i = 0; i < n_blocks; i++) {
        if (!blocks[i].ready || blocks[i].row_end <= before)
            return fail(STSTHIP_ERR_INVALID, "source blocks need an event each and ascending row ends");
        before = blocks[i].row_end;
    }
    source_arrival().assign(blocks, blocks + n_blocks);
    return STSTHIP_OK;
}

int ststhip_suggest_upload_blocks(uint64_t rows, uint64_t row_bytes, uint32_t *n_blocks) {
    if (!n_blocks)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    *n_blocks = 1;
    if (!opt().stream_upload || rows == 0 || row_bytes == 0)
        return STSTHIP_OK;
    // Up to eight blocks of 128 MiB or more (2.3 ms over PCIe) and 256 rows or more: the first block is what the chip
    // waits for with nothing to do; the number of blocks is how thin the tiles of the skewed passes get, and a tile
    // of fewer cells than that leaves most of the chip idle (HotSpot 8192^2 in eight blocks of 64 MiB: 7 % SLOWER than one
    // copy in front of the first pass, profiles/r04_stream_upload.txt).
    const std::uint64_t bytes = rows * row_bytes;
    const std::uint64_t block_bytes = opt().upload_block_mib > 0 ? (std::uint64_t(opt().upload_block_mib) << 20) : 0;
    std::uint64_t n = block_bytes ? (bytes + block_bytes - 1) / block_bytes : std::min<std::uint64_t>(8, bytes >> 27);
    n = std::min<std::uint64_t>(n, rows / 256);
    n = std::min<std::uint64_t>(n, 64);
    *n_blocks = std::uint32_t(std::max<std::uint64_t>(n, 1));
    return STSTHIP_OK;
}

int ststhip_upload_streams(ststhip_stream *copies, ststhip_stream *work) {
    if (!copies || !work)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    if (int rc = ststhip_init(-1))
        return rc;
    // the two band streams: they idle while a source arrives (the tiles of the skewed passes run on the caller's
    // stream and the side stream), and streams of their own would be dealt onto the hardware queues of those (see
    // shared_side_streams)
    std::vector<hipStream_t> band;
    if (!band_streams_for(nullptr, 2, band))
        return fail(STSTHIP_ERR_HIP, "could not create the upload streams");
    *copies = static_cast<ststhip_stream>(band[1]);
    *work = static_cast<ststhip_stream>(band[0]);
    return STSTHIP_OK;
}

// ------------------------------------------------------------------ pass driver
// Side streams for "virtual strips": one launch per pass has a ramp-up and a ragged tail during which
// part of the chip idles (at 16384^2 a pass is only ~3 residency rounds long).  Splitting the rows
// into V strips that advance on V streams, coupled only through their G-row boundary bands, lets
// the tail of one strip's kernel overlap with the next kernels of the other strips.
static std::vector<std::uint32_t> plan_depths(std::uint64_t n_iterations, std::uint32_t max_generations,
                                              std::uint32_t depth_cap = 0) {
    std::vector<std::uint32_t> depths;
    int cap = opt().max_generations > 0 ? opt().max_generations : int(max_generations);
    if (depth_cap > 0)
        cap = std::min(cap, int(depth_cap));
    std::uint64_t remaining = n_iterations;
    while (remaining > 0) {
        std::uint32_t t = max_generations;
        while (t > 1 && (t > remaining || int(t) > cap))
            t /= 2;
        depths.push_back(t);
        remaining -= t;
    }
    return depths;
}

// Depths chosen by measurement (ststhip_sweep_desc::alt_generations), per kernel family and grid shape, for the process.
extern "C++" {
namespace {
struct TunedKey {
    std::uint64_t key, height, width;
    bool operator<(TunedKey const &o) const { return std::tie(key, height, width) < std::tie(o.key, o.height, o.width); }
};
std::map<TunedKey, std::uint32_t> &tuned_depths() {
    static std::map<TunedKey, std::uint32_t> m;
    return m;
}
} // namespace
} // extern "C++"

int ststhip_tuned_depth(uint64_t tune_key, uint64_t height, uint64_t width, uint32_t *depth) {
    if (!depth)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> guard(rt().lock);
    auto it = tuned_depths().find(TunedKey{tune_key, height, width});
    *depth = it == tuned_depths().end() ? 0u : it->second;
    return STSTHIP_OK;
}

int ststhip_app_scratch_bytes(const char *app, uint32_t n_generations, size_t *bytes_per_work_item) {
    const AppEntry *e = find_app(app);
    if (!e)
        return fail(STSTHIP_ERR_UNKNOWN_APP, "unknown transition function");
    if (!bytes_per_work_item)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    if (int rc = ststhip_init(-1))
        return rc;
    return e->scratch_bytes(n_generations, bytes_per_work_item);
}

int ststhip_app_tuned_depth(const char *app, uint64_t height, uint64_t width, uint32_t *depth) {
    const AppEntry *e = find_app(app);
    if (!e)
        return fail(STSTHIP_ERR_UNKNOWN_APP, "unknown transition function");
    return ststhip_tuned_depth(reinterpret_cast<std::uintptr_t>(e), height, width, depth);
}

int ststhip_run_passes(ststhip_sweep_fn sweep, void *ctx, const ststhip_sweep_desc *desc,
                       const ststhip_domain *dom, const void *const *src, void *const *dst,
                       uint64_t iteration_offset, uint64_t n_iterations, int blocking, int profiling,
                       ststhip_stream stream, ststhip_run_info *info) {
    // (the list of a source that is still arriving is for this call, whatever becomes of it)
    std::vector<ststhip_source_block> arrival;
    arrival.swap(source_arrival());
    if (!sweep || !desc || !dom || !src || !dst)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    if (desc->n_planes < 1 || desc->n_planes > 16 || desc->max_generations < 1)
        return fail(STSTHIP_ERR_INVALID, "bad sweep description");
    if (dom->row_origin != 0 || dom->local_rows != dom->global_height)
        return fail(STSTHIP_ERR_INVALID, "the pass driver works on whole grids (row_origin 0)");
    if (int rc = ststhip_init(-1))
        return rc;
    hipStream_t s = resolve(stream);
    const unsigned n_planes = desc->n_planes;
    const std::size_t plane_cells = std::size_t(dom->local_rows) * dom->pitch;
    const std::uint64_t H = dom->global_height;
    auto started = std::chrono::high_resolution_clock::now();

    // The depth plan.  With a second candidate depth (desc->alt_generations) the first call for a grid shape times its
    // own first passes at both depths and keeps the faster (ststhip.h, ststhip_sweep_desc); later calls look it up.
    const std::uint32_t deep = desc->max_generations, alt = desc->alt_generations;
    const bool tunable = desc->tune_key != 0 && alt >= 2 && alt < deep && deep % alt == 0 && !profiling &&
                         opt().tune_depth == 1 && opt().max_generations <= 0;
    // (a family with a second depth runs it wherever nothing has been measured: that is the depth its rule trusts)
    std::uint32_t depth_cap = (alt >= 1 && alt < deep) ? alt : 0;
    bool probing = false;
    // STSTHIP_TUNE_DEPTH >= 2 names the depth outright (profiling runs must plan the launches of the unprofiled run)
    if (depth_cap && opt().tune_depth >= 2 && (std::uint32_t(opt().tune_depth) == alt || std::uint32_t(opt().tune_depth) == deep)) {
        depth_cap = std::uint32_t(opt().tune_depth);
    } else if (tunable) {
        std::uint32_t known = 0;
        ststhip_tuned_depth(desc->tune_key, H, dom->global_width, &known);
        if (known)
            depth_cap = known;
        else
            probing = n_iterations >= 6ull * deep;
    }
    std::vector<std::uint32_t> depths;
    // a probing call: [deep (untimed: clocks and caches settle)] [deep, deep] [alt x 2*deep/alt] then the rest at the
    // winner's depth, planned once the probes have been timed.  The ping-pong parity of the targets must be fixed
    // before that: it is that of the plan that continues at `deep`; if the other plan wins with the other parity, one
    // of its passes is split into two of half its depth.
    const std::size_t probe_passes = probing ? 3 + 2 * (deep / alt) : 0;
    std::vector<std::uint32_t> rest_alt;
    // (a lambda: planned again for what is left once the passes behind an arriving source have run)
    auto plan = [&](std::uint64_t n) {
        depths.clear();
        rest_alt.clear();
        if (probing) {
            for (int i = 0; i < 3; i++)
                depths.push_back(deep);
            for (std::uint32_t i = 0; i < 2 * (deep / alt); i++)
                depths.push_back(alt);
            const std::uint64_t left = n - 5ull * deep;
            const std::vector<std::uint32_t> rest_deep = plan_depths(left, deep);
            rest_alt = plan_depths(left, deep, alt);
            if ((rest_alt.size() + rest_deep.size()) % 2 != 0) {
                // split the first pass of depth `alt` (there is one: left >= deep) into two of alt / 2
                for (std::size_t i = 0; i < rest_alt.size(); i++)
                    if (rest_alt[i] == alt) {
                        rest_alt[i] = alt / 2;
                        rest_alt.insert(rest_alt.begin() + i, alt / 2);
                        break;
                    }
            }
            depths.insert(depths.end(), rest_deep.begin(), rest_deep.end()); // replaced by rest_alt if alt wins
        } else {
            depths = plan_depths(n, deep, depth_cap);
        }
    };
    plan(n_iterations);
    // The passes the whole call takes; which of `dst` and the scratch planes a pass writes follows from its index and
    // this number alone (the last one writes `dst`).
    const std::size_t total_passes = depths.size();

    // A source that is still arriving (ststhip_set_source_arrival): the first `streamed` passes run as row tiles
    // behind it, at the depth the plan starts with.  How many is decided while they run, so the plan of the rest
    // must have the same length for every choice: it has when the choice is a multiple of `quantum` (a plan is
    // greedy: whole launches of its depth first; in a probing call 2 * deep / alt of them make two launches of `deep`).
    const std::uint32_t tile_depth = depths.empty() ? 0 : (probing ? alt : depths.front());
    const std::uint64_t tile_g = std::uint64_t(tile_depth) * desc->halo_depth_per_generation;
    const std::uint32_t quantum = probing ? 2 * (deep / alt) : 1;
    std::uint64_t stream_limit = 0; // passes that may run behind the source, at most
    if (arrival.size() >= 2 && !profiling && opt().stream_upload && tile_depth > 0) {
        std::uint64_t thinnest = ~0ull, before = 0;
        for (auto const &blk : arrival) {
            thinnest = std::min(thinnest, std::min(blk.row_end, H) - std::min(before, H));
            before = blk.row_end;
        }
        // pass p of a block's column ends (p + 1) * g rows above the block's end: inside the block for every p
        const std::uint64_t by_rows = thinnest > tile_g ? (thinnest - 1) / tile_g - 1 : 0;
        const std::uint64_t spare = probing ? n_iterations - 6ull * deep : n_iterations;
        stream_limit = std::min(by_rows, spare / tile_depth);
        stream_limit -= stream_limit % quantum;
        if (before < H || stream_limit < 2)
            stream_limit = 0;
    }
    hipEvent_t probe_events[3] = {nullptr, nullptr, nullptr};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timed;
    EventPool sync_events;
    std::uint64_t n_launches = 0;
    int rc = STSTHIP_OK;
    void *scratch[16] = {nullptr};
    void *tdv_table = nullptr;
    // events and stream waits carry the ordering between strips: a failure there must fail the run
    auto ordered = [&](hipError_t err, const char *what) {
        if (err != hipSuccess && rc == STSTHIP_OK)
            rc = hip_fail(err, what);
    };

    // how many virtual strips: only worth it for grids with many rows per strip
    const std::uint64_t g_max = std::uint64_t((!probing && depth_cap) ? std::min(deep, depth_cap) : deep) *
                                desc->halo_depth_per_generation;
    int strips = profiling ? 1
                           : suggest_row_strips(H, dom->global_width, desc->strip_width, g_max, depths.size());

    // a source in blocks that this call does not follow block by block: everything waits for all of it
    if (stream_limit == 0)
        for (auto const &blk : arrival)
            ordered(hipStreamWaitEvent(s, static_cast<hipEvent_t>(blk.ready), 0), "hipStreamWaitEvent");
    std::uint64_t streamed = 0; // passes that ran behind the arriving source

    if (depths.empty()) {
        for (unsigned p = 0; p < n_planes && rc == STSTHIP_OK; p++)
            rc = ststhip_memcpy_d2d(dst[p], src[p], plane_cells * desc->plane_elem_size[p], s);
    } else {
        if (depths.size() > 1)
            for (unsigned p = 0; p < n_planes && rc == STSTHIP_OK; p++)
                rc = ststhip_malloc_async(&scratch[p], plane_cells * desc->plane_elem_size[p], s);

        // one device table of time-dependent values for the whole call
        if (rc == STSTHIP_OK && desc->tdv_size > 0 && (desc->fill_tdv || desc->tdv_device_table)) {
            const std::size_t bytes = std::size_t(desc->tdv_size) * n_iterations;
            if (desc->tdv_device_table) {
                g_tdv_table = desc->tdv_device_table;
            } else {
                rc = ststhip_malloc_async(&tdv_table, bytes, s);
                if (rc == STSTHIP_OK) {
                    std::vector<unsigned char> values(bytes);
                    desc->fill_tdv(ctx, iteration_offset, n_iterations, values.data());
                    // `values` is pageable and goes out of scope: the copy must have read it before that
                    ordered(hipMemcpyAsync(tdv_table, values.data(), bytes, hipMemcpyHostToDevice, s), "hipMemcpyAsync");
                    ordered(hipStreamSynchronize(s), "hipStreamSynchronize");
                    g_tdv_table = tdv_table;
                }
            }
            g_tdv_first = iteration_offset;
            g_tdv_count = n_iterations;
            g_tdv_size = desc->tdv_size;
        }
        auto new_event = [&]() {
            hipEvent_t e = sync_events.take();
            if (!e)
                ordered(hipErrorUnknown, "hipEventCreateWithFlags");
            return e;
        };
        // which planes pass k of the call writes: the last one `dst`, alternating backwards from there
        auto target_of = [&](std::uint64_t k) -> void *const * {
            return ((total_passes - 1 - k) % 2) == 0 ? dst : scratch;
        };

        // ---- passes behind a source that is still arriving (ststhip.h, ststhip_set_source_arrival) ----
        // frontier[p] = rows [0, frontier[p]) of pass p are done (or queued).  When block k has arrived, rows
        // [0, end_k) of the source are there and pass p may advance to end_k - (p + 1) * g: a tile of pass p reads g rows
        // beyond its own on both sides in the output of pass p - 1, and pass p - 1 has come g rows further.  The tiles of
        // one block (its "column": p = 0, 1, ...) form a chain on one stream; the columns of consecutive blocks run on
        // different streams, one pass apart (a tile waits for the tiles of the pass before it that overlap what it
        // reads -- the same tiles are the last readers of the rows it overwrites, two passes share a buffer set).
        if (stream_limit > 0 && rc == STSTHIP_OK) {
            struct Tile {
                std::uint64_t begin, end;
                hipEvent_t done;
                hipStream_t on;
            };
            std::vector<hipStream_t> tile_lane(1, s), extra;
            if (side_streams_for(s, 1, extra))
                tile_lane.push_back(extra[0]);
            {
                // (the scratch planes were allocated for work on `s`)
                hipEvent_t begin = new_event();
                ordered(hipEventRecord(begin, s), "hipEventRecord");
                for (std::size_t v = 1; v < tile_lane.size(); v++)
                    ordered(hipStreamWaitEvent(tile_lane[v], begin, 0), "hipStreamWaitEvent");
            }
            std::vector<std::vector<Tile>> tiles(stream_limit);
            std::vector<std::uint64_t> frontier(stream_limit, 0);
            // How deep a column goes: as deep as it takes to keep the chip busy until the next block is there -- and a
            // little busier: a tile of one block's rows cannot fill the chip on its own (a 2048 x 16384 tile of the
            // Jacobi example runs at 57 % of the whole grid's rate), two columns side by side do better, so the aim is ONE
            // or two earlier columns still running when a block arrives.  None: the chip has idled, and the column
            // before is timed against the blocks' arrival; three or more: the chip is behind, the next column goes
            // less deep (passes that are begun are completed by the last column whatever the later columns did).
            std::uint64_t row_bytes_all_planes = 0;
            for (unsigned p = 0; p < n_planes; p++)
                row_bytes_all_planes += dom->pitch * desc->plane_elem_size[p];
            std::uint64_t deep_now = std::min<std::uint64_t>(stream_limit, 16);
            deep_now = std::max<std::uint64_t>(deep_now - deep_now % quantum, quantum);
            std::uint64_t deepest = 0;
            std::vector<hipEvent_t> clocks; // timing events of the columns: begin, end
            g_launch_concurrency = 2;
            const bool narrate = std::getenv("STSTHIP_TRACE_STREAM") != nullptr; // (a debugging aid: host times to stderr)
            auto host_ms = [&] {
                return std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - started).count();
            };
            for (std::size_t k = 0; k < arrival.size() && rc == STSTHIP_OK; k++) {
                const double waiting_since = host_ms();
                ordered(hipEventSynchronize(static_cast<hipEvent_t>(arrival[k].ready)), "hipEventSynchronize");
                if (rc != STSTHIP_OK)
                    break;
                const bool last = k + 1 == arrival.size();
                int behind = 0; // earlier columns that are still running
                for (std::size_t j = 0; j < k; j++)
                    behind += event_done(clocks[2 * j + 1]) ? 0 : 1;
                if (narrate)
                    std::fprintf(stderr, "[ststhip] block %zu: host waited from %.2f to %.2f ms of the call, %d columns running", k,
                                 waiting_since, host_ms(), behind);
                if (k > 0 && !last) {
                    std::uint64_t want = deep_now;
                    if (behind == 0) {
                        // t = what a tile of the column before took (alone, or beside the one before it: the estimate
                        // is then on the careful side); two columns side by side finish a tile every 0.8 t
                        // (a block's time over PCIe by its bytes at 55 GB/s: the gaps between the arrival events are no
                        // measure -- the first blocks of an upload have been seen to arrive 5-6 ms ahead of the rest)
                        float column_ms = 0.0f;
                        const double block_ms = double(std::min(arrival[k].row_end, H) - std::min(arrival[k - 1].row_end, H)) *
                                                double(row_bytes_all_planes) / 55e6;
                        double grow = 1.25;
                        if (hipEventElapsedTime(&column_ms, clocks[2 * (k - 1)], clocks[2 * (k - 1) + 1]) == hipSuccess &&
                            column_ms > 0.0f)
                            grow = std::min(4.0, std::max(1.25, block_ms / (0.8 * double(column_ms))));
                        if (narrate)
                            std::fprintf(stderr, " (column before: %.2f ms, block: %.2f ms)", column_ms, block_ms);
                        want = std::uint64_t(double(deep_now) * grow + 0.5);
                    } else if (behind >= 3) {
                        want = deep_now * 3 / 4;
                    }
                    want -= want % quantum;
                    deep_now = std::min(std::max<std::uint64_t>(want, quantum), stream_limit);
                }
                deepest = std::max(deepest, deep_now);
                const std::uint64_t arrived = last ? H : std::min(arrival[k].row_end, H);
                hipStream_t on = tile_lane[k % tile_lane.size()];
                hipEvent_t c0 = nullptr, c1 = nullptr;
                ordered(hipEventCreate(&c0), "hipEventCreate");
                ordered(hipEventCreate(&c1), "hipEventCreate");
                clocks.push_back(c0);
                clocks.push_back(c1);
                if (rc != STSTHIP_OK)
                    break;
                ordered(hipEventRecord(c0, on), "hipEventRecord");
                // (the last column completes every pass that was begun)
                for (std::uint64_t p = 0; p < (last ? deepest : deep_now) && rc == STSTHIP_OK; p++) {
                    const std::uint64_t above = p == 0 ? arrived : frontier[p - 1];
                    const std::uint64_t upto = above >= H ? H : (above > tile_g ? above - tile_g : 0);
                    if (upto <= frontier[p])
                        break;
                    const std::uint64_t from_row = frontier[p];
                    if (p > 0)
                        for (Tile const &t : tiles[p - 1])
                            if (t.on != on && t.end + tile_g > from_row && t.begin < upto + tile_g)
                                ordered(hipStreamWaitEvent(on, t.done, 0), "hipStreamWaitEvent");
                    g_target_holds_constants = p >= 2 ? 1 : 0;
                    rc = sweep(ctx, dom, p == 0 ? src : const_cast<const void *const *>(target_of(p - 1)), target_of(p),
                               from_row, upto, iteration_offset + p * tile_depth, tile_depth, on);
                    n_launches++;
                    Tile t{from_row, upto, new_event(), on};
                    ordered(hipEventRecord(t.done, on), "hipEventRecord");
                    tiles[p].push_back(t);
                    frontier[p] = upto;
                }
                ordered(hipEventRecord(c1, on), "hipEventRecord");
                if (narrate)
                    std::fprintf(stderr, ", column of %llu passes queued by %.2f ms\n",
                                 (unsigned long long)(last ? deepest : deep_now), host_ms());
                if (last)
                    streamed = deepest;
            }
            for (std::size_t v = 1; v < tile_lane.size(); v++) {
                hipEvent_t done = new_event();
                ordered(hipEventRecord(done, tile_lane[v]), "hipEventRecord");
                ordered(hipStreamWaitEvent(s, done, 0), "hipStreamWaitEvent");
            }
            if (narrate && rc == STSTHIP_OK && hipStreamSynchronize(s) == hipSuccess) {
                for (std::size_t k = 0; 2 * k + 1 < clocks.size(); k++) {
                    float ms = 0.0f, since = 0.0f;
                    (void)hipEventElapsedTime(&ms, clocks[2 * k], clocks[2 * k + 1]);
                    (void)hipEventElapsedTime(&since, clocks[0], clocks[2 * k]);
                    std::fprintf(stderr, "[ststhip] column %zu ran from %.2f for %.2f ms\n", k, since, ms);
                }
                std::fprintf(stderr, "[ststhip] streamed passes done at %.2f ms of the call\n", host_ms());
            }
            for (hipEvent_t ev : clocks)
                if (ev)
                    (void)hipEventDestroy(ev);
            g_launch_concurrency = 1;
            g_target_holds_constants = 0;
            // every pass that was begun is complete now (the last block's column took each of them to the last row)
            for (std::uint64_t p = 0; p < streamed && rc == STSTHIP_OK; p++)
                if (frontier[p] != H)
                    rc = fail(STSTHIP_ERR_INVALID, "internal: a streamed pass is incomplete");
            if (rc == STSTHIP_OK && streamed > 0) {
                plan(n_iterations - streamed * tile_depth);
                // (a probing call's plans differ in length by whole pairs of passes)
                if ((streamed + depths.size()) % 2 != total_passes % 2)
                    rc = fail(STSTHIP_ERR_INVALID, "internal: the plan behind the streamed passes has another parity");
                strips = suggest_row_strips(H, dom->global_width, desc->strip_width, g_max, depths.size());
            }
        }

        // streams of the strips: strip 0 runs on the caller's stream
        std::vector<hipStream_t> lane(strips, s);
        if (strips > 1 && rc == STSTHIP_OK) {
            std::vector<hipStream_t> extra;
            if (side_streams_for(s, strips - 1, extra)) {
                hipEvent_t begin = new_event();
                ordered(hipEventRecord(begin, s), "hipEventRecord");
                for (int v = 1; v < strips; v++) {
                    lane[v] = extra[v - 1];
                    ordered(hipStreamWaitEvent(lane[v], begin, 0), "hipStreamWaitEvent");
                }
            } else {
                strips = 1;
                lane.assign(1, s);
            }
        }
        std::vector<std::uint64_t> bound(strips + 1);
        for (int v = 0; v <= strips; v++)
            bound[v] = H * std::uint64_t(v) / std::uint64_t(strips);
        // unequal strips drift out of phase, so one strip's tail meets the other's bulk (400 while the bands sat in
        // front of the interiors); with the bands beside the interiors equal strips are as good or better
        if (strips == 2)
            bound[1] = H * std::uint64_t(opt().strip_skew_permille > 0 ? opt().strip_skew_permille
                                                                        : (opt().bands_beside_interior ? 500 : 400)) / 1000;
        // Boundary bands on streams of their own (highest priority), beside the interior of the same strip and
        // pass: a band's output only feeds the NEXT pass, and in the strip's own stream it sat in front of the
        // interior for 110-165 us per pass (12 rows of work that queue behind the other strip's resident waves;
        // profiles/r02_bench_summary.json, launch shape 12544).  Dependencies per pass p and strip v:
        //   bands(v, p)    after interior(v, p-1), bands(v-1, p-1), bands(v+1, p-1)   [own bands(p-1): stream order]
        //   interior(v, p) after bands(v-1 .. v+1, p-1)                               [own interior(p-1): stream order]
        // -- every row a launch reads was written by one of those, and every row it overwrites (the other buffer
        // set) was last read by one of those.
        std::vector<hipStream_t> band_lane(strips, nullptr);
        const bool bands_beside = strips > 1 && opt().bands_beside_interior != 0 &&
                                  band_streams_for(s, strips, band_lane);
        if (bands_beside) {
            hipEvent_t begin = new_event();
            ordered(hipEventRecord(begin, s), "hipEventRecord");
            for (int v = 0; v < strips; v++)
                ordered(hipStreamWaitEvent(band_lane[v], begin, 0), "hipStreamWaitEvent");
        }
        // Strips with moving boundaries instead of strips with boundary bands (see the pass loop).  The boundaries move
        // around their places of rest: by what the call's plan moves them, at most a quarter of a strip's rows.
        std::uint64_t skew_span = 0;
        for (std::size_t i = 0; i < depths.size(); i++)
            skew_span += std::uint64_t(std::max(depths[i], i ? depths[i - 1] : 0u)) * desc->halo_depth_per_generation;
        skew_span = std::min<std::uint64_t>(skew_span, std::min<std::uint64_t>(H / std::uint64_t(2 * strips), 2048 + 2 * g_max));
        const bool skewed_strips = strips >= 2 && !profiling && opt().skewed_strips != 0 &&
                                   H >= std::uint64_t(strips) * 32 * g_max && skew_span >= 2 * g_max;
        std::uint64_t skew_offset = 0, skew_g_before = 0; // how far below their highest places the boundaries are
        std::vector<hipEvent_t> skew_done(strips, nullptr); // per strip: its launch of the previous pass
        std::vector<hipEvent_t> bands_done(strips, nullptr);    // per strip: bands of the previous pass
        std::vector<hipEvent_t> interior_done(strips, nullptr); // per strip: interior of the previous pass
        g_launch_concurrency = strips;

        // the last pass must land in dst; the input is never written
        const void *const *from = streamed ? const_cast<const void *const *>(target_of(streamed - 1)) : src;
        std::uint64_t iteration = iteration_offset + streamed * tile_depth;
        // a point in time on every stream of the call: everything queued so far has finished before `ev`, nothing
        // queued later starts before it (the boundaries of the depth probes' timed groups)
        auto fence_all = [&](hipEvent_t ev) {
            for (int v = 1; v < strips; v++)
                ordered(hipStreamWaitEvent(s, [&] { hipEvent_t e = new_event(); ordered(hipEventRecord(e, lane[v]), "hipEventRecord"); return e; }(), 0),
                        "hipStreamWaitEvent");
            if (bands_beside)
                for (int v = 0; v < strips; v++)
                    ordered(hipStreamWaitEvent(s, [&] { hipEvent_t e = new_event(); ordered(hipEventRecord(e, band_lane[v]), "hipEventRecord"); return e; }(), 0),
                            "hipStreamWaitEvent");
            ordered(hipEventRecord(ev, s), "hipEventRecord");
            for (int v = 1; v < strips; v++)
                ordered(hipStreamWaitEvent(lane[v], ev, 0), "hipStreamWaitEvent");
            if (bands_beside)
                for (int v = 0; v < strips; v++)
                    ordered(hipStreamWaitEvent(band_lane[v], ev, 0), "hipStreamWaitEvent");
        };
        if (probing)
            for (hipEvent_t &ev : probe_events)
                ordered(hipEventCreate(&ev), "hipEventCreate");
        for (std::size_t pass = 0; pass < depths.size() && rc == STSTHIP_OK; pass++) {
            if (probing && (pass == 1 || pass == 3 || pass == probe_passes)) {
                const int boundary = pass == 1 ? 0 : (pass == 3 ? 1 : 2);
                fence_all(probe_events[boundary]);
                if (boundary == 2) {
                    // the host waits for the probes here (a few milliseconds, once per kernel family and grid shape)
                    float ms_deep = 0.0f, ms_alt = 0.0f;
                    ordered(hipEventSynchronize(probe_events[2]), "hipEventSynchronize");
                    ordered(hipEventElapsedTime(&ms_deep, probe_events[0], probe_events[1]), "hipEventElapsedTime");
                    ordered(hipEventElapsedTime(&ms_alt, probe_events[1], probe_events[2]), "hipEventElapsedTime");
                    if (rc != STSTHIP_OK)
                        break;
                    // (the shallower depth has to win by 2 %: a tie keeps the plan that is already laid out)
                    const bool take_alt = ms_alt < 0.98f * ms_deep;
                    if (std::getenv("STSTHIP_TRACE_STREAM"))
                        std::fprintf(stderr, "[ststhip] depth probe: 2 launches of %u generations %.3f ms, %u of %u %.3f ms -> %u\n",
                                     deep, ms_deep, 2 * (deep / alt), alt, ms_alt, take_alt ? alt : deep);
                    if (take_alt) {
                        depths.resize(probe_passes);
                        depths.insert(depths.end(), rest_alt.begin(), rest_alt.end());
                    }
                    std::lock_guard<std::mutex> guard(rt().lock);
                    tuned_depths()[TunedKey{desc->tune_key, H, dom->global_width}] = take_alt ? alt : deep;
                }
            }
            void *const *to = target_of(streamed + pass);
            // targets alternate, and every pass writes all rows: from the third pass on the target already holds
            // what the pass before the previous one stored there, in particular the fields that never change
            g_target_holds_constants = streamed + pass >= 2 ? 1 : 0;
            const std::uint64_t g = std::uint64_t(depths[pass]) * desc->halo_depth_per_generation;
            hipEvent_t t0 = nullptr, t1 = nullptr;
            if (profiling) {
                ordered(hipEventCreate(&t0), "hipEventCreate");
                ordered(hipEventCreate(&t1), "hipEventCreate");
                ordered(hipEventRecord(t0, s), "hipEventRecord");
            }
            std::vector<hipEvent_t> bands_now(strips, nullptr), interior_now(strips, nullptr);
            if (skewed_strips) {
                // Strips whose common boundaries move UP by a launch's ghost rows from pass to pass (the larger of this
                // pass's and the one's before): a strip of pass p then reads nothing the strip BELOW it wrote in pass
                // p - 1 and overwrites nothing that one read -- it waits for the strip ABOVE it of the pass before and
                // for its own, nothing else; the uppermost strip's launches are a chain of their own.  No boundary
                // bands, one launch per strip and pass.  When the boundaries have moved a span, every strip waits for
                // the one below it once and the boundaries start over.
                const std::uint64_t shift = std::max(g, skew_g_before);
                bool restart = false;
                if (skew_offset + shift > skew_span) {
                    skew_offset = 0;
                    restart = true;
                } else {
                    skew_offset += shift;
                }
                skew_g_before = g;
                std::vector<hipEvent_t> done_now(strips, nullptr);
                for (int v = 0; v < strips && rc == STSTHIP_OK; v++) {
                    const std::uint64_t top = v == 0 ? 0 : bound[v] + skew_span / 2 - skew_offset;
                    const std::uint64_t end = v + 1 == strips ? H : bound[v + 1] + skew_span / 2 - skew_offset;
                    if (v > 0 && skew_done[v - 1])
                        ordered(hipStreamWaitEvent(lane[v], skew_done[v - 1], 0), "hipStreamWaitEvent");
                    if (restart && v + 1 < strips && skew_done[v + 1])
                        ordered(hipStreamWaitEvent(lane[v], skew_done[v + 1], 0), "hipStreamWaitEvent");
                    rc = sweep(ctx, dom, from, to, top, end, iteration, depths[pass], lane[v]);
                    n_launches++;
                    done_now[v] = new_event();
                    ordered(hipEventRecord(done_now[v], lane[v]), "hipEventRecord");
                }
                skew_done.swap(done_now);
                if (profiling) {
                    ordered(hipEventRecord(t1, s), "hipEventRecord");
                    timed.emplace_back(t0, t1);
                }
                from = const_cast<const void *const *>(to);
                iteration += depths[pass];
                continue;
            }
            for (int v = 0; v < strips && rc == STSTHIP_OK; v++) {
                const std::uint64_t a = bound[v], b = bound[v + 1];
                if (strips == 1) {
                    rc = sweep(ctx, dom, from, to, a, b, iteration, depths[pass], lane[v]);
                    n_launches++;
                    continue;
                }
                hipStream_t bands_on = bands_beside ? band_lane[v] : lane[v];
                // bands read the neighbours' bands of the previous pass (and will overwrite rows
                // the neighbours' previous bands read): wait for them
                if (v > 0 && bands_done[v - 1])
                    ordered(hipStreamWaitEvent(bands_on, bands_done[v - 1], 0), "hipStreamWaitEvent");
                if (v + 1 < strips && bands_done[v + 1])
                    ordered(hipStreamWaitEvent(bands_on, bands_done[v + 1], 0), "hipStreamWaitEvent");
                if (bands_beside && interior_done[v])
                    ordered(hipStreamWaitEvent(bands_on, interior_done[v], 0), "hipStreamWaitEvent");
                const std::uint64_t top_end = (v > 0) ? std::min(a + g, b) : a;
                const std::uint64_t bot_begin = (v + 1 < strips) ? std::max(b - std::min(g, b - a), top_end) : b;
                if (top_end > a) {
                    rc = sweep(ctx, dom, from, to, a, top_end, iteration, depths[pass], bands_on);
                    n_launches++;
                }
                if (rc == STSTHIP_OK && bot_begin < b) {
                    rc = sweep(ctx, dom, from, to, bot_begin, b, iteration, depths[pass], bands_on);
                    n_launches++;
                }
                bands_now[v] = new_event();
                ordered(hipEventRecord(bands_now[v], bands_on), "hipEventRecord");
                if (bands_beside) {
                    // the interior reads this strip's previous bands, and -- when this pass is shallower than the
                    // previous one -- overwrites rows next to them that the neighbours' previous bands read
                    for (int w = std::max(v - 1, 0); w <= std::min(v + 1, strips - 1); w++)
                        if (bands_done[w])
                            ordered(hipStreamWaitEvent(lane[v], bands_done[w], 0), "hipStreamWaitEvent");
                }
                if (rc == STSTHIP_OK && top_end < bot_begin) {
                    rc = sweep(ctx, dom, from, to, top_end, bot_begin, iteration, depths[pass], lane[v]);
                    n_launches++;
                }
                if (bands_beside) {
                    interior_now[v] = new_event();
                    ordered(hipEventRecord(interior_now[v], lane[v]), "hipEventRecord");
                }
            }
            bands_done.swap(bands_now);
            interior_done.swap(interior_now);
            if (profiling) {
                ordered(hipEventRecord(t1, s), "hipEventRecord");
                timed.emplace_back(t0, t1);
            }
            from = const_cast<const void *const *>(to);
            iteration += depths[pass];
        }
        // the band streams join the caller's stream too
        if (bands_beside)
            for (int v = 0; v < strips; v++) {
                hipEvent_t done = new_event();
                ordered(hipEventRecord(done, band_lane[v]), "hipEventRecord");
                ordered(hipStreamWaitEvent(s, done, 0), "hipStreamWaitEvent");
            }
        g_launch_concurrency = 1;
        g_target_holds_constants = 0;
        g_tdv_table = nullptr;
        g_tdv_count = 0;
        // join: the caller's stream continues after every strip has finished
        for (int v = 1; v < strips; v++) {
            hipEvent_t done = new_event();
            ordered(hipEventRecord(done, lane[v]), "hipEventRecord");
            ordered(hipStreamWaitEvent(s, done, 0), "hipStreamWaitEvent");
        }
    }
    if (streamed > 0 && std::getenv("STSTHIP_TRACE_STREAM")) {
        const double queued = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - started).count();
        (void)hipStreamSynchronize(s);
        std::fprintf(stderr, "[ststhip] all %zu + %llu passes queued at %.2f ms, done at %.2f ms of the call\n", depths.size(),
                     (unsigned long long)streamed, queued,
                     std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - started).count());
    }
    if (rc == STSTHIP_OK && (blocking || profiling)) {
        hipError_t err = hipStreamSynchronize(s);
        if (err != hipSuccess)
            rc = hip_fail(err, "hipStreamSynchronize");
    }
    // released on the caller's stream, which every strip has been joined into above: the blocks go to
    // another stream (or the host) only after the event recorded here -- also when a launch failed midway
    for (unsigned p = 0; p < n_planes; p++)
        if (scratch[p])
            ststhip_free_async(scratch[p], s);
    if (tdv_table)
        ststhip_free_async(tdv_table, s);
    for (hipEvent_t ev : probe_events)
        if (ev)
            (void)hipEventDestroy(ev);
    double kernel_s = 0.0;
    for (auto &ev : timed) {
        float ms = 0.0f;
        if (rc == STSTHIP_OK && hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess)
            kernel_s += double(ms) * 1e-3;
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    if (info) {
        std::chrono::duration<double> elapsed = std::chrono::high_resolution_clock::now() - started;
        info->walltime_s = elapsed.count();
        info->kernel_time_s = kernel_s;
        info->n_launches = n_launches;
        info->n_processed_cells = n_iterations * dom->global_height * dom->global_width;
        info->n_streamed_passes = streamed;
    }
    return rc;
}

namespace {
struct AppCall {
    const AppEntry *entry;
    const void *tf_params;
    const void *halo_cell;
};
// Jacobi5Uniform: which of the four kernels a launch needs depends on whether it contains the first
// and / or the last generation of the run.
struct UniformJacobiCall {
    const AppEntry *middle = nullptr, *first = nullptr, *last = nullptr, *only = nullptr;
    float c = 0.0f;
    const void *halo_cell = nullptr;
    std::uint64_t first_iteration = 0, last_iteration = 0;
};
int uniform_jacobi_trampoline(void *ctx, const ststhip_domain *dom, const void *const *src, void *const *dst,
                              uint64_t out_begin, uint64_t out_end, uint64_t iteration, uint32_t n_generations,
                              ststhip_stream stream) {
    const UniformJacobiCall *call = static_cast<const UniformJacobiCall *>(ctx);
    const bool has_first = iteration == call->first_iteration;
    const bool has_last = iteration + n_generations - 1 == call->last_iteration;
    const AppEntry *entry = has_first ? (has_last ? call->only : call->first) : (has_last ? call->last : call->middle);
    struct {
        float c;
    } block = {call->c};
    return entry->sweep(&block, call->halo_cell, dom, src, dst, out_begin, out_end, iteration, n_generations, stream);
}

void app_fill_tdv(void *ctx, uint64_t iteration_offset, uint64_t n_iterations, void *values) {
    const AppCall *call = static_cast<const AppCall *>(ctx);
    call->entry->fill_tdv(call->tf_params, iteration_offset, n_iterations, values);
}

int app_sweep_trampoline(void *ctx, const ststhip_domain *dom, const void *const *src, void *const *dst,
                         uint64_t out_begin, uint64_t out_end, uint64_t iteration, uint32_t n_generations,
                         ststhip_stream stream) {
    const AppCall *call = static_cast<const AppCall *>(ctx);
    return call->entry->sweep(call->tf_params, call->halo_cell, dom, src, dst, out_begin, out_end, iteration,
                              n_generations, stream);
}
} // namespace

int ststhip_suggest_row_strips(const char *app, uint64_t rows, uint64_t width, uint64_t n_passes) {
    const AppEntry *e = find_app(app);
    if (!e || ststhip_init(-1) != STSTHIP_OK)
        return 1;
    return suggest_row_strips(rows, width, e->info.strip_width,
                              std::uint64_t(e->info.max_generations) * e->info.halo_depth_per_generation,
                              n_passes);
}

// What ststhip_app_run and the strip driver launch for (app, parameters, halo, geometry): the registered sweep, or
// one of the two bit-identical fast forms the library switches to on its own (ststhip.h, ststhip_app_run).
namespace {
struct ResolvedApp {
    const AppEntry *entry = nullptr; // geometry and planes (the middle-launch kernel for the uniform form)
    ststhip_sweep_fn trampoline = nullptr;
    void *ctx = nullptr;
    bool uniform = false, packed = false;
    UniformJacobiCall uniform_call;
    AppCall call;
    std::uint32_t dead_word = 0;
    ststhip_domain words; // packed Game of Life: the domain in 32-bit words
    ststhip_sweep_desc desc;
    // the form is chosen for a run of generations [first, last]; a later run may move the window
    void set_run(std::uint64_t iteration_offset, std::uint64_t n_iterations) {
        uniform_call.first_iteration = iteration_offset;
        uniform_call.last_iteration = iteration_offset + n_iterations - 1;
    }
};

// `r` must stay where it is afterwards (its context pointers point into it).  `dom` may be replaced by r.words.
int resolve_app(ResolvedApp &r, const char *app, const void *tf_params, const void *halo_cell,
                const ststhip_domain *&dom, const void *const *src, void *const *dst, bool have_planes) {
    const AppEntry *e = find_app(app);
    if (!e)
        return fail(STSTHIP_ERR_UNKNOWN_APP, "unknown transition function");
    // Jacobi5General with five equal positive coefficients and a +0 halo: bit-identical results from the
    // product-carrying form with 5 instead of 9 flops per cell (apps/jacobi.hpp, Jacobi5Uniform)
    if (std::strcmp(app, "jacobi5general") == 0 && opt().jacobi_fastpath) {
        const ststhip_jacobi_params *jp = static_cast<const ststhip_jacobi_params *>(tf_params);
        std::uint32_t halo_bits;
        std::memcpy(&halo_bits, halo_cell, sizeof halo_bits);
        bool same = jp->coef[0] > 0.0f && halo_bits == 0u;
        for (int i = 1; i < 5 && same; i++)
            same = std::memcmp(&jp->coef[i], &jp->coef[0], sizeof(float)) == 0;
        UniformJacobiCall &u = r.uniform_call;
        u.middle = find_app("jacobi5uniform");
        u.first = find_app("jacobi5uniform_first");
        u.last = find_app("jacobi5uniform_last");
        u.only = find_app("jacobi5uniform_only");
        if (same && u.middle && u.first && u.last && u.only) {
            u.c = jp->coef[0];
            u.halo_cell = halo_cell;
            r.uniform = true;
            e = u.middle; // same geometry for all four variants
        }
    }
    // Game of Life with a dead halo on a grid whose width and pitch are multiples of four cells: the same
    // rule on 32-bit words of four cells (apps/conway.hpp, ConwayPacked), swept as a grid of words
    if (std::strcmp(app, "conway") == 0 && dom->local_cols == 0 && dom->global_width > 0 && dom->global_width % 4 == 0 &&
        dom->pitch % 4 == 0 && *static_cast<const unsigned char *>(halo_cell) == 0 &&
        (!have_planes || (reinterpret_cast<std::uintptr_t>(src[0]) % 4 == 0 &&
                          reinterpret_cast<std::uintptr_t>(dst[0]) % 4 == 0)) &&
        opt().conway_fastpath) {
        if (const AppEntry *packed = find_app("conway_packed")) {
            e = packed;
            r.words = *dom;
            r.words.global_width = dom->global_width / 4;
            r.words.pitch = dom->pitch / 4;
            dom = &r.words;
            r.packed = true;
        }
    }
    r.entry = e;
    std::memset(&r.desc, 0, sizeof r.desc);
    r.desc.n_planes = e->info.n_planes;
    r.desc.max_generations = e->info.max_generations;
    r.desc.halo_depth_per_generation = e->info.halo_depth_per_generation;
    r.desc.strip_width = e->info.strip_width;
    for (unsigned p = 0; p < e->info.n_planes; p++)
        r.desc.plane_elem_size[p] = e->info.plane_elem_size[p];
    // a family compiled deeper than its rule trusts: the unmeasured depth, and the key its measurements are kept under
    if (e->info.default_generations > 0 && e->info.default_generations < e->info.max_generations) {
        r.desc.alt_generations = e->info.default_generations;
        r.desc.tune_key = reinterpret_cast<std::uintptr_t>(e);
    }
    if (r.uniform) {
        r.trampoline = uniform_jacobi_trampoline;
        r.ctx = &r.uniform_call;
    } else {
        r.call = AppCall{e, tf_params, r.packed ? static_cast<const void *>(&r.dead_word) : halo_cell};
        r.trampoline = app_sweep_trampoline;
        r.ctx = &r.call;
        if (e->fill_tdv) { // one device table of time-dependent values per call
            r.desc.tdv_size = e->info.tdv_size;
            r.desc.fill_tdv = app_fill_tdv;
        }
    }
    return STSTHIP_OK;
}
} // namespace

int ststhip_app_run(const char *app, const void *tf_params, const void *halo_cell,
                    const ststhip_domain *dom, const void *const *src, void *const *dst,
                    uint64_t iteration_offset, uint64_t n_iterations, int blocking, int profiling,
                    ststhip_stream stream, ststhip_run_info *info) {
    const AppEntry *e = find_app(app);
    if (!e)
        return fail(STSTHIP_ERR_UNKNOWN_APP, "unknown transition function");
    if (!tf_params || !halo_cell || !src || !dst || !dom)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    if (dom->pitch < dom->global_width || dom->local_cols != 0)
        return fail(STSTHIP_ERR_INVALID, "ststhip_app_run works on whole grids: pitch >= width, no column range");
    for (unsigned p = 0; p < e->info.n_planes; p++)
        if (!src[p] || !dst[p] || src[p] == dst[p])
            return fail(STSTHIP_ERR_INVALID, "source and target planes must be distinct non-null buffers");
    ResolvedApp r;
    if (n_iterations == 0) { // a copy: no form to choose
        ststhip_sweep_desc desc;
        std::memset(&desc, 0, sizeof desc);
        desc.n_planes = e->info.n_planes;
        desc.max_generations = e->info.max_generations;
        desc.halo_depth_per_generation = e->info.halo_depth_per_generation;
        desc.strip_width = e->info.strip_width;
        for (unsigned p = 0; p < e->info.n_planes; p++)
            desc.plane_elem_size[p] = e->info.plane_elem_size[p];
        AppCall call{e, tf_params, halo_cell};
        return ststhip_run_passes(app_sweep_trampoline, &call, &desc, dom, src, dst, iteration_offset, 0, blocking,
                                  profiling, stream, info);
    }
    if (int rc = resolve_app(r, app, tf_params, halo_cell, dom, src, dst, true))
        return rc;
    r.set_run(iteration_offset, n_iterations);
    const int rc = ststhip_run_passes(r.trampoline, r.ctx, &r.desc, dom, src, dst, iteration_offset, n_iterations,
                                      blocking, profiling, stream, info);
    if (r.packed && info)
        info->n_processed_cells *= 4; // counted in words by the pass driver
    return rc;
}

// ------------------------------------------------------------------ multi-GPU
int ststhip_comm_unique_id(unsigned char id[STSTHIP_COMM_ID_BYTES]) {
    if (!id)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    if (int rc = load_rccl())
        return rc;
    NCCL_TRY(rccl().GetUniqueId(id));
    return STSTHIP_OK;
}

int ststhip_comm_create(const unsigned char id[STSTHIP_COMM_ID_BYTES], int rank, int n_ranks,
                        ststhip_comm *comm) {
    if (!id || !comm || rank < 0 || rank >= n_ranks)
        return fail(STSTHIP_ERR_INVALID, "bad communicator arguments");
    if (int rc = ststhip_init(-1))
        return rc;
    if (int rc = load_rccl())
        return rc;
    Id128 uid;
    std::memcpy(uid.bytes, id, sizeof uid.bytes);
    Comm *c = new Comm;
    c->rank = rank;
    c->n_ranks = n_ranks;
    c->up = rank - 1; // -1 for the first strip
    c->down = rank + 1 < n_ranks ? rank + 1 : -1;
    int rc = rccl().CommInitRank(&c->nccl, n_ranks, uid, rank);
    if (rc != 0) {
        delete c;
        return nccl_fail(rc, "ncclCommInitRank");
    }
    *comm = c;
    return STSTHIP_OK;
}

int ststhip_comm_destroy(ststhip_comm comm) {
    Comm *c = static_cast<Comm *>(comm);
    if (!c)
        return STSTHIP_OK;
    if (c->nccl)
        rccl().CommDestroy(c->nccl);
    delete c;
    return STSTHIP_OK;
}

int ststhip_comm_set_neighbours(ststhip_comm comm, int up, int down) {
    Comm *c = static_cast<Comm *>(comm);
    if (!c || up < -1 || down < -1 || up >= c->n_ranks || down >= c->n_ranks)
        return fail(STSTHIP_ERR_INVALID, "neighbours must be ranks of the communicator, or -1");
    c->up = up;
    c->down = down;
    return STSTHIP_OK;
}

int ststhip_comm_neighbours(ststhip_comm comm, int *up, int *down) {
    Comm *c = static_cast<Comm *>(comm);
    if (!c)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    if (up)
        *up = c->up;
    if (down)
        *down = c->down;
    return STSTHIP_OK;
}

// Grouped send / receive with the two neighbours of one direction (`first`: the rank whose rows lie above / whose
// columns lie to the left; `second`: below / to the right).
// RCCL matches the messages between two ranks in the order they were posted.  Per plane: both sends, then the receive
// from `second`, then the one from `first` -- so that when both name the SAME rank (two blocks of a ring, or a rank
// that is its own neighbour: the one-GPU loopback of tests/test_strip_native_gpu.py) that rank's first-side data land
// in the ghost cells on the receiver's second side and vice versa.  In a chain two ranks share one message per plane
// and direction and any order would do.
static int exchange_with(Comm *c, int first, int second, int n_planes, const void *const *send_first,
                         const void *const *send_second, void *const *recv_first, void *const *recv_second,
                         const size_t *bytes_per_unit, size_t n_units, hipStream_t s) {
    const int ncclChar = 0;
    const bool has_first = first >= 0, has_second = second >= 0;
    if ((has_first && (!send_first || !recv_first)) || (has_second && (!send_second || !recv_second)))
        return fail(STSTHIP_ERR_INVALID, "bad exchange arguments");
    NCCL_TRY(rccl().GroupStart());
    int failed = 0; // a group that was opened is closed again whatever happens inside
    const char *what = "";
    auto post = [&](int code, const char *call) {
        if (code != 0 && failed == 0) {
            failed = code;
            what = call;
        }
    };
    for (int p = 0; p < n_planes && failed == 0; p++) {
        const size_t bytes = bytes_per_unit[p] * n_units;
        if (has_first)
            post(rccl().Send(send_first[p], bytes, ncclChar, first, c->nccl, s), "ncclSend (up / left)");
        if (has_second)
            post(rccl().Send(send_second[p], bytes, ncclChar, second, c->nccl, s), "ncclSend (down / right)");
        if (has_second)
            post(rccl().Recv(recv_second[p], bytes, ncclChar, second, c->nccl, s), "ncclRecv (down / right)");
        if (has_first)
            post(rccl().Recv(recv_first[p], bytes, ncclChar, first, c->nccl, s), "ncclRecv (up / left)");
    }
    const int ended = rccl().GroupEnd();
    if (failed != 0)
        return nccl_fail(failed, what);
    if (ended != 0)
        return nccl_fail(ended, "ncclGroupEnd");
    return STSTHIP_OK;
}

int ststhip_comm_exchange_rows(ststhip_comm comm, int n_planes, const void *const *send_up,
                               const void *const *send_down, void *const *recv_up,
                               void *const *recv_down, const size_t *row_bytes, size_t n_rows,
                               ststhip_stream stream) {
    Comm *c = static_cast<Comm *>(comm);
    if (!c || n_planes < 1 || n_planes > 16 || !row_bytes)
        return fail(STSTHIP_ERR_INVALID, "bad exchange arguments");
    if (n_rows == 0)
        return STSTHIP_OK;
    return exchange_with(c, c->up, c->down, n_planes, send_up, send_down, recv_up, recv_down, row_bytes, n_rows,
                         resolve(stream));
}

int ststhip_comm_set_column_neighbours(ststhip_comm comm, int left, int right) {
    Comm *c = static_cast<Comm *>(comm);
    if (!c || left < -1 || right < -1 || left >= c->n_ranks || right >= c->n_ranks)
        return fail(STSTHIP_ERR_INVALID, "neighbours must be ranks of the communicator, or -1");
    c->left = left;
    c->right = right;
    return STSTHIP_OK;
}

int ststhip_comm_set_mesh(ststhip_comm comm, int mesh_rows, int mesh_cols) {
    Comm *c = static_cast<Comm *>(comm);
    if (!c || mesh_rows < 1 || mesh_cols < 1 || mesh_rows * mesh_cols != c->n_ranks)
        return fail(STSTHIP_ERR_INVALID, "the mesh must have exactly the communicator's ranks");
    const int r = c->rank / mesh_cols, k = c->rank % mesh_cols;
    c->up = r > 0 ? c->rank - mesh_cols : -1;
    c->down = r + 1 < mesh_rows ? c->rank + mesh_cols : -1;
    c->left = k > 0 ? c->rank - 1 : -1;
    c->right = k + 1 < mesh_cols ? c->rank + 1 : -1;
    return STSTHIP_OK;
}

int ststhip_comm_exchange_columns(ststhip_comm comm, int n_planes, const void *const *send_left,
                                  const void *const *send_right, void *const *recv_left, void *const *recv_right,
                                  const size_t *block_bytes, ststhip_stream stream) {
    Comm *c = static_cast<Comm *>(comm);
    if (!c || n_planes < 1 || n_planes > 16 || !block_bytes)
        return fail(STSTHIP_ERR_INVALID, "bad exchange arguments");
    return exchange_with(c, c->left, c->right, n_planes, send_left, send_right, recv_left, recv_right, block_bytes, 1,
                         resolve(stream));
}


// ------------------------------------------------------------------ row-strip driver (one strip per process / GPU)
namespace {
struct Strip {
    std::string app;
    std::vector<unsigned char> params, halo;
    ResolvedApp resolved;
    int rank = 0, n_ranks = 1;
    Comm *comm = nullptr;
    ststhip_exchange_fn exchange = nullptr;
    void *exchange_ctx = nullptr;
    std::uint64_t total_rows = 0, width = 0, row_begin = 0, row_end = 0, g_max = 0, local_rows = 0;
    std::uint64_t ghost = 0;   // ghost rows the buffers hold above and below the owned rows: g_max * exchange_every
    int exchange_every = 1;    // launches per ghost exchange (fixed at creation: it sizes the buffers)
    std::int64_t row_origin = 0;
    unsigned n_planes = 0;
    std::size_t elem[16] = {0};
    void *planes[2][16] = {{nullptr}};
    int current = 0;
    hipStream_t compute = nullptr, comm_stream = nullptr;
    std::vector<hipStream_t> band; // the boundary bands' stream (highest priority), created when first needed
    hipStream_t side = nullptr;    // the lower of two sub-strips with a moving boundary (ststhip_strip_advance), created when first needed
    ststhip_domain dom;        // geometry of the buffers (in words for the packed Game of Life)
    std::uint64_t n_launches = 0, n_exchanges = 0;
    // a block of a 2-D decomposition (ststhip_block_create): a column range and ghost columns as well
    bool is_block = false;
    int mesh_rows = 1, mesh_cols = 1, mesh_r = 0, mesh_c = 0;
    std::uint64_t total_cols = 0, col_begin = 0, col_end = 0, ghost_cols = 0, local_cols = 0;
    std::int64_t col_origin = 0;
    ststhip_exchange_fn exchange_cols = nullptr;
    void *exchange_cols_ctx = nullptr;
    void *stage[4][16] = {{nullptr}}; // packed ghost columns per plane: to the left, to the right, from the left, from the right
};

void strip_bounds(std::uint64_t total, int n, int r, std::uint64_t &a, std::uint64_t &b) {
    // as even as possible, earlier ranks take the remainder (stencilstream_amd/dist.py: split_rows)
    const std::uint64_t base = total / n, extra = total % n;
    a = base * r + std::min<std::uint64_t>(r, extra);
    b = a + base + (std::uint64_t(r) < extra ? 1 : 0);
}

// ghost rows of depth g next to the owned rows of buffer set `set`, on the comm stream
int strip_exchange(Strip &st, int set, std::uint64_t g) {
    if (st.n_ranks == 1 || g == 0)
        return STSTHIP_OK;
    const void *send_up[16], *send_down[16];
    void *recv_up[16], *recv_down[16];
    std::size_t row_bytes[16];
    const std::uint64_t o_start = st.ghost, o_stop = st.ghost + (st.row_end - st.row_begin);
    for (unsigned p = 0; p < st.n_planes; p++) {
        row_bytes[p] = std::size_t(st.dom.pitch) * st.elem[p];
        unsigned char *base = static_cast<unsigned char *>(st.planes[set][p]);
        send_up[p] = base + o_start * row_bytes[p];
        recv_up[p] = base + (o_start - g) * row_bytes[p];
        send_down[p] = base + (o_stop - g) * row_bytes[p];
        recv_down[p] = base + o_stop * row_bytes[p];
    }
    st.n_exchanges++;
    if (st.comm)
        return ststhip_comm_exchange_rows(st.comm, int(st.n_planes), send_up, send_down, recv_up, recv_down, row_bytes,
                                          std::size_t(g), st.comm_stream);
    return st.exchange(st.exchange_ctx, int(st.n_planes), send_up, send_down, recv_up, recv_down, row_bytes,
                       std::size_t(g), st.comm_stream);
}
// the 2-D block driver (below)
extern "C++" {
int block_exchange(Strip &st, int set, std::uint64_t g);
int block_advance(Strip *st, std::uint64_t iteration_offset, std::uint64_t n_generations, int blocking);
}
} // namespace

namespace {
// the part of strip creation that does not depend on where the sweep comes from: geometry, streams, buffers
int finish_strip(Strip *st, const ststhip_domain *dom, ststhip_strip *strip) {
    const ststhip_sweep_desc &d = st->resolved.desc;
    if (d.n_planes == 0 || d.n_planes > 16 || d.max_generations == 0) {
        delete st;
        return fail(STSTHIP_ERR_INVALID, "bad sweep description");
    }
    st->n_planes = d.n_planes;
    // (the deepest launch the strip driver plans: the family's trusted depth, ststhip_strip_advance)
    st->g_max = std::uint64_t(d.alt_generations && d.alt_generations < d.max_generations ? d.alt_generations : d.max_generations) *
                d.halo_depth_per_generation;
    std::uint64_t thinnest = st->total_rows;
    for (int r = 0; r < st->n_ranks; r++) {
        std::uint64_t a, b;
        strip_bounds(st->total_rows, st->n_ranks, r, a, b);
        thinnest = std::min(thinnest, b - a);
    }
    // launches per ghost exchange (STSTHIP_EXCHANGE_EVERY, else by the strips' height): a thin strip's launch is as
    // short as the band -> exchange chain in front of the next one, so it groups four launches per exchange, a thick
    // one two (one MI355X, the compute side of a rank, profiles/r03_thin_strips.txt: 2048 rows 3910 / 4010 / 4090
    // Gcell/s per GPU at 1 / 2 / 4 launches per exchange)
    st->exchange_every = 1;
    if (st->n_ranks > 1) {
        int every = opt().exchange_every > 0 ? std::min(opt().exchange_every, 16) : (thinnest <= 4096 ? 4 : 2);
        while (every > 1 && thinnest < 4 * st->g_max * std::uint64_t(every))
            every--;
        st->exchange_every = every;
    }
    st->ghost = st->g_max * std::uint64_t(st->exchange_every);
    if (st->n_ranks > 1 && thinnest < 2 * st->ghost) {
        delete st;
        return fail(STSTHIP_ERR_INVALID, "strips are thinner than two ghost depths: use fewer ranks, a larger grid or a "
                                         "smaller STSTHIP_EXCHANGE_EVERY");
    }
    st->row_origin = std::int64_t(st->row_begin) - std::int64_t(st->ghost);
    st->local_rows = (st->row_end - st->row_begin) + 2 * st->ghost;
    st->dom = *dom; // width and pitch possibly in words
    st->dom.row_origin = st->row_origin;
    st->dom.local_rows = st->local_rows;
    int rc = STSTHIP_OK;
    hipError_t err = hipStreamCreateWithFlags(&st->compute, hipStreamNonBlocking);
    // the exchange stream has normal priority: as a third highest-priority stream beside the band streams of two
    // sub-strips it cost 8-30 % (streams of one priority share few hardware queues, and a band waiting for its events
    // holds up whatever sits behind it in the same queue: profiles/r02_thin_strips.txt section 7)
    if (err == hipSuccess)
        err = opt().comm_stream_priority
                  ? create_band_stream(&st->comm_stream)
                  : hipStreamCreateWithFlags(&st->comm_stream, hipStreamNonBlocking);
    for (int set = 0; set < 2 && err == hipSuccess && rc == STSTHIP_OK; set++)
        for (unsigned p = 0; p < st->n_planes && rc == STSTHIP_OK; p++) {
            st->elem[p] = d.plane_elem_size[p];
            const std::size_t bytes = std::size_t(st->local_rows) * st->dom.pitch * st->elem[p];
            rc = ststhip_malloc_async(&st->planes[set][p], bytes, st->compute);
            if (rc == STSTHIP_OK)
                err = hipMemsetAsync(st->planes[set][p], 0, bytes, st->compute);
        }
    if (err != hipSuccess)
        rc = hip_fail(err, "strip set-up");
    if (rc == STSTHIP_OK && (err = hipStreamSynchronize(st->compute)) != hipSuccess)
        rc = hip_fail(err, "strip set-up");
    if (rc != STSTHIP_OK) {
        ststhip_strip_destroy(st);
        return rc;
    }
    *strip = st;
    return STSTHIP_OK;
}
} // namespace

int ststhip_strip_create(const char *app, const void *tf_params, const void *halo_cell, uint64_t total_rows,
                         uint64_t width, int rank, int n_ranks, ststhip_comm comm, ststhip_exchange_fn exchange,
                         void *exchange_ctx, ststhip_strip *strip) {
    const AppEntry *e = find_app(app);
    if (!e)
        return fail(STSTHIP_ERR_UNKNOWN_APP, "unknown transition function");
    if (!tf_params || !halo_cell || !strip || n_ranks < 1 || rank < 0 || rank >= n_ranks || width == 0)
        return fail(STSTHIP_ERR_INVALID, "bad strip arguments");
    if (n_ranks > 1 && !comm && !exchange)
        return fail(STSTHIP_ERR_INVALID, "several strips need a communicator or an exchange callback");
    if (int rc = ststhip_init(-1))
        return rc;
    Strip *st = new Strip;
    st->app = app;
    st->params.assign(static_cast<const unsigned char *>(tf_params),
                      static_cast<const unsigned char *>(tf_params) + std::max<std::uint32_t>(e->info.params_size, 1));
    st->halo.assign(static_cast<const unsigned char *>(halo_cell),
                    static_cast<const unsigned char *>(halo_cell) + e->info.cell_size);
    st->rank = rank;
    st->n_ranks = n_ranks;
    st->comm = static_cast<Comm *>(comm);
    st->exchange = exchange;
    st->exchange_ctx = exchange_ctx;
    st->total_rows = total_rows;
    st->width = width;
    strip_bounds(total_rows, n_ranks, rank, st->row_begin, st->row_end);
    ststhip_domain whole = {};
    whole.global_height = total_rows;
    whole.global_width = width;
    whole.pitch = width;
    whole.row_origin = 0;
    whole.local_rows = total_rows;
    const ststhip_domain *dom = &whole;
    int rc = resolve_app(st->resolved, st->app.c_str(), st->params.data(), st->halo.data(), dom, nullptr, nullptr, false);
    if (rc != STSTHIP_OK) {
        delete st;
        return rc;
    }
    return finish_strip(st, dom, strip);
}

int ststhip_strip_create_custom(ststhip_sweep_fn sweep, void *ctx, const ststhip_sweep_desc *desc, uint64_t total_rows,
                                uint64_t width, int rank, int n_ranks, ststhip_comm comm, ststhip_exchange_fn exchange,
                                void *exchange_ctx, ststhip_strip *strip) {
    if (!sweep || !desc || !strip || n_ranks < 1 || rank < 0 || rank >= n_ranks || width == 0)
        return fail(STSTHIP_ERR_INVALID, "bad strip arguments");
    if (n_ranks > 1 && !comm && !exchange)
        return fail(STSTHIP_ERR_INVALID, "several strips need a communicator or an exchange callback");
    if (int rc = ststhip_init(-1))
        return rc;
    Strip *st = new Strip;
    st->rank = rank;
    st->n_ranks = n_ranks;
    st->comm = static_cast<Comm *>(comm);
    st->exchange = exchange;
    st->exchange_ctx = exchange_ctx;
    st->total_rows = total_rows;
    st->width = width;
    strip_bounds(total_rows, n_ranks, rank, st->row_begin, st->row_end);
    st->resolved.entry = nullptr; // the caller's sweep: no registry entry, no run window to maintain
    st->resolved.trampoline = sweep;
    st->resolved.ctx = ctx;
    st->resolved.desc = *desc;
    ststhip_domain whole = {};
    whole.global_height = total_rows;
    whole.global_width = width;
    whole.pitch = width;
    whole.row_origin = 0;
    whole.local_rows = total_rows;
    return finish_strip(st, &whole, strip);
}

int ststhip_strip_destroy(ststhip_strip strip) {
    Strip *st = static_cast<Strip *>(strip);
    if (!st)
        return STSTHIP_OK;
    if (st->compute)
        (void)hipStreamSynchronize(st->compute);
    if (st->comm_stream)
        (void)hipStreamSynchronize(st->comm_stream);
    for (hipStream_t lane : st->band) {
        (void)hipStreamSynchronize(lane);
        (void)hipStreamDestroy(lane);
    }
    if (st->side) {
        (void)hipStreamSynchronize(st->side);
        (void)hipStreamDestroy(st->side);
    }
    for (auto &set : st->planes)
        for (void *plane : set)
            if (plane)
                ststhip_free(plane);
    for (auto &side : st->stage)
        for (void *buffer : side)
            if (buffer)
                ststhip_free(buffer);
    if (st->compute)
        (void)hipStreamDestroy(st->compute);
    if (st->comm_stream)
        (void)hipStreamDestroy(st->comm_stream);
    delete st;
    return STSTHIP_OK;
}

int ststhip_strip_rows(ststhip_strip strip, uint64_t *row_begin, uint64_t *row_end) {
    Strip *st = static_cast<Strip *>(strip);
    if (!st || !row_begin || !row_end)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    *row_begin = st->row_begin;
    *row_end = st->row_end;
    return STSTHIP_OK;
}

int ststhip_strip_plane(ststhip_strip strip, unsigned plane, void **owned_rows, size_t *row_bytes) {
    Strip *st = static_cast<Strip *>(strip);
    if (!st || plane >= st->n_planes || !owned_rows)
        return fail(STSTHIP_ERR_INVALID, "bad plane index or null argument");
    const std::size_t bytes = std::size_t(st->dom.pitch) * st->elem[plane];
    *owned_rows = static_cast<unsigned char *>(st->planes[st->current][plane]) + st->ghost * bytes;
    if (row_bytes)
        *row_bytes = bytes;
    return STSTHIP_OK;
}

int ststhip_strip_stream(ststhip_strip strip, ststhip_stream *stream) {
    Strip *st = static_cast<Strip *>(strip);
    if (!st || !stream)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    *stream = st->compute;
    return STSTHIP_OK;
}

int ststhip_strip_synchronize(ststhip_strip strip) {
    Strip *st = static_cast<Strip *>(strip);
    if (!st)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    HIP_TRY(hipStreamSynchronize(st->compute));
    return STSTHIP_OK;
}

int ststhip_strip_counters(ststhip_strip strip, uint64_t *n_launches, uint64_t *n_exchanges) {
    Strip *st = static_cast<Strip *>(strip);
    if (!st)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    if (n_launches)
        *n_launches = st->n_launches;
    if (n_exchanges)
        *n_exchanges = st->n_exchanges;
    return STSTHIP_OK;
}

int ststhip_strip_warm_up(ststhip_strip strip) {
    Strip *st = static_cast<Strip *>(strip);
    if (!st)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    if (st->n_ranks == 1)
        return STSTHIP_OK;
    HIP_TRY(hipStreamSynchronize(st->compute));
    if (int rc = st->is_block ? block_exchange(*st, st->current, st->g_max) : strip_exchange(*st, st->current, st->g_max))
        return rc;
    HIP_TRY(hipStreamSynchronize(st->comm_stream));
    return STSTHIP_OK;
}

// One call = n_generations generations of the whole distributed grid (every rank calls it with the same arguments).
//
// The launches of a call go in GROUPS of m = exchange_every launches with ONE ghost exchange per group.  Before a
// group the neighbours exchange G = the sum of the group's halo depths g_j rows; launch j of the group then produces
// the owned rows widened by E_j = g_(j+1) + ... + g_(m-1) on every side that has a neighbour (both ranks compute those
// rows: g*m*(m-1) redundant rows per group and boundary, 2 % of a 2048-row strip at m = 4, T = 12), so the next launch
// finds its halo without a message.  Only the LAST launch of a group is split: the G' rows next to each neighbour
// (G' = ghost depth of the next group) run as ONE band launch with a row hole on a highest-priority stream, the
// exchange for the next group follows it on the comm stream, and the interior runs beside both.  Per group: one band
// launch and one exchange, however many launches it has -- for thin strips, where the band -> exchange -> band chain
// of every launch was as long as the interior (m = 1 is the scheme of round 2: bands and exchange at every launch).
// The owned rows are NOT cut into sub-strips any more: with the staged sweep two sub-strips on two streams measured
// 3-20 % slower than one at 2048 ... 8192 rows, with band launches at their common boundary (round 2's scheme) and
// with redundant halos instead (profiles/r03_thin_strips.txt).
int ststhip_strip_advance(ststhip_strip strip, uint64_t iteration_offset, uint64_t n_generations, int blocking) {
    Strip *st = static_cast<Strip *>(strip);
    if (!st)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    if (st->is_block)
        return block_advance(st, iteration_offset, n_generations, blocking);
    const ststhip_sweep_desc &d = st->resolved.desc;
    // (every rank must plan the same launches, so nothing is measured here: a family with a second depth runs the one
    // its rule trusts)
    const std::vector<std::uint32_t> depths =
        plan_depths(n_generations, d.max_generations, d.alt_generations < d.max_generations ? d.alt_generations : 0);
    if (depths.empty())
        return STSTHIP_OK;
    st->resolved.set_run(iteration_offset, n_generations);
    const std::uint64_t hpg = d.halo_depth_per_generation;
    const std::uint64_t a = st->row_begin, b = st->row_end;
    const std::size_t m = std::size_t(st->exchange_every);
    int rc = STSTHIP_OK;
    EventPool events;
    auto ordered = [&](hipError_t err, const char *what) {
        if (err != hipSuccess && rc == STSTHIP_OK)
            rc = hip_fail(err, what);
    };
    auto record = [&](hipStream_t on) {
        hipEvent_t ev = events.take();
        if (!ev)
            ordered(hipErrorUnknown, "hipEventCreateWithFlags");
        else
            ordered(hipEventRecord(ev, on), "hipEventRecord");
        return ev;
    };
    auto wait = [&](hipStream_t who, hipEvent_t ev) {
        if (ev)
            ordered(hipStreamWaitEvent(who, ev, 0), "hipStreamWaitEvent");
    };

    const bool has_up = st->rank > 0, has_down = st->rank + 1 < st->n_ranks;
    hipStream_t lane = st->compute, band = nullptr;
    if (has_up || has_down) {
        // (created when first needed: streams are dealt onto a few hardware queues in creation order, idle ones included)
        if (st->band.empty()) {
            hipStream_t high;
            if (create_band_stream(&high) != hipSuccess)
                return hip_fail(hipErrorUnknown, "hipStreamCreateWithPriority");
            st->band.push_back(high);
        }
        band = st->band[0];
    }
    // Two sub-strips with a moving boundary (as the pass driver's strips, ststhip_run_passes): the owned rows are swept
    // as an upper and a lower launch per pass, on two streams; their common boundary moves up by the larger of this
    // launch's ghost rows and the ones of the launch before, so the upper sub-strip's launches never wait for the lower
    // one's and two launches are in flight -- a launch of a thin strip is too small to fill the chip on its own
    // (2048 x 16384 cells: +15 % for the same cells as two strips, profiles/r04_skewed_strips.txt).  The boundary bands
    // at the strip's ends and the exchange stay as they are; the boundary keeps moving across the groups of launches.
    hipStream_t lane_low = nullptr;
    std::uint64_t sub_span = 0;
    {
        for (std::size_t i = 0; i < depths.size(); i++)
            sub_span += std::uint64_t(std::max(depths[i], i ? depths[i - 1] : 0u)) * hpg;
        sub_span = std::min<std::uint64_t>(sub_span, (b - a) / 4);
        const int wanted = opt().strip_substrips < 0
                               ? suggest_row_strips(b - a, st->width, d.strip_width, st->g_max, depths.size())
                               : opt().strip_substrips;
        const bool usable = wanted >= 2 && opt().skewed_strips != 0 && depths.size() >= 2 && sub_span >= 2 * st->g_max &&
                            (b - a) >= 2 * (st->ghost + 2 * st->g_max) + 2 * sub_span + 64;
        if (usable) {
            if (!st->side && hipStreamCreateWithFlags(&st->side, hipStreamNonBlocking) != hipSuccess)
                return hip_fail(hipErrorUnknown, "hipStreamCreateWithFlags");
            lane_low = st->side;
        }
    }
    std::uint64_t sub_offset = 0, sub_g_before = 0; // how far below its highest place the boundary is
    hipEvent_t upper_done = nullptr, lower_done = nullptr;
    g_launch_concurrency = lane_low ? 2 : 1;

    // one device table of time-dependent values for the whole call, as in ststhip_run_passes
    void *tdv_table = nullptr;
    if (d.tdv_size > 0 && d.fill_tdv) {
        const std::size_t bytes = std::size_t(d.tdv_size) * n_generations;
        rc = ststhip_malloc_async(&tdv_table, bytes, lane);
        if (rc == STSTHIP_OK) {
            std::vector<unsigned char> values(bytes);
            d.fill_tdv(st->resolved.ctx, iteration_offset, n_generations, values.data());
            ordered(hipMemcpyAsync(tdv_table, values.data(), bytes, hipMemcpyHostToDevice, lane), "hipMemcpyAsync");
            ordered(hipStreamSynchronize(lane), "hipStreamSynchronize"); // `values` is pageable and goes out of scope
            g_tdv_table = tdv_table;
            g_tdv_first = iteration_offset;
            g_tdv_count = n_generations;
            g_tdv_size = d.tdv_size;
        }
    }
    auto group_depth = [&](std::size_t first) {
        std::uint64_t sum = 0;
        for (std::size_t i = first; i < std::min(first + m, depths.size()); i++)
            sum += depths[i] * hpg;
        return sum;
    };
    hipEvent_t begin = record(lane); // everything queued so far: a previous advance, uploads, the table
    wait(st->comm_stream, begin);
    if (band)
        wait(band, begin);
    if (lane_low)
        wait(lane_low, begin);
    hipEvent_t ghosts_ready = nullptr;
    if (st->n_ranks > 1 && rc == STSTHIP_OK) {
        rc = strip_exchange(*st, st->current, group_depth(0));
        ghosts_ready = record(st->comm_stream);
    }
    std::uint64_t iteration = iteration_offset;
    for (std::size_t first = 0; first < depths.size() && rc == STSTHIP_OK; first += m) {
        const std::size_t last = std::min(first + m, depths.size()) - 1;
        wait(lane, ghosts_ready); // the group's first launch reads the ghost rows of this group
        if (lane_low)
            wait(lane_low, ghosts_ready);
        std::uint64_t widen = group_depth(first); // E_(j-1): how far beyond the owned rows the launch's input is valid
        for (std::size_t i = first; i <= last && rc == STSTHIP_OK; i++) {
            const std::uint32_t depth = depths[i];
            widen -= depth * hpg; // E_j
            const void *const *src = const_cast<const void *const *>(st->planes[st->current]);
            void *const *dst = st->planes[st->current ^ 1];
            const std::uint64_t lo = has_up ? a - std::min(widen, a) : a;
            const std::uint64_t hi = has_down ? std::min(b + widen, st->total_rows) : b;
            auto sweep = [&](std::uint64_t r0, std::uint64_t r1, hipStream_t on) {
                if (r0 < r1 && rc == STSTHIP_OK) {
                    rc = st->resolved.trampoline(st->resolved.ctx, &st->dom, src, dst, r0, r1, iteration, depth, on);
                    st->n_launches++;
                }
            };
            // rows [r0, r1) as one launch, or as the two sub-strips' launches
            auto sweep_rows = [&](std::uint64_t r0, std::uint64_t r1) {
                if (!lane_low) {
                    sweep(r0, r1, lane);
                    return;
                }
                const std::uint64_t shift = std::max<std::uint64_t>(depth * hpg, sub_g_before);
                bool restart = false;
                if (sub_offset + shift > sub_span) {
                    sub_offset = 0;
                    restart = true;
                } else {
                    sub_offset += shift;
                }
                sub_g_before = depth * hpg;
                const std::uint64_t boundary = std::min(std::max((a + b) / 2 + sub_span / 2 - sub_offset, r0), r1);
                if (restart)
                    wait(lane, lower_done);
                sweep(r0, boundary, lane);
                hipEvent_t upper = record(lane);
                wait(lane_low, upper_done);
                sweep(boundary, r1, lane_low);
                lower_done = record(lane_low);
                upper_done = upper;
            };
            const bool feeds_exchange = i == last && last + 1 < depths.size() && (has_up || has_down);
            if (feeds_exchange) {
                const std::uint64_t next = group_depth(last + 1);
                const std::uint64_t top_end = has_up ? a + next : a;
                const std::uint64_t bot_begin = has_down ? b - next : b;
                wait(band, record(lane)); // the bands read what the previous launch left (and what `lane` waited for)
                if (lane_low)
                    wait(band, record(lane_low));
                if (has_up && has_down && top_end < bot_begin) {
                    // both bands as one launch with a hole where the interior is: one band latency in front of the exchange
                    g_row_hole_begin = top_end;
                    g_row_hole_end = bot_begin;
                    sweep(a, b, band);
                    g_row_hole_begin = g_row_hole_end = 0;
                } else {
                    sweep(a, top_end, band);
                    sweep(bot_begin, b, band);
                }
                hipEvent_t banded = record(band);
                if (st->n_ranks > 1 && rc == STSTHIP_OK) {
                    wait(st->comm_stream, banded);
                    rc = strip_exchange(*st, st->current ^ 1, next);
                    ghosts_ready = record(st->comm_stream);
                }
                sweep_rows(top_end, bot_begin);
                wait(lane, banded); // the next launch reads the bands' rows
                if (lane_low)
                    wait(lane_low, banded);
            } else {
                sweep_rows(lo, hi);
            }
            st->current ^= 1;
            iteration += depth;
        }
    }
    g_tdv_table = nullptr;
    g_tdv_count = 0;
    // the compute stream is the one callers synchronise with
    if (band)
        wait(lane, record(band));
    if (lane_low)
        wait(lane, record(lane_low));
    wait(lane, record(st->comm_stream));
    g_launch_concurrency = 1;
    if (rc == STSTHIP_OK && blocking) {
        hipError_t err = hipStreamSynchronize(lane);
        if (err != hipSuccess)
            rc = hip_fail(err, "hipStreamSynchronize");
    }
    if (tdv_table)
        ststhip_free_async(tdv_table, lane); // every stream has been joined into the compute stream above
    return rc; // the events go back to the pool (EventPool)
}


// ------------------------------------------------------------------ 2-D block driver (one block per process / GPU)
// The minimal form of the reference's tile geometry (StencilStream/tiling/Grid.hpp:305-450: a tile with its halo from the
// neighbouring tiles; tiling/StencilUpdate.hpp:216-247: one pass over the tiles per group of generations) on a mesh of
// GPUs: ghost columns are strided in memory, so they travel packed; corners arrive by exchanging the columns first and
// then the rows over the full buffer width.  No overlap of exchange and interior here (ststhip.h).
} // extern "C"

namespace {
// rows x width rectangle between pitched buffers, in units of U (4-byte words where everything is aligned, else bytes)
template <typename U>
__global__ void __launch_bounds__(256) copy_rect_kernel(U *dst, std::size_t dst_pitch, const U *src, std::size_t src_pitch,
                                                        unsigned width, unsigned rows) {
    const std::size_t i = blockIdx.x * std::size_t(256) + threadIdx.x;
    if (i >= std::size_t(width) * rows)
        return;
    const std::size_t r = i / width, c = i % width;
    dst[r * dst_pitch + c] = src[r * src_pitch + c];
}

int copy_rect(void *dst, std::size_t dst_pitch_bytes, const void *src, std::size_t src_pitch_bytes, std::size_t width_bytes,
              std::size_t rows, hipStream_t stream) {
    if (width_bytes == 0 || rows == 0)
        return STSTHIP_OK;
    const bool words = (reinterpret_cast<std::uintptr_t>(dst) | reinterpret_cast<std::uintptr_t>(src) | dst_pitch_bytes |
                        src_pitch_bytes | width_bytes) % 4 == 0;
    const std::size_t unit = words ? 4 : 1;
    const std::size_t n = width_bytes / unit * rows;
    if (n >= (1ull << 32) || width_bytes / unit >= (1ull << 32))
        return fail(STSTHIP_ERR_INVALID, "ghost-column block too large");
    const unsigned blocks = unsigned((n + 255) / 256);
    if (words)
        hipLaunchKernelGGL(copy_rect_kernel<std::uint32_t>, dim3(blocks), dim3(256), 0, stream, static_cast<std::uint32_t *>(dst),
                           dst_pitch_bytes / 4, static_cast<const std::uint32_t *>(src), src_pitch_bytes / 4,
                           unsigned(width_bytes / 4), unsigned(rows));
    else
        hipLaunchKernelGGL(copy_rect_kernel<unsigned char>, dim3(blocks), dim3(256), 0, stream, static_cast<unsigned char *>(dst),
                           dst_pitch_bytes, static_cast<const unsigned char *>(src), src_pitch_bytes, unsigned(width_bytes),
                           unsigned(rows));
    HIP_TRY(hipGetLastError());
    return STSTHIP_OK;
}

// ghost cells of depth g around the owned block of buffer set `set`, on the comm stream: columns (packed), then rows
int block_exchange(Strip &st, int set, std::uint64_t g) {
    if (g == 0 || (st.mesh_rows == 1 && st.mesh_cols == 1))
        return STSTHIP_OK;
    const bool has_left = st.mesh_c > 0, has_right = st.mesh_c + 1 < st.mesh_cols;
    const bool has_up = st.mesh_r > 0, has_down = st.mesh_r + 1 < st.mesh_rows;
    const std::uint64_t owned_rows = st.row_end - st.row_begin, owned_cols = st.col_end - st.col_begin;
    hipStream_t s = st.comm_stream;
    st.n_exchanges++;
    if (has_left || has_right) {
        const void *send_left[16], *send_right[16];
        void *recv_left[16], *recv_right[16];
        std::size_t bytes[16];
        for (unsigned p = 0; p < st.n_planes; p++) {
            const std::size_t pitch = std::size_t(st.dom.pitch) * st.elem[p], width = std::size_t(g) * st.elem[p];
            unsigned char *first_owned_row = static_cast<unsigned char *>(st.planes[set][p]) + st.ghost * pitch;
            bytes[p] = width * owned_rows;
            send_left[p] = st.stage[0][p];
            send_right[p] = st.stage[1][p];
            recv_left[p] = st.stage[2][p];
            recv_right[p] = st.stage[3][p];
            if (has_left)
                if (int rc = copy_rect(st.stage[0][p], width, first_owned_row + st.ghost_cols * st.elem[p], pitch, width, owned_rows, s))
                    return rc;
            if (has_right)
                if (int rc = copy_rect(st.stage[1][p], width, first_owned_row + (st.ghost_cols + owned_cols - g) * st.elem[p], pitch,
                                       width, owned_rows, s))
                    return rc;
        }
        int rc = st.comm ? ststhip_comm_exchange_columns(st.comm, int(st.n_planes), send_left, send_right, recv_left, recv_right,
                                                          bytes, s)
                         : st.exchange_cols(st.exchange_cols_ctx, int(st.n_planes), send_left, send_right, recv_left, recv_right,
                                            bytes, 1, s);
        if (rc != STSTHIP_OK)
            return rc;
        for (unsigned p = 0; p < st.n_planes; p++) {
            const std::size_t pitch = std::size_t(st.dom.pitch) * st.elem[p], width = std::size_t(g) * st.elem[p];
            unsigned char *first_owned_row = static_cast<unsigned char *>(st.planes[set][p]) + st.ghost * pitch;
            if (has_left)
                if (int rc2 = copy_rect(first_owned_row + (st.ghost_cols - g) * st.elem[p], pitch, st.stage[2][p], width, width, owned_rows, s))
                    return rc2;
            if (has_right)
                if (int rc2 = copy_rect(first_owned_row + (st.ghost_cols + owned_cols) * st.elem[p], pitch, st.stage[3][p], width, width,
                                        owned_rows, s))
                    return rc2;
        }
    }
    if (has_up || has_down) {
        // whole buffer rows, ghost columns included: what the left and right neighbours have just delivered goes on to
        // the blocks above and below -- their corners
        const void *send_up[16], *send_down[16];
        void *recv_up[16], *recv_down[16];
        std::size_t row_bytes[16];
        const std::uint64_t o_start = st.ghost, o_stop = st.ghost + owned_rows;
        for (unsigned p = 0; p < st.n_planes; p++) {
            row_bytes[p] = std::size_t(st.dom.pitch) * st.elem[p];
            unsigned char *base = static_cast<unsigned char *>(st.planes[set][p]);
            send_up[p] = base + o_start * row_bytes[p];
            recv_up[p] = base + (o_start - g) * row_bytes[p];
            send_down[p] = base + (o_stop - g) * row_bytes[p];
            recv_down[p] = base + o_stop * row_bytes[p];
        }
        if (st.comm)
            return ststhip_comm_exchange_rows(st.comm, int(st.n_planes), send_up, send_down, recv_up, recv_down, row_bytes,
                                              std::size_t(g), s);
        return st.exchange(st.exchange_ctx, int(st.n_planes), send_up, send_down, recv_up, recv_down, row_bytes, std::size_t(g), s);
    }
    return STSTHIP_OK;
}

int block_advance(Strip *st, std::uint64_t iteration_offset, std::uint64_t n_generations, int blocking) {
    const ststhip_sweep_desc &d = st->resolved.desc;
    const std::vector<std::uint32_t> depths =
        plan_depths(n_generations, d.max_generations, d.alt_generations < d.max_generations ? d.alt_generations : 0);
    if (depths.empty())
        return STSTHIP_OK;
    st->resolved.set_run(iteration_offset, n_generations);
    const std::uint64_t hpg = d.halo_depth_per_generation;
    const std::size_t m = std::size_t(st->exchange_every);
    int rc = STSTHIP_OK;
    EventPool events;
    auto ordered = [&](hipError_t err, const char *what) {
        if (err != hipSuccess && rc == STSTHIP_OK)
            rc = hip_fail(err, what);
    };
    auto record = [&](hipStream_t on) {
        hipEvent_t ev = events.take();
        if (!ev)
            ordered(hipErrorUnknown, "hipEventCreateWithFlags");
        else
            ordered(hipEventRecord(ev, on), "hipEventRecord");
        return ev;
    };
    auto wait = [&](hipStream_t who, hipEvent_t ev) {
        if (ev)
            ordered(hipStreamWaitEvent(who, ev, 0), "hipStreamWaitEvent");
    };
    hipStream_t lane = st->compute;
    g_launch_concurrency = 1;
    void *tdv_table = nullptr;
    if (d.tdv_size > 0 && d.fill_tdv) {
        const std::size_t bytes = std::size_t(d.tdv_size) * n_generations;
        rc = ststhip_malloc_async(&tdv_table, bytes, lane);
        if (rc == STSTHIP_OK) {
            std::vector<unsigned char> values(bytes);
            d.fill_tdv(st->resolved.ctx, iteration_offset, n_generations, values.data());
            ordered(hipMemcpyAsync(tdv_table, values.data(), bytes, hipMemcpyHostToDevice, lane), "hipMemcpyAsync");
            ordered(hipStreamSynchronize(lane), "hipStreamSynchronize"); // `values` is pageable and goes out of scope
            g_tdv_table = tdv_table;
            g_tdv_first = iteration_offset;
            g_tdv_count = n_generations;
            g_tdv_size = d.tdv_size;
        }
    }
    auto group_depth = [&](std::size_t first) {
        std::uint64_t sum = 0;
        for (std::size_t i = first; i < std::min(first + m, depths.size()); i++)
            sum += depths[i] * hpg;
        return sum;
    };
    const bool has_left = st->mesh_c > 0, has_right = st->mesh_c + 1 < st->mesh_cols;
    const bool has_up = st->mesh_r > 0, has_down = st->mesh_r + 1 < st->mesh_rows;
    const bool alone = st->mesh_rows * st->mesh_cols == 1;
    wait(st->comm_stream, record(lane)); // everything queued so far: a previous advance, uploads, the table
    hipEvent_t ghosts_ready = nullptr;
    if (!alone && rc == STSTHIP_OK) {
        rc = block_exchange(*st, st->current, group_depth(0));
        ghosts_ready = record(st->comm_stream);
    }
    std::uint64_t iteration = iteration_offset;
    for (std::size_t first = 0; first < depths.size() && rc == STSTHIP_OK; first += m) {
        const std::size_t last = std::min(first + m, depths.size()) - 1;
        wait(lane, ghosts_ready);
        std::uint64_t widen = group_depth(first);
        for (std::size_t i = first; i <= last && rc == STSTHIP_OK; i++) {
            const std::uint32_t depth = depths[i];
            widen -= depth * hpg; // how far beyond the owned block this launch still has to produce
            const std::uint64_t lo = has_up ? st->row_begin - std::min(widen, st->row_begin) : st->row_begin;
            const std::uint64_t hi = has_down ? std::min(st->row_end + widen, st->total_rows) : st->row_end;
            const std::uint64_t clo = has_left ? st->col_begin - std::min(widen, st->col_begin) : st->col_begin;
            const std::uint64_t chi = has_right ? std::min(st->col_end + widen, st->total_cols) : st->col_end;
            g_col_begin = clo;
            g_col_end = chi;
            rc = st->resolved.trampoline(st->resolved.ctx, &st->dom, const_cast<const void *const *>(st->planes[st->current]),
                                         st->planes[st->current ^ 1], lo, hi, iteration, depth, lane);
            g_col_begin = g_col_end = 0;
            st->n_launches++;
            st->current ^= 1;
            iteration += depth;
        }
        if (last + 1 < depths.size() && !alone && rc == STSTHIP_OK) {
            wait(st->comm_stream, record(lane));
            rc = block_exchange(*st, st->current, group_depth(last + 1));
            ghosts_ready = record(st->comm_stream);
        }
    }
    g_tdv_table = nullptr;
    g_tdv_count = 0;
    wait(lane, record(st->comm_stream));
    if (rc == STSTHIP_OK && blocking) {
        hipError_t err = hipStreamSynchronize(lane);
        if (err != hipSuccess)
            rc = hip_fail(err, "hipStreamSynchronize");
    }
    if (tdv_table)
        ststhip_free_async(tdv_table, lane);
    return rc;
}
} // namespace

extern "C" {

extern "C++" {
namespace {
// where a block lies in the mesh and whom it talks to
void place_block(Strip *st, std::uint64_t total_rows, std::uint64_t total_cols, int rank, int mesh_rows, int mesh_cols,
                 ststhip_comm comm, ststhip_exchange_fn exchange_rows, void *exchange_rows_ctx,
                 ststhip_exchange_fn exchange_cols, void *exchange_cols_ctx) {
    st->is_block = true;
    st->rank = rank;
    st->n_ranks = mesh_rows * mesh_cols;
    st->mesh_rows = mesh_rows;
    st->mesh_cols = mesh_cols;
    st->mesh_r = rank / mesh_cols;
    st->mesh_c = rank % mesh_cols;
    st->comm = static_cast<Comm *>(comm);
    st->exchange = exchange_rows;
    st->exchange_ctx = exchange_rows_ctx;
    st->exchange_cols = exchange_cols;
    st->exchange_cols_ctx = exchange_cols_ctx;
    st->total_rows = total_rows;
    st->total_cols = st->width = total_cols;
    strip_bounds(total_rows, mesh_rows, st->mesh_r, st->row_begin, st->row_end);
    strip_bounds(total_cols, mesh_cols, st->mesh_c, st->col_begin, st->col_end);
}

// ghost depths, buffers and streams of a block whose sweep (st->resolved) is known; owns `st` from here on
int finish_block(Strip *st, ststhip_strip *block) {
    const std::uint64_t total_rows = st->total_rows, total_cols = st->total_cols;
    const int mesh_rows = st->mesh_rows, mesh_cols = st->mesh_cols;
    int rc = STSTHIP_OK;
    const ststhip_sweep_desc &d = st->resolved.desc;
    st->n_planes = d.n_planes;
    st->g_max = std::uint64_t(d.alt_generations && d.alt_generations < d.max_generations ? d.alt_generations : d.max_generations) *
                d.halo_depth_per_generation;
    std::uint64_t thinnest = std::min(total_rows / mesh_rows, total_cols / mesh_cols);
    int every = 1;
    if (st->n_ranks > 1) {
        every = opt().exchange_every > 0 ? std::min(opt().exchange_every, 16) : 2;
        while (every > 1 && thinnest < 4 * st->g_max * std::uint64_t(every))
            every--;
    }
    st->exchange_every = every;
    st->ghost = st->g_max * std::uint64_t(every);
    if (st->n_ranks > 1 && thinnest < 2 * st->ghost) {
        delete st;
        return fail(STSTHIP_ERR_INVALID, "blocks are thinner than two ghost depths: use a smaller mesh, a larger grid or a "
                                         "smaller STSTHIP_EXCHANGE_EVERY");
    }
    // ghost columns: the exchanged depth, and a few more so that the strips next to the block's sides run the check-free
    // code (a wave's footprint is rounded up to whole lanes); the extra columns are never exchanged and never matter
    st->ghost_cols = (st->ghost + 8 + 3) / 4 * 4;
    st->row_origin = std::int64_t(st->row_begin) - std::int64_t(st->ghost);
    st->local_rows = (st->row_end - st->row_begin) + 2 * st->ghost;
    st->col_origin = std::int64_t(st->col_begin) - std::int64_t(st->ghost_cols);
    st->local_cols = (st->col_end - st->col_begin) + 2 * st->ghost_cols;
    st->dom = ststhip_domain{};
    st->dom.global_height = total_rows;
    st->dom.global_width = total_cols;
    st->dom.row_origin = st->row_origin;
    st->dom.local_rows = st->local_rows;
    st->dom.pitch = st->local_cols;
    st->dom.col_origin = st->col_origin;
    st->dom.local_cols = st->local_cols;
    hipError_t err = hipStreamCreateWithFlags(&st->compute, hipStreamNonBlocking);
    if (err == hipSuccess)
        err = hipStreamCreateWithFlags(&st->comm_stream, hipStreamNonBlocking);
    for (int set = 0; set < 2 && err == hipSuccess && rc == STSTHIP_OK; set++)
        for (unsigned p = 0; p < st->n_planes && rc == STSTHIP_OK; p++) {
            st->elem[p] = d.plane_elem_size[p];
            const std::size_t bytes = std::size_t(st->local_rows) * st->local_cols * st->elem[p];
            rc = ststhip_malloc_async(&st->planes[set][p], bytes, st->compute);
            if (rc == STSTHIP_OK)
                err = hipMemsetAsync(st->planes[set][p], 0, bytes, st->compute);
        }
    if (st->mesh_cols > 1)
        for (int side = 0; side < 4 && rc == STSTHIP_OK; side++)
            for (unsigned p = 0; p < st->n_planes && rc == STSTHIP_OK; p++)
                rc = ststhip_malloc_async(&st->stage[side][p], std::size_t(st->row_end - st->row_begin) * st->ghost * st->elem[p],
                                          st->compute);
    if (err != hipSuccess)
        rc = hip_fail(err, "block set-up");
    if (rc == STSTHIP_OK && (err = hipStreamSynchronize(st->compute)) != hipSuccess)
        rc = hip_fail(err, "block set-up");
    if (rc != STSTHIP_OK) {
        ststhip_strip_destroy(st);
        return rc;
    }
    *block = st;
    return STSTHIP_OK;
}
} // namespace
} // extern "C++"

int ststhip_block_create(const char *app, const void *tf_params, const void *halo_cell, uint64_t total_rows,
                         uint64_t total_cols, int rank, int mesh_rows, int mesh_cols, ststhip_comm comm,
                         ststhip_exchange_fn exchange_rows, void *exchange_rows_ctx,
                         ststhip_exchange_fn exchange_cols, void *exchange_cols_ctx, ststhip_strip *block) {
    const AppEntry *e = find_app(app);
    if (!e)
        return fail(STSTHIP_ERR_UNKNOWN_APP, "unknown transition function");
    if (!tf_params || !halo_cell || !block || mesh_rows < 1 || mesh_cols < 1 || rank < 0 || rank >= mesh_rows * mesh_cols ||
        total_rows == 0 || total_cols == 0)
        return fail(STSTHIP_ERR_INVALID, "bad block arguments");
    if (!comm && ((mesh_rows > 1 && !exchange_rows) || (mesh_cols > 1 && !exchange_cols)))
        return fail(STSTHIP_ERR_INVALID, "several blocks need a communicator or exchange callbacks for rows and columns");
    if (int rc = ststhip_init(-1))
        return rc;
    Strip *st = new Strip;
    st->app = app;
    st->params.assign(static_cast<const unsigned char *>(tf_params),
                      static_cast<const unsigned char *>(tf_params) + std::max<std::uint32_t>(e->info.params_size, 1));
    st->halo.assign(static_cast<const unsigned char *>(halo_cell), static_cast<const unsigned char *>(halo_cell) + e->info.cell_size);
    place_block(st, total_rows, total_cols, rank, mesh_rows, mesh_cols, comm, exchange_rows, exchange_rows_ctx, exchange_cols,
                exchange_cols_ctx);
    ststhip_domain whole = {};
    whole.global_height = total_rows;
    whole.global_width = total_cols;
    whole.pitch = total_cols;
    whole.local_rows = total_rows;
    whole.local_cols = total_cols; // (a block: the sweeps run on cells, never on the packed words of the Game of Life)
    const ststhip_domain *dom = &whole;
    int rc = resolve_app(st->resolved, st->app.c_str(), st->params.data(), st->halo.data(), dom, nullptr, nullptr, false);
    if (rc != STSTHIP_OK) {
        delete st;
        return rc;
    }
    return finish_block(st, block);
}

int ststhip_block_create_custom(ststhip_sweep_fn sweep, void *ctx, const ststhip_sweep_desc *desc, uint64_t total_rows,
                                uint64_t total_cols, int rank, int mesh_rows, int mesh_cols, ststhip_comm comm,
                                ststhip_exchange_fn exchange_rows, void *exchange_rows_ctx,
                                ststhip_exchange_fn exchange_cols, void *exchange_cols_ctx, ststhip_strip *block) {
    if (!sweep || !desc || !block || mesh_rows < 1 || mesh_cols < 1 || rank < 0 || rank >= mesh_rows * mesh_cols ||
        total_rows == 0 || total_cols == 0)
        return fail(STSTHIP_ERR_INVALID, "bad block arguments");
    if (!comm && ((mesh_rows > 1 && !exchange_rows) || (mesh_cols > 1 && !exchange_cols)))
        return fail(STSTHIP_ERR_INVALID, "several blocks need a communicator or exchange callbacks for rows and columns");
    if (int rc = ststhip_init(-1))
        return rc;
    Strip *st = new Strip;
    place_block(st, total_rows, total_cols, rank, mesh_rows, mesh_cols, comm, exchange_rows, exchange_rows_ctx, exchange_cols,
                exchange_cols_ctx);
    st->resolved.entry = nullptr; // the caller's sweep: no registry entry, no run window to maintain
    st->resolved.trampoline = sweep;
    st->resolved.ctx = ctx;
    st->resolved.desc = *desc;
    return finish_block(st, block);
}

int ststhip_block_geometry(ststhip_strip block, uint64_t *row_begin, uint64_t *row_end, uint64_t *col_begin,
                           uint64_t *col_end) {
    Strip *st = static_cast<Strip *>(block);
    if (!st)
        return fail(STSTHIP_ERR_INVALID, "null argument");
    if (row_begin)
        *row_begin = st->row_begin;
    if (row_end)
        *row_end = st->row_end;
    if (col_begin)
        *col_begin = st->is_block ? st->col_begin : 0;
    if (col_end)
        *col_end = st->is_block ? st->col_end : st->width;
    return STSTHIP_OK;
}

static int block_copy(Strip *st, unsigned plane, void *host, size_t host_pitch_bytes, bool upload) {
    if (!st || plane >= st->n_planes || !host)
        return fail(STSTHIP_ERR_INVALID, "bad plane index or null argument");
    const std::size_t e = st->elem[plane], pitch = std::size_t(st->dom.pitch) * e;
    const std::size_t ghost_cols = st->is_block ? st->ghost_cols : 0;
    const std::size_t owned_cols = st->is_block ? st->col_end - st->col_begin : st->dom.global_width;
    unsigned char *owned = static_cast<unsigned char *>(st->planes[st->current][plane]) + st->ghost * pitch + ghost_cols * e;
    if (host_pitch_bytes < owned_cols * e)
        return fail(STSTHIP_ERR_INVALID, "host pitch smaller than the block's row");
    HIP_TRY(upload ? hipMemcpy2DAsync(owned, pitch, host, host_pitch_bytes, owned_cols * e, st->row_end - st->row_begin,
                                      hipMemcpyHostToDevice, st->compute)
                   : hipMemcpy2DAsync(host, host_pitch_bytes, owned, pitch, owned_cols * e, st->row_end - st->row_begin,
                                      hipMemcpyDeviceToHost, st->compute));
    HIP_TRY(hipStreamSynchronize(st->compute));
    return STSTHIP_OK;
}
int ststhip_block_upload(ststhip_strip block, unsigned plane, const void *host_cells, size_t host_pitch_bytes) {
    return block_copy(static_cast<Strip *>(block), plane, const_cast<void *>(host_cells), host_pitch_bytes, true);
}
int ststhip_block_download(ststhip_strip block, unsigned plane, void *host_cells, size_t host_pitch_bytes) {
    return block_copy(static_cast<Strip *>(block), plane, host_cells, host_pitch_bytes, false);
}

} // extern "C"
