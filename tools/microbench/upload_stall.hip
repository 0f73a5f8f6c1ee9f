// Where does a host block while it queues an upload in row blocks?  (round 4: one hipMemcpyAsync of eight did not return
// for 7.6 ms in the example binaries, never from Python with torch loaded.)  Plain HIP, no library.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/upload_stall.hip -o build/ab/upload_stall && build/ab/upload_stall
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const size_t n = size_t(1) << 30;
    const int blocks = 8;
    hipStream_t main_stream, up;
    hipStreamCreateWithFlags(&main_stream, hipStreamNonBlocking);
    int least = 0, greatest = 0;
    hipDeviceGetStreamPriorityRange(&least, &greatest);
    hipStreamCreateWithPriority(&up, hipStreamNonBlocking, greatest);
    void *host = nullptr, *dev = nullptr, *small_h = nullptr, *small_d = nullptr;
    hipHostMalloc(&host, n, hipHostMallocDefault);
    std::memset(host, 1, n);
    hipHostMalloc(&small_h, 1 << 20, hipHostMallocDefault);
    hipMalloc(&small_d, 1 << 20);
    hipMemcpyAsync(small_d, small_h, 1 << 20, hipMemcpyHostToDevice, main_stream);
    hipMemcpyAsync(small_h, small_d, 1 << 20, hipMemcpyDeviceToHost, main_stream);
    hipStreamSynchronize(main_stream);
    if (variant == 3) { // warm the upload stream with a LARGE copy from another buffer
        void *h2 = nullptr, *d2 = nullptr;
        hipHostMalloc(&h2, 256 << 20, hipHostMallocDefault);
        std::memset(h2, 1, 256 << 20);
        hipMalloc(&d2, 256 << 20);
        hipMemcpyAsync(d2, h2, 128 << 20, hipMemcpyHostToDevice, up);
        hipMemcpyAsync((char *)d2 + (128 << 20), (char *)h2 + (128 << 20), 128 << 20, hipMemcpyHostToDevice, up);
        hipStreamSynchronize(up);
    }
    void *other[2] = {nullptr, nullptr};
    if (variant >= 4) { // the buffers an update allocates in front of the upload: result grid, swap planes
        hipMalloc(&other[0], n);
        hipMalloc(&other[1], n);
    }
    hipMalloc(&dev, n);
    if (variant == 5) { // ... and some time between the allocations and the copies
        const double until = now_ms() + 15.0;
        while (now_ms() < until) {
        }
    }
    if (variant == 6) // ... or the allocations touched first
        for (void *p : {other[0], other[1], dev})
            hipMemsetAsync(p, 0, 256, main_stream);
    if (variant == 6)
        hipStreamSynchronize(main_stream);
    for (int rep = 0; rep < 4; rep++) {
        std::vector<hipEvent_t> ev(blocks);
        for (auto &e : ev)
            hipEventCreate(&e);
        const double t0 = now_ms();
        std::vector<double> queued, arrived;
        hipStream_t on = variant == 2 ? main_stream : up;
        if (variant == 9) { // ... the host waits instead
            hipEvent_t first;
            hipEventCreate(&first);
            hipMemsetAsync(dev, 0, 256, main_stream);
            hipEventRecord(first, main_stream);
            hipEventSynchronize(first);
        }
        if (variant == 7 || variant == 8) { // the copies' stream waits for what the main stream has queued
            hipEvent_t first;
            hipEventCreate(&first);
            if (variant == 8)
                hipMemsetAsync(dev, 0, 256, main_stream);
            hipEventRecord(first, main_stream);
            hipStreamWaitEvent(on, first, 0);
        }
        for (int b = 0; b < blocks; b++) {
            hipMemcpyAsync((char *)dev + b * (n / blocks), (char *)host + b * (n / blocks), n / blocks, hipMemcpyHostToDevice, on);
            hipEventRecord(ev[b], on);
            queued.push_back(now_ms() - t0);
            if (variant == 1) { // one copy in flight at a time
                hipEventSynchronize(ev[b]);
                arrived.push_back(now_ms() - t0);
            }
        }
        if (variant != 1)
            for (int b = 0; b < blocks; b++) {
                hipEventSynchronize(ev[b]);
                arrived.push_back(now_ms() - t0);
            }
        std::printf("variant %d rep %d: queued at", variant, rep);
        for (double q : queued)
            std::printf(" %.2f", q);
        std::printf(" ms; arrived at");
        for (double a : arrived)
            std::printf(" %.2f", a);
        std::printf(" ms\n");
    }
    return 0;
}
