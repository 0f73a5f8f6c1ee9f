// stencil::hip::Grid -- the grid container of the MI355X backend.
//
// Interface parity with StencilStream/cuda/Grid.hpp:50-188 (== cpu/Grid.hpp): constructors from
// (rows, cols), sycl::range<2> and sycl::buffer<Cell,2>; copies are shared handles (:97);
// copy_from_buffer / copy_to_buffer throw std::range_error on an extent mismatch (:109-134);
// GridAccessor<mode> gives ac[id<2>] and ac[r][c] (:145-153); getters (:158-176); make_similar.
//
// Storage: row-major AoS cells in HBM (canonical copy, produced and consumed by the sweep
// kernels) plus a lazily created pinned host mirror.  Validity flags make the transfers implicit,
// which is what the reference gets from SYCL buffers: constructing a GridAccessor brings the host
// mirror up to date (device -> host), and a non-read accessor invalidates the device copy so the
// next StencilUpdate uploads it again.
#pragma once
#include "../Concepts.hpp"
#include "internal/Runtime.hpp"

#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

namespace stencil {
namespace hip {

template <typename Cell> class Grid {
    static_assert(std::is_trivially_copyable_v<Cell>,
                  "cells are moved between host and HBM as raw bytes");

    struct Storage {
        sycl::range<2> extent;
        Cell *host = nullptr;
        void *device = nullptr;
        bool host_valid = false;
        bool device_valid = false;
        // an upload queued by start_upload() that the host has not waited for yet: the pinned mirror must not be
        // written (or freed) before it has been read
        ststhip_event upload_event = nullptr;
        bool upload_pending = false;
        // events of an upload in row blocks (start_upload_in_blocks): per block "copied" and "ready" (the kernel a host
        // runs behind the copy has finished); `work_pending`: the last "ready" has not been waited for
        std::vector<ststhip_event> block_events;
        ststhip_event last_ready = nullptr;
        bool work_pending = false;

        explicit Storage(sycl::range<2> extent) : extent(extent) {}
        Storage(Storage const &) = delete;
        ~Storage() {
            if (upload_pending)
                ststhip_event_synchronize(upload_event);
            if (work_pending)
                ststhip_event_synchronize(last_ready); // (a kernel that still reads the device cells)
            if (upload_event)
                ststhip_event_destroy(upload_event);
            for (ststhip_event ev : block_events)
                ststhip_event_destroy(ev);
            if (host)
                ststhip_host_free(host);
            if (device)
                ststhip_free(device);
        }
        std::size_t bytes() const { return extent.size() * sizeof(Cell); }

        // `overwritten`: the caller fills every cell right away (a download), so the
        // value-initialisation a fresh grid owes its cells can be skipped
        void need_host(bool overwritten = false) {
            if (!host) {
                internal::ensure_runtime(-1);
                host = static_cast<Cell *>(internal::pinned_alloc(bytes()));
                if (!overwritten)
                    for (std::size_t i = 0; i < extent.size(); i++)
                        new (host + i) Cell();
            }
        }
        void need_device() {
            if (!device) {
                internal::ensure_runtime(-1);
                device = internal::device_alloc_on(bytes(), internal::default_stream());
            }
        }
        void sync_to_host() {
            need_host(!host_valid && device_valid);
            if (!host_valid && device_valid) {
                ststhip_stream s = internal::default_stream();
                internal::check(ststhip_memcpy_d2h(host, device, bytes(), s), "grid download");
                internal::check(ststhip_stream_synchronize(s), "grid download");
            }
            host_valid = true;
        }
        void wait_upload() {
            if (upload_pending) {
                internal::check(ststhip_event_synchronize(upload_event), "grid upload");
                upload_pending = false;
            }
            if (work_pending) {
                internal::check(ststhip_event_synchronize(last_ready), "grid upload");
                work_pending = false;
            }
        }
        // The device copy is up to date for work queued on `s` after this call; the host does not wait for the
        // transfer (allocating the buffers of the update that follows takes as long as a good part of it).
        void start_upload(ststhip_stream s) {
            need_device();
            if (!device_valid) {
                need_host();
                internal::check(ststhip_memcpy_h2d(device, host, bytes(), s), "grid upload");
                if (!upload_event)
                    internal::check(ststhip_event_create(&upload_event), "grid upload");
                internal::check(ststhip_event_record(upload_event, s), "grid upload");
                upload_pending = true;
                host_valid = true;
            }
            device_valid = true;
        }
        // The same in row blocks, first rows first, on the runtime's upload streams: `blocks` names them for the pass
        // driver (ststhip_set_source_arrival), which starts on the rows that have arrived instead of idling for the
        // whole transfer (1 GiB: 18.7 ms over PCIe).  `after_block(work_stream, first_row, end_row)` may queue a
        // kernel per block in front of its event (the scatter into per-field planes; on a stream of its own, behind
        // the block's copy: on the copies' stream it would hold up the next transfer); `has_work` says whether it does.
        // Nothing to upload, or a grid the rule does not split: `blocks` stays empty and the call is start_upload(s).
        template <typename AfterBlock>
        void start_upload_in_blocks(ststhip_stream s, std::vector<ststhip_source_block> &blocks, bool has_work,
                                    AfterBlock &&after_block) {
            blocks.clear();
            std::uint32_t n_blocks = 1;
            if (!device_valid)
                internal::check(ststhip_suggest_upload_blocks(extent[0], extent[1] * sizeof(Cell), &n_blocks), "grid upload");
            if (n_blocks < 2) {
                start_upload(s);
                return;
            }
            need_device();
            need_host();
            ststhip_stream up = nullptr, work = nullptr;
            internal::check(ststhip_upload_streams(&up, &work), "grid upload");
            if (!upload_event)
                internal::check(ststhip_event_create(&upload_event), "grid upload");
            // What `s` has queued so far comes first (buffers from the pool may still be in use by work queued there).
            // The HOST waits for it -- `s` is idle when a grid is uploaded, as a rule --, not the copies' stream: the
            // first copy of a process that is queued behind another stream's event costs a hipMemcpyAsync that does not
            // return for 7.6 ms, with the chip idle behind it (tools/microbench/upload_stall.hip, variants 8 / 9).
            internal::check(ststhip_event_record(upload_event, s), "grid upload");
            internal::check(ststhip_event_synchronize(upload_event), "grid upload");
            while (block_events.size() < 2 * std::size_t(n_blocks)) {
                ststhip_event ev = nullptr;
                internal::check(ststhip_event_create(&ev), "grid upload");
                block_events.push_back(ev);
            }
            const std::size_t rows = extent[0], row_bytes = extent[1] * sizeof(Cell);
            for (std::uint32_t b = 0; b < n_blocks; b++) {
                const std::size_t first = rows * b / n_blocks, end = rows * (b + 1) / n_blocks;
                internal::check(ststhip_memcpy_h2d(static_cast<char *>(device) + first * row_bytes,
                                                   reinterpret_cast<char const *>(host) + first * row_bytes,
                                                   (end - first) * row_bytes, up),
                                "grid upload");
                ststhip_event copied = block_events[2 * b], ready = copied;
                internal::check(ststhip_event_record(copied, up), "grid upload");
                if (has_work) {
                    ready = block_events[2 * b + 1];
                    internal::check(ststhip_stream_wait_event(work, copied), "grid upload");
                    after_block(work, first, end);
                    internal::check(ststhip_event_record(ready, work), "grid upload");
                }
                blocks.push_back(ststhip_source_block{end, ready});
                last_ready = ready;
            }
            work_pending = has_work;
            // (what wait_upload() waits for: the pinned mirror has been read)
            internal::check(ststhip_event_record(upload_event, up), "grid upload");
            upload_pending = true;
            host_valid = true;
            device_valid = true;
        }
        // ... and for any stream, and the pinned mirror may be rewritten by the host right after this call
        void sync_to_device() {
            start_upload(internal::default_stream());
            wait_upload();
        }
    };

  public:
    static constexpr std::size_t dimensions = 2;

    Grid(std::size_t n_rows, std::size_t n_columns)
        : storage(std::make_shared<Storage>(sycl::range<2>(n_rows, n_columns))) {}
    Grid(sycl::range<2> extent) : storage(std::make_shared<Storage>(extent)) {}
    Grid(sycl::buffer<Cell, 2> source) : Grid(source.get_range()) { copy_from_buffer(source); }
    Grid(Grid const &) = default; // shares the cells
    Grid &operator=(Grid const &) = default;

    void copy_from_buffer(sycl::buffer<Cell, 2> source) {
        require_same_extent(source.get_range());
        storage->wait_upload();
        storage->need_host(/*overwritten=*/true);
        std::memcpy(static_cast<void *>(storage->host), source.data(), storage->bytes());
        storage->host_valid = true;
        storage->device_valid = false;
    }

    void copy_to_buffer(sycl::buffer<Cell, 2> target) {
        require_same_extent(target.get_range());
        storage->sync_to_host();
        std::memcpy(static_cast<void *>(target.data()), storage->host, storage->bytes());
    }

    template <sycl::access::mode access_mode = sycl::access::mode::read_write> class GridAccessor {
        static constexpr bool read_only = (access_mode == sycl::access::mode::read);
        using Ref = std::conditional_t<read_only, Cell const &, Cell &>;
        using Ptr = std::conditional_t<read_only, Cell const *, Cell *>;

      public:
        GridAccessor(Grid &grid) : keep_alive(grid.storage), width(grid.get_grid_width()) {
            keep_alive->sync_to_host();
            if constexpr (!read_only) {
                keep_alive->wait_upload(); // the mirror is about to be written
                keep_alive->device_valid = false;
            }
            cells = keep_alive->host;
        }
        Ref operator[](sycl::id<2> at) const { return cells[at[0] * width + at[1]]; }
        Ptr operator[](std::size_t row) const { return cells + row * width; }
        sycl::range<2> get_range() const { return keep_alive->extent; }
        Ptr get_pointer() const { return cells; }
        std::size_t byte_size() const { return keep_alive->bytes(); }

      private:
        std::shared_ptr<Storage> keep_alive;
        std::size_t width;
        Ptr cells;
    };

    std::size_t get_grid_height() const { return storage->extent[0]; }
    std::size_t get_grid_width() const { return storage->extent[1]; }
    sycl::range<2> get_grid_range() const { return storage->extent; }
    Grid make_similar() const { return Grid(storage->extent); }

    // ---- used by stencil::hip::StencilUpdate and by device-side tooling ----
    // AoS cells in HBM, uploaded from the host mirror if that is newer.
    Cell const *device_cells() {
        storage->sync_to_device();
        return static_cast<Cell const *>(storage->device);
    }
    // The same for work queued on `stream` after this call: the upload is queued there and not waited for.
    Cell const *device_cells_on(ststhip_stream stream) {
        storage->start_upload(stream);
        return static_cast<Cell const *>(storage->device);
    }
    // Where the AoS cells lie in HBM, whatever they hold (for work that is ordered against an upload by other means).
    Cell const *device_cells_base() {
        storage->need_device();
        return static_cast<Cell const *>(storage->device);
    }
    // The same with the upload in row blocks the pass driver can follow (Storage::start_upload_in_blocks): `blocks` is
    // what to hand to ststhip_set_source_arrival right before the update's ststhip_run_passes -- empty when there is
    // nothing to follow, and then everything is ordered on `stream` as with device_cells_on.
    template <typename AfterBlock>
    Cell const *device_cells_arriving(ststhip_stream stream, std::vector<ststhip_source_block> &blocks, bool has_work,
                                      AfterBlock &&after_block) {
        storage->start_upload_in_blocks(stream, blocks, has_work, after_block);
        return static_cast<Cell const *>(storage->device);
    }
    // AoS cells in HBM about to be overwritten completely by a kernel on the runtime's stream.
    Cell *device_cells_for_overwrite() {
        storage->wait_upload();
        storage->need_device();
        storage->device_valid = true;
        storage->host_valid = false;
        return static_cast<Cell *>(storage->device);
    }

  private:
    void require_same_extent(sycl::range<2> other) const {
        if (other != storage->extent)
            throw std::range_error("The target buffer has not the same size as the grid");
    }
    std::shared_ptr<Storage> storage;
};

} // namespace hip
} // namespace stencil
