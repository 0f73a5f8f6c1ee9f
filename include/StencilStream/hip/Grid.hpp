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

        explicit Storage(sycl::range<2> extent) : extent(extent) {}
        Storage(Storage const &) = delete;
        ~Storage() {
            if (upload_pending)
                ststhip_event_synchronize(upload_event);
            if (upload_event)
                ststhip_event_destroy(upload_event);
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
