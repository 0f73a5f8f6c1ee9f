// hip::Grid uploads without making the host wait (the update that follows allocates its buffers meanwhile): what the
// host does to the source grid AFTER a non-blocking update must not reach that update -- a write accessor, a
// copy_from_buffer and the grid's destruction wait for the upload first.  (hipcc, needs a GPU to run.)
#include "mini_test.hpp"
#include <StencilStream/BaseTransitionFunction.hpp>
#include <StencilStream/cuda/StencilUpdate.hpp>
#include <memory>
#include <vector>

using namespace stencil;

struct Shift : public BaseTransitionFunction {
    using Cell = float;
    float operator()(Stencil<float, 1> const &s) const { return s[0][-1] + 1.0f; }
};
using SU = cuda::StencilUpdate<Shift>;

static float input_at(std::size_t r, std::size_t c) { return float((r * 131 + c * 7) % 1009); }

// after n generations cell (r, c) holds input(r, c - n) + n, or halo + (c + 1) where the halo has reached it
static bool as_expected(SU::GridImpl &out, std::size_t h, std::size_t w, std::size_t n, float halo) {
    SU::GridImpl::GridAccessor<sycl::access::mode::read> ac(out);
    bool same = true;
    for (std::size_t r = 0; r < h; r++)
        for (std::size_t c = 0; c < w; c++)
            same = same && ac[r][c] == (c >= n ? input_at(r, c - n) + float(n) : halo + float(c + 1));
    return same;
}

int main() {
    // large enough for the upload to take a while: 4096 x 8192 floats = 128 MiB
    const std::size_t h = 4096, w = 8192, n = 5;
    const float halo = -3.0f;
    SU update({.transition_function = Shift{}, .halo_value = halo, .n_iterations = n, .blocking = false});
    {
        SU::GridImpl grid(h, w);
        {
            SU::GridImpl::GridAccessor<sycl::access::mode::read_write> ac(grid);
            for (std::size_t r = 0; r < h; r++)
                for (std::size_t c = 0; c < w; c++)
                    ac[r][c] = input_at(r, c);
        }
        SU::GridImpl out = update(grid);
        {
            // the host scribbles over the source right away
            SU::GridImpl::GridAccessor<sycl::access::mode::read_write> ac(grid);
            for (std::size_t r = 0; r < h; r++)
                for (std::size_t c = 0; c < w; c++)
                    ac[r][c] = -1.0f;
        }
        REQUIRE(as_expected(out, h, w, n, halo));
        // ... and the scribbled values are what the next update sees
        SU::GridImpl second = update(grid);
        SU::GridImpl::GridAccessor<sycl::access::mode::read> ac(second);
        REQUIRE(ac[7][w - 1] == -1.0f + float(n) && ac[7][0] == halo + 1.0f);
    }
    {
        // copy_from_buffer after an update, and a source grid that dies while its upload may still be running
        sycl::buffer<float, 2> a(sycl::range<2>(h, w)), b(sycl::range<2>(h, w));
        for (std::size_t r = 0; r < h; r++)
            for (std::size_t c = 0; c < w; c++) {
                a.data()[r * w + c] = input_at(r, c);
                b.data()[r * w + c] = 0.0f;
            }
        auto grid = std::make_unique<SU::GridImpl>(a);
        SU::GridImpl out = update(*grid);
        grid->copy_from_buffer(b);
        SU::GridImpl out2 = update(*grid);
        grid.reset();
        REQUIRE(as_expected(out, h, w, n, halo));
        SU::GridImpl::GridAccessor<sycl::access::mode::read> ac(out2);
        REQUIRE(ac[100][w - 1] == float(n));
    }
    return finish("grid_upload_test");
}
