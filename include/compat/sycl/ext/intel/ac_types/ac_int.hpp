// ac_int<W, Signed> -- minimal arbitrary-width integer used by the FDTD example as a ring index
// (examples/fdtd/src/defines.hpp:46).  Value semantics only: stored in the smallest standard
// integer that holds W bits and wrapped to W bits after every assignment.
#pragma once
#include "../../../detail_hd.hpp"
#include <cstdint>
#include <type_traits>

template <int W, bool Signed = true> class ac_int {
    static_assert(W >= 1 && W <= 64);
    using U = std::conditional_t<(W <= 8), std::uint8_t,
                                 std::conditional_t<(W <= 16), std::uint16_t,
                                                    std::conditional_t<(W <= 32), std::uint32_t,
                                                                       std::uint64_t>>>;
    using S = std::make_signed_t<U>;

  public:
    using storage_t = std::conditional_t<Signed, S, U>;

    STST_HD constexpr ac_int() : bits(0) {}
    template <typename I>
        requires std::is_arithmetic_v<I>
    STST_HD constexpr ac_int(I v) : bits(wrap(static_cast<U>(v))) {}

    STST_HD constexpr operator storage_t() const { return bits; }

    STST_HD constexpr ac_int &operator++() {
        bits = wrap(static_cast<U>(static_cast<U>(bits) + 1));
        return *this;
    }
    STST_HD constexpr ac_int operator++(int) {
        ac_int old = *this;
        ++*this;
        return old;
    }
    STST_HD constexpr ac_int &operator+=(storage_t o) {
        bits = wrap(static_cast<U>(static_cast<U>(bits) + static_cast<U>(o)));
        return *this;
    }

  private:
    STST_HD static constexpr storage_t wrap(U v) {
        if constexpr (W == sizeof(U) * 8) {
            return static_cast<storage_t>(v);
        } else {
            U m = static_cast<U>((U(1) << W) - 1);
            v &= m;
            if constexpr (Signed) {
                U sign = U(1) << (W - 1);
                return static_cast<storage_t>((v ^ sign) - sign);
            } else {
                return static_cast<storage_t>(v);
            }
        }
    }
    storage_t bits;
};
