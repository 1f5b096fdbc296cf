// me_fast.hpp -- specialised motion-search kernels (filled in after the
// generic kernel is parity-green).
#pragma once
#include "vcs_common.hpp"

namespace vcs {

inline bool me_fast_supported(const Geom &) { return false; }

inline int launch_me_fast(const Geom &, hipStream_t, const uint8_t *, const uint8_t *, size_t, int,
                          int16_t *, uint32_t *, unsigned long long *) {
    return -1;
}

}  // namespace vcs
