// jump_tables.h -- host half of the succ_char / pred_char jump tables: the
// device kernels give, per jump block, the first / last row holding each
// character; this turns them into "first row >= block start" (suffix minimum)
// and "last row < block start" (exclusive prefix) per character.
#pragma once
#include <stdint.h>

#include <vector>

#include "device_layout.h"

namespace colbwt {

inline void finish_jump_tables(const std::vector<uint32_t> &first, const std::vector<uint32_t> &last, uint64_t nblk,
                               uint32_t sigma, std::vector<uint32_t> &next, std::vector<uint32_t> &prev) {
    next.resize(first.size());
    prev.resize(first.size());
    for (uint32_t c = 0; c < sigma; ++c) {
        uint32_t carry = kNone;
        for (uint64_t b = nblk; b-- > 0;) {
            if (first[b * sigma + c] != kNone) carry = first[b * sigma + c];
            next[b * sigma + c] = carry;
        }
        carry = kNone;
        for (uint64_t b = 0; b < nblk; ++b) {
            prev[b * sigma + c] = carry;
            if (last[b * sigma + c] != kNone) carry = last[b * sigma + c];
        }
    }
}

}  // namespace colbwt
