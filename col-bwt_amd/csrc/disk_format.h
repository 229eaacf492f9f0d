// disk_format.h -- constants of the reference's on-disk .col_pml format
// (common.hpp:46-54; col_bwt.hpp:81-115).
#pragma once
#include <stdint.h>

namespace colbwt {
constexpr uint32_t kRowBytesDisk = 18;  // sizeof(col_thr), packed
constexpr uint32_t kHeaderBytes = 32;   // bwt_r, n, r, size (4 x u64)
}  // namespace colbwt
