// index.h -- the run table resident in HBM (host-side owner object).
// Mirrors col_pml's load half (col_bwt.hpp:375-380 -> LF_table.hpp:347-357).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "dev_mem.h"
#include "device_layout.h"
#include "query_kernels.h"
#include "sk_layout.h"

namespace colbwt {

// Index::load only (not a public layout): mismatch lines, deep when the table then still leaves room
// for batches (fat_build.hip kDeepReserve), plain otherwise -- what COLBWT_LAYOUT_AUTO asks for first.
constexpr int kLayoutMismatchLinesAuto = 7;

struct IndexError {
    int code;
    std::string msg;
};

class Index {
public:
    Index() = default;
    ~Index();
    Index(const Index &) = delete;
    Index &operator=(const Index &) = delete;

    // `bytes` is the whole .col_pml image (header + rows) in host memory.
    // Returns 0 or a COLBWT_ERR_* code with `err` filled.
    // layout: 1 = one-step (device_layout.h); 2 / 3 = K-step (sk_layout.h, refined from 1);
    // 4 = line rows (fat_layout.h) with `steps` own steps; 5 = the same with mismatch lines; 6 = with deep ones;
    // kLayoutMismatchLinesAuto = 6 when there is room, else 5 (layout() says which).
    int load(const uint8_t *bytes, uint64_t len, int device, int layout, std::string &err, int steps = 0);

    const DevTable &table() const { return tbl_; }
    const SKTable &table_k() const { return tblk_; }
    const FatTable &table_fat() const { return tblf_; }
    int layout() const { return layout_; }
    bool line_rows() const { return layout_ >= 4 && layout_ <= 6; }
    uint64_t table_rows() const { return line_rows() ? tblf_.r : (layout_ >= 2 ? tblk_.r : tbl_.r); }
    int device() const { return device_; }
    uint64_t bwt_r() const { return bwt_r_; }
    uint64_t n() const { return tbl_.n; }
    uint64_t r() const { return tbl_.r; }
    uint32_t sigma() const { return tbl_.sigma; }
    const uint32_t *cid_set() const { return cid_set_; }            // 256-bit set of the col ids the table's rows hold
    uint64_t device_bytes() const;                                  // HBM held now
    uint64_t peak_device_bytes() const { return peak_device_bytes_; }  // ... and at most while loading
    // after a failed line-row load: the refinement level that did not fit (query_kernels.h build_fat), else 0
    int fat_failed_level() const { return fat_failed_level_; }

private:
    void release();
    void release_one_step();
    DevTable tbl_{};
    SKTable tblk_{};
    SKBuffers bufk_;
    FatTable tblf_{};
    FatBuffers buff_;
    int layout_ = 1;
    int fat_failed_level_ = 0;
    uint32_t cid_set_[8] = {};
    uint64_t bwt_r_ = 0;
    int device_ = -1;
    uint64_t peak_device_bytes_ = 0;
    DevPtr d_rows_, d_idx_, d_thr_, d_next_, d_prev_, d_cmap_;
};

// Selects `device` after checking that a usable gfx950-class HIP device exists.
int select_device(int device, std::string &err);

}  // namespace colbwt
