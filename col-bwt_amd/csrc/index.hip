// index.hip -- host side of the loader: .col_pml image -> HBM layout.
//
// File format (col_bwt.hpp:360-370 + LF_table.hpp:325-342):
//   u64 bwt_r, u64 n, u64 r, u64 size, then size raw 18-byte col_thr rows.
// The reference reads this blindly (UB on a bad file);
// this loader validates before anything is dereferenced on the device.
#include "index.h"

#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/colbwt.h"
#include "dev_mem.h"
#include "jump_tables.h"
#include "query_kernels.h"

namespace colbwt {

// A failed HIP call ends the load with the matching C-ABI code.  Every allocation is owned by a
// DevPtr (member tables, local temporaries), so a failed load leaves nothing behind.
#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                   \
            (void)hipGetLastError();                                                   \
            release();                                                                 \
            return e_ == hipErrorOutOfMemory ? COLBWT_ERR_NOMEM : COLBWT_ERR_HIP;      \
        }                                                                              \
    } while (0)

int select_device(int device, std::string &err) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        err = "no HIP device available (the query path has no CPU fallback)";
        return COLBWT_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) {
        err = "device ordinal " + std::to_string(device) + " out of range (" + std::to_string(count) + " devices)";
        return COLBWT_ERR_ARG;
    }
    e = hipSetDevice(device);
    if (e != hipSuccess) {
        err = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return COLBWT_ERR_NO_DEVICE;
    }
    return COLBWT_OK;
}

Index::~Index() { release(); }

void Index::release_one_step() {
    for (DevPtr *p : {&d_rows_, &d_idx_, &d_thr_, &d_next_, &d_prev_}) p->reset();
    tbl_.rows = nullptr;
    tbl_.idx = tbl_.thr = nullptr;
    tbl_.next_tbl = tbl_.prev_tbl = nullptr;
}

void Index::release() {
    if (device_ >= 0) (void)hipSetDevice(device_);
    release_one_step();
    d_cmap_.reset();
    bufk_.release();
    buff_.release();
}

uint64_t Index::device_bytes() const {
    return d_rows_.bytes() + d_idx_.bytes() + d_thr_.bytes() + d_next_.bytes() + d_prev_.bytes() + d_cmap_.bytes() +
           bufk_.bytes() + buff_.bytes();
}

namespace {
// the budget of the open running on this thread (dev_mem.h), dropped when the load returns
struct BudgetScope {
    DevBudget b;
    VmmScope keep_granules;   // freed arrays' memory stays with the process until the open returns (dev_vmm.h)
    BudgetScope() { b.limit = env_budget_bytes(); current_budget() = &b; }
    ~BudgetScope() { current_budget() = nullptr; }
};
}  // namespace

static inline uint64_t rd_u64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}

int Index::load(const uint8_t *bytes, uint64_t len, int device, int layout, std::string &err, int steps) {
    if (!bytes || len < kHeaderBytes) {
        err = "index image shorter than its 32-byte header";
        return COLBWT_ERR_FORMAT;
    }
    const uint64_t bwt_r = rd_u64(bytes + 0), n = rd_u64(bytes + 8), r = rd_u64(bytes + 16), size = rd_u64(bytes + 24);
    if (size != r) {
        err = "header: vector size " + std::to_string(size) + " != r " + std::to_string(r);
        return COLBWT_ERR_FORMAT;
    }
    if (r == 0 || r > 0xFFFFFFFFull - 1) {
        err = "header: r = " + std::to_string(r) + " outside [1, 2^32-2] (RUN_BYTES = 4, common.hpp:53)";
        return COLBWT_ERR_FORMAT;
    }
    if (n < r || n >= (1ull << 40)) {
        err = "header: n = " + std::to_string(n) + " outside [r, 2^40) (BWT_BYTES = 5, common.hpp:52)";
        return COLBWT_ERR_FORMAT;
    }
    if (len != kHeaderBytes + r * (uint64_t)kRowBytesDisk) {
        err = "file length " + std::to_string(len) + " != 32 + 18*" + std::to_string(r) +
              " (row widths other than the shipped 5/4/2/8 are not supported)";
        return COLBWT_ERR_FORMAT;
    }
    int rc = select_device(device, err);
    if (rc != COLBWT_OK) return rc;
    release();
    (void)hipGetLastError();   // a stale error of an earlier attempt must not fail this one
    BudgetScope budget;
    LoadClock clock;
    device_ = device;
    bwt_r_ = bwt_r;

    const uint8_t *rows_disk = bytes + kHeaderBytes;
    const uint32_t nblk = (uint32_t)((r + (1u << kBlockShift) - 1) >> kBlockShift);

    // r rows + the sentinel, padded to whole 128-byte lines (8 rows) so line-wide loads stay in bounds
    const uint64_t rows_alloc = ((r + 1 + 7) & ~7ull) * sizeof(uint4);
    {
        PlainAllocScope whole;             // the one-step layout's query table (dev_mem.h)
        HIP_TRY(d_rows_.alloc(rows_alloc));
    }
    HIP_TRY(hipMemset(d_rows_.get(), 0, rows_alloc));
    HIP_TRY(d_idx_.alloc((r + 1) * sizeof(uint64_t)));
    HIP_TRY(d_thr_.alloc(r * sizeof(uint64_t)));
    HIP_TRY(d_cmap_.alloc(256));
    clock.lap("one-step tables allocated, zeroed");

    // ---- upload packed rows chunk by chunk and re-lay them out on the device
    RelayoutReport h_report{};
    h_report.first_bad = kNone;
    {
        DevPtr report_buf, raw_buf;
        HIP_TRY(report_buf.alloc(sizeof(RelayoutReport)));
        RelayoutReport *d_report = report_buf.as<RelayoutReport>();
        HIP_TRY(hipMemcpy(d_report, &h_report, sizeof(h_report), hipMemcpyHostToDevice));
        const uint64_t chunk_rows = 16ull << 20;  // 288 MiB of packed rows per staging pass
        const uint64_t stage_rows = std::min<uint64_t>(chunk_rows, r);
        HIP_TRY(raw_buf.alloc((stage_rows + 3) * kRowBytesDisk + 64));
        uint8_t *d_raw = raw_buf.as<uint8_t>();
        for (uint64_t row0 = 0; row0 < r; row0 += chunk_rows) {
            const uint64_t count = std::min<uint64_t>(chunk_rows, r - row0);
            const uint64_t with_next = std::min<uint64_t>(count + 3, r - row0);  // following rows' idx for run lengths
            HIP_TRY(hipMemcpy(d_raw, rows_disk + row0 * kRowBytesDisk, with_next * kRowBytesDisk, hipMemcpyHostToDevice));
            launch_relayout(d_raw, row0, count, r, n, d_rows_.as<uint4>(), d_idx_.as<uint64_t>(), d_thr_.as<uint64_t>(),
                            d_report, 0);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(0));
        }
        HIP_TRY(hipMemcpy(&h_report, d_report, sizeof(h_report), hipMemcpyDeviceToHost));
    }
    clock.lap("rows uploaded and re-laid out");
    for (int q = 0; q < 8; ++q) cid_set_[q] = h_report.cids[q];
    if (h_report.flags) {
        err = "corrupt .col_pml near row " + std::to_string(h_report.first_bad) + ":";
        if (h_report.flags & 1u) err += " idx not strictly increasing;";
        if (h_report.flags & 2u) err += " interval >= r;";
        if (h_report.flags & 4u) err += " last idx >= n;";
        if (h_report.flags & 8u) err += " idx[0] != 0;";
        release();
        return COLBWT_ERR_FORMAT;
    }

    // ---- character map: byte -> dense index over the characters present
    // (ordered by decreasing row count so the hint slots cover the frequent characters)
    uint8_t cmap[256];
    uint32_t sigma = 0;
    {
        std::vector<uint32_t> chars;
        for (uint32_t c = 0; c < 256; ++c) {
            cmap[c] = (uint8_t)kAbsent;
            if ((h_report.present[c >> 5] >> (c & 31)) & 1u) chars.push_back(c);
        }
        std::stable_sort(chars.begin(), chars.end(),
                         [&](uint32_t a, uint32_t b) { return h_report.count[a] > h_report.count[b]; });
        sigma = (uint32_t)chars.size();
        for (uint32_t k = 0; k < sigma && k < 255; ++k) cmap[chars[k]] = (uint8_t)k;
    }
    if (sigma > 255) {  // 256 distinct bytes: index 255 would collide with kAbsent
        err = "all 256 byte values occur in the BWT; not supported";
        release();
        return COLBWT_ERR_FORMAT;
    }
    HIP_TRY(hipMemcpy(d_cmap_.get(), cmap, 256, hipMemcpyHostToDevice));

    // ---- jump tables bounding succ_char / pred_char (LF_table.hpp:271-298)
    const uint64_t tbl_entries = (uint64_t)nblk * sigma;
    HIP_TRY(d_next_.alloc(std::max<uint64_t>(tbl_entries, 1) * sizeof(uint32_t)));
    HIP_TRY(d_prev_.alloc(std::max<uint64_t>(tbl_entries, 1) * sizeof(uint32_t)));
    {
        launch_block_first_last(d_rows_.as<uint4>(), (uint32_t)r, nblk, sigma, d_cmap_.as<uint8_t>(), d_next_.as<uint32_t>(),
                                d_prev_.as<uint32_t>(), 0);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(0));
        std::vector<uint32_t> first(tbl_entries), last(tbl_entries);
        HIP_TRY(hipMemcpy(first.data(), d_next_.get(), tbl_entries * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(last.data(), d_prev_.get(), tbl_entries * sizeof(uint32_t), hipMemcpyDeviceToHost));
        // next[b][c] = first run >= b*B holding c ; prev[b][c] = last run < b*B holding c
        std::vector<uint32_t> next, prev;
        finish_jump_tables(first, last, nblk, sigma, next, prev);
        HIP_TRY(hipMemcpy(d_next_.get(), next.data(), tbl_entries * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_prev_.get(), prev.data(), tbl_entries * sizeof(uint32_t), hipMemcpyHostToDevice));
    }

    tbl_.rows = d_rows_.as<uint4>();
    tbl_.idx = d_idx_.as<uint64_t>();
    tbl_.thr = d_thr_.as<uint64_t>();
    tbl_.next_tbl = d_next_.as<uint32_t>();
    tbl_.prev_tbl = d_prev_.as<uint32_t>();
    tbl_.cmap = d_cmap_.as<uint8_t>();
    tbl_.n = n;
    tbl_.r = (uint32_t)r;
    tbl_.sigma = sigma;
    tbl_.nblk = nblk;
    tbl_.use_hints = 0;

    // ---- per-row threshold hints for the (up to) 5 most frequent characters
    {
        HintChars hc{};
        for (uint32_t c = 0; c < 256; ++c)
            if (cmap[c] != kAbsent && cmap[c] < 8) hc.c[cmap[c]] = (uint8_t)c;
        clock.lap("jump tables");
        launch_hints(tbl_, d_rows_.as<uint4>(), hc, 0);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(0));
        tbl_.use_hints = 1;
        clock.lap("threshold hints");

        // ---- optional K-step layout on top (sk_layout.h).  The one-step tables are only the
        // source of the refinement: they are freed as soon as the last pass that reads them is
        // done (before level 3 is allocated), and the K-step index keeps just cmap.
        layout_ = 1;
        if (layout == 2 || layout == 3) {
            rc = build_sk(tbl_, hc, layout, tblk_, bufk_, err, [this] { release_one_step(); });
            if (rc != COLBWT_OK) {
                release();
                return rc;
            }
            layout_ = layout;
        } else if (layout >= 4 && layout <= kLayoutMismatchLinesAuto) {
            rc = build_fat(tbl_, hc, steps, layout - 4, tblf_, buff_, err, [this] { release_one_step(); }, &fat_failed_level_);
            if (rc != COLBWT_OK) {
                release();
                return rc;
            }
            layout_ = tblf_.slot_line0 == 0 ? 4 : (tblf_.entry_shift == 7 ? 6 : 5);   // what the table became
        }
    }
    clock.lap("K-step / line-row layout");
    peak_device_bytes_ = budget.b.peak;
    return COLBWT_OK;
}

}  // namespace colbwt
