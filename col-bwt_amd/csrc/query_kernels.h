// query_kernels.h -- host-visible launchers of the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_layout.h"

namespace colbwt {

constexpr uint32_t kQueryBlock = 256;  // 4 waves; one lane per read

// col_pml::query_pml for n_reads reads resident in HBM (col_bwt.hpp:498-529).
// d_order (nullable): read indices by decreasing length (lane assignment for ragged batches).
void launch_pml_query(const DevTable &T, const uint8_t *d_bases, const uint64_t *d_read_off,
                      uint64_t n_reads, void *d_pml, int pml_bytes, uint8_t *d_cid, const uint32_t *d_order,
                      hipStream_t stream);

}  // namespace colbwt
#include <functional>
#include <string>

#include "dev_mem.h"
#include "fat_layout.h"
#include "sk_layout.h"
namespace colbwt {

// The same query over a K-step layout (sk_query.hip); K = T.steps.
void launch_sk_query(const SKTable &T, const uint8_t *d_bases, const uint64_t *d_read_off, uint64_t n_reads, uint64_t n_bases,
                     void *d_pml, int pml_bytes, uint8_t *d_cid, const uint32_t *d_order, hipStream_t stream);
// ... three-step rows with persistent lanes and pair-fetched rows (sk3_query.hip; launch_sk_query dispatches to it)
void launch_sk3_query(const SKTable &T, const uint8_t *d_bases, const uint64_t *d_read_off, uint64_t n_reads, uint64_t n_bases,
                      void *d_pml, int pml_bytes, uint8_t *d_cid, hipStream_t stream);

// ---- load-time kernels (index_kernels.hip) --------------------------------
struct RelayoutReport {
    uint32_t flags;      // bit0 idx not strictly increasing, bit1 interval >= r, bit2 idx >= n, bit3 idx[0] != 0
    uint32_t first_bad;  // smallest offending row
    uint32_t present[8]; // 256-bit set of characters seen
    uint32_t cids[8];    // 256-bit set of col ids seen (the dictionary of the gather codec)
    uint32_t count[256]; // rows per character (orders the dense character indices by frequency)
};

// Packed 18-byte rows [row0, row0+count) (plus up to 3 following rows, whose idx
// give run lengths) -> 16-byte rows + idx + thresholds; validates as it goes.
void launch_relayout(const uint8_t *d_raw, uint64_t row0, uint64_t count, uint64_t r, uint64_t n,
                     uint4 *d_rows, uint64_t *d_idx, uint64_t *d_thr, RelayoutReport *d_report,
                     hipStream_t stream);
void launch_block_first_last(const uint4 *d_rows, uint32_t r, uint32_t nblk, uint32_t sigma,
                             const uint8_t *d_cmap, uint32_t *d_first, uint32_t *d_last, hipStream_t stream);

// Threshold hints (device_layout.h); chars.c[cidx] = the byte with dense index cidx.
struct HintChars {
    uint8_t c[8];
};
void launch_hints(const DevTable &T, uint4 *d_rows_rw, const HintChars &chars, hipStream_t stream);

// gather_codec.hip: one bit per base for the PML values on their way to rank 0
void launch_pml_pack(const uint16_t *d_pml, uint64_t n_bases, uint32_t *d_mask, hipStream_t stream);
void launch_read_end_mask(const uint64_t *d_read_off, uint64_t n_reads, uint32_t *d_mask, hipStream_t stream);
void launch_pml_unpack(const uint32_t *d_flag, const uint32_t *d_last, uint64_t first_word, uint64_t n_words,
                       uint64_t total_words, uint16_t *d_pml, hipStream_t stream);
// ... and `bits` bits per base for the col ids: codes of a dictionary (the col ids the table holds), as bit planes
struct CidLut { uint8_t v[256]; };       // pack: col id -> code; unpack: code -> col id
void launch_cid_pack(const uint8_t *d_cid, uint64_t n_bases, const CidLut &code_of, uint32_t bits, uint32_t *d_planes, hipStream_t stream);
void launch_cid_unpack(const uint32_t *d_planes, uint64_t first_word, uint64_t n_words, const CidLut &id_of, uint32_t bits,
                       uint8_t *d_cid, hipStream_t stream);

// Device allocations of one K-step table.
struct SKBuffers {
    DevPtr lines, idx, thr, next, prev;
    uint64_t bytes() const;
    void release();
};
// K-step layout (steps = 2 or 3) from the one-step tables (sk_build.hip).  COLBWT_OK, or
// COLBWT_ERR_NOMEM when it cannot be built for lack of room (HBM, or more than 2^32-2 refined
// rows -- a shallower layout may fit), or COLBWT_ERR_HIP; `err` says what failed and nothing
// stays allocated.  `source_done` is called once the one-step tables are no longer read.
int build_sk(const DevTable &T, const HintChars &chars, int steps, SKTable &out, SKBuffers &buf, std::string &err,
             const std::function<void()> &source_done);

// Device allocations of a line-row table (fat_layout.h).
struct FatBuffers {
    DevPtr lines, chr, idx, thr, next, prev;
    uint64_t bytes() const;
    void release();
};
// Line rows with `steps` own steps (fat_build.hip), with in-row mismatch slots (mismatch_lines = 0),
// with mismatch lines (1), with deep ones (2: fat_layout.h), or with deep ones when the table then
// still leaves room for batches and plain ones otherwise (3: what AUTO asks for; out.entry_shift
// says which it became); same contract as build_sk.
// fat_steps_supported: the step counts compiled in.  On failure *failed_level (nullable) says how
// far the build got: the refinement level (2 .. steps) that could not be built -- a build with
// at least that many steps fails the same way -- or steps + 1 when the levels fit and the final
// tables did not.
bool fat_steps_supported(int steps);
int build_fat(const DevTable &T, const HintChars &chars, int steps, int mismatch_lines, FatTable &out, FatBuffers &buf,
              std::string &err, const std::function<void()> &source_done, int *failed_level);
// The query over line rows (fat_query.hip).
void launch_fat_query(const FatTable &T, const uint8_t *d_bases, const uint64_t *d_read_off, uint64_t n_reads, uint64_t n_bases,
                      void *d_pml, int pml_bytes, uint8_t *d_cid, const uint32_t *d_order, hipStream_t stream);
// ... over line rows with mismatch lines (fat2_query.hip; launch_fat_query dispatches to it)
void launch_fat2_query(const FatTable &T, const uint8_t *d_bases, const uint64_t *d_read_off, uint64_t n_reads, uint64_t n_bases,
                       void *d_pml, int pml_bytes, uint8_t *d_cid, hipStream_t stream);
void launch_fat_synth_reads(const FatTable &T, uint64_t n_reads, uint32_t read_len, uint32_t sub_permille,
                            uint64_t seed, uint8_t *d_bases, uint64_t *d_read_off, hipStream_t stream);

// Backward-walk read sampler (synthetic benchmark input); the same reads come
// out of every layout (a read is a function of its start position in the BWT).
void launch_synth_reads(const DevTable &T, uint64_t n_reads, uint32_t read_len, uint32_t sub_permille,
                        uint64_t seed, uint8_t *d_bases, uint64_t *d_read_off, hipStream_t stream);
void launch_sk_synth_reads(const SKTable &T, uint64_t n_reads, uint32_t read_len, uint32_t sub_permille,
                           uint64_t seed, uint8_t *d_bases, uint64_t *d_read_off, hipStream_t stream);

}  // namespace colbwt
