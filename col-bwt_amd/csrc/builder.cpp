// builder.cpp -- `.col_pml` construction from a split RLBWT (SURVEY.md 8(f)
// "next" #1): the in-repo builder `build_col_bwt <prefix>`
// (src/build_col_bwt.cpp:38-52), i.e.
//   col_pml(heads, lengths, col_ids, thresholds, splits)
//     = col_bwt ctor           col_bwt.hpp:124-230   (row emission at run heads and split bits)
//     + compute_table          LF_table.hpp:365-387  ((interval, offset) from the F order)
//     + read_thresholds        col_bwt.hpp:440-457   (a run's threshold copied to its sub-runs)
//     + serialize              col_bwt.hpp:360-370, LF_table.hpp:325-342
// Host code (index construction is build-time, I/O bound); linear time and
// streaming-friendly: no vector-of-vectors per character as in the reference.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/colbwt.h"
#include "disk_format.h"

namespace {

inline void put_le(uint8_t *p, uint64_t v, unsigned nbytes) {
    for (unsigned b = 0; b < nbytes; ++b) p[b] = (uint8_t)(v >> (8 * b));
}

bool read_file(const std::string &path, std::vector<uint8_t> &out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long long len = ftello(f);
    fseek(f, 0, SEEK_SET);
    out.resize((size_t)len);
    const size_t got = len ? fread(out.data(), 1, (size_t)len, f) : 0;
    fclose(f);
    return (long long)got == len;
}

void widen5(const std::vector<uint8_t> &raw, std::vector<uint64_t> &out) {  // RW_BYTES = 5 (common.hpp:46)
    out.resize(raw.size() / 5);
    for (size_t k = 0; k < out.size(); ++k) {
        uint64_t v = 0;
        for (int b = 0; b < 5; ++b) v |= (uint64_t)raw[5 * k + b] << (8 * b);
        out[k] = v;
    }
}

struct SubRun {
    uint64_t idx;
    uint8_t ch;
    uint8_t id;
};

}  // namespace

extern "C" int colbwt_build_col_pml_arrays(const uint8_t *heads, uint64_t n_heads, const uint64_t *lens,
                                           const uint8_t *col_ids, uint64_t n_ids, const uint64_t *split_pos,
                                           uint64_t n_splits, const uint64_t *thr_pos, uint64_t n_thr, void *out,
                                           uint64_t out_cap, uint64_t *out_len) {
    if (!heads || !lens || !out_len || (n_ids && !col_ids) || (n_splits && !split_pos) || (n_thr && !thr_pos))
        return COLBWT_ERR_ARG;
    using namespace colbwt;

    // ---- 1. sub-runs: a row starts at every BWT run head and at every split bit
    // strictly inside a run; its id is the id of the last split bit at or before
    // its start (ids are consumed one per split bit, col_bwt.hpp:177-181,197-199;
    // 0 before the first bit; the last value persists when .col_ids runs out).
    std::vector<SubRun> rows;
    rows.reserve(n_heads + n_splits);
    uint64_t n = 0, bwt_r = 0, s = 0, ids_used = 0;
    uint8_t cur_id = 0;
    auto consume_split = [&]() {
        if (ids_used < n_ids) cur_id = col_ids[ids_used++];
        ++s;
    };
    for (uint64_t h = 0; h < n_heads; ++h) {
        const uint8_t b = heads[h];
        if (b == 0xFF) break;  // `char c; (c = heads.get()) != EOF` ends at byte 0xFF (col_bwt.hpp:167)
        // `if (c <= TERMINATOR) c = TERMINATOR` on a signed char: 0x00, 0x01 and every byte >= 0x80 (:171)
        const uint8_t ch = (b <= 1 || b >= 0x80) ? 1 : b;
        uint64_t len = lens[h];
        const uint64_t run_end = n + len;
        if (s < n_splits && split_pos[s] == n) consume_split();   // :177-181
        while (s < n_splits && split_pos[s] < run_end) {
            rows.push_back({n, ch, cur_id});
            len -= split_pos[s] - n;
            n = split_pos[s];
            consume_split();
        }
        if (len > 0) {
            rows.push_back({n, ch, cur_id});
            n += len;
        }
        ++bwt_r;
    }
    const uint64_t r = rows.size();
    const uint64_t need = kHeaderBytes + r * (uint64_t)kRowBytesDisk;
    *out_len = need;
    if (!out || out_cap < need) return COLBWT_ERR_ARG;
    uint8_t *img = (uint8_t *)out;
    uint8_t *row = img + kHeaderBytes;
    auto idx_of = [&](uint64_t i) { return i < r ? rows[i].idx : n; };

    // ---- 2. (interval, offset): rows holding character c, in row order, tile F
    // contiguously from C[c]; a row's F start lies in row `interval` at `offset`
    // (compute_table, LF_table.hpp:365-387).  One pass with a destination cursor
    // per character; both fields keep only their bit-field widths (:375-376).
    uint64_t fpos[256], dst[256];
    {
        uint64_t total[256];
        memset(total, 0, sizeof(total));
        for (uint64_t i = 0; i < r; ++i) total[rows[i].ch] += idx_of(i + 1) - rows[i].idx;
        uint64_t acc = 0;
        for (int c = 0; c < 256; ++c) {
            fpos[c] = acc;
            acc += total[c];
            uint64_t lo = 0, hi = r;  // row containing F position fpos[c]
            while (hi - lo > 1) {
                const uint64_t mid = (lo + hi) >> 1;
                if (idx_of(mid) <= fpos[c]) lo = mid; else hi = mid;
            }
            dst[c] = lo;
        }
    }
    // ---- 3. thresholds: the k-th value goes to the k-th maximal group of consecutive
    // rows with equal character (read_thresholds' do/while, col_bwt.hpp:448-451);
    // rows beyond the last value keep 0.
    uint64_t group = 0;
    for (uint64_t i = 0; i < r; ++i) {
        const uint8_t c = rows[i].ch;
        if (i > 0 && rows[i - 1].ch != c) ++group;
        uint64_t &f = fpos[c];
        while (dst[c] + 1 < r && f >= idx_of(dst[c] + 1)) ++dst[c];
        uint8_t *p = row + i * kRowBytesDisk;
        p[0] = c;
        put_le(p + 1, rows[i].idx, 5);
        put_le(p + 6, dst[c], 4);
        put_le(p + 10, f - idx_of(dst[c]), 2);
        p[12] = rows[i].id;
        put_le(p + 13, group < n_thr ? thr_pos[group] : 0, 5);
        f += idx_of(i + 1) - rows[i].idx;
    }
    put_le(img + 0, bwt_r, 8);
    put_le(img + 8, n, 8);
    put_le(img + 16, r, 8);
    put_le(img + 24, r, 8);
    return COLBWT_OK;
}

// build_col_bwt <prefix> (src/build_col_bwt.cpp:14-52): reads
//   <prefix>.bwt.heads, <prefix>.bwt.len, <prefix>.col_ids, <prefix>.col_runs, <prefix>.thr_pos
// and writes <prefix>.col_pml (or out_path).  `.col_runs` is read as the plain
// sdsl::bit_vector that col_split writes under that name (col_split.hpp:384-386:
// u64 length in bits, then ceil(len/64) u64 words) -- the reference's own builder
// loads an sd_vector from the same name (build_col_bwt.cpp:24-25), a format
// mismatch in the reference itself; sdsl's serialisation is not
// in the container, so this container format is "parity unpinned".
extern "C" int colbwt_build_col_pml(const char *prefix, const char *out_path) {
    if (!prefix) return COLBWT_ERR_ARG;
    const std::string p = prefix;
    std::vector<uint8_t> heads, len_raw, ids, runs_raw, thr_raw;
    if (!read_file(p + ".bwt.heads", heads) || !read_file(p + ".bwt.len", len_raw) ||
        !read_file(p + ".col_ids", ids) || !read_file(p + ".col_runs", runs_raw) || !read_file(p + ".thr_pos", thr_raw))
        return COLBWT_ERR_IO;
    std::vector<uint64_t> lens, thr, splits;
    widen5(len_raw, lens);
    widen5(thr_raw, thr);
    if (lens.size() < heads.size()) lens.resize(heads.size(), 0);  // a short .bwt.len reads as length 0 (:169-170)
    if (runs_raw.size() < 8) return COLBWT_ERR_FORMAT;
    uint64_t nbits = 0;
    memcpy(&nbits, runs_raw.data(), 8);
    if (runs_raw.size() < 8 + ((nbits + 63) / 64) * 8) return COLBWT_ERR_FORMAT;
    for (uint64_t w = 0; w < (nbits + 63) / 64; ++w) {
        uint64_t word;
        memcpy(&word, runs_raw.data() + 8 + 8 * w, 8);
        while (word) {
            const uint64_t bit = (uint64_t)__builtin_ctzll(word);
            if (w * 64 + bit < nbits) splits.push_back(w * 64 + bit);
            word &= word - 1;
        }
    }
    uint64_t need = 0;
    colbwt_build_col_pml_arrays(heads.data(), heads.size(), lens.data(), ids.data(), ids.size(), splits.data(),
                                splits.size(), thr.data(), thr.size(), nullptr, 0, &need);
    std::vector<uint8_t> img(need);
    int rc = colbwt_build_col_pml_arrays(heads.data(), heads.size(), lens.data(), ids.data(), ids.size(), splits.data(),
                                         splits.size(), thr.data(), thr.size(), img.data(), need, &need);
    if (rc != COLBWT_OK) return rc;
    const std::string outp = out_path ? std::string(out_path) : p + ".col_pml";  // col_bwt.hpp:434-437
    FILE *f = fopen(outp.c_str(), "wb");
    if (!f) return COLBWT_ERR_IO;
    const bool ok = fwrite(img.data(), 1, img.size(), f) == img.size();
    return (fclose(f) == 0 && ok) ? COLBWT_OK : COLBWT_ERR_IO;
}
