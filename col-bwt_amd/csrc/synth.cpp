// synth.cpp -- direct move-table synthesis of a .col_pml image (host, linear
// time), the benchmark/test input generator described in SURVEY.md 8(d):
// the reference ships no data and its index builders (mumemto, Movi) are not
// available, so large indices are synthesised directly in the on-disk format
//  that col_pml::load reads (col_bwt.hpp:375-380).
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/colbwt.h"
#include "disk_format.h"

namespace {

inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

inline void put_le(uint8_t *p, uint64_t v, unsigned nbytes) {
    for (unsigned b = 0; b < nbytes; ++b) p[b] = (uint8_t)(v >> (8 * b));
}

}  // namespace

extern "C" uint64_t colbwt_synth_index_bytes(uint64_t rows) {
    return colbwt::kHeaderBytes + rows * (uint64_t)colbwt::kRowBytesDisk;
}

extern "C" int colbwt_synth_index(uint64_t rows, uint32_t mean_len, uint32_t split_permille, uint64_t seed, void *out,
                                  uint64_t out_len) {
    return colbwt_synth_index_thr(rows, mean_len, split_permille, seed, COLBWT_SYNTH_THR_UNIFORM, out, out_len);
}

extern "C" int colbwt_synth_index_thr(uint64_t rows, uint32_t mean_len, uint32_t split_permille, uint64_t seed,
                                      int thr_mode, void *out, uint64_t out_len) {
    using namespace colbwt;
    if (thr_mode != COLBWT_SYNTH_THR_UNIFORM && thr_mode != COLBWT_SYNTH_THR_BETWEEN_RUNS) return COLBWT_ERR_ARG;
    if (!out || rows < 2 || rows > 0xFFFFFFFEull || mean_len < 1 || mean_len > 60000 || split_permille >= 1000 ||
        out_len < colbwt_synth_index_bytes(rows))
        return COLBWT_ERR_ARG;
    uint8_t *img = (uint8_t *)out;
    uint8_t *row = img + kHeaderBytes;
    const uint64_t r = rows;
    static const uint8_t acgt[4] = {'A', 'C', 'G', 'T'};
    static const uint8_t ids[9] = {0, 0, 0, 1, 2, 3, 17, 200, 255};
    const double inv_log_q = mean_len > 1 ? 1.0 / log(1.0 - 1.0 / (double)mean_len) : 0.0;

    // pass 1: characters, lengths -> idx, col ids; count BWT runs and per-char totals
    uint64_t n = 0, bwt_r = 0;
    uint64_t char_total[256];
    memset(char_total, 0, sizeof(char_total));
    uint32_t prev = 0xFFFF;
    uint32_t cur4 = 0;
    for (uint64_t i = 0; i < r; ++i) {
        const uint64_t h = splitmix64(seed ^ (i * 0x9E3779B97F4A7C15ull));
        uint32_t ch;
        uint64_t len;
        if (i == r / 2) {  // the single terminator run (TERMINATOR = 1, common.hpp:72)
            ch = 1;
            len = 1;
        } else {
            const bool split = prev != 0xFFFF && prev != 1 && (uint32_t)(h % 1000) < split_permille;
            if (!split) {
                if (prev == 0xFFFF) cur4 = (uint32_t)((h >> 10) & 3);
                else cur4 = (cur4 + 1 + (uint32_t)((h >> 10) % 3)) & 3;  // adjacent-distinct walk
            }
            ch = acgt[cur4];
            if (mean_len == 1) {
                len = 1;
            } else {
                const double u = ((double)((h >> 11) | 1)) * (1.0 / 9007199254740992.0);  // (0,1)
                len = 1 + (uint64_t)(log(u) * inv_log_q);
                if (len > 60000) len = 60000;
            }
        }
        if (ch != prev) ++bwt_r;
        prev = ch;
        uint8_t *p = row + i * kRowBytesDisk;
        p[0] = (uint8_t)ch;
        put_le(p + 1, n, 5);
        p[12] = ids[(h >> 44) % 9];
        char_total[ch] += len;
        n += len;
    }
    if (n >= (1ull << 40)) return COLBWT_ERR_ARG;

    // pass 2: thresholds; sub-runs of one BWT run share a threshold (col_pml::read_thresholds
    // copies a run's threshold to all its sub-runs, col_bwt.hpp:448-454).
    //   UNIFORM       uniform in [0, n) (the SURVEY.md 8(d) recipe)
    //   BETWEEN_RUNS  where real thresholds live: a position between the end of the previous run
    //                 of the same character and the head of this run; 0 for a character's first run
    {
        uint64_t thr = 0;
        uint64_t last_end[256];
        bool seen[256];
        memset(last_end, 0, sizeof(last_end));
        memset(seen, 0, sizeof(seen));
        auto idx_at = [&](uint64_t i) -> uint64_t {
            if (i >= r) return n;
            const uint8_t *q = row + i * kRowBytesDisk + 1;
            uint64_t v = 0;
            for (int b = 0; b < 5; ++b) v |= (uint64_t)q[b] << (8 * b);
            return v;
        };
        for (uint64_t i = 0; i < r; ++i) {
            uint8_t *p = row + i * kRowBytesDisk;
            const uint8_t c = p[0];
            if (i == 0 || c != (p - kRowBytesDisk)[0]) {      // head of a BWT run
                const uint64_t h = splitmix64(seed ^ 0xABCDull ^ (i * 0xD6E8FEB86659FD93ull));
                if (thr_mode == COLBWT_SYNTH_THR_UNIFORM) {
                    thr = h % n;
                } else {
                    const uint64_t head = idx_at(i);
                    thr = seen[c] ? last_end[c] + h % (head - last_end[c] + 1) : 0;
                }
            }
            put_le(p + 13, thr, 5);
            seen[c] = true;
            last_end[c] = idx_at(i + 1);                      // one past the run's last position
        }
    }

    // pass 3: (interval, offset) from the stable char-sorted F order
    // (LF_table::compute_table, LF_table.hpp:365-387): rows holding character c, in
    // row order, tile F contiguously starting at C[c].
    uint64_t fstart[256];
    {
        uint64_t acc = 0;
        for (int c = 0; c < 256; ++c) { fstart[c] = acc; acc += char_total[c]; }
    }
    auto idx_of = [&](uint64_t i) -> uint64_t {
        if (i >= r) return n;
        const uint8_t *p = row + i * kRowBytesDisk + 1;
        uint64_t v = 0;
        for (int b = 0; b < 5; ++b) v |= (uint64_t)p[b] << (8 * b);
        return v;
    };
    // one forward pass with a destination cursor per character (each cursor only moves forward)
    uint64_t dst[256], dst_end[256];
    for (int c = 0; c < 256; ++c) {
        dst[c] = dst_end[c] = 0;
        if (!char_total[c]) continue;
        uint64_t lo = 0, hi = r;  // row containing F position fstart[c]
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (idx_of(mid) <= fstart[c]) lo = mid; else hi = mid;
        }
        dst[c] = lo;
        dst_end[c] = idx_of(lo + 1);
    }
    {
        uint64_t idx_i = 0;
        for (uint64_t i = 0; i < r; ++i) {
            uint8_t *p = row + i * kRowBytesDisk;
            const uint32_t c = p[0];
            const uint64_t idx_next = idx_of(i + 1);
            uint64_t &f = fstart[c];
            while (f >= dst_end[c]) { ++dst[c]; dst_end[c] = idx_of(dst[c] + 1); }
            put_le(p + 6, dst[c], 4);
            put_le(p + 10, f - idx_of(dst[c]), 2);
            f += idx_next - idx_i;
            idx_i = idx_next;
        }
    }

    put_le(img + 0, bwt_r, 8);
    put_le(img + 8, n, 8);
    put_le(img + 16, r, 8);
    put_le(img + 24, r, 8);
    return COLBWT_OK;
}
