/*
 * colbwt.h -- C-ABI of the MI355X-native col-bwt PML / chain-statistic query
 * engine (libcolbwt.so, hand-written HIP for gfx950).
 *
 * The reference (drnatebrown/col-bwt) has no FFI layer; the entry points below
 * are the ones a binding of its query path would need.  Each cites the
 * reference interface (file:line under /root/reference) it replaces.
 *
 *   col_pml tbl; tbl.load(ifstream)            col_bwt.hpp:375-380, LF_table.hpp:347-357
 *        -> colbwt_index_open / colbwt_index_open_memory
 *   tbl.query_pml(const char*, size_t)         col_bwt.hpp:409-412 (one read)
 *        -> colbwt_query_batch (many reads per call; read k of the batch is
 *           what one query_pml call returns: pml[k'], cid[k'] <-> pattern[k'])
 *   pml_to_vec / main of pml_query             pml_query.cpp:65-90, 92-143
 *        -> colbwt_query_file (FASTA/FASTQ[.gz] in, text .pml/.cid out)
 *   ~col_pml                                   (implicit)
 *        -> colbwt_index_close
 *
 * Conventions: plain pointers and sizes, caller-owned buffers, int return
 * codes (0 = ok, <0 = error; message via colbwt_last_error(), thread-local),
 * no exceptions cross the ABI.  An index may be queried from several host
 * threads at once on distinct batches (each call uses its own stream).
 * There is NO CPU fallback: every query entry point fails with
 * COLBWT_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef COLBWT_H
#define COLBWT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COLBWT_OK 0
#define COLBWT_ERR_ARG (-1)        /* bad argument (null, misaligned, sizes)  */
#define COLBWT_ERR_IO (-2)         /* file missing / short read               */
#define COLBWT_ERR_FORMAT (-3)     /* .col_pml fails validation               */
#define COLBWT_ERR_NO_DEVICE (-4)  /* no usable HIP device                    */
#define COLBWT_ERR_HIP (-5)        /* a HIP call failed                       */
#define COLBWT_ERR_NOMEM (-6)

typedef struct colbwt_index colbwt_index; /* opaque: table resident in HBM */

/* On-disk widths of the reference build (common.hpp:46-54).  Only the shipped
 * values are accepted; the struct exists so a binding can assert them. */
typedef struct colbwt_widths {
    uint32_t bwt_bytes; /* BWT_BYTES 5 */
    uint32_t run_bytes; /* RUN_BYTES 4 */
    uint32_t len_bytes; /* LEN_BYTES 2 */
    uint32_t id_bits;   /* ID_BITS   8 */
} colbwt_widths;

typedef struct colbwt_info {
    uint64_t bwt_r;        /* maximal BWT runs      (col_bwt.hpp:383)  */
    uint64_t n;            /* BWT length            (LF_table.hpp:360) */
    uint64_t r;            /* rows (sub-runs)       (LF_table.hpp:361) */
    uint32_t sigma;        /* distinct characters present in the table */
    uint32_t device;       /* HIP device ordinal                       */
    uint64_t device_bytes; /* HBM held by the index                    */
    uint32_t layout;       /* COLBWT_LAYOUT_ONE_STEP / _TWO_ / _THREE_ / _LINE_ROWS / _MISMATCH_LINES[_DEEP] */
    uint32_t layout_shape; /* line rows: own steps << 8 | steps per mismatch slot; else 0 */
    uint64_t table_rows;   /* rows of the HBM table actually queried   */
    uint32_t n_devices;    /* replicas of the table (colbwt_index_open_devices); the fields above describe the first */
    uint32_t reserved_;
} colbwt_info;

typedef struct colbwt_stats {
    uint64_t n_reads;
    uint64_t n_bases;
    double h2d_ms;     /* reads host->HBM (0 for the device entry point)    */
    double kernel_ms;  /* HIP-event time of the query kernel(s), on-stream  */
    double d2h_ms;     /* results HBM->host (0 for the device entry point)  */
    uint64_t algorithmic_bytes; /* 27 B/base, SURVEY.md section 8(d)         */
} colbwt_stats;

const char *colbwt_version(void);
const char *colbwt_last_error(void);

/* col_pml::load (col_bwt.hpp:375-380).  `prefix_or_file`: either the index
 * prefix (".col_pml" is appended, pml_query.cpp:110-111; extension from
 * col_bwt.hpp:434-437) or the path of the .col_pml file itself.  `widths` may
 * be NULL (shipped widths).  Unlike the reference (UB on a bad file) the
 * loader validates: size == r, file length == 32 + 18*size, idx strictly
 * increasing from 0 and < n, interval < r. */
int colbwt_index_open(const char *prefix_or_file, const colbwt_widths *widths, int device,
                      colbwt_index **out);
/* Same over an in-memory image of the .col_pml file. */
int colbwt_index_open_memory(const void *col_pml_bytes, uint64_t len, const colbwt_widths *widths,
                             int device, colbwt_index **out);
/* HBM table layouts (results are identical).  ONE_STEP:
 * 16-byte rows, one LF step per row load.  TWO_STEP / THREE_STEP: rows split at
 * the pre-images of row boundaries (allowed: the query is a function of BWT
 * positions) so that a row also knows the characters / col ids of the next one /
 * two steps and the landings of LF^2 / LF^3 -- one 128-byte line fill serves up
 * to K bases while the read keeps matching; about 4x / 7x the HBM footprint.
 * AUTO = the engine's choice: line rows when they can be built (fewer than 2^32-1
 * refined rows at every level, enough HBM), else the deepest K-step layout that can. */
#define COLBWT_LAYOUT_AUTO 0
#define COLBWT_LAYOUT_ONE_STEP 1
#define COLBWT_LAYOUT_TWO_STEP 2
#define COLBWT_LAYOUT_THREE_STEP 3
/* LINE_ROWS: one whole 128-byte line per row, fetched lane-cooperatively (8 lanes x 16 bytes per
 * row, one instruction): up to 8 look-ahead steps and, for the three most frequent other
 * characters, where a mismatch re-orients to and what the next step meets from there -- a
 * mismatch costs no line fill of its own.  About 4x the footprint of THREE_STEP.  The depth
 * (number of look-ahead steps K, 4..8) may be given; 0 = the engine's default. */
#define COLBWT_LAYOUT_LINE_ROWS 4
#define COLBWT_LAYOUT_LINE_ROWS_STEPS(K) (COLBWT_LAYOUT_LINE_ROWS | ((K) << 8))
/* MISMATCH_LINES: line rows whose mismatch information lives in a table of its own, one 64-byte
 * entry per (row of the file cut at its thresholds, character): an entry resolves the mismatching
 * base and the base after it whatever that base is, and a stretch of mismatching bases goes from
 * entry to entry -- a line fill per two bases instead of one per base.  LINE_ROWS + ~190 bytes per
 * row of the file. */
#define COLBWT_LAYOUT_MISMATCH_LINES 5
#define COLBWT_LAYOUT_MISMATCH_LINES_STEPS(K) (COLBWT_LAYOUT_MISMATCH_LINES | ((K) << 8))
/* MISMATCH_LINES_DEEP: the same with 128-byte entries that also resolve the base after those two when
 * it matches -- between two mismatches of a stretch there is often exactly one matching base, and
 * with it resolved in the entry the lane goes from entry to entry.  MISMATCH_LINES + ~190 bytes per
 * row of the file.  Measured on C2: 37.1 line fetches per read instead of 40.7, 10.6-11.0 ms per
 * launch instead of 11.5-11.6.  AUTO takes the deep entries when the table with
 * them still leaves 32 GB of what the open could allocate (HBM, COLBWT_HBM_BUDGET_MB) for batches
 * and their results, the 64-byte entries otherwise; asking for MISMATCH_LINES gets the 64-byte ones. */
#define COLBWT_LAYOUT_MISMATCH_LINES_DEEP 6
#define COLBWT_LAYOUT_MISMATCH_LINES_DEEP_STEPS(K) (COLBWT_LAYOUT_MISMATCH_LINES_DEEP | ((K) << 8))
int colbwt_index_open_layout(const char *prefix_or_file, const colbwt_widths *widths, int device, int layout,
                             colbwt_index **out);
int colbwt_index_open_memory_layout(const void *col_pml_bytes, uint64_t len, const colbwt_widths *widths,
                                    int device, int layout, colbwt_index **out);
/* The table replicated on several devices -- the `device_mask` of SURVEY.md 8(b) as a list, so a
 * device may appear twice (two replicas in one HBM).  The reference processes its reads one
 * after the other in one process (pml_query.cpp:74-86); here the host entry points
 * (colbwt_query_batch[_u32], colbwt_query_file) cut every batch into contiguous shards of equal
 * base count, one per replica, queried side by side and copied straight into the caller's arrays:
 * no exchange between devices.  Every replica gets the layout the first one ended up with.  The
 * device-resident entry points address the replica on the buffers' device. */
int colbwt_index_open_devices(const char *prefix_or_file, const colbwt_widths *widths, const int *devices,
                              int n_devices, int layout, colbwt_index **out);
int colbwt_index_open_memory_devices(const void *col_pml_bytes, uint64_t len, const colbwt_widths *widths,
                                     const int *devices, int n_devices, int layout, colbwt_index **out);
void colbwt_index_close(colbwt_index *idx);
int colbwt_index_info(const colbwt_index *idx, colbwt_info *out);

/* col_pml::query_pml for a batch of reads held in HOST memory
 * (col_bwt.hpp:409-412 -> :460-472 -> :498-529).  Read k is
 * bases[read_off[k] .. read_off[k+1]); read_off has n_reads+1 entries,
 * read_off[0] == 0.  Outputs are indexed like `bases`.  PML is u16; a batch
 * containing a read longer than 65535 must use the _u32 form (ERR_ARG
 * otherwise).  `stats` may be NULL. */
int colbwt_query_batch(colbwt_index *idx, const uint8_t *bases, const uint64_t *read_off,
                       uint64_t n_reads, uint16_t *pml, uint8_t *cid, colbwt_stats *stats);
int colbwt_query_batch_u32(colbwt_index *idx, const uint8_t *bases, const uint64_t *read_off,
                           uint64_t n_reads, uint32_t *pml, uint8_t *cid, colbwt_stats *stats);

/* Same computation with every buffer already resident in HBM on the index's
 * device (the benchmark / multi-GPU path).  Requirements: d_bases has at least
 * 64 readable bytes past read_off[n_reads] (the kernel reads whole 64-byte
 * blocks); d_bases and d_cid are 16-byte aligned, d_pml 32-byte aligned;
 * pml_bytes is 2 or 4 (2 needs every read <= 65535 bases).  `hip_stream` is a
 * hipStream_t (NULL = default stream); the call is asynchronous unless
 * `stats` is non-NULL, in which case it records HIP events on the stream,
 * synchronises it and fills kernel_ms.
 * Concurrency contract: a launch keeps no state outside its arguments (the
 * kernels' work counters live in LDS), so ANY number of calls may be in flight
 * on one index at once -- from any host threads, on any streams -- as long as
 * their output buffers are distinct; there is no bound to respect and nothing
 * to serialise (tests/test_gpu_parity.py queues 48 launches on 24 streams). */
int colbwt_query_device(colbwt_index *idx, const uint8_t *d_bases, const uint64_t *d_read_off,
                        uint64_t n_reads, uint64_t n_bases, void *d_pml, int pml_bytes,
                        uint8_t *d_cid, void *hip_stream, colbwt_stats *stats);

/* Same with an explicit lane assignment for ragged batches: d_order (device,
 * n_reads entries, nullable) lists the read indices by decreasing length, so the
 * 64 lanes of a wave walk reads of similar length.  Results are identical with
 * or without it (each read's values only depend on that read); the host entry
 * points build the order themselves when a batch is ragged.  d_order is
 * ADVISORY: the K-step and one-step layouts assign lanes by it, the line-row
 * layout (COLBWT_LAYOUT_LINE_ROWS, the default) ignores it -- its persistent
 * lanes claim chunks of consecutive reads and balance ragged batches themselves. */
int colbwt_query_device_ordered(colbwt_index *idx, const uint8_t *d_bases, const uint64_t *d_read_off,
                                uint64_t n_reads, uint64_t n_bases, void *d_pml, int pml_bytes,
                                uint8_t *d_cid, const uint32_t *d_order, void *hip_stream,
                                colbwt_stats *stats);

/* pml_query in vec mode (pml_query.cpp:92-143): reads FASTA/FASTQ (optionally
 * gzip) from pattern_path, writes text pml_path / cid_path (NULL => pattern +
 * ".pml" / ".cid", pml_query.cpp:124-125) in the byte format of pml_to_vec
 * (pml_query.cpp:78-85).  `batch_bases` bounds the bases per GPU batch
 * (0 = default). */
int colbwt_query_file(colbwt_index *idx, const char *pattern_path, const char *pml_path,
                      const char *cid_path, uint64_t batch_bases, colbwt_stats *stats);

/* The same program writing the results as binary containers -- what `col-bwt query` produces
 * (scripts/col-bwt.py:194-198: PATTERN.split.pml.bin / .split.cid.bin, written there by the
 * un-vendored Movi fork).  Record shape as SURVEY.md 8(c) recalls it of upstream Movi -- "Movi-like,
 * UNVERIFIED", parity unpinned: per read  u16 name_len | name | u64 count | count values,
 * values in computation order (the read's last base first); u16 lengths (saturated at 65535) in
 * pml_bin_path (NULL => pattern + ".pml.bin"), u8 col ids in cid_bin_path (".cid.bin").  The text
 * files stay the bit-exact contract; this exists because formatting ~4 text bytes per base
 * bounds pml_query end to end. */
int colbwt_query_file_binary(colbwt_index *idx, const char *pattern_path, const char *pml_bin_path,
                             const char *cid_bin_path, uint64_t batch_bases, colbwt_stats *stats);
/* Container -> the reference's text (pml_query.cpp:78-85): `col-bwt view`.  value_bytes: 2 for
 * .pml.bin, 1 for .cid.bin.  Host code. */
int colbwt_binary_to_text(const char *bin_path, int value_bytes, const char *text_path);

/* ---- index construction (SURVEY.md 8(f) "next" #1) ------------------------ */

/* build_col_bwt <prefix> (src/build_col_bwt.cpp:14-52): reads <prefix>.bwt.heads,
 * .bwt.len, .col_ids, .col_runs (plain sdsl::bit_vector as col_split writes it,
 * col_split.hpp:384-386), .thr_pos and writes <prefix>.col_pml (or out_path)
 * exactly as col_pml(heads, lengths, col_ids, thresholds, splits) + serialize
 * would (col_bwt.hpp:124-230, 391-395, 440-457, 360-370).  Host code. */
int colbwt_build_col_pml(const char *prefix, const char *out_path);
/* Same over decoded arrays: split_pos = ascending positions of the set bits of
 * .col_runs; lens / thr_pos already widened from 5 bytes.  *out_len receives
 * the image size (also when out is NULL / too small, then ERR_ARG). */
int colbwt_build_col_pml_arrays(const uint8_t *heads, uint64_t n_heads, const uint64_t *lens,
                                const uint8_t *col_ids, uint64_t n_ids, const uint64_t *split_pos,
                                uint64_t n_splits, const uint64_t *thr_pos, uint64_t n_thr, void *out,
                                uint64_t out_cap, uint64_t *out_len);

/* ---- sub-run splitting (SURVEY.md 8(f) "next" #2) -------------------------- */

/* build_FL + col_split (src/build_FL.cpp:27-74, src/col_split.cpp:62-140; include/ds/FL_table.hpp,
 * include/col_split.hpp): multi-MUMs -> where sub-runs start and their chain statistic.
 * `col_split <prefix> -m tunnels|all -s <rate>`: reads <prefix>.bwt.heads, .bwt.len and .col_mums
 * (5-byte num_docs, then 5-byte (length, position) pairs; position = rank in F / suffix-array
 * order of the first of num_docs consecutive suffixes, ascending), writes <prefix>.col_runs (a plain
 * sdsl::bit_vector: u64 length in bits + words) and <prefix>.col_ids (one byte per set bit) --
 * the inputs of colbwt_build_col_pml.  The FL table is rebuilt from the RLBWT instead of read from
 * the reference's .FL_table file (which embeds an sdsl sd_vector).  Every multi-MUM is FL-stepped on
 * the device (the reference steps them one after the other, twice); the overlap sweep
 * (find_col_runs, col_split.hpp:258-342) runs on the host.  The reference's -o overlap option is
 * parsed there but never used (col_split.hpp:215), so it has no counterpart.  `all` mode: up to
 * 1024 documents.  Needs 4 (tunnels) or 8 (all) bytes of HBM per BWT position. */
#define COLBWT_SPLIT_TUNNELS 0
#define COLBWT_SPLIT_ALL 1
int colbwt_col_split(const char *prefix, int mode, int split_rate, int device);
/* Same over decoded arrays; results as colbwt_build_col_pml_arrays takes them: split_pos = the
 * ascending positions of the set bits of .col_runs (room for `cap`), col_ids one byte each,
 * *n_split their number (also when the arrays are too small, then ERR_ARG), *bwt_len = n. */
int colbwt_col_split_arrays(const uint8_t *heads, uint64_t n_heads, const uint64_t *lens, const uint64_t *mum_len,
                            const uint64_t *mum_pos, uint64_t n_mums, uint32_t num_docs, int mode, int split_rate,
                            int device, uint64_t *split_pos, uint64_t cap, uint64_t *n_split, uint8_t *col_ids,
                            uint64_t *bwt_len);
const char *colbwt_col_split_error(void);

/* ---- RLBWT, thresholds and multi-MUMs from FASTA (SURVEY.md 8(f) "next" #4) ---- */

/* What the reference's driver takes from `mumemto mum -K -R -T -l <min> [-r]` (scripts/col-bwt.py:
 * 121-145): <prefix>.bwt.heads (one character per BWT run), .bwt.len (5-byte run lengths),
 * .thr_pos (5-byte threshold position per run, col_bwt.hpp:446-448) and .col_mums (5-byte num_docs,
 * then 5-byte (length, position) pairs, col_split.cpp:90-106) -- the inputs of colbwt_col_split and
 * colbwt_build_col_pml.  mumemto itself is an un-vendored dependency, so its conventions are
 * restated here and UNVERIFIED against it ("parity unpinned"):
 *   text        every record of every file as it is + separator 1 (+ its reverse complement + 1 when
 *               `revcomp`; one document per file), then one final 0; suffixes compare byte-wise
 *   runs        maximal stretches of equal characters with every byte <= 1 (the final 0, the
 *               separators) taken as ONE character, head 1 -- the class the reference folds them to
 *               when it reads .bwt.heads (col_bwt.hpp:167-171) and groups thresholds by
 *               (col_bwt.hpp:446-451): entry k of .thr_pos belongs to group k of the builder
 *   threshold   first position of the minimum LCP in (end of the previous run of the character,
 *               head of this run]; 0 for a character's first run.  LCPs are cut at the first
 *               separator: what lies behind one can never be matched by a pattern
 *   multi-MUM   num_docs consecutive suffixes, one from every document, that share >= min_mum
 *               characters (never across a separator), more than with either neighbouring suffix,
 *               and are not all preceded by the same character; position = suffix-array rank of
 *               the first, length = the shared prefix
 * The suffix array (prefix doubling over rocPRIM radix sorts), the LCP array, the runs, the
 * thresholds and the multi-MUM scan all run on the device.  Text < 2^32-1 characters, <= 4096
 * documents; HBM 29 bytes per character + 4 per doubling round. */
typedef struct colbwt_rlbwt colbwt_rlbwt;
typedef struct colbwt_rlbwt_view {
    uint64_t n;        /* BWT length */
    uint64_t n_runs;
    uint64_t n_mums;
    uint32_t n_docs;
    int32_t rounds;    /* prefix-doubling rounds the suffix sort took */
    const uint8_t *heads;      /* n_runs */
    const uint64_t *lens;      /* n_runs */
    const uint64_t *thr_pos;   /* n_runs */
    const uint64_t *mum_len;   /* n_mums */
    const uint64_t *mum_pos;   /* n_mums, ascending */
} colbwt_rlbwt_view;
/* text[0..n): separators 1 in place, text[n-1] its only 0; doc_start[d] = first character of
 * document d, ascending from 0. */
int colbwt_rlbwt_build_text(const uint8_t *text, uint64_t n, const uint64_t *doc_start, uint32_t n_docs,
                            uint64_t min_mum, int device, colbwt_rlbwt **out);
/* FASTA/FASTQ(.gz) files, one document each; writes the four files when out_prefix is not NULL,
 * hands the result back when `out` is not NULL. */
int colbwt_rlbwt_build_files(const char *const *fastas, uint32_t n_files, int revcomp, uint64_t min_mum, int device,
                             const char *out_prefix, colbwt_rlbwt **out);
void colbwt_rlbwt_get(const colbwt_rlbwt *h, colbwt_rlbwt_view *view);   /* pointers live until _free */
void colbwt_rlbwt_free(colbwt_rlbwt *h);
const char *colbwt_rlbwt_error(void);

/* ---- multi-GPU gather codec (the path's one exchange step) -----------------
 * The reference has no counterpart: its reads are processed by one process
 * (pml_query.cpp:74).  With the reads sharded over N GPUs the per-base results
 * are gathered on rank 0; the PML values of a read are determined by where
 * they are zero (length + 1 per match, 0 at a mismatch: col_bwt.hpp:516-521),
 * so a rank sends one bit per base and rank 0 rebuilds the 16-bit values
 * (the col ids: see colbwt_cid_pack_device below).
 * Device pointers; masks are uint32 words, bit b of word w = base 32w + b of
 * the rank's concatenated reads; all calls are asynchronous on `hip_stream`.
 *   pack     d_mask[(n_bases+31)/32] <- (d_pml[k] == 0); d_pml 32-byte aligned
 *   end mask bit (read_off[r+1]-1) set for every non-empty read; d_mask must be
 *            zeroed by the caller, (n_bases+31)/32 words
 *   unpack   rebuilds d_pml[32*first_word .. 32*(first_word+n_words)) from the
 *            zero mask and the end mask (both total_words long; the range must
 *            end at a read end or at total_words); d_pml needs room for whole
 *            32-value blocks and 64-byte alignment */
int colbwt_pml_pack_device(const uint16_t *d_pml, uint64_t n_bases, uint32_t *d_mask, void *hip_stream);
/* The col ids of the results are ids that rows of the table hold (col_bwt.hpp:513), and every rank
 * holds the same table: they travel as codes of that dictionary, colbwt_cid_code_bits(n_ids) =
 * max(1, ceil(log2(n_ids))) bits per base -- the same on every rank, so the gathers keep equal,
 * known sizes -- laid out as bit planes: `bits` words per 32 bases, word p = bit p of the 32 codes.
 * 7 distinct ids (the C2 index): 3 bits, so results travel at 0.5 bytes per base instead of 3.
 *   dictionary  ids[0 .. *n_ids) <- the distinct col ids of the table's rows, ascending (ids: 256 bytes)
 *   pack        d_planes[bits * (n_bases+31)/32] <- codes of d_cid[0 .. n_bases) (16-byte aligned);
 *               `ids` is a HOST array (the dictionary), an id outside it packs as code 0
 *   unpack      d_cid[32*first_word .. 32*(first_word+n_words)) <- ids of the codes in the planes of
 *               those words (d_planes is the whole array: word w's planes at d_planes[bits * w]);
 *               d_cid 32-byte aligned with room for whole 32-id blocks */
int colbwt_index_cid_dictionary(const colbwt_index *idx, uint8_t *ids, uint32_t *n_ids);
uint32_t colbwt_cid_code_bits(uint32_t n_ids);
int colbwt_cid_pack_device(const uint8_t *d_cid, uint64_t n_bases, const uint8_t *ids, uint32_t n_ids, uint32_t *d_planes,
                           void *hip_stream);
int colbwt_cid_unpack_device(const uint32_t *d_planes, uint64_t first_word, uint64_t n_words, const uint8_t *ids, uint32_t n_ids,
                             uint8_t *d_cid, void *hip_stream);
int colbwt_read_end_mask_device(const uint64_t *d_read_off, uint64_t n_reads, uint32_t *d_mask, void *hip_stream);
int colbwt_pml_unpack_device(const uint32_t *d_zero_mask, const uint32_t *d_end_mask, uint64_t first_word,
                             uint64_t n_words, uint64_t total_words, uint16_t *d_pml, void *hip_stream);

/* ---- synthetic inputs (benchmark / test generators) ------ */

/* Bytes needed for a synthetic .col_pml image of `rows` rows. */
uint64_t colbwt_synth_index_bytes(uint64_t rows);
/* Direct move-table synthesis: adjacent-distinct ACGT heads (+ one 0x01 run of
 * length 1 at row rows/2), Geometric(mean_len) lengths, (interval, offset)
 * from the stable char-sorted F order, col ids from {0,0,0,1,2,3,17,200,255},
 * thresholds uniform in [0,n).  `split_permille`: per-mille probability that a
 * row repeats the previous row's character (a sub-run split; 0 => bwt_r == r).
 * Writes the file image into `out` (colbwt_synth_index_bytes(rows) bytes). */
int colbwt_synth_index(uint64_t rows, uint32_t mean_len, uint32_t split_permille, uint64_t seed,
                       void *out, uint64_t out_len);
/* As above with a choice of where the thresholds fall: UNIFORM in [0, n) (the
 * SURVEY.md 8(d) recipe), or BETWEEN_RUNS: between the previous run of the same
 * character and the run's head, as in a real index (mumemto's .thr_pos). */
#define COLBWT_SYNTH_THR_UNIFORM 0
#define COLBWT_SYNTH_THR_BETWEEN_RUNS 1
int colbwt_synth_index_thr(uint64_t rows, uint32_t mean_len, uint32_t split_permille, uint64_t seed,
                           int thr_mode, void *out, uint64_t out_len);
/* Backward-walk reads sampled ON THE DEVICE from the loaded index:
 * read[m-1-k] = char at LF^k(p0), p0 uniform; 0x01 -> 'A'; substitutions at
 * `sub_permille`/1000.  Writes n_reads*read_len bytes (+64 pad bytes zeroed)
 * to d_bases and n_reads+1 offsets to d_read_off (both device pointers). */
int colbwt_synth_reads_device(colbwt_index *idx, uint64_t n_reads, uint32_t read_len,
                              uint32_t sub_permille, uint64_t seed, uint8_t *d_bases,
                              uint64_t *d_read_off, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* COLBWT_H */
