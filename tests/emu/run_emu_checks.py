#!/usr/bin/env python3
"""CPU-tier parity of the product's HIP sources compiled against the SIMT
emulator (tests/emu/hip/hip_runtime.h) vs the oracle.  Run by
tests/test_emu_parity.py in a subprocess with libasan preloaded, so every
out-of-bounds access of the kernels / host logic is fatal.

This is a test instrument: the shipped libcolbwt.so has no host path.
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from __graft_entry__ import load_oracle, load_package  # noqa: E402
import helpers  # noqa: E402

pkg = load_package()
pkg.LIB_PATH = os.path.join(HERE, "libcolbwt_emu.so")   # emulated build instead of the HIP one
# opens that leave the layout to the engine use two-step rows here: the emulator runs one OS
# thread per lane, and the kernels with persistent lanes (three-step rows, line rows: every lane of
# a workgroup stays until the last read is done) are exercised by the explicit cases below
os.environ["COLBWT_LAYOUT"] = "2"
oracle = load_oracle()
GOLD = os.path.join(ROOT, "tests", "golden")


LINE_ROWS = 4                     # include/colbwt.h COLBWT_LAYOUT_LINE_ROWS (| steps << 8)
MIS_LINES = 5                     # COLBWT_LAYOUT_MISMATCH_LINES (| steps << 8)
MIS_DEEP = 6                      # COLBWT_LAYOUT_MISMATCH_LINES_DEEP (| steps << 8)


def check(image, reads, label, wide=False, extra_layouts=(MIS_LINES,), base_layouts=(1, 2, 3)):
    image = bytes(image)
    bases, off = helpers.concat_reads(reads)
    ref = oracle.OracleIndex(image)
    epml, ecid = ref.query_batch(bases, off, wide=wide)
    for layout in tuple(base_layouts) + tuple(extra_layouts):   # one-step rows / K-step refined rows / line rows: identical results
        tbl = pkg.ColPml.from_bytes(image, layout=layout)
        assert tbl.info().layout == layout & 0xFF
        pml, cid, _ = tbl.query_batch(bases, off, wide=wide)
        assert np.array_equal(pml, epml), f"{label}/L{layout}: PML differs at {np.flatnonzero(pml != epml)[:5]}"
        assert np.array_equal(cid, ecid), f"{label}/L{layout}: CID differs at {np.flatnonzero(cid != ecid)[:5]}"
        rows2 = tbl.info().table_rows
        tbl.close()
    print(f"ok {label}: {len(reads)} reads, {int(off[-1])} bases (three-step table: {rows2} rows)")


def rand_reads(rng, n, lo, hi, alphabet=b"ACGT"):
    return [rng.choice(np.frombuffer(alphabet, np.uint8), size=int(m)) for m in rng.integers(lo, hi + 1, size=n)]


def main():
    rng = np.random.default_rng(11)

    # 1. golden KAT
    img = open(os.path.join(GOLD, "kat_d.col_pml"), "rb").read()
    kat = [np.frombuffer(s, np.uint8) for s in (b"GATTACA", b"TTACCGATNACA", b"CCCC")]
    check(img, kat, "kat_d")
    tbl = pkg.ColPml.from_bytes(img)
    exp = {b"GATTACA": ([5, 4, 3, 2, 1, 0, 1], [1, 3, 1, 3, 3, 1, 3]),
           b"CCCC": ([0, 0, 1, 0], [3, 3, 0, 3])}
    for s, (p, c) in exp.items():
        gp, gc = tbl.query_pml(s)
        assert gp.tolist() == p and gc.tolist() == c
    # text path: FASTA in, .pml/.cid out, byte-identical to the reference's files
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "kat_d.fa")
        open(fa, "wb").write(open(os.path.join(GOLD, "kat_d.fa"), "rb").read())
        tbl.query_file(fa)
        for ext in (".pml", ".cid"):
            assert open(fa + ext, "rb").read() == open(os.path.join(GOLD, "kat_d.fa" + ext), "rb").read(), ext
    tbl.close()
    print("ok kat_d text")

    # 1b. the pipelined file path: several batches, multi-threaded formatting (> 1 MiB of
    #     values per batch), values of five digits
    img = pkg.synth_index(3000, mean_len=6, split_permille=50, seed=21)
    reads = helpers.backward_walk_reads(img, 1500, 750, 0.002, seed=21)
    reads += helpers.backward_walk_reads(img, 1, 11000, 0.0, seed=22) + rand_reads(rng, 200, 0, 300)
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "big.fa")
        helpers.write_fasta(fa, reads, width=80)
        tbl = pkg.ColPml.from_bytes(img)
        st = tbl.query_file(fa, batch_bases=450_000)
        assert st.n_reads == len(reads)
        oracle.OracleIndex(bytes(img)).pml_query_files(fa, fa + ".opml", fa + ".ocid")
        for ext in ("pml", "cid"):
            assert open(fa + "." + ext, "rb").read() == open(fa + ".o" + ext, "rb").read(), ext
        assert b" 10000 " in open(fa + ".pml", "rb").read()
        tbl.close()
    print("ok big text")

    # 1b'. binary containers and the many-threaded FASTA reader: a file whose tail turns into FASTQ
    #      (qualities that start with '>'), CRLF line ends; text == oracle, container -> text == text
    img = pkg.synth_index(2500, mean_len=6, split_permille=50, seed=23)
    rds = helpers.backward_walk_reads(img, 260, 90, 0.02, seed=23) + rand_reads(rng, 80, 0, 130)
    with tempfile.TemporaryDirectory() as d:
        mixed = os.path.join(d, "mixed.fx")
        with open(mixed, "wb") as f:
            for k, rd in enumerate(rds[:200]):
                f.write(b">m%d c\r\n" % k + bytes(rd[:40]) + b"\r\n" + bytes(rd[40:]) + b"\r\n")
            for k, rd in enumerate(rds[200:]):
                f.write(b"@q%d\n" % k + bytes(rd) + b"\n+\n" + b">" * len(rd) + b"\n")
        tbl = pkg.ColPml.from_bytes(img)
        tbl.query_file(mixed, batch_bases=6000)
        tbl.query_file_binary(mixed, batch_bases=6000)
        oracle.OracleIndex(bytes(img)).pml_query_files(mixed, mixed + ".opml", mixed + ".ocid")
        for ext, width in (("pml", 2), ("cid", 1)):
            text = open(f"{mixed}.{ext}", "rb").read()
            assert text == open(f"{mixed}.o{ext}", "rb").read(), ext
            pkg.binary_to_text(f"{mixed}.{ext}.bin", width, f"{mixed}.{ext}.view")
            assert open(f"{mixed}.{ext}.view", "rb").read() == text, ext
        two = pkg.ColPml.from_bytes(img, devices=[0, 0])       # two replicas: every batch in two shards
        assert two.info().n_devices == 2
        two.query_file(mixed, mixed + ".pml2", mixed + ".cid2", batch_bases=6000)
        assert open(mixed + ".pml2", "rb").read() == open(mixed + ".pml", "rb").read()
        assert open(mixed + ".cid2", "rb").read() == open(mixed + ".cid", "rb").read()
        tbl.close(), two.close()
    print("ok binary containers, FASTA -> FASTQ hand-over, two replicas")

    # 1c. gather codec (multi-GPU exchange step): PML values <-> one bit per base
    def aligned(n, dt):
        raw = np.zeros(n * np.dtype(dt).itemsize + 64, np.uint8)
        o = (-raw.ctypes.data) % 64
        return raw[o:o + n * np.dtype(dt).itemsize].view(dt)
    creads = reads[:300] + [np.zeros(0, np.uint8)] * 3 + rand_reads(rng, 40, 1, 5)
    cb, coff = helpers.concat_reads(creads)
    epml, _ = oracle.OracleIndex(bytes(img)).query_batch(cb, coff)
    nb = int(coff[-1])
    nw = (nb + 31) // 32
    d_pml = aligned(nw * 32, np.uint16)
    d_pml[:nb] = epml
    d_pml[nb:] = 7                                   # padding must not leak into the mask
    zero, last = aligned(nw, np.uint32), aligned(nw, np.uint32)
    d_off = aligned(len(coff), np.uint64)
    d_off[:] = coff
    pkg.pml_pack_device(d_pml.ctypes.data, nb, zero.ctypes.data)
    pkg.read_end_mask_device(d_off.ctypes.data, len(creads), last.ctypes.data)
    bits = np.unpackbits(zero.view(np.uint8), bitorder="little")[:nb]
    assert np.array_equal(bits.astype(bool), epml == 0)
    ends = np.unpackbits(last.view(np.uint8), bitorder="little")[:nb]
    assert ends.sum() == sum(1 for r in creads if len(r)) and all(ends[int(e) - 1] for e in coff[1:] if e > 0)
    out = aligned(nw * 32, np.uint16)
    cutw = int(coff[150]) // 32                      # a word boundary inside a read: two calls
    cut_read_end = None
    for e in coff[100:]:
        if int(e) % 32 == 0 and e > 0:
            cut_read_end = int(e) // 32
            break
    first = cut_read_end if cut_read_end else 0      # chunked: [0, first) then [first, nw)
    del cutw
    pkg.pml_unpack_device(zero.ctypes.data, last.ctypes.data, 0, first, nw, out.ctypes.data)
    pkg.pml_unpack_device(zero.ctypes.data, last.ctypes.data, first, nw - first, nw, out.ctypes.data)
    assert np.array_equal(out[:nb], epml), np.flatnonzero(out[:nb] != epml)[:10]
    print(f"ok gather codec: {len(creads)} reads, {nb} bases, split at word {first}")
    # ... and the col ids as codes of the table's dictionary of ids, bit planes
    _, ecid = oracle.OracleIndex(bytes(img)).query_batch(cb, coff)
    tblc = pkg.ColPml.from_bytes(bytes(img))
    ids = tblc.cid_dictionary()
    tblc.close()
    assert ids.tolist() == sorted(set(helpers.unpack_col_pml(bytes(img))["cid"].tolist())) and set(ecid.tolist()) <= set(ids.tolist())
    for dict_ids in (ids, np.arange(256, dtype=np.uint8)):
        cbits = pkg.cid_code_bits(len(dict_ids))
        d_cid = aligned(nw * 32, np.uint8)
        d_cid[:nb] = ecid
        planes = aligned(cbits * nw, np.uint32)
        back = aligned(nw * 32, np.uint8)
        pkg.cid_pack_device(d_cid.ctypes.data, nb, dict_ids, planes.ctypes.data)
        pkg.cid_unpack_device(planes.ctypes.data, 0, first, dict_ids, back.ctypes.data)
        pkg.cid_unpack_device(planes.ctypes.data, first, nw - first, dict_ids, back.ctypes.data)
        assert np.array_equal(back[:nb], ecid), (cbits, np.flatnonzero(back[:nb] != ecid)[:10])
    print(f"ok col-id codec: {len(ids)} ids in the table's dictionary, {pkg.cid_code_bits(len(ids))} bit planes")

    # 2. true BWT index, reads with substitutions, N and lowercase (no case folding)
    seqs = []
    base = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=300)
    for _ in range(4):
        s = base.copy()
        mut = rng.random(300) < 0.03
        s[mut] = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(mut.sum()))
        seqs.append(bytes(s))
    img, text = helpers.true_bwt_index(seqs, seed=5, extra_splits=40)
    reads = helpers.reads_from_text(text, 60, (1, 120), 0.05, seed=6, extra=b"Nacgt")
    reads += [np.zeros(0, np.uint8), np.frombuffer(b"A", np.uint8)]       # empty + 1-base reads
    reads += rand_reads(rng, 20, 1, 40)
    check(img, reads, "true_bwt", extra_layouts=(LINE_ROWS, MIS_LINES, MIS_DEEP))

    # 3. synthetic tables spanning several jump blocks; sub-run splits
    for rows, split, seed in ((700, 0, 1), (3000, 150, 2), (257, 300, 3), (256, 0, 4)):
        img = pkg.synth_index(rows, mean_len=6, split_permille=split, seed=seed)
        reads = helpers.backward_walk_reads(img, 40, 70, 0.02, seed=seed)
        reads += rand_reads(rng, 40, 0, 90)
        reads += rand_reads(rng, 10, 1, 50, alphabet=b"ACGTN\x01")        # absent byte + terminator
        check(img, reads, f"synth_{rows}_{split}", extra_layouts={700: (LINE_ROWS, MIS_LINES | (6 << 8), MIS_DEEP | (7 << 8)), 257: (LINE_ROWS | (5 << 8), MIS_LINES | (4 << 8), MIS_DEEP | (4 << 8))}.get(rows, ()))
    img = pkg.synth_index(2500, mean_len=6, split_permille=50, seed=8, thr_mode=1)   # thresholds inside rows: cut out
    check(img, helpers.backward_walk_reads(img, 60, 80, 0.05, seed=8) + rand_reads(rng, 30, 0, 90), "synth_thr_between_runs",
          extra_layouts=(MIS_LINES, MIS_DEEP | (4 << 8)))                            # origin rows cut at thresholds; depth K straddles

    # 4. rare character far away: scans must leave the block and use the jump tables
    r = 2000
    chars = np.tile(np.frombuffer(b"AC", np.uint8), r // 2)
    chars[3] = ord("G"); chars[1500] = ord("G"); chars[900] = 1
    lens = rng.integers(1, 9, size=r)
    idx = np.concatenate(([0], np.cumsum(lens)[:-1]))
    n = int(lens.sum())
    interval, offset = helpers.lf_columns(chars, idx, n)
    thr = rng.integers(0, n, size=r)
    cid = rng.integers(0, 256, size=r)
    img = helpers.pack_col_pml(r, n, chars, idx, interval, offset, cid, thr)
    reads = rand_reads(rng, 60, 1, 60, alphabet=b"ACGGG\x01T")
    check(img, reads, "rare_char")

    # 4b. threshold hints: thresholds inside rows (sigma <= 5, hints on) and sigma = 7 (hints off)
    for label, alpha in (("hints_sigma4", b"ACGT"), ("hints_sigma5", b"\x01ACGT"), ("nohints_sigma7", b"\x01ACGNTac")):
        img = helpers.random_table(rng, 1500, alphabet=alpha)
        check(img, rand_reads(rng, 80, 1, 70, alphabet=alpha + b"N"), label,
              extra_layouts=(LINE_ROWS, MIS_LINES, MIS_DEEP) if label == "nohints_sigma7" else (MIS_LINES | (5 << 8),))

    # 5. long runs: len >= 65535 (len16 escape) incl. the last row, offsets near 2^16
    r = 600
    chars = np.tile(np.frombuffer(b"ACGT", np.uint8), r // 4)
    lens = rng.integers(1, 50, size=r)
    lens[[5, 77, 300, r - 1]] = [65535, 70000, 200000, 66000]
    idx = np.concatenate(([0], np.cumsum(lens)[:-1]))
    n = int(lens.sum())
    interval, offset = helpers.lf_columns(chars, idx, n)
    offset = offset & np.uint64(0xFFFF)            # what the 16-bit field keeps (LF_table.hpp:39)
    thr = rng.integers(0, n, size=r)
    img = helpers.pack_col_pml(r, n, chars, idx, interval, offset, rng.integers(0, 256, size=r), thr)
    check(img, rand_reads(rng, 50, 1, 80), "long_runs", extra_layouts=(LINE_ROWS, MIS_LINES))

    # 6. u32 PML: a read longer than 65535 bases
    img = pkg.synth_index(500, mean_len=4, split_permille=0, seed=9)
    long_read = helpers.backward_walk_reads(img, 1, 66000, 0.0005, seed=10)
    # (one lane per read and no persistent lanes: the kernels of the one- and two-step rows; the others
    # take the same code path on the short reads below)
    check(img, long_read + rand_reads(rng, 3, 1, 30), "wide_pml", wide=True, extra_layouts=(), base_layouts=(1, 2))
    short = [long_read[0][-900:]] + rand_reads(rng, 3, 1, 30)             # the u32 kernels of the line rows, small
    check(img, short, "wide_pml_line_rows", wide=True, extra_layouts=(LINE_ROWS, MIS_LINES, MIS_DEEP))
    try:
        bases, off = helpers.concat_reads(long_read)
        pkg.ColPml.from_bytes(bytes(img)).query_batch(bases, off, wide=False)
        raise SystemExit("u16 call with a 66000-base read must fail")
    except pkg.ColbwtError as e:
        assert e.code == -1

    # 7. loader validation (the reference has none: UB on a bad file)
    good = bytearray(pkg.synth_index(300, mean_len=5, seed=12).tobytes())
    for label, mutate in (
            ("short", lambda b: b[:-1]),
            ("size!=r", lambda b: b[:24] + (299).to_bytes(8, "little") + b[32:]),
            ("interval>=r", lambda b: b[:32 + 18 * 10 + 6] + (10**6).to_bytes(4, "little") + b[32 + 18 * 10 + 10:]),
            ("idx order", lambda b: b[:32 + 18 * 20 + 1] + (0).to_bytes(5, "little") + b[32 + 18 * 20 + 6:])):
        bad = bytes(mutate(bytes(good)))
        try:
            pkg.ColPml.from_bytes(bad)
            raise SystemExit(f"loader accepted a corrupt image ({label})")
        except pkg.ColbwtError as e:
            assert e.code == -3, (label, e)
    print("ok loader validation")

    # 8. the read sampler gives the same reads from every layout (the K-step index no longer
    #    holds the one-step tables)
    img = pkg.synth_index(1200, mean_len=6, split_permille=80, seed=31)
    got = []
    for layout in (1, 2, 3, 4):
        tbl = pkg.ColPml.from_bytes(img, layout=layout)
        b = np.zeros(300 * 90 + 64, np.uint8)
        o = np.zeros(301, np.uint64)
        tbl.synth_reads_device(300, 90, 20, 77, b.ctypes.data, o.ctypes.data)
        got.append(b[:300 * 90].copy())
        assert np.array_equal(o, np.arange(301, dtype=np.uint64) * 90)
        tbl.close()
    assert all(np.array_equal(got[0], x) for x in got[1:])
    print("ok sampler: identical reads from layouts 1, 2, 3, 4")

    # 9. HBM budget (COLBWT_HBM_BUDGET_MB): AUTO falls back to the deepest layout that fits; an
    #    explicit layout that does not fit is COLBWT_ERR_NOMEM; nothing is left allocated
    img = pkg.synth_index(1_000, mean_len=8, split_permille=0, seed=5)
    reads = helpers.backward_walk_reads(img, 30, 60, 0.02, seed=5)
    bases, off = helpers.concat_reads(reads)
    epml, ecid = oracle.OracleIndex(bytes(img)).query_batch(bases, off)
    full = {}
    for layout in (1, 2, 3, 4, 5, 6):
        tbl = pkg.ColPml.from_bytes(img, layout=layout)
        full[layout] = tbl.info().device_bytes
        tbl.close()
    assert full[1] < full[2] < full[3] < full[4] < full[5] < full[6]
    del os.environ["COLBWT_LAYOUT"]                   # the engine's own choice from here on
    # no budget: deep mismatch entries, there is room for them (fat_build.hip kDeepReserve); any budget
    # below the reserve: plain ones, then the ladder (capi.hip): line rows at K = 8 / 6 / 4, three-, two-, one-step rows
    tbl = pkg.ColPml.from_bytes(img, layout=0)
    assert tbl.info().layout == MIS_DEEP, tbl.info().layout
    pml, cid, _ = tbl.query_batch(bases, off)
    assert np.array_equal(pml, epml) and np.array_equal(cid, ecid)
    tbl.close()
    for budget_mb, expect in ((10_000, (5,)), (full[5] / 2**20 - 0.01, (4,)), (full[4] / 2**20 - 0.01, (4, 3)),
                              (full[3] / 2**20 - 0.01, (2,)), (full[2] / 2**20 - 0.01, (1,))):
        os.environ["COLBWT_HBM_BUDGET_MB"] = str(budget_mb)
        tbl = pkg.ColPml.from_bytes(img, layout=0)
        info = tbl.info()
        assert info.layout in expect and info.device_bytes <= budget_mb * 2**20, (budget_mb, info.layout, full)
        if budget_mb < full[4] / 2**20 and info.layout == 4:
            assert info.layout_shape >> 8 < 8              # not the depth that was just ruled out
        pml, cid, _ = tbl.query_batch(bases, off)
        assert np.array_equal(pml, epml) and np.array_equal(cid, ecid)
        tbl.close()
    os.environ["COLBWT_HBM_BUDGET_MB"] = str(full[2] / 2**20)
    try:
        pkg.ColPml.from_bytes(img, layout=3)
        raise SystemExit("a three-step open beyond the budget must fail")
    except pkg.ColbwtError as e:
        assert e.code == -6, e
    os.environ["COLBWT_HBM_BUDGET_MB"] = "0.01"
    try:
        pkg.ColPml.from_bytes(img, layout=0)
        raise SystemExit("nothing fits in 10 KB")
    except pkg.ColbwtError as e:
        assert e.code == -6, e
    del os.environ["COLBWT_HBM_BUDGET_MB"]
    print(f"ok HBM budget fallback (index bytes by layout: {full})")
    print("EMU-ALL-OK")


if __name__ == "__main__":
    main()
