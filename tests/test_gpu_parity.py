"""GPU tier (-m gpu): the HIP path, called through the C-ABI (libcolbwt.so via
ctypes), against the oracle and the committed golden vector -- bit-exact
(integer / byte / index work, no tolerance).

Edge cases follow what the reference's semantics make observable (the
reference has no tests of its own): empty and ragged
reads, bytes absent from the BWT, no case folding, the terminator, sub-run
splits, unbounded succ/pred scans, run lengths beyond the 16-bit offset field,
reads longer than 65535 bases (wide PML), corrupt files.
"""
import os
import shutil

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

# HBM layouts every parity case runs on (include/colbwt.h): 1 = one-step, 2 / 3 = K-step rows,
# 4 = line rows with the default number of look-ahead steps and with two more (steps << 8),
# 5 / 6 = line rows with mismatch lines / deep mismatch lines, likewise
LAYOUTS = (1, 2, 3, 4, 4 | (5 << 8), 4 | (4 << 8), 5, 5 | (4 << 8), 6, 6 | (6 << 8), 6 | (4 << 8))


def _loaded_hip_lib(pkg):
    path = pkg.LIB_PATH
    assert path.endswith("col-bwt_amd/libcolbwt.so") and os.path.exists(path), "HIP extension missing"
    return pkg.lib()


def _check(pkg, oracle, image, reads, wide=False):
    image = bytes(image)
    bases, off = helpers.concat_reads(reads)
    ref = oracle.OracleIndex(image)
    epml, ecid = ref.query_batch(bases, off, wide=wide, threads=8)
    for layout in LAYOUTS:  # one-step rows, K-step refined rows and line rows must all be bit-exact
        tbl = pkg.ColPml.from_bytes(image, layout=layout)
        assert tbl.info().layout == layout & 0xFF
        pml, cid, st = tbl.query_batch(bases, off, wide=wide)
        assert np.array_equal(pml, epml), f"layout {layout}: PML differs at {np.flatnonzero(pml != epml)[:8]}"
        assert np.array_equal(cid, ecid), f"layout {layout}: col-id differs at {np.flatnonzero(cid != ecid)[:8]}"
        tbl.close()
    return st


def _rand_reads(rng, n, lo, hi, alphabet=b"ACGT"):
    return [rng.choice(np.frombuffer(alphabet, np.uint8), size=int(m)) for m in rng.integers(lo, hi + 1, size=n)]


def test_golden_kat_values_and_text(pkg, oracle, golden_dir, tmp_path):
    _loaded_hip_lib(pkg)
    tbl = pkg.ColPml.load(os.path.join(golden_dir, "kat_d"))      # prefix form, pml_query.cpp:110-111
    info = tbl.info()
    assert (info.bwt_r, info.n, info.r, info.sigma) == (13, 22, 15, 5)
    exp = {
        b"GATTACA": ([5, 4, 3, 2, 1, 0, 1], [1, 3, 1, 3, 3, 1, 3]),
        b"TTACCGATNACA": ([4, 3, 2, 1, 0, 3, 2, 1, 0, 1, 0, 1], [1, 0, 3, 0, 2, 1, 3, 1, 3, 3, 1, 3]),
        b"CCCC": ([0, 0, 1, 0], [3, 3, 0, 3]),
    }
    for read, (pml, cid) in exp.items():
        p, c = tbl.query_pml(read)
        assert p.tolist() == pml and c.tolist() == cid
    fa = tmp_path / "kat_d.fa"
    shutil.copy(os.path.join(golden_dir, "kat_d.fa"), fa)
    tbl.query_file(str(fa))
    for ext in (".pml", ".cid"):
        assert open(str(fa) + ext, "rb").read() == open(os.path.join(golden_dir, "kat_d.fa" + ext), "rb").read()


def test_true_bwt_index_ragged_reads(pkg, oracle):
    rng = np.random.default_rng(21)
    base = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=1500)
    seqs = []
    for _ in range(4):
        s = base.copy()
        mut = rng.random(1500) < 0.02
        s[mut] = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(mut.sum()))
        seqs.append(bytes(s))
    image, text = helpers.true_bwt_index(seqs, seed=22, extra_splits=120)
    reads = helpers.reads_from_text(text, 400, (1, 300), 0.03, seed=23, extra=b"Nacgt")
    reads += [np.zeros(0, np.uint8), np.frombuffer(b"T", np.uint8), np.zeros(0, np.uint8)]
    reads += _rand_reads(rng, 100, 1, 64)
    _check(pkg, oracle, image, reads)


def test_c1_four_related_megabase_sequences(pkg, oracle):
    """BASELINE.json configs[0] (C1): 4 x 1 Mbp sequences (one random, three copies with 1 %
    SNPs), their true BWT with min-LCP thresholds and sub-run splits, 10 k x 100 bp reads with
    1 % substitutions -- a real run-length / threshold structure at scale, all HBM layouts."""
    rng = np.random.default_rng(1)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    base = rng.choice(acgt, size=1_000_000)
    seqs = [bytes(base)]
    for k in range(3):
        s = base.copy()
        mut = rng.random(len(s)) < 0.01
        s[mut] = rng.choice(acgt, size=int(mut.sum()))
        seqs.append(bytes(s))
    image, text = helpers.true_bwt_index_large(seqs, seed=2, extra_splits=20_000)
    t = helpers.unpack_col_pml(image)
    assert t["n"] == 4_000_001 and t["r"] > 100_000
    reads = helpers.reads_from_text(text, 10_000, 100, 0.01, seed=5)
    st = _check(pkg, oracle, image, reads)
    assert st.n_bases == 1_000_000


def test_empty_batch_and_all_empty_reads(pkg):
    tbl = pkg.ColPml.from_bytes(pkg.synth_index(400, 5, 0, 1))
    pml, cid, _ = tbl.query_batch(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert pml.size == 0 and cid.size == 0
    pml, cid, _ = tbl.query_batch(np.zeros(0, np.uint8), np.zeros(5, np.uint64))
    assert pml.size == 0 and cid.size == 0


@pytest.mark.parametrize("rows,split,seed", [(256, 0, 4), (257, 300, 3), (700, 0, 1), (50_000, 150, 2)])
def test_synthetic_tables(pkg, oracle, rows, split, seed):
    rng = np.random.default_rng(seed)
    image = pkg.synth_index(rows, mean_len=6, split_permille=split, seed=seed)
    reads = helpers.backward_walk_reads(image.tobytes(), 1000, 150, 0.01, seed=seed)   # match-heavy
    reads += _rand_reads(rng, 600, 0, 200)                                            # mismatch-heavy, ragged
    reads += _rand_reads(rng, 400, 1, 120, alphabet=b"ACGTN\x01acgt")                 # absent bytes, terminator
    _check(pkg, oracle, image, reads)


def test_thresholds_between_runs_are_cut_out_of_refined_rows(pkg, oracle):
    """Realistic thresholds (inside the rows between two runs of a character): the K-step
    build cuts rows there; results must not change, the refined tables get the extra rows."""
    rng = np.random.default_rng(8)
    image = pkg.synth_index(60_000, mean_len=6, split_permille=50, seed=8, thr_mode=1)
    uniform = pkg.synth_index(60_000, mean_len=6, split_permille=50, seed=8, thr_mode=0)
    reads = helpers.backward_walk_reads(image.tobytes(), 1500, 150, 0.03, seed=8) + _rand_reads(rng, 500, 0, 200)
    _check(pkg, oracle, image, reads)
    a = pkg.ColPml.from_bytes(image.tobytes(), layout=3)
    b = pkg.ColPml.from_bytes(uniform.tobytes(), layout=3)
    assert a.info().table_rows > b.info().table_rows
    a.close(), b.close()


def test_rare_character_uses_jump_tables(pkg, oracle):
    rng = np.random.default_rng(31)
    r = 20_000
    chars = np.tile(np.frombuffer(b"AC", np.uint8), r // 2)
    chars[[3, 15_000, 19_999]] = ord("G")
    chars[9_000] = 1
    lens = rng.integers(1, 9, size=r)
    idx = np.concatenate(([0], np.cumsum(lens)[:-1]))
    n = int(lens.sum())
    interval, offset = helpers.lf_columns(chars, idx, n)
    image = helpers.pack_col_pml(r, n, chars, idx, interval, offset, rng.integers(0, 256, size=r),
                                 rng.integers(0, n, size=r))
    _check(pkg, oracle, image, _rand_reads(rng, 800, 1, 100, alphabet=b"ACGGG\x01T"))


@pytest.mark.parametrize("alphabet", [b"ACGT", b"\x01ACGT", b"\x01ACGNTac"])
def test_threshold_hints_on_and_off(pkg, oracle, alphabet):
    """Thresholds that fall inside rows (offset-dependent `pos < thr`,
    col_bwt.hpp:560) with the 2-bit hints active (sigma <= 5) and inactive."""
    rng = np.random.default_rng(len(alphabet))
    image = helpers.random_table(rng, 30_000, alphabet=alphabet)
    _check(pkg, oracle, image, _rand_reads(rng, 1500, 1, 120, alphabet=alphabet + b"N"))


def test_long_runs_len16_escape(pkg, oracle):
    rng = np.random.default_rng(41)
    r = 4000
    chars = np.tile(np.frombuffer(b"ACGT", np.uint8), r // 4)
    lens = rng.integers(1, 50, size=r)
    lens[[5, 77, 300, 2222, r - 1]] = [65535, 70000, 200000, 65534, 66000]
    idx = np.concatenate(([0], np.cumsum(lens)[:-1]))
    n = int(lens.sum())
    interval, offset = helpers.lf_columns(chars, idx, n)
    offset = offset & np.uint64(0xFFFF)            # the 16-bit field silently truncates (LF_table.hpp:39,376)
    image = helpers.pack_col_pml(r, n, chars, idx, interval, offset, rng.integers(0, 256, size=r),
                                 rng.integers(0, n, size=r))
    _check(pkg, oracle, image, _rand_reads(rng, 1000, 1, 150))


def test_wide_pml_for_reads_over_65535(pkg, oracle):
    image = pkg.synth_index(3000, mean_len=4, split_permille=0, seed=9)
    long_reads = helpers.backward_walk_reads(image.tobytes(), 2, 70_000, 0.0002, seed=10)
    rng = np.random.default_rng(5)
    st = _check(pkg, oracle, image, long_reads + _rand_reads(rng, 5, 1, 30), wide=True)
    assert st.n_bases > 140_000
    bases, off = helpers.concat_reads(long_reads)
    with pytest.raises(pkg.ColbwtError):
        pkg.ColPml.from_bytes(image).query_batch(bases, off, wide=False)


def test_loader_rejects_corrupt_images(pkg):
    good = pkg.synth_index(300, mean_len=5, seed=12).tobytes()
    bad = [good[:-1],
           good[:24] + (299).to_bytes(8, "little") + good[32:],
           good[:32 + 18 * 10 + 6] + (10**6).to_bytes(4, "little") + good[32 + 18 * 10 + 10:],
           good[:32 + 18 * 20 + 1] + (0).to_bytes(5, "little") + good[32 + 18 * 20 + 6:]]
    for b in bad:
        with pytest.raises(pkg.ColbwtError) as ei:
            pkg.ColPml.from_bytes(b)
        assert ei.value.code == -3
    with pytest.raises(pkg.ColbwtError) as ei:
        pkg.ColPml.load("/nonexistent/prefix")
    assert ei.value.code == -2


def test_fasta_fastq_gz_text_outputs_match_oracle(pkg, oracle, tmp_path):
    """pml_query end to end (FASTA multi-line, FASTQ, gzip) vs the oracle's
    restatement of the same program: byte-identical .pml/.cid."""
    import gzip
    rng = np.random.default_rng(51)
    image = pkg.synth_index(5000, mean_len=6, split_permille=100, seed=52)
    reads = helpers.backward_walk_reads(image.tobytes(), 200, 151, 0.02, seed=53) + _rand_reads(rng, 50, 0, 90)
    names = [f"read{k}/1" for k in range(len(reads))]
    fa = tmp_path / "reads.fa"
    helpers.write_fasta(fa, reads, [nm + " extra comment" for nm in names], width=60)
    fq = tmp_path / "reads.fq.gz"
    with gzip.open(fq, "wb") as f:
        for nm, rd in zip(names, reads):
            f.write(b"@" + nm.encode() + b"\n" + bytes(rd) + b"\n+\n" + b"I" * len(rd) + b"\n")
    tbl = pkg.ColPml.from_bytes(image)
    ref = oracle.OracleIndex(image.tobytes())
    for path in (fa, fq):
        tbl.query_file(str(path), batch_bases=7000)            # several GPU batches
        ref.pml_query_files(str(path), str(path) + ".opml", str(path) + ".ocid")
        assert open(str(path) + ".pml", "rb").read() == open(str(path) + ".opml", "rb").read()
        assert open(str(path) + ".cid", "rb").read() == open(str(path) + ".ocid", "rb").read()


def test_device_resident_entry_point_and_read_sampler(pkg, oracle):
    """colbwt_query_device on torch-owned HBM buffers (the bench / multi-GPU
    path) incl. chunked launches with absolute offsets; the device read sampler
    feeds it."""
    import torch
    dev = torch.device("cuda", 0)
    image = pkg.synth_index(2_000_000, mean_len=8, split_permille=0, seed=42)
    tbl = pkg.ColPml.from_bytes(image)
    n_reads, m = 60_000, 150
    d_bases = torch.zeros(n_reads * m + 128, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    tbl.synth_reads_device(n_reads, m, 10, 43, d_bases.data_ptr(), d_off.data_ptr(), s)
    d_pml = torch.zeros(n_reads * m + 16, dtype=torch.int16, device=dev)
    d_cid = torch.zeros(n_reads * m + 16, dtype=torch.uint8, device=dev)
    for lo, hi in ((0, 25_000), (25_000, 60_000)):
        st = tbl.query_device(d_bases.data_ptr(), d_off.data_ptr() + 8 * lo, hi - lo, (hi - lo) * m,
                              d_pml.data_ptr(), d_cid.data_ptr(), 2, s, timed=True)
        assert st.kernel_ms > 0
    torch.cuda.synchronize()
    off = d_off.cpu().numpy().astype(np.uint64)
    assert off[-1] == n_reads * m and np.all(np.diff(off) == m)
    bases = d_bases[:n_reads * m].cpu().numpy()
    assert set(np.unique(bases).tolist()) <= set(b"ACGT")
    ref = oracle.OracleIndex(image.tobytes())
    epml, ecid = ref.query_batch(bases, off, threads=16)
    assert np.array_equal(d_pml[:n_reads * m].cpu().numpy().view(np.uint16), epml)
    assert np.array_equal(d_cid[:n_reads * m].cpu().numpy(), ecid)
    resets = float((epml == 0).mean())
    assert 0.02 < resets < 0.6          # the recipe's mix of extends and resets


@pytest.mark.parametrize("layout", [1, 2, 3, 4, 5, 6])
def test_full_scale_properties(pkg, oracle, layout, c2_image):
    """BASELINE config C2 scale (2e8 rows): size-independent properties --
    idempotence (two runs, identical bytes), batch-position independence (a
    permuted sub-batch gives the same per-read values) and oracle agreement on
    a sample.  Rows can be lowered with COLBWT_TEST_ROWS for rehearsals."""
    import torch
    n_reads, m = int(os.environ.get("COLBWT_TEST_READS", "2000000")), 150
    dev = torch.device("cuda", 0)
    image = c2_image
    tbl = pkg.ColPml.from_bytes(image, layout=layout)
    d_bases = torch.zeros(n_reads * m + 128, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    tbl.synth_reads_device(n_reads, m, 10, 43, d_bases.data_ptr(), d_off.data_ptr(), s)

    def run():
        p = torch.zeros(n_reads * m + 16, dtype=torch.int16, device=dev)
        c = torch.zeros(n_reads * m + 16, dtype=torch.uint8, device=dev)
        tbl.query_device(d_bases.data_ptr(), d_off.data_ptr(), n_reads, n_reads * m, p.data_ptr(), c.data_ptr(), 2, s)
        torch.cuda.synchronize()
        return p, c
    p1, c1 = run()
    p2, c2 = run()
    assert torch.equal(p1, p2) and torch.equal(c1, c2)                       # idempotent / deterministic
    # permuted sub-batch through the host entry point
    rng = np.random.default_rng(1)
    pick = rng.choice(n_reads, size=3000, replace=False)
    hb = d_bases[:n_reads * m].view(n_reads, m)[torch.from_numpy(pick).to(dev)].cpu().numpy()
    off = (np.arange(3001, dtype=np.uint64) * np.uint64(m))
    sp, sc, _ = tbl.query_batch(hb.reshape(-1), off)
    full_p = p1[:n_reads * m].view(n_reads, m)[torch.from_numpy(pick).to(dev)].cpu().numpy().view(np.uint16)
    full_c = c1[:n_reads * m].view(n_reads, m)[torch.from_numpy(pick).to(dev)].cpu().numpy()
    assert np.array_equal(sp.reshape(3000, m), full_p) and np.array_equal(sc.reshape(3000, m), full_c)
    # oracle on the same sample
    ref = oracle.OracleIndex(image)
    ep, ec = ref.query_batch(hb.reshape(-1), off, threads=16)
    assert np.array_equal(sp, ep) and np.array_equal(sc, ec)


def test_cli_pml_query_and_col_bwt_launcher(golden_dir, tmp_path):
    """The drop-in command lines: `pml_query -p <fa> <prefix>` (pml_query.cpp:92-143,
    same getopt string) and `col-bwt query -p PATTERN index` (col-bwt.py:226-228)
    reproduce the golden .pml/.cid bytes; a missing index is a non-zero exit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "col-bwt_amd", "pml_query")
    launcher = os.path.join(root, "col-bwt_amd", "col-bwt")
    for k, cmd in enumerate(([exe, "-v", "-p"], [sys.executable, launcher, "query", "-p"])):
        d = tmp_path / f"run{k}"
        d.mkdir()
        fa = d / "kat_d.fa"
        shutil.copy(os.path.join(golden_dir, "kat_d.fa"), fa)
        out = subprocess.run(cmd + [str(fa), os.path.join(golden_dir, "kat_d")], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        for ext in (".pml", ".cid"):
            assert open(str(fa) + ext, "rb").read() == open(os.path.join(golden_dir, "kat_d.fa" + ext), "rb").read()
    # the launcher also leaves the artefacts the reference's `query` leaves (col-bwt.py:194,198):
    # PATTERN.split.pml.bin / .split.cid.bin, here links to the containers; `view` turns them back
    # into the golden text
    for kind, width in (("pml", 2), ("cid", 1)):
        split = str(fa) + f".split.{kind}.bin"
        assert os.path.islink(split) and os.path.exists(split) and os.path.exists(str(fa) + f".{kind}.bin")
        view = subprocess.run([sys.executable, launcher, "view", split, "-o", str(d / f"view.{kind}")], capture_output=True, text=True)
        assert view.returncode == 0, view.stderr
        assert open(d / f"view.{kind}", "rb").read() == open(os.path.join(golden_dir, f"kat_d.fa.{kind}"), "rb").read()
    for flag, present, absent in (("-b", (".pml.bin", ".split.pml.bin", ".split.cid.bin"), (".pml", ".cid")),
                                  ("-t", (".pml", ".cid"), (".pml.bin", ".split.pml.bin"))):
        d = tmp_path / f"run{flag}"
        d.mkdir()
        fa = d / "kat_d.fa"
        shutil.copy(os.path.join(golden_dir, "kat_d.fa"), fa)
        out = subprocess.run([sys.executable, launcher, "query", flag, "-p", str(fa), os.path.join(golden_dir, "kat_d")],
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        assert all(os.path.exists(str(fa) + e) for e in present) and not any(os.path.lexists(str(fa) + e) for e in absent)
    bad = subprocess.run([exe, "-p", str(fa), "/nonexistent/prefix"], capture_output=True, text=True)
    assert bad.returncode != 0 and "[ERROR]" in bad.stderr
    bad = subprocess.run([sys.executable, launcher, "query", "-p", str(fa), "/nonexistent/prefix"], capture_output=True, text=True)
    assert bad.returncode != 0 and not os.path.lexists(str(fa) + ".split.pml.bin")


def test_large_file_parallel_text_formatting(pkg, oracle, tmp_path):
    """A FASTA big enough for the multi-threaded formatter (> 1 MiB of values)
    and for several GPU batches: still byte-identical to the sequential
    reference format (pml_query.cpp:78-85)."""
    rng = np.random.default_rng(61)
    image = pkg.synth_index(200_000, mean_len=8, split_permille=50, seed=62)
    reads = helpers.backward_walk_reads(image.tobytes(), 12_000, 150, 0.01, seed=63) + _rand_reads(rng, 3_000, 0, 400)
    fa = tmp_path / "big.fa"
    helpers.write_fasta(fa, reads, width=70)
    tbl = pkg.ColPml.from_bytes(image)
    st = tbl.query_file(str(fa), batch_bases=900_000)
    assert st.n_reads == len(reads)
    oracle.OracleIndex(image.tobytes()).pml_query_files(str(fa), str(fa) + ".opml", str(fa) + ".ocid")
    assert open(str(fa) + ".pml", "rb").read() == open(str(fa) + ".opml", "rb").read()
    assert open(str(fa) + ".cid", "rb").read() == open(str(fa) + ".ocid", "rb").read()


def test_concurrent_host_threads_share_one_index(pkg, oracle):
    """The C-ABI promises thread-safety on distinct batches (include/colbwt.h):
    four host threads query one index at once; every result equals the oracle's."""
    import threading
    image = pkg.synth_index(300_000, mean_len=8, split_permille=0, seed=71)
    tbl = pkg.ColPml.from_bytes(image)
    ref = oracle.OracleIndex(image.tobytes())
    jobs = []
    for t in range(4):
        reads = helpers.backward_walk_reads(image.tobytes(), 3000, 100 + 10 * t, 0.02, seed=80 + t)
        jobs.append(helpers.concat_reads(reads))
    results = [None] * 4

    def run(t):
        for _ in range(3):
            results[t] = tbl.query_batch(*jobs[t])[:2]

    ths = [threading.Thread(target=run, args=(t,)) for t in range(4)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    for t in range(4):
        ep, ec = ref.query_batch(*jobs[t], threads=8)
        assert np.array_equal(results[t][0], ep) and np.array_equal(results[t][1], ec)


@pytest.mark.parametrize("layout", [4, 6])
def test_many_overlapping_launches_on_one_index(pkg, oracle, layout):
    """include/colbwt.h: any number of colbwt_query_device launches may be in flight on one index.
    48 asynchronous launches on 24 streams (two per stream, nothing waited for in between) against
    one line-row index (both kinds), each over a batch of its own large enough to run for a while and to claim
    chunks from the workgroups' counters; every result equals the oracle's."""
    import torch
    dev = torch.device("cuda", 0)
    image = pkg.synth_index(2_000_000, mean_len=8, split_permille=0, seed=42)
    tbl = pkg.ColPml.from_bytes(image, layout=layout)
    assert tbl.info().layout == layout
    n_launch, n_streams, n_reads, m = 48, 24, 120_000, 150
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    d_bases = torch.zeros((n_launch, n_reads * m + 128), dtype=torch.uint8, device=dev)
    d_off = torch.zeros((n_launch, n_reads + 1), dtype=torch.int64, device=dev)
    d_pml = torch.full((n_launch, n_reads * m + 64), -1, dtype=torch.int16, device=dev)
    d_cid = torch.full((n_launch, n_reads * m + 64), 0xEE, dtype=torch.uint8, device=dev)
    s0 = torch.cuda.current_stream().cuda_stream
    for q in range(n_launch):
        tbl.synth_reads_device(n_reads, m, 10, 900 + q, d_bases[q].data_ptr(), d_off[q].data_ptr(), s0)
    torch.cuda.synchronize()
    for q in range(n_launch):                      # all queued back to back: launches overlap on the device
        tbl.query_device(d_bases[q].data_ptr(), d_off[q].data_ptr(), n_reads, n_reads * m,
                         d_pml[q].data_ptr(), d_cid[q].data_ptr(), 2, streams[q % n_streams].cuda_stream)
    torch.cuda.synchronize()
    ref = oracle.OracleIndex(image.tobytes())
    off = d_off[0].cpu().numpy().astype(np.uint64)
    for q in range(n_launch):
        bases = d_bases[q, :n_reads * m].cpu().numpy()
        epml, ecid = ref.query_batch(bases, off, threads=16)
        assert np.array_equal(d_pml[q, :n_reads * m].cpu().numpy().view(np.uint16), epml), f"launch {q}: PML"
        assert np.array_equal(d_cid[q, :n_reads * m].cpu().numpy(), ecid), f"launch {q}: col ids"
        assert (d_pml[q, n_reads * m:] == -1).all() and (d_cid[q, n_reads * m:] == 0xEE).all()
    tbl.close()


def test_open_close_does_not_leak_hbm(pkg):
    """Index::release frees every table of every layout (incl. the temporary
    level the three-step build refines from)."""
    import torch
    image = pkg.synth_index(1_000_000, mean_len=8, split_permille=0, seed=5)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    for layout in (1, 2, 3, 4, 5, 6, 0) * 2:
        tbl = pkg.ColPml.from_bytes(image, layout=layout)
        assert tbl.info().device_bytes > 0
        tbl.close()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(0)[0]
    assert free0 - free1 < 64 << 20, f"HBM leak: {free0 - free1} bytes"


@pytest.mark.parametrize("seed0", [0, 100])
def test_fuzz_random_tables_all_layouts(pkg, oracle, seed0):
    """Randomised sweep: alphabets of 2..9 arbitrary bytes, run lengths 1..300 with a few rows
    beyond the 16-bit length field, sub-run splits, arbitrary thresholds; random + walk reads."""
    for seed in range(seed0, seed0 + 25):
        rng = np.random.default_rng(seed)
        sigma = int(rng.integers(2, 10))
        alpha = bytes(rng.choice(np.arange(1, 128), size=sigma, replace=False).astype(np.uint8).tolist())
        r = int(rng.integers(3, 40_000))
        img = helpers.random_table(rng, r, alphabet=alpha, max_len=int(rng.choice([1, 2, 9, 40, 300])),
                                   split_prob=float(rng.choice([0, 0.1, 0.5])))
        if seed % 4 == 0:
            t = helpers.unpack_col_pml(img)
            lens = np.diff(np.append(t["idx"].astype(np.int64), t["n"]))
            lens[rng.integers(0, r, size=3)] = rng.integers(65530, 200000, size=3)
            idx = np.concatenate(([0], np.cumsum(lens)[:-1]))
            n = int(lens.sum())
            itv, off = helpers.lf_columns(t["char"], idx, n)
            img = helpers.pack_col_pml(int(t["bwt_r"]), n, t["char"], idx, itv, off & np.uint64(0xFFFF), t["cid"],
                                       rng.integers(0, n, size=r))
        img = bytes(img)
        reads = _rand_reads(rng, 300, 0, 200, alphabet=alpha + b"\xfe")
        reads += helpers.backward_walk_reads(img, 300, int(rng.integers(1, 400)), 0.02, seed=seed)
        _check(pkg, oracle, img, reads)


def test_gather_codec_round_trip_and_pipeline(pkg, oracle):
    """The multi-GPU exchange step on one GPU: PML values -> one bit per base -> values, through
    the same GatherPipeline / PmlCodec wiring bench.py uses, with the collective replaced by a
    device copy (every 'rank' contributes this GPU's results)."""
    import torch
    import __graft_entry__  # noqa: F401  (registers colbwt_amd)
    from colbwt_amd import multi_gpu
    dev = torch.device("cuda", 0)
    image = pkg.synth_index(100_000, mean_len=8, seed=3)
    tbl = pkg.ColPml.from_bytes(image.tobytes())
    n_reads, m, world = 20_011, 150, 3            # n_reads * m is not a multiple of 32
    nb = n_reads * m
    d_bases = torch.zeros(nb + 128, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    tbl.synth_reads_device(n_reads, m, 20, 7, d_bases.data_ptr(), d_off.data_ptr())
    d_pml = torch.zeros(nb + 16, dtype=torch.int16, device=dev)
    d_cid = torch.zeros(nb + 16, dtype=torch.uint8, device=dev)
    stream, comm = torch.cuda.current_stream(), torch.cuda.Stream(device=dev)
    words = (nb + 31) // 32
    d_zero = torch.zeros(words, dtype=torch.int32, device=dev)
    d_end = torch.zeros(words, dtype=torch.int32, device=dev)
    g_zero = torch.zeros((world, 4 * words), dtype=torch.uint8, device=dev)
    g_pml = torch.full((world, words * 32), -1, dtype=torch.int16, device=dev)
    pkg.read_end_mask_device(d_off.data_ptr(), n_reads, d_end.data_ptr(), stream.cuda_stream)

    nccl_stream = torch.cuda.Stream(device=dev)

    class Work:                                    # like ProcessGroupNCCL's: wait() orders the CURRENT stream
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)
            return True

    class FakeDist:
        """rank 0 of `world`: every rank sends what this GPU holds.  The transfer runs on its own
        stream, behind whatever the calling stream had queued when gather() was called -- the
        stream semantics of an RCCL collective with async_op=True."""
        @staticmethod
        def gather(src, glist, dst=0, async_op=False):
            nccl_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(nccl_stream):
                for g in glist:
                    g.copy_(src, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(nccl_stream)
            return Work(ev)

    def query_chunk(lo, hi):
        tbl.query_device(d_bases.data_ptr(), d_off.data_ptr() + 8 * lo, hi - lo, (hi - lo) * m,
                         d_pml.data_ptr(), d_cid.data_ptr(), 2, stream.cuda_stream)

    codec = multi_gpu.PmlCodec(
        d_zero.view(torch.uint8),
        lambda lo_base, n: pkg.pml_pack_device(d_pml.data_ptr() + 2 * lo_base, n, d_zero.data_ptr() + 4 * (lo_base // 32),
                                               stream.cuda_stream),
        lambda r, w0, nw: pkg.pml_unpack_device(g_zero[r].data_ptr(), d_end.data_ptr(), w0, nw, words,
                                                g_pml[r].data_ptr(), comm.cuda_stream),
        g_zero)
    # the col ids as codes of the table's dictionary (7 ids in the synthetic recipe: 3 bit planes)
    ids = tbl.cid_dictionary()
    assert ids.tolist() == sorted(set(helpers.unpack_col_pml(image.tobytes())["cid"].tolist()))
    bits = pkg.cid_code_bits(len(ids))
    assert bits == 3
    d_planes = torch.zeros(bits * words, dtype=torch.int32, device=dev)
    g_planes = torch.zeros((world, 4 * bits * words), dtype=torch.uint8, device=dev)
    g_cid = torch.full((world, words * 32), 0xEE, dtype=torch.uint8, device=dev)
    cid_codec = multi_gpu.PmlCodec(
        d_planes.view(torch.uint8),
        lambda lo_base, n: pkg.cid_pack_device(d_cid.data_ptr() + lo_base, n, ids, d_planes.data_ptr() + 4 * bits * (lo_base // 32),
                                               stream.cuda_stream),
        lambda r, w0, nw: pkg.cid_unpack_device(g_planes[r].data_ptr(), w0, nw, ids, g_cid[r].data_ptr(), comm.cuda_stream),
        g_planes, bits=bits)
    pipe = multi_gpu.GatherPipeline(FakeDist, 0, world, n_reads, m, 4, [(d_cid[:nb], 1)], dev, (stream, comm),
                                    codecs=[codec, cid_codec])
    assert pipe.bounds[0] == 0 and pipe.bounds[-1] == n_reads and all(b % 32 == 0 for b in pipe.bounds[:-1])
    for _ in range(2):
        pipe.step(query_chunk)
    pipe.finish()
    torch.cuda.synchronize()
    hb = d_bases[:nb].cpu().numpy()
    ho = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(m)
    ep, ec = oracle.OracleIndex(image.tobytes()).query_batch(hb, ho, threads=8)
    assert np.array_equal(d_pml[:nb].cpu().numpy().view(np.uint16), ep)
    for r in range(world):
        assert np.array_equal(g_pml[r][:nb].cpu().numpy().view(np.uint16), ep), r
        assert np.array_equal(pipe.gathered[0][r].cpu().numpy(), ec), r
        assert np.array_equal(g_cid[r][:nb].cpu().numpy(), ec), r
    # a dictionary of 200 ids (8 bit planes) round-trips as well
    many = np.arange(3, 203, dtype=np.uint8)
    src = torch.from_numpy(many[np.random.default_rng(5).integers(0, 200, size=words * 32)]).to(dev)
    pl8 = torch.zeros(8 * words, dtype=torch.int32, device=dev)
    back = torch.zeros(words * 32, dtype=torch.uint8, device=dev)
    assert pkg.cid_code_bits(200) == 8 and pkg.cid_code_bits(1) == 1 and pkg.cid_code_bits(2) == 1 and pkg.cid_code_bits(17) == 5
    pkg.cid_pack_device(src.data_ptr(), words * 32 - 5, many, pl8.data_ptr())
    pkg.cid_unpack_device(pl8.data_ptr(), 0, words, many, back.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(back[:words * 32 - 5], src[:words * 32 - 5])
    tbl.close()


def test_host_entry_large_batch_uses_staged_copy(pkg):
    """colbwt_query_batch with > 256 MB of results in pageable memory: the results come back
    through the pinned staging buffers (several 128 MB chunks, two buffers alternating) and
    must equal what the device entry point left in HBM."""
    import torch
    dev = torch.device("cuda", 0)
    image = pkg.synth_index(2_000_000, mean_len=8, seed=12)
    tbl = pkg.ColPml.from_bytes(image.tobytes())
    n_reads, m = 2_200_003, 150
    nb = n_reads * m
    d_bases = torch.zeros(nb + 128, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    tbl.synth_reads_device(n_reads, m, 15, 5, d_bases.data_ptr(), d_off.data_ptr())
    d_pml = torch.zeros(nb + 16, dtype=torch.int16, device=dev)
    d_cid = torch.zeros(nb + 16, dtype=torch.uint8, device=dev)
    tbl.query_device(d_bases.data_ptr(), d_off.data_ptr(), n_reads, nb, d_pml.data_ptr(), d_cid.data_ptr())
    torch.cuda.synchronize()
    bases = d_bases[:nb].cpu().numpy()
    off = d_off.cpu().numpy().astype(np.uint64)
    for _ in range(2):                                   # second call reuses the staging buffers
        pml, cid, st = tbl.query_batch(bases, off)
        assert st.n_bases == nb
        assert np.array_equal(pml.view(np.int16), d_pml[:nb].cpu().numpy())
        assert np.array_equal(cid, d_cid[:nb].cpu().numpy())
    tbl.close()


def test_two_replicas_shard_every_batch(pkg, oracle, tmp_path):
    """colbwt_index_open_devices with the device list [0, 0] (SURVEY.md 8(b) `device_mask`): two
    replicas of the table on the one GPU, every host batch cut into two shards by base count and
    queried side by side -- bit-equal to the single-replica result and to the oracle, through
    colbwt_query_batch, colbwt_query_batch_u32, colbwt_query_file (text and binary) and the
    `pml_query -d 0,0` command line.  More than one GPU: unmeasured here."""
    import subprocess
    rng = np.random.default_rng(91)
    image = pkg.synth_index(150_000, mean_len=7, split_permille=60, seed=91)
    reads = helpers.backward_walk_reads(image.tobytes(), 3000, 150, 0.02, seed=92) + _rand_reads(rng, 1500, 0, 300)
    reads = [reads[i] for i in rng.permutation(len(reads))]
    bases, off = helpers.concat_reads(reads)
    ep, ec = oracle.OracleIndex(image.tobytes()).query_batch(bases, off, threads=8)
    one = pkg.ColPml.from_bytes(image)
    two = pkg.ColPml.from_bytes(image, devices=[0, 0])
    assert one.info().n_devices == 1 and two.info().n_devices == 2 and two.info().layout == one.info().layout
    for tbl in (one, two):
        p, c, st = tbl.query_batch(bases, off)
        assert st.n_reads == len(reads) and st.n_bases == off[-1]
        assert np.array_equal(p, ep) and np.array_equal(c, ec)
        p32, c32, _ = tbl.query_batch(bases, off, wide=True)
        assert np.array_equal(p32, ep.astype(np.uint32)) and np.array_equal(c32, ec)
    # degenerate shards: fewer reads than replicas, all bases in one read
    for sub in ([reads[0]], [np.zeros(0, np.uint8), reads[1]], []):
        b1, o1 = helpers.concat_reads(sub)
        a = one.query_batch(b1, o1)
        b = two.query_batch(b1, o1)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    fa = tmp_path / "reads.fa"
    helpers.write_fasta(fa, reads, width=80)
    two.query_file(str(fa), batch_bases=100_000)
    oracle.OracleIndex(image.tobytes()).pml_query_files(str(fa), str(fa) + ".opml", str(fa) + ".ocid")
    for ext in ("pml", "cid"):
        assert open(f"{fa}.{ext}", "rb").read() == open(f"{fa}.o{ext}", "rb").read()
    one.close(), two.close()
    # the command line with a device list (and -l, which the reference only honours for one read:
    # col_bwt.hpp:477-495 -- here it writes the normal files)
    idx_file = tmp_path / "tbl.col_pml"
    idx_file.write_bytes(image.tobytes())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in (["-d", "0,0"], ["-l"]):
        fa2 = tmp_path / ("cli" + "".join(extra).replace(",", "_") + ".fa")
        shutil.copy(fa, fa2)
        out = subprocess.run([os.path.join(root, "col-bwt_amd", "pml_query"), "-v"] + extra + ["-p", str(fa2), str(tmp_path / "tbl")],
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        if extra[0] == "-d":
            assert "2 replicas" in out.stdout
        for ext in ("pml", "cid"):
            assert open(f"{fa2}.{ext}", "rb").read() == open(f"{fa}.o{ext}", "rb").read()


def test_granule_store_small_tables(pkg, oracle, tmp_path):
    """csrc/dev_vmm.h: arrays of 512 MiB and more are address ranges over recycled granules of physical
    memory, which the tables of this tier never reach.  Here the knobs are shrunk (granules of 2 MiB,
    every array of 1 MiB and more through the store) in child processes -- the store reads them once --
    and `pml_query` must write the oracle's bytes for every layout: refinement levels, final tables and
    the run-time tables all live on granules that earlier arrays of the same open gave back."""
    import subprocess
    rng = np.random.default_rng(404)
    image = pkg.synth_index(400_000, mean_len=7, split_permille=80, seed=404)
    reads = helpers.backward_walk_reads(image.tobytes(), 1500, 150, 0.02, seed=405) + _rand_reads(rng, 500, 0, 300)
    fa = tmp_path / "reads.fa"
    helpers.write_fasta(fa, reads, width=80)
    oracle.OracleIndex(image.tobytes()).pml_query_files(str(fa), str(fa) + ".opml", str(fa) + ".ocid")
    (tmp_path / "tbl.col_pml").write_bytes(image.tobytes())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for layout in (6, 5, 4, 3, 2, 1):
        fa2 = tmp_path / f"store_{layout}.fa"
        shutil.copy(fa, fa2)
        env = dict(os.environ, COLBWT_VMM_GRANULE_MB="2", COLBWT_VMM_MIN_MB="1", COLBWT_ALLOC_LOG="1", COLBWT_LAYOUT=str(layout))
        out = subprocess.run([os.path.join(root, "col-bwt_amd", "pml_query"), "-v", "-p", str(fa2), str(tmp_path / "tbl")],
                             capture_output=True, text=True, env=env)
        assert out.returncode == 0, out.stdout + out.stderr
        assert " failed (" not in out.stderr, out.stderr
        for ext in ("pml", "cid"):
            assert open(f"{fa2}.{ext}", "rb").read() == open(f"{fa}.o{ext}", "rb").read(), (layout, ext)


def test_binary_containers_round_trip(pkg, oracle, tmp_path):
    """`.pml.bin` / `.cid.bin` (include/colbwt.h: Movi-like record shape, unverified against Movi --
    the reference names these outputs, scripts/col-bwt.py:194, but their writer is not in its
    tree): values read back from the containers == the text files' values == the oracle's, and
    `col-bwt view` / colbwt_binary_to_text reproduce the reference's text byte for byte.  FASTA
    (parsed by several threads), a FASTA whose tail turns into FASTQ (the parallel reader hands
    over to the sequential one), CRLF line ends, gzip."""
    import gzip
    import subprocess
    import sys
    rng = np.random.default_rng(95)
    image = pkg.synth_index(60_000, mean_len=6, split_permille=80, seed=95)
    reads = helpers.backward_walk_reads(image.tobytes(), 2500, 120, 0.02, seed=96) + _rand_reads(rng, 700, 0, 260)
    names = [f"rd{k}.{k % 7}" for k in range(len(reads))]
    ref = oracle.OracleIndex(image.tobytes())
    tbl = pkg.ColPml.from_bytes(image)
    plain = tmp_path / "plain.fa"
    helpers.write_fasta(plain, reads, [nm + " some comment" for nm in names], width=50)
    crlf = tmp_path / "crlf.fa"
    with open(crlf, "wb") as f:
        for nm, rd in zip(names, reads):
            f.write(b">" + nm.encode() + b"\r\n")
            for s in range(0, len(rd), 61):
                f.write(bytes(rd[s:s + 61]) + b"\r\n")
            f.write(b"\r\n")
    mixed = tmp_path / "mixed.fx"                            # FASTA records, then FASTQ records
    with open(mixed, "wb") as f:
        for nm, rd in list(zip(names, reads))[:1500]:
            f.write(b">" + nm.encode() + b"\n" + bytes(rd) + b"\n")
        for nm, rd in list(zip(names, reads))[1500:]:
            f.write(b"@" + nm.encode() + b"\n" + bytes(rd) + b"\n+\n" + b">" * len(rd) + b"\n")   # qualities of '>'
    gz = tmp_path / "plain.fa.gz"
    with gzip.open(gz, "wb") as f:
        f.write(open(plain, "rb").read())
    for path in (plain, crlf, mixed, gz):
        tbl.query_file(str(path), batch_bases=40_000)                      # many batches
        tbl.query_file_binary(str(path), batch_bases=40_000)
        ref.pml_query_files(str(path), str(path) + ".opml", str(path) + ".ocid")
        for ext, width in (("pml", 2), ("cid", 1)):
            text = open(f"{path}.{ext}", "rb").read()
            assert text == open(f"{path}.o{ext}", "rb").read(), (path, ext)
            pkg.binary_to_text(f"{path}.{ext}.bin", width, f"{path}.{ext}.view")
            assert open(f"{path}.{ext}.view", "rb").read() == text, (path, ext)
            recs = pkg.read_binary(f"{path}.{ext}.bin", width)
            lines = text.split(b"\n")
            assert len(recs) * 2 == len(lines) - 1
            for k in (0, 1, len(recs) // 2, len(recs) - 1):
                assert lines[2 * k] == b">" + recs[k][0].encode() + b" "
                assert recs[k][1].tolist() == [int(x) for x in lines[2 * k + 1].split()]
    # against the oracle's values directly, record by record
    bases, off = helpers.concat_reads(reads)
    ep, ec = ref.query_batch(bases, off, threads=8)
    for (nm, vals), k in zip(pkg.read_binary(f"{plain}.pml.bin", 2), range(len(reads))):
        assert nm == names[k] and np.array_equal(vals, ep[off[k]:off[k + 1]])
    for (nm, vals), k in zip(pkg.read_binary(f"{plain}.cid.bin", 1), range(len(reads))):
        assert np.array_equal(vals, ec[off[k]:off[k + 1]])
    tbl.close()
    # the launcher: `col-bwt query -b` then `col-bwt view`
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    idx_file = tmp_path / "tbl.col_pml"
    idx_file.write_bytes(image.tobytes())
    launcher = os.path.join(root, "col-bwt_amd", "col-bwt")
    fa2 = tmp_path / "launch.fa"
    shutil.copy(plain, fa2)
    out = subprocess.run([sys.executable, launcher, "query", "-b", "-p", str(fa2), str(tmp_path / "tbl")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert open(f"{fa2}.pml.bin", "rb").read() == open(f"{plain}.pml.bin", "rb").read()
    out = subprocess.run([sys.executable, launcher, "view", f"{fa2}.cid.bin"], capture_output=True)
    assert out.returncode == 0 and out.stdout == open(f"{plain}.cid", "rb").read()
