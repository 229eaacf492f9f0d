"""Front of the pipeline (SURVEY.md 8(f) "next" #4): FASTA documents -> RLBWT, thresholds, multi-MUMs.

PARITY UNPINNED (mumemto, which writes these files for the reference, is neither in the reference
tree nor in this image): the tests pin the product to oracle/rlbwt_oracle.py -- the published
meaning of the files written independently -- and the oracle's multi-MUMs to a brute-force
enumeration of substrings that uses no suffix array.

CPU tier: oracle self-checks; the product's HIP sources under the SIMT emulator == oracle.
GPU tier (-m gpu): the device construction == oracle on related sequences (with and without reverse
complements); the whole chain FASTA -> build_rlbwt -> col_split -> build_col_bwt -> query equals the
oracle chain, file by file.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import rlbwt_oracle as ro  # noqa: E402

ACGT = np.frombuffer(b"ACGT", np.uint8)


def related_docs(rng, n_docs, length, rate, records=1, with_n=False):
    """n_docs documents of `records` records each: copies of one random sequence with substitutions,
    a few insertions/deletions, optionally a stretch of N."""
    base = rng.choice(ACGT, size=length)
    docs = []
    for _ in range(n_docs):
        s = base.copy()
        mut = rng.random(length) < rate
        s[mut] = rng.choice(ACGT, size=int(mut.sum()))
        s = s.tolist()
        for _ in range(int(rng.integers(0, 3))):
            at = int(rng.integers(0, len(s)))
            if rng.random() < 0.5:
                del s[at:at + int(rng.integers(1, 5))]
            else:
                s[at:at] = rng.choice(ACGT, size=int(rng.integers(1, 5))).tolist()
        if with_n:
            at = int(rng.integers(0, max(1, len(s) - 10)))
            s[at:at + 6] = [ord("N")] * 6
        s = bytes(s)
        cuts = sorted(rng.choice(np.arange(1, len(s)), size=records - 1, replace=False).tolist()) if records > 1 else []
        docs.append([s[a:b] for a, b in zip([0] + cuts, cuts + [len(s)])])
    return docs


def test_oracle_on_the_survey_text():
    """GATTACAGATTACCGATAACA: the multi-MUM of the two halves' GATTAC is
    at suffix-array rank 15 in the survey's single-terminator text; here the same text as two
    documents."""
    res = ro.build([[b"GATTACA"], [b"GATTACCGATAACA"]], min_len=3)
    text = res["text"]
    assert text == b"GATTACA\x01GATTACCGATAACA\x01\x00"
    assert sorted(res["sa"]) == list(range(len(text))) and all(text[a:] < text[b:] for a, b in zip(res["sa"], res["sa"][1:]))
    assert sum(res["lens"]) == len(text) and all(a != b for a, b in zip(res["heads"], res["heads"][1:]))
    assert res["mums"] == ro.brute_force_mums(text, res["doc_start"], 3)
    (length, rank), = [m for m in res["mums"] if m[0] == 6]
    assert text[res["sa"][rank]:][:6] == b"GATTAC" and text[res["sa"][rank + 1]:][:6] == b"GATTAC"


def test_oracle_mums_equal_the_brute_force_definition():
    rng = np.random.default_rng(3)
    found = 0
    for case in range(12):
        nd = int(rng.integers(2, 5))
        docs = related_docs(rng, nd, int(rng.integers(30, 90)), 0.08, records=int(rng.integers(1, 3)), with_n=case % 3 == 0)
        min_len = int(rng.integers(1, 7))
        for rc in (False, True):
            res = ro.build(docs, min_len=min_len, revcomp=rc)
            assert res["mums"] == ro.brute_force_mums(res["text"], res["doc_start"], min_len), (case, rc)
            found += len(res["mums"])
    assert found > 40
    # and both suffix sorters agree
    text, _ = ro.build_text(related_docs(rng, 3, 3000, 0.02))
    assert ro.suffix_array(text) == sorted(range(len(text)), key=lambda i: text[i:])


def test_oracle_thresholds_equal_the_helpers_definition():
    """tests/helpers._index_from_bwt_lcp (used by the query tests since round 1) takes the same minimum --
    here over the runs of the folded characters and the separator-capped LCPs."""
    rng = np.random.default_rng(4)
    res = ro.build(related_docs(rng, 3, 400, 0.03), min_len=10)
    bwt = np.maximum(np.frombuffer(res["bwt"], np.uint8), 1)
    lcp = np.array(ro.capped_lcp(res["text"], res["sa"]), np.int64)
    n = len(bwt)
    heads = np.flatnonzero(np.concatenate(([True], bwt[1:] != bwt[:-1])))
    ends = np.append(heads[1:], n) - 1
    expect = np.zeros(len(heads), np.int64)
    last = {}
    for j, c in enumerate(bwt[heads]):
        if c in last:
            seg = lcp[ends[last[c]] + 1:heads[j] + 1]
            expect[j] = ends[last[c]] + 1 + int(np.argmin(seg))
        last[c] = j
    assert res["thr"] == expect.tolist()


def _folded_groups(heads):
    """The groups read_thresholds (col_bwt.hpp:446-451) forms: maximal stretches of equal characters
    after bytes <= 1 (and >= 0x80) are folded to the terminator (col_bwt.hpp:167-171)."""
    f = [1 if (c <= 1 or c >= 0x80) else c for c in heads]
    return [k for k in range(len(f)) if k == 0 or f[k] != f[k - 1]]


def test_documents_sharing_a_prefix_get_one_threshold_per_group_of_the_builder(oracle):
    """The suffix-array neighbours of suffix 0 are other record starts, so when documents share a
    prefix the final 0 of the BWT touches a run of separators: written as two runs, the builder (which
    folds both to the terminator and gives the k-th threshold to the k-th GROUP) would hand every later
    run its predecessor's threshold.  Runs are runs of the folded characters: one threshold per group,
    and every row of the built index carries its own run's value."""
    cases = [[[b"ACGTTGCA"], [b"ACGTAGCA"]]]
    rng = np.random.default_rng(12)
    cases.append(related_docs(rng, 3, 400, 0.01))
    for docs in cases:
        for rc in (False, True):
            res = ro.build(docs, min_len=4, revcomp=rc)
            raw_heads = [c for k, c in enumerate(res["bwt"]) if k == 0 or c != res["bwt"][k - 1]]
            assert len(_folded_groups(raw_heads)) < len(raw_heads)         # the 0 does touch a run of 1s here
            heads, lens, thr = res["heads"], res["lens"], res["thr"]
            assert 0 not in heads and len(_folded_groups(heads)) == len(heads) == len(thr)
            starts = np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.uint64)
            image = oracle.build_col_pml(heads, lens, np.zeros(len(heads), np.uint8), starts, thr)
            t = helpers.unpack_col_pml(image)
            assert t["r"] == len(heads) and t["n"] == len(res["text"])
            assert t["thr"].tolist() == thr and t["idx"].tolist() == starts.tolist()


def test_thresholds_ignore_what_lies_behind_a_separator():
    """All separators are the same byte, so the raw LCP of two suffixes that reach one together runs
    on into the NEXT records -- context no pattern can match.  The two conventions differ on this
    text; the files hold the capped one."""
    docs = [[b"CAT", b"GGGGA"], [b"CAT", b"GGGGC"], [b"TAT", b"GGGGA"]]
    res = ro.build(docs, min_len=2)
    raw = ro.thresholds(res["heads"], res["lens"], res["lcp"])
    assert res["thr"] == ro.thresholds(res["heads"], res["lens"], ro.capped_lcp(res["text"], res["sa"]))
    assert raw != res["thr"]


def test_construction_fails_loudly_without_a_device_and_on_bad_text(pkg, tmp_path):
    """No CPU fallback: argument and format errors first (they need no device), then NO_DEVICE."""
    with pytest.raises(pkg.ColbwtError) as e:
        pkg.rlbwt_from_text(b"ACGT\x01", [0])                       # no final 0
    assert e.value.code == -3
    with pytest.raises(pkg.ColbwtError) as e:
        pkg.rlbwt_from_text(b"AC\x00GT\x01\x00", [0])                # a 0 inside
    assert e.value.code == -3
    with pytest.raises(pkg.ColbwtError) as e:
        pkg.rlbwt_from_text(b"ACGT\x01ACGT\x01\x00", [0, 7, 5])       # document starts must ascend
    assert e.value.code == -1
    with pytest.raises(pkg.ColbwtError) as e:
        pkg.rlbwt_from_fastas([str(tmp_path / "missing.fa")], out_prefix=str(tmp_path / "x"))
    assert e.value.code == -2
    empty = tmp_path / "empty.fa"
    empty.write_bytes(b"")
    with pytest.raises(pkg.ColbwtError) as e:
        pkg.rlbwt_from_fastas([str(empty)], out_prefix=str(tmp_path / "x"))
    assert e.value.code == -2 and "no record" in str(e.value)
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(pkg.ColbwtError) as e:
            pkg.rlbwt_from_text(b"ACGT\x01ACGA\x01\x00", [0, 5])
        assert e.value.code == -4
    exe = os.path.join(ROOT, "col-bwt_amd", "build_rlbwt")
    out = subprocess.run([exe, "a.fa"], capture_output=True, text=True)              # no -o
    assert out.returncode == 1 and "usage" in out.stderr
    out = subprocess.run([exe, "-o", str(tmp_path / "x"), str(tmp_path / "missing.fa")], capture_output=True, text=True)
    assert out.returncode == 1 and "cannot open" in out.stderr
    launcher = subprocess.run([sys.executable, os.path.join(ROOT, "col-bwt_amd", "col-bwt"), "build", "-o", str(tmp_path / "y")],
                              capture_output=True, text=True)
    assert launcher.returncode == 1 and "required" in launcher.stdout


def compare(pkg, docs, min_len, revcomp, label):
    res = ro.build(docs, min_len=min_len, revcomp=revcomp)
    got = pkg.rlbwt_from_text(res["text"], res["doc_start"], min_mum=min_len)
    assert got["n"] == len(res["text"]), label
    assert got["heads"].tolist() == res["heads"], label
    assert got["lens"].tolist() == res["lens"], label
    assert got["thr"].tolist() == res["thr"], (label, np.flatnonzero(got["thr"] != np.array(res["thr"], np.uint64))[:5])
    assert list(zip(got["mum_len"].tolist(), got["mum_pos"].tolist())) == res["mums"], label
    return res, got


def emu_env():
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    return dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")


def test_emulated_construction_matches_oracle():
    """The product's construction (HIP sources under the SIMT emulator + ASan; rocPRIM's sorts and
    scans replaced by <algorithm>) == the oracle."""
    emu = os.path.join(HERE, "emu")
    subprocess.check_call(["make", "-C", emu, "libcolbwt_emu.so"], stdout=subprocess.DEVNULL)
    out = subprocess.run([sys.executable, os.path.join(HERE, "test_rlbwt.py"), "emu"], env=emu_env(),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "RLBWT-EMU-OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def emu_main():
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    pkg = load_package()
    pkg.LIB_PATH = os.path.join(HERE, "emu", "libcolbwt_emu.so")
    rng = np.random.default_rng(21)
    compare(pkg, [[b"GATTACA"], [b"GATTACCGATAACA"]], 3, False, "survey")
    compare(pkg, [[b"A"], [b"A"]], 1, False, "tiny")
    for k in range(5):
        nd = int(rng.integers(1, 5))
        docs = related_docs(rng, nd, int(rng.integers(40, 700)), 0.04, records=int(rng.integers(1, 3)), with_n=k == 2)
        compare(pkg, docs, int(rng.integers(1, 12)), bool(k & 1), f"random{k}")
    # a very repetitive text: many doubling rounds
    compare(pkg, [[b"ACACACAC" * 40], [b"ACACACAC" * 40 + b"G"]], 5, False, "repeats")
    # long gaps between the runs of a rare character: the threshold pass steps over whole 2048-position blocks
    compare(pkg, related_docs(rng, 2, 5200, 0.02, with_n=True), 8, False, "blocks")
    print("RLBWT-EMU-OK")


@pytest.mark.gpu
def test_gpu_construction_matches_oracle(pkg):
    rng = np.random.default_rng(8)
    compare(pkg, [[b"GATTACA"], [b"GATTACCGATAACA"]], 3, False, "survey")
    compare(pkg, [[b"A"], [b"A"]], 1, False, "tiny")
    compare(pkg, [[b"ACACACAC" * 500], [b"ACACACAC" * 500 + b"G"]], 5, False, "repeats")
    total = 0
    for k in range(8):
        nd = int(rng.choice([1, 2, 3, 8, 33]))
        docs = related_docs(rng, nd, int(rng.integers(500, 60_000 // nd + 600)), 0.01, records=int(rng.integers(1, 4)), with_n=k % 3 == 0)
        res, got = compare(pkg, docs, int(rng.choice([5, 12, 20])), bool(k & 1), f"random{k}")
        total += len(res["mums"])
    assert total > 50
    # many short documents: windows of 300 suffixes, a document bit set of ten words
    base = rng.choice(ACGT, size=90)
    docs = []
    for _ in range(300):                                             # substitutions in the first half only: the rest is shared
        s = base.copy()
        at = rng.integers(0, 45, size=2)
        s[at] = rng.choice(ACGT, size=2)
        docs.append([bytes(s)])
    res, got = compare(pkg, docs, 6, False, "many_docs")
    assert len(res["mums"]) >= 1 and max(m[0] for m in res["mums"]) >= 45
    with pytest.raises(pkg.ColbwtError) as e:                       # more documents than the scan's bit set holds
        pkg.rlbwt_from_text(b"A\x01" * 4097 + b"\x00", list(range(0, 2 * 4097, 2)))
    assert e.value.code == -3
    # one larger text: 4 x 250 kbp with reverse complements (2 Mbp, > 15 doubling rounds not needed: random)
    docs = related_docs(rng, 4, 250_000, 0.005)
    res, got = compare(pkg, docs, 20, True, "large")
    assert len(res["mums"]) > 1000


@pytest.mark.gpu
def test_gpu_chain_from_fasta_to_query(pkg, oracle, tmp_path):
    """build_rlbwt -> col_split -> build_col_bwt -> pml_query from FASTA files, every intermediate
    file equal to the oracle chain's (rlbwt_oracle -> colsplit_oracle -> builder restatement ->
    query oracle)."""
    rng = np.random.default_rng(17)
    docs = related_docs(rng, 5, 6000, 0.01, records=2)
    paths = []
    for d, records in enumerate(docs):
        p = str(tmp_path / f"g{d}.fa")
        with open(p, "wb") as f:
            for k, rec in enumerate(records):
                f.write(b">doc%d_%d\n" % (d, k))
                for a in range(0, len(rec), 70):
                    f.write(rec[a:a + 70] + b"\n")
        paths.append(p)
    prefix = str(tmp_path / "idx.fa")
    lst = str(tmp_path / "list.txt")
    open(lst, "w").write("".join(f"{p} {k + 1}\n" for k, p in enumerate(paths)))
    exe = lambda name: os.path.join(ROOT, "col-bwt_amd", name)
    out = subprocess.run([exe("build_rlbwt"), "-r", "-l", "20", "-i", lst, "-o", prefix], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    res = ro.build(docs, min_len=20, revcomp=True)
    assert len(res["mums"]) > 20
    for ext, want in zip((".bwt.heads", ".bwt.len", ".thr_pos", ".col_mums"), ro.file_bytes(res, len(docs))):
        assert open(prefix + ext, "rb").read() == want, ext
    out = subprocess.run([exe("col_split"), "-m", "tunnels", "-s", "2", prefix], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    heads, lens = np.array(res["heads"], np.uint8), np.array(res["lens"], np.uint64)
    mlen = np.array([m[0] for m in res["mums"]], np.uint64)
    mpos = np.array([m[1] for m in res["mums"]], np.uint64)
    epos, eids, en, stats = oracle.col_split(heads, lens, mlen, mpos, len(docs), "tunnels", 2)
    n = len(res["text"])
    raw = np.fromfile(prefix + ".col_runs", np.uint64)
    assert raw[0] == n == en
    assert np.array_equal(np.flatnonzero(np.unpackbits(raw[1:].view(np.uint8), bitorder="little")[:n]), epos)
    assert np.array_equal(np.fromfile(prefix + ".col_ids", np.uint8), eids)
    out = subprocess.run([exe("build_col_bwt"), prefix], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    image = open(prefix + ".col_pml", "rb").read()
    assert image == oracle.build_col_pml(heads, lens, eids, epos, np.array(res["thr"], np.uint64)).tobytes()
    reads = helpers.reads_from_text(res["text"], 400, (30, 300), 0.02, seed=3)
    bases, off = helpers.concat_reads(reads)
    ep, ec = oracle.OracleIndex(image).query_batch(bases, off)
    tbl = pkg.ColPml.load(prefix)
    p, c, _ = tbl.query_batch(bases, off)
    tbl.close()
    assert np.array_equal(p, ep) and np.array_equal(c, ec) and (ec > 0).any()
    # PMLs of error-free stretches grow to the read's length (a real BWT: matches are found)
    assert int(ep.max()) >= 100
    # the launcher's `build` (the reference's command line, col-bwt.py:209-222) leaves the same index
    outp = str(tmp_path / "launched")
    out = subprocess.run([sys.executable, exe("col-bwt"), "build", "-r", "-l", "20", "-m", "tunnels", "-s", "2", "-o", outp] + paths,
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert open(outp + ".col_pml", "rb").read() == image
    assert not os.path.exists(outp + ".fa.col_mums")           # intermediates removed without --keep
    # --keep leaves them; a second run then skips the first step unless --force (col-bwt.py:121-135,
    # 219); --clean removes them whatever --keep says (col-bwt.py:184-186, 223)
    common = ["-r", "-l", "20", "-m", "tunnels", "-s", "2", "-o", outp] + paths
    out = subprocess.run([sys.executable, exe("col-bwt"), "build", "--keep"] + common, capture_output=True, text=True)
    assert out.returncode == 0 and os.path.exists(outp + ".fa.col_mums") and "skipping" not in out.stdout
    out = subprocess.run([sys.executable, exe("col-bwt"), "build", "--keep"] + common, capture_output=True, text=True)
    assert out.returncode == 0 and "already found, skipping" in out.stdout
    out = subprocess.run([sys.executable, exe("col-bwt"), "build", "--keep", "--force", "--clean"] + common, capture_output=True, text=True)
    assert out.returncode == 0 and "skipping" not in out.stdout and not os.path.exists(outp + ".fa.col_mums")
    assert open(outp + ".col_pml", "rb").read() == image


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "emu":
    emu_main()
