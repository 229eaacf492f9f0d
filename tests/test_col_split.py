"""Sub-run splitter (SURVEY.md 8(f) "next" #2: src/col_split.cpp + src/build_FL.cpp).

CPU tier: the oracle's restatement (oracle/colsplit_oracle.c) against the only reference-produced
numbers there are -- the counts SURVEY.md Appendix C.6 recorded from the compiled reference -- and
the product (col-bwt_amd/csrc/col_split.hip, compiled against the SIMT emulator) against the oracle.
GPU tier (-m gpu): the HIP kernels against the oracle on larger random inputs, then the chain
col_split -> build_col_pml -> query against the oracle's chain, end to end.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers

HERE = os.path.dirname(os.path.abspath(__file__))
TEXT_D = b"GATTACAGATTACCGATAACA\x01"


def rlbwt_of(text):
    n = len(text)
    sa = sorted(range(n), key=lambda i: text[i:])
    bwt = bytes(text[i - 1] for i in sa)
    heads, lens = [], []
    for c in bwt:
        if heads and heads[-1] == c:
            lens[-1] += 1
        else:
            heads.append(c)
            lens.append(1)
    return sa, np.array(heads, np.uint8), np.array(lens, np.uint64)


def test_oracle_reproduces_the_survey_counts(oracle):
    """Appendix C.6: the Appendix D text with one multi-MUM (GATTAC, len 6, SA rank 15, N = 2):
    tunnels => 15 sub-runs / 5 col runs / 10 col chars; all => 16 / 7 / 12."""
    sa, heads, lens = rlbwt_of(TEXT_D)
    assert [k for k, i in enumerate(sa) if TEXT_D[i:i + 6] == b"GATTAC"] == [15, 16]
    pos, ids, n, stats = oracle.col_split(heads, lens, [6], [15], 2, "tunnels", 1)
    assert n == 22 and stats == (5, 15, 10) and len(pos) == len(ids) == 15
    pos, ids, n, stats = oracle.col_split(heads, lens, [6], [15], 2, "all", 1)
    assert stats == (7, 16, 12) and len(pos) == len(ids) == 16
    # every BWT run head is a sub-run start, ids are 0 / 1 (one multi-MUM)
    run_heads = np.concatenate(([0], np.cumsum(lens)[:-1]))
    assert set(run_heads.tolist()) <= set(pos.tolist()) and set(ids.tolist()) <= {0, 1}


def random_case(rng, r, n_mums, max_docs, alphabet=b"ACGT"):
    """Random RLBWT + multi-MUM list (positions ascending, ranges inside [0, n))."""
    alpha = np.frombuffer(alphabet, np.uint8)
    heads = rng.choice(alpha, size=r)
    for k in range(1, r):                                     # adjacent runs differ
        while heads[k] == heads[k - 1]:
            heads[k] = rng.choice(alpha)
    heads[rng.integers(0, r)] = 1                             # one terminator run
    lens = rng.integers(1, 12, size=r).astype(np.uint64)
    n = int(lens.sum())
    docs = int(rng.integers(1, max_docs + 1))
    mpos = np.sort(rng.integers(0, max(1, n - docs), size=n_mums)).astype(np.uint64)
    mlen = rng.integers(1, 40, size=n_mums).astype(np.uint64)
    return heads, lens, mlen, mpos, docs


def emu_env():
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    return dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")


def test_emulated_splitter_matches_oracle():
    """The product's splitter (HIP sources under the SIMT emulator + ASan) == the oracle, both modes,
    split rates 1 / 3 / 10, the Appendix C.6 case included."""
    emu = os.path.join(HERE, "emu")
    subprocess.check_call(["make", "-C", emu, "libcolbwt_emu.so"], stdout=subprocess.DEVNULL)
    out = subprocess.run([sys.executable, os.path.join(HERE, "test_col_split.py"), "emu"], env=emu_env(),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "SPLIT-EMU-OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def compare(pkg, oracle, heads, lens, mlen, mpos, docs, label):
    for mode in ("tunnels", "all"):
        for rate in (1, 3, 10):
            epos, eids, n, _ = oracle.col_split(heads, lens, mlen, mpos, docs, mode, rate)
            gpos, gids, gn = pkg.col_split_arrays(heads, lens, mlen, mpos, docs, mode, rate)
            assert gn == n, (label, mode, rate)
            assert np.array_equal(gpos, epos), (label, mode, rate, np.flatnonzero(gpos[:len(epos)] != epos[:len(gpos)])[:5])
            assert np.array_equal(gids, eids), (label, mode, rate)


def emu_main():
    sys.path.insert(0, os.path.dirname(HERE))
    from __graft_entry__ import load_oracle, load_package
    pkg, oracle = load_package(), load_oracle()
    pkg.LIB_PATH = os.path.join(HERE, "emu", "libcolbwt_emu.so")
    sa, heads, lens = rlbwt_of(TEXT_D)
    compare(pkg, oracle, heads, lens, [6], [15], 2, "appendix_c6")
    rng = np.random.default_rng(7)
    for k in range(6):
        heads, lens, mlen, mpos, docs = random_case(rng, int(rng.integers(5, 400)), int(rng.integers(0, 25)), 5)
        compare(pkg, oracle, heads, lens, mlen, mpos, docs, f"random{k}")
    print("SPLIT-EMU-OK")


@pytest.mark.gpu
def test_gpu_splitter_matches_oracle(pkg, oracle):
    rng = np.random.default_rng(11)
    sa, heads, lens = rlbwt_of(TEXT_D)
    compare(pkg, oracle, heads, lens, [6], [15], 2, "appendix_c6")
    for k in range(12):
        heads, lens, mlen, mpos, docs = random_case(rng, int(rng.integers(50, 60_000)), int(rng.integers(0, 3000)),
                                                    int(rng.choice([1, 2, 7, 64, 200])))
        compare(pkg, oracle, heads, lens, mlen, mpos, docs, f"random{k}")
    # MUM list that goes backwards: the reference's loop stalls there for good (col_split.hpp:77)
    heads, lens, mlen, mpos, docs = random_case(rng, 3000, 200, 4)
    mpos[120] = 0
    compare(pkg, oracle, heads, lens, mlen, mpos, docs, "backwards")


@pytest.mark.gpu
def test_gpu_split_build_query_chain(pkg, oracle, tmp_path):
    """col_split -> build_col_pml -> pml_query on the files a mumemto run would leave (a true BWT
    with min-LCP thresholds, multi-MUMs found by brute force on the suffix array): every file the
    product writes equals the oracle chain's, and the queries agree."""
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    base = rng.choice(acgt, size=1200)
    docs = []
    for _ in range(4):
        s = base.copy()
        mut = rng.random(len(s)) < 0.03
        s[mut] = rng.choice(acgt, size=int(mut.sum()))
        docs.append(bytes(s))
    text = b"".join(d + b"\x02" for d in docs)[:-1] + b"\x01"   # 0x02 separates documents (folds to the terminator class)
    sa, heads, lens = rlbwt_of(text)
    n = len(text)
    # multi-MUM candidates: N = 4 consecutive suffixes sharing >= 20 characters, one per document
    bounds = np.cumsum([len(d) + 1 for d in docs])
    doc_of = lambda i: int(np.searchsorted(bounds, i, side="right"))
    mums = []
    k = 0
    while k + 4 <= n:
        grp = sa[k:k + 4]
        l = 0
        while all(g + l < n for g in grp) and len({text[g + l] for g in grp}) == 1 and text[grp[0] + l] > 2:
            l += 1
        if l >= 20 and len({doc_of(g) for g in grp}) == 4:
            mums.append((l, k))
            k += 4
        else:
            k += 1
    assert len(mums) >= 3
    mlen = np.array([m[0] for m in mums], np.uint64)
    mpos = np.array([m[1] for m in mums], np.uint64)
    thr = np.zeros(len(heads), np.uint64)                      # any thresholds do for the chain; keep them in range
    thr[:] = rng.integers(0, n, size=len(heads))
    prefix = str(tmp_path / "idx")

    def le5(a):
        return b"".join(int(x).to_bytes(5, "little") for x in a)
    open(prefix + ".bwt.heads", "wb").write(bytes(heads))
    open(prefix + ".bwt.len", "wb").write(le5(lens))
    open(prefix + ".thr_pos", "wb").write(le5(thr))
    open(prefix + ".col_mums", "wb").write(le5([4]) + b"".join(le5([l, p]) for l, p in zip(mlen, mpos)))
    root = os.path.dirname(HERE)
    for mode in ("tunnels", "all"):
        out = subprocess.run([os.path.join(root, "col-bwt_amd", "col_split"), "-m", mode, "-s", "2", prefix], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        epos, eids, en, stats = oracle.col_split(heads, lens, mlen, mpos, 4, mode, 2)
        raw = np.fromfile(prefix + ".col_runs", np.uint64)
        assert raw[0] == n == en
        got = np.flatnonzero(np.unpackbits(raw[1:].view(np.uint8), bitorder="little")[:n])
        assert np.array_equal(got, epos) and np.array_equal(np.fromfile(prefix + ".col_ids", np.uint8), eids)
        assert stats[0] > 0                                    # some col runs exist
        pkg.build_col_pml(prefix)                              # product builder on the product splitter's files
        image = open(prefix + ".col_pml", "rb").read()
        expect = oracle.build_col_pml(heads, lens, eids, epos, thr)
        assert image == expect.tobytes()
        reads = helpers.reads_from_text(text, 300, (20, 200), 0.02, seed=9)
        bases, off = helpers.concat_reads(reads)
        ep, ec = oracle.OracleIndex(image).query_batch(bases, off)
        tbl = pkg.ColPml.load(prefix)
        p, c, _ = tbl.query_batch(bases, off)
        assert np.array_equal(p, ep) and np.array_equal(c, ec)
        assert (ec > 0).any()                                  # chain statistics show up in the query
        tbl.close()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "emu":
    emu_main()


@pytest.mark.gpu
def test_builder_cases_on_the_gpu_box(pkg, oracle, golden_dir, tmp_path):
    """SURVEY.md 8(f) "next" #1 in the driver's GPU record too: the builder is host code, its CPU-tier
    cases (tests/test_builder.py: the Appendix D index byte for byte, random inputs against the
    oracle's restatement of the reference constructor) run unchanged here."""
    import test_builder
    for name in sorted(dir(test_builder)):
        if name.startswith("test_"):
            fn = getattr(test_builder, name)
            kwargs = {k: v for k, v in (("pkg", pkg), ("oracle", oracle), ("golden_dir", golden_dir), ("tmp_path", tmp_path))
                      if k in fn.__code__.co_varnames[:fn.__code__.co_argcount]}
            fn(**kwargs)
