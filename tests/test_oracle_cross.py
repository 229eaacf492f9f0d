"""C oracle vs an independent pure-Python restatement of SURVEY.md B.2 on
small true-BWT and synthetic indices (incl. N / lowercase / terminator bytes)."""
import numpy as np

import helpers
import py_restatement


def _compare(oracle, image, reads):
    x = oracle.OracleIndex(bytes(image))
    for rd in reads:
        p, c = x.query_pml(bytes(rd))
        ep, ec = py_restatement.query_pml(image, bytes(rd))
        assert p.tolist() == ep and c.tolist() == ec


def test_true_bwt(oracle):
    rng = np.random.default_rng(1)
    base = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=200)
    seqs = []
    for _ in range(3):
        s = base.copy()
        mut = rng.random(200) < 0.04
        s[mut] = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(mut.sum()))
        seqs.append(bytes(s))
    image, text = helpers.true_bwt_index(seqs, seed=2, extra_splits=25)
    reads = helpers.reads_from_text(text, 25, (1, 80), 0.05, seed=3, extra=b"Nacgt\x01")
    _compare(oracle, image, reads + [np.zeros(0, np.uint8)])


def test_true_bwt_exact_substring_locks_on(oracle):
    """Semantic sanity: an exact substring of the text, once PML has locked on,
    extends by one per base (README.md:16-22)."""
    rng = np.random.default_rng(4)
    seq = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=400))
    image, text = helpers.true_bwt_index([seq], seed=5, extra_splits=10)
    x = oracle.OracleIndex(bytes(image))
    p, _ = x.query_pml(text[100:180])
    p = p.tolist()                      # p[k] for pattern[k]; processing goes k = m-1 .. 0
    tail = p[:40]                       # the last 40 processed bases
    assert all(tail[k] == tail[k + 1] + 1 for k in range(len(tail) - 1))


def test_synthetic_rows(oracle, pkg):
    rng = np.random.default_rng(6)
    for rows, split in ((300, 0), (900, 200)):
        image = pkg.synth_index(rows, mean_len=5, split_permille=split, seed=rows).tobytes()
        reads = helpers.backward_walk_reads(image, 10, 60, 0.03, seed=7)
        reads += [rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=int(m)) for m in rng.integers(1, 50, size=10)]
        _compare(oracle, image, reads)


def test_large_index_builder_equals_the_naive_one():
    """tests/helpers.true_bwt_index_large (prefix doubling + rank lifting, used for the C1-shaped
    GPU test) against the quadratic textbook construction on a small text."""
    import helpers
    rng = np.random.default_rng(3)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    base = rng.choice(acgt, size=400)
    seqs = []
    for _ in range(4):
        s = base.copy()
        mut = rng.random(len(s)) < 0.03
        s[mut] = rng.choice(acgt, size=int(mut.sum()))
        seqs.append(bytes(s))
    a, ta = helpers.true_bwt_index(seqs, seed=5, extra_splits=30)
    b, tb = helpers.true_bwt_index_large(seqs, seed=5, extra_splits=30)
    assert ta == tb and bytes(a) == bytes(b)
