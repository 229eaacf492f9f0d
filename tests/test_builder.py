"""SURVEY.md 8(f) "next" #1 -- the `.col_pml` builder (build_col_bwt,
src/build_col_bwt.cpp:38-52): the product's host builder
(col-bwt_amd/csrc/builder.cpp, through the C-ABI) vs the oracle's line-by-line
restatement of the reference constructor, both pinned by the Appendix D KAT
(whose 302-byte index the reference constructor wrote),
plus a third independent implementation (tests/helpers.true_bwt_index)."""
import os
import struct

import numpy as np

import helpers

KAT_BWT = b"ACTATTCGGGAAACC\x01AATTAA"


def _rle(bwt):
    b = np.frombuffer(bwt, np.uint8)
    heads_at = np.flatnonzero(np.concatenate(([True], b[1:] != b[:-1])))
    lens = np.diff(np.append(heads_at, len(b)))
    return b[heads_at], lens.astype(np.uint64), heads_at


def _kat_inputs():
    heads, lens, heads_at = _rle(KAT_BWT)
    splits = np.array(sorted(set(heads_at.tolist()) | {5, 8}), np.uint64)     # every run head + positions 5 and 8
    ids = np.array([(7 * k) % 5 for k in range(len(splits))], np.uint8)       # ids (7k mod 5)
    thr = np.array([0, 0, 0, 1, 3, 2, 0, 10, 10, 0, 14, 10, 20], np.uint64)   # per BWT run (from the KAT rows)
    return heads, lens, ids, splits, thr


def test_kat_index_rebuilt_byte_for_byte(pkg, oracle, golden_dir):
    gold = open(os.path.join(golden_dir, "kat_d.col_pml"), "rb").read()
    heads, lens, ids, splits, thr = _kat_inputs()
    assert oracle.build_col_pml(heads, lens, ids, splits, thr).tobytes() == gold
    assert pkg.build_col_pml_arrays(heads, lens, ids, splits, thr).tobytes() == gold


def test_file_interface_build_col_bwt(pkg, golden_dir, tmp_path):
    """build_col_bwt <prefix>: the flat input files of SURVEY.md Appendix A."""
    heads, lens, ids, splits, thr = _kat_inputs()
    prefix = str(tmp_path / "kat")
    open(prefix + ".bwt.heads", "wb").write(heads.tobytes())
    open(prefix + ".bwt.len", "wb").write(b"".join(int(v).to_bytes(5, "little") for v in lens))
    open(prefix + ".col_ids", "wb").write(ids.tobytes())
    open(prefix + ".thr_pos", "wb").write(b"".join(int(v).to_bytes(5, "little") for v in thr))
    n = int(lens.sum())
    words = [0] * ((n + 63) // 64)
    for p in splits.tolist():
        words[p // 64] |= 1 << (p % 64)
    open(prefix + ".col_runs", "wb").write(struct.pack("<Q", n) + struct.pack(f"<{len(words)}Q", *words))
    pkg.build_col_pml(prefix)
    assert open(prefix + ".col_pml", "rb").read() == open(os.path.join(golden_dir, "kat_d.col_pml"), "rb").read()
    try:
        pkg.build_col_pml(str(tmp_path / "missing"))
        raise AssertionError("missing inputs must fail")
    except pkg.ColbwtError as e:
        assert e.code == -2


def test_random_inputs_match_reference_restatement(pkg, oracle):
    rng = np.random.default_rng(7)
    for trial in range(60):
        n_runs = int(rng.integers(2, 200))
        alpha = np.frombuffer(b"ACGT\x01N", np.uint8) if trial % 3 else np.array([0, 1, 65, 67, 0x80, 0xC8, 71], np.uint8)
        heads = rng.choice(alpha, size=n_runs)
        for k in range(1, n_runs):                       # RLBWT: adjacent heads differ
            while heads[k] == heads[k - 1]:
                heads[k] = rng.choice(alpha)
        lens = rng.integers(1, 30, size=n_runs).astype(np.uint64)
        if trial % 7 == 0:
            lens[rng.integers(0, n_runs)] = 70000        # offsets beyond the 16-bit field
        n = int(lens.sum())
        heads_at = np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.int64)
        marked = heads_at if trial % 4 else heads_at[rng.random(n_runs) < 0.8]   # some run heads unmarked
        extra = rng.choice(n, size=min(n, int(rng.integers(0, 80))), replace=False)
        splits = np.array(sorted(set(marked.tolist()) | set(extra.tolist())), np.uint64)
        if splits.size == 0:
            splits = np.array([0], np.uint64)
        n_ids = splits.size if trial % 5 else max(1, splits.size - 3)            # short .col_ids
        ids = rng.integers(0, 256, size=n_ids).astype(np.uint8)
        n_thr = n_runs if trial % 6 else max(1, n_runs - 2)                      # short .thr_pos
        thr = rng.integers(0, n, size=n_thr).astype(np.uint64)
        a = oracle.build_col_pml(heads, lens, ids, splits, thr)
        b = pkg.build_col_pml_arrays(heads, lens, ids, splits, thr)
        assert a.tobytes() == b.tobytes(), f"trial {trial}"


def test_agrees_with_python_true_bwt_pipeline(pkg, oracle):
    """Third implementation: helpers.true_bwt_index builds the same kind of table
    with numpy; feeding its ingredients through the builder gives the same bytes,
    and the result answers queries like the python-built one."""
    rng = np.random.default_rng(3)
    seq = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=600))
    image, text = helpers.true_bwt_index([seq], seed=4, extra_splits=50)
    t = helpers.unpack_col_pml(image)
    n = int(t["n"])
    # recover the builder's inputs from the python-built table
    heads_mask = np.concatenate(([True], t["char"][1:] != t["char"][:-1]))
    run_start = t["idx"][heads_mask]
    heads = t["char"][heads_mask]
    lens = np.diff(np.append(run_start, n)).astype(np.uint64)
    splits = t["idx"]                                       # every sub-run start is a split bit
    built = pkg.build_col_pml_arrays(heads, lens, t["cid"], splits, t["thr"][heads_mask])
    assert built.tobytes() == bytes(image)
    assert oracle.build_col_pml(heads, lens, t["cid"], splits, t["thr"][heads_mask]).tobytes() == bytes(image)
