"""The oracle vs the only reference-produced vector available: SURVEY.md
Appendix D (tests/golden/kat_d.*, written by make_kat_appendix_d.py)."""
import os
import shutil

import numpy as np


def test_kat_values(oracle, golden_dir):
    x = oracle.OracleIndex(os.path.join(golden_dir, "kat_d.col_pml"))
    assert (x.bwt_r, x.n, x.r) == (13, 22, 15)
    exp = {
        b"GATTACA": ([5, 4, 3, 2, 1, 0, 1], [1, 3, 1, 3, 3, 1, 3]),
        b"TTACCGATNACA": ([4, 3, 2, 1, 0, 3, 2, 1, 0, 1, 0, 1], [1, 0, 3, 0, 2, 1, 3, 1, 3, 3, 1, 3]),
        b"CCCC": ([0, 0, 1, 0], [3, 3, 0, 3]),
    }
    for read, (pml, cid) in exp.items():
        p, c = x.query_pml(read)
        assert p.tolist() == pml and c.tolist() == cid


def test_kat_text_files(oracle, golden_dir, tmp_path):
    """pml_to_vec byte format incl. the header's trailing space and the kseq
    name rule (">q1 desc" -> "q1"), pml_query.cpp:78-85 / io.hpp:24-26."""
    x = oracle.OracleIndex(os.path.join(golden_dir, "kat_d.col_pml"))
    fa = tmp_path / "kat_d.fa"
    shutil.copy(os.path.join(golden_dir, "kat_d.fa"), fa)
    x.pml_query_files(str(fa))
    for ext in (".pml", ".cid"):
        assert open(str(fa) + ext, "rb").read() == open(os.path.join(golden_dir, "kat_d.fa" + ext), "rb").read()


def test_kat_file_is_what_the_script_writes(golden_dir, tmp_path):
    import subprocess, sys
    for f in os.listdir(golden_dir):
        shutil.copy(os.path.join(golden_dir, f), tmp_path / f)
    subprocess.check_call([sys.executable, str(tmp_path / "make_kat_appendix_d.py")])
    for f in ("kat_d.col_pml", "kat_d.fa", "kat_d.fa.pml", "kat_d.fa.cid"):
        assert open(tmp_path / f, "rb").read() == open(os.path.join(golden_dir, f), "rb").read()


def test_batch_matches_single(oracle, golden_dir):
    x = oracle.OracleIndex(os.path.join(golden_dir, "kat_d.col_pml"))
    reads = [b"GATTACA", b"", b"TTACCGATNACA", b"CCCC"]
    bases = np.frombuffer(b"".join(reads), np.uint8)
    off = np.cumsum([0] + [len(r) for r in reads]).astype(np.uint64)
    for threads in (1, 3):
        pml, cid = x.query_batch(bases, off, threads=threads)
        for k, rd in enumerate(reads):
            p, c = x.query_pml(rd)
            assert pml[int(off[k]):int(off[k + 1])].tolist() == p.tolist()
            assert cid[int(off[k]):int(off[k + 1])].tolist() == c.tolist()
