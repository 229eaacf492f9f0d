"""CPU tier: the C-ABI library loads and exports every symbol include/colbwt.h
declares; without a GPU the engine fails loudly instead of computing on the CPU."""
import ctypes as C
import re

import numpy as np
import pytest


def _declared(header_path):
    src = open(header_path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(colbwt_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(pkg):
    names = _declared(pkg.HEADER_PATH)
    assert set(names) == set(pkg.EXPORTS)
    L = pkg.lib()
    for nm in names:
        assert getattr(L, nm) is not None
    assert b"gfx950" in L.colbwt_version()


def test_no_cpu_fallback_without_device(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    image = pkg.synth_index(100, mean_len=4, seed=1)
    with pytest.raises(pkg.ColbwtError) as ei:
        pkg.ColPml.from_bytes(image)
    assert ei.value.code == -4          # COLBWT_ERR_NO_DEVICE


def test_synth_index_is_a_consistent_move_table(pkg):
    """Host-side generator: idx = prefix sums, (interval, offset) = compute_table
    (LF_table.hpp:365-387), sub-runs of a BWT run share a threshold."""
    import helpers
    for rows, split in ((1000, 0), (5000, 250)):
        img = pkg.synth_index(rows, mean_len=7, split_permille=split, seed=rows)
        t = helpers.unpack_col_pml(img.tobytes())
        assert t["r"] == rows and t["idx"][0] == 0 and np.all(np.diff(t["idx"].astype(np.int64)) > 0)
        assert int(t["idx"][-1]) < t["n"]
        itv, off = helpers.lf_columns(t["char"], t["idx"], t["n"])
        assert np.array_equal(itv, t["interval"]) and np.array_equal(off, t["offset"])
        heads = np.concatenate(([True], t["char"][1:] != t["char"][:-1]))
        assert int(heads.sum()) == t["bwt_r"]
        same = ~heads[1:]
        assert np.all(t["thr"][1:][same] == t["thr"][:-1][same])
        assert (t["char"] == 1).sum() == 1


def test_synth_index_thresholds_between_runs(pkg):
    """thr_mode 1: the threshold of a BWT run lies between the end of the previous run of
    the same character and the run's head (0 for a character's first run); everything else
    is the same table as thr_mode 0."""
    import helpers
    img0 = helpers.unpack_col_pml(pkg.synth_index(4000, mean_len=6, split_permille=100, seed=5).tobytes())
    t = helpers.unpack_col_pml(pkg.synth_index(4000, mean_len=6, split_permille=100, seed=5, thr_mode=1).tobytes())
    for k in ("char", "idx", "interval", "offset", "cid"):
        assert np.array_equal(img0[k], t[k])
    idx = t["idx"].astype(np.int64)
    end = np.append(idx[1:], t["n"])
    last_end = {}
    for i in range(int(t["r"])):
        c = int(t["char"][i])
        if i == 0 or t["char"][i - 1] != c:
            lo = last_end.get(c)
            if lo is None:
                assert t["thr"][i] == 0
            else:
                assert lo <= int(t["thr"][i]) <= idx[i]
            head_thr = t["thr"][i]
        assert t["thr"][i] == head_thr
        last_end[c] = int(end[i])
