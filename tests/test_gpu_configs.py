"""GPU tier (-m gpu): the BASELINE.json configurations beyond C1/C2's 150 bp batch, and the
failure paths the capacity configuration depends on.

  C3  the per-rank shard of the 8-GPU run is the C2 batch (same index, 10 M x 150 bp per GPU):
      covered by test_gpu_parity.py::test_full_scale_properties and named here.
  C4  2e8-row index, >= 100 k reads of ~10 kbp (+-20 % length jitter, 5 % substitutions), all
      four HBM layouts, device entry point with and without the length order, host entry point;
      and the configuration's own size, 1 M reads (1e10 bases), on line rows: two runs and the
      three-step layout compared on the device, the oracle on the first, middle and last reads.
  C5  capacity: the largest indices one MI355X takes -- 1e9 rows opened with AUTO (line rows do
      not fit, which the build finds out after one counting pass: three-step rows, ~130 GB resident), 1.7e9 rows where the three-step refinement passes 2^32-2 rows and
      AUTO must settle for two-step rows -- plus the HBM-budget fallback and what a failed open
      leaves behind.
Sizes shrink with COLBWT_TEST_ROWS / COLBWT_TEST_BIG_ROWS for rehearsals.
"""
import gc
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

C2_ROWS = int(os.environ.get("COLBWT_TEST_ROWS", "200000000"))
BIG_ROWS = int(os.environ.get("COLBWT_TEST_BIG_ROWS", "1000000000"))
LIMIT_ROWS = int(os.environ.get("COLBWT_TEST_LIMIT_ROWS", "1700000000"))


def _free_hbm():
    import torch
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return torch.cuda.mem_get_info(0)[0]


def _ragged_suffixes(torch, d_fixed, n_reads, m_max, lens):
    """Read k := the last lens[k] bases of fixed-length read k (a suffix of a backward-walk
    read is a backward-walk read).  Returns (bases with 128 pad bytes, int64 offsets)."""
    dev = d_fixed.device
    lens = lens.to(dev)
    off = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(lens, 0)
    total = int(off[-1])
    rid = torch.repeat_interleave(torch.arange(n_reads, device=dev), lens)
    src = torch.arange(total, device=dev) - off[rid] + rid * m_max + (m_max - lens)[rid]
    out = torch.zeros(total + 128, dtype=torch.uint8, device=dev)
    out[:total] = d_fixed[src]
    return out, off


def test_c3_per_rank_shard_is_the_c2_batch(pkg):
    """BASELINE configs[2] (8 GPUs, 80 M x 150 bp): every rank holds the whole index and queries
    10 M x 150 bp -- exactly C2.  Single-GPU parity at that shape:
    test_gpu_parity.py::test_full_scale_properties (all layouts) and bench.py's on-box oracle
    check; the exchange step: test_gather_codec_round_trip_and_pipeline + the gloo tests."""
    from colbwt_amd import multi_gpu
    off = np.arange(0, 150 * 80_000 + 1, 150, dtype=np.uint64)
    shards = multi_gpu.shard_reads(off, 8)
    assert [hi - lo for lo, hi in shards] == [10_000] * 8


def test_c4_long_reads_all_layouts(pkg, oracle, c2_image):
    """BASELINE configs[3]: ONT-length reads.  100 k reads of 8-12 kbp (1e9 bases) by backward
    walk with 5 % substitutions on the 2e8-row index: every layout gives the same bytes, with
    and without the length order, twice; the host entry point returns the same; the oracle
    agrees on a >= 50 Mbase sample."""
    import torch
    dev = torch.device("cuda", 0)
    n_reads = int(os.environ.get("COLBWT_TEST_LONG_READS", "100000"))
    m_max = 12_000
    rng = np.random.default_rng(4)
    lens = torch.from_numpy(rng.integers(8_000, m_max + 1, size=n_reads))
    ref = oracle.OracleIndex(c2_image)
    expect = None
    host_result = None
    for layout in (4, 3, 2, 1):
        _free_hbm()
        tbl = pkg.ColPml.from_bytes(c2_image, layout=layout)
        assert tbl.info().layout == layout
        d_fixed = torch.zeros(n_reads * m_max + 128, dtype=torch.uint8, device=dev)
        d_foff = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
        tbl.synth_reads_device(n_reads, m_max, 50, 44, d_fixed.data_ptr(), d_foff.data_ptr())
        torch.cuda.synchronize()
        d_bases, d_off = _ragged_suffixes(torch, d_fixed, n_reads, m_max, lens)
        del d_fixed, d_foff
        nb = int(d_off[-1])
        d_order = torch.argsort(lens.to(dev), descending=True, stable=True).to(torch.int32)

        def run(order):
            p = torch.zeros(nb + 16, dtype=torch.int16, device=dev)
            c = torch.zeros(nb + 16, dtype=torch.uint8, device=dev)
            tbl.query_device(d_bases.data_ptr(), d_off.data_ptr(), n_reads, nb, p.data_ptr(), c.data_ptr(), 2, 0,
                             d_order=order.data_ptr() if order is not None else None)
            torch.cuda.synchronize()
            return p[:nb], c[:nb]
        p1, c1 = run(d_order)
        p2, c2 = run(d_order)
        assert torch.equal(p1, p2) and torch.equal(c1, c2)                  # idempotent
        p3, c3 = run(None)
        assert torch.equal(p1, p3) and torch.equal(c1, c3)                  # lane assignment is invisible
        del p2, c2, p3, c3
        hp, hc = p1.cpu().numpy().view(np.uint16), c1.cpu().numpy()
        if expect is None:
            # oracle on the first reads (>= 50 Mbase)
            k = min(n_reads, 5_500)
            off = d_off[:k + 1].cpu().numpy().astype(np.uint64)
            bases = d_bases[:int(off[-1])].cpu().numpy()
            ep, ec = ref.query_batch(bases, off, threads=16)
            assert off[-1] >= min(50_000_000, nb * 0.9)
            assert np.array_equal(hp[:int(off[-1])], ep) and np.array_equal(hc[:int(off[-1])], ec)
            resets = float((ep == 0).mean())
            assert 0.05 < resets < 0.7
            expect = (hp, hc)
            # host entry point, whole batch (pageable numpy buffers: staged D2H)
            hb = d_bases[:nb].cpu().numpy()
            ho = d_off.cpu().numpy().astype(np.uint64)
            qp, qc, st = tbl.query_batch(hb, ho)
            assert st.n_bases == nb
            host_result = bool(np.array_equal(qp, hp) and np.array_equal(qc, hc))
            del hb, qp, qc
        else:
            assert np.array_equal(hp, expect[0]) and np.array_equal(hc, expect[1]), f"layout {layout} differs"
            # host entry point on a slice of the batch
            k = min(n_reads, 2_000)
            off = d_off[:k + 1].cpu().numpy().astype(np.uint64)
            qp, qc, _ = tbl.query_batch(d_bases[:int(off[-1])].cpu().numpy(), off)
            assert np.array_equal(qp, hp[:int(off[-1])]) and np.array_equal(qc, hc[:int(off[-1])])
        del p1, c1, d_bases, d_off
        tbl.close()
    assert host_result


def test_c4_at_baseline_size_one_million_long_reads(pkg, oracle, c2_image):
    """BASELINE configs[3] at its own size: 1 M reads of 8-12 kbp (1e10 bases, 5 % substitutions,
    backward walks on the 2e8-row index) on the layout AUTO picks (line rows with mismatch lines) --
    five reads per persistent lane, so every
    lane claims further chunks from its workgroup's counter and the last tenth of every share goes
    out read by read (fat_query.hip ChunkPlan; the 100 k-read case above never gets there).
    Two runs are compared ON THE DEVICE, then with the three-step layout's run on the same
    buffers; the oracle (col_bwt.hpp:498-529 restated) checks >= 50 Mbase taken from the FIRST,
    MIDDLE and LAST reads of the batch -- the tail is where single-read chunks are claimed."""
    import torch
    dev = torch.device("cuda", 0)
    n_reads = int(os.environ.get("COLBWT_TEST_C4_READS", "1000000"))
    m_max, chunk = 12_000, 20_000
    rng = np.random.default_rng(14)
    lens_np = rng.integers(8_000, m_max + 1, size=n_reads)
    off_np = np.zeros(n_reads + 1, np.int64)
    off_np[1:] = np.cumsum(lens_np)
    nb = int(off_np[-1])
    _free_hbm()
    # AUTO under a budget that keeps room for this batch (10 GB of bases, 30 GB of results, twice the
    # results for the comparison): line rows with plain mismatch lines -- without the budget AUTO takes
    # deep entries on this index (229 GB of 288) and the batch does not fit next to it
    os.environ["COLBWT_HBM_BUDGET_MB"] = "230000"
    try:
        tbl = pkg.ColPml.from_bytes(c2_image, layout=0)
    finally:
        os.environ.pop("COLBWT_HBM_BUDGET_MB", None)
    assert tbl.info().layout == 5
    d_bases = torch.zeros(nb + 128, dtype=torch.uint8, device=dev)
    d_off = torch.from_numpy(off_np).to(dev)
    d_fixed = torch.zeros(chunk * m_max + 128, dtype=torch.uint8, device=dev)
    d_foff = torch.zeros(chunk + 1, dtype=torch.int64, device=dev)
    for lo in range(0, n_reads, chunk):                      # ragged suffixes of fixed-length walks, chunk by chunk
        k = min(chunk, n_reads - lo)
        tbl.synth_reads_device(k, m_max, 50, 500 + lo // chunk, d_fixed.data_ptr(), d_foff.data_ptr())
        part, part_off = _ragged_suffixes(torch, d_fixed, k, m_max, torch.from_numpy(lens_np[lo:lo + k]))
        assert int(part_off[-1]) == off_np[lo + k] - off_np[lo]
        d_bases[off_np[lo]:off_np[lo + k]] = part[:int(part_off[-1])]
        del part, part_off
    del d_fixed, d_foff
    torch.cuda.synchronize()

    def run(t):
        p = torch.full((nb + 64,), -1, dtype=torch.int16, device=dev)
        c = torch.full((nb + 64,), 0xEE, dtype=torch.uint8, device=dev)
        st = t.query_device(d_bases.data_ptr(), d_off.data_ptr(), n_reads, nb, p.data_ptr(), c.data_ptr(), 2, 0, timed=True)
        assert (p[nb:] == -1).all() and (c[nb:] == 0xEE).all()           # nothing written past the batch
        return p[:nb], c[:nb], st.kernel_ms
    p1, c1, ms1 = run(tbl)
    p2, c2, _ = run(tbl)
    assert torch.equal(p1, p2) and torch.equal(c1, c2)
    del p2, c2
    print(f"C4 at BASELINE size: {n_reads} reads, {nb} bases, line rows + mismatch lines {ms1:.1f} ms = {nb / ms1 / 1e6:.1f} Gbase/s")
    # the oracle on the first, middle and last reads
    ref = oracle.OracleIndex(c2_image)
    per = max(1, min(n_reads // 3, 1_800))
    checked = 0
    for lo in (0, (n_reads - per) // 2, n_reads - per):
        b0, b1 = int(off_np[lo]), int(off_np[lo + per])
        ep, ec = ref.query_batch(d_bases[b0:b1].cpu().numpy(), (off_np[lo:lo + per + 1] - b0).astype(np.uint64), threads=16)
        assert np.array_equal(p1[b0:b1].cpu().numpy().view(np.uint16), ep), f"PML differs from the oracle in reads {lo}.."
        assert np.array_equal(c1[b0:b1].cpu().numpy(), ec), f"col ids differ from the oracle in reads {lo}.."
        assert 0.05 < float((ep == 0).mean()) < 0.7
        checked += b1 - b0
    assert checked >= min(50_000_000, nb * 0.9)
    tbl.close()
    _free_hbm()
    tbl3 = pkg.ColPml.from_bytes(c2_image, layout=3)
    assert tbl3.info().layout == 3
    p3, c3, ms3 = run(tbl3)
    assert torch.equal(p1, p3) and torch.equal(c1, c3), "three-step rows and line rows disagree"
    print(f"  three-step rows {ms3:.1f} ms = {nb / ms3 / 1e6:.1f} Gbase/s")
    tbl3.close()


def _oracle_sample_check(pkg, oracle_index, tbl, n_reads, m, seed, sample):
    """Device-sampled backward-walk reads through the device entry point; the first `sample`
    reads against the oracle; two runs identical."""
    import torch
    dev = torch.device("cuda", 0)
    d_bases = torch.zeros(n_reads * m + 128, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    tbl.synth_reads_device(n_reads, m, 10, seed, d_bases.data_ptr(), d_off.data_ptr())
    outs = []
    for _ in range(2):
        p = torch.zeros(n_reads * m + 16, dtype=torch.int16, device=dev)
        c = torch.zeros(n_reads * m + 16, dtype=torch.uint8, device=dev)
        tbl.query_device(d_bases.data_ptr(), d_off.data_ptr(), n_reads, n_reads * m, p.data_ptr(), c.data_ptr())
        torch.cuda.synchronize()
        outs.append((p, c))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    hb = d_bases[:sample * m].cpu().numpy()
    ho = np.arange(sample + 1, dtype=np.uint64) * np.uint64(m)
    ep, ec = oracle_index.query_batch(hb, ho, threads=16)
    assert np.array_equal(outs[0][0][:sample * m].cpu().numpy().view(np.uint16), ep)
    assert np.array_equal(outs[0][1][:sample * m].cpu().numpy(), ec)
    return float((ep == 0).mean())


def test_hbm_budget_fallback_and_failed_open_leaves_nothing(pkg, oracle, c2_image):
    """The AUTO fallback (capi.hip) on the C2 index under COLBWT_HBM_BUDGET_MB: a three-step open
    beyond the budget fails with COLBWT_ERR_NOMEM and gives every byte back (the refinement's
    temporaries used to leak exactly here); AUTO then lands on the deepest layout that fits and
    still answers bit-exactly.  Without a budget the K-step index no longer keeps the one-step
    tables (6.4 GB at C2 scale)."""
    base = _free_hbm()
    ref = oracle.OracleIndex(c2_image)
    scale = C2_ROWS / 200_000_000
    try:
        os.environ["COLBWT_HBM_BUDGET_MB"] = str(int(24_000 * scale))
        with pytest.raises(pkg.ColbwtError) as ei:
            pkg.ColPml.from_bytes(c2_image, layout=3)
        assert ei.value.code == -6, ei.value
        after_fail = _free_hbm()
        assert base - after_fail < 64 << 20, f"failed open left {base - after_fail} bytes of HBM allocated"
        tbl = pkg.ColPml.from_bytes(c2_image, layout=0)
        assert tbl.info().layout == 2, tbl.info().layout
        assert tbl.info().device_bytes < 24_000 * scale * (1 << 20)
        _oracle_sample_check(pkg, ref, tbl, 200_000, 150, 7, 20_000)
        tbl.close()
        os.environ["COLBWT_HBM_BUDGET_MB"] = str(int(12_000 * scale))
        tbl = pkg.ColPml.from_bytes(c2_image, layout=0)
        assert tbl.info().layout == 1
        _oracle_sample_check(pkg, ref, tbl, 200_000, 150, 8, 20_000)
        tbl.close()
    finally:
        os.environ.pop("COLBWT_HBM_BUDGET_MB", None)
    assert base - _free_hbm() < 64 << 20
    tbl = pkg.ColPml.from_bytes(c2_image, layout=3)
    info = tbl.info()
    if C2_ROWS == 200_000_000:
        assert 24e9 < info.device_bytes < 27e9, info.device_bytes     # 31.9 GB with the one-step tables kept
    tbl.close()
    assert base - _free_hbm() < 64 << 20
    tbl = pkg.ColPml.from_bytes(c2_image, layout=0)                  # no budget: AUTO = line rows + deep mismatch lines, K = 8
    info = tbl.info()
    assert info.layout == (6 if C2_ROWS == 200_000_000 else info.layout) and info.layout in (5, 6) and info.layout_shape >> 8 == 8
    if C2_ROWS == 200_000_000:
        assert 0.9e9 < info.table_rows < 1.2e9 and 215e9 < info.device_bytes < 240e9, (info.table_rows, info.device_bytes)
    _oracle_sample_check(pkg, ref, tbl, 200_000, 150, 9, 20_000)
    tbl.close()
    assert base - _free_hbm() < 64 << 20
    # the ladder between the two (capi.hip): plain mismatch lines, line rows without them at K = 8, then shallower
    if C2_ROWS == 200_000_000:
        try:
            for budget_gb, layout_want, want in ((215, 5, (8,)), (185, 4, (8,)), (135, 4, (6, 4))):   # 215: no room for deep entries
                os.environ["COLBWT_HBM_BUDGET_MB"] = str(budget_gb * 1000)
                tbl = pkg.ColPml.from_bytes(c2_image, layout=0)
                info = tbl.info()
                assert info.layout == layout_want and info.layout_shape >> 8 in want, (budget_gb, info.layout, info.layout_shape >> 8)
                assert info.device_bytes < budget_gb * 1000 * (1 << 20)
                _oracle_sample_check(pkg, ref, tbl, 100_000, 150, 10, 10_000)
                tbl.close()
        finally:
            os.environ.pop("COLBWT_HBM_BUDGET_MB", None)
        assert base - _free_hbm() < 64 << 20


def test_c5_capacity_auto_layout_1e9_rows(pkg, oracle):
    """BASELINE configs[4] (capacity): a 1e9-row index with sub-run splits on 10 % of the rows
    (the density `col_split -m tunnels -s 10` leaves) opened with AUTO: three-step rows, about
    130 GB resident after a ~215 GB build peak; 64-bit indexing (n > 2^32); oracle on a sample."""
    base = _free_hbm()
    image = pkg.synth_index(BIG_ROWS, mean_len=8, split_permille=100, seed=45)
    tbl = pkg.ColPml.from_bytes(image, layout=0)
    info = tbl.info()
    assert info.r == BIG_ROWS and info.bwt_r < info.r and info.layout == 3
    if BIG_ROWS == 1_000_000_000:
        assert info.n > 2**32
        assert 2.2e9 < info.table_rows < 3.0e9
        assert 100e9 < info.device_bytes < 160e9, info.device_bytes
    ref = oracle.OracleIndex(image)
    resets = _oracle_sample_check(pkg, ref, tbl, 2_000_000, 150, 46, 30_000)
    assert 0.02 < resets < 0.6
    tbl.close()
    del ref, image
    assert base - _free_hbm() < 64 << 20


def test_c5_refined_row_limit_falls_back_to_two_step(pkg, oracle):
    """1.7e9 rows: the three-step refinement would need more than 2^32-2 rows (row numbers are 32
    bits, RUN_BYTES = 4), which the build reports as COLBWT_ERR_NOMEM after the counting pass;
    AUTO settles for two-step rows; an explicit three-step open fails cleanly."""
    if LIMIT_ROWS < 1_650_000_000:
        pytest.skip("needs >= 1.65e9 rows to pass the 2^32-2 refined-row limit")
    base = _free_hbm()
    image = pkg.synth_index(LIMIT_ROWS, mean_len=8, split_permille=0, seed=47)
    with pytest.raises(pkg.ColbwtError) as ei:
        pkg.ColPml.from_bytes(image, layout=3)
    assert ei.value.code == -6 and "2^32-2" in str(ei.value), ei.value
    assert base - _free_hbm() < 64 << 20
    tbl = pkg.ColPml.from_bytes(image, layout=0)
    info = tbl.info()
    assert info.layout == 2 and 3.0e9 < info.table_rows < 2**32 - 2
    ref = oracle.OracleIndex(image)
    _oracle_sample_check(pkg, ref, tbl, 1_000_000, 150, 48, 20_000)
    tbl.close()
    del ref, image
    assert base - _free_hbm() < 64 << 20
