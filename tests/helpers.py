"""Test input builders (numpy): `.col_pml` images and reads.

Format: SURVEY.md Appendix A (col_bwt.hpp:360-370, LF_table.hpp:325-342).
These are INPUT generators; expected outputs always come from the oracle
(oracle/) or the committed golden vectors (tests/golden/).
"""
import struct

import numpy as np

ROW = 18
HDR = 32


def pack_col_pml(bwt_r, n, chars, idx, interval, offset, cid, thr):
    """Raw memory image of vector<col_thr> behind the 4xu64 header."""
    r = len(chars)
    rows = np.zeros((r, ROW), np.uint8)
    rows[:, 0] = np.asarray(chars, np.uint8)
    idx = np.asarray(idx, np.uint64)
    thr = np.asarray(thr, np.uint64)
    interval = np.asarray(interval, np.uint64)
    offset = np.asarray(offset, np.uint64)
    for b in range(5):
        rows[:, 1 + b] = (idx >> np.uint64(8 * b)) & np.uint64(0xFF)
        rows[:, 13 + b] = (thr >> np.uint64(8 * b)) & np.uint64(0xFF)
    for b in range(4):
        rows[:, 6 + b] = (interval >> np.uint64(8 * b)) & np.uint64(0xFF)
    for b in range(2):
        rows[:, 10 + b] = (offset >> np.uint64(8 * b)) & np.uint64(0xFF)
    rows[:, 12] = np.asarray(cid, np.uint8)
    return struct.pack("<4Q", bwt_r, n, r, r) + rows.tobytes()


def unpack_col_pml(image):
    """-> dict of numpy arrays (char, idx, interval, offset, cid, thr) + header."""
    image = bytes(image)
    bwt_r, n, r, size = struct.unpack_from("<4Q", image, 0)
    rows = np.frombuffer(image, np.uint8, size * ROW, HDR).reshape(size, ROW).astype(np.uint64)

    def le(lo, nb):
        v = np.zeros(size, np.uint64)
        for b in range(nb):
            v |= rows[:, lo + b] << np.uint64(8 * b)
        return v
    return dict(bwt_r=bwt_r, n=n, r=r, char=rows[:, 0].astype(np.uint8), idx=le(1, 5), interval=le(6, 4),
                offset=le(10, 2), cid=rows[:, 12].astype(np.uint8), thr=le(13, 5))


def lf_columns(chars, idx, n):
    """(interval, offset) of every row: LF_table::compute_table (LF_table.hpp:365-387):
    rows sorted stably by character tile F; a row's F start lands in row
    `interval` at `offset`."""
    chars = np.asarray(chars, np.uint8)
    idx = np.asarray(idx, np.int64)
    r = len(chars)
    lens = np.diff(np.append(idx, n))
    order = np.argsort(chars, kind="stable")
    fstart = np.zeros(r, np.int64)
    fstart[order] = np.concatenate(([0], np.cumsum(lens[order])[:-1]))
    interval = np.searchsorted(idx, fstart, side="right") - 1
    offset = fstart - idx[interval]
    return interval.astype(np.uint64), offset.astype(np.uint64)


def _index_from_bwt_lcp(bwt, lcp, rng, extra_splits, ids):
    """Run heads, thresholds = position of the minimum LCP between consecutive runs of a
    character (0 for a character's first run), random sub-run splits and ids -> image bytes."""
    n = len(bwt)
    heads = np.flatnonzero(np.concatenate(([True], bwt[1:] != bwt[:-1])))
    bwt_r = len(heads)
    ends = np.append(heads[1:], n) - 1                       # last position of every run
    run_thr = np.zeros(bwt_r, np.int64)
    key = lcp.astype(np.int64) * n + np.arange(n)            # min key = (min lcp, first position)
    head_chars = bwt[heads]
    for c in np.unique(head_chars):
        runs = np.flatnonzero(head_chars == c)
        if len(runs) < 2:
            continue
        lo = ends[runs[:-1]] + 1                             # first position after the previous c-run
        hi = heads[runs[1:]] + 1                             # one past this run's head
        cuts = np.empty(2 * len(lo), np.int64)
        cuts[0::2], cuts[1::2] = lo, hi
        keep = cuts < n
        mins = np.minimum.reduceat(key, cuts[keep])
        run_thr[runs[1:]] = (mins[0::2] % n)[:len(lo)]
    split = np.zeros(n, bool)
    split[heads] = True
    cand = np.flatnonzero(~split)
    if len(cand) and extra_splits:
        split[rng.choice(cand, size=min(extra_splits, len(cand)), replace=False)] = True
    starts = np.flatnonzero(split).astype(np.int64)
    chars = bwt[starts]
    run_of = np.searchsorted(heads, starts, side="right") - 1
    thr = run_thr[run_of]
    cid = rng.choice(np.array(ids, np.uint8), size=len(starts))
    interval, offset = lf_columns(chars, starts, n)
    return pack_col_pml(bwt_r, n, chars, starts, interval, offset, cid, thr)


def true_bwt_index(seqs, seed=0, extra_splits=20, ids=(0, 0, 0, 1, 2, 3, 17, 200, 255), term=1):
    """A real BWT index of concat(seqs)+terminator: suffix array by sorting,
    run heads, thresholds = min-LCP position between consecutive same-char
    runs (0 for the first run of a character), random sub-run splits and ids.
    Returns (image_bytes, text_bytes).  Quadratic: small texts only."""
    rng = np.random.default_rng(seed)
    text = b"".join(seqs) + bytes([term])
    n = len(text)
    sa = sorted(range(n), key=lambda i: text[i:])
    bwt = np.frombuffer(bytes(text[i - 1] for i in sa), np.uint8)
    lcp = np.zeros(n, np.int64)
    for k in range(1, n):
        a, b = text[sa[k - 1]:], text[sa[k]:]
        l = 0
        while l < len(a) and l < len(b) and a[l] == b[l]:
            l += 1
        lcp[k] = l
    return _index_from_bwt_lcp(bwt, lcp, rng, extra_splits, ids), text


def true_bwt_index_large(seqs, seed=0, extra_splits=20, ids=(0, 0, 0, 1, 2, 3, 17, 200, 255), term=1):
    """The same index as true_bwt_index for texts of millions of characters (BASELINE.json's
    C1 shape): suffix array by prefix doubling, LCP by binary lifting over the rank arrays of
    the doubling rounds, all in numpy.  The terminator must be the smallest byte and unique."""
    rng = np.random.default_rng(seed)
    text = b"".join(seqs) + bytes([term])
    t = np.frombuffer(text, np.uint8)
    n = len(t)
    assert (t[:-1] > term).all()
    rank = t.astype(np.int64)
    ranks = []                                   # ranks[j] orders suffixes by their first 2^j characters
    k = 1
    while True:
        ranks.append(rank)
        if rank.max() == n - 1 and len(np.unique(rank)) == n:
            break
        nxt = np.full(n, -1, np.int64)           # beyond the end sorts first (cannot tie: unique terminator)
        nxt[:n - k] = rank[k:]
        order = np.lexsort((nxt, rank))
        r_s, n_s = rank[order], nxt[order]
        new = np.concatenate(([0], np.cumsum((r_s[1:] != r_s[:-1]) | (n_s[1:] != n_s[:-1]))))
        rank = np.empty(n, np.int64)
        rank[order] = new
        k *= 2
    sa = np.empty(n, np.int64)
    sa[ranks[-1]] = np.arange(n)
    bwt = t[(sa - 1) % n]
    a, b = sa[:-1], sa[1:]
    l = np.zeros(n - 1, np.int64)
    for j in range(len(ranks) - 2, -1, -1):      # ranks[-1] is unique; level j compares 2^j characters
        ia, ib = a + l, b + l
        ok = (ia < n) & (ib < n)
        same = np.zeros(n - 1, bool)
        same[ok] = ranks[j][ia[ok]] == ranks[j][ib[ok]]
        l += same * (1 << j)
    lcp = np.concatenate(([0], l))
    return _index_from_bwt_lcp(bwt, lcp, rng, extra_splits, ids), text


def random_table(rng, r, alphabet=b"ACGT", max_len=9, split_prob=0.1, thr_inside=True):
    """A random but self-consistent move table over `alphabet` (any sigma):
    random run characters (repeats allowed = sub-run splits), lengths, ids;
    thresholds shared by the sub-runs of a BWT run, some placed inside nearby
    rows so the `pos < thr` comparison depends on the offset."""
    alpha = np.frombuffer(alphabet, np.uint8)
    chars = rng.choice(alpha, size=r)
    rep = rng.random(r) < split_prob
    for k in range(1, r):
        if rep[k]:
            chars[k] = chars[k - 1]
    lens = rng.integers(1, max_len + 1, size=r)
    idx = np.concatenate(([0], np.cumsum(lens)[:-1]))
    n = int(lens.sum())
    interval, offset = lf_columns(chars, idx, n)
    thr = rng.integers(0, n, size=r)
    if thr_inside:
        near = rng.random(r) < 0.5
        back = rng.integers(0, 6, size=r)
        j = np.maximum(np.arange(r) - back, 0)
        thr = np.where(near, idx[j] + rng.integers(0, max_len, size=r) % lens[j], thr)
    heads = np.concatenate(([True], chars[1:] != chars[:-1]))
    run_of = np.cumsum(heads) - 1
    thr = thr[np.flatnonzero(heads)][run_of]
    cid = rng.integers(0, 256, size=r)
    return pack_col_pml(int(heads.sum()), n, chars, idx, interval, offset, cid, thr)


def reads_from_text(text, n_reads, read_len, sub_rate, seed, alphabet=b"ACGT", extra=b""):
    """Substrings of `text` with substitutions; `extra` bytes (e.g. b"Nacgt")
    are sprinkled in to exercise absent characters / no case folding."""
    rng = np.random.default_rng(seed)
    body = np.frombuffer(text, np.uint8)
    out = []
    for _ in range(n_reads):
        m = int(read_len if np.isscalar(read_len) else rng.integers(read_len[0], read_len[1] + 1))
        m = min(m, len(body) - 1)
        s = int(rng.integers(0, len(body) - m)) if len(body) - m > 0 else 0
        rd = body[s:s + m].copy()
        rd[rd <= 1] = ord("A")
        mut = rng.random(m) < sub_rate
        rd[mut] = rng.choice(np.frombuffer(alphabet, np.uint8), size=int(mut.sum()))
        if extra:
            ex = rng.random(m) < 0.02
            rd[ex] = rng.choice(np.frombuffer(extra, np.uint8), size=int(ex.sum()))
        out.append(rd)
    return out


def concat_reads(reads):
    off = np.zeros(len(reads) + 1, np.uint64)
    if reads:
        off[1:] = np.cumsum([len(r) for r in reads])
        bases = np.concatenate(reads).astype(np.uint8) if off[-1] else np.zeros(0, np.uint8)
    else:
        bases = np.zeros(0, np.uint8)
    return bases, off


def backward_walk_reads(image, n_reads, read_len, sub_rate, seed):
    """SURVEY.md 8(d) read recipe on any index image, vectorised over reads:
    read[m-1-k] = char at LF^k(p0), p0 uniform; 0x01 -> 'A'; substitutions."""
    t = unpack_col_pml(image)
    rng = np.random.default_rng(seed)
    idx = t["idx"].astype(np.int64)
    n, r = int(t["n"]), int(t["r"])
    idx_ext = np.append(idx, n)
    p0 = rng.integers(0, n, size=n_reads)
    i = np.searchsorted(idx, p0, side="right") - 1
    o = p0 - idx[i]
    out = np.zeros((n_reads, read_len), np.uint8)
    interval = t["interval"].astype(np.int64)
    offset = t["offset"].astype(np.int64)
    for k in range(read_len):
        ch = t["char"][i].copy()
        ch[ch <= 1] = ord("A")
        out[:, read_len - 1 - k] = ch
        j = interval[i]
        tt = offset[i] + o
        while True:
            ln = idx_ext[j + 1] - idx_ext[j]
            ff = (tt >= ln) & (j < r - 1)
            if not ff.any():
                break
            tt = np.where(ff, tt - ln, tt)
            j = np.where(ff, j + 1, j)
        i, o = j, tt
    mut = rng.random(out.shape) < sub_rate
    out[mut] = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(mut.sum()))
    return [row for row in out]


def write_fasta(path, reads, names=None, width=60):
    with open(path, "wb") as f:
        for k, rd in enumerate(reads):
            nm = names[k] if names else f"r{k}"
            f.write(b">" + nm.encode() + b"\n")
            b = bytes(rd)
            for s in range(0, len(b), width):
                f.write(b[s:s + width] + b"\n")
            if not b:
                f.write(b"\n")
