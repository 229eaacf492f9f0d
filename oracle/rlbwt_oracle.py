"""CPU restatement of the front of the pipeline: documents -> RLBWT, thresholds, multi-MUMs
(SURVEY.md 8(f) "next" #4; what the reference's driver takes from `mumemto mum -K -R -T`,
scripts/col-bwt.py:121-145, in the byte formats of SURVEY.md Appendix A: col_bwt.hpp:167-171,
446-448; col_split.cpp:90-106).

TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product package.

PARITY UNPINNED: mumemto is an un-vendored dependency (thirdparty/CMakeLists.txt, pinned by branch
name only), the reference tree holds neither its source nor any file it wrote, and the reference's
own code only consumes these files.  What is restated here is therefore the published meaning of
the files (BWT of the concatenated documents, min-LCP thresholds as col_bwt.hpp:531-574 uses them,
multi-MUM = maximal match occurring exactly once in every document) with this repository's
conventions for what the formats leave open (include/colbwt.h): separators 1, a final 0, byte-wise
suffix order, runs of the folded characters (bytes <= 1 are one class), LCPs cut at separators,
first minimum for ties.  Everything is written for clarity, independently of the
product's algorithms (plain sorting, Kasai, direct character comparison); `brute_force_mums`
does not use a suffix array at all.
"""
import numpy as np

COMPLEMENT = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")


def build_text(docs, revcomp=False):
    """docs: list of documents, each a list of records (bytes).  -> (text, doc_start)."""
    out, starts = bytearray(), []
    for records in docs:
        starts.append(len(out))
        for rec in records:
            assert all(b > 1 for b in rec)
            out += rec + b"\x01"
            if revcomp:
                out += rec.translate(COMPLEMENT)[::-1] + b"\x01"
    out += b"\x00"
    return bytes(out), starts


def suffix_array(text):
    """Plain sorting for short texts, prefix doubling (numpy) for long ones."""
    n = len(text)
    if n <= 4000:
        return sorted(range(n), key=lambda i: text[i:])
    t = np.frombuffer(text, np.uint8)
    rank = t.astype(np.int64)
    k = 1
    while True:
        nxt = np.full(n, -1, np.int64)
        nxt[:n - k] = rank[k:]
        order = np.lexsort((nxt, rank))
        r_s, n_s = rank[order], nxt[order]
        new = np.concatenate(([0], np.cumsum((r_s[1:] != r_s[:-1]) | (n_s[1:] != n_s[:-1]))))
        rank = np.empty(n, np.int64)
        rank[order] = new
        if new[-1] == n - 1:
            return order.tolist()
        k *= 2


def lcp_array(text, sa):
    """Kasai et al.: lcp[k] = common prefix of suffixes sa[k-1], sa[k]; lcp[0] = 0."""
    n = len(text)
    isa = [0] * n
    for k, i in enumerate(sa):
        isa[i] = k
    lcp = [0] * n
    l = 0
    for i in range(n):
        k = isa[i]
        if k == 0:
            l = 0
            continue
        j = sa[k - 1]
        while i + l < n and j + l < n and text[i + l] == text[j + l]:
            l += 1
        lcp[k] = l
        if l:
            l -= 1
    return lcp


def rlbwt(text, sa):
    """Runs of the characters as the consumers see them: bytes <= 1 (the final 0, the separators)
    are one class, the terminator 1 (col_bwt.hpp:167-171, FL_table.hpp:102-104), so that the k-th
    run is the k-th group read_thresholds (col_bwt.hpp:446-451) gives a threshold to."""
    bwt = bytes(text[i - 1] for i in sa)            # text[-1] for i == 0: the final 0
    heads, lens = [], []
    for c in bwt:
        c = max(c, 1)
        if heads and heads[-1] == c:
            lens[-1] += 1
        else:
            heads.append(c)
            lens.append(1)
    return bwt, heads, lens


def capped_lcp(text, sa):
    """LCP of neighbouring suffixes cut at the first separator: what a pattern can tell apart."""
    return [0] + [_match_len(text, sa[k - 1], sa[k]) for k in range(1, len(text))]


def thresholds(heads, lens, lcp):
    """Per run: first position of the minimum (separator-capped) LCP in (end of the previous run
    of its character, its head]; 0 for a character's first run."""
    start, thr, last_end = 0, [], {}
    for c, ln in zip(heads, lens):
        if c in last_end:
            lo = last_end[c] + 1
            best = lo
            for k in range(lo, start + 1):
                if lcp[k] < lcp[best]:
                    best = k
            thr.append(best)
        else:
            thr.append(0)
        start += ln
        last_end[c] = start - 1
    return thr


def _match_len(text, a, b):
    """Common prefix of two suffixes that holds no separator."""
    l = 0
    while text[a + l] > 1 and text[b + l] > 1 and text[a + l] == text[b + l]:
        l += 1
    return l


def _doc_of(doc_start, i):
    return int(np.searchsorted(doc_start, i, side="right")) - 1


def multi_mums(text, sa, doc_start, min_len):
    """[(length, suffix-array rank of the first suffix)], ascending by rank."""
    n, nd = len(text), len(doc_start)
    if nd < 2:
        return []
    adj = capped_lcp(text, sa)
    out = []
    for i in range(0, n - nd + 1):
        inner = min(adj[i + 1:i + nd])
        if inner < max(1, min_len) or adj[i] >= inner or (i + nd < n and adj[i + nd] >= inner):
            continue
        if len({_doc_of(doc_start, sa[k]) for k in range(i, i + nd)}) != nd:
            continue
        before = [text[sa[k] - 1] for k in range(i, i + nd)]      # sa[k] == 0: text[-1], the final 0
        if len(set(before)) == 1 and before[0] > 1:
            continue
        out.append((inner, i))
    return out


def brute_force_mums(text, doc_start, min_len):
    """The definition itself, no suffix array: every separator-free substring that occurs exactly
    len(doc_start) times, once per document, and cannot be extended to either side; reported as
    (length, number of suffixes smaller than its smallest occurrence)."""
    n, nd = len(text), len(doc_start)
    if nd < 2:
        return []
    seen, out = set(), []
    end0 = doc_start[1]
    for s in range(end0):
        for e in range(s + max(1, min_len), end0 + 1):
            w = text[s:e]
            if min(w) <= 1:
                break
            if w in seen:
                continue
            seen.add(w)
            occ = [i for i in range(n - len(w) + 1) if text[i:i + len(w)] == w]
            if len(occ) != nd or len({_doc_of(doc_start, i) for i in occ}) != nd:
                continue
            right = {text[i + len(w)] for i in occ}
            left = {text[i - 1] for i in occ}
            if (len(right) == 1 and min(right) > 1) or (len(left) == 1 and min(left) > 1):
                continue
            smallest = min(occ, key=lambda i: text[i:])
            rank = sum(1 for j in range(n) if text[j:] < text[smallest:])
            out.append((len(w), rank))
    return sorted(out, key=lambda m: m[1])


def build(docs, min_len=20, revcomp=False):
    """Everything at once -> dict(text, doc_start, sa, lcp, bwt, heads, lens, thr, mums)."""
    text, doc_start = build_text(docs, revcomp)
    sa = suffix_array(text)
    lcp = lcp_array(text, sa)
    bwt, heads, lens = rlbwt(text, sa)
    return dict(text=text, doc_start=doc_start, sa=sa, lcp=lcp, bwt=bwt, heads=heads, lens=lens,
                thr=thresholds(heads, lens, capped_lcp(text, sa)), mums=multi_mums(text, sa, doc_start, min_len))


def file_bytes(res, n_docs):
    """The four files' contents: heads, len, thr_pos, col_mums."""
    def le5(a):
        return b"".join(int(x).to_bytes(5, "little") for x in a)
    return (bytes(res["heads"]), le5(res["lens"]), le5(res["thr"]),
            le5([n_docs]) + b"".join(le5(m) for m in res["mums"]))
