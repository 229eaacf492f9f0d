"""Independent pure-Python restatement of SURVEY.md Appendix B.2 (written from
the spec, not from the C oracle) -- used only to cross-check the C oracle on
small cases (tests/test_oracle_cross.py)."""
from helpers import unpack_col_pml


def query_pml(image, pattern):
    t = unpack_col_pml(image)
    n, r = int(t["n"]), int(t["r"])
    ch, idx = t["char"].tolist(), [int(v) for v in t["idx"]]
    itv, off = [int(v) for v in t["interval"]], [int(v) for v in t["offset"]]
    cid, thr = t["cid"].tolist(), [int(v) for v in t["thr"]]

    def ln(i):
        return (n if i == r - 1 else idx[i + 1]) - idx[i]

    m = len(pattern)
    pml, cids = [0] * m, [0] * m
    i, o, pos, L = r - 1, ln(r - 1) - 1, n - 1, 0
    for k in range(m - 1, -1, -1):
        c = pattern[k]
        cids[k] = cid[i]
        if ch[i] == c:
            L += 1
        else:
            L = 0
            ni, no, th = i, o, n
            s = next((j for j in range(i, r) if ch[j] == c), None)
            if s is not None:
                ni, no, th = s, 0, thr[s]
            if pos < th:
                q = next((j for j in range(i, -1, -1) if ch[j] == c), None)
                if q is not None:
                    ni, no = q, ln(q) - 1
            i, o = ni, no
        pml[k] = L
        j, tt = itv[i], off[i] + o
        while tt >= ln(j):
            tt -= ln(j)
            j += 1
        i, o, pos = j, tt, idx[j] + tt
    return pml, cids
