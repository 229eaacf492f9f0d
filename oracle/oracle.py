"""ctypes binding of the CPU oracle (oracle/colbwt_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package.  Parity pinned by
the SURVEY.md Appendix D known-answer vector only (see colbwt_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcolbwt_oracle.so")


def build(force=False):
    """Compile the C restatement (gcc); idempotent."""
    srcs = [os.path.join(_HERE, f) for f in ("colbwt_oracle.c", "colsplit_oracle.c", "colbwt_oracle.h", "colsplit_oracle.h")]
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs)):
        subprocess.check_call(["make", "-C", _HERE, "libcolbwt_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Index(C.Structure):
    _fields_ = [("bwt_r", C.c_uint64), ("n", C.c_uint64), ("r", C.c_uint64),
                ("size", C.c_uint64), ("rows", C.c_void_p), ("owned", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.oracle_load_file.argtypes = [C.c_char_p, C.POINTER(_Index)]
        L.oracle_load_file.restype = C.c_int
        L.oracle_load_memory.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(_Index)]
        L.oracle_load_memory.restype = C.c_int
        L.oracle_free.argtypes = [C.POINTER(_Index)]
        L.oracle_query_pml.argtypes = [C.POINTER(_Index), C.c_void_p, C.c_uint64,
                                       C.c_void_p, C.c_void_p]
        for nm in ("oracle_query_batch_u16", "oracle_query_batch_u32"):
            getattr(L, nm).argtypes = [C.POINTER(_Index), C.c_void_p, C.c_void_p, C.c_uint64,
                                       C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_pml_query_files.argtypes = [C.POINTER(_Index), C.c_char_p, C.c_char_p, C.c_char_p]
        L.oracle_pml_query_files.restype = C.c_int
        L.oracle_get_length.argtypes = [C.POINTER(_Index), C.c_uint64]
        L.oracle_get_length.restype = C.c_uint64
        _lib = L
    return _lib


class OracleIndex:
    """A loaded `.col_pml` (col_pml::load, col_bwt.hpp:375-380)."""

    def __init__(self, source):
        self._x = _Index()
        self._keep = None
        if isinstance(source, (str, os.PathLike)):
            rc = lib().oracle_load_file(os.fsencode(source), C.byref(self._x))
        else:
            arr = np.ascontiguousarray(np.frombuffer(source, dtype=np.uint8))
            self._keep = arr
            rc = lib().oracle_load_memory(arr.ctypes.data, arr.size, C.byref(self._x))
        if rc != 0:
            raise ValueError(f"oracle: cannot load index (rc={rc})")

    n = property(lambda s: s._x.n)
    r = property(lambda s: s._x.r)
    bwt_r = property(lambda s: s._x.bwt_r)

    def query_pml(self, pattern: bytes):
        """col_pml::query_pml (col_bwt.hpp:409): returns (pml, cid) as uint64 arrays."""
        m = len(pattern)
        p = np.frombuffer(bytes(pattern), dtype=np.uint8) if m else np.zeros(0, np.uint8)
        p = np.ascontiguousarray(p)
        pml = np.zeros(m, np.uint64)
        cid = np.zeros(m, np.uint64)
        lib().oracle_query_pml(C.byref(self._x), p.ctypes.data, m, pml.ctypes.data, cid.ctypes.data)
        return pml, cid

    def query_batch(self, bases: np.ndarray, read_off: np.ndarray, wide=False, threads=1):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
        n_reads = read_off.size - 1
        total = int(read_off[-1])
        pml = np.zeros(total, np.uint32 if wide else np.uint16)
        cid = np.zeros(total, np.uint8)
        fn = lib().oracle_query_batch_u32 if wide else lib().oracle_query_batch_u16
        fn(C.byref(self._x), bases.ctypes.data, read_off.ctypes.data, n_reads,
           pml.ctypes.data, cid.ctypes.data, int(threads))
        return pml, cid

    def pml_query_files(self, pattern_path, pml_path=None, cid_path=None):
        """pml_query vec mode (pml_query.cpp:92-143): writes <pattern>.pml/.cid."""
        pml_path = pml_path or str(pattern_path) + ".pml"
        cid_path = cid_path or str(pattern_path) + ".cid"
        rc = lib().oracle_pml_query_files(C.byref(self._x), os.fsencode(pattern_path),
                                          os.fsencode(pml_path), os.fsencode(cid_path))
        if rc != 0:
            raise OSError(f"oracle_pml_query_files rc={rc}")
        return pml_path, cid_path

    def close(self):
        lib().oracle_free(C.byref(self._x))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def build_col_pml(heads, lens, col_ids, split_pos, thr_pos):
    """Reference constructor restatement (col_bwt.hpp:124-230 etc.) -> image bytes."""
    L = lib()
    L.oracle_build_col_pml.restype = C.c_uint64
    L.oracle_build_col_pml.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                       C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    heads = np.ascontiguousarray(heads, np.uint8)
    lens = np.ascontiguousarray(lens, np.uint64)
    col_ids = np.ascontiguousarray(col_ids, np.uint8)
    split_pos = np.ascontiguousarray(split_pos, np.uint64)
    thr_pos = np.ascontiguousarray(thr_pos, np.uint64)
    cap = 32 + 18 * (heads.size + split_pos.size + 1)
    out = np.zeros(cap, np.uint8)
    n = L.oracle_build_col_pml(heads.ctypes.data, heads.size, lens.ctypes.data, col_ids.ctypes.data, col_ids.size,
                               split_pos.ctypes.data, split_pos.size, thr_pos.ctypes.data, thr_pos.size,
                               out.ctypes.data, cap)
    return out[:n].copy()


class _FL(C.Structure):
    _fields_ = [("n", C.c_uint64), ("r", C.c_uint64), ("ch", C.c_void_p), ("idx", C.c_void_p),
                ("interval", C.c_void_p), ("offset", C.c_void_p), ("L_head", C.c_void_p)]


def col_split(heads, lens, mum_len, mum_pos, num_docs, mode, split_rate=1):
    """build_FL + col_split (src/build_FL.cpp, src/col_split.cpp; colsplit_oracle.c): returns
    (split_positions uint64 ascending, col_ids uint8 as .col_ids would hold them, n, stats) with
    stats = (col id runs, total runs, col chars) as the reference prints them."""
    L = lib()
    L.oracle_fl_build.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(_FL)]
    L.oracle_fl_free.argtypes = [C.POINTER(_FL)]
    L.oracle_col_split.argtypes = [C.POINTER(_FL), C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p]
    heads = np.ascontiguousarray(heads, np.uint8)
    lens = np.ascontiguousarray(lens, np.uint64)
    mum_len = np.ascontiguousarray(mum_len, np.uint64)
    mum_pos = np.ascontiguousarray(mum_pos, np.uint64)
    t = _FL()
    if L.oracle_fl_build(heads.ctypes.data, heads.size, lens.ctypes.data, C.byref(t)) != 0:
        raise ValueError("oracle_fl_build failed")
    n = int(t.n)
    bits = np.zeros((n + 63) // 64 + 1, np.uint64)
    cap = n + 8
    ids = np.zeros(cap, np.uint8)
    n_ids = C.c_uint64(0)
    stats = np.zeros(3, np.uint64)
    L.oracle_col_split(C.byref(t), mum_len.ctypes.data, mum_pos.ctypes.data, mum_len.size, int(num_docs),
                       1 if mode == "all" else 0, int(split_rate), bits.ctypes.data, ids.ctypes.data, cap,
                       C.byref(n_ids), stats.ctypes.data)
    L.oracle_fl_free(C.byref(t))
    pos = np.flatnonzero(np.unpackbits(bits.view(np.uint8), bitorder="little")[:n]).astype(np.uint64)
    return pos, ids[:n_ids.value].copy(), n, tuple(int(x) for x in stats)
