"""colbwt_amd -- host-side Python mirror of the reference's query interface
(`col_pml`, include/col_bwt.hpp:386-575; `pml_query`, src/pml_query.cpp) over
the MI355X-native C-ABI library `libcolbwt.so` (include/colbwt.h).

This module is plumbing: ctypes over the C-ABI.  All computation happens in
the hand-written HIP kernels of `csrc/`.  There is no CPU fallback: if the
library is missing or no HIP device is usable, calls raise `ColbwtError`.

The directory is named `col-bwt_amd` (not importable by that name); load it
with `__graft_entry__.load_package()` which registers it as `colbwt_amd`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcolbwt.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "colbwt.h")

# every symbol include/colbwt.h declares (checked by tests/test_capi_symbols.py)
EXPORTS = (
    "colbwt_version", "colbwt_last_error", "colbwt_index_open", "colbwt_index_open_memory",
    "colbwt_index_open_layout", "colbwt_index_open_memory_layout",
    "colbwt_index_open_devices", "colbwt_index_open_memory_devices",
    "colbwt_index_close", "colbwt_index_info", "colbwt_query_batch", "colbwt_query_batch_u32",
    "colbwt_query_device", "colbwt_query_device_ordered", "colbwt_query_file", "colbwt_query_file_binary",
    "colbwt_binary_to_text", "colbwt_synth_index_bytes", "colbwt_synth_index", "colbwt_synth_index_thr", "colbwt_pml_pack_device", "colbwt_read_end_mask_device", "colbwt_pml_unpack_device",
    "colbwt_index_cid_dictionary", "colbwt_cid_code_bits", "colbwt_cid_pack_device", "colbwt_cid_unpack_device",
    "colbwt_synth_reads_device", "colbwt_build_col_pml", "colbwt_build_col_pml_arrays",
    "colbwt_col_split", "colbwt_col_split_arrays", "colbwt_col_split_error",
    "colbwt_rlbwt_build_text", "colbwt_rlbwt_build_files", "colbwt_rlbwt_get", "colbwt_rlbwt_free", "colbwt_rlbwt_error",
)


class ColbwtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"colbwt error {code}: {msg}")
        self.code = code


class Info(C.Structure):
    _fields_ = [("bwt_r", C.c_uint64), ("n", C.c_uint64), ("r", C.c_uint64), ("sigma", C.c_uint32),
                ("device", C.c_uint32), ("device_bytes", C.c_uint64), ("layout", C.c_uint32),
                ("layout_shape", C.c_uint32), ("table_rows", C.c_uint64), ("n_devices", C.c_uint32),
                ("reserved_", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_bases", C.c_uint64), ("h2d_ms", C.c_double),
                ("kernel_ms", C.c_double), ("d2h_ms", C.c_double), ("algorithmic_bytes", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def lib():
    """Loads libcolbwt.so; raises loudly when the HIP extension is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ColbwtError(-100, f"{LIB_PATH} not built: run __graft_entry__.build() "
                                "(the query path has no CPU fallback)")
    # One HIP runtime per process: PyTorch bundles its own libamdhip64 (same
    # soname as /opt/rocm's).  Importing torch first makes libcolbwt.so bind to
    # that copy; loading ours first would leave torch with a second runtime
    # that sees no GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    L.colbwt_version.restype = C.c_char_p
    L.colbwt_last_error.restype = C.c_char_p
    L.colbwt_index_open.argtypes = [C.c_char_p, vp, i32, C.POINTER(vp)]
    L.colbwt_index_open_memory.argtypes = [vp, u64, vp, i32, C.POINTER(vp)]
    L.colbwt_index_open_layout.argtypes = [C.c_char_p, vp, i32, i32, C.POINTER(vp)]
    L.colbwt_index_open_memory_layout.argtypes = [vp, u64, vp, i32, i32, C.POINTER(vp)]
    L.colbwt_index_open_devices.argtypes = [C.c_char_p, vp, C.POINTER(i32), i32, i32, C.POINTER(vp)]
    L.colbwt_index_open_memory_devices.argtypes = [vp, u64, vp, C.POINTER(i32), i32, i32, C.POINTER(vp)]
    L.colbwt_index_close.argtypes = [vp]
    L.colbwt_index_close.restype = None
    L.colbwt_index_info.argtypes = [vp, C.POINTER(Info)]
    L.colbwt_query_batch.argtypes = [vp, vp, vp, u64, vp, vp, C.POINTER(Stats)]
    L.colbwt_query_batch_u32.argtypes = [vp, vp, vp, u64, vp, vp, C.POINTER(Stats)]
    L.colbwt_query_device.argtypes = [vp, vp, vp, u64, u64, vp, i32, vp, vp, C.POINTER(Stats)]
    L.colbwt_query_device_ordered.argtypes = [vp, vp, vp, u64, u64, vp, i32, vp, vp, vp, C.POINTER(Stats)]
    L.colbwt_query_file.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, u64, C.POINTER(Stats)]
    L.colbwt_query_file_binary.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, u64, C.POINTER(Stats)]
    L.colbwt_binary_to_text.argtypes = [C.c_char_p, i32, C.c_char_p]
    L.colbwt_synth_index_bytes.argtypes = [u64]
    L.colbwt_synth_index_bytes.restype = u64
    L.colbwt_synth_index.argtypes = [u64, C.c_uint32, C.c_uint32, u64, vp, u64]
    L.colbwt_synth_index_thr.argtypes = [u64, C.c_uint32, C.c_uint32, u64, C.c_int, vp, u64]
    L.colbwt_pml_pack_device.argtypes = [vp, u64, vp, vp]
    L.colbwt_read_end_mask_device.argtypes = [vp, u64, vp, vp]
    L.colbwt_pml_unpack_device.argtypes = [vp, vp, u64, u64, u64, vp, vp]
    L.colbwt_index_cid_dictionary.argtypes = [vp, vp, C.POINTER(C.c_uint32)]
    L.colbwt_cid_code_bits.argtypes = [C.c_uint32]
    L.colbwt_cid_code_bits.restype = C.c_uint32
    L.colbwt_cid_pack_device.argtypes = [vp, u64, vp, C.c_uint32, vp, vp]
    L.colbwt_cid_unpack_device.argtypes = [vp, u64, u64, vp, C.c_uint32, vp, vp]
    L.colbwt_synth_reads_device.argtypes = [vp, u64, C.c_uint32, C.c_uint32, u64, vp, vp, vp]
    L.colbwt_build_col_pml.argtypes = [C.c_char_p, C.c_char_p]
    L.colbwt_build_col_pml_arrays.argtypes = [vp, u64, vp, vp, u64, vp, u64, vp, u64, vp, u64, C.POINTER(u64)]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise ColbwtError(rc, lib().colbwt_last_error().decode("utf-8", "replace"))


def version():
    return lib().colbwt_version().decode()


class ColPml:
    """The reference's `col_pml` (col_bwt.hpp:386): load an index, query reads.

    `ColPml.load(prefix)` <-> `col_pml tbl; tbl.load(ifstream(prefix + ".col_pml"))`
    (pml_query.cpp:109-112); `query_pml(pattern)` <-> col_bwt.hpp:403-412.
    """

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def load(cls, prefix_or_file, device=0, layout=0, devices=None):
        """layout: 0 = engine default, 1 = one-step rows, 2 / 3 = K-step rows, 4 = line rows, 5 = line rows with
        mismatch lines (same results)."""
        h = C.c_void_p()
        if devices is not None:
            dv = (C.c_int * len(devices))(*[int(d) for d in devices])
            _check(lib().colbwt_index_open_devices(os.fsencode(prefix_or_file), None, dv, len(devices), int(layout),
                                                   C.byref(h)))
        else:
            _check(lib().colbwt_index_open_layout(os.fsencode(prefix_or_file), None, int(device), int(layout),
                                                  C.byref(h)))
        return cls(h)

    @classmethod
    def from_bytes(cls, image, device=0, layout=0, devices=None):
        """devices: a list of device ordinals (one replica each; a device may repeat) instead of `device`."""
        arr = np.ascontiguousarray(np.frombuffer(image, dtype=np.uint8))
        h = C.c_void_p()
        if devices is not None:
            dv = (C.c_int * len(devices))(*[int(d) for d in devices])
            _check(lib().colbwt_index_open_memory_devices(arr.ctypes.data, arr.size, None, dv, len(devices), int(layout),
                                                          C.byref(h)))
        else:
            _check(lib().colbwt_index_open_memory_layout(arr.ctypes.data, arr.size, None, int(device), int(layout),
                                                         C.byref(h)))
        return cls(h)

    def info(self):
        out = Info()
        _check(lib().colbwt_index_info(self._h, C.byref(out)))
        return out

    # -- col_pml::query_pml(const char*, size_t), col_bwt.hpp:409-412 ---------
    def query_pml(self, pattern):
        """One read -> (pml, cid) uint64 arrays, index k <-> pattern[k]."""
        p = np.frombuffer(bytes(pattern), dtype=np.uint8)
        off = np.array([0, p.size], dtype=np.uint64)
        pml, cid, _ = self.query_batch(p, off, wide=p.size > 65535)
        return pml.astype(np.uint64), cid.astype(np.uint64)

    def query_batch(self, bases, read_off, wide=False):
        """Many reads at once (host buffers in, host buffers out)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
        n_reads = read_off.size - 1
        total = int(read_off[-1]) if n_reads >= 0 and read_off.size else 0
        pml = np.zeros(total, np.uint32 if wide else np.uint16)
        cid = np.zeros(total, np.uint8)
        st = Stats()
        fn = lib().colbwt_query_batch_u32 if wide else lib().colbwt_query_batch
        _check(fn(self._h, bases.ctypes.data, read_off.ctypes.data, max(n_reads, 0),
                  pml.ctypes.data, cid.ctypes.data, C.byref(st)))
        return pml, cid, st

    def query_device(self, d_bases, d_read_off, n_reads, n_bases, d_pml, d_cid, pml_bytes=2,
                     stream=0, timed=False, d_order=None):
        """Device-resident entry point: arguments are raw device pointers (ints).
        d_order: optional device array of read indices by decreasing length."""
        st = Stats()
        _check(lib().colbwt_query_device_ordered(self._h, d_bases, d_read_off, n_reads, n_bases, d_pml,
                                                 pml_bytes, d_cid, d_order, stream,
                                                 C.byref(st) if timed else None))
        return st

    # -- pml_query main, vec mode (pml_query.cpp:92-143) ----------------------
    def query_file(self, pattern_path, pml_path=None, cid_path=None, batch_bases=0):
        st = Stats()
        _check(lib().colbwt_query_file(self._h, os.fsencode(pattern_path),
                                       os.fsencode(pml_path) if pml_path else None,
                                       os.fsencode(cid_path) if cid_path else None,
                                       batch_bases, C.byref(st)))
        return st

    def query_file_binary(self, pattern_path, pml_bin_path=None, cid_bin_path=None, batch_bases=0):
        """`col-bwt query`'s binary outputs (<pattern>.pml.bin / .cid.bin; Movi-like container, unverified)."""
        st = Stats()
        _check(lib().colbwt_query_file_binary(self._h, os.fsencode(pattern_path),
                                              os.fsencode(pml_bin_path) if pml_bin_path else None,
                                              os.fsencode(cid_bin_path) if cid_bin_path else None,
                                              batch_bases, C.byref(st)))
        return st

    def cid_dictionary(self):
        """The distinct col ids the table's rows hold, ascending (uint8 array): the dictionary of the gather codec."""
        ids = np.zeros(256, np.uint8)
        n = C.c_uint32(0)
        _check(lib().colbwt_index_cid_dictionary(self._h, ids.ctypes.data, C.byref(n)))
        return ids[:n.value].copy()

    def synth_reads_device(self, n_reads, read_len, sub_permille, seed, d_bases, d_read_off, stream=0):
        _check(lib().colbwt_synth_reads_device(self._h, n_reads, read_len, sub_permille, seed,
                                               d_bases, d_read_off, stream))

    def close(self):
        if self._h:
            lib().colbwt_index_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def binary_to_text(bin_path, value_bytes, text_path):
    """Container -> reference text (`col-bwt view`); value_bytes 2 = .pml.bin, 1 = .cid.bin."""
    _check(lib().colbwt_binary_to_text(os.fsencode(bin_path), int(value_bytes), os.fsencode(text_path)))


def read_binary(bin_path, value_bytes):
    """Parses a container: [(name, values in pattern order as a numpy array)]."""
    raw = np.fromfile(bin_path, dtype=np.uint8)
    out, at = [], 0
    dt = np.uint16 if value_bytes == 2 else np.uint8
    while at < raw.size:
        nl = int(raw[at:at + 2].view(np.uint16)[0])
        name = raw[at + 2:at + 2 + nl].tobytes().decode()
        m = int(raw[at + 2 + nl:at + 10 + nl].view(np.uint64)[0])
        at += 10 + nl
        out.append((name, raw[at:at + m * value_bytes].view(dt)[::-1].copy()))
        at += m * value_bytes
    return out


def pml_pack_device(d_pml, n_bases, d_mask, stream=0):
    """Gather codec (include/colbwt.h): one bit per base, set where the PML value is 0."""
    _check(lib().colbwt_pml_pack_device(d_pml, n_bases, d_mask, stream))


def read_end_mask_device(d_read_off, n_reads, d_mask, stream=0):
    """Bit set at the last base of every non-empty read (d_mask zeroed by the caller)."""
    _check(lib().colbwt_read_end_mask_device(d_read_off, n_reads, d_mask, stream))


def cid_code_bits(n_ids):
    """Bits per base of the col-id codes of a dictionary of n_ids ids."""
    return int(lib().colbwt_cid_code_bits(int(n_ids)))


def cid_pack_device(d_cid, n_bases, ids, d_planes, stream=0):
    """Gather codec: col ids -> codes of the dictionary `ids` (uint8 array), cid_code_bits(len(ids)) bit planes per 32 bases."""
    ids = np.ascontiguousarray(ids, np.uint8)
    _check(lib().colbwt_cid_pack_device(d_cid, n_bases, ids.ctypes.data, ids.size, d_planes, stream))


def cid_unpack_device(d_planes, first_word, n_words, ids, d_cid, stream=0):
    """Rebuilds the col ids of 32-base words [first_word, first_word + n_words) from their bit planes."""
    ids = np.ascontiguousarray(ids, np.uint8)
    _check(lib().colbwt_cid_unpack_device(d_planes, first_word, n_words, ids.ctypes.data, ids.size, d_cid, stream))


def pml_unpack_device(d_zero_mask, d_end_mask, first_word, n_words, total_words, d_pml, stream=0):
    """Rebuilds the u16 PML values of 32-base words [first_word, first_word + n_words)."""
    _check(lib().colbwt_pml_unpack_device(d_zero_mask, d_end_mask, first_word, n_words, total_words, d_pml, stream))


def synth_index(rows, mean_len=8, split_permille=0, seed=42, thr_mode=0):
    """Synthetic `.col_pml` image (SURVEY.md 8(d) recipe) as a uint8 numpy array.
    thr_mode 0: thresholds uniform in [0, n); 1: between consecutive runs of a character."""
    nbytes = lib().colbwt_synth_index_bytes(rows)
    out = np.empty(nbytes, np.uint8)
    _check(lib().colbwt_synth_index_thr(rows, mean_len, split_permille, seed, thr_mode, out.ctypes.data, nbytes))
    return out


def build_col_pml(prefix, out_path=None):
    """`build_col_bwt <prefix>` (src/build_col_bwt.cpp:38-52): writes <prefix>.col_pml."""
    _check_build(lib().colbwt_build_col_pml(os.fsencode(prefix), os.fsencode(out_path) if out_path else None))


def build_col_pml_arrays(heads, lens, col_ids, split_pos, thr_pos):
    """col_pml(heads, lengths, col_ids, thresholds, splits) + serialize -> image bytes (uint8 array)."""
    heads = np.ascontiguousarray(heads, np.uint8)
    lens = np.ascontiguousarray(lens, np.uint64)
    col_ids = np.ascontiguousarray(col_ids, np.uint8)
    split_pos = np.ascontiguousarray(split_pos, np.uint64)
    thr_pos = np.ascontiguousarray(thr_pos, np.uint64)
    need = C.c_uint64(0)
    args = [heads.ctypes.data, heads.size, lens.ctypes.data, col_ids.ctypes.data, col_ids.size,
            split_pos.ctypes.data, split_pos.size, thr_pos.ctypes.data, thr_pos.size]
    lib().colbwt_build_col_pml_arrays(*args, None, 0, C.byref(need))
    out = np.zeros(need.value, np.uint8)
    _check_build(lib().colbwt_build_col_pml_arrays(*args, out.ctypes.data, out.size, C.byref(need)))
    return out


def col_split(prefix, mode="tunnels", split_rate=1, device=0):
    """`col_split <prefix> -m mode -s rate` (src/col_split.cpp:62-140, with build_FL folded in): writes
    <prefix>.col_runs and <prefix>.col_ids."""
    L = lib()
    L.colbwt_col_split_error.restype = C.c_char_p
    rc = L.colbwt_col_split(os.fsencode(prefix), 1 if mode == "all" else 0, int(split_rate), int(device))
    if rc != 0:
        raise ColbwtError(rc, L.colbwt_col_split_error().decode())


def col_split_arrays(heads, lens, mum_len, mum_pos, num_docs, mode="tunnels", split_rate=1, device=0):
    """-> (split positions uint64 ascending, col ids uint8, n): what .col_runs / .col_ids would hold."""
    L = lib()
    L.colbwt_col_split_error.restype = C.c_char_p
    L.colbwt_col_split_arrays.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                          C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p,
                                          C.POINTER(C.c_uint64)]
    heads = np.ascontiguousarray(heads, np.uint8)
    lens = np.ascontiguousarray(lens, np.uint64)
    mum_len = np.ascontiguousarray(mum_len, np.uint64)
    mum_pos = np.ascontiguousarray(mum_pos, np.uint64)
    cap = int(lens.sum()) + 8
    pos = np.zeros(cap, np.uint64)
    ids = np.zeros(cap, np.uint8)
    k, n = C.c_uint64(0), C.c_uint64(0)
    rc = L.colbwt_col_split_arrays(heads.ctypes.data, heads.size, lens.ctypes.data, mum_len.ctypes.data, mum_pos.ctypes.data,
                                   mum_len.size, int(num_docs), 1 if mode == "all" else 0, int(split_rate), int(device),
                                   pos.ctypes.data, cap, C.byref(k), ids.ctypes.data, C.byref(n))
    if rc != 0:
        raise ColbwtError(rc, L.colbwt_col_split_error().decode())
    return pos[:k.value].copy(), ids[:k.value].copy(), int(n.value)


class _RlbwtView(C.Structure):
    _fields_ = [("n", C.c_uint64), ("n_runs", C.c_uint64), ("n_mums", C.c_uint64), ("n_docs", C.c_uint32), ("rounds", C.c_int32),
                ("heads", C.POINTER(C.c_uint8)), ("lens", C.POINTER(C.c_uint64)), ("thr_pos", C.POINTER(C.c_uint64)),
                ("mum_len", C.POINTER(C.c_uint64)), ("mum_pos", C.POINTER(C.c_uint64))]


def _rlbwt_result(L, handle):
    v = _RlbwtView()
    L.colbwt_rlbwt_get.argtypes = [C.c_void_p, C.POINTER(_RlbwtView)]
    L.colbwt_rlbwt_free.argtypes = [C.c_void_p]
    L.colbwt_rlbwt_get(handle, C.byref(v))

    def arr(ptr, count, dtype):
        return np.ctypeslib.as_array(ptr, shape=(count,)).astype(dtype, copy=True) if count else np.zeros(0, dtype)
    out = dict(n=int(v.n), n_docs=int(v.n_docs), rounds=int(v.rounds), heads=arr(v.heads, v.n_runs, np.uint8),
               lens=arr(v.lens, v.n_runs, np.uint64), thr=arr(v.thr_pos, v.n_runs, np.uint64),
               mum_len=arr(v.mum_len, v.n_mums, np.uint64), mum_pos=arr(v.mum_pos, v.n_mums, np.uint64))
    L.colbwt_rlbwt_free(handle)
    return out


def rlbwt_from_text(text, doc_start, min_mum=20, device=0):
    """RLBWT, thresholds and multi-MUMs of a prepared text (separators 1, final 0) on the device:
    -> dict(n, n_docs, rounds, heads, lens, thr, mum_len, mum_pos)."""
    L = lib()
    L.colbwt_rlbwt_error.restype = C.c_char_p
    L.colbwt_rlbwt_build_text.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.POINTER(C.c_void_p)]
    t = np.frombuffer(bytes(text), np.uint8)
    ds = np.ascontiguousarray(doc_start, np.uint64)
    h = C.c_void_p()
    rc = L.colbwt_rlbwt_build_text(t.ctypes.data, t.size, ds.ctypes.data, ds.size, int(min_mum), int(device), C.byref(h))
    if rc != 0:
        raise ColbwtError(rc, L.colbwt_rlbwt_error().decode())
    return _rlbwt_result(L, h)


def rlbwt_from_fastas(paths, out_prefix=None, min_mum=20, revcomp=False, device=0):
    """`mumemto mum -K -R -T` of the reference's driver (col-bwt.py:121-145): one document per file;
    writes <out_prefix>.bwt.heads / .bwt.len / .thr_pos / .col_mums when given; returns the arrays."""
    L = lib()
    L.colbwt_rlbwt_error.restype = C.c_char_p
    L.colbwt_rlbwt_build_files.argtypes = [C.POINTER(C.c_char_p), C.c_uint32, C.c_int, C.c_uint64, C.c_int, C.c_char_p,
                                           C.POINTER(C.c_void_p)]
    arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
    h = C.c_void_p()
    rc = L.colbwt_rlbwt_build_files(arr, len(paths), int(bool(revcomp)), int(min_mum), int(device),
                                    os.fsencode(out_prefix) if out_prefix else None, C.byref(h))
    if rc != 0:
        raise ColbwtError(rc, L.colbwt_rlbwt_error().decode())
    return _rlbwt_result(L, h)


def _check_build(rc):
    if rc != 0:
        raise ColbwtError(rc, "index construction failed (missing/short input file or bad argument)")
