"""ctypes binding of the CPU oracle (oracle/libqm2_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None


class Stats(C.Structure):
    _fields_ = [("n_bases", C.c_uint64), ("n_kmers", C.c_uint64),
                ("n_probes", C.c_uint64), ("n_hits", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Dict(C.Structure):
    _fields_ = [("k", C.c_uint32), ("hdr_bytes", C.c_uint8 * 4),
                ("hash_size", C.c_uint64), ("first_idx", C.c_uint64),
                ("keys", C.POINTER(C.c_uint64)), ("chain", C.POINTER(C.c_uint32))]


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(ORACLE_DIR, "libqm2_oracle.so")
    src = os.path.join(ORACLE_DIR, "qm2_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", ORACLE_DIR, "libqm2_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    L = C.CDLL(so)
    u64p = C.POINTER(C.c_uint64)
    L.qm2o_djb.restype = C.c_uint64
    L.qm2o_djb.argtypes = [C.c_uint64]
    L.qm2o_find.restype = C.c_int
    L.qm2o_find.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, u64p, u64p]
    L.qm2o_extract.restype = C.c_uint64
    L.qm2o_extract.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_uint64, u64p]
    L.qm2o_count_lines.restype = C.c_int
    L.qm2o_count_lines.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_char_p, C.c_size_t,
                                   C.c_void_p, C.POINTER(Stats)]
    L.qm2o_count_lines_mt.restype = C.c_int
    L.qm2o_count_lines_mt.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_char_p, C.c_size_t,
                                      C.c_void_p, C.c_int, C.POINTER(Stats)]
    L.qm2o_dict_load.restype = C.c_int
    L.qm2o_dict_load.argtypes = [C.c_char_p, C.POINTER(Dict)]
    L.qm2o_dict_free.argtypes = [C.POINTER(Dict)]
    L.qm2o_strip_reads.restype = C.c_void_p
    L.qm2o_strip_reads.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_size_t)]
    L.qm2o_epilogue.restype = C.c_uint64
    L.qm2o_epilogue.argtypes = [C.POINTER(Dict), C.c_void_p, C.c_char_p, C.c_char_p,
                                C.POINTER(C.c_double)]
    L.qm2o_count_main.restype = C.c_int
    L.qm2o_count_main.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int,
                                  C.POINTER(Stats), C.POINTER(C.c_double)]
    _LIB = L
    return L


def load_qm(path):
    """Parse a .qm with numpy (format: SURVEY Appendix A.1).  Returns dict."""
    with open(path, "rb") as f:
        hdr = f.read(24)
        k = hdr[4]
        H = int.from_bytes(hdr[8:16], "little")
        first = int.from_bytes(hdr[16:24], "little")
        keys = np.fromfile(f, dtype=np.uint64, count=H)
        chain = np.fromfile(f, dtype=np.uint32, count=H)
    assert len(keys) == H and len(chain) == H
    return {"k": k, "H": H, "first": first, "keys": keys, "chain": chain, "hdr": hdr}


def extract(lines: bytes, k: int):
    L = lib()
    nb = C.c_uint64(0)
    n = L.qm2o_extract(lines, len(lines), k, None, 0, C.byref(nb))
    out = np.empty(n, dtype=np.uint64)
    nb = C.c_uint64(0)
    L.qm2o_extract(lines, len(lines), k, out.ctypes.data, n, C.byref(nb))
    return out, int(nb.value)


def count_lines(keys: np.ndarray, k: int, lines: bytes, threads=0, depth=None):
    """Oracle count of a sequence-lines buffer.  Returns (depth u16[H], stats dict)."""
    L = lib()
    H = len(keys)
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    if depth is None:
        depth = np.zeros(H, dtype=np.uint16)
    st = Stats()
    rc = L.qm2o_count_lines_mt(keys.ctypes.data, H, k, lines, len(lines), depth.ctypes.data,
                               threads, C.byref(st))
    assert rc == 0
    return depth, st.as_dict()


def zero_slot(keys: np.ndarray) -> int:
    """The empty slot k-mer 0 resolves to (Find_hash of 0 stops on the first empty slot and
    reports a match, QuicKmer.c:96-98)."""
    L = lib()
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    idx = C.c_uint64(0)
    assert L.qm2o_find(keys.ctypes.data, len(keys), 0, C.byref(idx), None) == 1
    return int(idx.value)


def remove_drain_padding(depth: np.ndarray, keys: np.ndarray, n_kmers: int) -> np.ndarray:
    """The threaded reference pads its last batch with zeros (QuicKmer.c:460-466); each padding
    zero bumps the counter of zero_slot(keys), which is never on the chain.  Undo that so a
    threaded-oracle counter array can be compared slot for slot with a serial / GPU one."""
    pad = 4096 - (n_kmers % 4096)
    out = depth.copy()
    z = zero_slot(keys)
    out[z] = np.uint16((int(out[z]) - pad) & 0xFFFF)
    return out


def strip_reads(path, seekable=True) -> bytes:
    L = lib()
    n = C.c_size_t(0)
    p = L.qm2o_strip_reads(path.encode(), 1 if seekable else 0, C.byref(n))
    assert p
    data = C.string_at(p, n.value)
    C.CDLL(None).free(C.c_void_p(p))
    return data


def chain_order(qm):
    """Slots in genome (chain) order: the epilogue's walk, :494-516."""
    first = np.uint32(qm["first"] & 0xFFFFFFFF)
    chain = qm["chain"]
    out = []
    at = int(first)
    while True:
        out.append(at)
        at = int(chain[at])
        if at == int(first):
            break
    return np.asarray(out, dtype=np.int64)


def count_main(ref_prefix, reads, out_prefix, threads=0):
    L = lib()
    st = Stats()
    sec = C.c_double(0)
    rc = L.qm2o_count_main(ref_prefix.encode(), reads.encode(), out_prefix.encode(), threads,
                           C.byref(st), C.byref(sec))
    return rc, st.as_dict(), sec.value
