"""ctypes binding of the C-ABI in include/qm2.h (libqm2_hip.so).

This is the reference-side stub a Python caller would use (INTEGRATION.md shows
the C one).  It holds no algorithm: every call goes straight to the HIP
library.  There is no CPU fallback — if the library is missing or no GPU is
usable the calls raise.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QM2_LIB", os.path.join(HERE, "libqm2_hip.so"))   # QM2_LIB: tuning variants only

FLAG_TIMING = 0x1
FLAG_NO_STATS = 0x2
FLAG_DIRECT = 0x4
FLAG_PARTITION = 0x8
FLAG_KEYS_LATER = 0x10
FLAG_RAW_STRICT = 0x20
FLAG_SUBMIT_ASCII = 0x40
PACKED_BLOCK_BYTES = 24
KERNEL_CLASSES = ["count_direct", "split_text", "split_kmer", "bucket_count", "probe", "encode", "lines", "-"]


class Opts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("flags", C.c_uint32),
                ("n_staging", C.c_uint32), ("staging_bytes", C.c_uint64), ("batch_bytes", C.c_uint64),
                ("part_max_bits", C.c_uint32), ("kmers_per_kib", C.c_uint32),
                ("ext_keys", C.c_void_p), ("ext_counters", C.c_void_p), ("stream", C.c_void_p),
                ("span_lo", C.c_uint64), ("span_hi", C.c_uint64), ("own_lo", C.c_uint64), ("own_hi", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("n_bases", C.c_uint64), ("n_kmers", C.c_uint64),
                ("n_probes", C.c_uint64), ("n_hits", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Profile(C.Structure):
    _fields_ = [("n_launches", C.c_uint64), ("kernel_ms", C.c_double), ("last_kernel_ms", C.c_double),
                ("class_launches", C.c_uint64 * 8), ("class_ms", C.c_double * 8),
                ("partitioned_batches", C.c_uint64), ("direct_batches", C.c_uint64), ("fallback_batches", C.c_uint64),
                ("allocations", C.c_uint64), ("spilled_kmers", C.c_uint64), ("staged_bytes", C.c_uint64), ("refused_batches", C.c_uint64), ("replans", C.c_uint64)]


class Packed(C.Structure):
    """qm2_packed: a piece of packed sequence (include/qm2.h, "packed sequence")"""
    _fields_ = [("n_blocks", C.c_uint64), ("n_newlines", C.c_uint64), ("n_wraps", C.c_uint64)]


class LineState(C.Structure):
    _fields_ = [("fastq", C.c_uint32), ("skip", C.c_uint32), ("desync", C.c_uint32), ("lines", C.c_uint32)]


class Range(C.Structure):
    _fields_ = [("start", C.c_uint64), ("len", C.c_uint64)]


class Dup(C.Structure):
    _fields_ = [("dst", C.c_uint64), ("src", C.c_uint64), ("len", C.c_uint64)]


# every symbol include/qm2.h and include/qm2_tools.h declare
ABI_SYMBOLS = ["qm2_flush", "qm2_merge_counters", "qm2_route_device", "qm2_route_device_carry", "qm2_probe_slabs_device", "qm2_probe_subslabs_device", "qm2_route_xcd_device", "qm2_route_xcd_device_carry", "qm2_prepare", "qm2_stats_peek", "qm2_open", "qm2_submit", "qm2_staging_acquire", "qm2_staging_commit", "qm2_submit_device", "qm2_sync", "qm2_finish", "qm2_reset", "qm2_keys_changed",
               "qm2_close", "qm2_counters_device", "qm2_keys_device", "qm2_get_profile",
               "qm2_encode_device", "qm2_probe_device", "qm2_chain_gather", "qm2_chain_fetch", "qm2_keys_from_fd", "qm2_chain_gather_fd", "qm2_chain_gc_sums_fd", "qm2_submit_raw_device", "qm2_staging_commit_raw", "qm2_raw_verdict",
               "qm2_packed_bound", "qm2_pack_lines", "qm2_pack_reads", "qm2_pack_isa", "qm2_pack_isa_select", "qm2_staging_commit_packed", "qm2_submit_packed_device", "qm2_submit_threads",
               "qm2_shard_span", "qm2_bind_near_device", "qm2_last_error",
               "qm2_abi_version"]
TOOL_SYMBOLS = ["qm2t_genome_generate", "qm2t_dict_build", "qm2t_reads_generate",
                "qm2t_read_offsets", "qm2t_microbench", "qm2t_checksum_u32"]

_LIB = None


def lib():
    """Load libqm2_hip.so; raises (never falls back) when it is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libqm2_hip.so is not built (run `make -C quick-mer2_amd` or "
                           "__graft_entry__.build()); there is no CPU fallback")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7.  Importing
    # torch first makes libqm2_hip.so's NEEDED libamdhip64.so.7 resolve to that already-loaded
    # copy, so device pointers and streams are interchangeable between torch and this library
    # (loading ours first would leave torch unable to see the GPU).  Without torch installed the
    # library binds to /opt/rocm's runtime, as the C CLI does.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    L.qm2_open.restype = C.c_int
    L.qm2_open.argtypes = [C.POINTER(vp), vp, u64, u32, C.POINTER(Opts)]
    L.qm2_submit.restype = C.c_int
    L.qm2_submit.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.qm2_staging_acquire.restype = C.c_int
    L.qm2_staging_acquire.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.qm2_staging_commit.restype = C.c_int
    L.qm2_staging_commit.argtypes = [vp, vp, C.c_size_t]
    L.qm2_submit_device.restype = C.c_int
    L.qm2_submit_device.argtypes = [vp, vp, C.c_size_t, vp, u64]
    L.qm2_sync.restype = C.c_int
    L.qm2_sync.argtypes = [vp]
    L.qm2_finish.restype = C.c_int
    L.qm2_finish.argtypes = [vp, vp, C.POINTER(Stats)]
    L.qm2_reset.restype = C.c_int
    L.qm2_reset.argtypes = [vp]
    L.qm2_keys_changed.restype = C.c_int
    L.qm2_keys_changed.argtypes = [vp]
    L.qm2_prepare.restype = C.c_int
    L.qm2_prepare.argtypes = [vp, u64, u32]
    L.qm2_stats_peek.restype = C.c_int
    L.qm2_stats_peek.argtypes = [vp, C.POINTER(Stats)]
    L.qm2_close.restype = None
    L.qm2_close.argtypes = [vp]
    L.qm2_counters_device.restype = vp
    L.qm2_counters_device.argtypes = [vp]
    L.qm2_keys_device.restype = vp
    L.qm2_keys_device.argtypes = [vp]
    L.qm2_get_profile.restype = C.c_int
    L.qm2_get_profile.argtypes = [vp, C.POINTER(Profile)]
    L.qm2_encode_device.restype = C.c_int
    L.qm2_encode_device.argtypes = [vp, vp, C.c_size_t, vp, u64, vp, u64, u32, vp]
    L.qm2_probe_device.restype = C.c_int
    L.qm2_probe_device.argtypes = [vp, vp, u64, vp]
    L.qm2_flush.restype = C.c_int
    L.qm2_flush.argtypes = [vp]
    L.qm2_merge_counters.restype = C.c_int
    L.qm2_merge_counters.argtypes = [vp, vp]
    L.qm2_route_device.restype = C.c_int
    L.qm2_route_device.argtypes = [vp, vp, C.c_size_t, vp, u64, vp, u64, u32, vp, vp]
    L.qm2_route_device_carry.restype = C.c_int
    L.qm2_route_device_carry.argtypes = [vp, vp, C.c_size_t, u32, vp, u64, vp, u64, u32, vp, vp]
    L.qm2_probe_slabs_device.restype = C.c_int
    L.qm2_probe_slabs_device.argtypes = [vp, vp, u64, u32, vp]
    L.qm2_probe_subslabs_device.restype = C.c_int
    L.qm2_probe_subslabs_device.argtypes = [vp, vp, u64, u32, u32, u32, vp, u64]
    L.qm2_route_xcd_device.restype = C.c_int
    L.qm2_route_xcd_device.argtypes = [vp, vp, C.c_size_t, vp, u64, vp, u64, u32, vp, vp]
    L.qm2_route_xcd_device_carry.restype = C.c_int
    L.qm2_route_xcd_device_carry.argtypes = [vp, vp, C.c_size_t, u32, vp, u64, vp, u64, u32, vp, vp]
    L.qm2_chain_gather.restype = C.c_int
    L.qm2_chain_gather.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.qm2_submit_raw_device.restype = C.c_int
    L.qm2_submit_raw_device.argtypes = [vp, vp, C.c_size_t, u32, u32, C.POINTER(u32)]
    L.qm2_staging_commit_raw.restype = C.c_int
    L.qm2_staging_commit_raw.argtypes = [vp, vp, C.c_size_t, C.c_size_t, u32, u32]
    L.qm2_raw_verdict.restype = u32
    L.qm2_raw_verdict.argtypes = [vp]
    L.qm2_packed_bound.restype = C.c_size_t
    L.qm2_packed_bound.argtypes = [C.c_size_t]
    L.qm2_pack_lines.restype = C.c_int
    L.qm2_pack_lines.argtypes = [vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(Packed)]
    L.qm2_pack_reads.restype = C.c_int
    L.qm2_pack_reads.argtypes = [vp, C.c_size_t, C.POINTER(LineState), vp, C.c_size_t, vp, C.c_size_t, C.POINTER(Packed)]
    L.qm2_pack_isa.restype = C.c_char_p
    L.qm2_pack_isa_select.restype = C.c_char_p
    L.qm2_pack_isa_select.argtypes = [C.c_char_p]
    L.qm2_staging_commit_packed.restype = C.c_int
    L.qm2_staging_commit_packed.argtypes = [vp, vp, C.POINTER(Packed), vp]
    L.qm2_submit_threads.restype = C.c_int
    L.qm2_submit_threads.argtypes = [vp, vp, C.c_size_t, u32]
    L.qm2_submit_packed_device.restype = C.c_int
    L.qm2_submit_packed_device.argtypes = [vp, vp, C.POINTER(Packed), vp]
    L.qm2_keys_from_fd.restype = C.c_int
    L.qm2_keys_from_fd.argtypes = [vp, C.c_int, u64]
    L.qm2_chain_gather_fd.restype = C.c_int
    L.qm2_chain_gather_fd.argtypes = [vp, C.c_int, u64, u64, C.POINTER(u64)]
    L.qm2_chain_gc_sums_fd.restype = C.c_int
    L.qm2_chain_gc_sums_fd.argtypes = [vp, C.c_int, u64, vp, vp, vp, vp]
    L.qm2_chain_fetch.restype = C.c_int
    L.qm2_chain_fetch.argtypes = [vp, vp, u64, u64]
    L.qm2_shard_span.restype = C.c_int
    L.qm2_shard_span.argtypes = [vp, u64, u32, u32] + [C.POINTER(u64)] * 4
    L.qm2_bind_near_device.restype = C.c_int
    L.qm2_bind_near_device.argtypes = [C.c_int, C.POINTER(u32)]
    L.qm2_last_error.restype = C.c_char_p
    L.qm2_abi_version.restype = u32
    L.qm2t_genome_generate.restype = C.c_int
    L.qm2t_genome_generate.argtypes = [vp, u64, u64, C.c_float, C.c_float, C.POINTER(Range), u32,
                                       C.POINTER(Dup), u32, vp]
    L.qm2t_dict_build.restype = C.c_int
    L.qm2t_dict_build.argtypes = [vp, u64, u32, vp, u64, vp, C.POINTER(u64), C.POINTER(u64), vp]
    L.qm2t_reads_generate.restype = C.c_int
    L.qm2t_reads_generate.argtypes = [vp, u64, vp, vp, u64, u64, u32, u32, u64, C.c_float, C.c_float,
                                      C.POINTER(Range), u32, vp]
    L.qm2t_read_offsets.restype = None
    L.qm2t_read_offsets.argtypes = [vp, u64, u64, u32, u32, u64]
    L.qm2t_microbench.restype = C.c_int
    L.qm2t_microbench.argtypes = [u64, u64, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.qm2t_checksum_u32.restype = C.c_int
    L.qm2t_checksum_u32.argtypes = [vp, u64, C.POINTER(u64 * 3), vp]
    _LIB = L
    return L


class Qm2Error(RuntimeError):
    pass


def _check(rc, what):
    if rc != 0:
        raise Qm2Error("%s failed: %s" % (what, lib().qm2_last_error().decode(errors="replace")))


class Counter:
    """One counting context (qm2_ctx).  keys: host uint64[H] (or None with ext_keys)."""

    def __init__(self, keys, hash_size=None, k=30, device=-1, flags=0, n_staging=0, staging_bytes=0,
                 batch_bytes=0, part_max_bits=0, ext_keys=None, ext_counters=None, stream=None, span=None, own=None,
                 kmers_per_kib=0):
        L = lib()
        self._h = C.c_void_p()
        o = Opts()
        o.struct_size = C.sizeof(Opts)
        o.device = device
        o.flags = flags
        o.n_staging = n_staging
        o.staging_bytes = staging_bytes
        o.batch_bytes = batch_bytes
        o.part_max_bits = part_max_bits
        o.kmers_per_kib = int(kmers_per_kib)
        o.ext_keys = ext_keys
        o.ext_counters = ext_counters
        o.stream = stream
        if span is not None:
            o.span_lo, o.span_hi = int(span[0]), int(span[1])
        if own is not None:
            o.own_lo, o.own_hi = int(own[0]), int(own[1])
        kp = None
        if keys is not None:
            keys = np.ascontiguousarray(keys, dtype=np.uint64)
            kp = keys.ctypes.data
            if hash_size is None:
                hash_size = len(keys)
        self.H = int(hash_size)
        self.n_slots = (int(span[1]) - int(span[0])) if span is not None else self.H
        self.k = k
        _check(L.qm2_open(C.byref(self._h), kp, self.H, k, C.byref(o)), "qm2_open")

    def submit(self, lines: bytes):
        _check(lib().qm2_submit(self._h, lines, len(lines)), "qm2_submit")

    def submit_threads(self, lines, threads, addr=None, n=None):
        """qm2_submit_threads: lines = bytes, or (addr, n) of host memory"""
        if addr is None:
            view = np.frombuffer(lines, np.uint8)
            addr, n = view.ctypes.data, len(view)
        _check(lib().qm2_submit_threads(self._h, addr, n, threads), "qm2_submit_threads")

    def submit_zero_copy(self, lines: bytes):
        """qm2_staging_acquire / fill / qm2_staging_commit, one staging buffer at a time."""
        L = lib()
        off = 0
        while off < len(lines):
            buf, cap = C.c_void_p(), C.c_size_t()
            _check(L.qm2_staging_acquire(self._h, C.byref(buf), C.byref(cap)), "qm2_staging_acquire")
            take = min(cap.value, len(lines) - off)
            if off + take < len(lines):
                take = lines.rfind(b"\n", off, off + take) + 1 - off
                if take <= 0:
                    L.qm2_staging_commit(self._h, buf, 0)
                    raise Qm2Error("a line is longer than a staging buffer")
            C.memmove(buf, lines[off:off + take], take)
            _check(L.qm2_staging_commit(self._h, buf, take), "qm2_staging_commit")
            off += take

    def staging_acquire(self):
        """-> (address of a pinned staging buffer, its capacity)"""
        buf, cap = C.c_void_p(), C.c_size_t()
        _check(lib().qm2_staging_acquire(self._h, C.byref(buf), C.byref(cap)), "qm2_staging_acquire")
        return buf.value, cap.value

    def staging_commit(self, addr, n):
        _check(lib().qm2_staging_commit(self._h, addr, n), "qm2_staging_commit")

    def staging_commit_raw(self, addr, offset, n, fastq, skip):
        _check(lib().qm2_staging_commit_raw(self._h, addr, offset, n, 1 if fastq else 0, skip), "qm2_staging_commit_raw")

    def raw_verdict(self):
        return int(lib().qm2_raw_verdict(self._h))

    def feed_host(self, addr, n, unit, raw=None, threads=8, depth=6):
        """Stream n bytes of HOST memory at `addr` through the pinned staging buffers the way the CLI's
        reader does: `threads` threads copy into acquired buffers (ctypes.memmove releases the GIL),
        this thread commits them in order.  Pieces are whole multiples of `unit` bytes (a line, or a
        FASTQ record).  raw = None: stripped sequence lines (qm2_staging_commit); raw = (fastq, skip):
        raw text with the line state at the start of every piece (qm2_staging_commit_raw)."""
        from concurrent.futures import ThreadPoolExecutor
        pend, off = [], 0
        with ThreadPoolExecutor(max_workers=threads) as pool:
            while off < n or pend:
                while off < n and len(pend) < depth:
                    buf, cap = self.staging_acquire()
                    take = min((cap // unit) * unit, n - off)
                    pend.append((buf, take, pool.submit(C.memmove, buf, addr + off, take)))
                    off += take
                buf, take, fut = pend.pop(0)
                fut.result()
                if raw is None:
                    self.staging_commit(buf, take)
                else:
                    self.staging_commit_raw(buf, 0, take, raw[0], raw[1])

    def staging_commit_packed(self, addr, packed, wraps=None):
        _check(lib().qm2_staging_commit_packed(self._h, addr, C.byref(packed), wraps.ctypes.data if wraps is not None and len(wraps) else None),
               "qm2_staging_commit_packed")

    def submit_packed(self, lines: bytes, piece=None):
        """qm2_staging_acquire / qm2_pack_lines into the buffer / qm2_staging_commit_packed, a staging buffer at a time
        (piece: at most that many text bytes per buffer)."""
        L = lib()
        off = 0
        while off < len(lines):
            buf, cap = self.staging_acquire()
            take = min((cap // PACKED_BLOCK_BYTES) * 64, len(lines) - off, piece or (1 << 62))
            if off + take < len(lines):
                take = lines.rfind(b"\n", off, off + take) + 1 - off
                if take <= 0:                         # a line longer than a piece: the ASCII way (qm2_submit cuts it)
                    self.staging_commit(buf, 0)
                    end = lines.find(b"\n", off) + 1
                    _check(L.qm2_submit(self._h, lines[off:end], end - off), "qm2_submit")
                    off = end
                    continue
            p, wraps = Packed(), np.zeros(4096, np.uint64)
            view = np.frombuffer(lines, np.uint8, take, off)
            rc = L.qm2_pack_lines(view.ctypes.data, take, buf, cap, wraps.ctypes.data, len(wraps), C.byref(p))
            if rc:
                self.staging_commit(buf, 0)
                raise Qm2Error("qm2_pack_lines: %d" % rc)
            self.staging_commit_packed(buf, p, wraps[:p.n_wraps])
            off += take

    def submit_packed_device(self, ptr, packed, wraps_ptr=None):
        _check(lib().qm2_submit_packed_device(self._h, ptr, C.byref(packed), wraps_ptr), "qm2_submit_packed_device")

    def feed_host_packed(self, addr, n, unit, reads=None, threads=8, depth=None, piece_text=16 << 20):
        """feed_host with the packing on the threads (qm2_pack.c; the calls release the GIL): every piece of whole
        `unit`-byte records is packed into an acquired pinned buffer — reads = None: the text is sequence lines
        (qm2_pack_lines); reads = (fastq, skip): raw FASTA/FASTQ with that line state at the start of every piece
        (qm2_pack_reads) — and committed in order (qm2_staging_commit_packed).  Pieces of piece_text bytes of text (the
        CLI's reader: 32 MiB blocks), threads + 2 of them in flight (the context must have that many staging buffers + 1).
        Returns the bytes that crossed the link."""
        from concurrent.futures import ThreadPoolExecutor
        L = lib()
        pend, off, sent = [], 0, 0
        depth = depth or threads + 2

        def pack(buf, cap, src, take):
            p, wraps = Packed(), np.zeros(64, np.uint64)
            if reads is None:
                rc = L.qm2_pack_lines(src, take, buf, cap, wraps.ctypes.data, len(wraps), C.byref(p))
            else:
                st = LineState(1 if reads[0] else 0, reads[1], 0)
                rc = L.qm2_pack_reads(src, take, C.byref(st), buf, cap, wraps.ctypes.data, len(wraps), C.byref(p))
                if st.desync:
                    rc = 3
            return rc, p, wraps

        with ThreadPoolExecutor(max_workers=threads) as pool:
            while off < n or pend:
                while off < n and len(pend) < depth:
                    buf, cap = self.staging_acquire()
                    per = min((cap // PACKED_BLOCK_BYTES) * 64 - 64 if reads is None else cap, piece_text)
                    take = min(max(1, per // unit) * unit, n - off)
                    pend.append((buf, pool.submit(pack, buf, cap, addr + off, take)))
                    off += take
                buf, fut = pend.pop(0)
                rc, p, wraps = fut.result()
                if rc:
                    self.staging_commit(buf, 0)
                    raise Qm2Error("packing a piece failed (%d)" % rc)
                self.staging_commit_packed(buf, p, wraps[:p.n_wraps])
                sent += int(p.n_blocks) * PACKED_BLOCK_BYTES
        return sent

    def submit_device(self, ptr, n, suppress_ptr=None, n_suppress=0):
        _check(lib().qm2_submit_device(self._h, ptr, n, suppress_ptr, n_suppress), "qm2_submit_device")

    def encode_device(self, ptr, n, out_ptr, bucket_cap, n_buckets, counts_ptr, suppress_ptr=None, n_suppress=0):
        _check(lib().qm2_encode_device(self._h, ptr, n, suppress_ptr, n_suppress, out_ptr, bucket_cap,
                                       n_buckets, counts_ptr), "qm2_encode_device")

    def route_device(self, ptr, n, slabs_ptr, slab_cap, n_owners, fill_ptr, call_ptr, suppress_ptr=None, n_suppress=0):
        _check(lib().qm2_route_device(self._h, ptr, n, suppress_ptr, n_suppress, slabs_ptr, slab_cap, n_owners, fill_ptr,
                                      call_ptr), "qm2_route_device")

    def route_device_carry(self, ptr, n, carry, slabs_ptr, slab_cap, n_owners, fill_ptr, call_ptr, suppress_ptr=None, n_suppress=0):
        _check(lib().qm2_route_device_carry(self._h, ptr, n, carry, suppress_ptr, n_suppress, slabs_ptr, slab_cap, n_owners,
                                            fill_ptr, call_ptr), "qm2_route_device_carry")

    def probe_slabs_device(self, slabs_ptr, slab_cap, n_slabs, fill_ptr):
        _check(lib().qm2_probe_slabs_device(self._h, slabs_ptr, slab_cap, n_slabs, fill_ptr), "qm2_probe_slabs_device")

    def probe_subslabs_device(self, slabs_ptr, sub_cap, n_peers, n_subs, fill_ptr, kmers_bound=0, peer_stride=0):
        _check(lib().qm2_probe_subslabs_device(self._h, slabs_ptr, sub_cap, n_peers, n_subs, peer_stride, fill_ptr, int(kmers_bound)),
               "qm2_probe_subslabs_device")

    def route_xcd_device(self, ptr, n, slabs_ptr, region_cap, n_regions, fill_ptr, call_ptr, suppress_ptr=None, n_suppress=0):
        _check(lib().qm2_route_xcd_device(self._h, ptr, n, suppress_ptr, n_suppress, slabs_ptr, region_cap, n_regions, fill_ptr,
                                          call_ptr), "qm2_route_xcd_device")

    def probe_device(self, kmers_ptr, n, n_ptr=None):
        _check(lib().qm2_probe_device(self._h, kmers_ptr, n, n_ptr), "qm2_probe_device")

    def sync(self):
        _check(lib().qm2_sync(self._h), "qm2_sync")

    def finish(self, want_depth=True):
        st = Stats()
        depth = np.empty(self.n_slots, dtype=np.uint16) if want_depth else None
        _check(lib().qm2_finish(self._h, depth.ctypes.data if want_depth else None, C.byref(st)), "qm2_finish")
        return depth, st.as_dict()

    def reset(self):
        _check(lib().qm2_reset(self._h), "qm2_reset")

    def keys_changed(self):
        _check(lib().qm2_keys_changed(self._h), "qm2_keys_changed")

    def prepare(self, batch_bytes=0, host=False, raw=False, fastq=False, no_part=False, packed=False):
        what = (1 if host else 0) | (2 if raw else 0) | (4 if fastq else 0) | (8 if no_part else 0) | (16 if packed else 0)
        _check(lib().qm2_prepare(self._h, int(batch_bytes), what), "qm2_prepare")

    def flush(self):
        _check(lib().qm2_flush(self._h), "qm2_flush")

    def merge_counters(self, src):
        _check(lib().qm2_merge_counters(self._h, src._h), "qm2_merge_counters")

    def stats_peek(self):
        st = Stats()
        _check(lib().qm2_stats_peek(self._h, C.byref(st)), "qm2_stats_peek")
        return st.as_dict()

    def chain_gather(self, chain, first_idx, window=None):
        """Epilogue on the device: counts in genome (chain) order, = the .bin payload."""
        chain = np.ascontiguousarray(chain, dtype=np.uint32)
        n = C.c_uint64(0)
        _check(lib().qm2_chain_gather(self._h, chain.ctypes.data, int(first_idx), C.byref(n)), "qm2_chain_gather")
        out = np.empty(int(n.value), dtype=np.uint16)
        step = window or max(1, len(out))
        for a in range(0, len(out), step):
            m = min(step, len(out) - a)
            _check(lib().qm2_chain_fetch(self._h, out[a:].ctypes.data, a, m), "qm2_chain_fetch")
        return out

    def submit_raw_device(self, ptr, n, fastq, skip=0):
        """Raw FASTA/FASTQ text on the device; returns the verdict (0 = counted)."""
        v = C.c_uint32(0)
        _check(lib().qm2_submit_raw_device(self._h, ptr, n, 1 if fastq else 0, skip, C.byref(v)), "qm2_submit_raw_device")
        return int(v.value)

    def keys_from_fd(self, fd, file_offset=24):
        """Fill the key table of a context opened with FLAG_KEYS_LATER from a .qm file."""
        _check(lib().qm2_keys_from_fd(self._h, int(fd), int(file_offset)), "qm2_keys_from_fd")

    def chain_gather_fd(self, fd, first_idx, file_offset=None):
        """chain_gather with the chain read from the .qm (offset 24 + 8 H by default)."""
        n = C.c_uint64(0)
        off = 24 + 8 * self.H if file_offset is None else int(file_offset)
        _check(lib().qm2_chain_gather_fd(self._h, int(fd), off, int(first_idx), C.byref(n)), "qm2_chain_gather_fd")
        out = np.empty(int(n.value), dtype=np.uint16)
        if len(out):
            _check(lib().qm2_chain_fetch(self._h, out.ctypes.data, 0, len(out)), "qm2_chain_fetch")
        return out

    def chain_gc_sums_fd(self, fd, n_file, stale=None):
        """The GC curve's sums over the gathered counts (QuicKmer.c:504-509): -> (sum, sumsq, cnt), int64/int64/uint64 [401]."""
        s, q, c = np.zeros(401, np.int64), np.zeros(401, np.int64), np.zeros(401, np.uint64)
        if stale is not None:
            stale = np.ascontiguousarray(stale, dtype=np.uint16)
            assert len(stale) == 1 << 20
        _check(lib().qm2_chain_gc_sums_fd(self._h, int(fd), int(n_file), stale.ctypes.data if stale is not None else None,
                                          s.ctypes.data, q.ctypes.data, c.ctypes.data), "qm2_chain_gc_sums_fd")
        return s, q, c

    def profile(self):
        p = Profile()
        _check(lib().qm2_get_profile(self._h, C.byref(p)), "qm2_get_profile")
        return {"n_launches": int(p.n_launches), "kernel_ms": float(p.kernel_ms),
                "last_kernel_ms": float(p.last_kernel_ms),
                "class_launches": {KERNEL_CLASSES[i]: int(p.class_launches[i]) for i in range(7)},
                "class_ms": {KERNEL_CLASSES[i]: float(p.class_ms[i]) for i in range(7)},
                "partitioned_batches": int(p.partitioned_batches), "direct_batches": int(p.direct_batches),
                "fallback_batches": int(p.fallback_batches), "allocations": int(p.allocations),
                "spilled_kmers": int(p.spilled_kmers), "staged_bytes": int(p.staged_bytes), "refused_batches": int(p.refused_batches), "replans": int(p.replans)}

    @property
    def counters_ptr(self):
        return lib().qm2_counters_device(self._h)

    @property
    def keys_ptr(self):
        return lib().qm2_keys_device(self._h)

    def close(self):
        if self._h:
            lib().qm2_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_lines(lines: bytes, wraps_cap=4096):
    """qm2_pack_lines on the host -> (blocks as uint64[3 n_blocks], Packed, wraps)"""
    out = np.empty(lib().qm2_packed_bound(len(lines)) // 8, np.uint64)
    p, wraps = Packed(), np.zeros(wraps_cap, np.uint64)
    src = np.frombuffer(lines, np.uint8)                  # (bytes, or anything with the buffer protocol)
    rc = lib().qm2_pack_lines(src.ctypes.data if len(src) else None, len(src), out.ctypes.data, out.nbytes, wraps.ctypes.data, wraps_cap, C.byref(p))
    if rc:
        raise Qm2Error("qm2_pack_lines: %d" % rc)
    return out[:3 * p.n_blocks], p, wraps[:p.n_wraps]


def pack_reads(text: bytes, fastq, skip=0, wraps_cap=4096, room=None):
    """qm2_pack_reads on the host -> (blocks, Packed, wraps, LineState after the last line)"""
    out = np.zeros((lib().qm2_packed_bound(len(text)) if room is None else room) // 8, np.uint64)
    p, wraps, st = Packed(), np.zeros(wraps_cap, np.uint64), LineState(1 if fastq else 0, skip, 0)
    src = np.frombuffer(text, np.uint8)
    rc = lib().qm2_pack_reads(src.ctypes.data if len(text) else None, len(text), C.byref(st), out.ctypes.data, out.nbytes, wraps.ctypes.data, wraps_cap, C.byref(p))
    if rc:
        raise Qm2Error("qm2_pack_reads: %d" % rc)
    return out[:3 * p.n_blocks], p, wraps[:p.n_wraps], st


def bind_near_device(device=-1):
    """qm2_bind_near_device: this thread (and the ones it starts) onto the CPUs of the device's NUMA node; -> CPUs it may run on"""
    n = C.c_uint32(0)
    _check(lib().qm2_bind_near_device(device, C.byref(n)), "qm2_bind_near_device")
    return int(n.value)


def shard_span(keys, n_shards, g):
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    v = [C.c_uint64() for _ in range(4)]
    _check(lib().qm2_shard_span(keys.ctypes.data, len(keys), n_shards, g, *[C.byref(x) for x in v]),
           "qm2_shard_span")
    return tuple(int(x.value) for x in v)   # own_lo, own_hi, span_lo, span_hi


class tools:
    """Synthetic workload tooling (include/qm2_tools.h).  Device pointers are ints."""

    @staticmethod
    def genome_generate(d_ptr, n, seed, gc_lo=0.32, gc_hi=0.62, n_blocks=(), dups=(), stream=None):
        nb = (Range * max(1, len(n_blocks)))(*[Range(int(a), int(b)) for a, b in n_blocks])
        dp = (Dup * max(1, len(dups)))(*[Dup(int(a), int(b), int(c)) for a, b, c in dups])
        _check(lib().qm2t_genome_generate(d_ptr, n, seed, gc_lo, gc_hi, nb, len(n_blocks), dp, len(dups),
                                          stream), "qm2t_genome_generate")

    @staticmethod
    def dict_build(d_genome, n, k, d_keys, H, d_scratch, stream=None):
        a, b = C.c_uint64(0), C.c_uint64(0)
        _check(lib().qm2t_dict_build(d_genome, n, k, d_keys, H, d_scratch, C.byref(a), C.byref(b), stream),
               "qm2t_dict_build")
        return int(a.value), int(b.value)

    @staticmethod
    def read_offsets(n_reads, first_read, len_lo, len_hi, seed):
        off = np.empty(n_reads + 1, dtype=np.uint64)
        lib().qm2t_read_offsets(off.ctypes.data, n_reads, first_read, len_lo, len_hi, seed)
        return off

    @staticmethod
    def reads_generate(d_genome, n, d_out, d_offsets, n_reads, first_read, len_lo, len_hi, seed,
                       sub_rate=0.005, n_read_frac=0.01, sample_from=(), stream=None):
        fr = (Range * max(1, len(sample_from)))(*[Range(int(a), int(b)) for a, b in sample_from])
        _check(lib().qm2t_reads_generate(d_genome, n, d_out, d_offsets, n_reads, first_read, len_lo, len_hi,
                                         seed, sub_rate, n_read_frac, fr, len(sample_from), stream),
               "qm2t_reads_generate")

    @staticmethod
    def checksum_u32(d_ptr, n, stream=None):
        """-> (position-weighted checksum mod 2^64, plain sum, non-zero slots) of n u32 counters on the device"""
        out = (C.c_uint64 * 3)()
        _check(lib().qm2t_checksum_u32(d_ptr, n, C.byref(out), stream), "qm2t_checksum_u32")
        return int(out[0]), int(out[1]), int(out[2])

    @staticmethod
    def microbench(slots, n_ops, mode, repeats=3):
        ms = C.c_double(0)
        _check(lib().qm2t_microbench(slots, n_ops, mode, repeats, C.byref(ms)), "qm2t_microbench")
        return float(ms.value)
