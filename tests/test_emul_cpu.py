"""CPU-only check of the kernels' arithmetic: the product's device header
(SWAR pack, window assembly, funnel-shift k-mer extraction, uint16-wrap
suppression) and its staging planner are compiled for the host
(tests/emul/qm2_emul.hip) and run lane by lane; the k-mer stream must equal the
oracle's, for every k, staging size and buffer alignment."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
import synth

HERE = os.path.dirname(os.path.abspath(__file__))
EMUL_DIR = os.path.join(HERE, "emul")
_L = None


def emul():
    global _L
    if _L is not None:
        return _L
    so = os.path.join(EMUL_DIR, "libqm2_emul.so")
    srcs = [os.path.join(EMUL_DIR, "qm2_emul.hip"),
            os.path.join(ol.ROOT, "quick-mer2_amd", "csrc", "qm2_device.hip.h"),
            os.path.join(ol.ROOT, "quick-mer2_amd", "csrc", "qm2_plan.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "--offload-arch=gfx950", "-fPIC",
                        "-shared", "-o", so, srcs[0]], check=True)
    L = C.CDLL(so)
    L.qm2e_extract.restype = C.c_uint64
    L.qm2e_extract.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_size_t, C.c_uint, C.c_void_p,
                               C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.qm2e_extract_packed.restype = C.c_uint64
    L.qm2e_extract_packed.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.qm2e_djb32.restype = C.c_uint32
    L.qm2e_djb32.argtypes = [C.c_uint64]
    _L = L
    return L


def emul_extract(blob, k, staging=1 << 20, misalign=0):
    L = emul()
    nb, npart = C.c_uint64(0), C.c_uint64(0)
    cap = len(blob) + 16
    out = np.empty(cap, dtype=np.uint64)
    n = L.qm2e_extract(blob, len(blob), k, staging, misalign, out.ctypes.data, cap, C.byref(nb), C.byref(npart))
    return out[:n], int(nb.value), int(npart.value)


def product():
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ol.ROOT, "quick-mer2_amd", "libqm2_hip.so")):
        ge.build()
    return ge.load_package()


def emul_extract_packed(blob, k, reads=None):
    """The product's host packer (qm2_pack_lines, or qm2_pack_reads on raw text with reads = (fastq, skip)), then the
    product's device-side reading of the blocks, lane by lane on the CPU -> (k-mers, n_bases as the library reports it)."""
    P = product().pyqm2
    if reads is None:
        blocks, p, wraps = P.pack_lines(blob, wraps_cap=1 << 12)
    else:
        blocks, p, wraps, _ = P.pack_reads(blob, reads[0], reads[1], wraps_cap=1 << 12)
    L = emul()
    cap = 64 * int(p.n_blocks) + 16
    out = np.empty(cap, dtype=np.uint64)
    npos = C.c_uint64(0)
    blocks = np.ascontiguousarray(blocks)
    wraps = np.ascontiguousarray(wraps, dtype=np.uint64)
    n = L.qm2e_extract_packed(blocks.ctypes.data if len(blocks) else None, int(p.n_blocks), k, wraps.ctypes.data if len(wraps) else None, len(wraps),
                              out.ctypes.data, cap, C.byref(npos))
    return out[:n], int(npos.value) - int(p.n_newlines)


def fuzz_blob(seed, n_lines=300, maxlen=400):
    rng = np.random.default_rng(seed)
    ref = synth.kat_reference().tobytes()
    parts = []
    for _ in range(n_lines):
        L = int(rng.integers(0, maxlen))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            s = rng.integers(0, 256, L, dtype=np.uint8).tobytes().replace(b"\n", b"A")
        elif kind == 1:
            p = int(rng.integers(0, len(ref) - maxlen))
            s = bytearray(ref[p:p + L])
            for _ in range(int(rng.integers(0, 4))):
                if L:
                    s[int(rng.integers(0, L))] = int(rng.choice(list(b"NnXacgt\r\x00\xff")))
            s = bytes(s)
        elif kind == 2:
            s = bytes([int(rng.choice(list(b"ACGTN")))]) * L
        else:
            p = int(rng.integers(0, len(ref) - maxlen))
            s = ref[p:p + L].lower()
        parts.append(s + b"\n")
    return b"".join(parts)


def test_djb32_is_low_half_of_djb():
    L, O = emul(), ol.lib()
    rng = np.random.default_rng(1)
    for x in [0, 1, 0xFF, (1 << 60) - 1, (1 << 64) - 1] + [int(v) for v in rng.integers(0, 1 << 62, 2000)]:
        assert L.qm2e_djb32(x) == (O.qm2o_djb(x) & 0xFFFFFFFF)


@pytest.mark.parametrize("k", [30, 1, 3, 12, 16, 25, 29, 31])
def test_fuzz_stream_equals_oracle(k):
    for seed in range(3):
        blob = fuzz_blob(100 * k + seed)
        exp, exp_nb = ol.extract(blob, k)
        for staging, mis in ((1 << 20, 0), (4096, 5), (700, 15), (97, 9)):
            got, nb, nparts = emul_extract(blob, k, staging, mis)
            assert nb == exp_nb, (k, seed, staging)
            assert np.array_equal(got, exp), (k, seed, staging, mis)


@pytest.mark.parametrize("k", [30, 25, 31, 12])
def test_long_lines_uint16_wrap(k):
    """Lines >= 65536 bases: the reference's uint16 run counter drops k windows per wrap."""
    tile = synth.kat_reference().tobytes()
    for L in (65535, 65536, 65537, 65566, 70000, 131072, 131100, 200000):
        s = (tile * (L // len(tile) + 1))[:L]
        blob = b"ACGT\n" + s + b"\n" + tile[:100] + b"\n"
        exp, exp_nb = ol.extract(blob, k)
        for staging, mis in ((1 << 22, 0), (65536, 3), (50000, 0), (9973, 11)):
            got, nb, _ = emul_extract(blob, k, staging, mis)
            assert nb == exp_nb and np.array_equal(got, exp), (k, L, staging)
    s = bytearray((tile * 60)[:250000])
    for p in (1234, 70000, 70001, 135535, 200000):
        s[p] = ord("N")
    blob = bytes(s) + b"\n" + bytes(s[:66000]) + b"\n"
    exp, exp_nb = ol.extract(blob, k)
    for staging in (1 << 22, 40000):
        got, nb, _ = emul_extract(blob, k, staging, 0)
        assert nb == exp_nb and np.array_equal(got, exp)


@pytest.mark.parametrize("k", [30, 1, 12, 25, 29, 31])
def test_packed_stream_equals_oracle(k):
    """host packer -> the device's reading of packed input == the oracle's k-mer stream and base count"""
    for seed in range(3):
        blob = fuzz_blob(100 * k + seed)
        exp, exp_nb = ol.extract(blob, k)
        got, nb = emul_extract_packed(blob, k)
        assert nb == exp_nb and np.array_equal(got, exp), (k, seed)
    for blob in (b"", b"\n", b"\n\n\n", b"A\n", b"ACGT\n" * 3, b"N" * 100 + b"\n", synth.kat_reference().tobytes()[:30] + b"\n",
                 synth.kat_reference().tobytes()[:29] + b"\n" + synth.kat_reference().tobytes()[:31] + b"\n",
                 synth.kat_reference().tobytes()[:63] + b"\n", synth.kat_reference().tobytes()[:64] + b"\n"):
        exp, exp_nb = ol.extract(blob, k)
        got, nb = emul_extract_packed(blob, k)
        assert nb == exp_nb and np.array_equal(got, exp), (k, blob[:20])


@pytest.mark.parametrize("k", [30, 25, 31])
def test_packed_long_lines_uint16_wrap(k):
    tile = synth.kat_reference().tobytes()
    for L in (65535, 65536, 65537, 65566, 70000, 131100):
        s = (tile * (L // len(tile) + 1))[:L]
        blob = b"ACGT\n" + s + b"\n" + tile[:100] + b"\n"
        exp, exp_nb = ol.extract(blob, k)
        got, nb = emul_extract_packed(blob, k)
        assert nb == exp_nb and np.array_equal(got, exp), (k, L)
        got, nb = emul_extract_packed(b">h\n" + blob + b">t\n", k, reads=(False, 0))       # the same through the line discipline
        assert nb == exp_nb and np.array_equal(got, exp), (k, L)
    s = bytearray((tile * 60)[:250000])
    for p in (1234, 70000, 70001, 135535, 200000):
        s[p] = ord("N")
    blob = bytes(s) + b"\n" + bytes(s[:66000]) + b"\n"
    exp, exp_nb = ol.extract(blob, k)
    got, nb = emul_extract_packed(blob, k)
    assert nb == exp_nb and np.array_equal(got, exp)


def test_packed_reads_of_the_kat_files():
    """raw FASTA / FASTQ vectors -> qm2_pack_reads -> the device's reading == the oracle on the oracle's own strip"""
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        fa, ctrl, rfa, rfq = synth.mini_inputs(td)
        for path in (rfa, rfq):
            text = open(path, "rb").read()
            fastq = text[:1] == b"@"
            body = text[text.index(b"\n") + 1:] if fastq else text
            for k in (30, 25):
                exp, exp_nb = ol.extract(ol.strip_reads(path), k)
                got, nb = emul_extract_packed(body, k, reads=(fastq, 0))
                assert nb == exp_nb and np.array_equal(got, exp), (path, k)


def test_edges():
    for blob in (b"\n", b"\n\n\n", b"A\n", b"ACGT\n" * 3, b"N" * 100 + b"\n",
                 synth.kat_reference().tobytes()[:30] + b"\n",
                 synth.kat_reference().tobytes()[:29] + b"\n" + synth.kat_reference().tobytes()[:31] + b"\n"):
        for k in (30, 25, 31):
            exp, exp_nb = ol.extract(blob, k)
            for mis in (0, 1, 15):
                got, nb, _ = emul_extract(blob, k, 1 << 20, mis)
                assert nb == exp_nb and np.array_equal(got, exp)


def test_kat_case_streams():
    cases = synth.kat_cases(synth.kat_reference())
    for name in ("polyA", "junk_bytes", "crlf", "n_lower", "n_upper", "split_lines", "short_reads", "long_with_N"):
        data = cases[name][1]
        blob = b"".join(l + b"\n" for l in data.split(b"\n") if l and not l.startswith(b">"))
        for k in (30, 25):
            exp, exp_nb = ol.extract(blob, k)
            got, nb, _ = emul_extract(blob, k, 5000, 7)
            assert nb == exp_nb and np.array_equal(got, exp), name


def _decode_record(r, w, bucket, lgH, lg_s):
    """The k-mer of bucket record (r, w) in bucket `bucket` of a table of 2^lgH slots, by the argument of
    qm2_device.hip.h ("bucket records"): the hash from bucket + w, V from the hash and b5..b7, its base-33 digits
    b4..b1 each from its residue and its top three bits, b0 what is left."""
    C33 = (5381 * pow(33, 8, 2 ** 32)) % 2 ** 32
    slot = w & ((1 << lg_s) - 1)
    assert (w >> lg_s) & 3 == 0                      # a k-mer's code bits
    ext = w >> (lg_s + 2)                            # (x : h) >> lgH
    h = ((ext << lgH) | (bucket << lg_s) | slot) & 0xFFFFFFFF
    x = ((ext << lgH) >> 32) & 3
    t4, mid, t2, t1, t3 = r & 7, (r >> 3) & 0xFFFFF, (r >> 23) & 7, (r >> 26) & 7, (r >> 29) & 7
    b5, b6, b7 = mid & 0xFF, (mid >> 8) & 0xFF, (mid >> 16) | (x << 4)
    V = ((h - C33 - (1089 * b5 + 33 * b6 + b7)) * pow(33 ** 3, -1, 2 ** 32)) % 2 ** 32
    digits = {}
    for i, t in ((4, t4), (3, t3), (2, t2), (1, t1)):
        cand = [v for v in range(32 * t, 32 * t + 32) if v % 33 == V % 33]
        assert len(cand) == 1
        digits[i] = cand[0]
        V = (V - cand[0]) // 33
    assert 0 <= V < 256
    return (V | digits[1] << 8 | digits[2] << 16 | digits[3] << 24 | digits[4] << 32 | b5 << 40 | b6 << 48 | b7 << 56)


@pytest.mark.parametrize("k,lgH,lg_s", [(30, 32, 12), (31, 32, 12), (30, 28, 12), (25, 21, 12), (31, 20, 12), (12, 16, 12), (30, 30, 12)])
def test_bucket_records_are_one_to_one(k, lgH, lg_s):
    """rec_r / rec_w (qm2_device.hip.h): the 6- or 8-byte record of a k-mer inside its finest bucket — what the
    partition's last level writes and bucket_count compares — decodes back to the k-mer, so two k-mers of one bucket
    are equal iff their records are; the record's slot bits are the home slot inside the bucket (QuicKmer.c:92); and
    the bits beside slot and code fit the width rec_ext_bits promises."""
    L = emul()
    L.qm2e_records.restype = None
    L.qm2e_records.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    L.qm2e_rec_ext_bits.restype = C.c_uint32
    L.qm2e_rec_ext_bits.argtypes = [C.c_uint32, C.c_uint32]
    rng = np.random.default_rng(k * 100 + lgH)
    mask = (1 << (2 * k)) - 1
    kmers = rng.integers(0, 2 ** 63, 20000, dtype=np.uint64) & np.uint64(mask)
    # corner cases: poly-A, all ones, single bytes, neighbours of a random k-mer in every byte
    extra = [0, mask, 1, 0xFF, 0xFF00, 0xFF << 32, 0x1F << 8, 0x20 << 8, 0x21 << 8]
    base = int(kmers[0])
    for byte in range(8):
        for d in (1, 32, 33, 66, 255):
            extra.append((base ^ (d << (8 * byte))) & mask)
    kmers = np.concatenate([kmers, np.array(extra, dtype=np.uint64) & np.uint64(mask)])
    r = np.zeros(len(kmers), dtype=np.uint32)
    w = np.zeros(len(kmers), dtype=np.uint32)
    L.qm2e_records(kmers.ctypes.data, len(kmers), lgH, lg_s, r.ctypes.data, w.ctypes.data)
    ext_bits = int(L.qm2e_rec_ext_bits(k, lgH))
    assert ext_bits == (32 - lgH) + (2 if k == 31 else 0)
    assert int(w.max()) >> (lg_s + 2 + ext_bits) == 0
    for km, ri, wi in zip(kmers.tolist(), r.tolist(), w.tolist()):
        h = int(L.qm2e_djb32(km))
        home = h & ((1 << lgH) - 1)
        assert wi & ((1 << lg_s) - 1) == home & ((1 << lg_s) - 1)
        assert _decode_record(ri, wi, (home >> lg_s), lgH, lg_s) == km


@pytest.mark.parametrize("lgH,lg_s,halo,fill,seed", [(14, 8, 32, 0.80, 1), (14, 8, 32, 0.93, 2), (13, 6, 16, 0.90, 3), (16, 12, 256, 0.85, 4)])
def test_record_compare_is_the_references_compare(lgH, lg_s, halo, fill, seed):
    """bucket_count's rule on the CPU, slot for slot: a k-mer walks the slice of its bucket (own slots + halo on the side
    walks leave through, QuicKmer.c:93-97) and "equals" a staged table-image record iff R and the second word are equal
    (rec_image_w as the loaders stage it, rec_staged_w).  Against Find_hash on the keys themselves (:90-99) for every
    dictionary k-mer and as many absent ones, on tables filled until runs of occupied slots cross bucket edges, several
    buckets and the middle of the table: the same slot or the same miss, whenever the walk stays inside the slice (the
    kernel finishes the others in global memory on the keys)."""
    L = emul()
    u32p = C.c_void_p
    L.qm2e_table_image.restype = None
    L.qm2e_table_image.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, u32p, u32p]
    L.qm2e_stage_words.restype = None
    L.qm2e_stage_words.argtypes = [u32p, C.c_uint64, C.c_int, C.c_int, C.c_uint32, u32p, C.c_void_p]
    L.qm2e_records.restype = None
    L.qm2e_records.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    H, S = 1 << lgH, 1 << lg_s
    rng = np.random.default_rng(seed)
    cand = np.unique(rng.integers(1, 1 << 60, int(H * fill * 2.2), dtype=np.uint64))
    rng.shuffle(cand)
    n_dict = int(H * fill)
    present, absent = cand[:n_dict], cand[n_dict:2 * n_dict]
    homes = lambda a: np.array([int(L.qm2e_djb32(int(x))) & (H - 1) for x in a], dtype=np.int64)
    keys = np.zeros(H, dtype=np.uint64)
    for key, i in zip(present.tolist(), homes(present).tolist()):          # the reference's insertion (:846-860)
        step = -1 if i & (H >> 1) else 1
        while keys[i] != 0:
            i += step
        keys[i] = key
    img_r = np.zeros(H, dtype=np.uint32)
    img_w = np.zeros(H, dtype=np.uint32)
    L.qm2e_table_image(keys.ctypes.data, H, lg_s, img_r.ctypes.data, img_w.ctypes.data)
    # every image word staged all four ways
    staged, empty = {}, {}
    for hl in (0, 1):
        for dn in (0, 1):
            o = np.zeros(H, dtype=np.uint32)
            e = np.zeros(H, dtype=np.uint8)
            L.qm2e_stage_words(img_w.ctypes.data, H, hl, dn, lg_s, o.ctypes.data, e.ctypes.data)
            staged[hl, dn], empty[hl, dn] = o, e
    assert np.array_equal(empty[0, 0] != 0, keys == 0) and np.array_equal(empty[1, 1] != 0, keys == 0)
    probes = np.concatenate([present, absent])
    pr = np.zeros(len(probes), dtype=np.uint32)
    pw = np.zeros(len(probes), dtype=np.uint32)
    L.qm2e_records(probes.ctypes.data, len(probes), lgH, lg_s, pr.ctypes.data, pw.ctypes.data)
    inside = crossed_middle = in_halo = 0
    for key, r, w, home in zip(probes.tolist(), pr.tolist(), pw.tolist(), homes(probes).tolist()):
        b = home >> lg_s
        down = 1 if home & (H >> 1) else 0
        step = -1 if down else 1
        lo, hi = (b * S - halo, b * S + S) if down else (b * S, b * S + S + halo)        # the staged slice
        # the reference
        i, ref = home, None
        while True:
            if keys[i] == key:
                ref = i
                break
            if keys[i] == 0:
                break
            i += step
        last = i
        # the records, while the walk is inside the slice
        i, got, left = home, None, False
        while True:
            if not (max(lo, 0) <= i < min(hi, H)):
                left = True
                break
            hl = 0 if b * S <= i < b * S + S else 1
            if img_r[i] == r and staged[hl, down][i] == w:
                got = i
                break
            if empty[hl, down][i]:
                break
            i += step
        if left:
            assert not (max(lo, 0) <= last < min(hi, H))           # ... only when the reference's walk leaves it too
            continue
        inside += 1
        assert got == ref, (hex(key), home, ref, got)
        in_halo += 1 if ref is not None and not (b * S <= ref < b * S + S) else 0
        crossed_middle += 1 if ref is not None and (ref >> (lgH - 1)) != (home >> (lgH - 1)) else 0
    assert inside > 0.6 * len(probes) and in_halo > 0      # (at 93 % fill a tenth of the walks outrun a 32-slot halo)
    if fill >= 0.9:
        assert crossed_middle > 0


def test_partition_buffer_layouts_are_one_to_one():
    """slot_of (qm2_device.hip.h): the child-major and the segmented [segment][region] layout of a partition
    level's buffer each map (region, k-mer index) one-to-one onto the buffer's slots, and in the segmented one
    the two k-mers of an aligned 16-byte load (even index, split_kmer's fetch) are neighbours inside one segment."""
    L = emul()
    L.qm2e_slot_of.restype = C.c_uint64
    L.qm2e_slot_of.argtypes = [C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
    L.qm2e_seg_kmers.restype = C.c_uint32
    seg = int(L.qm2e_seg_kmers())
    lg_seg = seg.bit_length() - 1
    regions, n_seg = 16, 3                       # 8 sub-regions of 2 children, 3 segments each
    cap = n_seg * seg
    lg_row = 4 + lg_seg
    for segmented in (0, 1):
        seen = np.zeros(regions * cap, dtype=np.uint8)
        for r in range(regions):
            es = np.unique(np.concatenate([np.arange(0, 64), np.arange(seg - 64, seg + 64), np.arange(cap - 64, cap),
                                           np.random.default_rng(r).integers(0, cap, 4000)]))
            for e in es:
                at = L.qm2e_slot_of(segmented, r, int(e), cap, lg_row)
                assert at < regions * cap
                seen[at] += 1
                if e % 2 == 0 and e + 1 < cap:
                    assert L.qm2e_slot_of(segmented, r, int(e) + 1, cap, lg_row) == at + 1
        assert seen.max() == 1
    # a region's segment s starts at row s, column region
    assert L.qm2e_slot_of(1, 5, 2 * seg + 7, cap, lg_row) == (2 << lg_row) + 5 * seg + 7
    assert L.qm2e_slot_of(0, 5, 2 * seg + 7, cap, lg_row) == 5 * cap + 2 * seg + 7


def emul_extract_dealt(blob, k, piece_cap, batch_bytes, n_ranks):
    L = emul()
    L.qm2e_extract_dealt.restype = C.c_uint64
    L.qm2e_extract_dealt.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p,
                                     C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    nb, nr = C.c_uint64(0), C.c_uint64(0)
    cap = len(blob) + 16
    out = np.empty(cap, dtype=np.uint64)
    n = L.qm2e_extract_dealt(blob, len(blob), k, piece_cap, batch_bytes, n_ranks, out.ctypes.data, cap, C.byref(nb), C.byref(nr))
    return out[:n], int(nb.value), int(nr.value)


@pytest.mark.parametrize("k", [30, 25, 31, 12])
def test_shards_dealing_of_long_lines(k):
    """`quicKmer2 count --shards` deals the stream of sequence lines to its ranks in staged pieces (qm2_shards.hip with
    qm2_plan.h's next_piece / LineStream): round-robin by commit, a line longer than a staging buffer in pieces that stay
    on one rank, and — when that rank's batch ends under the line — on behind a round with its last 32 bytes as look-back
    (which emit nothing), the uint16 run-counter wraps followed across pieces and batches.  The k-mers of all rounds are
    the oracle's, as a multiset (the ranks' order is not the stream's), and so are the bases."""
    tile = synth.kat_reference().tobytes()
    s = bytearray((tile * 60)[:250000])
    for p in (1234, 65500, 70000, 70001, 135535, 200000):
        s[p] = ord("N")
    long_n = bytes(s)
    blobs = [fuzz_blob(7 + k, n_lines=120),
             b"ACGT\n" + (tile * 30)[:140000] + b"\n" + tile[:100] + b"\n" + (tile * 30)[7:131100] + b"\n",
             long_n + b"\n" + long_n[:66000] + b"\n" + fuzz_blob(3, n_lines=40) + long_n[100:70000] + b"\n"]
    for bi, blob in enumerate(blobs):
        exp, exp_nb = ol.extract(blob, k)
        exp = np.sort(exp)
        for piece, batch, ranks in ((1 << 20, 1 << 22, 1), (20000, 40000, 4), (4096, 65536, 2), (97, 500, 3), (33, 66, 1),
                                    (65536, 65536, 2), (50000, 1 << 22, 4)):
            got, nb, rounds = emul_extract_dealt(blob, k, piece, batch, ranks)
            assert nb == exp_nb, (k, bi, piece, batch, ranks)
            assert np.array_equal(np.sort(got), exp), (k, bi, piece, batch, ranks)
            if batch * ranks < len(blob) // 2:
                assert rounds >= 2
