/*
 * qm2_emul.cpp — TEST INFRASTRUCTURE: runs the product's device-side encoder
 * (quick-mer2_amd/csrc/qm2_device.hip.h: SWAR pack, window assembly, funnel-
 * shift k-mer extraction, DJB) and its host-side staging planner
 * (qm2_plan.h) on the CPU, lane by lane, so that the arithmetic of the HIP
 * kernels is checked against the oracle in the CPU-only test suite.  Compiled
 * by hipcc for the host (same headers, same code; only the three wave shuffles
 * of make_window are replaced by array look-ups).
 */
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include "../../quick-mer2_amd/csrc/qm2_device.hip.h"
#include "../../quick-mer2_amd/csrc/qm2_plan.h"

using namespace qm2;

template <bool K30, bool PK = false>
static void emul_part(const SeqView &sv, uint32_t k, std::vector<uint64_t> &out, uint64_t *n_bases)
{
    const uint64_t n_tiles = (sv.n_total + kTileBytes - 1) / kTileBytes;
    for (uint64_t tile = 0; tile < n_tiles; tile++) {
        Packed p[kWave];
        for (int lane = 0; lane < kWave; lane++) {
            const int64_t g = lane_offset(tile * kTileBytes, lane);
            p[lane] = PK ? unpack16(sv, g, load16_packed_raw(sv, g)) : pack16(load16(sv, g));
        }
        for (int lane = 0; lane < kWave; lane++) {
            /* __shfl_up(x, d): lanes < d keep their own value */
            const Packed p1 = p[lane >= 1 ? lane - 1 : lane], p2 = p[lane >= 2 ? lane - 2 : lane];
            const Window W = assemble_window<K30>(sv, lane_offset(tile * kTileBytes, lane), lane < kHaloLanes,
                                                  p[lane], p1, p2, k);
            *n_bases += W.n_bases;
#define E(J) if (W.emit & (0x8000u >> J)) out.push_back(kmer_at<K30, J>(W, k));
            E(0) E(1) E(2) E(3) E(4) E(5) E(6) E(7) E(8) E(9) E(10) E(11) E(12) E(13) E(14) E(15)
#undef E
        }
    }
}

extern "C" {

/* Emulates qm2_submit(buf, n) with the given staging capacity and device-buffer
 * misalignment; returns the emitted canonical k-mers in stream order. */
uint64_t qm2e_extract(const char *buf, size_t n, uint32_t k, size_t staging_bytes, unsigned misalign,
                      uint64_t *out, uint64_t cap, uint64_t *n_bases, uint64_t *n_parts)
{
    std::vector<Part> parts;
    plan_parts(buf, n, staging_bytes, staging_bytes, parts);
    std::vector<uint64_t> km;
    uint64_t bases = 0;
    for (const Part &pt : parts) {
        const size_t staged = pt.carry + pt.take;
        /* a 16-byte aligned "device" buffer with `misalign` leading bytes and a poisoned tail */
        std::vector<uint8_t> dev(staged + 64 + 32, 0x41 /* poison that would count as a base */);
        uint8_t *aligned = (uint8_t *)(((uintptr_t)dev.data() + 15) & ~(uintptr_t)15);
        uint8_t *data = aligned + (misalign & 15);
        memcpy(data, buf + pt.off - pt.carry, staged);
        SeqView sv;
        sv.lead = (uintptr_t)data & 15u;
        sv.base = data - sv.lead;
        sv.emit_from = sv.lead + pt.carry;
        sv.n_total = sv.lead + staged;
        sv.suppress = pt.wraps.empty() ? nullptr : pt.wraps.data();
        sv.n_suppress = pt.wraps.size();
        if (k == 30) emul_part<true>(sv, k, km, &bases);
        else emul_part<false>(sv, k, km, &bases);
    }
    for (uint64_t i = 0; i < km.size() && i < cap; i++) out[i] = km[i];
    if (n_bases) *n_bases = bases;
    if (n_parts) *n_parts = parts.size();
    return km.size();
}

/* Emulates the dealing of quick-mer2_amd/csrc/qm2_shards.hip (qm2_shards_submit -> stage -> run_round) with the product's
 * own planner pieces (qm2_plan.h: next_piece, LineStream): the stream is staged in pieces of at most `piece_cap` bytes,
 * round-robin over `n_ranks` texts of `batch_bytes` each; an over-long line stays on one rank and, when that rank's batch
 * ends under it, goes on behind a round with its look-back.  Every round "routes" each rank's text through the encoder
 * (carry bytes emit nothing, wraps as suppress entries).  Returns the k-mers of all rounds (order: by round and rank). */
uint64_t qm2e_extract_dealt(const char *buf, size_t n, uint32_t k, size_t piece_cap, size_t batch_bytes, int n_ranks,
                            uint64_t *out, uint64_t cap, uint64_t *n_bases, uint64_t *n_rounds)
{
    struct RankText { std::vector<uint8_t> text; size_t carry0 = 0; std::vector<uint64_t> wraps; };
    std::vector<RankText> R((size_t)n_ranks);
    std::vector<uint64_t> km;
    uint64_t bases = 0, rounds = 0;
    LineStream ls;
    int feed = 0;
    auto run_round = [&]() {
        bool any = false;
        for (RankText &r : R) {
            if (r.text.empty()) continue;
            any = true;
            std::vector<uint8_t> dev(r.text.size() + 64 + 32, 0x41);
            uint8_t *data = (uint8_t *)(((uintptr_t)dev.data() + 15) & ~(uintptr_t)15) + 5;       /* (misaligned on purpose) */
            memcpy(data, r.text.data(), r.text.size());
            SeqView sv;
            sv.lead = (uintptr_t)data & 15u;
            sv.base = data - sv.lead;
            sv.emit_from = sv.lead + r.carry0;
            sv.n_total = sv.lead + r.text.size();
            sv.suppress = r.wraps.empty() ? nullptr : r.wraps.data();
            sv.n_suppress = r.wraps.size();
            if (k == 30) emul_part<true>(sv, k, km, &bases);
            else emul_part<false>(sv, k, km, &bases);
            r.text.clear(); r.carry0 = 0; r.wraps.clear();
        }
        if (any) rounds++;
        feed = 0;
    };
    size_t at = 0;
    while (at < n) {
        bool ends_line = true;
        const size_t take = next_piece(buf + at, n - at, piece_cap, ls.open, &ends_line);
        if (R[(size_t)feed].text.size() + take > batch_bytes) {
            run_round();
            char look[kCarry];
            RankText &r0 = R[(size_t)feed];
            const size_t carry = ls.carry_into(look, r0.wraps);
            if (carry) { r0.text.assign(look, look + carry); r0.carry0 = carry; }
        }
        RankText &r = R[(size_t)feed];
        std::vector<uint64_t> w;
        ls.piece(buf + at, take, ends_line, w);
        for (uint64_t p : w) r.wraps.push_back(p + r.text.size());
        r.text.insert(r.text.end(), buf + at, buf + at + take);
        if (ends_line) feed = (feed + 1) % n_ranks;
        at += take;
    }
    run_round();
    for (uint64_t i = 0; i < km.size() && i < cap; i++) out[i] = km[i];
    if (n_bases) *n_bases = bases;
    if (n_rounds) *n_rounds = rounds;
    return km.size();
}

/* The device's reading of PACKED input (qm2_device.hip.h: load16_packed_raw, unpack16) on n_blocks blocks as the host
 * packers write them (quick-mer2_amd/csrc/qm2_pack.c); `wraps` as positions.  *n_positions = what the kernels add to
 * n_bases before the host's line ends are taken off. */
uint64_t qm2e_extract_packed(const void *blocks, uint64_t n_blocks, uint32_t k, const uint64_t *wraps, uint64_t n_wraps,
                             uint64_t *out, uint64_t cap, uint64_t *n_positions)
{
    std::vector<uint8_t> dev(n_blocks * 24 + 64, 0xFF);       /* (a poisoned tail behind the last block) */
    uint8_t *data = (uint8_t *)(((uintptr_t)dev.data() + 15) & ~(uintptr_t)15);
    memcpy(data, blocks, n_blocks * 24);
    SeqView sv;
    sv.base = data;
    sv.lead = 0;
    sv.emit_from = 0;
    sv.n_total = n_blocks * 64;
    sv.suppress = n_wraps ? wraps : nullptr;
    sv.n_suppress = n_wraps;
    sv.packed = 1;
    std::vector<uint64_t> km;
    uint64_t bases = 0;
    if (n_blocks) {
        if (k == 30) emul_part<true, true>(sv, k, km, &bases);
        else emul_part<false, true>(sv, k, km, &bases);
    }
    for (uint64_t i = 0; i < km.size() && i < cap; i++) out[i] = km[i];
    if (n_positions) *n_positions = bases;
    return km.size();
}

uint32_t qm2e_djb32(uint64_t kmer) { return djb32(kmer); }

/* bucket records (qm2_device.hip.h): the two words of n k-mers */
void qm2e_records(const uint64_t *kmers, uint64_t n, uint32_t lgH, uint32_t lg_s, uint32_t *r, uint32_t *w)
{
    for (uint64_t i = 0; i < n; i++) { r[i] = rec_r(kmers[i]); w[i] = rec_w(kmers[i], djb32(kmers[i]), lgH, lg_s); }
}
uint32_t qm2e_rec_ext_bits(uint32_t k, uint32_t lgH) { return rec_ext_bits(k, lgH); }
/* the table image of a whole table (what table_image_kernel makes), and image words as bucket_count's loaders stage them */
void qm2e_table_image(const uint64_t *keys, uint64_t H, uint32_t lg_s, uint32_t *img_r, uint32_t *img_w)
{
    uint32_t lgH = 0;
    while ((1ull << lgH) < H) lgH++;
    for (uint64_t i = 0; i < H; i++) {
        img_r[i] = keys[i] ? rec_r(keys[i]) : 0u;
        img_w[i] = rec_image_w(keys[i], i, (uint32_t)(H - 1), lgH, lg_s);
    }
}
void qm2e_stage_words(const uint32_t *w, uint64_t n, int halo, int down, uint32_t lg_s, uint32_t *out, uint8_t *empty)
{
    for (uint64_t i = 0; i < n; i++) { out[i] = rec_staged_w(w[i], halo != 0, down != 0, lg_s); empty[i] = rec_staged_empty(out[i], lg_s); }
}

/* the buffer layouts of a partition level (slot_of): where k-mer e of region r goes */
uint64_t qm2e_slot_of(int segmented, uint64_t region, uint32_t e, uint32_t cap, uint32_t lg_row)
{
    return segmented ? slot_of<true>(region, e, cap, lg_row) : slot_of<false>(region, e, cap, lg_row);
}
uint32_t qm2e_seg_kmers(void) { return kSegKmers; }

}
