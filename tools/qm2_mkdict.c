/*
 * qm2_mkdict.c — dictionary writer: `quicKmer2 search -e 0` re-stated.
 *
 * TOOLING, not part of the count path: `search` is upstream of it and stays with
 * the reference.  This exists so that tests and end-to-end runs on a box without
 * the reference binary can make format-exact <ref.fa>.qm / .qgc / .bed files; it
 * is verified byte for byte against the real `search -k K -s SIZE -e 0 -w W
 * [-c ctrl.bed] ref.fa` (tests/test_mkdict_cpu.py, md5s in tests/golden/).
 *
 * Written from the behaviour of QuicKmer.c (cited by line), quirks included,
 * because byte-identical output depends on them:
 *   pass 1  hash_from_fasta :824-923      every canonical k-mer (k-mer 0 excluded, :864) is
 *                                         inserted by the probe rule of :866-876 with a
 *                                         saturating occurrence count (:888)
 *   prune   main_search :1218-1231        occurrences > 1 are deleted
 *   rehash  main_search :1232-1271        in place, outside-in, with the 1024-entry side
 *                                         buffer for keys whose home is in the other half
 *   pass 2  dump_kmer_list :925-1073      genome-order chain, GC/control word per k-mer
 *                                         (.qgc), window table (.bed)
 *   write   main_search :1284-1299
 *   grow    hash_from_fasta :891-895,     a table that passes 80 % fill at the end of a FASTA line is doubled and
 *           Resize_hash_table :738-779    re-placed in place, upper half of the old table first (downwards), then the
 *                                         lower half (upwards): the order decides where colliding keys end up
 * Not supported (the reference would): edit-distance filtering (-e 1/2).
 *
 * usage: qm2-mkdict [-k K] [-s SIZE[K|M|G]] [-w W] [-d D] [-c ctrl.bed] ref.fa
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static uint64_t g_k = 30, g_mask = ((uint64_t)1 << 60) - 1, g_H = 0x2000000;   /* :14-16 */
static uint64_t *g_keys;

static uint64_t djb(uint64_t x)                                    /* :66-76 */
{
    uint64_t h = 5381;
    for (int i = 0; i < 8; i++) { h = h * 33 + (x & 0xFF); x >>= 8; }
    return h;
}

/* :90-99 — slot where `km` sits, or the empty slot its walk ends on */
static uint64_t find_slot(uint64_t km, int *found)
{
    uint64_t i = djb(km) & (g_H - 1);
    const uint64_t step = (i & (g_H >> 1)) ? (uint64_t)-1 : 1;
    while (g_keys[i] && g_keys[i] != km) i += step;
    if (found) *found = g_keys[i] == km;
    return i;
}

/* rolling encoder state shared by both passes (:854-863 and :1003-1017) */
typedef struct { uint64_t fwd, rc; uint8_t run; } roll;

static inline void roll_reset(roll *r) { r->fwd = 0; r->rc = 0; r->run = 0; }

/* feeds one character; returns the canonical k-mer, or 0 when none is complete
 * (k-mer 0 itself is never used by search, :864 / :1018) */
static inline uint64_t roll_push(roll *r, unsigned char c)
{
    const uint64_t code = (c >> 1) & 3;
    r->fwd = (r->fwd << 2) | code;
    r->rc = (r->rc | (((code - 2) & 3) << 60)) >> 2;
    uint64_t km = r->fwd & g_mask;
    if (km > r->rc) km = r->rc;
    if (r->run < g_k) r->run++;
    return r->run == g_k ? km : 0;
}

/* ---- growth (:738-779) ------------------------------------------------------ */

/* Double the table and re-place every key under the new size, in the reference's order.  A key moves only if
 * its walk in the doubled table does not end on itself (Find_hash returns the slot of the key or of the first
 * empty slot on the way, :90-99); its occurrence count moves with it.  The reference leaves the new half of
 * Kmer_occr to realloc (:749): fresh pages, i.e. zeros, stated here. */
static int grow(uint8_t **occ_p)
{
    const uint64_t old = g_H, nw = old << 1;
    printf("Resize %f\n", (float)nw / old);                        /* :739 */
    if (nw > ((uint64_t)1 << 32)) { puts("hash size above 2^32 does not fit the uint32 chain"); return 1; }
    uint64_t *nk = realloc(g_keys, nw * sizeof(uint64_t));
    if (!nk) { puts("Realloc failed!\n"); return 1; }
    g_keys = nk;
    uint8_t *no = realloc(*occ_p, nw);
    if (!no) { puts("Realloc failed!\n"); return 1; }
    *occ_p = no;
    puts("Allocation success");
    memset(g_keys + old, 0, sizeof(uint64_t) * (nw - old));        /* :755 */
    memset(no + old, 0, nw - old);
    g_H = nw;
    for (uint64_t index = old - 1; index >= (old >> 1); index--) {   /* :757-771 */
        if (g_keys[index]) {
            const uint64_t to = find_slot(g_keys[index], NULL);
            if (to != index) { g_keys[to] = g_keys[index]; g_keys[index] = 0; no[to] = no[index]; no[index] = 0; }
        }
        if (index == 0) break;                                     /* (old = 1 cannot happen: H >= 2) */
    }
    for (uint64_t index = 0; index < (old >> 1); index++) {        /* :772-786 */
        if (g_keys[index]) {
            const uint64_t to = find_slot(g_keys[index], NULL);
            if (to != index) { g_keys[to] = g_keys[index]; g_keys[index] = 0; no[to] = no[index]; no[index] = 0; }
        }
    }
    return 0;
}

/* ---- pass 1 --------------------------------------------------------------- */

static int pass1(FILE *fa, uint8_t **occ_p)
{
    uint8_t *occ = *occ_p;
    char line[200];
    roll r;
    roll_reset(&r);
    uint64_t distinct = 0;
    uint32_t worst = 0;
    uint64_t hist_n = 0;
    double hist_sum = 0;
    while (fgets(line, sizeof line, fa) && line[0]) {              /* :834 */
        if (line[0] == '>') {                                      /* :837-844 */
            roll_reset(&r);
            printf("%s", line);
            continue;
        }
        for (const char *p = line; *p && *p != '\n'; p++) {        /* :845 */
            if (*p == 'N') { roll_reset(&r); continue; }           /* :847-853 */
            const uint64_t km = roll_push(&r, (unsigned char)*p);
            if (!km) continue;
            uint64_t i = djb(km) & (g_H - 1);
            const uint64_t step = (i & (g_H >> 1)) ? (uint64_t)-1 : 1;
            uint32_t moved = 0;
            while (g_keys[i] && g_keys[i] != km) { i += step; moved++; }
            if (!g_keys[i]) {
                distinct++;
                if (moved > worst) { worst = moved; printf("Worst %u\n", worst); }
                hist_n++;
                hist_sum += moved;
                g_keys[i] = km;
            }
            if (occ[i] < 255) occ[i]++;                            /* :888 */
        }
        if ((double)distinct > 0.8 * (double)g_H) {                /* :891-895 */
            if (grow(occ_p)) return 1;
            occ = *occ_p;
            hist_n = 0; hist_sum = 0; worst = 0;
        }
    }
    printf("Average %f, fill %f%% \n", hist_n ? hist_sum / (double)hist_n : 0.0, ((float)distinct * 100) / g_H);
    uint64_t once = 0;
    for (uint64_t i = 0; i < g_H; i++) once += occ[i] == 1;
    printf("Uniq count %lu, total %lu\n", (unsigned long)once, (unsigned long)distinct);
    return 0;
}

/* ---- prune + in-place rehash (:1218-1271) ---------------------------------- */

static void prune_and_rehash(const uint8_t *occ)
{
    uint64_t gone = 0;
    for (uint64_t i = 0; i < g_H; i++)
        if (occ[i] > 1) { g_keys[i] = 0; gone++; }
    printf("%u kmers deleted\n", (unsigned)gone);
    /* Survivors are re-placed walking the table from both ends towards the middle.  A key whose
     * home lies in the OTHER half must wait until that half has been cleaned up: it is parked
     * (at most 1024 of them, as in the reference) and inserted last, newest first. */
    uint64_t parked[1024];
    uint32_t n_parked = 0;
    const uint64_t half = g_H >> 1;
    for (uint64_t lo = 0; lo < half; lo++) {
        const uint64_t hi = g_H - 1 - lo;
        if (g_keys[lo]) {
            const uint64_t km = g_keys[lo];
            g_keys[lo] = 0;
            if ((djb(km) & (g_H - 1)) >= half) parked[n_parked++] = km;
            else g_keys[find_slot(km, NULL)] = km;
        }
        if (g_keys[hi]) {
            const uint64_t km = g_keys[hi];
            g_keys[hi] = 0;
            if ((djb(km) & (g_H - 1)) < half) parked[n_parked++] = km;
            else g_keys[find_slot(km, NULL)] = km;
        }
    }
    while (n_parked) {
        const uint64_t km = parked[--n_parked];
        g_keys[find_slot(km, NULL)] = km;
    }
}

/* ---- pass 2 ---------------------------------------------------------------- */

/* One contig of the FASTA, newlines removed, as dump_kmer_list holds it (:956-969).  The
 * reference reads with fgets into one big buffer; the only observable consequences are kept:
 * the text is NUL-terminated, and for the LAST contig of the file the final newline stays in
 * front of that NUL and is later treated as one more base (it ends the reading loop without
 * ever being overwritten). */
typedef struct { char *s; size_t cap; uint32_t n_bases; } contig;

static int next_contig(FILE *fa, contig *c, char *name, size_t name_cap)
{
    if (!fgets(name, (int)name_cap, fa)) return 0;                 /* :943 */
    uint32_t at = 0;
    for (;;) {
        if (feof(fa)) break;                                       /* :956 */
        if ((size_t)at + 256 > c->cap) {
            c->cap = c->cap ? c->cap * 2 : ((size_t)64 << 20);
            c->s = realloc(c->s, c->cap);
            if (!c->s) { puts("Memory allocation failed"); exit(1); }
        }
        const long before = ftell(fa);
        if (!fgets(c->s + at, 200, fa)) {
            /* end of file: the newline of the last line is still at c->s[at] (:958-967) */
            break;
        }
        const uint8_t got = (uint8_t)(ftell(fa) - before);
        if (c->s[at] == '>') {                                     /* next contig: give the line back */
            fseek(fa, -(long)got, SEEK_CUR);
            c->s[at] = 0;
            break;
        }
        at += got - 1;
        if (c->s[at] != '\n') at++;                                /* a last line without newline */
    }
    c->n_bases = at;
    return 1;
}

static int write_dict(FILE *fa, FILE *qm, FILE *bed, FILE *ctrl, FILE *qgc, uint32_t win_kmers,
                      uint8_t e_byte, uint8_t d_byte)
{
    uint32_t *chain = calloc(g_H, sizeof(uint32_t));               /* :1273 */
    uint16_t *gcbuf = qgc ? malloc(((size_t)1 << 20) * sizeof(uint16_t)) : NULL;
    if (!chain || (qgc && !gcbuf)) { puts("Memory allocation failed"); return 1; }
    puts("Generate chaining table");
    const uint16_t gc_win = 400;                                   /* :1280 */
    const uint16_t wsize = (uint16_t)win_kmers;                    /* uint16_t parameter, :926 */
    const uint32_t lead = (uint32_t)((gc_win - g_k) / 2), lag = (uint32_t)((gc_win + g_k) / 2);
    uint32_t gc_fill = 0;
    uint64_t n_out = 0, first = 0, last = 0;
    contig c = {NULL, 0, 0};
    char header[200];
    char ctrl_chr[200];
    rewind(fa);                                                    /* :928 */
    while (next_contig(fa, &c, header, sizeof header)) {
        char *name = header + 1;                                   /* :930, :979 */
        name[strlen(name) - 1] = 0;
        printf("%s %i\n", name, (int)c.n_bases);
        /* GC window: counts over the ~gc_win bases centred on the k-mer, kept by adding the base
         * `lead` ahead and dropping the one `lag` behind (:982-1002).  A base "is GC" when bit 1
         * of its byte is set — true for C, G and also N, so the N counter never moves. */
        uint16_t win_bp = 0, gc_bp = 0, n_bp = 0;
        for (uint32_t i = 0; i < lead; i++) {
            if (c.s[i] & 2) gc_bp++; else if (c.s[i] == 'N') n_bp++;
            win_bp++;
        }
        int ctrl_started = 0, ctrl_absent = 0;
        uint32_t ctrl_s = 0, ctrl_e = 0;
        uint64_t win_start = 0;
        uint32_t win_first = (uint32_t)n_out;
        roll r;
        roll_reset(&r);
        for (uint32_t pos = 0; c.s[pos]; pos++) {                  /* :989 */
            if (pos + lead < c.n_bases) {
                const char a = c.s[pos + lead];
                if (a & 2) gc_bp++; else if (a == 'N') n_bp++;
                win_bp++;
            }
            if (pos >= lag) {
                const char d = c.s[pos - lag];
                if (d & 2) gc_bp--; else if (d == 'N') n_bp--;
                win_bp--;
            }
            if (c.s[pos] == 'N') { roll_reset(&r); continue; }     /* :1004-1008 */
            const uint64_t km = roll_push(&r, (unsigned char)c.s[pos]);
            if (!km) continue;
            int found;
            const uint64_t slot = find_slot(km, &found);
            if (!found) continue;                                  /* :1021 */
            if (qgc) {
                uint16_t word = (uint16_t)((n_bp + gc_bp * 2) * 200 / win_bp);   /* :1025 */
                if (!ctrl_started) {                               /* :1027-1032 */
                    fseek(ctrl, 0, SEEK_SET);
                    while (fscanf(ctrl, "%s\t%d\t%d", ctrl_chr, (int *)&ctrl_s, (int *)&ctrl_e) == 3 &&
                           (ctrl_absent = strcmp(ctrl_chr, name)))
                        ;
                    ctrl_started = 1;
                } else {                                           /* :1033-1038 */
                    while (pos > ctrl_e && !ctrl_absent &&
                           fscanf(ctrl, "%s\t%d\t%d", ctrl_chr, (int *)&ctrl_s, (int *)&ctrl_e) == 3)
                        if (strcmp(ctrl_chr, name)) { ctrl_absent = 1; break; }
                }
                if (!ctrl_absent && (uint64_t)pos + 1 - g_k > ctrl_s) word |= 0x8000;     /* :1039-1040 */
                gcbuf[gc_fill++] = word;
                if (gc_fill == (1u << 20)) { fwrite(gcbuf, 2, gc_fill, qgc); gc_fill = 0; }
            }
            if (!n_out) first = slot; else chain[last] = (uint32_t)slot;         /* :1049-1051 */
            last = slot;
            n_out++;
            if (bed && n_out % wsize == 0) {                       /* :1054-1059 */
                fprintf(bed, "%s\t%u\t%u\t%u\t%u\n", name, (unsigned)win_start, (unsigned)pos, (unsigned)win_first,
                        (unsigned)n_out);
                win_start = pos;
                win_first = (uint32_t)n_out;
            }
        }
    }
    if (qgc) { fwrite(gcbuf, 2, gc_fill, qgc); fclose(qgc); }      /* :1066-1069 */
    chain[last] = (uint32_t)first;                                 /* :1070 */
    printf("Total output %u k-mers\n", (unsigned)n_out);
    puts("Writing index file");
    rewind(qm);                                                    /* :1284-1299 */
    const uint8_t kb = (uint8_t)g_k;
    fwrite("QM11", 1, 4, qm);
    fwrite(&kb, 1, 1, qm);
    fwrite(&e_byte, 1, 1, qm);
    fwrite(&d_byte, 1, 1, qm);
    fwrite(&d_byte, 1, 1, qm);
    fwrite(&g_H, 8, 1, qm);
    fwrite(&first, 8, 1, qm);
    fwrite(g_keys, 8, g_H, qm);
    puts("Writing chain list");
    fwrite(chain, 4, g_H, qm);
    fclose(qm);
    free(chain); free(gcbuf); free(c.s);
    puts("Kmer search finished!");
    return 0;
}

static void usage(void)
{
    puts("\nqm2-mkdict [Options] ref.fa   (the reference's `quicKmer2 search -e 0`, byte-identical output)\n\nOptions:");
    puts("-k [num]\tSize of K-mer. 3-31. Default 30");
    puts("-s [num]\tSize of hash dictionary. Can use suffix G,M,K");
    puts("-d [num]\tstored in the header only (edit-distance depth threshold). Default 100");
    puts("-w [num]\tOutput window definition size. Default 1000");
    puts("-c [filename]\tInput bedfile for GC control regions");
}

int main(int argc, char **argv)
{
    uint32_t win = 1000;
    uint8_t d_byte = 100;
    FILE *ctrl = NULL;
    int opt;
    if (argc < 2) { usage(); return 1; }
    while ((opt = getopt(argc, argv, "hk:t:s:e:d:w:c:")) != -1) {   /* :1103 */
        switch (opt) {
        case 'h': usage(); return 1;
        case 'k':
            g_k = (uint8_t)atoi(optarg);
            if (g_k < 1 || g_k > 31) { puts("k must be 1..31"); return 1; }
            g_mask = ((uint64_t)1 << (g_k * 2)) - 1;
            printf("[Option] Set %i-mer\n", (int)g_k);
            break;
        case 't': break;
        case 's': {                                                 /* :1119-1136 */
            const size_t len = strlen(optarg);
            uint64_t v;
            switch (len ? optarg[len - 1] : 0) {
            case 'G': v = (uint64_t)atoi(optarg) << 30; break;
            case 'M': v = (uint64_t)(atoi(optarg) << 20); break;
            case 'K': v = (uint64_t)(atoi(optarg) << 10); break;
            default: v = (uint64_t)atoi(optarg);
            }
            g_H = (uint64_t)1 << (uint8_t)ceil(log2((double)v));
            printf("[Option] Set hash space 0x%lX\n", (unsigned long)g_H);
            break;
        }
        case 'c':
            ctrl = fopen(optarg, "r");
            printf("[Option] Control bedfile %s\n", optarg);
            break;
        case 'w': win = (uint32_t)atoi(optarg); printf("[Option] Write window size %s\n", optarg); break;
        case 'e':
            if (atoi(optarg) != 0) { puts("only -e 0 is supported by this writer"); return 1; }
            printf("[Option] Edit distance %i\n", 0);
            break;
        case 'd': d_byte = (uint8_t)atoi(optarg); break;
        default: puts("Option error, check help"); usage(); return 1;
        }
    }
    if (optind >= argc) { usage(); return 1; }
    if (g_H > ((uint64_t)1 << 32)) { puts("hash size above 2^32 does not fit the uint32 chain"); return 1; }
    const char *ref = argv[argc - 1];
    g_keys = calloc(g_H, sizeof(uint64_t));                        /* :1163-1164 */
    uint8_t *occ = calloc(g_H, 1);
    if (!g_keys || !occ) { puts("Memory allocation failed"); return 1; }
    char *path = malloc(strlen(ref) + 8);
    sprintf(path, "%s.qm", ref);
    FILE *qm = fopen(path, "w");
    if (!qm) { puts("File creation failed"); return 1; }
    sprintf(path, "%s.bed", ref);
    FILE *bed = fopen(path, "w");
    if (!bed) { puts("Window creation failed"); return 1; }
    FILE *fa = fopen(ref, "r");
    if (!fa) { puts("Input open fail"); return 1; }
    if (pass1(fa, &occ)) return 1;
    prune_and_rehash(occ);
    free(occ);
    FILE *qgc = NULL;
    if (ctrl) {
        sprintf(path, "%s.qgc", ref);
        qgc = fopen(path, "w");
    }
    const int rc = write_dict(fa, qm, bed, ctrl, qgc, win, 0, d_byte);
    fclose(bed);
    fclose(fa);
    free(path);
    free(g_keys);
    return rc;
}
