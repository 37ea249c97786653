/*
 * qm2_est.c — `quicKmer2 est ref.fa sample_prefix output.bed`, the
 * consumer of count's .bin: windowed, GC-corrected copy number.  Host C, no GPU: it
 * is one streaming pass over 4 bytes per dictionary k-mer.
 *
 * Follows main_estimate (QuicKmer.c:555-685) step by step, number formats and
 * float/double conversions included, so output.bed is byte-identical given the same
 * correction curve:
 *   - mean depth from <sample>.txt: sum of (float depth * count) over the 401 GC bins
 *     divided by the total count (:622-629);
 *   - the 401-float correction curve comes from the helper `smooth_GC_mrsfast.py
 *     <sample>.txt` on PATH, read from its stdout (:632-640) — unchanged interface, so
 *     the reference's own script (or a fixed one: it breaks under numpy 2) plugs in;
 *   - one pass over .qgc/.bin in 1 MiB reads; a window [left, right) of k-mer indices
 *     from <ref>.bed is emitted when the running index reaches `right` (:651-679): the
 *     last window is therefore written only if k-mers follow it — kept as is.
 * Deviations, both towards failing loudly:
 *   - the reference does not check the helper: a missing / crashing script silently
 *     yields a bed of zeros or garbage (:638-639).  Here fewer than 401 floats is an
 *     error and nothing is written;
 *   - a missing <sample>.txt is an error (count always writes it); the reference's
 *     regeneration branch (:603-620) reads one element past what it loaded and
 *     cannot be restated exactly.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qm2_host.h"

#define EST_BUF (1024 * 1024)            /* bytes per read, as the reference's buffer_size (:11) */
#define EST_BINS 401

static void help_est(void)
{
    puts("quicKmer2 est ref.fa sample_prefix output.bed");
    puts("\tref.fa\t\tPrefix to genome reference. Program requires .qgc and .bed definition");
    puts("\tsample_prefix\tPrefix to sample.bin");
    puts("\toutput.bed\tOutput bedfile for copy number");
    puts("\nNo options available\n");
}

static FILE *open_with_suffix(const char *prefix, const char *suffix, const char *mode)
{
    const size_t n = strlen(prefix) + strlen(suffix) + 1;
    char *path = malloc(n);
    if (!path) return NULL;
    snprintf(path, n, "%s%s", prefix, suffix);
    FILE *f = fopen(path, mode);
    free(path);
    return f;
}

int qm2_main_est(int argc, char **argv)
{
    if (argc < 4) {                                                        /* :563-566, plus the argv[argc-3] it needs */
        help_est();
        return 1;
    }
    const char *ref = argv[argc - 3], *sample = argv[argc - 2], *out_path = argv[argc - 1];
    FILE *gc_ctrl = open_with_suffix(ref, ".qgc", "r");
    if (!gc_ctrl) { puts("GC control file missing."); help_est(); return 1; }
    FILE *windows = open_with_suffix(ref, ".bed", "r");
    if (!windows) { puts("Window segmentation file missing."); help_est(); fclose(gc_ctrl); return 1; }
    FILE *depth_bin = open_with_suffix(sample, ".bin", "r");
    if (!depth_bin) { puts("Sample file missing (Use prefix)."); help_est(); fclose(gc_ctrl); fclose(windows); return 1; }
    int rc = 1;
    FILE *out = NULL;
    uint16_t *gc = malloc(EST_BUF), *depth = malloc(EST_BUF);
    if (!gc || !depth) { puts("Memory allocation failed"); goto done; }

    /* mean depth over the control k-mers, from the GC curve count wrote (:622-631) */
    double total_depth = 0;
    uint64_t total_count = 0;
    {
        FILE *curve = open_with_suffix(sample, ".txt", "r");
        if (!curve) {
            puts("Depth control (.txt) not found: re-run quicKmer2 count for this sample.");
            goto done;
        }
        float percent, d;
        uint32_t cnt;
        char word[255];
        while (fscanf(curve, "%f\t%f\t%u\t%254s\n", &percent, &d, &cnt, word) == 4) {
            total_depth += d * cnt;                                         /* float product, as :626 */
            total_count += cnt;
        }
        fclose(curve);
    }
    total_depth /= total_count;
    printf("Mean sequencing depth: %.2f\n", total_depth);

    /* the correction curve: 401 floats on the helper's stdout (:632-640) */
    float correction[EST_BINS];
    {
        const size_t n = strlen(sample) + 64;
        char *cmd = malloc(n);
        if (!cmd) { puts("Memory allocation failed"); goto done; }
        snprintf(cmd, n, "smooth_GC_mrsfast.py %s.txt", sample);
        FILE *pipe_in = popen(cmd, "r");
        free(cmd);
        size_t got = pipe_in ? fread(correction, sizeof(float), EST_BINS, pipe_in) : 0;
        const int status = pipe_in ? pclose(pipe_in) : -1;
        if (got != EST_BINS || status != 0) {
            printf("GC correction helper failed (smooth_GC_mrsfast.py on PATH must print %d floats; got %zu, exit status %d)\n",
                   EST_BINS, got, status);
            goto done;
        }
    }
    out = fopen(out_path, "w");
    if (!out) { puts("Cannot open the output bed file."); goto done; }

    /* one pass: windows are [left, right) in dictionary k-mer (chain) order (:642-679) */
    double window_depth = 0.0;
    uint64_t kmer_idx = 0;
    char chrom[64], win_begin[64], win_end[64];
    uint32_t left = 0, right = 0;
    if (fscanf(windows, "%63s\t%63s\t%63s\t%u\t%u\n", chrom, win_begin, win_end, &left, &right) != 5) {
        rc = 0;                                                             /* no window: empty output, like the reference */
        goto done;
    }
    size_t read_size;
    while ((read_size = fread(gc, 1, EST_BUF, gc_ctrl)) != 0) {
        if (fread(depth, 1, read_size, depth_bin) != read_size) {
            puts("Sample .bin is shorter than the reference's .qgc: not the same dictionary?");
            goto done;
        }
        read_size >>= 1;
        size_t i = 0;
        while (i < read_size) {
            if (kmer_idx >= right) {
                window_depth /= right - left;
                window_depth /= total_depth / 2;
                fprintf(out, "%s\t%s\t%s\t%f\n", chrom, win_begin, win_end, window_depth);
                /* out of windows: the reference leaves only the inner loop (:672-673), so every further
                 * 1 MiB block prints the last window once more, divided again — kept as is */
                if (fscanf(windows, "%63s\t%63s\t%63s\t%u\t%u\n", chrom, win_begin, win_end, &left, &right) != 5) break;
                window_depth = 0.0;
            }
            if (kmer_idx < right && kmer_idx >= left) {
                const unsigned bin = gc[i] & 0x1FF;
                if (bin >= EST_BINS) { puts("GC bin outside 0..400 in the .qgc"); goto done; }
                window_depth += correction[bin] * depth[i];                 /* float product, as :676 */
            }
            i++;
            kmer_idx++;
        }
    }
    rc = 0;
done:
    if (out && fclose(out) != 0) rc = 1;
    free(gc);
    free(depth);
    fclose(gc_ctrl);
    fclose(windows);
    fclose(depth_bin);
    return rc;
}
