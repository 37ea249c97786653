// tools/micro/superkmer_probe.hip — what would a minimizer front end cost, and how many bytes per k-mer would it hand on?
//
// EXPERIMENTS.md section 7.1: the three passes of the partitioned pipeline move 8 bytes per k-mer twice; a run of
// consecutive k-mers that share their canonical minimizer (a "super-k-mer") could travel as one record, the bases
// once.  This probe measures the two numbers that decide whether that is worth a pipeline: the time of the
// minimizer pass over a batch of 150-base reads (synthetic text: 150 bases + '\n' per line, no N), and the number
// of records it would emit, for m = 9 .. 15 at k = 30.  It computes, per k-mer, the minimum over its k - m + 1
// canonical m-mers of a mixed hash of the m-mer, marks the k-mers whose minimizer differs from the previous
// k-mer's (record starts), and counts k-mers, records and the records' bytes (2 header bytes + the bases packed
// four to a byte) per 1024 hash bins of the minimizer — nothing is written out: the emission would add the
// byte-granular variant of split_text's LDS sort, on a tenth as many items.
//
// build: hipcc -O3 --offload-arch=gfx950 -o tools/micro/superkmer_probe tools/micro/superkmer_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int kThreads = 256, kPer = 16, kTile = kThreads * kPer;      // bases per workgroup and step
constexpr int kK = 30, kLine = 151;

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
    return x;
}

template <int M>
__global__ __launch_bounds__(kThreads) void minimizer_kernel(const uint8_t *text, uint64_t n, unsigned long long *totals, unsigned int *bins)
{
    constexpr int W = kK - M + 1;                                     // m-mers per k-mer
    constexpr int kHalo = kK + 2;                                     // bases behind the tile that its last k-mers need (+ the previous k-mer's)
    __shared__ uint8_t code[kTile + 2 * kHalo + 16];                  // 2-bit codes, 4 = not a base
    __shared__ uint32_t h[kTile + kHalo + 16];                        // hash of the canonical m-mer that starts here (~0u: none)
    __shared__ uint32_t s_bins[1024];
    const int tid = threadIdx.x;
    unsigned long long kmers = 0, records = 0, bytes = 0;
    for (int i = tid; i < 1024; i += kThreads) s_bins[i] = 0;
    for (uint64_t tile = blockIdx.x; tile * kTile < n; tile += gridDim.x) {
        const uint64_t base = tile * kTile;
        __syncthreads();
        // position -1 (the previous k-mer's minimizer decides whether the first k-mer of the tile starts a record) .. tile + halo
        for (int i = tid; i < kTile + 2 * kHalo; i += kThreads) {
            const int64_t p = (int64_t)base - 1 + i;
            uint8_t c = 4;
            if (p >= 0 && (uint64_t)p < n) {
                const uint8_t b = text[p];
                c = b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : b == 'T' ? 3 : 4;
            }
            code[i] = c;
        }
        __syncthreads();
        for (int i = tid; i < kTile + kHalo; i += kThreads) {
            uint32_t f = 0, r = 0, bad = 0;
#pragma unroll
            for (int j = 0; j < M; j++) {
                const uint32_t c = code[i + j];
                bad |= c >> 2;
                f = (f << 2) | (c & 3u);
                r |= ((c & 3u) ^ 3u) << (2 * j);
            }
            h[i] = bad ? ~0u : mix(f < r ? f : r);
        }
        __syncthreads();
        // thread t: the k-mers that start at tile positions 16 t .. 16 t + 15 (array index + 1), and the one before its first
        uint32_t prev = 0;
#pragma unroll
        for (int j = -1; j < kPer; j++) {
            const int i = tid * kPer + j + 1;
            uint32_t mn = ~0u, worst = 0;
#pragma unroll
            for (int w = 0; w < W; w++) { const uint32_t v = h[i + w]; mn = mn < v ? mn : v; worst = worst > v ? worst : v; }
            const bool valid = worst != ~0u && (uint64_t)((int64_t)base - 1 + i) + kK <= n && (int64_t)base - 1 + i >= 0;
            const uint32_t cur = valid ? mn : ~0u;
            if (j >= 0 && valid) {
                kmers++;
                if (cur != prev) {
                    records++;
                    atomicAdd(&s_bins[cur & 1023u], 1u);
                }
            }
            prev = cur;
        }
    }
    __syncthreads();
    for (int i = tid; i < 1024; i += kThreads) if (s_bins[i]) atomicAdd(&bins[i], s_bins[i]);
    (void)bytes;
    for (int o = 32; o; o >>= 1) { kmers += __shfl_down(kmers, o); records += __shfl_down(records, o); }
    if ((tid & 63) == 0) { atomicAdd(&totals[0], kmers); atomicAdd(&totals[1], records); }
}

template <int M>
static int run(const uint8_t *d_text, uint64_t n, unsigned long long *d_tot, unsigned int *d_bins, int reps)
{
    hipEvent_t a, b;
    OK(hipEventCreate(&a)); OK(hipEventCreate(&b));
    float best = 1e30f;
    unsigned long long tot[2] = {0, 0};
    std::vector<unsigned int> bins(1024);
    for (int r = 0; r < reps; r++) {
        OK(hipMemset(d_tot, 0, 16));
        OK(hipMemset(d_bins, 0, 4096));
        OK(hipEventRecord(a));
        hipLaunchKernelGGL(minimizer_kernel<M>, dim3(256 * 8), dim3(kThreads), 0, 0, d_text, n, d_tot, d_bins);
        OK(hipEventRecord(b));
        OK(hipEventSynchronize(b));
        float ms = 0;
        OK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    OK(hipMemcpy(tot, d_tot, 16, hipMemcpyDeviceToHost));
    OK(hipMemcpy(bins.data(), d_bins, 4096, hipMemcpyDeviceToHost));
    unsigned int mx = 0;
    for (auto v : bins) mx = v > mx ? v : mx;
    const double per = (double)tot[0] / (double)tot[1];
    // a record: 2 header bytes + (k - 1 + its k-mers) bases, four to a byte
    const double bytes_per_kmer = (2.0 + (kK - 1 + per) / 4.0 + 0.375) / per;
    printf("m = %2d: %7.2f ms per %.2f G bases (%.0f G bases/s), %.2f k-mers per record, %.2f bytes per k-mer handed on (8 today), fullest of 1024 bins %.2f x the mean\n",
           M, best, n * 150.0 / 151.0 / 1e9, n * 150.0 / 151.0 / best / 1e6, per, bytes_per_kmer, mx / ((double)tot[1] / 1024.0));
    fflush(stdout);
    return 0;
}

int main(int argc, char **argv)
{
    const uint64_t reads = argc > 1 ? strtoull(argv[1], nullptr, 10) : 20000000ull;
    const uint64_t n = reads * kLine;
    std::vector<uint8_t> text(n);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (uint64_t i = 0; i < n; i++) {
        if (i % kLine == kLine - 1) { text[i] = '\n'; continue; }
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        text[i] = "ACGT"[(s >> 33) & 3];
    }
    uint8_t *d_text = nullptr;
    unsigned long long *d_tot = nullptr;
    unsigned int *d_bins = nullptr;
    OK(hipMalloc((void **)&d_text, n));
    OK(hipMalloc((void **)&d_tot, 16));
    OK(hipMalloc((void **)&d_bins, 4096));
    OK(hipMemcpy(d_text, text.data(), n, hipMemcpyHostToDevice));
    printf("%llu reads of 150 bases (uniform random text: a genome's low-complexity stretches make the bins more uneven than this)\n", (unsigned long long)reads);
    if (run<9>(d_text, n, d_tot, d_bins, 3)) return 1;
    if (run<11>(d_text, n, d_tot, d_bins, 3)) return 1;
    if (run<13>(d_text, n, d_tot, d_bins, 3)) return 1;
    if (run<15>(d_text, n, d_tot, d_bins, 3)) return 1;
    return 0;
}
