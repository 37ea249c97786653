// tools/micro/ring_probe.hip — does a write -> read round trip through a small ring stay on the die (256 MiB Infinity Cache)?
//
// The question behind EXPERIMENTS.md section 7 ("take the level-2 round trip out of HBM"): the k-mers of one level-1 parent
// (~47 MB at 50 M reads per batch) are written by split_kmer and read once by bucket_count.  If parents were processed a
// few at a time through a ring of level-2 buffers, would the ring's lines be served from the Infinity Cache instead of
// HBM — and how much other traffic between the write and the read (the parent's own level-1 reads, the 48 MB table
// slice) does that survive?  This program only moves bytes; it has none of the product's code.
//
//   for every "parent" p (chunk of C MB), ring of S slots:
//       W  write  C MB   into slot p % S               (plain 16-byte stores, every workgroup a contiguous piece)
//       X  stream E MB   of an unrelated 8 GB buffer   (the traffic between the write and the read; plain or nontemporal loads)
//       R  read   C MB   of slot p % S                 (16-byte loads, summed)
//   as separate launches on one stream (kernel boundaries, like per-parent-group launches of the existing kernels), or
//   as ONE launch per parent in which the workgroups of one half of the chip write parent p while the other half reads
//   parent p - 1 ("fused": no kernel boundary between a slot's write and its read, but S >= 2).
//
// Output: one line per configuration: ms per parent and GB/s over the bytes moved (W + X + R).
// S * C well under 256 MiB vs S * C = 4 GB (never re-used before eviction) is the comparison that answers the question.
//
// build: hipcc -O3 --offload-arch=gfx950 -o tools/micro/ring_probe tools/micro/ring_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// mode 0: write, 1: read (plain), 2: read (nontemporal)
template <int MODE>
__device__ __forceinline__ void sweep(u32x4 *p, uint64_t n16, uint32_t wg, uint32_t n_wg, unsigned long long *sink, uint32_t tag)
{
    // workgroup `wg` of `n_wg` takes a contiguous piece, 1 KiB per wave instruction
    const uint64_t per = (n16 + n_wg - 1) / n_wg, lo = (uint64_t)wg * per, hi = lo + per < n16 ? lo + per : n16;
    uint32_t acc = 0;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        if (MODE == 0) p[i] = u32x4{(uint32_t)i, tag, (uint32_t)(i >> 32), tag ^ 0x5A5A5A5Au};
        else {
            const u32x4 v = MODE == 2 ? __builtin_nontemporal_load(p + i) : p[i];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (MODE != 0 && acc == 0x12345678u) atomicAdd(sink, 1ull);          // keeps the loads alive
}

template <int MODE>
__global__ __launch_bounds__(1024) void sweep_kernel(u32x4 *p, uint64_t n16, unsigned long long *sink, uint32_t tag)
{
    sweep<MODE>(p, n16, blockIdx.x, gridDim.x, sink, tag);
}

// one launch per parent: even workgroups write slot `w`, odd workgroups read slot `r` (written by the previous launch)
template <int XMODE>
__global__ __launch_bounds__(1024) void fused_kernel(u32x4 *w, u32x4 *r, u32x4 *x, uint64_t n16, uint64_t x16, unsigned long long *sink, uint32_t tag)
{
    const uint32_t half = gridDim.x / 2, me = blockIdx.x >> 1;
    if (blockIdx.x & 1u) {
        if (x16) sweep<XMODE>(x, x16, me, half, sink, tag);               // e.g. the table slice the reader also streams
        sweep<1>(r, n16, me, half, sink, tag);
    } else sweep<0>(w, n16, me, half, sink, tag);
}

int main(int argc, char **argv)
{
    const int parents = argc > 1 ? atoi(argv[1]) : 400;
    const uint64_t MB = 1000000ull;
    unsigned long long *sink = nullptr;
    OK(hipMalloc((void **)&sink, 8));
    OK(hipMemset(sink, 0, 8));
    const uint64_t ring_bytes = 4096 * MB, extra_bytes = 8192 * MB;
    u32x4 *ring = nullptr, *extra = nullptr;
    OK(hipMalloc((void **)&ring, ring_bytes));
    OK(hipMalloc((void **)&extra, extra_bytes));
    OK(hipMemset(ring, 1, ring_bytes));
    OK(hipMemset(extra, 2, extra_bytes));
    hipStream_t s;
    OK(hipStreamCreate(&s));
    hipEvent_t a, b;
    OK(hipEventCreate(&a));
    OK(hipEventCreate(&b));
    struct Cfg { const char *name; uint64_t chunk_mb; uint32_t slots; uint64_t extra_mb; int xmode; int fused; };
    std::vector<Cfg> cfgs;
    for (uint64_t c : {47ull, 94ull, 188ull, 376ull}) {
        cfgs.push_back({"separate launches, no other traffic    ", c, 2, 0, 1, 0});
        cfgs.push_back({"separate launches, no other traffic    ", c, (uint32_t)(4096 / c), 0, 1, 0});
    }
    for (int xm : {1, 2}) {
        cfgs.push_back({xm == 1 ? "separate, 48 MB plain stream between   " : "separate, 48 MB nt stream between      ", 47, 2, 48, xm, 0});
        cfgs.push_back({xm == 1 ? "separate, 96 MB plain stream between   " : "separate, 96 MB nt stream between      ", 47, 2, 96, xm, 0});
        cfgs.push_back({xm == 1 ? "separate, 192 MB plain stream between  " : "separate, 192 MB nt stream between     ", 47, 2, 192, xm, 0});
        cfgs.push_back({xm == 1 ? "separate, 96 MB plain stream between   " : "separate, 96 MB nt stream between      ", 47, 87, 96, xm, 0});
    }
    for (uint64_t c : {47ull, 94ull}) {
        cfgs.push_back({"fused halves (W p | R p-1), no stream  ", c, 2, 0, 1, 1});
        cfgs.push_back({"fused halves (W p | R p-1), no stream  ", c, (uint32_t)(4096 / c), 0, 1, 1});
        cfgs.push_back({"fused halves, reader also streams 48 MB", c, 2, 48, 1, 1});
        cfgs.push_back({"fused halves, reader streams 48 MB nt  ", c, 2, 48, 2, 1});
        cfgs.push_back({"fused halves, reader also streams 48 MB", c, (uint32_t)(4096 / c), 48, 1, 1});
    }
    printf("%d parents per configuration; ring allocation %.1f GB, unrelated stream buffer %.1f GB\n", parents, ring_bytes / 1e9, extra_bytes / 1e9);
    for (int rep = 0; rep < 2; rep++)
    for (const Cfg &c : cfgs) {
        const uint64_t n16 = c.chunk_mb * MB / 16, x16 = c.extra_mb * MB / 16;
        const uint64_t slot16 = (n16 + 4095) & ~4095ull;
        if ((uint64_t)c.slots * slot16 * 16 > ring_bytes) continue;
        uint64_t xat = 0;
        OK(hipEventRecord(a, s));
        for (int p = 0; p < parents; p++) {
            u32x4 *slot = ring + (uint64_t)(p % c.slots) * slot16;
            u32x4 *xs = extra + xat;
            if (x16) { xat += x16; if ((xat + x16) * 16 > extra_bytes) xat = 0; }
            if (!c.fused) {
                hipLaunchKernelGGL(sweep_kernel<0>, dim3(256), dim3(1024), 0, s, slot, n16, sink, (uint32_t)p);
                if (x16) {
                    if (c.xmode == 2) hipLaunchKernelGGL(sweep_kernel<2>, dim3(256), dim3(1024), 0, s, xs, x16, sink, (uint32_t)p);
                    else hipLaunchKernelGGL(sweep_kernel<1>, dim3(256), dim3(1024), 0, s, xs, x16, sink, (uint32_t)p);
                }
                hipLaunchKernelGGL(sweep_kernel<1>, dim3(256), dim3(1024), 0, s, slot, n16, sink, (uint32_t)p);
            } else {
                u32x4 *prev = ring + (uint64_t)((p + c.slots - 1) % c.slots) * slot16;
                if (c.xmode == 2) hipLaunchKernelGGL(fused_kernel<2>, dim3(256), dim3(1024), 0, s, slot, prev, xs, n16, x16, sink, (uint32_t)p);
                else hipLaunchKernelGGL(fused_kernel<1>, dim3(256), dim3(1024), 0, s, slot, prev, xs, n16, x16, sink, (uint32_t)p);
            }
        }
        OK(hipGetLastError());
        OK(hipEventRecord(b, s));
        OK(hipEventSynchronize(b));
        float ms = 0;
        OK(hipEventElapsedTime(&ms, a, b));
        const double bytes = (double)parents * (2.0 * c.chunk_mb + c.extra_mb) * 1e6;
        printf("%s chunk %3llu MB x %3u slots (ring %4llu MB)%s: %7.1f us per parent, %6.0f GB/s over W+X+R bytes\n", c.name,
               (unsigned long long)c.chunk_mb, c.slots, (unsigned long long)(c.chunk_mb * c.slots), rep ? " [2]" : "    ",
               1e3 * ms / parents, bytes / (ms * 1e-3) / 1e9);
    }
    return 0;
}
