// tools/micro/front_probe.hip — how fast does the memory system take the text level's WRITE pattern, by buffer layout?
//
// split_text's store pattern without the rest of the kernel: every workgroup appends runs of 16 k-mers (128 B) to
// each of 1024 "fronts" in turn, a run's place reserved with an atomic on the front's cursor, four fronts per
// 64-lane store.  Layouts:
//   child  S   front f occupies [f * S, (f + 1) * S)                       (S = k-mers per front)
//   seg    lg  segments of 2^lg k-mers laid out [segment][front]
//   stag   S   child-major, each front started a pseudo-random multiple of 512 B (< 2 MiB) further on
//   xcd    S   child-major, every front cut into 8 sub-regions of S/8, one per XCD (workgroup b runs on XCD b % 8):
//              a cache line is only ever written through ONE L2
// Prints ms and GB/s per layout, `reps` times each.
//
// build: hipcc -O3 --offload-arch=gfx950 -o tools/micro/front_probe tools/micro/front_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Layout { int mode; uint32_t lg; uint64_t stride; };    // mode 0 child, 1 seg, 2 stag

__device__ __forceinline__ uint64_t place(const Layout &L, uint32_t f, uint32_t e)
{
    if (L.mode == 3) return (uint64_t)f * (L.stride >> 3) + e;      /* f already includes the XCD: xcd * 1024 + front */
    if (L.mode == 1) return ((uint64_t)(e >> L.lg) << (10 + L.lg)) + (((uint64_t)f << L.lg) | (e & ((1u << L.lg) - 1u)));
    uint64_t at = (uint64_t)f * L.stride + e;
    if (L.mode == 2) at += ((f * 0x9E3779B1u) >> 8) & 0x3FFC0u;
    return at;
}

__global__ __launch_bounds__(1024) void fronts_kernel(uint64_t *buf, uint32_t *cursors, Layout L, uint32_t chunks, uint32_t run, uint32_t ragged)
{
    __shared__ uint32_t base[2][1024];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t per = 64u / run;                       // fronts per store instruction
    const uint32_t nf = 16384u / run;                     // fronts a chunk appends to (run 16: all 1024)
    for (uint32_t c = 0; c < chunks; c++) {
        // a different rotation per workgroup and chunk, so that the workgroups are not in step
        const uint32_t rot = (blockIdx.x * 37u + c * 101u) & 1023u;
        // all of the chunk's reservations first, like the real kernel (one atomic per thread), then the stores
        // ragged: run lengths run/2 .. run (mean 3/4 run), so that runs start and end anywhere in a sector, as the real ones do
        auto len_of = [&](uint32_t i) { const uint32_t raw = run / 2u + ((i * 0x9E3779B1u + c * 0x85EBCA6Bu + blockIdx.x * 0xC2B2AE35u) >> 20) % (run / 2u + 1u);
                                        return ragged ? (raw + ragged - 1u) / ragged * ragged : run; };   /* (ragged = 1: as is; 4, 8: padded to that many k-mers) */
        const uint32_t sub = L.mode == 3 ? (blockIdx.x & 7u) << 10 : 0u;
        if (threadIdx.x < nf) base[c & 1][threadIdx.x] = atomicAdd(&cursors[sub + ((rot + threadIdx.x) & 1023u)], len_of(threadIdx.x));
        __syncthreads();
        for (uint32_t step = 0; step < nf / (16u * per); step++) {
            const uint32_t i = (step * 16u + wave) * per + lane / run;
            const uint32_t f = sub + ((rot + i) & 1023u);
            const uint32_t e = base[c & 1][i] + lane % run;
            if (lane % run < len_of(i)) buf[place(L, f, e)] = ((uint64_t)f << 32) | e;
        }
    }
}

// The other way round: thread t owns front (rot + t) of the chunk and writes its whole run itself, one k-mer (wide = 1) or
// two (wide = 2: 16-byte stores; run lengths padded to even, so every run starts 16-byte aligned) per store — every lane
// of a store instruction in a different 128-byte line, the pieces of a line arriving in consecutive instructions of
// one wave.  (What a copy-out without the per-element child lookup would look like.)
template <int WIDE>
__global__ __launch_bounds__(1024) void lanes_kernel(uint64_t *buf, uint32_t *cursors, Layout L, uint32_t chunks, uint32_t run, uint32_t ragged)
{
    typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
    for (uint32_t c = 0; c < chunks; c++) {
        const uint32_t rot = (blockIdx.x * 37u + c * 101u) & 1023u;
        const uint32_t i = threadIdx.x;
        const uint32_t raw = run / 2u + ((i * 0x9E3779B1u + c * 0x85EBCA6Bu + blockIdx.x * 0xC2B2AE35u) >> 20) % (run / 2u + 1u);
        const uint32_t len = ragged ? (raw + ragged - 1u) / ragged * ragged : run;
        const uint32_t sub = L.mode == 3 ? (blockIdx.x & 7u) << 10 : 0u;
        const uint32_t f = sub + ((rot + i) & 1023u);
        const uint32_t base = atomicAdd(&cursors[f], len);
        uint64_t *dst = buf + place(L, f, base);
        for (uint32_t e = 0; e < len; e += WIDE) {
            if (WIDE == 2) *reinterpret_cast<u64x2 *>(dst + e) = u64x2{((uint64_t)f << 32) | e, e + 1u};
            else dst[e] = ((uint64_t)f << 32) | e;
        }
    }
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 2;
    const uint32_t wgs = 256, chunks = argc > 2 ? atoi(argv[2]) : 1400;
    const uint64_t total = (uint64_t)wgs * chunks * 16384u;            // k-mers written
    const uint64_t per_front = total / 1024u;
    const uint64_t cap = ((per_front + per_front / 8u) + 262144u + (1u << 22)) & ~(uint64_t)((1u << 22) - 1u);   // whole 32 MiB segments fit
    uint64_t *buf = nullptr;
    uint32_t *cur = nullptr;
    const uint64_t alloc_elems = 1024u * (cap + 2048u);
    OK(hipMalloc((void **)&buf, alloc_elems * 8u));
    OK(hipMalloc((void **)&cur, 32768));
    printf("buffer %p, %.1f GB; %llu k-mers per launch (%.1f GB), cap %llu per front\n", (void *)buf, 1024.0 * cap * 8 / 1e9,
           (unsigned long long)total, total * 8 / 1e9, (unsigned long long)cap);
    struct { const char *name; Layout L; uint32_t run; uint32_t ragged; int lanes; } cfg[] = {
        {"per-XCD RAGGED 8..16 (rows)  ", {3, 0, cap}, 16, 1},
        {"per-XCD 8..16 lane-per-run 8B", {3, 0, cap}, 16, 1, 1},
        {"per-XCD 8..16 pad2, rows     ", {3, 0, cap}, 16, 2},
        {"per-XCD 8..16 pad2, lane 16B ", {3, 0, cap}, 16, 2, 2},
        {"per-XCD 16..32 (rows)        ", {3, 0, cap}, 32, 1},
        {"per-XCD 16..32 lane-per-run 8B", {3, 0, cap}, 32, 1, 1},
        {"per-XCD 16..32 pad2 lane 16B ", {3, 0, cap}, 32, 2, 2},
        {"child-major stride=cap      ", {0, 0, cap}, 16},
        {"child-major stride=cap+64   ", {0, 0, cap + 64}, 16},
        {"child-major stride=cap+1088 ", {0, 0, cap + 1088}, 16},
        {"staggered   stride=cap      ", {2, 0, cap}, 16},
        {"segmented   512 B           ", {1, 6, 0}, 16},
        {"segmented   4 KiB           ", {1, 9, 0}, 16},
        {"segmented   64 KiB          ", {1, 13, 0}, 16},
        {"segmented   1 MiB           ", {1, 17, 0}, 16},
        {"segmented   32 MiB          ", {1, 22, 0}, 16},
        {"child-major RAGGED 8..16    ", {0, 0, cap}, 16, 1},
        {"per-XCD     RAGGED 8..16    ", {3, 0, cap}, 16, 1},
        {"per-XCD     aligned 16      ", {3, 0, cap}, 16, 0},
        {"per-XCD RAGGED padded to 4  ", {3, 0, cap}, 16, 4},
        {"per-XCD RAGGED padded to 8  ", {3, 0, cap}, 16, 8},
        {"per-XCD RAGGED 16..32       ", {3, 0, cap}, 32, 1},
        {"per-XCD RAGGED 16..32 pad 4 ", {3, 0, cap}, 32, 4},
        {"staggered   RAGGED 8..16    ", {2, 0, cap}, 16, 1},
        {"segmented 4 KiB  RAGGED     ", {1, 9, 0}, 16, 1},
        {"segmented 64 KiB RAGGED     ", {1, 13, 0}, 16, 1},
        {"segmented 1 MiB  RAGGED     ", {1, 17, 0}, 16, 1},
        {"segmented 32 MiB RAGGED     ", {1, 22, 0}, 16, 1},
        {"child-major RAGGED 16..32   ", {0, 0, cap}, 32, 1},
        {"segmented 64 KiB RAGGED 32  ", {1, 13, 0}, 32, 1},
        {"child-major runs of 32      ", {0, 0, cap}, 32},
        {"segmented 64 KiB runs of 32 ", {1, 13, 0}, 32},
        {"child-major runs of 64      ", {0, 0, cap}, 64},
        {"segmented 64 KiB runs of 64 ", {1, 13, 0}, 64},
    };
    hipEvent_t a, b;
    OK(hipEventCreate(&a)); OK(hipEventCreate(&b));
    for (auto &c : cfg)
        for (int r = 0; r < reps; r++) {
            {   // the highest k-mer slot this layout can touch must lie inside the allocation
                const uint64_t top = c.L.mode == 1 ? ((per_front >> c.L.lg) + 1) << (10 + c.L.lg)
                                                   : c.L.mode == 3 ? 8191u * (c.L.stride >> 3) + per_front / 8u + per_front / 64u : 1023u * c.L.stride + per_front + (c.L.mode == 2 ? 0x40000u : 0u);
                if (top > alloc_elems) { printf("%s skipped: does not fit the buffer\n", c.name); break; }
            }
            OK(hipMemset(cur, 0, 32768));
            OK(hipEventRecord(a));
            // (lane-per-run: a chunk is 1024 runs whatever their length; the k-mers written are counted below)
            if (c.lanes == 1) hipLaunchKernelGGL(lanes_kernel<1>, dim3(wgs), dim3(1024), 0, 0, buf, cur, c.L, chunks * 16u / c.run, c.run, c.ragged);
            else if (c.lanes == 2) hipLaunchKernelGGL(lanes_kernel<2>, dim3(wgs), dim3(1024), 0, 0, buf, cur, c.L, chunks * 16u / c.run, c.run, c.ragged);
            else
            hipLaunchKernelGGL(fronts_kernel, dim3(wgs), dim3(1024), 0, 0, buf, cur, c.L, chunks, c.run, c.ragged);
            OK(hipEventRecord(b));
            OK(hipEventSynchronize(b));
            float ms = 0;
            OK(hipEventElapsedTime(&ms, a, b));
            printf("%s rep %d: %7.2f ms  %6.0f GB/s\n", c.name, r, ms, total * (c.ragged ? 6 : 8) / ms / 1e6);
            fflush(stdout);
        }
    return 0;
}
