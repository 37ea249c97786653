// tools/micro/tlb_probe.hip — what does the per-CU translation cache (UTCL1) hold?
//
// split_text writes 1024 streams that lie far apart; rocprofv3 shows one UTCL1 miss per run it writes and
// the shared UTCL2 busy for the whole kernel (EXPERIMENTS.md section 4.1).  This probe measures the UTCL1's reach:
// every wave stores 512 bytes to stream 0, then to stream 1, ... stream N-1, round and round; streams are S
// bytes apart.  Run under
//     rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum --output-format csv -d out -- ./tlb_probe
// and join the dispatches (in launch order) with the lines this program prints.
//
// build: hipcc -O3 --offload-arch=gfx950 -o tools/micro/tlb_probe tools/micro/tlb_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void streams_kernel(uint64_t *base, uint64_t stride_elems, uint32_t n_streams, uint32_t rounds,
                                                      uint32_t wg_span_elems)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // every workgroup has its own 2 KiB piece per round inside each stream; waves of a workgroup are 512 B apart
    uint64_t at = (uint64_t)blockIdx.x * wg_span_elems + wave * 64u + lane;
    for (uint32_t r = 0; r < rounds; r++) {
        for (uint32_t s = 0; s < n_streams; s++) base[(uint64_t)s * stride_elems + at] = at + s;
        at += 256u;          // next round: the next 2 KiB of this workgroup's span
    }
}

int main(int argc, char **argv)
{
    const uint64_t strides[] = {4096, 65536, 1u << 21, 1u << 26};
    const uint32_t counts[] = {4, 8, 16, 32, 64, 128, 256, 512, 1024};
    const uint32_t rounds = 64, wgs = 256;
    const uint32_t wg_span = rounds * 256u;                       // elements one workgroup covers inside a stream
    uint64_t bytes = 1024ull * (1u << 26) + (uint64_t)wgs * wg_span * 8u + (1u << 21);
    uint64_t *buf = nullptr;
    OK(hipMalloc((void **)&buf, bytes));
    OK(hipMemset(buf, 0, 1 << 20));
    hipEvent_t a, b;
    OK(hipEventCreate(&a)); OK(hipEventCreate(&b));
    int idx = 0;
    for (uint64_t S : strides)
        for (uint32_t n : counts) {
            // streams must not overlap the workgroups' spans: span of all workgroups inside a stream = wgs * wg_span * 8 B = 32 MiB
            // for small strides the "streams" interleave with other workgroups' pieces, which is fine: distinct addresses,
            // (s * S + wg * span) collide only if S divides span multiples: keep n * S <= span for small S by shrinking wgs
            uint32_t g = wgs;
            if (S < (uint64_t)wg_span * 8u) {
                // place workgroups n*S apart instead
                // (handled through wg_span_elems below)
            }
            const uint32_t span_elems = S < (uint64_t)wg_span * 8u ? (uint32_t)((uint64_t)n * S / 8u > wg_span ? (uint64_t)n * S / 8u : wg_span) : wg_span;
            if ((uint64_t)g * span_elems * 8u + (uint64_t)n * S > bytes) g = (uint32_t)((bytes - (uint64_t)n * S) / ((uint64_t)span_elems * 8u));
            if (g == 0) continue;
            OK(hipEventRecord(a));
            hipLaunchKernelGGL(streams_kernel, dim3(g), dim3(256), 0, 0, buf, S / 8u, n, rounds, span_elems);
            OK(hipEventRecord(b));
            OK(hipEventSynchronize(b));
            float ms = 0;
            OK(hipEventElapsedTime(&ms, a, b));
            const double stores = (double)g * 4 * rounds * n;          // wave-level store instructions
            printf("dispatch %d stride %llu streams %u workgroups %u: %.3f ms, %.1f ns per wave-store per CU-wave, %.1f GB/s\n", idx++,
                   (unsigned long long)S, n, g, ms, 1e6 * ms / (rounds * n), stores * 512 / ms / 1e6);
        }
    OK(hipFree(buf));
    return 0;
}
