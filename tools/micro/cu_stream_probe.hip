// tools/micro/cu_stream_probe.hip — what can ONE CU pull from HBM, and does that change with how many CUs are pulling?
//
// The question behind EXPERIMENTS.md section 4.1.3 ("time x CUs is constant"): bucket_count moves 16 GB/s per CU and needs
// exactly twice as long on half the CUs.  Is 16 GB/s close to what a CU's memory pipe can do at all — in which case the
// chip's 8 TB/s is only there for kernels whose 256 CUs each run at their own limit — or is it the kernel's latency hiding?
//
// One resident workgroup of 1024 threads per "CU" (grid = N workgroups, N = 32 .. 256: the other CUs idle), each streaming
// its own contiguous piece of a large buffer with 16-byte loads (1 KiB per wave instruction), D loads in flight per thread
// before the first is used (D = 1, 2, 4, 8: the software pipelining depth); or copying it (loads + 16-byte stores).
// Output: GB/s in total and per workgroup (= per CU).
//
// build: hipcc -O3 --offload-arch=gfx950 -o tools/micro/cu_stream_probe tools/micro/cu_stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int D, bool COPY>
__global__ __launch_bounds__(1024) void stream_kernel(const u32x4 *src, u32x4 *dst, uint64_t n16_per_wg, unsigned long long *sink)
{
    __shared__ uint32_t whole_cu[24 * 1024];                              // 96 KiB: no second workgroup fits the CU
    whole_cu[threadIdx.x] = threadIdx.x;
    const u32x4 *p = src + (uint64_t)blockIdx.x * n16_per_wg;
    u32x4 *q = dst + (uint64_t)blockIdx.x * n16_per_wg;
    uint32_t acc = 0;
    // n16_per_wg is a multiple of 1024 * D: no tail
    for (uint64_t i = threadIdx.x; i < n16_per_wg; i += (uint64_t)1024 * D) {
        u32x4 v[D];
#pragma unroll
        for (int d = 0; d < D; d++) v[d] = p[i + (uint64_t)d * 1024];
#pragma unroll
        for (int d = 0; d < D; d++) {
            if (COPY) q[i + (uint64_t)d * 1024] = v[d];
            else acc += v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
        }
    }
    if ((!COPY && acc == 0x12345678u) || whole_cu[(threadIdx.x * 7u) & 1023u] == 0xFFFFFFFFu) atomicAdd(sink, 1ull);   // keeps the loads (and the LDS) alive
}

template <int D, bool COPY>
static int run(const u32x4 *src, u32x4 *dst, uint64_t bytes_total, int n_wg, unsigned long long *sink)
{
    // every workgroup gets the same piece whatever the grid: 256 pieces in the buffer, the first n_wg are used
    const uint64_t per = bytes_total / 256 / 16 / (1024 * D) * (1024 * D);
    hipEvent_t a, b;
    OK(hipEventCreate(&a)); OK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        OK(hipEventRecord(a, 0));
        stream_kernel<D, COPY><<<dim3(n_wg), dim3(1024), 0, 0>>>(src, dst, per, sink);
        OK(hipEventRecord(b, 0));
        OK(hipEventSynchronize(b));
        float ms = 0;
        OK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    const double moved = (double)per * 16 * n_wg * (COPY ? 2 : 1);
    printf("%s  %3d workgroups  depth %d : %7.3f ms  %7.1f GB/s  = %5.1f GB/s per CU\n", COPY ? "copy" : "read", n_wg, D, best, moved / best / 1e6,
           moved / best / 1e6 / n_wg);
    OK(hipEventDestroy(a)); OK(hipEventDestroy(b));
    return 0;
}

int main()
{
    const uint64_t bytes = 16ull << 30;
    u32x4 *src = nullptr, *dst = nullptr;
    unsigned long long *sink = nullptr;
    OK(hipMalloc((void **)&src, bytes));
    OK(hipMalloc((void **)&dst, bytes));
    OK(hipMalloc((void **)&sink, 8));
    OK(hipMemset(src, 1, bytes));
    OK(hipMemset(dst, 2, bytes));
    OK(hipMemset(sink, 0, 8));
    OK(hipDeviceSynchronize());
    const int grids[] = {256, 192, 128, 64, 32};
    for (int g : grids) {
        if (run<1, false>(src, dst, bytes, g, sink)) return 1;
        if (run<2, false>(src, dst, bytes, g, sink)) return 1;
        if (run<4, false>(src, dst, bytes, g, sink)) return 1;
        if (run<8, false>(src, dst, bytes, g, sink)) return 1;
    }
    for (int g : grids) {
        if (run<2, true>(src, dst, bytes, g, sink)) return 1;
        if (run<4, true>(src, dst, bytes, g, sink)) return 1;
    }
    OK(hipFree(src)); OK(hipFree(dst)); OK(hipFree(sink));
    return 0;
}
