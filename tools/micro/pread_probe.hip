// tools/micro/pread_probe.hip — what bounds the CLI's count phase: pread(2) from the page cache (a file in /dev/shm)
// into the staging buffers.  Does the KIND of destination memory matter?  Per kind and thread count: aggregate GB/s of
// 32 MiB preads, every thread into its own 64 MiB buffer, ~2 s each.
//
//   malloc            ordinary pageable memory (the ceiling of the copy itself)
//   malloc+huge       the same with MADV_HUGEPAGE
//   hipHostMalloc     what the contexts use today (default flags: coherent, fine-grained)
//   noncoherent       hipHostMallocNonCoherent (coarse-grained: visible to the device at synchronisation points — enough for staging)
//   writecombined     hipHostMallocWriteCombined
//   registered        malloc + hipHostRegister
//
// build: hipcc -O2 --offload-arch=gfx950 -o tools/micro/pread_probe tools/micro/pread_probe.hip -pthread
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static const size_t kBlock = 32u << 20, kBuf = 64u << 20;
static int g_fd;
static uint64_t g_file_bytes;
static std::atomic<uint64_t> g_next;
static std::atomic<int> g_stop;

struct Job { char *buf; uint64_t bytes; };

static void *reader(void *arg)
{
    Job *j = (Job *)arg;
    const uint64_t n_blocks = g_file_bytes / kBlock;
    while (!g_stop.load(std::memory_order_relaxed)) {
        const uint64_t b = g_next.fetch_add(1) % n_blocks;
        size_t got = 0;
        while (got < kBlock) {
            const ssize_t r = pread(g_fd, j->buf + (b & 1) * kBlock + got, kBlock - got, (off_t)(b * kBlock + got));
            if (r <= 0) return nullptr;
            got += (size_t)r;
        }
        j->bytes += kBlock;
    }
    return nullptr;
}

static double now()
{
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

int main(int argc, char **argv)
{
    const char *path = argc > 1 ? argv[1] : "/dev/shm/qm2_pread_probe.bin";
    const uint64_t gb = argc > 2 ? strtoull(argv[2], nullptr, 10) : 8;
    const double secs = argc > 3 ? atof(argv[3]) : 2.0;
    g_file_bytes = gb << 30;
    {
        const int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0600);
        if (fd < 0) { perror("open"); return 1; }
        std::vector<char> block(kBlock);
        for (size_t i = 0; i < kBlock; i++) block[i] = "ACGT\n"[(i * 2654435761u >> 13) % 5];
        for (uint64_t at = 0; at < g_file_bytes; at += kBlock)
            if (write(fd, block.data(), kBlock) != (ssize_t)kBlock) { perror("write"); return 1; }
        close(fd);
    }
    g_fd = open(path, O_RDONLY);
    int dev_count = 0;
    if (hipGetDeviceCount(&dev_count) != hipSuccess || dev_count == 0) { puts("no GPU"); return 1; }
    const char *kinds[] = {"malloc", "malloc+huge", "hipHostMalloc", "noncoherent", "writecombined", "registered"};
    const int threads[] = {1, 4, 8, 16, 24};
    for (int kind = 0; kind < 6; kind++)
        for (int nt : threads) {
            std::vector<Job> jobs(nt);
            std::vector<pthread_t> th(nt);
            bool ok = true;
            for (int t = 0; t < nt && ok; t++) {
                void *p = nullptr;
                switch (kind) {
                case 0: p = aligned_alloc(1 << 21, kBuf); break;
                case 1: p = aligned_alloc(1 << 21, kBuf); if (p) madvise(p, kBuf, MADV_HUGEPAGE); break;
                case 2: ok = hipHostMalloc(&p, kBuf, hipHostMallocDefault) == hipSuccess; break;
                case 3: ok = hipHostMalloc(&p, kBuf, hipHostMallocNonCoherent) == hipSuccess; break;
                case 4: ok = hipHostMalloc(&p, kBuf, hipHostMallocWriteCombined) == hipSuccess; break;
                case 5: p = aligned_alloc(1 << 21, kBuf); ok = p && hipHostRegister(p, kBuf, hipHostRegisterDefault) == hipSuccess; break;
                }
                if (!p) ok = false;
                if (ok) memset(p, 1, kBuf);
                jobs[t] = {(char *)p, 0};
            }
            if (!ok) { printf("%-14s %2d threads: allocation refused\n", kinds[kind], nt); continue; }
            g_stop = 0;
            g_next = 0;
            const double t0 = now();
            for (int t = 0; t < nt; t++) pthread_create(&th[t], nullptr, reader, &jobs[t]);
            while (now() - t0 < secs) usleep(10000);
            g_stop = 1;
            for (int t = 0; t < nt; t++) pthread_join(th[t], nullptr);
            const double dt = now() - t0;
            uint64_t total = 0;
            for (auto &j : jobs) total += j.bytes;
            printf("%-14s %2d threads: %6.1f GB/s\n", kinds[kind], nt, total / dt / 1e9);
            fflush(stdout);
            for (int t = 0; t < nt; t++) {
                if (kind == 2 || kind == 3 || kind == 4) (void)hipHostFree(jobs[t].buf);
                else { if (kind == 5) (void)hipHostUnregister(jobs[t].buf); free(jobs[t].buf); }
            }
        }
    close(g_fd);
    unlink(path);
    return 0;
}
