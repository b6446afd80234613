// io_probe.cpp -- what a file <-> pinned memory <-> HBM path can move on the GPU box (round 4).
// Build: hipcc -O2 -std=c++17 io_probe.cpp -o io_probe -pthread ; run: ./io_probe /dev/shm/oip_probe.bin 4
//   1. serial write() of G GiB into a tmpfs file (the reference's WriteBufferToFile loop, imageop.h:84-97)
//   2. pread into a pinned buffer with T threads (page cache -> pinned)
//   3. the same pipelined with H2D through four 32 MiB slots
//   4. writes: serial write(), T-thread pwrite on one inode, T-thread memcpy into a fresh MAP_SHARED mapping
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <typename F> static void par(int T, F fn)
{
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back([&, t] { fn(t); });
    fn(0);
    for (auto &x : th) x.join();
}

int main(int argc, char **argv)
{
    const char *path = argc > 1 ? argv[1] : "/dev/shm/oip_probe.bin";
    const size_t G = (size_t)(argc > 2 ? atoi(argv[2]) : 4) << 30;
    const size_t SLOT = (size_t)32 << 20;
    void *pin[4];
    for (auto &p : pin) if (hipHostMalloc(&p, SLOT, hipHostMallocDefault) != hipSuccess) { puts("hipHostMalloc failed"); return 1; }
    void *dev = nullptr;
    if (hipMalloc(&dev, G) != hipSuccess) { puts("hipMalloc failed"); return 1; }
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t ev[4];
    for (auto &e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (auto p : pin) memset(p, 0x5a, SLOT);
    printf("host threads: %u\n", std::thread::hardware_concurrency());

    // 1. serial write (skipped with a third argument "reuse": the file was written by ANOTHER process, as the CLI's inputs are)
    if (argc > 3) {
        struct stat st;
        if (stat(path, &st) != 0 || (size_t)st.st_size < G) { puts("reuse: file missing or short"); return 1; }
        puts("reading a file another process wrote");
    } else {
        int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        double t0 = now();
        for (size_t off = 0; off < G; off += SLOT) if (write(fd, pin[0], SLOT) != (ssize_t)SLOT) { puts("write failed"); return 1; }
        close(fd);
        printf("serial write() of %zu GiB into a new file: %.2f GB/s\n", G >> 30, G / (now() - t0) / 1e9);
    }
    // 2. pread into pinned, T threads, slot by slot
    int fd = open(path, O_RDONLY);
    for (int T : {1, 4, 8, 16, 32, 64}) {
        double t0 = now();
        for (size_t off = 0; off < G; off += SLOT) {
            char *dst = (char *)pin[(off / SLOT) & 3];
            const size_t part = SLOT / T;
            par(T, [&](int t) { size_t o = t * part, n = t == T - 1 ? SLOT - o : part; if (pread(fd, dst + o, n, off + o) != (ssize_t)n) puts("short pread"); });
        }
        printf("pread -> pinned slot, %2d threads (spawned per slot): %.2f GB/s\n", T, G / (now() - t0) / 1e9);
    }
    // 2b. pread with persistent threads each owning an interleaved share of every slot (no per-slot spawn): upper bound of the pool
    for (int T : {8, 16, 32, 64}) {
        double t0 = now();
        par(T, [&](int t) {
            const size_t part = SLOT / T;
            for (size_t off = 0; off < G; off += SLOT) {
                char *dst = (char *)pin[(off / SLOT) & 3];
                size_t o = t * part, n = t == T - 1 ? SLOT - o : part;
                if (pread(fd, dst + o, n, off + o) != (ssize_t)n) puts("short pread");
            }
        });
        printf("pread -> pinned, %2d persistent threads, no barrier per slot: %.2f GB/s\n", T, G / (now() - t0) / 1e9);
    }
    // 3. pread + H2D through the four slots
    for (int T : {8, 16, 32}) {
        bool used[4] = {false, false, false, false};
        double t0 = now();
        for (size_t off = 0; off < G; off += SLOT) {
            const int i = (off / SLOT) & 3;
            if (used[i]) hipEventSynchronize(ev[i]);
            char *dst = (char *)pin[i];
            const size_t part = SLOT / T;
            par(T, [&](int t) { size_t o = t * part, n = t == T - 1 ? SLOT - o : part; if (pread(fd, dst + o, n, off + o) != (ssize_t)n) puts("short pread"); });
            hipMemcpyAsync((char *)dev + off, dst, SLOT, hipMemcpyHostToDevice, st);
            hipEventRecord(ev[i], st);
            used[i] = true;
        }
        hipStreamSynchronize(st);
        printf("file -> pinned -> HBM, %2d threads: %.2f GB/s\n", T, G / (now() - t0) / 1e9);
    }
    close(fd);
    if (argc > 3) return 0;
    // 4. writes of a fresh file each
    for (int T : {1, 8, 16, 32}) {
        unlink(path);
        int wfd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        double t0 = now();
        for (size_t off = 0; off < G; off += SLOT) {
            const char *src = (const char *)pin[(off / SLOT) & 3];
            const size_t part = SLOT / T;
            par(T, [&](int t) { size_t o = t * part, n = t == T - 1 ? SLOT - o : part; if (pwrite(wfd, src + o, n, off + o) != (ssize_t)n) puts("short pwrite"); });
        }
        close(wfd);
        printf("pwrite from pinned, %2d threads, new file: %.2f GB/s\n", T, G / (now() - t0) / 1e9);
    }
    for (int T : {1, 8, 16, 32, 64}) {
        unlink(path);
        int wfd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        double t0 = now();
        if (ftruncate(wfd, (off_t)G)) { puts("ftruncate failed"); return 1; }
        char *map = (char *)mmap(nullptr, G, PROT_READ | PROT_WRITE, MAP_SHARED, wfd, 0);
        if (map == MAP_FAILED) { puts("mmap failed"); return 1; }
        for (size_t off = 0; off < G; off += SLOT) {
            const char *src = (const char *)pin[(off / SLOT) & 3];
            const size_t part = SLOT / T;
            par(T, [&](int t) { size_t o = t * part, n = t == T - 1 ? SLOT - o : part; memcpy(map + off + o, src + o, n); });
        }
        munmap(map, G);
        close(wfd);
        printf("memcpy from pinned into a MAP_SHARED mapping of a new file, %2d threads: %.2f GB/s\n", T, G / (now() - t0) / 1e9);
    }
    // 4b. HBM -> pinned -> mapping, pipelined over two slots
    for (int T : {16, 32}) {
        unlink(path);
        int wfd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        double t0 = now();
        if (ftruncate(wfd, (off_t)G)) { puts("ftruncate failed"); return 1; }
        char *map = (char *)mmap(nullptr, G, PROT_READ | PROT_WRITE, MAP_SHARED, wfd, 0);
        long prev = -1;
        auto drain = [&](long k) {
            hipEventSynchronize(ev[k & 3]);
            const char *src = (const char *)pin[k & 3];
            const size_t off = (size_t)k * SLOT, part = SLOT / T;
            par(T, [&](int t) { size_t o = t * part, n = t == T - 1 ? SLOT - o : part; memcpy(map + off + o, src + o, n); });
        };
        for (size_t off = 0; off < G; off += SLOT) {
            const long k = (long)(off / SLOT);
            hipMemcpyAsync(pin[k & 3], (char *)dev + off, SLOT, hipMemcpyDeviceToHost, st);
            hipEventRecord(ev[k & 3], st);
            if (prev >= 0) drain(prev);
            prev = k;
        }
        drain(prev);
        munmap(map, G);
        close(wfd);
        printf("HBM -> pinned -> mapping of a new file, %2d threads: %.2f GB/s\n", T, G / (now() - t0) / 1e9);
    }
    unlink(path);
    return 0;
}
