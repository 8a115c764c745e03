// How fast can ONE file (the .h5 of a dbgh5 run) take records from many threads? (round 5, the drop-in's sink)  g++ -O2 -pthread filewrite_probe.cpp -o filewrite_probe
//   ./filewrite_probe [dir=/dev/shm] [GB=8] [piece MB=2] [source buffer MB=0: one cache-resident piece]
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/dev/shm"; const size_t GB = argc > 2 ? atoi(argv[2]) : 8, piece = (size_t)(argc > 3 ? atoi(argv[3]) : 2) << 20;
    const size_t total = GB << 30, np = total / piece;
    const size_t src_total = (size_t)(argc > 4 ? atoi(argv[4]) : 0) << 20;        // > 0: the pieces come from a buffer of this many MB walked round robin (not cache resident), like slots the DMA just filled
    std::vector<char> src(src_total ? src_total : piece, 7);
    auto src_of = [&](size_t i) { return src.data() + (src_total ? (i * piece) % (src_total - piece + 1) / piece * piece : 0); };
    const std::string path = dir + "/filewrite_probe.bin";
    auto run = [&](const char* name, int nthreads, int mode) {
        unlink(path.c_str());
        int fd = open(path.c_str(), O_RDWR | O_CREAT, 0644);
        if (mode >= 2) { if (ftruncate(fd, total) != 0) { printf("ftruncate failed\n"); return; } }
        char* whole = nullptr;
        if (mode == 3 || mode == 4) { whole = (char*)mmap(0, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); if (whole == MAP_FAILED) { printf("mmap failed\n"); return; } }
        if (mode == 4) madvise(whole, total, MADV_HUGEPAGE);
        std::atomic<size_t> next(0);
        double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++) th.emplace_back([&] {
            int myfd = open(path.c_str(), O_RDWR);
            for (;;) { size_t i = next.fetch_add(1); if (i >= np) break;
                if (mode <= 1) { size_t left = piece; off_t at = i * piece; const char* p = src_of(i); while (left) { ssize_t w = pwrite(myfd, p, left, at); if (w <= 0) break; p += w; at += w; left -= w; } }
                else if (mode == 2) { char* m = (char*)mmap(0, piece, PROT_READ | PROT_WRITE, MAP_SHARED, myfd, i * piece); memcpy(m, src_of(i), piece); munmap(m, piece); }
                else memcpy(whole + i * piece, src_of(i), piece);
            }
            close(myfd); });
        for (auto& x : th) x.join();
        double dt = now() - t0;
        if (whole) munmap(whole, total);
        close(fd);
        printf("%-64s %3d threads: %6.2f s  %6.2f GB/s\n", name, nthreads, dt, total / dt / 1e9); fflush(stdout);
    };
    printf("# %zu GB into %s in pieces of %zu MB\n", GB, path.c_str(), piece >> 20);
    run("pwrite", 1, 0);
    for (int n : {2, 3, 4, 8, 32, 128}) run("pwrite, disjoint ranges", n, 1);
    if (getenv("PROBE_PWRITE_ONLY")) { unlink(path.c_str()); return 0; }
    for (int n : {1, 8, 32, 128, 256}) run("mmap + memcpy + munmap per piece", n, 2);
    for (int n : {8, 32, 128, 256}) run("one mapping of the file, memcpy per piece", n, 3);
    for (int n : {32, 128}) run("one mapping + MADV_HUGEPAGE, memcpy per piece", n, 4);
    unlink(path.c_str());
    return 0;
}
