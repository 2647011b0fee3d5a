// Test infrastructure: exercises tulip.jl_amd/csrc/hostcopy.cpp (the thread pool and the staging copy of the host-pointer entry
// points) without a GPU.  Built and run by tests/test_hostcopy.py.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../tulip.jl_amd/csrc/hostcopy.hpp"

using namespace tlpk;

static int check_indices(int n) {
    std::vector<std::atomic<int>> hit((size_t)n);
    for (auto &h : hit) h.store(0);
    host_parallel_for(n, [&](int i) { hit[(size_t)i].fetch_add(1); });
    for (int i = 0; i < n; ++i) if (hit[(size_t)i].load() != 1) { std::printf("index %d of %d ran %d times\n", i, n, hit[(size_t)i].load()); return 1; }
    return 0;
}

int main() {
    int bad = 0;
    // every index exactly once, for job sizes around the worker count, many jobs back to back (late workers of one job must never
    // draw an index of the next)
    for (int rep = 0; rep < 2000; ++rep) bad += check_indices(1 + rep % 37);
    bad += check_indices(100000);
    // two posting threads share the pool (two handles driven from two host threads)
    {
        std::atomic<int> fails{0};
        auto poster = [&] { for (int rep = 0; rep < 500; ++rep) if (check_indices(1 + rep % 19)) fails.fetch_add(1); };
        std::thread a(poster), b(poster);
        a.join(); b.join();
        bad += fails.load();
    }
    // staging copy: aligned destination, arbitrary source offset and length
    {
        const size_t N = 3u << 20;
        std::vector<double> src(N + 8), dst(N + 8);
        for (size_t i = 0; i < src.size(); ++i) src[i] = (double)i * 0.5 + 1.0;
        char *d0 = reinterpret_cast<char *>(dst.data());
        d0 += (16 - (reinterpret_cast<uintptr_t>(d0) & 15)) & 15;
        for (size_t bytes : {(size_t)0, (size_t)8, (size_t)4095, (size_t)4096, (size_t)4104, (size_t)65536 * 8 + 24, N * 8 - 64}) {
            for (size_t so : {(size_t)0, (size_t)8}) {
                std::memset(d0, 0xAB, bytes + 16);
                copy_to_staging(d0, reinterpret_cast<const char *>(src.data()) + so, bytes);
                if (std::memcmp(d0, reinterpret_cast<const char *>(src.data()) + so, bytes) != 0) { std::printf("copy mismatch bytes=%zu so=%zu\n", bytes, so); ++bad; }
                if ((unsigned char)d0[bytes] != 0xAB) { std::printf("copy overran bytes=%zu\n", bytes); ++bad; }
            }
        }
        // the piece-wise parallel copy the entry points perform
        std::vector<double> out(N);
        const size_t piece = 65536;
        host_parallel_for((int)((N + piece - 1) / piece), [&](int i) {
            const size_t o = (size_t)i * piece, c = std::min(piece, N - o);
            std::memcpy(out.data() + o, src.data() + o, c * 8);
        });
        if (std::memcmp(out.data(), src.data(), N * 8) != 0) { std::printf("parallel copy mismatch\n"); ++bad; }
    }
    std::printf("hostcopy_check: %d failures, %d copy threads\n", bad, host_copy_threads());
    return bad ? 1 : 0;
}
