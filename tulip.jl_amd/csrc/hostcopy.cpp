// hostcopy.cpp -- see hostcopy.hpp: a small persistent pool of host threads for the staging copies of the host-pointer
// entry points, and the non-temporal copy into the pinned staging area.
#include "hostcopy.hpp"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include <unistd.h>

namespace tlpk {
namespace {

inline void cpu_relax() { __builtin_ia32_pause(); }

struct Job {
    const std::function<void(int)> *fn;
    int n;
    std::atomic<int> next{0}, done{0};
    int inside = 0;                    // workers that hold a pointer to this job (guarded by Pool::mu)
};

// One pool per process.  A job lives on its poster's stack: a worker registers (inside++) under the mutex while the job is still
// posted, the poster withdraws the job under the same mutex and then waits for inside == 0 before it returns -- no worker can hold
// a dangling pointer, and an index is never drawn from one job and run against another.
struct Pool {
    std::vector<std::thread> th;
    std::mutex mu, post_mu;            // post_mu: one job at a time (handles of different host threads share the pool)
    std::condition_variable cv;
    Job *cur = nullptr;
    std::atomic<uint64_t> gen{0};
    bool stop = false;

    static void work(Job *j) {
        for (;;) {
            const int i = j->next.fetch_add(1, std::memory_order_relaxed);
            if (i >= j->n) break;
            (*j->fn)(i);
            j->done.fetch_add(1, std::memory_order_release);
        }
    }
    void worker() {
        uint64_t seen = 0;
        for (;;) {
            // spin briefly (the calls of a Newton step follow each other within a few hundred microseconds), then sleep
            const auto t0 = std::chrono::steady_clock::now();
            bool fresh = false;
            for (int it = 0;; ++it) {
                if (gen.load(std::memory_order_acquire) != seen) { fresh = true; break; }
                cpu_relax();
                if ((it & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(100)) break;
            }
            Job *j = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                if (!fresh) cv.wait(lk, [&] { return stop || gen.load(std::memory_order_relaxed) != seen; });
                if (stop) return;
                seen = gen.load(std::memory_order_relaxed);
                j = cur;
                if (j) ++j->inside;
            }
            if (!j) continue;                                   // the job was finished before this worker woke up
            work(j);
            { std::lock_guard<std::mutex> lk(mu); --j->inside; }
        }
    }
    explicit Pool(int nworkers) {
        try { for (int t = 0; t < nworkers; ++t) th.emplace_back(&Pool::worker, this); } catch (...) { /* fewer workers */ }
    }
    ~Pool() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto &t : th) t.join();
    }
    void run(int n, const std::function<void(int)> &fn) {
        if (n <= 0) return;
        if (th.empty() || n == 1) { for (int i = 0; i < n; ++i) fn(i); return; }
        std::lock_guard<std::mutex> one(post_mu);
        Job job; job.fn = &fn; job.n = n;
        { std::lock_guard<std::mutex> lk(mu); cur = &job; gen.fetch_add(1, std::memory_order_release); }
        cv.notify_all();
        work(&job);
        while (job.done.load(std::memory_order_acquire) < n) cpu_relax();
        { std::lock_guard<std::mutex> lk(mu); cur = nullptr; }
        for (;;) {                                               // late workers that registered but found no index left
            { std::lock_guard<std::mutex> lk(mu); if (job.inside == 0) break; }
            cpu_relax();
        }
    }
};

int configured_workers() {
    int w = 4;
    if (const char *e = std::getenv("TLPK_COPY_THREADS")) w = std::atoi(e);
    const unsigned hc = std::thread::hardware_concurrency();
    if (hc > 0 && (unsigned)w + 1 > hc) w = (int)hc - 1;
    return w < 0 ? 0 : (w > 31 ? 31 : w);
}
// One pool per PROCESS (round-5 advisor finding): after fork() -- Python's multiprocessing with the fork start method -- the child inherits the parent's pool
// object but none of its threads; joining those handles at exit is undefined behaviour.  The holder remembers the pid that created the pool: a child builds its
// own on first use and abandons the inherited object (its threads do not exist there), and only the owning process joins its workers at exit.
struct PoolHolder {
    Pool *p = nullptr; pid_t owner = 0; std::mutex m;
    ~PoolHolder() { if (p && owner == getpid()) delete p; }
};
Pool &pool() {
    static PoolHolder h;
    std::lock_guard<std::mutex> lk(h.m);
    const pid_t me = getpid();
    if (!h.p || h.owner != me) { h.p = new Pool(configured_workers()); h.owner = me; }
    return *h.p;
}

}  // namespace

void host_parallel_for(int n, const std::function<void(int)> &fn) { pool().run(n, fn); }
int host_copy_threads() { return configured_workers() + 1; }      // (the configured size: asking must not start the workers)

void copy_to_staging(void *dst, const void *src, size_t bytes) {
    static const bool nt = [] { const char *e = std::getenv("TLPK_COPY_NT"); return !e || std::atoi(e) != 0; }();
    if (!nt || (reinterpret_cast<uintptr_t>(dst) & 15) != 0 || bytes < 4096) { std::memcpy(dst, src, bytes); return; }
    typedef double v2d __attribute__((vector_size(16)));
    const char *s = static_cast<const char *>(src);
    char *d = static_cast<char *>(dst);
    size_t i = 0;
    for (; i + 64 <= bytes; i += 64) {
        v2d a, b, c, e;
        std::memcpy(&a, s + i, 16); std::memcpy(&b, s + i + 16, 16); std::memcpy(&c, s + i + 32, 16); std::memcpy(&e, s + i + 48, 16);
        __builtin_nontemporal_store(a, reinterpret_cast<v2d *>(d + i));
        __builtin_nontemporal_store(b, reinterpret_cast<v2d *>(d + i + 16));
        __builtin_nontemporal_store(c, reinterpret_cast<v2d *>(d + i + 32));
        __builtin_nontemporal_store(e, reinterpret_cast<v2d *>(d + i + 48));
    }
    if (i < bytes) std::memcpy(d + i, s + i, bytes - i);
    __builtin_ia32_sfence();                                     // the stores are globally visible before the DMA engine is told to read them
}

}  // namespace tlpk
