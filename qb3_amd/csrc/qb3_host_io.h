// qb3_amd/csrc/qb3_host_io.h -- host side of the host-pointer API (qb3_encode / qb3_read_data, reference QB3.h:110,141): moving
// the caller's pageable buffers over the host link.
//
// Measured on the MI355X box (tools/pcie_probe.cpp): pinned memory moves at 55 GB/s up, 57 GB/s down, 48 GB/s each way when
// both directions run at once; pageable memory through the runtime's own bounce buffers at 26 GB/s up; one host thread copies
// 30 GB/s, eight 130 GB/s; starting a thread costs 30 us.  So: a ring of pinned slices, filled / drained by a small pool of
// PERSISTENT copy threads (round 3 started threads per slice: a hundred slices of a 16384 x 16384 x 3 raster x 7 threads x 30 us
// was most of what the calls lost against the link), the DMA of one slice beside the host copy of the next, and -- where the
// coding can be cut into strips -- upload, kernels and download of different strips at once on three streams.
#pragma once
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <unistd.h>

namespace qb3host {

// A process-wide pool of copy threads.  Jobs are memcpy pieces; whoever waits for a batch works on the queue itself, so the
// pool is correct with zero workers (thread creation refused: a pids limit) and never sends an exception across the C ABI.
class CopyPool {
public:
    struct Batch { std::atomic<uint32_t> pending{0}; };
    static CopyPool &get() {
        static CopyPool *g = new CopyPool();                // (never destroyed: its threads may outlive static destructors)
        if (g->pid_ != getpid()) g->reset_after_fork();
        return *g;
    }
    // queues the copy in pieces; returns at once.  b.pending counts the pieces not yet done.
    void submit(void *dst, const void *src, size_t n, Batch &b) {
        if (!n) return;
        start_workers();
        const size_t piece = n > ((size_t)8 << 20) ? (size_t)1 << 20 : std::max<size_t>((size_t)256 << 10, (n / 8 + 4095) & ~(size_t)4095);
        uint32_t count = (uint32_t)((n + piece - 1) / piece);
        b.pending.fetch_add(count, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> l(mu_);
            for (size_t off = 0; off < n; off += piece) q_.push_back({(uint8_t *)dst + off, (const uint8_t *)src + off, std::min(piece, n - off), &b});
        }
        cv_.notify_all();
    }
    bool done(const Batch &b) const { return b.pending.load(std::memory_order_acquire) == 0; }
    // works on the queue until the batch is done
    void wait(Batch &b) {
        while (!done(b)) {
            Job j;
            if (pop(j, false)) run(j);
            else std::this_thread::yield();
        }
    }
    bool help_one() {                                       // one piece of whatever is queued, if anything is
        Job j;
        if (!pop(j, false)) return false;
        run(j);
        return true;
    }
    void copy(void *dst, const void *src, size_t n) {       // parallel memcpy, returns when done
        if (n < ((size_t)1 << 20)) { memcpy(dst, src, n); return; }
        Batch b;
        submit(dst, src, n, b);
        wait(b);
    }
private:
    struct Job { uint8_t *dst; const uint8_t *src; size_t n; Batch *b; };
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Job> q_;
    unsigned nworkers_ = 0;
    bool started_ = false;
    pid_t pid_ = getpid();
    static unsigned want_workers() {
        const unsigned h = std::thread::hardware_concurrency();
        return h >= 32 ? 8u : h >= 16 ? 6u : h >= 8 ? 3u : h >= 4 ? 1u : 0u;       // (+ the waiting thread itself)
    }
    void start_workers() {
        if (started_) return;
        std::lock_guard<std::mutex> l(mu_);
        if (started_) return;
        started_ = true;
        const unsigned n = want_workers();
        for (unsigned i = 0; i < n; i++) {
            try { std::thread([this] { worker(); }).detach(); nworkers_++; }
            catch (...) { break; }                          // fewer workers: the waiting thread does the rest
        }
    }
    void reset_after_fork() {                               // the child has none of the parent's threads
        new (&mu_) std::mutex(); new (&cv_) std::condition_variable();
        q_.clear(); nworkers_ = 0; started_ = false; pid_ = getpid();
    }
    bool pop(Job &j, bool block) {
        std::unique_lock<std::mutex> l(mu_);
        if (block) cv_.wait(l, [this] { return !q_.empty(); });
        if (q_.empty()) return false;
        j = q_.front(); q_.pop_front();
        return true;
    }
    static void run(const Job &j) {
        memcpy(j.dst, j.src, j.n);
        j.b->pending.fetch_sub(1, std::memory_order_release);
    }
    void worker() { for (;;) { Job j; if (pop(j, true)) run(j); } }
};

// A ring of pinned slices with an event each.  Pinning memory costs milliseconds, so rings of handles that went away wait
// (two at most) for the next handle that asks (ring_acquire / ring_release); qb3x_trim lets them go.
struct PinnedRing {
    static constexpr size_t SLICE = (size_t)16 << 20, NSLOT = 6, MIN_BYTES = (size_t)4 << 20;
    uint8_t *slot[NSLOT] = {};
    hipEvent_t ev[NSLOT] = {};
    bool ready = false, failed = false;
    bool init() {
        if (ready || failed) return ready;
        for (size_t i = 0; i < NSLOT; i++) {
            if (hipHostMalloc((void **)&slot[i], SLICE, hipHostMallocDefault) != hipSuccess ||
                hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) { failed = true; (void)hipGetLastError(); release(); return false; }
        }
        return ready = true;
    }
    void release() {
        for (size_t i = 0; i < NSLOT; i++) {
            if (slot[i]) (void)hipHostFree(slot[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
            slot[i] = nullptr; ev[i] = nullptr;
        }
        ready = false;
    }
};

struct RingCache { std::mutex mu; PinnedRing *idle[2] = {nullptr, nullptr}; };
inline RingCache &ring_cache() { static RingCache *g = new RingCache(); return *g; }
inline PinnedRing *ring_acquire() {
    RingCache &c = ring_cache();
    {
        std::lock_guard<std::mutex> l(c.mu);
        for (auto &r : c.idle) if (r) { PinnedRing *got = r; r = nullptr; return got; }
    }
    PinnedRing *r = new (std::nothrow) PinnedRing();
    if (r && !r->init()) { delete r; r = nullptr; }
    return r;
}
inline void ring_release(PinnedRing *r) {
    if (!r) return;
    RingCache &c = ring_cache();
    {
        std::lock_guard<std::mutex> l(c.mu);
        for (auto &slot : c.idle) if (!slot) { slot = r; return; }
    }
    r->release();
    delete r;
}
inline void ring_trim() {
    RingCache &c = ring_cache();
    PinnedRing *out[2];
    { std::lock_guard<std::mutex> l(c.mu); for (int i = 0; i < 2; i++) { out[i] = c.idle[i]; c.idle[i] = nullptr; } }
    for (auto r : out) if (r) { r->release(); delete r; }
}

}  // namespace qb3host
