// qb3_amd/csrc/qb3_api.cpp -- the C ABI (include/QB3.h, include/qb3x.h) of the MI355X-native QB3 codec.
//
// Host side only: handle bookkeeping, container headers, STORED fallback (the RLE0 byte pass is a device pass: k_rle0.hip)
// and the HIP plumbing (buffers, copies, one synchronisation per call).  The block coding itself -- the hot
// path -- is in the k_*.hip files and always runs on the GPU; there is no CPU fallback for it.
//
// Behaviour mirrors the reference C API (reference QB3lib/QB3encode.cpp, QB3decode.cpp), including the
// quirks a drop-in has to keep: band state carried across qb3_encode calls until qb3_reset_encoder
// (QB3encode.h:446-449), sticky Z order (QB3encode.cpp:124-132), mode left at STORED after a fallback
// (QB3encode.cpp:464), stale error blocking the handle (QB3encode.cpp:514).
#include <hip/hip_runtime_api.h>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <chrono>
#include <mutex>
#include <new>
#include <vector>
#include <limits>
#include <thread>
#include <algorithm>
#include "../../include/QB3.h"
#include "../../include/qb3x.h"
#include "qb3_dev.h"
#include "qb3_host_io.h"

using namespace qb3dev;

#define QB3_API extern "C" __attribute__((visibility("default")))

static const int typesizes[8] = { 1, 1, 2, 2, 4, 4, 8, 8 };
static inline size_t szof(int dt) { return (dt < 0 || dt > QB3_I64) ? 0 : typesizes[dt]; }
static inline unsigned topbit(uint64_t v) { return 63u - (unsigned)__builtin_clzll(v); }

// ---------------------------------------------------------------- device buffers owned by a handle
// Device buffers of destroyed handles wait in a small per-process pool for the next handle: a caller that opens, decodes and
// closes a container per tile (the reference's calling pattern) would otherwise pay tens of milliseconds of hipMalloc / hipFree
// around a fraction of a millisecond of kernels.  Bounded (POOL_ITEMS buffers, POOL_BYTES bytes); qb3x_trim() empties it.
struct DevPool {
    struct Item { void *p; size_t cap; int dev; };
    static constexpr size_t POOL_ITEMS = 24;
    const size_t POOL_BYTES = [] { const char *e = getenv("QB3_POOL_MB"); return (e && e[0] ? (size_t)strtoull(e, nullptr, 10) : (size_t)3072) << 20; }();   // (0: nothing is kept)
    std::mutex mu;
    std::vector<Item> items;
    size_t bytes = 0;
    void *take(size_t n, int dev, size_t *cap) {           // the smallest pooled buffer of this device that holds n and is not more than twice that
        std::lock_guard<std::mutex> l(mu);
        size_t best = items.size();
        for (size_t i = 0; i < items.size(); i++)
            if (items[i].dev == dev && items[i].cap >= n && items[i].cap / 2 <= n && (best == items.size() || items[i].cap < items[best].cap)) best = i;
        if (best == items.size()) return nullptr;
        void *p = items[best].p;
        *cap = items[best].cap;
        bytes -= items[best].cap;
        items.erase(items.begin() + (long)best);
        return p;
    }
    bool give(void *p, size_t cap, int dev) {
        std::lock_guard<std::mutex> l(mu);
        if (items.size() >= POOL_ITEMS || bytes + cap > POOL_BYTES) return false;
        items.push_back({p, cap, dev});
        bytes += cap;
        return true;
    }
    void trim() {
        std::vector<Item> out;
        { std::lock_guard<std::mutex> l(mu); out.swap(items); bytes = 0; }
        int cur = 0;
        (void)hipGetDevice(&cur);
        for (auto &it : out) { (void)hipSetDevice(it.dev); (void)hipFree(it.p); }
        (void)hipSetDevice(cur);
    }
};
static DevPool &dev_pool() { static DevPool *g = new DevPool(); return *g; }      // (never destroyed: the HIP runtime may be gone before static destructors run)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int dev = 0;
    bool ensure(size_t n) {
        if (n <= cap) return true;
        release();
        (void)hipGetDevice(&dev);
        try {
            if ((p = dev_pool().take(n, dev, &cap)) != nullptr) return true;
        } catch (...) { p = nullptr; }
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {                              // out of memory with buffers idle in the pool: give them back and try once more
            (void)hipGetLastError();
            dev_pool().trim();
            e = hipMalloc(&p, n);
        }
        if (e != hipSuccess) { set_error("hipMalloc", (int)e); p = nullptr; cap = 0; return false; }
        cap = n;
        return true;
    }
    // hipFree waits for the device; a buffer that goes to the pool instead may be handed to another handle on another stream
    // at once, so the same wait comes first -- unless the caller has just made it (`idle`: a handle's buffers go one after
    // the other).  An error return in the middle of a call leaves kernels in flight; they end here, not in the next owner.
    void release(bool idle = false) {
        if (p) {
            if (!idle) (void)hipDeviceSynchronize();
            bool kept = false;
            try { kept = dev_pool().give(p, cap, dev); } catch (...) { kept = false; }
            if (!kept) (void)hipFree(p);
        }
        p = nullptr; cap = 0;
    }
};
template <class... B> static void release_all(B &...b) {
    bool any = false;
    for (bool h : {(b.p != nullptr)...}) any = any || h;
    if (any) (void)hipDeviceSynchronize();
    (void)std::initializer_list<int>{(b.release(true), 0)...};
}

static bool device_ok() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { set_error("no usable HIP device (the block codec has no CPU fallback)", (int)e); return false; }
    return true;
}

// Waiting for a stream whose work is short: the runtime's blocking wait costs tens of microseconds to wake up, a kernel
// sequence of this library takes a few hundred.  Poll for a bounded time first.
static hipError_t wait_stream(hipStream_t st) {
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t i = 0;; i++) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
        if ((i & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(4)) break;
    }
    return hipStreamSynchronize(st);
}
// A few result bytes from the device, then the wait: through pinned memory of the calling thread (a copy into pageable
// memory goes through the runtime's staging path)
static hipError_t fetch_small(void *dst, const void *d_src, size_t n, hipStream_t st) {
    static thread_local void *pinned = nullptr;
    if (!pinned && hipHostMalloc(&pinned, 1024, hipHostMallocDefault) != hipSuccess) pinned = nullptr;
    if (!pinned || n > 1024) {
        hipError_t e = hipMemcpyAsync(dst, d_src, n, hipMemcpyDeviceToHost, st);
        return e == hipSuccess ? hipStreamSynchronize(st) : e;
    }
    hipError_t e = hipMemcpyAsync(pinned, d_src, n, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = wait_stream(st);
    if (e == hipSuccess) memcpy(dst, pinned, n);
    return e;
}

// ---------------------------------------------------------------- host <-> device copies of the host-pointer API
// The reference API hands over pageable host memory (qb3_host_io.h: the ring of pinned slices, the pool of copy threads).
struct Stager {
    qb3host::PinnedRing *ring = nullptr;
    static constexpr size_t SLICE = qb3host::PinnedRing::SLICE, NSLOT = qb3host::PinnedRing::NSLOT, MIN_BYTES = qb3host::PinnedRing::MIN_BYTES;
    bool failed = false;
    bool init() {
        if (ring) return true;
        if (failed) return false;
        try { ring = qb3host::ring_acquire(); } catch (...) { ring = nullptr; }
        failed = !ring;
        return ring != nullptr;
    }
    void release() { try { qb3host::ring_release(ring); } catch (...) {} ring = nullptr; }
    uint8_t *slot(size_t i) const { return ring->slot[i % NSLOT]; }
    hipEvent_t ev(size_t i) const { return ring->ev[i % NSLOT]; }
};
static void parallel_memcpy(uint8_t *dst, const uint8_t *src, size_t n) {
    try { qb3host::CopyPool::get().copy(dst, src, n); }
    catch (...) { memcpy(dst, src, n); }                    // (a pool that cannot be had: the caller's thread copies)
}
// host -> device; returns once the host bytes have been consumed (the last slices may still be on the link)
static bool upload(Stager &sg, void *d_dst, const void *h_src, size_t bytes, hipStream_t st) {
    if (bytes < Stager::MIN_BYTES || !sg.init()) {
        hipError_t e = hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { set_error("upload", (int)e); return false; }
        return true;
    }
    size_t i = 0;
    for (size_t off = 0; off < bytes; off += Stager::SLICE, i++) {
        const size_t n = std::min(Stager::SLICE, bytes - off);
        if (i >= Stager::NSLOT && hipEventSynchronize(sg.ev(i)) != hipSuccess) { set_error("upload: slot wait", 0); return false; }
        parallel_memcpy(sg.slot(i), (const uint8_t *)h_src + off, n);
        hipError_t e = hipMemcpyAsync((uint8_t *)d_dst + off, sg.slot(i), n, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipEventRecord(sg.ev(i), st);
        if (e != hipSuccess) { set_error("upload", (int)e); return false; }
    }
    return true;
}
// device -> host; returns when the bytes are in h_dst (synchronises the stream)
static bool download(Stager &sg, void *h_dst, const void *d_src, size_t bytes, hipStream_t st) {
    if (bytes < Stager::MIN_BYTES || !sg.init()) {
        hipError_t e = hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_error("download", (int)e); return false; }
        return true;
    }
    const size_t nsl = (bytes + Stager::SLICE - 1) / Stager::SLICE;
    auto issue = [&](size_t i) -> hipError_t {
        const size_t off = i * Stager::SLICE, n = std::min(Stager::SLICE, bytes - off);
        hipError_t e = hipMemcpyAsync(sg.slot(i), (const uint8_t *)d_src + off, n, hipMemcpyDeviceToHost, st);
        return e == hipSuccess ? hipEventRecord(sg.ev(i), st) : e;
    };
    hipError_t e = hipSuccess;
    for (size_t i = 0; i < std::min(nsl, (size_t)Stager::NSLOT) && e == hipSuccess; i++) e = issue(i);
    for (size_t i = 0; i < nsl && e == hipSuccess; i++) {
        e = hipEventSynchronize(sg.ev(i));
        if (e != hipSuccess) break;
        const size_t off = i * Stager::SLICE, n = std::min(Stager::SLICE, bytes - off);
        parallel_memcpy((uint8_t *)h_dst + off, sg.slot(i), n);
        if (i + Stager::NSLOT < nsl) e = issue(i + Stager::NSLOT);
    }
    if (e != hipSuccess) { set_error("download", (int)e); return false; }
    return true;
}

// Streams and events of a pipelined host call (upload, kernels and download of different strips at once), kept by the handle
struct Pipe {
    hipStream_t up = nullptr, k = nullptr, dn = nullptr;
    std::vector<hipEvent_t> ev;
    uint64_t *words = nullptr;                      // pinned: a few result words the kernel stream copies down (WORDS of them)
    static constexpr size_t WORDS = 2048;
    bool failed = false;
    bool init() {
        if (up) return true;
        if (failed) return false;
        if (hipStreamCreateWithFlags(&up, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&k, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&dn, hipStreamNonBlocking) != hipSuccess || hipHostMalloc((void **)&words, 8 * WORDS, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError(); release(); failed = true; return false;
        }
        return true;
    }
    bool events(size_t n) {
        while (ev.size() < n) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
            ev.push_back(e);
        }
        return true;
    }
    void sync() { if (up) { (void)hipStreamSynchronize(up); (void)hipStreamSynchronize(k); (void)hipStreamSynchronize(dn); } }
    void release() {
        for (auto e : ev) (void)hipEventDestroy(e);
        ev.clear();
        if (up) (void)hipStreamDestroy(up);
        if (k) (void)hipStreamDestroy(k);
        if (dn) (void)hipStreamDestroy(dn);
        if (words) (void)hipHostFree(words);
        up = k = dn = nullptr; words = nullptr;
    }
};

struct band_state { size_t prev, runbits, cf; };

struct encs {
    size_t xsize, ysize, nbands, stride;
    uint64_t order, quanta;
    band_state band[QB3_MAXBANDS];
    size_t cband[QB3_MAXBANDS];
    int error;
    qb3_mode mode;
    qb3_dtype type;
    bool away;
    int ix_chunk;           // qb3x_set_encoder_index_chunk: embed the restart table ("ix" chunks); 2: with block lengths
    DevBuf d_img, d_out, d_ws, d_q, d_idx, d_rle;      // d_rle: workspace of the RLE0 passes (k_rle0.hip)
    Stager stager, stager2;                            // (stager2: the download ring of a pipelined host call)
    Pipe pipe;                                         // ... its streams and events
};

struct decs {
    size_t xsize, ysize, nbands, stride;
    uint64_t order, quanta;
    int error, stage;
    uint8_t cband[QB3_MAXBANDS];
    qb3_mode mode;
    qb3_dtype type;
    uint8_t *s_in;
    size_t s_size;
    uint8_t *s_start;       // the pointer given to qb3_read_start
    bool saw_cb;            // a CB chunk was present
    unsigned compat;
    size_t hdr_avail;       // bytes readable at s_start (the whole stream, or the header copy given to qb3x_read_start)
    size_t ix_off;          // restart table found in the container: offset of its first chunk from s_start (0: none)
    uint32_t ix_K, ix_blocks, ix_E, ix_per_chunk;
    bool ix_bl;             // ... its entries carry block lengths
    bool ix_pads, ix_bad;   // pad chunks behind the table chunks (version 2); the chunks seen do not form one table
    uint32_t ix_ver;        // version of the table's chunks (3: each carries a check of its entries)
    bool ix_heads_unchecked;    // the parser stepped over a regular table in one go: the chunk heads behind the first are checked on the device
    bool hdr_short;         // qb3_read_info read beyond the host copy of the header (whatever it then made of the zeros it got)
    size_t ix_need_off;     // ... and would have, but the bytes at this offset from s_start (the "DT" behind the table) are not on the host (0: no)
    std::vector<uint8_t> own_head, win2;    // qb3x_read_start_device: the handle's own copy of the container's first bytes, and of a few bytes further on
    size_t win2_off = 0;    // ... at this offset from s_start
    std::vector<uint8_t> tile_ok;   // qb3x_decode_tiles: per tile outcome of the last call
    uint32_t last_status = 0;       // status bits of the last decode call (qb3x_last_decode_status; tiles: of all tiles together)
    DevBuf d_in, d_img, d_ws, d_ix, d_rle, d_tab;      // d_rle: RLE0 workspace (+ the packed bytes of a host call); d_tab: the unit-length table a plain 8-bit stream is walked through
    Stager stager, stager2;                            // (stager2: the download ring of a pipelined host call)
    Pipe pipe;                                         // ... its streams and events
};

// ---------------------------------------------------------------- small host bit writer for headers
struct HdrWriter {
    uint8_t *d; size_t n = 0;
    explicit HdrWriter(uint8_t *dst) : d(dst) {}
    void put(uint64_t v, unsigned bytes) { for (unsigned i = 0; i < bytes; i++) d[n++] = (uint8_t)(v >> (8 * i)); }
    void sig(const char *s) { d[n++] = (uint8_t)s[0]; d[n++] = (uint8_t)s[1]; }
};

// reference QB3encode.cpp:189-268: main header, then CB / QV / SC chunks as needed, then DT.
// with_dt = false: the caller continues the header (the restart-table chunks and "DT" are written on the device)
static size_t write_headers(const encs *p, uint8_t *dst, bool with_dt = true) {
    HdrWriter w(dst);
    w.put(0x80334251u, 4);
    w.put(p->xsize - 1, 2); w.put(p->ysize - 1, 2); w.put(p->nbands - 1, 1);
    w.put((uint8_t)p->type, 1); w.put((uint8_t)p->mode, 1);
    bool diff = false;
    for (size_t c = 0; c < p->nbands; c++) diff |= p->cband[c] != c;
    if (p->mode != QB3M_STORED && diff) {
        w.sig("CB"); w.put(p->nbands, 2);
        for (size_t c = 0; c < p->nbands; c++) w.put(p->cband[c], 1);
    }
    if (p->quanta >= 2) {
        unsigned qb = 1 + topbit(p->quanta) / 8;
        w.sig("QV"); w.put(qb, 2); w.put(p->quanta, qb);
    }
    if (p->order != ZCURVE && p->mode != QB3M_STORED) {
        w.sig("SC"); w.put(8, 2); w.put(p->order ? p->order : HILBERT, 8);
    }
    if (with_dt) w.sig("DT");
    return w.n;
}

static size_t raw_size(const encs *p) { return p->xsize * p->ysize * p->nbands * szof(p->type); }

// ---------------------------------------------------------------- encoder handle
// No C++ exception crosses the C ABI: a failed allocation (std::vector, std::thread) inside a call is an error return with
// a message for qb3x_last_error, not std::terminate in the caller's process
template <class R, class F> static R abi_guard(R fail, F &&f) noexcept {
    try { return f(); }
    catch (const std::exception &e) { set_error(e.what(), -1); }
    catch (...) { set_error("C++ exception inside the library", -1); }
    return fail;
}

QB3_API encsp qb3_create_encoder(size_t w, size_t h, size_t b, qb3_dtype dt) {
    if (w == 0 || w > 0x10000 || h == 0 || h > 0x10000 || b == 0 || b > QB3_MAXBANDS || (int)dt < 0 || (int)dt > (int)QB3_I64)
        return nullptr;
    encs *p = new (std::nothrow) encs();
    if (!p) return nullptr;
    p->xsize = w; p->ysize = h; p->nbands = b; p->type = dt;
    p->stride = 0; p->order = 0; p->quanta = 1; p->away = false; p->mode = QB3M_DEFAULT; p->error = 0;
    { const char *e = getenv("QB3X_INDEX_CHUNK"); p->ix_chunk = e ? (atoi(e) >= 2 ? 2 : atoi(e) != 0) : 0; }
    for (size_t c = 0; c < QB3_MAXBANDS; c++) p->cband[c] = c < b ? c : 0;
    if (b == 3 || b == 4) p->cband[0] = p->cband[2] = 1;
    qb3_reset_encoder(p);
    return p;
}

QB3_API void qb3_reset_encoder(encsp p) {
    for (size_t c = 0; c < QB3_MAXBANDS; c++) p->band[c].prev = p->band[c].runbits = p->band[c].cf = 0;
    p->error = 0;
}

QB3_API void qb3_destroy_encoder(encsp p) {
    if (!p) return;
    release_all(p->d_img, p->d_out, p->d_ws, p->d_q, p->d_idx, p->d_rle);
    p->stager.release(); p->stager2.release(); p->pipe.release();
    delete p;
}

QB3_API bool qb3_set_encoder_coreband(encsp p, size_t b, size_t *bands) {
    if (b != p->nbands) return false;
    for (size_t i = 0; i < b; i++) p->cband[i] = (uint8_t)((bands[i] < b) ? bands[i] : i);
    for (size_t i = 0; i < b; i++) if (p->cband[i] != i) p->cband[p->cband[i]] = p->cband[i];
    for (size_t i = 0; i < b; i++) bands[i] = p->cband[i];
    return true;
}

QB3_API void qb3_set_encoder_stride(encsp p, size_t stride) { p->stride = stride; }

QB3_API bool qb3_set_encoder_quanta(encsp p, uint64_t q, bool away) {
    if (q < 1) return false;
    p->quanta = q; p->away = away;
    if (q == 1) return true;
    // the reference's fall-through range switch (QB3encode.cpp:96-107): a type is checked against its own
    // limit and against the limits of every wider type listed after it
    static const int order[7] = { QB3_I8, QB3_U8, QB3_I16, QB3_U16, QB3_I32, QB3_U32, QB3_I64 };
    static const uint64_t lim[7] = { 0x7f, 0xff, 0x7fff, 0xffff, 0x7fffffff, 0xffffffffull, 0x7fffffffffffffffull };
    bool bad = false;
    int start = -1;
    for (int i = 0; i < 7; i++) if (order[i] == (int)p->type) start = i;
    for (int i = start; i >= 0 && i < 7; i++) bad |= q > lim[i];
    return !bad;
}

// reference QB3encode.cpp:112-118
static size_t max_encoded_size_ref(const encs *p) {
    size_t n = 16 * ((p->xsize + 3) / 4) * ((p->ysize + 3) / 4) * p->nbands;
    double bits_per_value = 17.0 / 16.0 + 8 * szof(p->type);
    return 1024 + static_cast<size_t>(bits_per_value * n / 8);
}
static Geometry make_geometry(size_t w, size_t h, size_t bands, int dtype, size_t stride, uint64_t order, int mode,
                              const size_t *cband_sz, const uint8_t *cband_u8);
static bool is_rle_mode(int m) { return m == QB3M_RLE || m == QB3M_CF_RLE || m == QB3M_RLE_H || m == QB3M_CF_RLE_H; }
// bytes the restart-table chunks add to a container of this handle (0: none would be written)
static size_t ix_room(const encs *p) {
    if (!p->ix_chunk || p->xsize < 4 || p->ysize < 4 || p->xsize * p->ysize <= 16) return 0;
    // The bound must not depend on the mode (not even on QB3M_STORED, where a raw fallback leaves a handle: a caller that
    // sizes its buffer again then, and sets a coding mode afterwards, must not get less than the next call writes): the reference's callers size the buffer right after qb3_create_encoder and BEFORE
    // qb3_set_encoder_mode (reference cqb3.cpp:405-464, test_qb3.cpp:84-102).  The largest table any mode would write for this
    // raster: FTL and BASE streams share a layout, the common-factor modes have another.
    size_t room = 0;
    for (int m : {(int)QB3M_FTL, (int)QB3M_CF_H}) {
        const Geometry g = make_geometry(p->xsize, p->ysize, p->nbands, p->type, p->stride, p->order, m, p->cband, nullptr);
        room = std::max(room, ix_total_bytes(ix_layout(g, p->ix_chunk)));
    }
    return room;
}
QB3_API size_t qb3_max_encoded_size(const encsp p) { return max_encoded_size_ref(p) + ix_room(p); }

QB3_API qb3_mode qb3_set_encoder_mode(encsp p, qb3_mode mode) {
    if ((int)mode >= 0 && (int)mode < (int)QB3M_END) p->mode = mode;
    if ((int)p->mode <= (int)QB3M_CF_RLE) p->order = ZCURVE;
    return p->mode;
}

QB3_API int qb3_get_encoder_state(encsp p) { return p->error; }

// ---------------------------------------------------------------- geometry helpers
static CodecMode codec_mode(int mode) {
    if (mode == QB3M_FTL) return CM_FTL;
    if (mode == QB3M_BASE_H || mode == QB3M_BASE_Z || mode == QB3M_RLE || mode == QB3M_RLE_H) return CM_BASE;   // RLE0 only wraps the BASE stream
    return CM_BEST;
}

static Geometry make_geometry(size_t w, size_t h, size_t bands, int dtype, size_t stride, uint64_t order, int mode,
                              const size_t *cband_sz, const uint8_t *cband_u8) {
    Geometry g;
    memset(&g, 0, sizeof(g));
    g.w = (uint32_t)w; g.h = (uint32_t)h; g.bands = (uint32_t)bands; g.tsz = (uint32_t)szof(dtype);
    g.stride = stride ? stride : w * bands;
    g.order = order ? order : HILBERT;
    g.nbx = (uint32_t)((w + 3) / 4); g.nby = (uint32_t)((h + 3) / 4);
    g.nblocks = (uint64_t)g.nbx * g.nby;
    g.mode = codec_mode(mode);
    g.ulen_sz = ulen_size_for(g.tsz, g.mode, g.bands);
    for (size_t c = 0; c < bands; c++) g.cband[c] = cband_sz ? (uint8_t)cband_sz[c] : cband_u8[c];
    g.seg_blocks = seg_blocks_for(g);
    g.nseg = (g.nblocks + g.seg_blocks - 1) / g.seg_blocks;
    return g;
}

// narrow-image remap (reference QB3encode.cpp:351-389, implemented per its intent; the reference itself has
// a use-after-scope there).  Returns the packed pixels, sets the stand-in dimensions.
static std::vector<uint8_t> remap_small(const uint8_t *src, size_t w, size_t h, size_t pix, size_t stride_bytes,
                                        size_t &nw, size_t &nh) {
    const size_t ngroups = (w * h + 15) / 16;
    std::vector<uint8_t> t(ngroups * 16 * pix, 0);
    uint8_t *d = t.data();
    if (w < 4) {
        for (size_t y = 0; y < h; y++, d += w * pix) memcpy(d, src + y * stride_bytes, w * pix);
        nw = 4; nh = ngroups * 4;
    } else {
        for (size_t x = 0; x < w; x++)
            for (size_t y = 0; y < h; y++, d += pix) memcpy(d, src + y * stride_bytes + x * pix, pix);
        nw = ngroups * 4; nh = 4;
    }
    return t;
}

// (RLE0, reference QB3encode.cpp:271-332 / QB3decode.cpp:267-307, runs on the device: k_rle0.hip)

// ---------------------------------------------------------------- encode
static size_t stored_encode_host(encsp p, const void *source, void *destination) {
    uint8_t *d = (uint8_t *)destination;
    p->mode = QB3M_STORED;
    const size_t hdr = write_headers(p, d);
    if (p->error) return 0;
    const size_t tsz = szof(p->type), line = p->xsize * p->nbands * tsz;
    const size_t stride = (p->stride ? p->stride : p->xsize * p->nbands) * tsz;
    for (size_t y = 0; y < p->ysize; y++) memcpy(d + hdr + y * line, (const uint8_t *)source + y * stride, line);
    return hdr + raw_size(p);
}

#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(#x, (int)e_); p->error = QB3E_LIBERR; return 0; } } while (0)

// Runs the block coder on a device image.  d_out mirrors the destination buffer: the stream starts at byte
// `hdr`.  On success *bits receives the stream length; the handle's band state is updated when `carry`.
// d_index may be null.  Synchronises the stream.
static bool encode_blocks_device(encsp p, const Geometry &g, const void *d_img, uint8_t *d_out, size_t hdr,
                                 void *d_index, hipStream_t st, bool carry, uint64_t *bits, const uint8_t *hdrbytes,
                                 size_t hdr_stamp, const IxTable &ix, int *zero_run = nullptr) {
    EncPlan plan = plan_encode(g);
    if (!p->d_ws.ensure(plan.ws_bytes)) return false;
    BandState bs;
    memset(&bs, 0, sizeof(bs));
    for (size_t c = 0; c < p->nbands; c++) {
        bs.prev[c] = p->band[c].prev; bs.cf[c] = p->band[c].cf; bs.rung[c] = (uint8_t)p->band[c].runbits;
    }
    uint32_t *out32 = (uint32_t *)(d_out + (hdr & ~(size_t)3));
    EncResult res;
    const uint8_t *dres = (const uint8_t *)p->d_ws.p + plan.ws_bytes - sizeof(EncResult);
    if (launch_encode(g, plan, d_img, out32, (uint32_t)(8 * (hdr & 3)), bs, p->d_ws.p, d_index, st, TileBatch(), hdrbytes, (uint32_t)hdr_stamp, ix, zero_run != nullptr)) return false;
    const hipError_t e = fetch_small(&res, dres, sizeof(res), st);
    if (e != hipSuccess) { set_error("encode kernels", (int)e); return false; }
    prof_collect();
    *bits = res.total_bits;
    // (a chunk of the stream holds plan.nbp blocks of at least two bits a unit)
    { static const bool dbg = getenv("QB3_DEBUG_RLE") != nullptr; if (dbg && zero_run) fprintf(stderr, "encode: zero_run %llu ff_pairs %llu zero_dwords %llu (%u chunks)\n", (unsigned long long)res.zero_run, (unsigned long long)res.ff_pairs, (unsigned long long)res.zero_dwords, plan.nchunks); }
    if (zero_run) *zero_run = rle0_may_win(res) ? (rle0_no_uniform_chunk(res, (uint64_t)plan.nbp * g.bands / 4) ? 2 : 1) : 0;      // (2: and no 4 KB of the stream hold one byte value only)
    if (carry)
        for (size_t c = 0; c < p->nbands; c++) {
            p->band[c].prev = (size_t)res.prev[c]; p->band[c].runbits = res.rung[c]; p->band[c].cf = (size_t)res.cf[c];
        }
    return true;
}

// Puts the caller's mode back on every exit but the ones that are meant to change it (the reference leaves a handle
// at QB3M_STORED after a raw fallback, QB3encode.cpp:464, and demotes RLE modes only for the block pass, :495-506).
struct ModeGuard {
    encs *p; qb3_mode mode; bool armed = true;
    ModeGuard(encs *h) : p(h), mode(h->mode) {}
    ~ModeGuard() { if (armed) p->mode = mode; }
};


// ---------------------------------------------------------------- qb3_encode, pipelined
// The raster goes up the link in slices; as soon as the rows of a STRIP (a scan group of chunks, about 50 MB of an 8-bit RGB
// raster) are in device memory the strip is coded, scanned, moved into place and sealed (launch_encode_strip) -- its bits start
// where the strips before it ended, the dependency only points backwards -- and the part of the stream that is final comes
// down the link while later strips are still going up.  Three streams, two rings of pinned slices, host copies by the pool.
// Returns 1: the stream (and its table) is in host_dst behind the header's place, *bits_out and the handle's band state are
// set; 0: not taken (the caller goes the one-after-the-other way); -1: failed (p->error set).
static int encode_pipelined(encsp p, const Geometry &g, const void *host_src, void *host_dst, size_t src_bytes, size_t line, uint8_t *out_dev, size_t hdr,
                            const uint8_t *hdrbuf, size_t hdr_stamp, const IxTable &ixt, void *d_index, bool carry, uint64_t *bits_out) {
    using qb3host::CopyPool;
    constexpr size_t SLICE = Stager::SLICE, NSLOT = Stager::NSLOT;
    static const bool dbg = getenv("QB3_DEBUG_PIPE") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto ms_since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    const EncPlan plan = plan_encode(g);
    if (!encode_strips_ok(g, plan)) return 0;
    const uint32_t nstrips = encode_strip_count(plan);
    if (nstrips < 3 || nstrips + 64 > Pipe::WORDS) return 0;
    if (!p->pipe.init() || !p->pipe.events((size_t)nstrips + 1) || !p->stager.init() || !p->stager2.init()) return 0;
    if (!p->d_img.ensure(src_bytes) || !p->d_ws.ensure(plan.ws_bytes)) { p->error = QB3E_LIBERR; return -1; }
    BandState bs;
    memset(&bs, 0, sizeof(bs));
    for (size_t c = 0; c < p->nbands; c++) { bs.prev[c] = p->band[c].prev; bs.cf[c] = p->band[c].cf; bs.rung[c] = (uint8_t)p->band[c].runbits; }
    uint32_t *out32 = (uint32_t *)(out_dev + (hdr & ~(size_t)3));
    const uint32_t out_bit0 = (uint32_t)(8 * (hdr & 3));
    // raster bytes a strip needs in device memory: the rows of its last block (the block before its first one is in the strip before)
    std::vector<size_t> need(nstrips);
    for (uint32_t s = 0; s < nstrips; s++) {
        const uint64_t blocks = std::min<uint64_t>(g.nblocks, encode_strip_blocks(plan, s));
        const uint64_t brow = (blocks - 1) / g.nbx;
        need[s] = std::min(src_bytes, (size_t)std::min<uint64_t>(g.h, 4 * (brow + 1)) * line);
    }
    need[nstrips - 1] = src_bytes;
    struct Slice { const uint8_t *dev; uint8_t *host; size_t n; };
    std::vector<Slice> dn;
    const size_t n_up = (src_bytes + SLICE - 1) / SLICE;
    CopyPool &pool = CopyPool::get();
    CopyPool::Batch b_up[NSLOT], b_dn[NSLOT];
    Stager &r1 = p->stager, &r2 = p->stager2;
    hipStream_t sU = p->pipe.up, sK = p->pipe.k, sD = p->pipe.dn;
    uint64_t *totals = p->pipe.words;                       // [s]: stream bits behind strip s; behind them the EncResult
    EncResult *res_host = (EncResult *)(totals + nstrips);
    const uint8_t *dres = (const uint8_t *)p->d_ws.p + plan.ws_bytes - sizeof(EncResult);
    size_t up_started = 0, up_enq = 0, bytes_enq = 0, dn_issued = 0, dn_copy = 0, dn_freed = 0, x_prev = hdr;
    uint32_t next_strip = 0, known = 0;
    bool tail_seen = false, launch_failed = false;
    hipError_t e = hipSuccess;
    auto ok = [&] { return e == hipSuccess && !launch_failed; };
    auto add_span = [&](size_t a, size_t b) {               // bytes [a, b) of the container are final in device memory
        for (size_t off = a; off < b; off += SLICE) dn.push_back({out_dev + off, (uint8_t *)host_dst + off, std::min(SLICE, b - off)});
    };
    const double t_setup = ms_since();
    double t_up_done = 0, t_first_dn = 0;
    while (ok() && (up_enq < n_up || !tail_seen || dn_freed < dn.size())) {
        bool progress = false;
        // ---- up
        if (up_started < n_up && up_started - up_enq < 2 && (up_started < NSLOT || hipEventQuery(r1.ev(up_started)) == hipSuccess)) {
            const size_t off = up_started * SLICE;
            pool.submit(r1.slot(up_started), (const uint8_t *)host_src + off, std::min(SLICE, src_bytes - off), b_up[up_started % NSLOT]);
            up_started++; progress = true;
        }
        if (up_enq < up_started && pool.done(b_up[up_enq % NSLOT])) {
            const size_t off = up_enq * SLICE, n = std::min(SLICE, src_bytes - off);
            e = hipMemcpyAsync((uint8_t *)p->d_img.p + off, r1.slot(up_enq), n, hipMemcpyHostToDevice, sU);
            if (e == hipSuccess) e = hipEventRecord(r1.ev(up_enq), sU);
            bytes_enq += n;
            up_enq++; progress = true;
            if (dbg && up_enq == n_up) t_up_done = ms_since();
            while (ok() && next_strip < nstrips && need[next_strip] <= bytes_enq) {
                hipEvent_t ev_u = p->pipe.ev[nstrips];     // (one event for "the rows are on their way": recorded and waited for at once)
                e = hipEventRecord(ev_u, sU);
                if (e == hipSuccess) e = hipStreamWaitEvent(sK, ev_u, 0);
                if (e != hipSuccess) break;
                if (launch_encode_strip(g, plan, p->d_img.p, out32, out_bit0, bs, p->d_ws.p, d_index, sK, hdrbuf, (uint32_t)hdr_stamp, ixt, next_strip)) { launch_failed = true; break; }
                e = hipMemcpyAsync(&totals[next_strip], encode_strip_total(g, plan, p->d_ws.p, next_strip), 8, hipMemcpyDeviceToHost, sK);
                if (e == hipSuccess && next_strip + 1 == nstrips) {        // behind the last strip: index positions, the table, the result words
                    if (launch_encode_tail(g, plan, p->d_img.p, out32, out_bit0, bs, p->d_ws.p, d_index, sK, hdrbuf, (uint32_t)hdr_stamp, ixt)) { launch_failed = true; break; }
                    e = hipMemcpyAsync(res_host, dres, sizeof(EncResult), hipMemcpyDeviceToHost, sK);
                }
                if (e == hipSuccess) e = hipEventRecord(p->pipe.ev[next_strip], sK);
                next_strip++;
            }
        }
        // ---- a strip is done: what is final behind it (all but the dword its end falls into; the last strip: everything, and the table)
        if (ok() && known < next_strip) {
            const hipError_t q = hipEventQuery(p->pipe.ev[known]);
            if (q == hipSuccess) {
                const uint64_t P = totals[known];
                const bool last = known + 1 == nstrips;
                const size_t x = last ? hdr + (size_t)((P + 7) / 8) : (hdr & ~(size_t)3) + 4 * (size_t)((out_bit0 + P) >> 5);
                if (x > x_prev) { add_span(x_prev, x); x_prev = x; }
                if (last) { if (hdr > hdr_stamp) add_span(hdr_stamp, hdr); tail_seen = true; }
                known++; progress = true;
            } else if (q != hipErrorNotReady) e = q;
        }
        // ---- down
        if (ok() && dn_issued < dn.size() && dn_issued - dn_freed < NSLOT) {
            e = hipMemcpyAsync(r2.slot(dn_issued), dn[dn_issued].dev, dn[dn_issued].n, hipMemcpyDeviceToHost, sD);
            if (e == hipSuccess) e = hipEventRecord(r2.ev(dn_issued), sD);
            dn_issued++; progress = true;
        }
        if (ok() && dn_copy < dn_issued) {
            const hipError_t q = hipEventQuery(r2.ev(dn_copy));
            if (q == hipSuccess) { if (dbg && !dn_copy) t_first_dn = ms_since(); pool.submit(dn[dn_copy].host, r2.slot(dn_copy), dn[dn_copy].n, b_dn[dn_copy % NSLOT]); dn_copy++; progress = true; }
            else if (q != hipErrorNotReady) e = q;
        }
        while (dn_freed < dn_copy && pool.done(b_dn[dn_freed % NSLOT])) { dn_freed++; progress = true; }
        if (!progress && !pool.help_one()) std::this_thread::yield();
    }
    (void)hipGetLastError();                                // (hipErrorNotReady of the queries is not an error)
    for (size_t i = 0; i < NSLOT; i++) { pool.wait(b_up[i]); pool.wait(b_dn[i]); }
    p->pipe.sync();
    if (dbg) fprintf(stderr, "encode_pipelined: setup %.2f ms, last upload enqueued %.2f, first slice down %.2f, done %.2f (%u strips, %zu + %zu slices)\n", t_setup, t_up_done, t_first_dn, ms_since(), nstrips, n_up, dn.size());
    if (!ok()) { if (!launch_failed) set_error("pipelined encode", (int)e); p->error = QB3E_LIBERR; return -1; }
    prof_collect();
    *bits_out = res_host->total_bits;
    if (carry)
        for (size_t c = 0; c < p->nbands; c++) {
            p->band[c].prev = (size_t)res_host->prev[c]; p->band[c].runbits = res_host->rung[c]; p->band[c].cf = (size_t)res_host->cf[c];
        }
    return 1;
}

// Shared by qb3_encode (host buffers) and qb3x_encode_device (device buffers).
// host_src/host_dst are null in the device flavour; d_src/d_dst are null in the host flavour.
static size_t encode_common(encsp p, const void *host_src, void *host_dst, const void *d_src, void *d_dst,
                            void *d_index, hipStream_t st) {
    const bool on_host = host_src != nullptr;
    const size_t tsz = szof(p->type), line = p->xsize * p->nbands * tsz;
    const size_t src_stride_bytes = (p->stride ? p->stride : p->xsize * p->nbands) * tsz;
    const size_t src_span = src_stride_bytes * (p->ysize - 1) + line;      // bytes from the first to the last pixel
    if (p->xsize * p->ysize <= 16) {        // tiny images are stored (reference QB3encode.cpp:490)
        if (on_host) return stored_encode_host(p, host_src, host_dst);
        std::vector<uint8_t> tmp(src_span), out(64 + raw_size(p));
        if (!device_ok()) { p->error = QB3E_LIBERR; return 0; }
        HIPOK(hipMemcpyAsync(tmp.data(), d_src, tmp.size(), hipMemcpyDeviceToHost, st));
        HIPOK(hipStreamSynchronize(st));
        size_t n = stored_encode_host(p, tmp.data(), out.data());
        HIPOK(hipMemcpyAsync(d_dst, out.data(), n, hipMemcpyHostToDevice, st));
        HIPOK(hipStreamSynchronize(st));
        return n;
    }
    ModeGuard guard(p);
    const qb3_mode mode = p->mode;
    const bool rle = is_rle_mode(mode);
    const size_t ixroom = mode == QB3M_STORED ? 0 : ix_room(p);   // (a handle left at STORED writes no table)
    if (rle) p->mode = (qb3_mode)((int)mode - 2);       // RLE is a post pass over the base mode's stream
    uint8_t hdrbuf[80];
    size_t hdr = write_headers(p, hdrbuf);
    if (p->error) return 0;                               // stale error blocks the handle until reset
    if (!device_ok()) { p->error = QB3E_LIBERR; return 0; }

    // geometry, with the narrow-image stand-in where needed
    size_t w = p->xsize, h = p->ysize, stride = p->stride;
    const void *img_dev = d_src;
    std::vector<uint8_t> small;
    const bool narrow = w < 4 || h < 4;
    if (narrow) {
        std::vector<uint8_t> tmp;
        const uint8_t *hs = (const uint8_t *)host_src;
        if (!on_host) {
            tmp.resize(src_span);
            HIPOK(hipMemcpyAsync(tmp.data(), d_src, tmp.size(), hipMemcpyDeviceToHost, st));
            HIPOK(hipStreamSynchronize(st));
            hs = tmp.data();
        }
        small = remap_small(hs, p->xsize, p->ysize, p->nbands * tsz, src_stride_bytes, w, h);
        stride = 0;
    }
    // a large raster in host memory, coded as it is: upload, coding and download strip by strip, all three at once (encode_pipelined)
    static const bool no_pipeline = [] { const char *e = getenv("QB3_NO_PIPELINE"); return e && e[0] && e[0] != '0'; }();
    bool want_pipe = on_host && !narrow && !rle && p->quanta < 2 && src_stride_bytes == line && src_span >= ((size_t)64 << 20) && !no_pipeline;
    auto upload_now = [&]() -> bool {
        const uint8_t *hs = narrow ? small.data() : (const uint8_t *)host_src;
        const size_t bytes = narrow ? small.size() : src_span;
        return p->d_img.ensure(bytes) && upload(p->stager, p->d_img.p, hs, bytes, st);
    };
    if (on_host || narrow) {
        if (!want_pipe && !upload_now()) { p->error = QB3E_LIBERR; return 0; }
        if (want_pipe && !p->d_img.ensure(src_span)) { p->error = QB3E_LIBERR; return 0; }
        img_dev = p->d_img.p;
    }
    Geometry g = make_geometry(w, h, p->nbands, p->type, stride, p->order, p->mode, p->cband, nullptr);
    if (p->quanta >= 2) {
        // quantise into a compact device copy; like the reference (QB3encode.cpp:405-455) the band state then
        // lives on a copy of the handle and is not carried back
        if (!p->d_q.ensure((size_t)g.w * g.h * g.bands * tsz)) { p->error = QB3E_LIBERR; return 0; }
        if (launch_quantize(p->d_q.p, img_dev, g, (int)p->type, p->quanta, p->away, st)) { p->error = QB3E_LIBERR; return 0; }
        img_dev = p->d_q.p;
        g.stride = (uint64_t)g.w * g.bands;
    }
    const bool carry = !narrow && p->quanta < 2;
    const size_t maxsz = max_encoded_size_ref(p);         // the reference's bound: RLE0 decisions must not depend on the table
    uint8_t *out_dev = (uint8_t *)d_dst;
    if (on_host) {
        if (!p->d_out.ensure(maxsz + ixroom + 64)) { p->error = QB3E_LIBERR; return 0; }
        out_dev = (uint8_t *)p->d_out.p;
    }
    // check_info (reference QB3encode.h:364-373); cband is kept in range by the setter
    if (g.w < 4 || g.h < 4) { p->error = 1; return 0; }

    // optional restart table inside the container (a winning RLE0 pass keeps it in front of its bytes)
    IxTable ixt;
    size_t hdr_stamp = hdr;                               // header bytes prepared on the host
    if (ixroom && !narrow) {
        ixt = ix_layout(g, p->ix_chunk);
        hdr_stamp = write_headers(p, hdrbuf, false);
        hdr = hdr_stamp + ix_total_bytes(ixt) + 2;        // chunks, then "DT": both written by enc_finish_kernel
        ixt.base = out_dev + hdr_stamp;
        if (!d_index) {                                   // the table is a sample of the index: make one
            if (!p->d_idx.ensure(index_bytes(g))) { p->error = QB3E_LIBERR; return 0; }
            d_index = p->d_idx.p;
            ixt.own_index = true;
        }
    }
    uint64_t bits = 0;
    int has_run = 1;        // (RLE0 modes: the concatenation pass counts zero runs and pairs of 0xff; rle0_may_win, qb3_dev.h)
    bool piped = false;
    if (want_pipe) {
        const int r = encode_pipelined(p, g, host_src, host_dst, src_span, line, out_dev, hdr, hdrbuf, hdr_stamp, ixt, d_index, carry, &bits);
        if (r < 0) return 0;
        piped = r > 0;
        if (!piped && !upload_now()) { p->error = QB3E_LIBERR; return 0; }
    }
    if (!piped && !encode_blocks_device(p, g, img_dev, out_dev, hdr, d_index, st, carry, &bits, hdrbuf, hdr_stamp, ixt, rle ? &has_run : nullptr)) {   // the index describes the block stream, RLE0 wrapped or not
        p->error = QB3E_LIBERR; return 0;
    }
    p->error = 0;
    const size_t len = hdr + (size_t)((bits + 7) / 8);
    const size_t len_ref = len - (ixt.base ? ix_total_bytes(ixt) : 0);      // what the reference's container measures

    if (rle) {
        // the RLE0 post pass (reference QB3encode.cpp:536-565)
        p->mode = mode;
        // ... which can only win when the stream's zero runs outweigh its pairs of 0xff (has_run: the encoder kernels counted)
        if (len_ref <= maxsz / 2 && has_run) {
            // the byte pass on the device (k_rle0.hip): its size first, the bytes only when it wins -- into a buffer of its
            // own (the passes run in parallel: not in place), then behind the RLE mode's header
            const size_t n = len - hdr;
            uint64_t rsz64 = 0;
            if (!p->d_rle.ensure(rle0_ws_bytes(n)) || rle0_device_size(out_dev + hdr, n, p->d_rle.p, false, &rsz64, st, has_run == 2)) { p->error = QB3E_LIBERR; return 0; }
            const size_t rsz = (size_t)rsz64;
            if (rsz <= maxsz - len_ref && rsz < n) {
                // the RLE mode's header, the restart table when one was asked for (its chunks and the "DT" behind them stand in the
                // buffer as written: the table describes the block stream, which the decoder sees again once it has expanded the
                // bytes), then the RLE0 bytes
                uint8_t hdr2[80];
                const size_t tbl = ixt.base ? hdr - hdr_stamp : 0;      // (chunks + "DT")
                const size_t h2 = write_headers(p, hdr2, tbl == 0);
                if (!p->d_q.ensure(rsz) || rle0_device_write(out_dev + hdr, n, p->d_rle.p, false, p->d_q.p, st)) { p->error = QB3E_LIBERR; return 0; }
                if (on_host) {
                    memcpy(host_dst, hdr2, h2);
                    if (tbl && !download(p->stager, (uint8_t *)host_dst + h2, out_dev + hdr_stamp, tbl, st)) { p->error = QB3E_LIBERR; return 0; }
                    if (!download(p->stager, (uint8_t *)host_dst + h2 + tbl, p->d_q.p, rsz, st)) { p->error = QB3E_LIBERR; return 0; }
                } else {
                    // (device flavour: out_dev IS d_dst and h2 == hdr_stamp -- the modes' headers differ in one byte -- so the table stays where it is)
                    if (tbl && h2 != hdr_stamp) HIPOK(hipMemcpyAsync((uint8_t *)d_dst + h2, out_dev + hdr_stamp, tbl, hipMemcpyDeviceToDevice, st));
                    HIPOK(hipMemcpyAsync((uint8_t *)d_dst + h2 + tbl, p->d_q.p, rsz, hipMemcpyDeviceToDevice, st));
                    HIPOK(hipMemcpyAsync(d_dst, hdr2, h2, hipMemcpyHostToDevice, st));
                    HIPOK(hipStreamSynchronize(st));
                }
                return h2 + tbl + rsz;
            }
        }
    }
    // (the restart table does not take part in the decision: the same inputs give the same kind of container)
    if (raw_size(p) > len_ref) {
        if (on_host) {
            memcpy(host_dst, hdrbuf, hdr_stamp);
            if (!piped && !download(p->stager, (uint8_t *)host_dst + hdr_stamp, out_dev + hdr_stamp, len - hdr_stamp, st)) { p->error = QB3E_LIBERR; return 0; }
        }       // device flavour: the header was written by enc_finish_kernel, in stream order
        return len;
    }
    // not worth it: raw bypass (reference QB3encode.cpp:571-573), which leaves the handle's mode at STORED
    guard.armed = false;
    if (on_host) return stored_encode_host(p, host_src, host_dst);
    p->mode = QB3M_STORED;
    const size_t h2 = write_headers(p, hdrbuf);
    HIPOK(hipMemcpyAsync(d_dst, hdrbuf, h2, hipMemcpyHostToDevice, st));
    HIPOK(hipMemcpy2DAsync((uint8_t *)d_dst + h2, line, d_src, src_stride_bytes, line, p->ysize, hipMemcpyDeviceToDevice, st));
    HIPOK(hipStreamSynchronize(st));
    return h2 + raw_size(p);
}

QB3_API size_t qb3_encode(encsp p, void *source, void *destination) {
    if (!p || !source || !destination) return 0;
    return abi_guard<size_t>(0, [&] { return encode_common(p, source, destination, nullptr, nullptr, nullptr, nullptr); });
}

QB3_API size_t qb3x_encode_device(encsp p, const void *d_src, void *d_dst, void *d_index, void *stream) {
    if (!p || !d_src || !d_dst || ((uintptr_t)d_dst & 3)) { if (p) p->error = QB3E_EINV; return 0; }
    return abi_guard<size_t>(0, [&] { return encode_common(p, nullptr, nullptr, d_src, d_dst, d_index, (hipStream_t)stream); });
}

QB3_API void qb3x_set_encoder_index_chunk(encsp p, int on) { if (p) p->ix_chunk = on >= 2 ? 2 : on != 0; }

QB3_API size_t qb3x_index_size(const encsp p) {
    if (!p || p->xsize < 4 || p->ysize < 4) return 0;
    Geometry g = make_geometry(p->xsize, p->ysize, p->nbands, p->type, p->stride, p->order, p->mode, p->cband, nullptr);
    return index_bytes(g);
}

// One tile at a time (general path: RLE modes, quantisation, narrow or tiny tiles, STORED fallbacks)
static size_t encode_tiles_loop(encsp p, const void *d_src, size_t first, size_t n, size_t src_pitch, void *d_dst, size_t dst_pitch,
                                void *d_index, size_t isz, size_t *sizes, void *stream, qb3_mode mode) {
    size_t done = 0;
    for (size_t i = first; i < first + n; i++) {
        qb3_reset_encoder(p);
        p->mode = mode;
        sizes[i] = qb3x_encode_device(p, (const uint8_t *)d_src + i * src_pitch, (uint8_t *)d_dst + i * dst_pitch,
                                      d_index ? (uint8_t *)d_index + i * isz : nullptr, stream);
        done += sizes[i] != 0;
    }
    return done;
}

static size_t encode_tiles_body(encsp p, const void *d_src, size_t n, size_t src_pitch, void *d_dst, size_t dst_pitch,
                                void *d_index, size_t *sizes, void *stream);
QB3_API size_t qb3x_encode_tiles(encsp p, const void *d_src, size_t n, size_t src_pitch, void *d_dst, size_t dst_pitch,
                                 void *d_index, size_t *sizes, void *stream) {
    return abi_guard<size_t>(0, [&] { return encode_tiles_body(p, d_src, n, src_pitch, d_dst, dst_pitch, d_index, sizes, stream); });
}
static size_t encode_tiles_body(encsp p, const void *d_src, size_t n, size_t src_pitch, void *d_dst, size_t dst_pitch,
                                void *d_index, size_t *sizes, void *stream) {
    if (!p || !d_src || !d_dst || !sizes || (dst_pitch & 3) || ((uintptr_t)d_dst & 3)) return 0;
    const size_t isz = d_index ? qb3x_index_size(p) : 0;
    const qb3_mode mode = p->mode;
    const size_t tsz = szof(p->type);
    hipStream_t st = (hipStream_t)stream;
    // batched path: every tile of the call goes through ONE set of kernel launches (blockIdx.y = tile) and one
    // host synchronisation.  Anything unusual takes the one-by-one path.
    const bool batchable = !is_rle_mode(mode) && mode != QB3M_STORED && p->quanta < 2 && p->xsize >= 4 && p->ysize >= 4 &&
                           p->xsize * p->ysize > 16 && !p->error && device_ok();
    if (!batchable) {
        const size_t k = encode_tiles_loop(p, d_src, 0, n, src_pitch, d_dst, dst_pitch, d_index, isz, sizes, stream, mode);
        if (mode != QB3M_STORED) p->mode = mode;
        return k;
    }

    uint8_t hdrbuf[80];
    size_t hdr = write_headers(p, hdrbuf);
    Geometry g = make_geometry(p->xsize, p->ysize, p->nbands, p->type, p->stride, p->order, p->mode, p->cband, nullptr);
    EncPlan plan = plan_encode(g);
    size_t wsp = (plan.ws_bytes + 255) & ~(size_t)255;
    size_t batch = (size_t)8 << 30 >= wsp ? ((size_t)8 << 30) / wsp : 1;     // keep the workspace under 8 GiB
    if (batch > n) batch = n;
    if (batch > 65535) batch = 65535;
    if (!p->d_ws.ensure(batch * wsp)) { p->error = QB3E_LIBERR; return 0; }
    // self-indexing containers (qb3x_set_encoder_index_chunk): every tile gets its own restart table, at the same place
    IxTable ixt;
    size_t hdr_stamp = hdr, ix_bytes = 0, isz_all = isz;
    void *index_all = d_index;
    if (ix_room(p)) {                                     // (batchable: the mode is not QB3M_STORED)
        ixt = ix_layout(g, p->ix_chunk);
        hdr_stamp = write_headers(p, hdrbuf, false);
        ix_bytes = ix_total_bytes(ixt);
        hdr = hdr_stamp + ix_bytes + 2;                   // chunks, then "DT": both written by enc_finish_kernel
        if (!index_all) {                                 // the table is a sample of the index: make one per tile of a batch
            isz_all = (index_bytes(g) + 7) & ~(size_t)7;
            if (!p->d_idx.ensure(batch * isz_all)) { p->error = QB3E_LIBERR; return 0; }
            index_all = p->d_idx.p;
            ixt.own_index = true;
        }
    }
    BandState bs;
    memset(&bs, 0, sizeof(bs));             // tiles are independent streams: every tile starts from the reset state
    std::vector<EncResult> res(batch);
    size_t done = 0;
    for (size_t first = 0; first < n; first += batch) {
        const size_t cnt = (n - first < batch) ? n - first : batch;
        TileBatch tb;
        tb.n = (uint32_t)cnt; tb.src_pitch = src_pitch; tb.dst_pitch = dst_pitch; tb.ws_pitch = wsp; tb.idx_pitch = isz_all;
        uint8_t *out0 = (uint8_t *)d_dst + first * dst_pitch;
        if (ix_bytes) ixt.base = out0 + hdr_stamp;
        // (the caller's index array is indexed by tile; the internal one by tile of the batch)
        void *index_here = !index_all ? nullptr : (d_index ? (uint8_t *)d_index + first * isz : (uint8_t *)index_all);
        if (launch_encode(g, plan, (const uint8_t *)d_src + first * src_pitch, (uint32_t *)(out0 + (hdr & ~(size_t)3)), (uint32_t)(8 * (hdr & 3)), bs,
                          p->d_ws.p, index_here, st, tb, hdrbuf, (uint32_t)hdr_stamp, ixt)) { p->error = QB3E_LIBERR; return done; }
        const uint8_t *dres = (const uint8_t *)p->d_ws.p + plan.ws_bytes - sizeof(EncResult);
        hipError_t e = hipMemcpy2DAsync(res.data(), sizeof(EncResult), dres, wsp, sizeof(EncResult), cnt, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_error("encode kernels (tiles)", (int)e); p->error = QB3E_LIBERR; return done; }
        prof_collect();
        const size_t raw = p->xsize * p->ysize * p->nbands * tsz;
        for (size_t i = 0; i < cnt; i++) {
            const size_t len = hdr + (size_t)((res[i].total_bits + 7) / 8);
            if (raw > len - ix_bytes) { sizes[first + i] = len; done++; }         // (the table does not take part in the decision)
            else done += encode_tiles_loop(p, d_src, first + i, 1, src_pitch, d_dst, dst_pitch, d_index, isz, sizes, stream, mode);   // STORED fallback
        }
        // handle state as after a loop over the tiles: the state left by the last one
        for (size_t c = 0; c < p->nbands; c++) {
            p->band[c].prev = (size_t)res[cnt - 1].prev[c]; p->band[c].runbits = res[cnt - 1].rung[c]; p->band[c].cf = (size_t)res[cnt - 1].cf[c];
        }
    }
    p->mode = mode;         // a raw fallback of one tile must not turn the handle (and the next call) to QB3M_STORED
    p->error = 0;
    return done;
}

// ---------------------------------------------------------------- decoder handle
QB3_API void qb3_destroy_decoder(decsp p) {
    if (!p) return;
    release_all(p->d_in, p->d_img, p->d_ws, p->d_ix, p->d_rle, p->d_tab);
    p->stager.release(); p->stager2.release(); p->pipe.release();
    delete p;
}
QB3_API size_t qb3_decoded_size(const decsp p) { return p->xsize * p->ysize * p->nbands * szof(p->type); }
QB3_API qb3_dtype qb3_get_type(const decsp p) { return p->type; }
QB3_API qb3_mode qb3_get_mode(const decsp p) { return (2 == p->stage) ? p->mode : QB3M_INVALID; }
QB3_API uint64_t qb3_get_quanta(const decsp p) { return (2 == p->stage) ? p->quanta : 0; }
QB3_API uint64_t qb3_get_order(const decsp p) { return (p->stage != 2) ? 0 : (p->order ? p->order : ZCURVE); }
QB3_API bool qb3_get_coreband(const decsp p, size_t *coreband) {
    if (p->stage != 2) return false;
    for (size_t c = 0; c < p->nbands; c++) coreband[c] = p->cband[c];
    return true;
}
QB3_API void qb3_set_decoder_stride(decsp p, size_t stride) { p->stride = stride; }
QB3_API void qb3x_set_decoder_compat(decsp p, unsigned flags) { if (p) p->compat = flags; }

// reference QB3decode.cpp:130-172.  hdr_avail: bytes readable at `source` (the device flavour may hand over a copy of
// the container's head only, with source_size still the size of the whole container)
static decsp read_start_impl(void *source, size_t hdr_avail, size_t source_size, size_t *image_size) {
    if (!source || source_size < 15 || hdr_avail < 15 || !image_size) return nullptr;
    const uint8_t *b = (const uint8_t *)source;
    if (b[0] != 'Q' || b[1] != 'B' || b[2] != '3' || b[3] != 0x80) return nullptr;
    const size_t nb = 1 + (size_t)b[8];
    const int type = b[9], mode = b[10];
    if (nb > QB3_MAXBANDS || (mode >= (int)QB3M_END && mode != (int)QB3M_STORED) || ((b[11] | b[12]) & 0x80) || type > (int)QB3_I64)
        return nullptr;
    decs *p = new decs();
    p->xsize = 1 + (size_t)(b[4] | (b[5] << 8));
    p->ysize = 1 + (size_t)(b[6] | (b[7] << 8));
    p->nbands = nb; p->type = (qb3_dtype)type; p->mode = (qb3_mode)mode;
    p->stride = 0; p->order = 0; p->quanta = 0; p->error = QB3E_OK; p->stage = 1;
    memset(p->cband, 0, sizeof(p->cband));
    p->s_start = (uint8_t *)source;
    p->s_in = p->s_start + 11; p->s_size = source_size - 11;
    p->hdr_avail = hdr_avail < source_size ? hdr_avail : source_size;
    p->saw_cb = false; p->compat = 0;
    p->ix_off = 0; p->ix_K = p->ix_blocks = p->ix_E = p->ix_per_chunk = 0; p->ix_pads = false; p->ix_bad = false; p->ix_bl = false;
    p->ix_ver = 0; p->ix_heads_unchecked = false; p->ix_need_off = 0; p->hdr_short = false;
    image_size[0] = p->xsize; image_size[1] = p->ysize; image_size[2] = p->nbands;
    if (mode <= (int)QB3M_CF_RLE) p->order = ZCURVE;
    return p;
}
QB3_API decsp qb3_read_start(void *source, size_t source_size, size_t *image_size) {
    return abi_guard<decsp>(nullptr, [&] { return read_start_impl(source, source_size, source_size, image_size); });
}
QB3_API decsp qb3x_read_start(void *header, size_t header_size, size_t stream_size, size_t *image_size) {
    return abi_guard<decsp>(nullptr, [&] { return read_start_impl(header, header_size, stream_size, image_size); });
}
// Upper bound of the bytes in front of the block stream of a container that starts with these (at least 11) bytes:
// the fixed header, the reference's chunks and this library's restart-table chunks for the worst mode.
QB3_API size_t qb3x_header_size_bound(const void *container, size_t avail) {
    const uint8_t *b = (const uint8_t *)container;
    if (!b || avail < 11 || b[0] != 'Q' || b[1] != 'B' || b[2] != '3' || b[3] != 0x80) return 0;
    const size_t w = 1 + (size_t)(b[4] | (b[5] << 8)), h = 1 + (size_t)(b[6] | (b[7] << 8)), nb = 1 + (size_t)b[8];
    const size_t tsz = szof(b[9]);
    if (!tsz || nb > QB3_MAXBANDS) return 0;
    // an entry covers at least 12 units (one common-factor segment) and takes at most 6 + bands * (1 + 2 * tsz) bytes
    const size_t units = ((w + 3) / 4) * ((h + 3) / 4) * nb, E = 6 + nb * (1 + 2 * tsz);
    const size_t K = units / 12 + 1;
    // ... and a table of 8-bit data may carry ten bits per block on top (an entry per 64 blocks)
    const size_t nblk = ((w + 3) / 4) * ((h + 3) / 4);
    const size_t bl = tsz == 1 ? (nblk / 64 + 1) * (64 * IX_BL_BEST_BYTES) : tsz == 2 ? (nblk * (nb / 4 + 1) / 64 + 1) * ((128 * IX_BL_BITS + 7) / 8)
                               : std::max((nblk * nb * IX_BL_BITS_WIDE) / 8 + (nblk / 12 + 1) * 2 + 64,      // (32/64-bit: a length per unit, an odd byte per entry)
                                          nblk * IX_BL_BEST_BYTES + 64);                                    // (... or, one band, common factor: a field per block)
    // ... and a table of the lane-per-unit decoder's rasters a field per UNIT: three bytes (common factor) or twelve bits
    const size_t blu = nblk * nb * IX_BL_BEST_BYTES + 64 * IX_BL_BEST_BYTES;
    const size_t bytes = K * E + std::max(bl, blu);
    return 128 + bytes + (bytes / 60000 + 1) * (IX_HEAD + IX_PAD);
}

static bool valid_curve(uint64_t v) {
    unsigned mask = 0;
    for (int i = 0; i < 16; i++, v >>= 4) mask |= 1u << (v & 15);
    return mask == 0xffff;
}

// reference QB3decode.cpp:176-264; chunks are byte aligned, so this walks bytes
QB3_API bool qb3_read_info(decsp p) {
    if (p->stage != 1 || p->error || !p->s_in || p->s_size < 4) {
        if (QB3E_OK == p->error) p->error = QB3E_EINV;
        return false;
    }
    const uint8_t *s = p->s_in;
    const size_t n = p->s_size;
    const size_t avail = p->hdr_avail > 11 ? p->hdr_avail - 11 : 0;           // bytes readable at s (<= n)
    size_t pos = 0;
    bool short_copy = false;                                                   // the head copy ends before the header does
    auto have = [&](size_t at) -> bool {                                       // is the byte on the host
        return at < avail || (at + 11 >= p->win2_off && at + 11 - p->win2_off < p->win2.size());
    };
    auto rd = [&](size_t at) -> unsigned {                                     // reads past the end give zeros
        if (at < avail) return s[at];
        if (at + 11 >= p->win2_off && at + 11 - p->win2_off < p->win2.size()) return p->win2[at + 11 - p->win2_off];
        if (at < n) short_copy = true;
        return 0u;
    };
    do {
        const unsigned c0 = rd(pos), c1 = rd(pos + 1), len = rd(pos + 2) | (rd(pos + 3) << 8);
        if (c0 == 'Q' && c1 == 'V') {
            if (len > 4 || len < 1) { p->error = QB3E_EINV; break; }
            pos += 4;
            uint64_t q = 0;
            for (unsigned i = 0; i < len; i++) q |= (uint64_t)rd(pos + i) << (8 * i);
            pos += len;
            p->quanta = q;
            if (p->quanta < 2) p->error = QB3E_EINV;
        } else if (c0 == 'C' && c1 == 'B') {
            if (len != p->nbands) { p->error = QB3E_EINV; break; }
            pos += 4;
            for (size_t i = 0; i < p->nbands; i++) {
                p->cband[i] = (uint8_t)rd(pos++);
                if (p->cband[i] >= p->nbands) p->error = QB3E_EINV;
            }
            p->saw_cb = true;
        } else if (c0 == 'D' && c1 == 'T') {
            pos += 2;
            if (pos > n) pos = n;
            if (p->s_size <= pos) { p->error = QB3E_EINV; break; }
            p->s_in += pos; p->s_size -= pos; p->stage = 2;
        } else if (c0 == 'S' && c1 == 'C') {
            if (len != 8) { p->error = QB3E_EINV; break; }
            if ((int)p->mode < (int)QB3M_BASE_H || p->mode == QB3M_STORED) { p->error = QB3E_EINV; break; }
            pos += 4;
            uint64_t o = 0;
            for (unsigned i = 0; i < 8; i++) o |= (uint64_t)rd(pos + i) << (8 * i);
            pos += 8;
            p->order = o;
            if (!valid_curve(o)) { p->error = QB3E_EINV; break; }
        } else {
            // the reference skips an ignorable (lower case) chunk by `len` bytes from the chunk start
            // (QB3decode.cpp:254-255); a zero length would never terminate there, treat it as an error
            if (c0 == 'i' && c1 == 'x' && len >= IX_HEAD && rd(pos + 4) >= 1 && rd(pos + 4) <= 3 && p->mode != QB3M_STORED) {
                // this library's restart table (include/qb3x.h): a run of such chunks, all but the last of the same
                // size, each followed by a 4-byte pad chunk (version 2).  Remember where it is, check it later.
                const size_t tsz = szof(p->type);
                const uint32_t blocks = rd(pos + 8) | (rd(pos + 9) << 8) | (rd(pos + 10) << 16) | (rd(pos + 11) << 24);
                const bool bl = (rd(pos + 5) & 2) != 0;     // entries end with their blocks' bit lengths
                const bool cfe = (rd(pos + 5) & 1) != 0;     // entries carry the common factors
                const uint32_t E = (uint32_t)(6 + p->nbands * (1 + tsz * (cfe ? 2 : 1))) + (bl && blocks <= 4096 ? ix_bl_bytes((uint32_t)tsz, (uint32_t)p->nbands, blocks, cfe) : 0);
                const size_t at = (size_t)(p->s_in - p->s_start) + pos;
                const unsigned ver = rd(pos + 4);
                const bool v2 = ver >= 2;
                if ((len - IX_HEAD) % E || pos + len > n) p->ix_bad = true;
                else if (!p->ix_K) {        // the first chunk
                    p->ix_off = at; p->ix_E = E; p->ix_blocks = blocks; p->ix_pads = v2; p->ix_bl = bl; p->ix_ver = ver;
                    p->ix_per_chunk = p->ix_K = (len - IX_HEAD) / E;
                    // A regular table -- every chunk but the last full, a pad behind each, "DT" behind the last -- is stepped over in
                    // one go when "DT" stands where such a table ends: the heads in between are then checked on the device
                    // (ix_check_kernel) and need not be on the host at all (a 16384 x 16384 raster's level 2 table is 24 MB)
                    const uint64_t nblk = (uint64_t)((p->xsize + 3) / 4) * ((p->ysize + 3) / 4);
                    const uint64_t Kexp = blocks ? (nblk + blocks - 1) / blocks : 0;
                    // -- only then: with the whole container on the host the chunks are walked one by one, as the reference's
                    // parser walks them (garbage between the first chunk and "DT" is an error, not a table)
                    const bool all_here = p->hdr_avail >= 11 + n;
                    if (!all_here && v2 && p->ix_per_chunk && Kexp > p->ix_per_chunk && Kexp < 0xffffffffull) {
                        const uint64_t nch = (Kexp + p->ix_per_chunk - 1) / p->ix_per_chunk;
                        const uint64_t total = nch * (IX_HEAD + IX_PAD) + Kexp * E;
                        if (pos + total + 2 < n) {
                            if (have(pos + total) && have(pos + total + 1)) {
                                if (rd(pos + total) == 'D' && rd(pos + total + 1) == 'T') {
                                    p->ix_K = (uint32_t)Kexp; p->ix_heads_unchecked = true;
                                    pos += total;
                                    continue;
                                }
                            } else p->ix_need_off = 11 + pos + total;
                        }
                    }
                } else {                    // a further one: in place, same shape, and only the last may be short
                    const size_t full = IX_HEAD + (size_t)p->ix_per_chunk * E + (p->ix_pads ? IX_PAD : 0);
                    const uint32_t here = (len - IX_HEAD) / E;
                    if (!v2 || !p->ix_pads || ver != p->ix_ver || E != p->ix_E || blocks != p->ix_blocks || bl != p->ix_bl || p->ix_K % p->ix_per_chunk ||
                        at != p->ix_off + (p->ix_K / p->ix_per_chunk) * full || here > p->ix_per_chunk) p->ix_bad = true;
                    else p->ix_K += here;
                }
            }
            if ((c0 & 0x20) && len) pos += len;
            else p->error = QB3E_UNKN;
        }
        if (pos > n) pos = n;
    } while (p->stage != 2 && QB3E_OK == p->error && pos < n);
    if (QB3E_OK == p->error && 2 != p->stage) p->error = QB3E_EINV;
    if (short_copy && QB3E_OK == p->error) p->error = QB3E_EINV;               // qb3x_read_start: the head copy is too short
    p->hdr_short = short_copy;
    if (p->ix_bad) p->ix_K = 0;
    return QB3E_OK == p->error;
}

// qb3_read_start + qb3_read_info for a container in DEVICE memory: the handle keeps its own host copy of the container's
// first bytes (up to 512), and when a restart table pushes the "DT" mark beyond them, of the few bytes where a regular
// table ends -- two small copies instead of the whole table (24 MB for a 16384 x 16384 x 3 raster at level 2), whose
// chunk heads and checks the device verifies before the table is used (ix_check_kernel).  A table that is not regular
// is read whole.  Returns a handle in the state qb3_read_info leaves, or NULL.
static decsp read_start_device_body(const void *d_container, size_t nbytes, size_t *image_size, void *stream);
QB3_API decsp qb3x_read_start_device(const void *d_container, size_t nbytes, size_t *image_size, void *stream) {
    return abi_guard<decsp>(nullptr, [&] { return read_start_device_body(d_container, nbytes, image_size, stream); });
}
static decsp read_start_device_body(const void *d_container, size_t nbytes, size_t *image_size, void *stream) {
    if (!d_container || nbytes < 15 || !image_size || !device_ok()) return nullptr;
    hipStream_t st = (hipStream_t)stream;
    auto fetch = [&](std::vector<uint8_t> &dst, size_t off, size_t n) -> bool {
        dst.resize(n);
        return hipMemcpyAsync(dst.data(), (const uint8_t *)d_container + off, n, hipMemcpyDeviceToHost, st) == hipSuccess &&
               hipStreamSynchronize(st) == hipSuccess;
    };
    std::vector<uint8_t> head, win;
    size_t win_off = 0;
    if (!fetch(head, 0, std::min(nbytes, (size_t)512))) return nullptr;
    for (int turn = 0; turn < 3; turn++) {
        decs *p = read_start_impl(head.data(), head.size(), nbytes, image_size);
        if (!p) return nullptr;
        p->own_head.swap(head);                         // (the vector's buffer stays where it is: s_start stays valid)
        p->win2 = win; p->win2_off = win_off;
        if (qb3_read_info(p)) return p;
        const size_t need = p->ix_need_off;
        head.swap(p->own_head);
        const bool was_short = p->hdr_short;
        if (getenv("QB3_DEBUG_RS")) fprintf(stderr, "read_start_device turn %d: need %zu short %d err %d ix_off %zu K %u E %u per %u ver %u\n", turn, need, (int)was_short, p->error, p->ix_off, p->ix_K, p->ix_E, p->ix_per_chunk, p->ix_ver);
        qb3_destroy_decoder(p);
        if (!was_short) return nullptr;
        if (turn == 0 && need && need + 2 <= nbytes) {  // a regular table: the mark behind it (four bytes: the chunk loop reads a length field behind every tag)
            win_off = need;
            if (!fetch(win, need, std::min<size_t>(4, nbytes - need))) return nullptr;
        } else if (turn <= 1) {                         // something else: the whole head, as far as a table can reach
            const size_t bound = std::min(nbytes, qb3x_header_size_bound(head.data(), head.size()));
            if (bound <= head.size()) return nullptr;
            win.clear(); win_off = 0;
            if (!fetch(head, 0, bound)) return nullptr;
        } else return nullptr;
    }
    return nullptr;
}

QB3_API size_t qb3x_decoder_table_entries(const decsp p) { return (p && p->stage == 2) ? p->ix_K : 0; }

QB3_API size_t qb3x_decoder_index_size(const decsp p) {
    if (!p || p->stage != 2 || p->xsize < 4 || p->ysize < 4) return 0;
    Geometry g = make_geometry(p->xsize, p->ysize, p->nbands, p->type, 0, p->order, p->mode, nullptr, p->cband);
    return index_bytes(g);
}

#undef HIPOK
#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(#x, (int)e_); p->error = QB3E_LIBERR; return 0; } } while (0)

// A plain 8-bit stream (no index, no restart table) is walked through a table in device memory (qb3_dev.h): make sure
// the decoder holds one -- the whole call in one round, or walk_table_cap() and several rounds.  False: out of memory.
static bool walk_table_ready(decsp p, const Geometry &g, const DecPlan &plan, uint32_t ntiles, uint64_t max_bits) {
    if (!walk_table_applies(g, plan)) return true;
    size_t want = walk_memory_bytes(g, ntiles, max_bits);
    const size_t least = walk_table_min_bytes(ntiles, g.tsz);
    const size_t cap = walk_table_cap();
    if (want > cap) want = cap > least ? cap : least;
    return p->d_tab.cap >= want || p->d_tab.ensure(want);
}

// Decode the block stream at d_stream (+ byte offset off inside a 4-byte aligned device buffer) into d_img.
static bool decode_blocks_device(decsp p, const Geometry &g, const uint8_t *d_buf, size_t off, size_t nbytes,
                                 void *d_img, const void *d_index, hipStream_t st, const IxTable &ix = IxTable()) {
    DecPlan plan = plan_decode(g);
    if (!p->d_ws.ensure(plan.ws_bytes)) return false;
    uint32_t *d_status = nullptr;
    const uint32_t *in32 = (const uint32_t *)(d_buf + (off & ~(size_t)3));
    if (!d_index && !ix.base && !walk_table_ready(p, g, plan, 1, (uint64_t)nbytes * 8)) return false;
    uint32_t status = 0;
    IxTable table = ix;
    // (32/64-bit plain streams: the table of a band of sixteen rungs; a stream that leaves the band goes to the one-lane parser)
    bool walk_tab_ok = true, dropped_table = false;
    const uint32_t wide_band = 16;
    for (int turn = 0; turn < 3; turn++) {
        for (int full = 0; full < 2; full++) {      // (second turn: a 16-bit segment outgrew the staging sized for the stream's average)
            if (launch_decode(g, plan, in32, (uint32_t)(8 * (off & 3)), (uint64_t)nbytes * 8, d_img, d_index, p->d_ws.p, &d_status, st, TileBatch(), nullptr, table,
                              walk_tab_ok ? p->d_tab.p : nullptr, walk_tab_ok ? p->d_tab.cap : 0, full != 0, wide_band))
                return false;
            const hipError_t e = fetch_small(&status, d_status, 4, st);
            if (e != hipSuccess) { set_error("decode kernels", (int)e); return false; }
            { static const bool dbg = getenv("QB3_DEBUG_DEC") != nullptr; if (dbg) fprintf(stderr, "decode turn %d full %d: status %u (table %d)\n", turn, full, status, table.base != nullptr); }
            if (!(status & 16)) break;
        }
        // The container's restart table is a convenience the format does not protect: when its check fails (bit 5) or the
        // decode that relied on it does, the stream is decoded again WITHOUT it -- the plain walk, what the reference does
        // with such a container -- so a damaged table costs time, never pixels.
        if (!(status & (27 | 32)) || d_index) break;
        if (table.base) {
            table = IxTable();
            dropped_table = true;
            if (!walk_table_ready(p, g, plan, 1, (uint64_t)nbytes * 8)) return false;
        } else if (walk_tab_ok && p->d_tab.p && walk_table_applies(g, plan)) walk_tab_ok = false;      // (the last rung of the ladder, for every width and mode: the one-lane
            // parser, whose reader behaves like the reference's on a stream that ends early -- zeros behind the end, bitstream.h:36 -- where the table walks stop)
        else break;
    }
    prof_collect();
    p->last_status = status | (dropped_table ? 32u : 0u);        // (bit 5: the container's table was not used in the end, whatever made it so)
    // bit 0: corrupt unit, bit 1: more than 7 unused bits at the end (reference QB3decode.h:411,569,740).
    // bit 2 (ran past the end) is not an error in the reference, whose reader clamps (bitstream.h:36).
    // bit 3: the index handed in does not describe this stream (a segment longer than any valid one).
    if (status & 27) { set_error("decode: corrupt or over-long stream", 0); p->error = QB3E_ERR; return false; }
    return true;
}


// ---------------------------------------------------------------- qb3_read_data, pipelined
// A self-indexed container (level 2 table: an entry per segment with its blocks' fields) decodes segment by segment with
// nothing but its table, so the call can be cut into STRIPS of block rows: while the stream of strip k + 1 goes up the link,
// strip k is decoded and the rows of strip k - 1 come down -- three streams, two rings of pinned slices, the host copies
// (caller's memory <-> pinned) by the pool of copy threads.  The link moves 48 GB/s each way at once (tools/pcie_probe.cpp),
// so the call is bound by its larger direction -- the raster -- instead of the sum of both.
// Returns the decoded size; 0 with p->error == QB3E_OK: not taken or not trusted (a table that fails its check, a segment
// that does not decode) -- the caller goes on with the one-after-the-other path, which has the fallback ladder;
// 0 with p->error set: a HIP failure.
static size_t decode_pipelined(decsp p, const Geometry &g, const IxTable &ix_host, void *host_dst, size_t nbytes, size_t total, size_t line) {
    using qb3host::CopyPool;
    constexpr size_t SLICE = Stager::SLICE, NSLOT = Stager::NSLOT;
    static const bool dbg = getenv("QB3_DEBUG_PIPE") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto ms_since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    const DecPlan plan = plan_decode(g);
    IxTable ixt = ix_host;
    const size_t tab_bytes = ix_total_bytes(ixt) + 2;
    if (!p->d_ix.ensure(tab_bytes + 16)) return 0;
    ixt.base = (uint8_t *)p->d_ix.p;
    if (!decode_strips_ok(g, plan, ixt)) return 0;
    // strips of block rows, about 48 MB of raster each, the last one at least two block rows (a shifted last row overlaps the one before)
    const uint32_t nby = g.nby, nbx = g.nbx, NB = g.seg_blocks;
    uint32_t BR = (uint32_t)std::max<size_t>(2, ((size_t)48 << 20) / (4 * line));
    uint32_t nstrips = (nby + BR - 1) / BR;
    if (nstrips > 1 && nby - (nstrips - 1) * BR < 2) nstrips--;
    if (nstrips < 3) return 0;
    if (!p->pipe.init() || !p->pipe.events(2 * (size_t)nstrips) || !p->stager.init() || !p->stager2.init()) return 0;
    if (!p->d_in.ensure(nbytes + 8) || !p->d_img.ensure(total) || !p->d_ws.ensure(plan.ws_bytes)) return 0;
    // where the stream must have arrived for a strip to be decoded: the position of the segment behind its last one, from
    // the table in the caller's memory (untrusted: not monotonous or behind the stream's end means "not this way")
    const uint8_t *tab_host = p->s_start + p->ix_off;
    auto entry_pos = [&](uint64_t k) -> uint64_t {
        const uint64_t c = k / ixt.per_chunk, j = k - c * ixt.per_chunk;
        const uint8_t *e = tab_host + c * (IX_HEAD + (ixt.pads ? IX_PAD : 0) + (uint64_t)ixt.per_chunk * ixt.entry_bytes) + IX_HEAD + j * ixt.entry_bytes;
        uint64_t v = 0;
        for (int i = 0; i < 6; i++) v |= (uint64_t)e[i] << (8 * i);
        return v;
    };
    struct Strip { uint64_t seg0, nseg; size_t need, row0, row1; };
    std::vector<Strip> strips(nstrips);
    size_t prev_need = 0;
    for (uint32_t s = 0; s < nstrips; s++) {
        const uint32_t br0 = s * BR, br1 = s + 1 == nstrips ? nby : (s + 1) * BR;
        const uint64_t b0 = (uint64_t)br0 * nbx, b1 = (uint64_t)br1 * nbx;
        Strip &t = strips[s];
        t.seg0 = b0 / NB;
        const uint64_t seg1 = std::min<uint64_t>(g.nseg, (b1 + NB - 1) / NB);
        t.nseg = seg1 - t.seg0;
        const uint64_t pos = seg1 < g.nseg ? entry_pos(seg1) : (uint64_t)nbytes * 8;
        if (pos > (uint64_t)nbytes * 8) return 0;
        t.need = std::min(nbytes, (size_t)(pos / 8) + 16);
        if (t.need < prev_need) return 0;
        prev_need = t.need;
        t.row0 = (size_t)4 * br0; t.row1 = s + 1 == nstrips ? g.h : (size_t)4 * br1;
    }
    strips[nstrips - 1].need = nbytes;
    // the slices: up = the table, then the stream; down = every strip's rows
    struct Slice { uint8_t *dev; uint8_t *host; size_t n; uint32_t strip; };
    std::vector<Slice> up, dn;
    for (size_t off = 0; off < tab_bytes; off += SLICE) up.push_back({(uint8_t *)p->d_ix.p + off, const_cast<uint8_t *>(tab_host) + off, std::min(SLICE, tab_bytes - off), 0});
    const size_t n_tab = up.size();
    for (size_t off = 0; off < nbytes; off += SLICE) up.push_back({(uint8_t *)p->d_in.p + off, p->s_in + off, std::min(SLICE, nbytes - off), 0});
    for (uint32_t s = 0; s < nstrips; s++) {
        const size_t a = strips[s].row0 * line, b = strips[s].row1 * line;
        for (size_t off = a; off < b; off += SLICE) dn.push_back({(uint8_t *)p->d_img.p + off, (uint8_t *)host_dst + off, std::min(SLICE, b - off), s});
    }
    CopyPool &pool = CopyPool::get();
    CopyPool::Batch b_up[NSLOT], b_dn[NSLOT];
    Stager &r1 = p->stager, &r2 = p->stager2;
    hipStream_t sU = p->pipe.up, sK = p->pipe.k, sD = p->pipe.dn;
    const double t_setup = ms_since();
    double t_up_done = 0, t_first_dn = 0;
    size_t up_started = 0, up_enq = 0, stream_enq = 0, dn_avail = 0, dn_issued = 0, dn_copy = 0, dn_freed = 0;
    uint32_t next_strip = 0;
    int last_strip_waited = -1;
    uint32_t *d_status = nullptr;
    hipError_t e = hipSuccess;
    bool launch_failed = false;
    auto ok = [&] { return e == hipSuccess && !launch_failed; };
    while (ok() && (up_enq < up.size() || dn_freed < dn.size())) {
        bool progress = false;
        // ---- up: the host copy of a slice into its pinned slot (two in flight), then its DMA
        if (up_started < up.size() && up_started - up_enq < 2 && (up_started < NSLOT || hipEventQuery(r1.ev(up_started)) == hipSuccess)) {
            pool.submit(r1.slot(up_started), up[up_started].host, up[up_started].n, b_up[up_started % NSLOT]);
            up_started++; progress = true;
        }
        if (up_enq < up_started && pool.done(b_up[up_enq % NSLOT])) {
            e = hipMemcpyAsync(up[up_enq].dev, r1.slot(up_enq), up[up_enq].n, hipMemcpyHostToDevice, sU);
            if (e == hipSuccess) e = hipEventRecord(r1.ev(up_enq), sU);
            if (up_enq >= n_tab) stream_enq += up[up_enq].n;
            up_enq++; progress = true;
            if (dbg && up_enq == up.size()) t_up_done = ms_since();
            // strips whose stream is on its way: their kernel waits for it on the kernel stream
            while (ok() && up_enq >= n_tab && next_strip < nstrips && strips[next_strip].need <= stream_enq) {
                hipEvent_t ev_u = p->pipe.ev[2 * next_strip], ev_k = p->pipe.ev[2 * next_strip + 1];
                e = hipEventRecord(ev_u, sU);
                if (e == hipSuccess) e = hipStreamWaitEvent(sK, ev_u, 0);
                if (e != hipSuccess) break;
                const DecStrip st = { strips[next_strip].seg0, strips[next_strip].nseg, next_strip == 0 };
                if (launch_decode(g, plan, (const uint32_t *)p->d_in.p, 0, (uint64_t)nbytes * 8, p->d_img.p, nullptr, p->d_ws.p, &d_status, sK, TileBatch(), nullptr, ixt,
                                  nullptr, 0, false, 16, &st)) { launch_failed = true; break; }
                e = hipEventRecord(ev_k, sK);
                while (dn_avail < dn.size() && dn[dn_avail].strip == next_strip) dn_avail++;
                next_strip++;
            }
        }
        // ---- down: the DMA of a slice into its pinned slot once its strip's kernel is done, then the host copy out of it
        if (ok() && dn_issued < dn_avail && dn_issued - dn_freed < NSLOT) {
            if ((int)dn[dn_issued].strip != last_strip_waited) {
                e = hipStreamWaitEvent(sD, p->pipe.ev[2 * dn[dn_issued].strip + 1], 0);
                last_strip_waited = (int)dn[dn_issued].strip;
            }
            if (e == hipSuccess) e = hipMemcpyAsync(r2.slot(dn_issued), dn[dn_issued].dev, dn[dn_issued].n, hipMemcpyDeviceToHost, sD);
            if (e == hipSuccess) e = hipEventRecord(r2.ev(dn_issued), sD);
            dn_issued++; progress = true;
        }
        if (ok() && dn_copy < dn_issued) {
            const hipError_t q = hipEventQuery(r2.ev(dn_copy));
            if (q == hipSuccess) { if (dbg && !dn_copy) t_first_dn = ms_since(); pool.submit(dn[dn_copy].host, r2.slot(dn_copy), dn[dn_copy].n, b_dn[dn_copy % NSLOT]); dn_copy++; progress = true; }
            else if (q != hipErrorNotReady) e = q;
        }
        while (dn_freed < dn_copy && pool.done(b_dn[dn_freed % NSLOT])) { dn_freed++; progress = true; }
        if (!progress && !pool.help_one()) std::this_thread::yield();
    }
    (void)hipGetLastError();                                // (hipErrorNotReady of the queries is not an error)
    for (size_t i = 0; i < NSLOT; i++) { pool.wait(b_up[i]); pool.wait(b_dn[i]); }
    p->pipe.sync();
    if (dbg) fprintf(stderr, "decode_pipelined: setup %.2f ms, last upload enqueued %.2f, first slice down %.2f, done %.2f (%u strips, %zu + %zu slices)\n", t_setup, t_up_done, t_first_dn, ms_since(), nstrips, up.size(), dn.size());
    if (!ok()) { if (!launch_failed) set_error("pipelined decode", (int)e); p->error = QB3E_LIBERR; return 0; }
    uint32_t status = 0;
    e = hipMemcpy(&status, d_status, 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { set_error("pipelined decode: status", (int)e); p->error = QB3E_LIBERR; return 0; }
    prof_collect();
    p->last_status = status;
    if (status) return 0;                                   // not trusted: the caller decodes again, one step after the other
    return total;
}

static size_t decode_common(decsp p, void *host_dst, const void *d_src, void *d_dst, const void *d_index, hipStream_t st) {
    if (p->stage != 2 || p->error != QB3E_OK || p->s_in == nullptr || p->s_size == 0) {
        if (p->error == QB3E_OK) p->error = QB3E_EINV;
        return 0;
    }
    const bool on_host = host_dst != nullptr;
    const size_t tsz = szof(p->type), line = p->xsize * p->nbands * tsz, total = qb3_decoded_size(p);
    const size_t data_off = (size_t)(p->s_in - p->s_start);
    const size_t dst_stride = (p->stride ? p->stride : p->xsize * p->nbands) * tsz;
    // qb3x_read_start / qb3x_read_start_device handles hold a copy of the container's HEAD (hdr_avail bytes at s_start) while
    // s_size spans the whole container: the host-pointer call would read the stream, and the table, past that copy
    if (on_host && p->hdr_avail < data_off + p->s_size) { p->error = QB3E_EINV; return 0; }
    if (p->mode == QB3M_STORED) {           // reference QB3decode.cpp:356-375
        if (p->s_size != total) { p->error = QB3E_EINV; return 0; }
        if (on_host) {
            if (!p->stride) memcpy(host_dst, p->s_in, total);
            else for (size_t y = 0; y < p->ysize; y++) memcpy((uint8_t *)host_dst + y * dst_stride, p->s_in + y * line, line);   // stride in values, as everywhere (QB3.h:146-148)
        } else {
            HIPOK(hipMemcpy2DAsync(d_dst, dst_stride, (const uint8_t *)d_src + data_off, line, line, p->ysize, hipMemcpyDeviceToDevice, st));
            HIPOK(hipStreamSynchronize(st));
        }
        return total;
    }
    if (p->xsize * p->ysize < 16) { p->error = QB3E_EINV; return 0; }
    if (!device_ok()) { p->error = QB3E_LIBERR; return 0; }

    // a large self-indexed container in host memory: upload, decode and download strip by strip, all three at once
    static const bool no_pipeline = [] { const char *e = getenv("QB3_NO_PIPELINE"); return e && e[0] && e[0] != '0'; }();
    if (on_host && !is_rle_mode(p->mode) && p->ix_K && p->ix_bl && p->quanta <= 1 && p->xsize >= 4 && p->ysize >= 4 && dst_stride == line &&
        total >= ((size_t)64 << 20) && !no_pipeline) {
        uint8_t cb[QB3_MAXBANDS];
        for (size_t c = 0; c < QB3_MAXBANDS; c++) cb[c] = p->cband[c];
        if (!p->saw_cb && !(p->compat & QB3X_REF_CBAND0)) for (size_t c = 0; c < p->nbands; c++) cb[c] = (uint8_t)c;
        const Geometry gp = make_geometry(p->xsize, p->ysize, p->nbands, p->type, 0, p->order, p->mode, nullptr, cb);
        IxTable ixh;
        ixh.K = p->ix_K; ixh.blocks = p->ix_blocks; ixh.entry_bytes = p->ix_E; ixh.per_chunk = p->ix_per_chunk; ixh.pads = p->ix_pads; ixh.block_lens = p->ix_bl;
        ixh.version = p->ix_ver; ixh.check_heads = p->ix_heads_unchecked;
        const size_t r = decode_pipelined(p, gp, ixh, host_dst, p->s_size, total, line);
        if (r) return r;
        if (p->error != QB3E_OK) return 0;
    }
    // locate the block stream on the device
    const uint8_t *dev_buf = nullptr;
    size_t off = 0, nbytes = p->s_size;
    const bool rle = is_rle_mode(p->mode);
    if (rle) {
        // the RLE0 expansion on the device (k_rle0.hip; reference QB3decode.cpp:396-413): size first, then the bytes
        const uint8_t *packed = (const uint8_t *)d_src + data_off;
        const size_t wsb = (rle0_ws_bytes(p->s_size) + 15) & ~(size_t)15;
        if (!p->d_rle.ensure(wsb + (on_host ? p->s_size : 0))) { p->error = QB3E_LIBERR; return 0; }
        if (on_host) {
            uint8_t *up = (uint8_t *)p->d_rle.p + wsb;
            if (!upload(p->stager, up, p->s_in, p->s_size, st)) { p->error = QB3E_LIBERR; return 0; }
            packed = up;
        }
        uint64_t sz = 0;
        if (rle0_device_size(packed, p->s_size, p->d_rle.p, true, &sz, st)) { p->error = QB3E_LIBERR; return 0; }
        if (sz > total) { p->error = QB3E_ERR; return 0; }
        if (!p->d_in.ensure((size_t)sz + 8)) { p->error = QB3E_LIBERR; return 0; }
        if (rle0_device_write(packed, p->s_size, p->d_rle.p, true, p->d_in.p, st)) { p->error = QB3E_LIBERR; return 0; }
        HIPOK(hipMemsetAsync((uint8_t *)p->d_in.p + sz, 0, 8, st));       // (a stream that ends early reads as zeros behind its end, not as what the buffer held before)
        nbytes = (size_t)sz;
        dev_buf = (const uint8_t *)p->d_in.p; off = 0;
    } else if (on_host) {
        if (!p->d_in.ensure(nbytes + 8)) { p->error = QB3E_LIBERR; return 0; }
        if (!upload(p->stager, p->d_in.p, p->s_in, nbytes, st)) { p->error = QB3E_LIBERR; return 0; }
        HIPOK(hipMemsetAsync((uint8_t *)p->d_in.p + nbytes, 0, 8, st));    // (a stream that ends early reads as zeros behind its end, not as what the buffer held before)
        dev_buf = (const uint8_t *)p->d_in.p; off = 0;
    } else { dev_buf = (const uint8_t *)d_src; off = data_off; }

    // geometry, narrow images decode into their stand-in shape (reference QB3decode.cpp:321-353)
    size_t w = p->xsize, h = p->ysize;
    const bool narrow = w < 4 || h < 4;
    if (narrow) {
        const size_t ngroups = (w * h + 15) / 16;
        if (p->xsize < 4) { w = 4; h = ngroups * 4; } else { w = ngroups * 4; h = 4; }
    }
    uint8_t cband[QB3_MAXBANDS];
    for (size_t c = 0; c < QB3_MAXBANDS; c++) cband[c] = p->cband[c];
    // no CB chunk means identity; the reference leaves the map zero filled instead
    if (!p->saw_cb && !(p->compat & QB3X_REF_CBAND0)) for (size_t c = 0; c < p->nbands; c++) cband[c] = (uint8_t)c;
    const bool direct = !on_host && !narrow;        // decode straight into the caller's device buffer
    Geometry g = make_geometry(w, h, p->nbands, p->type, direct ? p->stride : 0, p->order, p->mode, nullptr, cband);
    void *img_dev = d_dst;
    if (!direct) {
        if (!p->d_img.ensure((size_t)g.w * g.h * g.bands * tsz)) { p->error = QB3E_LIBERR; return 0; }
        img_dev = p->d_img.p;
    }
    // a restart table inside the container stands in for a missing index (also under RLE0: the table describes the block stream,
    // which is what the expansion above has just made)
    IxTable ixt;
    if (!d_index && p->ix_K && !narrow) {
        ixt.K = p->ix_K; ixt.blocks = p->ix_blocks; ixt.entry_bytes = p->ix_E; ixt.per_chunk = p->ix_per_chunk; ixt.pads = p->ix_pads; ixt.block_lens = p->ix_bl;
        ixt.version = p->ix_ver; ixt.check_heads = p->ix_heads_unchecked;
        if (on_host) {
            const size_t bytes = ix_total_bytes(ixt) + 2;        // (+2: the "DT" behind the last chunk, which the check kernel looks at)
            if (!p->d_ix.ensure(bytes + 16)) { p->error = QB3E_LIBERR; return 0; }       // (+16: the check kernel reads whole sixteen-byte groups)
            HIPOK(hipMemcpyAsync(p->d_ix.p, p->s_start + p->ix_off, bytes, hipMemcpyHostToDevice, st));
            ixt.base = (uint8_t *)p->d_ix.p;
        } else ixt.base = (uint8_t *)d_src + p->ix_off;
    }
    if (!decode_blocks_device(p, g, dev_buf, off, nbytes, img_dev, d_index, st, ixt)) {
        if (p->error == QB3E_OK) p->error = QB3E_LIBERR;
        return 0;
    }
    if (p->quanta > 1 && launch_dequantize(img_dev, g, (int)p->type, p->quanta, st)) { p->error = QB3E_LIBERR; return 0; }

    if (narrow) {
        std::vector<uint8_t> t((size_t)g.w * g.h * g.bands * tsz);
        HIPOK(hipMemcpyAsync(t.data(), img_dev, t.size(), hipMemcpyDeviceToHost, st));
        HIPOK(hipStreamSynchronize(st));
        std::vector<uint8_t> outimg;
        uint8_t *dst = (uint8_t *)host_dst;
        if (!on_host) { outimg.resize(dst_stride * p->ysize); dst = outimg.data(); }
        const size_t pix = p->nbands * tsz;
        const uint8_t *s = t.data();
        if (p->xsize < 4) for (size_t y = 0; y < p->ysize; y++, s += p->xsize * pix) memcpy(dst + y * dst_stride, s, p->xsize * pix);
        else for (size_t x = 0; x < p->xsize; x++) for (size_t y = 0; y < p->ysize; y++, s += pix) memcpy(dst + y * dst_stride + x * pix, s, pix);
        if (!on_host) { HIPOK(hipMemcpyAsync(d_dst, dst, outimg.size(), hipMemcpyHostToDevice, st)); HIPOK(hipStreamSynchronize(st)); }
        return total;
    }
    if (on_host) {
        if (dst_stride == line) { if (!download(p->stager, host_dst, img_dev, total, st)) { p->error = QB3E_LIBERR; return 0; } }
        else {
            HIPOK(hipMemcpy2DAsync(host_dst, dst_stride, img_dev, line, line, p->ysize, hipMemcpyDeviceToHost, st));
            HIPOK(hipStreamSynchronize(st));
        }
    }
    return total;
}

QB3_API size_t qb3_read_data(decsp p, void *dst) {
    if (!p || !dst) return 0;
    return abi_guard<size_t>(0, [&] { return decode_common(p, dst, nullptr, nullptr, nullptr, nullptr); });
}

QB3_API size_t qb3x_decode_device(decsp p, const void *d_src, void *d_dst, const void *d_index, void *stream) {
    if (!p || !d_src || !d_dst || ((uintptr_t)d_src & 3)) { if (p) p->error = QB3E_EINV; return 0; }
    return abi_guard<size_t>(0, [&] { return decode_common(p, nullptr, d_src, d_dst, d_index, (hipStream_t)stream); });
}

// One tile through its own header: a host copy of its head is parsed into a handle of its own (a batch may hold
// containers of another kind than tile 0's: raw-stored tiles next to coded ones, QB3encode.cpp:571-573)
static bool decode_tile_alone(decsp ref, const uint8_t *d_tile, size_t size, void *d_out, const void *d_index, hipStream_t st) {
    size_t dims[3];
    decsp q = qb3x_read_start_device(d_tile, size, dims, st);
    if (!q) return false;
    bool ok = dims[0] == ref->xsize && dims[1] == ref->ysize && dims[2] == ref->nbands && q->type == ref->type;
    if (ok) {
        q->stride = ref->stride; q->compat = ref->compat;
        ok = 0 != qb3x_decode_device(q, d_tile, d_out, d_index, st);
    }
    qb3_destroy_decoder(q);
    return ok;
}

static size_t decode_tiles_body(decsp p, const void *d_src, size_t n, size_t src_pitch, const size_t *sizes,
                                void *d_dst, size_t dst_pitch, const void *d_index, void *stream);
QB3_API size_t qb3x_decode_tiles(decsp p, const void *d_src, size_t n, size_t src_pitch, const size_t *sizes,
                                 void *d_dst, size_t dst_pitch, const void *d_index, void *stream) {
    return abi_guard<size_t>(0, [&] { return decode_tiles_body(p, d_src, n, src_pitch, sizes, d_dst, dst_pitch, d_index, stream); });
}
static size_t decode_tiles_body(decsp p, const void *d_src, size_t n, size_t src_pitch, const size_t *sizes,
                                void *d_dst, size_t dst_pitch, const void *d_index, void *stream) {
    if (!p || !d_src || !d_dst || !sizes || (src_pitch & 3) || ((uintptr_t)d_src & 3)) return 0;
    if (p->stage != 2 || p->error != QB3E_OK) return 0;
    const size_t hdr = (size_t)(p->s_in - p->s_start);
    hipStream_t st = (hipStream_t)stream;
    p->tile_ok.assign(n, 0);
    p->last_status = 0;
    if (!n || !device_ok()) return 0;
    // the mode byte of every tile: tiles of tile 0's kind go through one set of launches, the others one by one
    std::vector<uint8_t> modes(n);
    {
        hipError_t e = hipMemcpy2DAsync(modes.data(), 1, (const uint8_t *)d_src + 10, src_pitch, 1, n, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_error("decode tiles: mode bytes", (int)e); p->error = QB3E_LIBERR; return 0; }
    }
    // the pitch of the caller's index array is the encoder's qb3x_index_size: a function of the image and of the coding
    // mode, which a raw-stored tile 0 does not tell -- take it from the first coded tile
    size_t isz = 0;
    if (d_index && p->xsize >= 4 && p->ysize >= 4) {
        int m = p->mode;
        for (size_t i = 0; i < n && m == QB3M_STORED; i++) m = modes[i];
        if (m != QB3M_STORED && m < (int)QB3M_END)
            isz = index_bytes(make_geometry(p->xsize, p->ysize, p->nbands, p->type, 0, m <= (int)QB3M_CF_RLE ? ZCURVE : p->order, m, nullptr, p->cband));
    }
    const bool batchable = !is_rle_mode(p->mode) && p->mode != QB3M_STORED && p->quanta <= 1 && p->xsize >= 4 && p->ysize >= 4 &&
                           p->xsize * p->ysize >= 16;
    // restart tables (tile 0 has one, parsed by qb3_read_info): usable for the batch when every tile of it has its "ix" tag
    // and its "DT" mark where tile 0 has them (equally shaped tiles written by one encoder do); else the plain walk
    bool use_ix = !d_index && p->ix_K && hdr >= 2;
    std::vector<uint8_t> tags;
    if (use_ix) {
        tags.resize(4 * n);
        hipError_t e = hipMemcpy2DAsync(tags.data(), 4, (const uint8_t *)d_src + p->ix_off, src_pitch, 2, n, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpy2DAsync(tags.data() + 2, 4, (const uint8_t *)d_src + hdr - 2, src_pitch, 2, n, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_error("decode tiles: table tags", (int)e); p->error = QB3E_LIBERR; return 0; }
    }
    auto in_batch = [&](size_t i) { return batchable && modes[i] == (uint8_t)p->mode && sizes[i] > hdr; };
    for (size_t i = 0; i < n && use_ix; i++)
        if (in_batch(i) && !(tags[4 * i] == 'i' && tags[4 * i + 1] == 'x' && tags[4 * i + 2] == 'D' && tags[4 * i + 3] == 'T')) use_ix = false;
    size_t done = 0;
    if (batchable) {
        uint8_t cband[QB3_MAXBANDS];
        for (size_t c = 0; c < QB3_MAXBANDS; c++) cband[c] = p->cband[c];
        if (!p->saw_cb && !(p->compat & QB3X_REF_CBAND0)) for (size_t c = 0; c < p->nbands; c++) cband[c] = (uint8_t)c;
        Geometry g = make_geometry(p->xsize, p->ysize, p->nbands, p->type, p->stride, p->order, p->mode, nullptr, cband);
        const DecPlan plan = plan_decode(g);
        const size_t wsp = (plan.ws_bytes + 255) & ~(size_t)255;
        size_t batch = d_index ? n : (((size_t)8 << 30) / wsp ? ((size_t)8 << 30) / wsp : 1);
        if (batch > n) batch = n;
        if (batch > 65535) batch = 65535;
        if (!p->d_ws.ensure(256 * ((batch + 63) / 64) + (d_index ? 0 : batch * wsp)) || !p->d_in.ensure(8 * batch)) { p->error = QB3E_LIBERR; return 0; }
        std::vector<uint64_t> bits(batch);
        std::vector<uint32_t> status(batch);
        for (size_t first = 0; first < n; first += batch) {
            const size_t cnt = (n - first < batch) ? n - first : batch;
            // a tile of another kind takes part with an empty stream: its lanes find nothing to read, its turn comes below
            for (size_t i = 0; i < cnt; i++) bits[i] = in_batch(first + i) ? (uint64_t)(sizes[first + i] - hdr) * 8 : 0;
            hipError_t e = hipMemcpyAsync(p->d_in.p, bits.data(), 8 * cnt, hipMemcpyHostToDevice, st);
            if (e != hipSuccess) { set_error("decode tiles: upload of stream lengths", (int)e); p->error = QB3E_LIBERR; return done; }
            TileBatch tb;
            tb.n = (uint32_t)cnt; tb.src_pitch = src_pitch; tb.dst_pitch = dst_pitch; tb.idx_pitch = isz;
            for (size_t i = 0; i < cnt; i++) if (bits[i] > tb.max_bits) tb.max_bits = bits[i];
            IxTable ixt;
            if (use_ix) {
                ixt.K = p->ix_K; ixt.blocks = p->ix_blocks; ixt.entry_bytes = p->ix_E; ixt.per_chunk = p->ix_per_chunk; ixt.pads = p->ix_pads; ixt.block_lens = p->ix_bl;
                ixt.version = p->ix_ver; ixt.check_heads = true;      // (only tile 0's heads were read on the host)
                ixt.base = (uint8_t *)d_src + first * src_pitch + p->ix_off;
            }
            if (!d_index && !use_ix && !walk_table_ready(p, g, plan, tb.n, tb.max_bits)) { p->error = QB3E_LIBERR; return done; }
            const uint8_t *src0 = (const uint8_t *)d_src + first * src_pitch;
            uint32_t *d_status = nullptr;
            bool walk_tab_ok = true;
            const uint32_t wide_band = 16;
            for (int turn = 0; turn < 3; turn++) {
                for (int full = 0; full < 2; full++) {      // (second turn: a 16-bit segment outgrew the staging sized for the streams' average)
                    if (launch_decode(g, plan, (const uint32_t *)(src0 + (hdr & ~(size_t)3)), (uint32_t)(8 * (hdr & 3)), 0, (uint8_t *)d_dst + first * dst_pitch,
                                      d_index ? (const uint8_t *)d_index + first * isz : nullptr, p->d_ws.p, &d_status, st, tb, (const uint64_t *)p->d_in.p,
                                      ixt, walk_tab_ok ? p->d_tab.p : nullptr, walk_tab_ok ? p->d_tab.cap : 0, full != 0, wide_band)) { p->error = QB3E_LIBERR; return done; }
                    e = hipMemcpyAsync(status.data(), d_status, 4 * cnt, hipMemcpyDeviceToHost, st);
                    if (e == hipSuccess) e = hipStreamSynchronize(st);
                    if (e != hipSuccess) { set_error("decode kernels (tiles)", (int)e); p->error = QB3E_LIBERR; return done; }
                    bool again = false;
                    for (size_t i = 0; i < cnt; i++) again = again || (status[i] & 16);
                    if (!again) break;
                }
                // a tile whose table fails its check, or whose decode from the table fails: the batch again without the tables
                bool table_trouble = false;
                for (size_t i = 0; i < cnt; i++) table_trouble = table_trouble || (bits[i] && (status[i] & (27 | 32)));
                if (!table_trouble || d_index) break;
                if (ixt.base) {
                    ixt = IxTable();
                    if (!walk_table_ready(p, g, plan, tb.n, tb.max_bits)) { p->error = QB3E_LIBERR; return done; }
                } else if (walk_tab_ok && (g.tsz >= 4 || g.mode == CM_BEST) && p->d_tab.p && walk_table_applies(g, plan)) walk_tab_ok = false;   // a stream left the band of rungs: the one-lane parser
                else break;
            }
            prof_collect();
            for (size_t i = 0; i < cnt; i++) { p->last_status |= status[i]; if (bits[i] && !(status[i] & 27)) { p->tile_ok[first + i] = 1; done++; } }
        }
    }
    for (size_t i = 0; i < n; i++) {
        if (in_batch(i)) continue;
        if (decode_tile_alone(p, (const uint8_t *)d_src + i * src_pitch, sizes[i], (uint8_t *)d_dst + i * dst_pitch,
                              d_index ? (const uint8_t *)d_index + i * isz : nullptr, st)) { p->tile_ok[i] = 1; done++; }
    }
    return done;
}

QB3_API int qb3x_decode_tile_ok(const decsp p, size_t i) { return (p && i < p->tile_ok.size()) ? p->tile_ok[i] : 0; }

// ---------------------------------------------------------------- misc
// returns the device buffers that destroyed handles left in the library's pool to the runtime
QB3_API unsigned qb3x_last_decode_status(const decsp p) { return p ? p->last_status : 0u; }
QB3_API void qb3x_trim(void) { try { dev_pool().trim(); qb3host::ring_trim(); } catch (...) {} }

QB3_API int qb3x_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
QB3_API const char *qb3x_last_error(void) { return last_error(); }
// FNV-1a, 64 bit, over host bytes; chain calls by passing the previous result as `seed` (0 starts a new hash)
QB3_API uint64_t qb3x_fnv1a64(const void *data, size_t n, uint64_t seed) {
    uint64_t h = seed ? seed : 0xcbf29ce484222325ull;
    const uint8_t *b = (const uint8_t *)data;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 0x100000001b3ull;
    return h;
}
// The RLE0 byte pass on device buffers (k_rle0.hip): the coded (decode = 0) or expanded (decode != 0) form of the n bytes
// at d_src.  d_dst == NULL: the size only.  Returns the size, 0 on failure or when it exceeds dst_cap.
QB3_API size_t qb3x_rle0_device(const void *d_src, size_t n, void *d_dst, size_t dst_cap, int decode, void *stream) {
    if (!d_src || !n) return 0;
    DevBuf ws;
    if (!ws.ensure(rle0_ws_bytes(n))) return 0;
    uint64_t total = 0;
    size_t ret = 0;
    if (rle0_device_size(d_src, n, ws.p, decode != 0, &total, stream) == 0) {
        if (!d_dst) ret = (size_t)total;
        else if (total <= dst_cap && rle0_device_write(d_src, n, ws.p, decode != 0, d_dst, stream) == 0 &&
                 hipStreamSynchronize((hipStream_t)stream) == hipSuccess) ret = (size_t)total;
    }
    ws.release();
    return ret;
}
QB3_API void qb3x_profile_enable(int level) { prof_enable(level < 0 ? 0 : level); }
QB3_API void qb3x_profile_reset(void) { prof_reset(); }
QB3_API int qb3x_profile_get(const char *kernel, double *total_ms, uint64_t *count) { return prof_get(kernel, total_ms, count) ? 1 : 0; }
QB3_API int qb3x_profile_names(char *buf, size_t bufsize) { return prof_names(buf, bufsize); }

QB3_API decsp qb3_create_decoder(void *source, size_t source_size, size_t *image_size) {
    decsp p = qb3_read_start(source, source_size, image_size);
    if (p && !qb3_read_info(p)) { qb3_destroy_decoder(p); p = nullptr; }
    return p;
}
QB3_API size_t qb3_decode(decsp p, void *destination) { return qb3_read_data(p, destination); }
