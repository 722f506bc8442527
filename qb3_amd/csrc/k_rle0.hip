// qb3_amd/csrc/k_rle0.hip -- the RLE0 byte pass of the *_RLE modes on the device (reference QB3encode.cpp:271-332,
// QB3decode.cpp:267-307), so that a stream never has to cross PCIe for it
#include "qb3_kernels.h"

namespace qb3dev {

// The reference codes a stream with a greedy, byte-serial loop: a pair of 0xff becomes ff ff ff, a run of 4..258 zero
// bytes becomes ff ff (n - 4) -- unless the byte emitted just before was a literal 0xff --, everything else is copied,
// and no code starts in the last two bytes.  What makes it parallel: a byte that is neither 00 nor ff is ALWAYS copied
// and resets the state, so the loop only has memory inside maximal REGIONS of 00/ff bytes.  A thread that finds the
// start of a region replays the reference loop over it (runs of equal bytes in closed form: full 258-zero codes, then
// the remainder; ff pairs), every other byte costs one byte; chunk sums, a scan, and the same walk again writes.
// Long runs are skipped a 4 KB chunk at a time through a table that says which chunks hold one byte value only.
constexpr uint32_t RLE_CHUNK = 4096, RLE_THREADS = 256, RLE_PER_THREAD = RLE_CHUNK / RLE_THREADS;
constexpr uint32_t RLE_SCAN_GROUP = 4096;                  // chunks per workgroup of rle0_scan_kernel

// sixteen bytes of s from p on as four dwords, the byte before them and the byte behind them, through aligned dword loads
// (p > 0 and p + 17 <= n: every dword read holds at least one byte of s)
struct Rle16 { uint32_t w[4], prev, next; };
__device__ __forceinline__ Rle16 rle_load16(const uint8_t *s, uint64_t p) {
    const uintptr_t A = (uintptr_t)(s + p);
    const uint32_t *d = (const uint32_t *)(A & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(A & 3) * 8;
    uint32_t v[6];
#pragma unroll
    for (int k = 0; k < 5; k++) v[k + 1] = d[k];
    v[0] = sh ? v[1] : d[-1];                               // (the byte before p is in d[0] unless p is dword aligned)
    Rle16 r;
#pragma unroll
    for (int k = 0; k < 4; k++) r.w[k] = sh ? __builtin_amdgcn_alignbit(v[k + 2], v[k + 1], sh) : v[k + 1];
    r.next = (sh ? __builtin_amdgcn_alignbit(0u, v[5], sh) : v[5]) & 0xffu;
    r.prev = sh ? (v[1] >> (sh - 8)) & 0xffu : v[0] >> 24;
    return r;
}
// 0x80 in every byte of v that is 00 or ff
__device__ __forceinline__ uint32_t rle_special4(uint32_t v) {
    const uint32_t a = ((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v, n = ~v, b = ((n & 0x7f7f7f7fu) + 0x7f7f7f7fu) | n;
    return ~(a & b) & 0x80808080u;
}
__device__ __forceinline__ uint32_t rle_ff4(uint32_t v) {
    const uint32_t n = ~v, b = ((n & 0x7f7f7f7fu) + 0x7f7f7f7fu) | n;
    return ~b & 0x80808080u;
}

// which 4 KB chunks hold one byte value only (and starts the size pass's total at n)
__global__ void __launch_bounds__(256) rle0_uniform_kernel(const uint8_t *s, uint64_t n, uint16_t *uniform, uint64_t *total) {
    const uint64_t c0 = (uint64_t)blockIdx.x * RLE_CHUNK;
    __shared__ uint32_t differs;
    if (threadIdx.x == 0) differs = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) total[0] = n;
    __syncthreads();
    const uint8_t v = s[c0];
    bool d = c0 + RLE_CHUNK > n;                            // a short last chunk never counts as uniform
    const uint64_t p = c0 + (uint64_t)threadIdx.x * RLE_PER_THREAD;
    if (!d) {
        if (p > 0 && p + RLE_PER_THREAD + 1 <= n) {
            const Rle16 r = rle_load16(s, p);
            const uint32_t sv = (uint32_t)v * 0x01010101u;
            d = r.w[0] != sv || r.w[1] != sv || r.w[2] != sv || r.w[3] != sv;
        } else
            for (uint32_t i = 0; i < RLE_PER_THREAD; i++) d = d || s[p + i] != v;
    }
    if (d) differs = 1;
    __syncthreads();
    if (threadIdx.x == 0) uniform[blockIdx.x] = differs ? (uint16_t)0x100 : (uint16_t)v;
}

// bytes equal to c from position i on (s[i] == c)
__device__ __forceinline__ uint64_t rle_run_len(const uint8_t *s, uint64_t n, uint64_t i, uint8_t c, const uint16_t *uniform) {
    uint64_t j = i + 1;
    while (j < n) {
        if ((j & (RLE_CHUNK - 1)) == 0) {
            while (j + RLE_CHUNK <= n && uniform[j / RLE_CHUNK] == c) j += RLE_CHUNK;
            if (j >= n) break;
        }
        if (s[j] != c) break;
        j++;
    }
    return j - i;
}
// bytes equal to c ending at position i (s[i] == c), going backwards
__device__ __forceinline__ uint64_t rle_run_len_back(const uint8_t *s, uint64_t i, uint8_t c, const uint16_t *uniform) {
    uint64_t j = i;                                         // first position known to hold c
    while (j > 0) {
        if ((j & (RLE_CHUNK - 1)) == 0) {
            while (j >= RLE_CHUNK && uniform[j / RLE_CHUNK - 1] == c) j -= RLE_CHUNK;
            if (j == 0) break;
        }
        if (s[j - 1] != c) break;
        j--;
    }
    return i - j + 1;
}

struct RleOut {
    uint8_t *out; uint64_t o;
    __device__ __forceinline__ void lit(uint8_t c) { if (out) out[o] = c; o++; }
    __device__ __forceinline__ void fill(uint8_t c, uint64_t k) { if (out) for (uint64_t q = 0; q < k; q++) out[o + q] = c; o += k; }
    __device__ __forceinline__ void toks(uint8_t c, uint64_t k) {       // k codes ff ff c
        if (out) for (uint64_t q = 0; q < k; q++) { out[o + 3 * q] = 0xff; out[o + 3 * q + 1] = 0xff; out[o + 3 * q + 2] = c; }
        o += 3 * k;
    }
};

// the bytes the reference emits for the region of 00/ff bytes that starts at a (reference QB3encode.cpp:283-310)
__device__ uint64_t rle0_region(const uint8_t *s, uint64_t n, uint64_t a, const uint16_t *uniform, uint8_t *out) {
    RleOut E{out, 0};
    const uint64_t lim = n > 2 ? n - 2 : 0;                 // no code starts at or beyond lim
    uint8_t last = 0;                                       // the literal emitted last (a region follows a copied byte that is not ff, or the start)
    uint64_t i = a;
    while (i < n) {
        const uint8_t c = s[i];
        if (c != 0 && c != 0xff) break;
        if (i >= lim) { E.lit(c); i++; continue; }          // the last two bytes are copied
        uint64_t L = rle_run_len(s, n, i, c, uniform);
        if (c == 0xff) {
            uint64_t pairs = L / 2;
            const uint64_t maxp = (lim - i + 1) / 2;        // pair k starts at i + 2k, which must be below lim
            if (pairs > maxp) pairs = maxp;
            E.toks(0xff, pairs);
            i += 2 * pairs; L -= 2 * pairs;
            if (pairs) last = 0;
            if (L && i < lim) { E.lit(0xff); last = 0xff; i++; }      // one ff left: the byte after it differs
        } else if (last == 0xff) { E.lit(0); last = 0; i++; }         // no code right after a literal ff (the run is measured again)
        else if (L >= 4) {
            const uint64_t full = L / 258, r = L % 258;
            E.toks(0xfe, full);
            i += 258 * full;
            if (r >= 4) { E.toks((uint8_t)(r - 4), 1); i += r; }
            else { E.fill(0, r); i += r; }
            last = 0;
        } else { E.fill(0, L); i += L; last = 0; }
    }
    return E.o;
}

__device__ __forceinline__ bool rle_special(uint8_t c) { return c == 0 || c == 0xff; }

// MODE 0: bytes per chunk; MODE 1: write (chunk_off known)
template <int MODE, bool DECODE>
__global__ void __launch_bounds__(256) rle0_pass_kernel(const uint8_t *s, uint64_t n, const uint16_t *uniform, uint32_t *chunk_out, const uint64_t *chunk_off, uint8_t *dst, uint64_t *total, const uint64_t *gsum);

// ---- decoding (reference QB3decode.cpp:267-291): ff ff x is a code wherever it STARTS at a code boundary, and a maximal
// run of ff bytes always starts at one (the byte before it is a copied byte or the count of a zero code).  Of a run of L
// ff bytes every three are a pair code (two ff out); one left over is a copied ff; two left over take the byte after the
// run as the count of a zero code (4 + x zero bytes out).  No code starts in the last two bytes.
__device__ uint64_t derle0_run(const uint8_t *s, uint64_t n, uint64_t a, const uint16_t *uniform, uint8_t *out) {
    RleOut E{out, 0};
    const uint64_t lim = n > 2 ? n - 2 : 0;
    uint64_t i = a, L = rle_run_len(s, n, a, 0xff, uniform);
    while (L) {
        if (i >= lim) { E.fill(0xff, L); break; }
        if (L >= 3) {
            uint64_t k = L / 3;
            const uint64_t maxk = (lim - i + 2) / 3;        // code k starts at i + 3k, which must be below lim
            if (k > maxk) k = maxk;
            E.fill(0xff, 2 * k);
            i += 3 * k; L -= 3 * k;
        } else if (L == 2) { E.fill(0, 4 + (uint64_t)s[i + 2]); break; }       // (i < n - 2: the count byte exists)
        else { E.lit(0xff); break; }
    }
    return E.o;
}

template <int MODE, bool DECODE>
__global__ void __launch_bounds__(256) rle0_pass_kernel(const uint8_t *s, uint64_t n, const uint16_t *uniform, uint32_t *chunk_out, const uint64_t *chunk_off, uint8_t *dst, uint64_t *total, const uint64_t *gsum) {
    __shared__ uint32_t part[4];
    const uint32_t tid = threadIdx.x;
    const uint64_t p0 = (uint64_t)blockIdx.x * RLE_CHUNK + (uint64_t)tid * RLE_PER_THREAD;
    // one pass over the thread's bytes: what each of them contributes.  MODE 1 runs it twice: sizes, then (after the
    // workgroup's scan) the bytes themselves.
    auto walk = [&](uint8_t *out) -> uint64_t {
        uint64_t o = 0;
        for (uint32_t q = 0; q < RLE_PER_THREAD; q++) {
            const uint64_t p = p0 + q;
            if (p >= n) break;
            const uint8_t c = s[p];
            if (!DECODE) {
                if (!rle_special(c)) { if (out) out[o] = c; o++; }
                else if (p == 0 || !rle_special(s[p - 1])) o += rle0_region(s, n, p, uniform, out ? out + o : nullptr);
            } else {
                if (c == 0xff) { if (p == 0 || s[p - 1] != 0xff) o += derle0_run(s, n, p, uniform, out ? out + o : nullptr); }
                else {
                    // the count of a zero code when the ff run that ends just before it leaves two over
                    const bool taken = p > 0 && s[p - 1] == 0xff && rle_run_len_back(s, p - 1, 0xff, uniform) % 3 == 2;
                    if (!taken) { if (out) out[o] = c; o++; }
                }
            }
        }
        return o;
    };
    // Most threads need no walk: a byte that is 00 or ff with no such byte next to it is copied like any other (a region of
    // one byte: no pair, no run of four), so sixteen bytes without two special bytes in a row -- the bytes before and behind
    // them included -- are sixteen bytes out; expanding, sixteen bytes with no ff among them or just before them are.  The
    // stream's first bytes and its end (no code starts in the last two bytes) take the walk.
    bool quick = false;
    if (p0 > 0 && p0 + RLE_PER_THREAD + 3 <= n) {
        const Rle16 r = rle_load16(s, p0);
        if (!DECODE) {
            uint32_t f[5];
#pragma unroll
            for (int k = 0; k < 4; k++) f[k] = rle_special4(r.w[k]);
            f[4] = rle_special4(r.next | 0x01010100u);
            uint32_t adj = rle_special4(r.prev | 0x01010100u) & f[0] & 0x80u;
#pragma unroll
            for (int k = 0; k < 4; k++) adj |= f[k] & __builtin_amdgcn_alignbit(f[k + 1], f[k], 8);
            quick = adj == 0;
        } else {
            // expanding: an ff with no ff next to it is a copied byte too (a run of one: QB3decode.cpp:267-291), and the byte
            // behind it is not a count -- sixteen bytes with no two ff in a row, the bytes around them included, are sixteen out
            uint32_t f[5];
#pragma unroll
            for (int k = 0; k < 4; k++) f[k] = rle_ff4(r.w[k]);
            f[4] = rle_ff4(r.next | 0x01010100u);
            uint32_t adj = rle_ff4(r.prev | 0x01010100u) & 0x80u;       // (an ff just before: the run it ends may make the first byte a count)
#pragma unroll
            for (int k = 0; k < 4; k++) adj |= f[k] & __builtin_amdgcn_alignbit(f[k + 1], f[k], 8);
            quick = adj == 0;
        }
    }
    const uint64_t mine = quick ? RLE_PER_THREAD : walk(nullptr);
    // (a thread's share is below 2^32: the output of one region is at most 3/2 of its bytes + 258 per code)
    // the workgroup's scan: DPP inside the waves, the four wave sums through LDS, one barrier
    const uint32_t inc_w = wave_iscan32((uint32_t)mine);
    if ((tid & 63) == 63) part[tid >> 6] = inc_w;
    __syncthreads();
    uint32_t before = 0, chunk_total = 0;
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) { const uint32_t v = part[i]; if (i < (tid >> 6)) before += v; chunk_total += v; }
    const uint32_t inc = before + inc_w;                    // inclusive prefix of the thread's bytes inside the chunk
    if (MODE == 0) {
        if (tid == 0) {     // total = n + what the chunks' sizes differ from their byte counts by: most chunks add nothing
            chunk_out[blockIdx.x] = chunk_total;
            const uint64_t c0 = (uint64_t)blockIdx.x * RLE_CHUNK, in = n - c0 < RLE_CHUNK ? n - c0 : RLE_CHUNK;
            if (chunk_total != in) atomicAdd((unsigned long long *)total, (unsigned long long)chunk_total - (unsigned long long)in);   // (wraps: two's complement)
        }
        return;
    }
    // The write.  Nearly every thread is "quick": its sixteen bytes go out as they are, only shifted to wherever the bytes
    // before them ended.  A byte at a time that is 435 M one-byte stores for a 16384 x 16384 x 3 raster's stream (1.3 ms, five
    // times what the bytes cost to move), so:
    //  * a wave in which EVERY thread is quick is a shifted copy of its kilobyte: each lane stores the sixteen output bytes
    //    of an ALIGNED group (one 16-byte store, the sources funnel-shifted out of aligned loads), the kilobyte's first
    //    and last few bytes -- the groups it shares with its neighbours -- go byte by byte;
    //  * elsewhere a quick thread stores its head bytes up to the next aligned dword, three or four dwords, its tail bytes.
    uint8_t *base = dst + gsum[blockIdx.x / RLE_SCAN_GROUP] + chunk_off[blockIdx.x];
    // (per WAVE: 64 quick threads are a shifted copy of their kilobyte -- of a stream's waves more than nine in ten)
    if (__all(quick)) {
        const uint32_t wv = tid >> 6, lane = tid & 63;
        const uint64_t w0 = (uint64_t)blockIdx.x * RLE_CHUNK + 1024 * wv;          // the wave's first input byte
        uint8_t *ow = base + before;                                            // ... and where its 1024 bytes go
        const uint32_t head = (uint32_t)((0 - (uintptr_t)ow) & 15);             // bytes in front of the first 16-byte aligned output address
        const uint32_t ngroups = (1024 - head) >> 4;                            // 63 or 64 aligned groups of sixteen
        if (lane < ngroups) {
            const Rle16 r = rle_load16(s, w0 + head + 16 * (uint64_t)lane);     // (w0 > 0 and three bytes at least follow the wave's: quick threads said so)
            *(uint4 *)(ow + head + 16 * (uint64_t)lane) = make_uint4(r.w[0], r.w[1], r.w[2], r.w[3]);
        }
        const uint32_t tail0 = head + 16 * ngroups;                             // bytes behind the last group (fewer than sixteen)
        if (lane < head) ow[lane] = s[w0 + lane];
        if (lane >= 32 && lane - 32 < 1024 - tail0) ow[tail0 + (lane - 32)] = s[w0 + tail0 + (lane - 32)];
        return;
    }
    if (!mine) return;
    uint8_t *o = base + (inc - (uint32_t)mine);
    if (quick) {
        const Rle16 r = rle_load16(s, p0);
        const uint32_t a = (uint32_t)((0 - (uintptr_t)o) & 3);                  // head bytes up to the next aligned dword
        const uint32_t w4[5] = { r.w[0], r.w[1], r.w[2], r.w[3], 0u };
#pragma unroll
        for (uint32_t k = 0; k < 3; k++) if (k < a) o[k] = (uint8_t)(r.w[0] >> (8 * k));
        uint32_t *od = (uint32_t *)(o + a);
#pragma unroll
        for (uint32_t k = 0; k < 3; k++) od[k] = a ? __builtin_amdgcn_alignbit(w4[k + 1], w4[k], 8 * a) : w4[k];
        if (a == 0) od[3] = r.w[3];
        else {
#pragma unroll
            for (uint32_t k = 0; k < 3; k++) if (k < 4 - a) o[a + 12 + k] = (uint8_t)(r.w[3] >> (8 * (a + k)));
        }
        return;
    }
    walk(o);
}

// chunk_off = exclusive prefix of chunk_out (only the write pass needs it); total[0] = the sum again.  A workgroup per 4096
// chunks (four a thread, 16-byte loads) leaves its chunks' offsets inside the group and the group's sum; the workgroup that
// finishes LAST (a counter behind the group sums) scans the group sums and adds them in -- enc_scan_kernel's scheme.
__global__ void __launch_bounds__(1024) rle0_scan_kernel(const uint32_t *chunk_out, uint64_t *chunk_off, uint64_t nchunks, uint64_t *total, uint64_t *gsum) {
    __shared__ uint64_t wsum[16];
    __shared__ uint32_t is_last;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ngroups = gridDim.x;
    const uint64_t i0 = (uint64_t)blockIdx.x * RLE_SCAN_GROUP + 4 * tid;
    uint32_t v[4];
    uint64_t sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = i0 + k < nchunks ? chunk_out[i0 + k] : 0u; sum += v[k]; }
    uint64_t x = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint64_t y = __shfl_up(x, d, 64); if (lane >= (uint32_t)d) x += y; }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint64_t off = x - sum, tot = 0;
    for (uint32_t i = 0; i < 16; i++) { const uint64_t w = wsum[i]; if (i < wave) off += w; tot += w; }
#pragma unroll
    for (int k = 0; k < 4; k++) if (i0 + k < nchunks) { chunk_off[i0 + k] = off; off += v[k]; }
    if (tid == 0) {
        gsum[blockIdx.x] = tot;
        __threadfence();
        is_last = atomicAdd((uint32_t *)&gsum[ngroups + 1], 1u) == ngroups - 1;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    if (tid == 0) {             // a few dozen groups: one thread
        uint64_t run = 0;
        for (uint32_t g = 0; g < ngroups; g++) { const uint64_t t = __hip_atomic_load(&gsum[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); gsum[g] = run; run += t; }
        gsum[ngroups] = run; gsum[ngroups + 1] = 0;         // (the counter is ready for the next call)
        total[0] = run;
    }
}

// ---- host entry points.  ws: rle0_ws_bytes(n) of device memory; the size pass leaves what the write pass needs in it.
size_t rle0_ws_bytes(uint64_t n) {
    const uint64_t nchunks = (n + RLE_CHUNK - 1) / RLE_CHUNK;
    const uint64_t ngroups = (nchunks + RLE_SCAN_GROUP - 1) / RLE_SCAN_GROUP;
    return (size_t)(16 + 8 * (ngroups + 2) + 8 * nchunks + 4 * nchunks + 2 * nchunks + 64);
}
struct RleWs { uint64_t *total, *gsum, *off; uint32_t *out; uint16_t *uniform; uint64_t nchunks, ngroups; };
static RleWs rle_ws(void *ws, uint64_t n) {
    RleWs w;
    w.nchunks = (n + RLE_CHUNK - 1) / RLE_CHUNK;
    uint8_t *p = (uint8_t *)ws;
    w.ngroups = (w.nchunks + RLE_SCAN_GROUP - 1) / RLE_SCAN_GROUP;
    w.total = (uint64_t *)p; p += 16;
    w.gsum = (uint64_t *)p; p += 8 * (w.ngroups + 2);       // per group of chunks: bytes before it; the total; the scan's count of finished workgroups
    w.off = (uint64_t *)p; p += 8 * w.nchunks;
    w.out = (uint32_t *)p; p += 4 * w.nchunks;
    w.uniform = (uint16_t *)p;
    return w;
}
// size of the coded (decode = false) or expanded (decode = true) form of the n bytes at d_src; synchronises the stream
__global__ void __launch_bounds__(256) rle0_no_uniform_kernel(uint16_t *uniform, uint64_t nchunks, uint64_t n, uint64_t *total) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nchunks) uniform[i] = (uint16_t)0x100;
    if (i == 0) total[0] = n;
}
int rle0_device_size(const void *d_src, uint64_t n, void *ws, bool decode, uint64_t *total, void *stream, bool no_uniform_chunk) {
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { *total = 0; return 0; }
    const RleWs w = rle_ws(ws, n);
    const uint8_t *s = (const uint8_t *)d_src;
    {
    ProfScope ps(decode ? "rle0_expand_size" : "rle0_size", st);
    if (no_uniform_chunk) hipLaunchKernelGGL(rle0_no_uniform_kernel, dim3((uint32_t)((w.nchunks + 255) / 256)), dim3(256), 0, st, w.uniform, w.nchunks, n, w.total);
    else hipLaunchKernelGGL(rle0_uniform_kernel, dim3((uint32_t)w.nchunks), dim3(256), 0, st, s, n, w.uniform, w.total);
    if (decode) hipLaunchKernelGGL((rle0_pass_kernel<0, true>), dim3((uint32_t)w.nchunks), dim3(256), 0, st, s, n, w.uniform, w.out, w.off, (uint8_t *)nullptr, w.total, (const uint64_t *)w.gsum);
    else hipLaunchKernelGGL((rle0_pass_kernel<0, false>), dim3((uint32_t)w.nchunks), dim3(256), 0, st, s, n, w.uniform, w.out, w.off, (uint8_t *)nullptr, w.total, (const uint64_t *)w.gsum);
    }
    HIPCHK(hipMemcpyAsync(total, w.total, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return 0;
}
// after rle0_device_size with the same arguments: the bytes, to d_dst (room for the size it returned)
int rle0_device_write(const void *d_src, uint64_t n, void *ws, bool decode, void *d_dst, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return 0;
    const RleWs w = rle_ws(ws, n);
    const uint8_t *s = (const uint8_t *)d_src;
    ProfScope ps(decode ? "rle0_expand" : "rle0_write", st);
    HIPCHK(hipMemsetAsync(w.gsum + w.ngroups + 1, 0, 8, st));                                                 // the scan's counter
    hipLaunchKernelGGL(rle0_scan_kernel, dim3((uint32_t)w.ngroups), dim3(1024), 0, st, w.out, w.off, w.nchunks, w.total + 1, w.gsum);      // (the size pass left the chunk sums)
    if (decode) hipLaunchKernelGGL((rle0_pass_kernel<1, true>), dim3((uint32_t)w.nchunks), dim3(256), 0, st, s, n, w.uniform, w.out, w.off, (uint8_t *)d_dst, w.total + 1, (const uint64_t *)w.gsum);
    else hipLaunchKernelGGL((rle0_pass_kernel<1, false>), dim3((uint32_t)w.nchunks), dim3(256), 0, st, s, n, w.uniform, w.out, w.off, (uint8_t *)d_dst, w.total + 1, (const uint64_t *)w.gsum);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace qb3dev
