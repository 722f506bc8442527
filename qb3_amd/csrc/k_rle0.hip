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
__global__ void __launch_bounds__(256) rle0_pass_kernel(const uint8_t *s, uint64_t n, const uint16_t *uniform, uint32_t *chunk_out, const uint64_t *chunk_off, uint8_t *dst, uint64_t *total);

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
__global__ void __launch_bounds__(256) rle0_pass_kernel(const uint8_t *s, uint64_t n, const uint16_t *uniform, uint32_t *chunk_out, const uint64_t *chunk_off, uint8_t *dst, uint64_t *total) {
    __shared__ uint32_t part[256];
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
        } else
            quick = (rle_ff4(r.w[0]) | rle_ff4(r.w[1]) | rle_ff4(r.w[2]) | rle_ff4(r.w[3]) | rle_ff4(r.prev | 0x01010100u)) == 0;
    }
    const uint64_t mine = quick ? RLE_PER_THREAD : walk(nullptr);
    part[tid] = (uint32_t)mine;                             // (a thread's share is below 2^32: the output of one region is at most 3/2 of its bytes + 258 per code)
    __syncthreads();
    if (MODE == 0) {
        for (uint32_t d = 128; d > 0; d >>= 1) { if (tid < d) part[tid] += part[tid + d]; __syncthreads(); }
        if (tid == 0) {     // total = n + what the chunks' sizes differ from their byte counts by: most chunks add nothing
            chunk_out[blockIdx.x] = part[0];
            const uint64_t c0 = (uint64_t)blockIdx.x * RLE_CHUNK, in = n - c0 < RLE_CHUNK ? n - c0 : RLE_CHUNK;
            if (part[0] != in) atomicAdd((unsigned long long *)total, (unsigned long long)part[0] - (unsigned long long)in);   // (wraps: two's complement)
        }
        return;
    }
    for (uint32_t d = 1; d < 256; d <<= 1) {                // inclusive scan
        const uint32_t y = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += y;
        __syncthreads();
    }
    if (mine) walk(dst + chunk_off[blockIdx.x] + (part[tid] - (uint32_t)mine));     // (a quick thread's sixteen bytes are copied by the walk)
}

// chunk_off = exclusive prefix of chunk_out (one workgroup; only the write pass needs it); total[0] = the sum again
__global__ void __launch_bounds__(1024) rle0_scan_kernel(const uint32_t *chunk_out, uint64_t *chunk_off, uint64_t nchunks, uint64_t *total) {
    __shared__ uint64_t part[1024];
    const uint32_t tid = threadIdx.x;
    const uint64_t per = (nchunks + 1023) / 1024, c0 = (uint64_t)tid * per, c1 = c0 + per < nchunks ? c0 + per : nchunks;
    uint64_t sum = 0;
    for (uint64_t c = c0; c < c1; c++) sum += chunk_out[c];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint64_t y = tid >= d ? part[tid - d] : 0ull;
        __syncthreads();
        part[tid] += y;
        __syncthreads();
    }
    uint64_t run = part[tid] - sum;
    for (uint64_t c = c0; c < c1; c++) { chunk_off[c] = run; run += chunk_out[c]; }
    if (tid == 1023) total[0] = part[1023];
}

// ---- host entry points.  ws: rle0_ws_bytes(n) of device memory; the size pass leaves what the write pass needs in it.
size_t rle0_ws_bytes(uint64_t n) {
    const uint64_t nchunks = (n + RLE_CHUNK - 1) / RLE_CHUNK;
    return (size_t)(16 + 8 * nchunks + 4 * nchunks + 2 * nchunks + 64);
}
struct RleWs { uint64_t *total, *off; uint32_t *out; uint16_t *uniform; uint64_t nchunks; };
static RleWs rle_ws(void *ws, uint64_t n) {
    RleWs w;
    w.nchunks = (n + RLE_CHUNK - 1) / RLE_CHUNK;
    uint8_t *p = (uint8_t *)ws;
    w.total = (uint64_t *)p; p += 16;
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
    if (decode) hipLaunchKernelGGL((rle0_pass_kernel<0, true>), dim3((uint32_t)w.nchunks), dim3(256), 0, st, s, n, w.uniform, w.out, w.off, (uint8_t *)nullptr, w.total);
    else hipLaunchKernelGGL((rle0_pass_kernel<0, false>), dim3((uint32_t)w.nchunks), dim3(256), 0, st, s, n, w.uniform, w.out, w.off, (uint8_t *)nullptr, w.total);
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
    hipLaunchKernelGGL(rle0_scan_kernel, dim3(1), dim3(1024), 0, st, w.out, w.off, w.nchunks, w.total + 1);      // (the size pass left the chunk sums)
    if (decode) hipLaunchKernelGGL((rle0_pass_kernel<1, true>), dim3((uint32_t)w.nchunks), dim3(256), 0, st, s, n, w.uniform, w.out, w.off, (uint8_t *)d_dst, w.total + 1);
    else hipLaunchKernelGGL((rle0_pass_kernel<1, false>), dim3((uint32_t)w.nchunks), dim3(256), 0, st, s, n, w.uniform, w.out, w.off, (uint8_t *)d_dst, w.total + 1);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace qb3dev
