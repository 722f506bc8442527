// qb3_amd/csrc/qb3_kernels.hip -- HIP kernels of the MI355X-native QB3 block codec (gfx950, wave64).
//
// What is computed is the reference's per-band 4x4 micro-block code (reference QB3lib/QB3encode.h:155-280,
// 376-451; QB3decode.h:142-412) -- bit-identical streams -- but organised for the GPU:
//
// ENCODE (thread per UNIT = one block of one band; a workgroup owns a CHUNK of consecutive blocks)
//   * the running predictor of the reference (`prv += g -= prv`, QB3encode.h:434) makes the value entering a
//     unit simply the last pixel the curve visited in the previous block, and the rung-switch code needs only
//     the previous block's rung.  So a chunk is self-contained once it also loads ONE halo block (slot 0).
//   * the 4-row tile of the chunk is staged in LDS with coalesced dword loads; each lane gathers its 16 values
//     in curve order, band-differences, deltas, mag-sign, ORs -> rung.
//   * per-unit bit lengths -> workgroup exclusive scan -> bit offsets inside the chunk.
//   * pass 1 (enc_kernel<EMIT=false>) only publishes the chunk's bit total; a single-workgroup scan turns the
//     totals into global bit offsets (64-bit: a 16384^2x3 stream exceeds 2^32 bits); pass 2 rebuilds the codes,
//     ORs them into an LDS staging buffer (ds_or_b32) and stores whole dwords coalesced; the partial dwords
//     at chunk seams are merged with global atomicOr into dwords the scan kernel zeroed.
//   * pass 2 also samples the coder state every `seg_blocks` blocks into the out-of-band decode index.
//
// DECODE (lane per SEGMENT of the index; sequential inside a segment like the reference, parallel across)
//   * a lane owns a private bit reader over aligned dword loads, decodes its blocks band by band into a small
//     LDS scratch block, adds the core band back and stores the 4 rows.
//   * a foreign stream (no index) first goes through dec_index_serial: ONE lane walks the whole stream and
//     rebuilds the index.  That pass is inherently serial (QB3decode.h:445-454) and latency bound.
//
// No MFMA anywhere: this is integer bit packing, bounded by VALU/LDS issue and, ultimately, HBM.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <type_traits>
#include "qb3_dev.h"

namespace qb3dev {

// ------------------------------------------------------------------ small helpers
template <typename T> struct UBits { static constexpr uint32_t v = sizeof(T) == 1 ? 3 : sizeof(T) == 2 ? 4 : sizeof(T) == 4 ? 5 : 6; };

template <typename T> __device__ __forceinline__ T mags_t(T v) {       // reference QB3common.h:127-130
    constexpr uint32_t B = 8 * sizeof(T);
    return (T)((T)(v << 1) ^ (T)(0 - (T)(v >> (B - 1))));
}
template <typename T> __device__ __forceinline__ T smag_t(T v) {       // reference QB3common.h:133-136
    return (T)((T)(v >> 1) ^ (T)(0 - (T)(v & 1)));
}
template <typename T> __device__ __forceinline__ T mabs_t(T v) { return (T)((v >> 1) + (v & 1)); }
template <typename T> __device__ __forceinline__ T mmul_t(T v, T m) { return (T)((T)(mabs_t<T>(v) * (T)(m << 1)) - (T)(v & 1)); }
__device__ __forceinline__ uint32_t topbit64(uint64_t v) { return 63u - (uint32_t)__clzll((long long)v); }
__device__ __forceinline__ uint32_t topbit32(uint32_t v) { return 31u - (uint32_t)__clz((int)v); }
template <typename T> __device__ __forceinline__ uint32_t topbit_t(T v) {
    if (sizeof(T) == 8) return topbit64((uint64_t)v | 1);
    return topbit32((uint32_t)v | 1);
}

// n / d for small n with magic = ceil(2^32 / d); d == 1 has no 32-bit magic
__device__ __forceinline__ uint32_t fastdiv(uint32_t n, uint32_t d, uint32_t magic) { return d == 1 ? n : __umulhi(n, magic); }

// curve nibble i (0 = first visited): x = nib & 3, y = nib >> 2 (reference QB3common.h:168-193)
__device__ __forceinline__ uint32_t curve_nib(uint64_t order, uint32_t i) { return (uint32_t)(order >> (60 - 4 * i)) & 15u; }

// length (incl. change flag) of the rung-switch code for delta in [0, 2^UB)  (reference QB3encode.h:79-89)
template <uint32_t UB> __device__ __forceinline__ uint32_t cs_len(uint32_t delta) {
    constexpr uint32_t n = 1u << UB;
    if (delta == 0) return 1;
    const uint32_t m = (delta < n / 2) ? 2 * (delta - 1) : 2 * (n - delta) - 1;
    return UB + (m >= (1u << (UB - 2))) + (m >= (1u << (UB - 1)));   // 1 flag + (UB-1) + extra bits
}
// the code itself, flag in bit 0
template <uint32_t UB> __device__ __forceinline__ uint32_t cs_code(uint32_t delta) {
    constexpr uint32_t n = 1u << UB, r = UB - 1, half = 1u << (r - 1), top = 1u << r;
    if (delta == 0) return 0;
    const uint32_t m = (delta < n / 2) ? 2 * (delta - 1) : 2 * (n - delta) - 1;
    uint32_t c = (m < half) ? (m << 1) : (m < top) ? (((m - half) << 2) | 1) : (((m - top) << 2) | 3);
    return (c << 1) | 1;
}


typedef const __attribute__((address_space(3))) uint32_t *LdsWords;   // explicit LDS pointer: loads become ds_read
__device__ __forceinline__ LdsWords lds_at(uint32_t byte_off) { return (LdsWords)(uintptr_t)byte_off; }
// 32 stream bits starting at bit `pos` (counted from LDS address 0)
__device__ __forceinline__ uint32_t lds_bits(uint32_t pos) {
    LdsWords p = lds_at((pos >> 3) & ~3u);
    return __builtin_amdgcn_alignbit(p[1], p[0], pos);
}

// wave-wide inclusive scan with DPP (row shifts inside the 16-lane rows, then the two row broadcasts of GFX9)
__device__ __forceinline__ uint32_t wave_iscan32(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);     // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);     // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);     // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);     // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3
    return x;
}
// workgroup exclusive scan of NW independent 32-bit words per lane; ONE barrier; the scratch (NW*4 words) must not
// be rewritten before the caller's next barrier
template <int NW>
__device__ __forceinline__ void block_exscan_dpp(uint32_t (&v)[NW], uint32_t *wsum) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) {
        inc[k] = wave_iscan32(v[k]);
        if (lane == 63) wsum[k * 4 + wave] = inc[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NW; k++) {
        uint32_t base = 0;
#pragma unroll
        for (uint32_t i = 0; i < 3; i++) if (i < wave) base += wsum[k * 4 + i];
        v[k] = base + inc[k] - v[k];
    }
}


// ---- code tables in LDS, generated from the code rules (never transcribed) ---------------------------------
// Encode: ENC_TAB_SIZE entries; rung r in 1..7 occupies [2^(r+1)-4, 2^(r+2)-4), indexed by the mag-sign value,
// entry = len<<12 | code with the middle swap applied (reference QB3encode.h:30-33, 132-141).
// Decode: DEC_TAB_SIZE entries; rung r in 1..7 occupies [2^(r+2)-8, 2^(r+3)-8), indexed by the next r+2 stream
// bits, entry = len<<12 | value with the swap undone (reference QB3decode.h:119-129).
constexpr uint32_t ENC_TAB_SIZE = 508, DEC_TAB_SIZE = 1016;
__device__ __forceinline__ uint32_t enc_tab_off(uint32_t r) { return (2u << r) - 4; }
__device__ __forceinline__ uint32_t dec_tab_off(uint32_t r) { return (4u << r) - 8; }
__device__ __forceinline__ void fill_enc_tab(uint16_t *tab) {
    for (uint32_t idx = threadIdx.x; idx < ENC_TAB_SIZE; idx += blockDim.x) {
        const uint32_t r = topbit32(idx + 4) - 1, top = 1u << r, half = top >> 1;
        uint32_t v = idx - enc_tab_off(r);
        if (v == top || v == top - 1) v ^= 2 * top - 1;
        const uint32_t code = (v < half) ? (v << 1) : (v < top) ? (((v - half) << 2) | 1) : (((v - top) << 2) | 3);
        tab[idx] = (uint16_t)(((r + (v >= half) + (v >= top)) << 12) | code);
    }
}
__device__ __forceinline__ void fill_dec_tab(uint16_t *tab) {
    for (uint32_t idx = threadIdx.x; idx < DEC_TAB_SIZE; idx += blockDim.x) {
        const uint32_t r = topbit32(idx + 8) - 2, top = 1u << r, half = top >> 1, x = idx - dec_tab_off(r);
        uint32_t v, len;
        if (!(x & 1)) { v = (x & (top - 1)) >> 1; len = r; }
        else if (!(x & 2)) { v = ((x >> 2) & (half - 1)) | half; len = r + 1; }
        else { v = ((x >> 2) & (top - 1)) | top; len = r + 2; }
        if (v == top || v == top - 1) v ^= 2 * top - 1;
        tab[idx] = (uint16_t)((len << 12) | v);
    }
}

// workgroup exclusive scan of one u32 per thread (blockDim.x multiple of 64, <= 1024); *total = sum
__device__ __forceinline__ uint32_t block_exscan(uint32_t v, uint32_t *wsum, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d, 64);
        if (lane >= (uint32_t)d) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (uint32_t i = 0; i < nw; i++) {
        uint32_t s = wsum[i];
        if (i < wave) base += s;
        tot += s;
    }
    __syncthreads();    // wsum may be reused
    *total = tot;
    return base + x - v;
}

template <typename V>
__device__ __forceinline__ V block_exscan_v(V v, V *wsum) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    V x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        V y = __shfl_up(x, d, 64);
        if (lane >= (uint32_t)d) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    V base = 0;
    for (uint32_t i = 0; i < nw; i++) if (i < wave) base += wsum[i];
    __syncthreads();
    return (V)(base + x - v);
}

// ------------------------------------------------------------------ encode
struct EncArgs {
    Geometry g;
    const void *img;
    uint32_t *out32;
    uint32_t out_bit0;
    uint32_t slots, nchunks, dpr, magic_dpr, magic_bands;
    uint32_t *chunk_bits;   // per chunk: bits produced
    uint64_t *chunk_off;    // per chunk: exclusive bit offset inside its scan group (SCAN_GROUP chunks)
    uint64_t *group_sum;    // per scan group: bits produced
    uint32_t *scratch;      // per chunk: slot_dw dwords, the chunk's bits starting at bit 0
    uint32_t slot_dw;
    uint32_t *seams;        // per chunk: first and last dword after shifting, for the dwords two chunks share
    uint8_t *cw_has;        // common-factor modes: per chunk and band, does the chunk overwrite the band's factor
    uint64_t *cw_val;       //   ... and with what (cf - 2)
    uint64_t *centry;       //   ... factor state on entering the chunk (after best_scan_kernel)
    uint32_t ntiles;
    uint64_t ts_img, ts_out, ts_ws, ts_idx;     // batched tiles: byte strides from tile to tile (blockIdx.y = tile)
    uint32_t hdr_len;       // container header bytes to stamp in front of the stream (write_header_kernel)
    uint32_t hdr_back;      // distance from the container start to the stream start (= hdr_len without an index chunk)
    uint8_t hdr[80];        // at most 11 + 20 (CB) + 12 (QV) + 12 (SC) + 12 (ix head) bytes
    uint32_t flags;         // tuning switches (QB3_ENC_FLAGS): bit 0 = codes from the LDS table instead of the rule
    uint32_t px_ng, px_magic_ng;    // 16-bit lane-per-block kernel: band groups per block (lanes per block), magic of it
    uint32_t px_aligned;            // lane-per-block kernels: every row of every block is dword aligned (plain dword loads)
    uint64_t *stamps;               // debugging: per-workgroup phase time stamps of enc_px_kernel (null: off)
    uint32_t stamps_n;
    uint8_t *ix_dst;                // coarse index chunk: where the entries go (null: none), "DT" right after them
    uint32_t ix_K, ix_spe, ix_E;    //   ... entries, fine segments per entry, bytes per entry
    EncResult *res;
    BandState st;
    IndexView idx;
    uint32_t have_idx;
};


// Batched tiles: every kernel of the encoder takes the tile from blockIdx.y and shifts its per-tile pointers.
template <typename P> __device__ __forceinline__ P *shift_ptr(P *p, uint64_t bytes) { return p ? (P *)((uint8_t *)p + bytes) : p; }
__device__ __forceinline__ EncArgs enc_for_tile(EncArgs a, uint32_t t) {
    if (t) {
        a.img = (const uint8_t *)a.img + t * a.ts_img;
        a.out32 = shift_ptr(a.out32, t * a.ts_out);
        const uint64_t w = t * a.ts_ws, x = t * a.ts_idx;
        a.chunk_bits = shift_ptr(a.chunk_bits, w); a.chunk_off = shift_ptr(a.chunk_off, w); a.group_sum = shift_ptr(a.group_sum, w);
        a.scratch = shift_ptr(a.scratch, w); a.seams = shift_ptr(a.seams, w); a.res = shift_ptr(a.res, w);
        a.cw_has = shift_ptr(a.cw_has, w); a.cw_val = shift_ptr(a.cw_val, w); a.centry = shift_ptr(a.centry, w);
        a.idx.bitpos = shift_ptr(a.idx.bitpos, x); a.idx.prev = shift_ptr(a.idx.prev, x); a.idx.cf = shift_ptr(a.idx.cf, x);
        a.idx.rung = shift_ptr(a.idx.rung, x); a.idx.ulen = shift_ptr(a.idx.ulen, x);
    }
    return a;
}

// LSB-first bit writer into a zeroed LDS dword buffer shared by the workgroup
struct LdsWriter {
    uint32_t *buf;
    uint64_t acc;
    uint32_t n, w;
    __device__ __forceinline__ void init(uint32_t *b, uint32_t bitpos) { buf = b; acc = 0; n = bitpos & 31; w = bitpos >> 5; }
    __device__ __forceinline__ void put(uint32_t code, uint32_t len) {     // len <= 32, code < 2^len
        acc |= (uint64_t)code << n;
        n += len;
        if (n >= 32) { atomicOr(&buf[w], (uint32_t)acc); w++; acc >>= 32; n -= 32; }
    }
    __device__ __forceinline__ void put64(uint64_t code, uint32_t len) {   // len <= 64
        const uint32_t l0 = len < 32 ? len : 32;
        put((uint32_t)code, l0);
        if (len > 32) put((uint32_t)(code >> 32), len - 32);
    }
    __device__ __forceinline__ void finish() { if (n) atomicOr(&buf[w], (uint32_t)acc); }
};

// one value code at rung r >= 1 (three-length code, reference QB3encode.h:132-141); v already swapped
template <typename T> __device__ __forceinline__ void put_value(LdsWriter &w, T v, uint32_t r) {
    if (sizeof(T) <= 2) {
        const uint32_t x = (uint32_t)v, half = 1u << (r - 1), top = 1u << r;
        const uint32_t code = (x < half) ? (x << 1) : (x < top) ? (((x - half) << 2) | 1) : (((x - top) << 2) | 3);
        const uint32_t len = r + (x >= half) + (x >= top);
        w.put(code, len);
    } else {
        const uint64_t x = (uint64_t)v, half = 1ull << (r - 1), top = 1ull << r;
        if (x < half) w.put64(x << 1, r);
        else if (x < top) w.put64(((x - half) << 2) | 1, r + 1);
        else {
            const uint64_t pay = x - top;           // < 2^r
            w.put(3, 2);
            w.put64(pay, r);                         // r <= 63; at r == 63 this is the reference's 64+1 bit split
        }
    }
}

constexpr uint32_t SCAN_GROUP = 4096;      // chunks per workgroup of enc_scan_kernel

// Everything a unit-per-lane encoder kernel needs before coding: LDS carve, tile staging, gather, deltas.
template <typename T> struct EncFront {
    uint64_t *slot_base; uint32_t *tile, *wsum; uint8_t *rungs; uint16_t *etab; uint32_t *outbuf;
    uint32_t s, c, cb, gblk, rung, nbp, chunk;
    bool valid, payload;
    T used, pv, lastv;
};

template <typename T>
__device__ __forceinline__ void enc_front(const EncArgs &a, const EncArgs &a0, uint8_t *smem, uint32_t outdw, EncFront<T> &f, T (&g)[16]) {
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    const uint32_t bands = a.g.bands, slots = a.slots, dpr = a.dpr, nbp = slots - 1;
    const uint32_t nblocks = (uint32_t)a.g.nblocks, nbx = a.g.nbx;
    const uint64_t stride = a.g.stride;
    const uint32_t rowdw = slots * dpr;

    // LDS carve (all offsets multiples of 8)
    f.slot_base = (uint64_t *)smem;
    f.tile = (uint32_t *)(f.slot_base + slots);
    f.wsum = f.tile + 4 * rowdw;                          // 64 dwords of scan scratch
    f.rungs = (uint8_t *)(f.wsum + 64);                   // slots*bands bytes, padded to 8
    f.etab = (uint16_t *)(f.rungs + ((slots * bands + 7) & ~7u));     // ENC_TAB_SIZE + pad
    f.outbuf = (uint32_t *)(f.etab + 512);
    fill_enc_tab(f.etab);
    uint64_t *slot_base = f.slot_base; uint32_t *tile = f.tile;

    const uint32_t chunk = blockIdx.x;
    const uint32_t g0 = chunk * nbp;                      // first payload block of this chunk
    f.nbp = nbp; f.chunk = chunk;

    // block index of slot s is g0 - 1 + s; invalid slots are clamped to a valid block so loads stay in bounds
    auto slot_block = [&](uint32_t s, bool &valid) -> uint32_t {
        const int64_t g = (int64_t)g0 - 1 + s;
        valid = g >= 0 && g < (int64_t)nblocks;
        return g < 0 ? 0u : (g >= (int64_t)nblocks ? nblocks - 1 : (uint32_t)g);
    };
    auto block_origin = [&](uint32_t g, uint32_t &x0, uint32_t &y0) {
        const uint32_t by = g / nbx, bx = g - by * nbx;
        x0 = (4 * bx + 4 > a.g.w) ? a.g.w - 4 : 4 * bx;     // last column / row is shifted, not padded
        y0 = (4 * by + 4 > a.g.h) ? a.g.h - 4 : 4 * by;     // (reference QB3encode.h:410-416)
    };

    if (tid < slots) {
        bool valid; uint32_t x0, y0;
        block_origin(slot_block(tid, valid), x0, y0);
        slot_base[tid] = (uint64_t)y0 * stride + (uint64_t)x0 * bands;
    }
    for (uint32_t i = tid; i < outdw; i += nthr) f.outbuf[i] = 0;
    __syncthreads();

    // ---- stage the 4-row tile: coalesced dword loads, [row][slot][pixel][band] as in memory
    const uint8_t *imgb = (const uint8_t *)a.img;
    // (four loads in flight per thread before the LDS stores: a load-store-load-store loop pays one memory round trip
    // per element)
    for (uint32_t j0 = 0; j0 < rowdw; j0 += nthr) {
        const uint32_t j = j0 + tid;
        const bool in = j < rowdw;
        const uint32_t s = fastdiv(in ? j : 0, dpr, a.magic_dpr), d = (in ? j : 0) - s * dpr;
        const uint8_t *p0 = imgb + slot_base[s] * sizeof(T) + 4 * d;
        uint32_t v[4];
#pragma unroll
        for (uint32_t r = 0; r < 4; r++) {
            const uint8_t *p = p0 + (uint64_t)r * stride * sizeof(T);
            v[r] = 0;
            if (in) {
                if (((uintptr_t)p & 3) == 0) v[r] = *(const uint32_t *)p;
                else v[r] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
            }
        }
#pragma unroll
        for (uint32_t r = 0; r < 4; r++) if (in) tile[r * rowdw + j] = v[r];
    }
    __syncthreads();

    // ---- per unit: gather in curve order, band difference, delta, mag-sign, rung
    const uint32_t s = fastdiv(tid, bands, a.magic_bands), c = tid - s * bands;
    bool valid = false;
    uint32_t gblk = 0;
    if (s < slots) gblk = slot_block(s, valid);
    const uint32_t cb = a0.g.cband[c < MAXBANDS ? c : 0];
    const T *tt = (const T *)tile;
    const uint64_t order = a.g.order;
    T used = 0, pv = 0, lastv = 0;
    uint32_t rung = 0;
    if (valid) {
        // value entering the unit: last visited pixel of the previous block, or the carried state
        const uint32_t n15 = curve_nib(order, 15);
        if (gblk == 0) pv = (T)a0.st.prev[c];
        else if (s >= 1) {
            const uint32_t e = (((n15 >> 2) * slots + (s - 1)) * 4 + (n15 & 3)) * bands;
            pv = tt[e + c];
            if (cb != c) pv = (T)(pv - tt[e + cb]);
        } else {            // halo unit: its predecessor block is not in the tile
            uint32_t x0, y0;
            block_origin(gblk - 1, x0, y0);
            const T *ip = (const T *)a.img + (uint64_t)(y0 + (n15 >> 2)) * stride + (uint64_t)(x0 + (n15 & 3)) * bands;
            pv = ip[c];
            if (cb != c) pv = (T)(pv - ip[cb]);
        }
        T prv = pv;
        const T cbmask = (cb != c) ? (T)~(T)0 : (T)0;
        // element index = lane part (slot, band) + a wave-uniform part per curve position (scalar arithmetic)
        const uint32_t ebase = s * 4 * bands, rowel = slots * 4 * bands;
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) {
            const uint32_t nib = curve_nib(order, i);
            const uint32_t e = ebase + ((nib >> 2) * rowel + (nib & 3) * bands);
            const T v = (T)(tt[e + c] - (tt[e + cb] & cbmask));    // core band read always: no branch, same-address LDS reads broadcast
            g[i] = mags_t<T>((T)(v - prv));
            used |= g[i];
            prv = v;
        }
        lastv = prv;
        rung = topbit_t<T>(used);
        f.rungs[tid] = (uint8_t)rung;
    }
    __syncthreads();
    f.s = s; f.c = c; f.cb = cb; f.gblk = gblk; f.rung = rung;
    f.valid = valid; f.payload = valid && s >= 1;
    f.used = used; f.pv = pv; f.lastv = lastv;
}

template <typename T, bool STEP>
__global__ void enc_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    constexpr uint32_t UB = UBits<T>::v, UMASK = (1u << UB) - 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    const uint32_t bands = a.g.bands, nblocks = (uint32_t)a.g.nblocks;
    T g[16];
    EncFront<T> f;
    enc_front<T>(a, a0, smem, (31 + (a.slots - 1) * bands * (UB + 2 + 16 * (8 * (uint32_t)sizeof(T) + 1))) / 32 + 1, f, g);
    const uint32_t c = f.c, gblk = f.gblk, rung = f.rung, chunk = f.chunk;
    const bool payload = f.payload;
    const T used = f.used, pv = f.pv, lastv = f.lastv;
    uint8_t *rungs = f.rungs; uint16_t *etab = f.etab; uint32_t *outbuf = f.outbuf, *wsum = f.wsum;

    // ---- unit bit string.  Short units (rung < 8, always the case for 8-bit data) are assembled BEFORE the scan
    // into six pieces of at most 27 bits -- [switch, c0, c1] [c2..c4] [c5..c7] [c8..c10] [c11..c13] [c14, c15] --
    // so that only six words and their packed lengths stay live across the scan (the 16 values die here).
    // Wider units keep their values and are coded from the rule after the scan.
    uint32_t len = 0, prung = 0, delta = 0;
    uint32_t pc[6] = {0, 0, 0, 0, 0, 0}, plens = 0;        // pieces and their lengths (5 bits each)
    bool pieces = false;
    if (payload) {
        prung = (gblk == 0) ? a0.st.rung[c] : rungs[tid - bands];
        delta = (rung - prung) & UMASK;
        const uint32_t csl = cs_len<UB>(delta), csc = cs_code<UB>(delta);
        len = csl;
        if (used <= 1) {            // flag, then the sixteen one-bit values if any is set (reference QB3encode.h:159-166)
            uint32_t bits = 0;
#pragma unroll
            for (uint32_t i = 0; i < 16; i++) bits |= (uint32_t)(g[i] & 1) << i;
            const uint32_t l = 1 + (used ? 16 : 0);
            pc[0] = csc | ((uint32_t)used << csl) | (bits << (csl + 1));
            plens = csl + l;        // <= 8 + 17
            len += l;
            pieces = true;
        } else {
            const T top = (T)((T)1 << rung);
            if (STEP) {     // clear the rung bit of the last value of a 1..10..0 rung-bit run (reference QB3encode.h:169-176)
                uint32_t bits = 0;
#pragma unroll
                for (uint32_t i = 0; i < 16; i++) bits |= (uint32_t)((g[i] >> rung) & 1) << i;
                if ((bits & (bits + 1)) == 0) {
                    const uint32_t n = __popc(bits);    // >= 1 here
#pragma unroll
                    for (uint32_t i = 0; i < 16; i++) if (i + 1 == n) g[i] ^= top;
                }
            }
            if (sizeof(T) == 1 || rung < 8) {
                // code and length per value from the rule, middle swap included (QB3encode.h:30-33,132-141),
                // or from the LDS table (flag bit 0)
                const uint32_t tp = 1u << rung, hf = tp >> 1;
                const uint16_t *tab = etab + enc_tab_off(rung);
                uint32_t acc = csc, al = csl, k = 0, lsum = 0;
#pragma unroll
                for (uint32_t i = 0; i < 16; i++) {
                    uint32_t code, l;
                    if (a.flags & 1) { const uint32_t e = tab[(uint32_t)g[i]]; code = e & 0xfff; l = e >> 12; }
                    else {
                        uint32_t v = (uint32_t)g[i];
                        if (v == tp || v == tp - 1) v ^= 2 * tp - 1;
                        code = (v < hf) ? (v << 1) : (v < tp) ? (((v - hf) << 2) | 1) : (((v - tp) << 2) | 3);
                        l = rung + (v >= hf) + (v >= tp);
                    }
                    acc |= code << al; al += l; lsum += l;
                    if (i == 1 || i == 4 || i == 7 || i == 10 || i == 13 || i == 15) {   // piece boundary (static)
                        pc[k] = acc; plens |= al << (5 * k); k++; acc = 0; al = 0;
                    }
                }
                len += lsum;
                pieces = true;
            } else {                                 // computed three-length code, no swap above rung 7
                uint32_t extra = 0;
                const T half = (T)(top >> 1);
#pragma unroll
                for (uint32_t i = 0; i < 16; i++) extra += (g[i] >= half) + (g[i] >= top);
                len += 16 * rung + extra;
            }
        }
    }
    uint32_t total;
    const uint32_t pos = block_exscan(len, wsum, &total);

    // ---- emit: the chunk's bits are assembled in LDS starting at bit 0 and go to the chunk's private slot.
    // Where they land in the stream is only known after all chunks are counted; enc_concat_kernel moves them.
    // (A single pass with a decoupled look-back was measured slower here: at ~160 chunks/us the prefix frontier
    // cannot keep up with L2 polling latency, and a ticket counter alone caps the kernel at ~88 chunks/us.)
    if (payload) {
        LdsWriter w;
        w.init(outbuf, pos);
        if (pieces) {
#pragma unroll
            for (uint32_t k = 0; k < 6; k++) w.put(pc[k], (plens >> (5 * k)) & 31);
        } else {
            w.put(cs_code<UB>(delta), cs_len<UB>(delta));
#pragma unroll
            for (uint32_t i = 0; i < 16; i++) put_value<T>(w, g[i], rung);
        }
        w.finish();
        // coder state on leaving the image, for handle statefulness (reference QB3encode.h:446-449)
        if (gblk == nblocks - 1) { a.res->prev[c] = (uint64_t)lastv; a.res->rung[c] = rung; a.res->cf[c] = a0.st.cf[c]; }
        if (a.have_idx) {
            if (a.g.ulen_sz == 1) ((uint8_t *)a.idx.ulen)[(uint64_t)gblk * bands + c] = (uint8_t)len;
            else if (a.g.ulen_sz == 2) ((uint16_t *)a.idx.ulen)[(uint64_t)gblk * bands + c] = (uint16_t)len;
            const uint32_t seg = gblk / a.g.seg_blocks;
            if (seg * a.g.seg_blocks == gblk) {
                ((T *)a.idx.prev)[(uint64_t)seg * bands + c] = pv;
                a.idx.rung[(uint64_t)seg * bands + c] = (uint8_t)prung;
                if (c == 0) a.idx.bitpos[seg] = ((uint64_t)chunk << 32) | pos;     // chunk-relative; fixed up by enc_seam_kernel
            }
        }
    }
    __syncthreads();
    const uint32_t nd = (total + 31) >> 5;
    uint32_t *slot = a.scratch + (uint64_t)chunk * a.slot_dw;
    for (uint32_t d = tid; d < nd; d += nthr) slot[d] = outbuf[d];
    if (tid == 0) a.chunk_bits[chunk] = total;
}

// step transform in place (QB3encode.h:169-176)
template <typename T> __device__ __forceinline__ void apply_step(T (&v)[16], uint32_t rung) {
    uint32_t bits = 0;
#pragma unroll
    for (uint32_t i = 0; i < 16; i++) bits |= (uint32_t)((v[i] >> rung) & 1) << i;
    if (bits && (bits & (bits + 1)) == 0) {
        const uint32_t n = __popc(bits);
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) if (i + 1 == n) v[i] ^= (T)((T)1 << rung);
    }
}
// ------------------------------------------------------------------ 8-bit, 1/3/4 bands: lane per BLOCK, in registers
// Specialisation of enc_kernel for the common rasters (uint8, grey / RGB / RGBA, width a multiple of 4, identity
// or default R-G,G,B-G band map, Hilbert or Z curve).  Same bit stream, different organisation.  The kernel is
// bound by instruction issue and memory latency, not by HBM bandwidth, so it is written for instruction count:
//   * a lane owns a whole block: it loads the four rows of the block straight from HBM (B dwords per row: 64
//     lanes x 4*B bytes are one contiguous run, so the loads are coalesced without an LDS tile) plus the one
//     dword that holds the previous block's last visited pixel;
//   * band count and curve are template parameters: v_perm_b32 gathers each band's bytes in curve order, four to
//     a register, and band difference, running delta and mag-sign are byte-parallel (SWAR) on those registers;
//   * the code table is a compile-time constant (code << 8 | length) copied from L2; a unit's bit string is six
//     pieces of at most 27 bits, each built BACKWARDS with one v_lshl_or_b32 per value (the shift count is the
//     entry itself: the hardware uses its low five bits) and its length is the low byte of the sum of the entries;
//   * rungs of the neighbouring block come from the neighbouring lane (DPP wave shift, LDS only across waves);
//     lane 0 of the workgroup is the halo block (computes rungs only), so a chunk is 255 blocks;
//   * one workgroup scan per chunk (block bit lengths, DPP), one 32-bit LDS bit writer per lane.
constexpr uint32_t order_nib(uint64_t order, int i) { return (uint32_t)(order >> (60 - 4 * i)) & 15u; }
// core band of band c under the default map: R-G, G, B-G (, A)   (reference QB3encode.cpp:41-45)
template <int B, bool RGB> constexpr int core_of(int c) { return (RGB && (c == 0 || c == 2)) ? 1 : c; }

// Encode table of the px kernel, built at compile time: rung r (1..7) at entries [2<<r, 4<<r), indexed by the
// mag-sign value; entry = code << 8 | length, middle swap applied (reference QB3encode.h:30-33, 132-141)
struct PxEncTab { alignas(16) uint32_t e[512]; };
constexpr PxEncTab make_px_enc_tab() {
    PxEncTab t{};
    for (uint32_t r = 1; r < 8; r++) {
        const uint32_t top = 1u << r, half = top >> 1;
        for (uint32_t m = 0; m < (2u << r); m++) {
            uint32_t v = m;
            if (v == top || v == top - 1) v ^= 2 * top - 1;
            const uint32_t code = (v < half) ? (v << 1) : (v < top) ? (((v - half) << 2) | 1) : (((v - top) << 2) | 3);
            t.e[(2u << r) + m] = (code << 8) | (r + (v >= half) + (v >= top));
        }
    }
    return t;
}
__device__ const PxEncTab px_enc_tab = make_px_enc_tab();
// rung-switch codes of 8-bit data (3-bit rungs): length in 4-bit fields, code in 8-bit fields, by delta
constexpr uint32_t cs3_len_c(uint32_t d) {
    if (d == 0) return 1;
    const uint32_t m = (d < 4) ? 2 * (d - 1) : 2 * (8 - d) - 1;
    return 3 + (m >= 2) + (m >= 4);
}
constexpr uint32_t cs3_code_c(uint32_t d) {
    if (d == 0) return 0;
    const uint32_t m = (d < 4) ? 2 * (d - 1) : 2 * (8 - d) - 1;
    const uint32_t c = (m < 2) ? (m << 1) : (m < 4) ? (((m - 2) << 2) | 1) : (((m - 4) << 2) | 3);
    return (c << 1) | 1;
}
constexpr uint32_t cs3_lens() { uint32_t v = 0; for (uint32_t d = 0; d < 8; d++) v |= cs3_len_c(d) << (4 * d); return v; }
constexpr uint64_t cs3_codes() { uint64_t v = 0; for (uint32_t d = 0; d < 8; d++) v |= (uint64_t)cs3_code_c(d) << (8 * d); return v; }

// four independent byte subtractions
__device__ __forceinline__ uint32_t swar_sub8(uint32_t x, uint32_t y) {
    return ((x | 0x80808080u) - (y & 0x7f7f7f7fu)) ^ (~(x ^ y) & 0x80808080u);
}
// mag-sign of four bytes (reference QB3common.h:127-130): (d << 1) ^ (d < 0 ? 0xff : 0)
__device__ __forceinline__ uint32_t swar_mags8(uint32_t d) {
    const uint32_t s = d & 0x80808080u, ff = (s << 1) - (s >> 7);
    return ((d << 1) & 0xfefefefeu) ^ ff;
}
// bytes of band c at curve positions 4q..4q+3 from the block's rows (w[y][k] = dword k of row y)
template <int B, uint64_t ORDER>
__device__ __forceinline__ uint32_t gather_quad(const uint32_t (&w)[4][B], int q, int c) {
    int ry[4], rk[4], rb[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int nib = (int)order_nib(ORDER, 4 * q + j), bi = (nib & 3) * B + c;
        ry[j] = nib >> 2; rk[j] = bi >> 2; rb[j] = bi & 3;
    }
    auto same = [&](int i, int j) { return ry[i] == ry[j] && rk[i] == rk[j]; };
    // at most two source registers: one v_perm_b32 (selector 0..3 = bytes of the second operand, 4..7 of the first)
    int other = -1;
    bool two = true;
#pragma unroll
    for (int j = 1; j < 4; j++)
        if (!same(j, 0)) { if (other < 0) other = j; else if (!same(j, other)) two = false; }
    if (two) {
        uint32_t sel = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) sel |= (uint32_t)(same(j, 0) ? rb[j] : 4 + rb[j]) << (8 * j);
        const int o = other < 0 ? 0 : other;
        return __builtin_amdgcn_perm(w[ry[o]][rk[o]], w[ry[0]][rk[0]], sel);
    }
    const uint32_t lo = __builtin_amdgcn_perm(w[ry[1]][rk[1]], w[ry[0]][rk[0]], (uint32_t)((4 + rb[1]) << 8 | rb[0]));
    const uint32_t hi = __builtin_amdgcn_perm(w[ry[3]][rk[3]], w[ry[2]][rk[2]], (uint32_t)((4 + rb[3]) << 8 | rb[2]));
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

// LDS bit writer with a 32-bit accumulator, for pieces of at most 27 bits
struct LdsWriter32 {
    uint32_t *buf;
    uint32_t acc, n, w;
    __device__ __forceinline__ void init(uint32_t *b, uint32_t bitpos) { buf = b; acc = 0; n = bitpos & 31; w = bitpos >> 5; }
    __device__ __forceinline__ void put(uint32_t code, uint32_t len) {     // len <= 27, code < 2^len
        acc |= code << n;
        const uint32_t n2 = n + len;
        if (n2 >= 32) {                                                    // then n >= 5
            atomicOr(&buf[w], acc); w++;
            acc = code >> (32 - n);
            n = n2 - 32;
        } else n = n2;
    }
    __device__ __forceinline__ void finish() { if (n) atomicOr(&buf[w], acc); }
};

template <int B, bool RGB, uint64_t ORDER, bool STEP>
__global__ void __launch_bounds__(256, 4) enc_px_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    constexpr uint32_t UMASK = 7;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint64_t t_start = a0.stamps ? clock64() : 0;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nblocks = (uint32_t)a.g.nblocks, nbx = a.g.nbx;
    const uint64_t stride = a.g.stride;

    uint32_t *etab = (uint32_t *)smem;                      // 512 entries
    uint32_t *wsum = etab + 512;                            // 64 dwords: scan scratch, [32..35] rungs of each wave's last lane
    uint32_t *outbuf = wsum + 64;                           // slot_dw dwords (a multiple of 4)
    // the code table is asked for now and written to LDS only before the first barrier: its round trip runs beside the
    // pixel loads instead of in front of them
    const uint4 tabv = ((const uint4 *)px_enc_tab.e)[tid & 127];
    for (uint32_t i = tid; i < a.slot_dw / 4; i += 256) ((uint4 *)outbuf)[i] = make_uint4(0, 0, 0, 0);
    const uint32_t etab_off = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)smem;

    const uint32_t chunk = blockIdx.x;
    const bool stamp = a0.stamps && chunk < a0.stamps_n && threadIdx.x == 64;
    if (stamp) a0.stamps[chunk * 8 + 0] = t_start;
    const int64_t gs = (int64_t)chunk * 255 - 1 + tid;     // lane 0 is the halo block
    const bool valid = gs >= 0 && gs < (int64_t)nblocks, payload = valid && tid >= 1;
    const uint32_t gblk = valid ? (uint32_t)gs : 0u;

    // ---- load the block (4 rows x B dwords) and the dword holding the previous block's last visited pixel
    uint32_t w[4][B];
    uint32_t pd = 0;
    constexpr uint32_t n15 = order_nib(ORDER, 15);
    // Rows need not be dword aligned (odd widths and strides, the shifted last column, any pointer): a row is read as the
    // aligned dwords that cover it -- one more than it has when it is not aligned -- and funnel-shifted into place.
    // Nothing is read beyond the aligned dword that holds the row's last byte.
    auto load_row = [&](const uint8_t *p, uint32_t (&row)[B]) {
        const uint32_t sh = 8 * ((uint32_t)(uintptr_t)p & 3);
        const uint32_t *q = (const uint32_t *)((uintptr_t)p & ~(uintptr_t)3);
        uint32_t d[B + 1];
#pragma unroll
        for (int t = 0; t < B; t++) d[t] = q[t];
        d[B] = sh ? q[B] : 0u;
#pragma unroll
        for (int t = 0; t < B; t++) row[t] = __builtin_amdgcn_alignbit(d[t + 1], d[t], sh);
    };
    if (valid) {
        const uint32_t by = gblk / nbx, bx = gblk - by * nbx;
        const uint32_t x0 = (4 * bx + 4 > a.g.w) ? a.g.w - 4 : 4 * bx;     // last column / row is shifted, not padded
        const uint32_t y0 = (4 * by + 4 > a.g.h) ? a.g.h - 4 : 4 * by;
        const uint8_t *p0 = (const uint8_t *)a.img + (uint64_t)y0 * stride + (uint64_t)x0 * B;
        const uint8_t *pp = nullptr;      // the four bytes that end the previous block's row holding its last visited pixel
        if (gblk) {
            const uint32_t pb = gblk - 1, pby = pb / nbx, pbx = pb - pby * nbx;
            const uint32_t px0 = (4 * pbx + 4 > a.g.w) ? a.g.w - 4 : 4 * pbx;
            const uint32_t py0 = (4 * pby + 4 > a.g.h) ? a.g.h - 4 : 4 * pby;
            pp = (const uint8_t *)a.img + (uint64_t)(py0 + (n15 >> 2)) * stride + (uint64_t)px0 * B + 4 * (B - 1);
        }
        if (a.px_aligned) {             // workgroup uniform: width, stride and pointer are multiples of 4
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t *rp = (const uint32_t *)(p0 + (uint64_t)r * stride);
#pragma unroll
                for (int t = 0; t < B; t++) w[r][t] = rp[t];
            }
            if (gblk) pd = *(const uint32_t *)pp;
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) load_row(p0 + (uint64_t)r * stride, w[r]);
            if (gblk) {
                const uint32_t sh = 8 * ((uint32_t)(uintptr_t)pp & 3);
                const uint32_t *q = (const uint32_t *)((uintptr_t)pp & ~(uintptr_t)3);
                pd = __builtin_amdgcn_alignbit(sh ? q[1] : 0u, q[0], sh);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int k = 0; k < B; k++) w[r][k] = 0;
    }

    if (stamp) a0.stamps[chunk * 8 + 1] = clock64();
    // ---- per band: bytes in curve order, band difference, running delta, mag-sign -- four values per register
    uint32_t cur[B][4];
#pragma unroll
    for (int c = 0; c < B; c++)
#pragma unroll
        for (int q = 0; q < 4; q++) cur[c][q] = gather_quad<B, ORDER>(w, q, c);
    uint32_t gp[B][4], usedv[B], lastv[B], pvv[B];
    uint32_t rp_packed = 0;
#pragma unroll
    for (int c = 0; c < B; c++) {
        const int cb = core_of<B, RGB>(c);
        uint32_t prv;
        if (gblk == 0) prv = (uint32_t)a0.st.prev[c] & 0xffu;
        else {      // pixel x = 3 of the previous block sits in the last dword of its row: byte c + 4 - B
            prv = (pd >> (8 * (c + 4 - B))) & 0xffu;
            if (cb != c) prv = (prv - ((pd >> (8 * (cb + 4 - B))) & 0xffu)) & 0xffu;
        }
        pvv[c] = prv;
        uint32_t x[4];
#pragma unroll
        for (int q = 0; q < 4; q++) x[q] = (cb != c) ? swar_sub8(cur[c][q], cur[cb][q]) : cur[c][q];
        uint32_t u = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t before = q ? __builtin_amdgcn_alignbit(x[q], x[q - 1], 24) : ((x[0] << 8) | prv);
            gp[c][q] = swar_mags8(swar_sub8(x[q], before));
            u |= gp[c][q];
        }
        u |= u >> 16; u |= u >> 8; u &= 0xffu;
        usedv[c] = u; lastv[c] = x[3] >> 24;
        rp_packed |= topbit32(u | 1) << (4 * c);
    }
    if (stamp) a0.stamps[chunk * 8 + 2] = clock64() + (rp_packed & 0);
    // rungs of the previous block: neighbouring lane, or the last lane of the previous wave through LDS
    uint32_t prp = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rp_packed, 0x138, 0xf, 0xf, false);      // wave_shr:1
    if (lane == 63) wsum[32 + wave] = rp_packed;
    if (tid < 128) ((uint4 *)etab)[tid] = tabv;
    __syncthreads();
    if (lane == 0 && wave) prp = wsum[32 + wave - 1];
    if (gblk == 0) { prp = 0;
#pragma unroll
        for (int c = 0; c < B; c++) prp |= ((uint32_t)a0.st.rung[c] & 15u) << (4 * c); }

    if (stamp) a0.stamps[chunk * 8 + 3] = clock64();
    // ---- per band: the unit's bit string as six pieces of at most 27 bits; pl = piece length (low byte)
    uint32_t pc[B][6], pl[B][6], lens[B], blen[1] = { 0 };
#pragma unroll
    for (int c = 0; c < B; c++) {
#pragma unroll
        for (int k = 0; k < 6; k++) { pc[c][k] = 0; pl[c][k] = 0; }
        lens[c] = 0;
        if (payload) {
            const uint32_t rung = (rp_packed >> (4 * c)) & 15u, prung = (prp >> (4 * c)) & 15u, used = usedv[c];
            const uint32_t delta = (rung - prung) & UMASK;
            const uint32_t csl = __builtin_amdgcn_ubfe(cs3_lens(), 4 * delta, 4), csc = (uint32_t)(cs3_codes() >> (8 * delta)) & 0xffu;
            if (used <= 1) {
                uint32_t bits = 0;
#pragma unroll
                for (int i = 0; i < 16; i++) bits |= ((gp[c][i >> 2] >> (8 * (i & 3))) & 1u) << i;
                // switch, the "not all zero" flag, then the 16 bits: split so that no piece exceeds 27 bits
                pc[c][0] = csc | (used << csl); pl[c][0] = csl + 1;
                pc[c][1] = bits; pl[c][1] = used ? 16 : 0;
                lens[c] = pl[c][0] + pl[c][1];
            } else {
                uint32_t g4[4] = {gp[c][0], gp[c][1], gp[c][2], gp[c][3]};
                if (STEP) {     // clear the rung bit of the last value of a 1..10..0 rung-bit run (reference QB3encode.h:169-176)
                    uint32_t bits = 0;
#pragma unroll
                    for (int i = 0; i < 16; i++) bits |= ((g4[i >> 2] >> (8 * (i & 3) + rung)) & 1u) << i;
                    if ((bits & (bits + 1)) == 0) {
                        const uint32_t n = __popc(bits) - 1;        // index of the value to change
#pragma unroll
                        for (int q = 0; q < 4; q++) if ((n >> 2) == (uint32_t)q) g4[q] ^= (1u << rung) << (8 * (n & 3));
                    }
                }
                const uint32_t tb = etab_off + (8u << rung);         // byte address of the rung's table region
                constexpr int first[7] = {0, 2, 5, 8, 11, 14, 16};  // piece k holds values first[k] .. first[k+1]-1
                uint32_t lsum = 0;
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    uint32_t acc = 0, s = 0;
#pragma unroll
                    for (int i = first[k + 1] - 1; i >= first[k]; i--) {
                        const uint32_t m = (g4[i >> 2] >> (8 * (i & 3))) & 0xffu;
                        const uint32_t e = *lds_at((m << 2) + tb);
                        acc = (acc << (e & 31u)) | (e >> 8);
                        s += e;
                    }
                    if (k == 0) { acc = (acc << csl) | csc; s += csl; }
                    pc[c][k] = acc; pl[c][k] = s & 0xffu; lsum += s & 0xffu;
                }
                lens[c] = lsum;
            }
            blen[0] += lens[c];
        }
    }
    if (stamp) a0.stamps[chunk * 8 + 4] = clock64() + (blen[0] & 0);
    const uint32_t mybits = blen[0];
    block_exscan_dpp<1>(blen, wsum);
    const uint32_t pos = blen[0], total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    (void)mybits;

    if (stamp) a0.stamps[chunk * 8 + 5] = clock64();
    if (payload) {
        LdsWriter32 wr;
        wr.init(outbuf, pos);
#pragma unroll
        for (int c = 0; c < B; c++)
#pragma unroll
            for (int k = 0; k < 6; k++) wr.put(pc[c][k], pl[c][k]);
        wr.finish();
        if (gblk == nblocks - 1) {
#pragma unroll
            for (int c = 0; c < B; c++) { a.res->prev[c] = lastv[c]; a.res->rung[c] = (rp_packed >> (4 * c)) & 15u; a.res->cf[c] = a0.st.cf[c]; }
        }
        if (a.have_idx) {
            uint8_t *ul = (uint8_t *)a.idx.ulen + (uint64_t)gblk * B;
#pragma unroll
            for (int c = 0; c < B; c++) ul[c] = (uint8_t)lens[c];
            const uint32_t seg = gblk / a.g.seg_blocks;
            if (seg * a.g.seg_blocks == gblk) {
#pragma unroll
                for (int c = 0; c < B; c++) {
                    ((uint8_t *)a.idx.prev)[(uint64_t)seg * B + c] = (uint8_t)pvv[c];
                    a.idx.rung[(uint64_t)seg * B + c] = (uint8_t)((prp >> (4 * c)) & 15u);
                }
                a.idx.bitpos[seg] = ((uint64_t)chunk << 32) | pos;
            }
        }
    }
    if (stamp) a0.stamps[chunk * 8 + 6] = clock64();
    __syncthreads();
    const uint32_t nd4 = (total + 127) >> 7;
    uint4 *slot = (uint4 *)(a.scratch + (uint64_t)chunk * a.slot_dw);
    for (uint32_t d = tid; d < nd4; d += 256) slot[d] = ((const uint4 *)outbuf)[d];
    if (tid == 0) a.chunk_bits[chunk] = total;
    if (stamp) a0.stamps[chunk * 8 + 7] = clock64();
}

// ------------------------------------------------------------------ 16-bit: lane per (block, band group), in registers
// The 16-bit counterpart of enc_px_kernel.  A lane owns BG <= 4 bands of one block (bands = NG x BG: 8-band data
// is two lanes per block); units of a block are consecutive in the stream, so lane order is still stream order.
// Two values per register: v_perm_b32 gathers curve-ordered pairs, band difference / running delta / mag-sign are
// packed 16-bit operations (v_pk_sub_u16, v_pk_lshlrev_b16, v_pk_ashrrev_i16).  Rungs up to 7 use the same
// compile-time code table as the 8-bit kernel, higher rungs the code rule in ALU (no middle swap above rung 7);
// pieces are 64 bits wide (three codes of at most 17 bits).  Slot 0 of a workgroup (its first NG lanes) is the halo
// block, so a chunk is 256/NG - 1 blocks.
typedef uint16_t u16x2_t __attribute__((ext_vector_type(2)));
typedef int16_t i16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_sub16(uint32_t x, uint32_t y) {
    return __builtin_bit_cast(uint32_t, (u16x2_t)(__builtin_bit_cast(u16x2_t, x) - __builtin_bit_cast(u16x2_t, y)));
}
__device__ __forceinline__ uint32_t pk_mags16(uint32_t d) {       // (d << 1) ^ (d >> 15), two 16-bit lanes
    const u16x2_t a = __builtin_bit_cast(u16x2_t, d) << (u16x2_t)(uint16_t)1;
    const i16x2_t s = __builtin_bit_cast(i16x2_t, d) >> (i16x2_t)(int16_t)15;
    return __builtin_bit_cast(uint32_t, a) ^ __builtin_bit_cast(uint32_t, s);
}
// values 2k, 2k+1 (curve order) of band c of the lane's group; w[y][j] = dword j of the lane's row y, halfword
// x*BG + c of it is band c of pixel x
template <int BG, uint64_t ORDER>
__device__ __forceinline__ uint32_t gather_pair16(const uint32_t (&w)[4][2 * BG], int k, int c) {
    const int n0 = (int)order_nib(ORDER, 2 * k), n1 = (int)order_nib(ORDER, 2 * k + 1);
    const int h0 = (n0 & 3) * BG + c, h1 = (n1 & 3) * BG + c;
    const uint32_t sel = (uint32_t)(2 * (h0 & 1)) | (uint32_t)(2 * (h0 & 1) + 1) << 8 |
                         (uint32_t)(4 + 2 * (h1 & 1)) << 16 | (uint32_t)(4 + 2 * (h1 & 1) + 1) << 24;
    return __builtin_amdgcn_perm(w[n1 >> 2][h1 >> 1], w[n0 >> 2][h0 >> 1], sel);
}

template <int BG, bool RGB, uint64_t ORDER, bool STEP>
__global__ void __launch_bounds__(256, 2) enc_px16_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    constexpr uint32_t UB = 4, UMASK = 15;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t nblocks = (uint32_t)a.g.nblocks, nbx = a.g.nbx, B = a.g.bands, NG = a.px_ng, S = 256 / NG;
    const uint64_t stride = a.g.stride;                     // in values
    const uint32_t slot = fastdiv(tid, NG, a.px_magic_ng), grp = tid - slot * NG, band0 = grp * BG;

    uint32_t *etab = (uint32_t *)smem;                      // 512 entries
    uint32_t *wsum = etab + 512;                            // 64 dwords of scan scratch
    uint32_t *rp_s = wsum + 64;                             // 256: every lane's packed rungs
    uint32_t *outbuf = rp_s + 256;                          // slot_dw dwords (a multiple of 4)
    const uint4 tabv = ((const uint4 *)px_enc_tab.e)[tid & 127];     // written to LDS before the first barrier, see enc_px_kernel
    for (uint32_t i = tid; i < a.slot_dw / 4; i += 256) ((uint4 *)outbuf)[i] = make_uint4(0, 0, 0, 0);
    const uint32_t etab_off = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)smem;

    const uint32_t chunk = blockIdx.x;
    const int64_t gs = (int64_t)chunk * (S - 1) - 1 + slot; // slot 0 is the halo block
    const bool valid = slot < S && gs >= 0 && gs < (int64_t)nblocks, payload = valid && slot >= 1;
    const uint32_t gblk = valid ? (uint32_t)gs : 0u;

    // ---- load the lane's bands of the block (4 rows) and of the previous block's last visited pixel
    uint32_t w[4][2 * BG], pvals[BG];
    constexpr uint32_t n15 = order_nib(ORDER, 15);
#pragma unroll
    for (int c = 0; c < BG; c++) pvals[c] = 0;
    // N dwords starting at a halfword address: when it is not dword aligned (odd strides, the shifted last column, odd
    // widths) the N+1 aligned dwords covering them are read and funnel-shifted; nothing is read beyond the aligned dword
    // holding the last halfword
    auto load_dw = [&](const uint16_t *p, uint32_t *dst, auto nconst) {
        constexpr int N = decltype(nconst)::value;
        if (a.px_aligned) {                 // workgroup uniform
            const uint32_t *q = (const uint32_t *)p;
#pragma unroll
            for (int t = 0; t < N; t++) dst[t] = q[t];
        } else {
            const uint32_t sh = 8 * ((uint32_t)(uintptr_t)p & 2);
            const uint32_t *q = (const uint32_t *)((uintptr_t)p & ~(uintptr_t)3);
            uint32_t d[N + 1];
#pragma unroll
            for (int t = 0; t < N; t++) d[t] = q[t];
            d[N] = sh ? q[N] : 0u;
#pragma unroll
            for (int t = 0; t < N; t++) dst[t] = __builtin_amdgcn_alignbit(d[t + 1], d[t], sh);
        }
    };
    if (valid) {
        const uint32_t by = gblk / nbx, bx = gblk - by * nbx;
        const uint32_t x0 = (4 * bx + 4 > a.g.w) ? a.g.w - 4 : 4 * bx;     // last column / row is shifted, not padded
        const uint32_t y0 = (4 * by + 4 > a.g.h) ? a.g.h - 4 : 4 * by;
        const uint16_t *p0 = (const uint16_t *)a.img + (uint64_t)y0 * stride + (uint64_t)x0 * B + band0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint16_t *rowp = p0 + (uint64_t)r * stride;
            if (BG % 2 == 0) {      // a pixel's BG values are whole dwords; pixels are B values apart
#pragma unroll
                for (int x = 0; x < 4; x++) load_dw(rowp + (uint64_t)x * B, &w[r][x * (BG / 2)], std::integral_constant<int, (BG / 2 ? BG / 2 : 1)>());
            } else                  // BG == bands: the row of the block is contiguous
                load_dw(rowp, &w[r][0], std::integral_constant<int, 2 * BG>());
        }
        if (gblk) {
            const uint32_t pb = gblk - 1, pby = pb / nbx, pbx = pb - pby * nbx;
            const uint32_t px0 = (4 * pbx + 4 > a.g.w) ? a.g.w - 4 : 4 * pbx;
            const uint32_t py0 = (4 * pby + 4 > a.g.h) ? a.g.h - 4 : 4 * pby;
            const uint16_t *q = (const uint16_t *)a.img + (uint64_t)(py0 + (n15 >> 2)) * stride + (uint64_t)(px0 + (n15 & 3)) * B + band0;
#pragma unroll
            for (int c = 0; c < BG; c++) pvals[c] = q[c];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int k = 0; k < 2 * BG; k++) w[r][k] = 0;
    }

    // ---- per band: values in curve order, band difference, running delta, mag-sign -- two values per register
    uint32_t cur[BG][8];
#pragma unroll
    for (int c = 0; c < BG; c++)
#pragma unroll
        for (int k = 0; k < 8; k++) cur[c][k] = gather_pair16<BG, ORDER>(w, k, c);
    uint32_t gp[BG][8], usedv[BG], lastv[BG], pvv[BG];
    uint32_t rp_packed = 0;
#pragma unroll
    for (int c = 0; c < BG; c++) {
        const int cb = core_of<BG, RGB>(c);
        uint32_t prv;
        // the R-G, G, B-G map applies to the first three bands of the image: group 0 only
        const bool diff = cb != c && grp == 0;
        if (gblk == 0) prv = (uint32_t)a0.st.prev[band0 + c] & 0xffffu;
        else prv = diff ? (pvals[c] - pvals[cb]) & 0xffffu : pvals[c];
        pvv[c] = prv;
        uint32_t x[8], u = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = (cb != c) ? pk_sub16(cur[c][k], diff ? cur[cb][k] : 0u) : cur[c][k];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t before = k ? __builtin_amdgcn_alignbit(x[k], x[k - 1], 16) : ((x[0] << 16) | prv);
            gp[c][k] = pk_mags16(pk_sub16(x[k], before));
            u |= gp[c][k];
        }
        u = (u | (u >> 16)) & 0xffffu;
        usedv[c] = u; lastv[c] = x[7] >> 16;
        rp_packed |= topbit32(u | 1) << (4 * c);
    }
    // rungs of the same bands of the previous block: NG lanes back
    rp_s[tid] = rp_packed;
    if (tid < 128) ((uint4 *)etab)[tid] = tabv;
    __syncthreads();
    uint32_t prp = tid >= NG ? rp_s[tid - NG] : 0u;
    if (gblk == 0) { prp = 0;
#pragma unroll
        for (int c = 0; c < BG; c++) prp |= ((uint32_t)a0.st.rung[band0 + c] & 15u) << (4 * c); }

    // ---- per band: the unit's bit string as six pieces (64-bit), pl = piece length
    uint64_t pc[BG][6];
    uint32_t pl[BG][6], lens[BG], blen[1] = { 0 };
#pragma unroll
    for (int c = 0; c < BG; c++) {
#pragma unroll
        for (int k = 0; k < 6; k++) { pc[c][k] = 0; pl[c][k] = 0; }
        lens[c] = 0;
        if (payload) {
            const uint32_t rung = (rp_packed >> (4 * c)) & 15u, prung = (prp >> (4 * c)) & 15u, used = usedv[c];
            const uint32_t delta = (rung - prung) & UMASK;
            const uint32_t csl = cs_len<UB>(delta), csc = cs_code<UB>(delta);
            if (used <= 1) {
                uint32_t bits = 0;
#pragma unroll
                for (int i = 0; i < 16; i++) bits |= ((gp[c][i >> 1] >> (16 * (i & 1))) & 1u) << i;
                pc[c][0] = csc | (used << csl); pl[c][0] = csl + 1;
                pc[c][1] = bits; pl[c][1] = used ? 16 : 0;
                lens[c] = pl[c][0] + pl[c][1];
            } else {
                uint32_t g8[8];
#pragma unroll
                for (int k = 0; k < 8; k++) g8[k] = gp[c][k];
                if (STEP) {     // clear the rung bit of the last value of a 1..10..0 rung-bit run (reference QB3encode.h:169-176)
                    uint32_t bits = 0;
#pragma unroll
                    for (int i = 0; i < 16; i++) bits |= ((g8[i >> 1] >> (16 * (i & 1) + rung)) & 1u) << i;
                    if ((bits & (bits + 1)) == 0) {
                        const uint32_t n = __popc(bits) - 1;        // index of the value to change
#pragma unroll
                        for (int k = 0; k < 8; k++) if ((n >> 1) == (uint32_t)k) g8[k] ^= (1u << rung) << (16 * (n & 1));
                    }
                }
                constexpr int first[7] = {0, 2, 5, 8, 11, 14, 16};  // piece k holds values first[k] .. first[k+1]-1
                uint32_t lsum = 0;
                if (rung <= 7) {                                     // all values below 256: the code table
                    const uint32_t tb = etab_off + (8u << rung);
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        uint32_t acc = 0, s = 0;
#pragma unroll
                        for (int i = first[k + 1] - 1; i >= first[k]; i--) {
                            const uint32_t m = (g8[i >> 1] >> (16 * (i & 1))) & 0xffffu;
                            const uint32_t e = *lds_at((m << 2) + tb);
                            acc = (acc << (e & 31u)) | (e >> 8);
                            s += e;
                        }
                        s &= 0xffu;
                        uint64_t a64 = acc;
                        if (k == 0) { a64 = (a64 << csl) | csc; s += csl; }
                        pc[c][k] = a64; pl[c][k] = s; lsum += s;
                    }
                } else {                                             // the code rule (reference QB3encode.h:30-33), no swap above rung 7
                    const uint32_t top = 1u << rung, half = top >> 1;
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        uint64_t acc = 0;
                        uint32_t s = 0;
#pragma unroll
                        for (int i = first[k + 1] - 1; i >= first[k]; i--) {
                            const uint32_t m = (g8[i >> 1] >> (16 * (i & 1))) & 0xffffu;
                            const bool c1 = m >= half, c2 = m >= top;
                            const uint32_t code = c2 ? (((m - top) << 2) | 3u) : c1 ? (((m - half) << 2) | 1u) : (m << 1);
                            const uint32_t len = rung + c1 + c2;
                            acc = (acc << len) | code;
                            s += len;
                        }
                        if (k == 0) { acc = (acc << csl) | csc; s += csl; }
                        pc[c][k] = acc; pl[c][k] = s; lsum += s;
                    }
                }
                lens[c] = lsum;
            }
            blen[0] += lens[c];
        }
    }
    block_exscan_dpp<1>(blen, wsum);
    const uint32_t pos = blen[0], total = wsum[0] + wsum[1] + wsum[2] + wsum[3];

    if (payload) {
        LdsWriter wr;
        wr.init(outbuf, pos);
#pragma unroll
        for (int c = 0; c < BG; c++)
#pragma unroll
            for (int k = 0; k < 6; k++) wr.put64(pc[c][k], pl[c][k]);
        wr.finish();
        if (gblk == nblocks - 1) {
#pragma unroll
            for (int c = 0; c < BG; c++) { a.res->prev[band0 + c] = lastv[c]; a.res->rung[band0 + c] = (rp_packed >> (4 * c)) & 15u; a.res->cf[band0 + c] = a0.st.cf[band0 + c]; }
        }
        if (a.have_idx) {
            uint16_t *ul = (uint16_t *)a.idx.ulen + (uint64_t)gblk * B + band0;
#pragma unroll
            for (int c = 0; c < BG; c++) ul[c] = (uint16_t)lens[c];
            const uint32_t seg = gblk / a.g.seg_blocks;
            if (seg * a.g.seg_blocks == gblk) {
#pragma unroll
                for (int c = 0; c < BG; c++) {
                    ((uint16_t *)a.idx.prev)[(uint64_t)seg * B + band0 + c] = (uint16_t)pvv[c];
                    a.idx.rung[(uint64_t)seg * B + band0 + c] = (uint8_t)((prp >> (4 * c)) & 15u);
                }
                if (grp == 0) a.idx.bitpos[seg] = ((uint64_t)chunk << 32) | pos;
            }
        }
    }
    __syncthreads();
    const uint32_t nd4 = (total + 127) >> 7;
    uint4 *slotp = (uint4 *)(a.scratch + (uint64_t)chunk * a.slot_dw);
    for (uint32_t d = tid; d < nd4; d += 256) slotp[d] = ((const uint4 *)outbuf)[d];
    if (tid == 0) a.chunk_bits[chunk] = total;
}

// ------------------------------------------------------------------ common-factor + index coding (BEST)
// Reference: encode_best (QB3encode.h:617-724), cfgenc (:283-361), ienc (:557-613).  A unit can be coded
// plainly, as common factor times a smaller group, or as up to eight distinct values plus indices.  The only
// state besides the rung is pcf, the previous factor of the band.  A unit overwrites pcf with cf-2 exactly when
// cf >= 2 and index coding does not beat the "factor differs" size -- a condition that does not involve pcf
// itself -- so pcf is a LAST-WRITER scan over units: pass 0 records each chunk's last writer per band,
// best_scan_kernel carries it across chunks, pass 1 codes with the right pcf.
template <typename T> __device__ __forceinline__ T mdiv_t(T v, T cf) { return (T)((T)((T)(mabs_t<T>(v) / cf) << 1) - (T)(v & 1)); }

template <typename T> __device__ __forceinline__ T gcf_t(const T (&g)[16]) {      // gcd of the non-zero magnitudes (QB3encode.h:98-126)
    // a magnitude of 1 settles it; so does an odd value next to an even one... only the first is cheap to see in every
    // lane at once, and on noisy data it spares most lanes the divergent Euclid loop
    bool one = false;
#pragma unroll
    for (uint32_t i = 0; i < 16; i++) one = one || mabs_t<T>(g[i]) == 1;
    if (one) return 1;
    T x = 0;
#pragma unroll 1
    for (uint32_t i = 0; i < 16 && x != 1; i++) {
        T y = mabs_t<T>(g[i]);
        while (y) { const T t = (T)(x % y); x = y; y = t; }
    }
    return x;
}
// bit length of one value coded on its own at rung r (reference qb3csztbl, QB3encode.h:144-150): rung 0 is one raw
// bit, rungs 1-2 plain, rungs 3-7 with the middle swap, above that plain
template <typename T> __device__ __forceinline__ uint32_t vlen_t(T v, uint32_t r) {
    if (r == 0) return 1;
    const T top = (T)((T)1 << r), half = (T)(top >> 1);
    if (r >= 3 && r < 8 && (v == top || v == (T)(top - 1))) v ^= (T)(2 * top - 1);
    return r + (v >= half) + (v >= top);
}
template <typename T> __device__ __forceinline__ void put_single(LdsWriter &w, T v, uint32_t r) {
    if (r == 0) { w.put((uint32_t)v & 1, 1); return; }
    const T top = (T)((T)1 << r);
    if (r >= 3 && r < 8 && (v == top || v == (T)(top - 1))) v ^= (T)(2 * top - 1);
    put_value<T>(w, v, r);
}
// 16 group codes at rung >= 1, values already stepped: total length / emission
template <typename T> __device__ __forceinline__ uint32_t group_len(const T (&v)[16], uint32_t rung) {
    const T top = (T)((T)1 << rung), half = (T)(top >> 1);
    uint32_t n = 16 * rung;
#pragma unroll
    for (uint32_t i = 0; i < 16; i++) {
        T x = v[i];
        if (rung < 8 && (x == top || x == (T)(top - 1))) x ^= (T)(2 * top - 1);
        n += (x >= half) + (x >= top);
    }
    return n;
}
template <typename T> __device__ __forceinline__ void put_group(LdsWriter &w, const T (&v)[16], uint32_t rung) {
    const T top = (T)((T)1 << rung);
#pragma unroll
    for (uint32_t i = 0; i < 16; i++) {
        T x = v[i];
        if (rung < 8 && (x == top || x == (T)(top - 1))) x ^= (T)(2 * top - 1);
        put_value<T>(w, x, rung);
    }
}
template <uint32_t UB> __device__ __forceinline__ uint32_t sw_noflag_len(uint32_t delta) {       // switch without flag, signal for "no change"
    const uint32_t l = cs_len<UB>(delta & ((1u << UB) - 1));
    return (l == 1 ? UB + 2 : l) - 1;
}
template <uint32_t UB> __device__ __forceinline__ void put_sw_noflag(LdsWriter &w, uint32_t delta) {
    delta &= (1u << UB) - 1;
    constexpr uint32_t r = UB - 1, sig = ((((1u << UB) - 2 - (1u << r)) << 2) | 3);    // code of 2^UB-2 at rung UB-1 (long form)
    if (delta == 0) w.put(sig, UB + 1);
    else w.put(cs_code<UB>(delta) >> 1, cs_len<UB>(delta) - 1);
}
template <uint32_t UB> __device__ __forceinline__ void put_signal(LdsWriter &w) {
    constexpr uint32_t r = UB - 1, sig = ((((1u << UB) - 2 - (1u << r)) << 2) | 3);
    w.put((sig << 1) | 1, UB + 2);
}

// Everything pass 0 and pass 1 agree on for one unit (used > 1)
template <typename T> struct BestUnit {
    T cf;                   // common factor (>= 1)
    uint32_t szN;           // plain coding size
    uint32_t szBase;        // cf coding: signal + switch + same/diff flag + divided group
    uint32_t szCf;          // cf coding: extra bits when the factor has to be written
    uint32_t idx;           // index coding size, 0xffffffff if more than 8 distinct values
    uint32_t trung;
    bool writer;            // overwrites pcf with cf-2
};

template <typename T>
__device__ __forceinline__ void best_analyse(const T (&g)[16], uint32_t rung, uint32_t oldrung, BestUnit<T> &u) {
    constexpr uint32_t UB = UBits<T>::v, UMASK = (1u << UB) - 1;
    u.cf = gcf_t<T>(g);
    u.szN = u.szBase = u.szCf = 0; u.trung = 0;
    if (u.cf >= 2) {
        T d[16], usedd = 0;
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) { d[i] = mdiv_t<T>(g[i], u.cf); usedd |= d[i]; }
        const T cfm = (T)(u.cf - 2);
        const uint32_t trung = topbit_t<T>(usedd), cfrung = topbit_t<T>(cfm);
        u.trung = trung;
        uint32_t grp = 16;
        if (trung) { apply_step<T>(d, trung); grp = group_len<T>(d, trung); }
        u.szBase = (UB + 2) + sw_noflag_len<UB>(trung - oldrung) + 1 + grp;
        if (trung >= cfrung && (trung < cfrung + UB || cfrung == 0)) u.szCf = 1 + vlen_t<T>(cfm, trung);
        else u.szCf = cs_len<UB>((cfrung - trung) & UMASK) + vlen_t<T>((T)(cfm ^ (T)((T)1 << cfrung)), cfrung - 1);
    } else {
        T v[16];
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) v[i] = g[i];
        apply_step<T>(v, rung);
        u.szN = cs_len<UB>((rung - oldrung) & UMASK) + group_len<T>(v, rung);
    }
    // index coding (QB3encode.h:557-613)
    u.idx = 0xffffffffu;
    // (first count the distinct values with plain comparisons, in registers and the same in every lane: more than 8
    // means no index coding, and the search below -- small arrays indexed at run time, divergent -- is skipped)
    uint32_t distinct = 0;
    if (rung > 3 && rung < 63) {
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) {
            bool seen = false;
#pragma unroll
            for (uint32_t j = 0; j < i; j++) seen = seen || g[j] == g[i];
            distinct += !seen;
        }
    }
    if (rung > 3 && rung < 63 && distinct <= 8) {
        T val[8]; uint32_t cnt[8], n = 0;
        bool fits = true;
#pragma unroll 1
        for (uint32_t i = 0; i < 16 && fits; i++) {
            uint32_t j = 0;
            while (j < n && val[j] != g[i]) j++;
            if (j == n) { if (n == 8) fits = false; else { val[n] = g[i]; cnt[n++] = 1; } }
            else cnt[j]++;
        }
        if (fits) {
            // stable sort by descending count (QB3encode.h:546-554)
#pragma unroll 1
            for (uint32_t i = 1; i < n; i++)
                for (uint32_t j = i; j > 0 && cnt[j] > cnt[j - 1]; j--) {
                    const T tv = val[j]; val[j] = val[j - 1]; val[j - 1] = tv;
                    const uint32_t tc = cnt[j]; cnt[j] = cnt[j - 1]; cnt[j - 1] = tc;
                }
            uint32_t bits = (UB + 2) + sw_noflag_len<UB>(UMASK - oldrung) + sw_noflag_len<UB>(rung - oldrung);
#pragma unroll 1
            for (uint32_t j = 0; j < n; j++) bits += cnt[j] * (2 + (j >= 2) + (j >= 4)) + vlen_t<T>(val[j], rung);   // plain rung-2 index codes
            u.idx = bits;
        }
    }
    const uint32_t thr = 36 + 3 * UB + 2 * rung;
    const uint32_t szDiff = u.szBase + u.szCf;
    u.writer = u.cf >= 2 && !(szDiff >= thr && u.idx < szDiff);
}

// per-band "last writer" inclusive scan over the lanes of the workgroup (lanes are slot-major, band-minor, so the
// band's units are `bands` lanes apart): key = 0 for "no writer", else anything non-zero; doubling in LDS
__device__ __forceinline__ void last_writer_scan(uint32_t *key, uint64_t *val, uint32_t n, uint32_t bands, uint32_t mykey, uint64_t myval) {
    const uint32_t tid = threadIdx.x;
    if (tid < n) { key[tid] = mykey; val[tid] = myval; }
    __syncthreads();
    for (uint32_t d = bands; d < n; d <<= 1) {
        uint32_t k = 0; uint64_t v = 0;
        const bool take = tid < n && tid >= d && key[tid] == 0;
        if (take) { k = key[tid - d]; v = val[tid - d]; }
        __syncthreads();
        if (take && k) { key[tid] = k; val[tid] = v; }
        __syncthreads();
    }
}

template <typename T, int PASS>
__global__ void enc_best_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    constexpr uint32_t UB = UBits<T>::v, UMASK = (1u << UB) - 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    const uint32_t bands = a.g.bands, nblocks = (uint32_t)a.g.nblocks, slots = a.slots, nunits = slots * bands;
    T g[16];
    EncFront<T> f;
    const uint32_t outdw = a.slot_dw;
    enc_front<T>(a, a0, smem, PASS ? outdw : 0, f, g);
    const uint32_t c = f.c, gblk = f.gblk, rung = f.rung, chunk = f.chunk;
    const bool payload = f.payload;
    const T used = f.used;
    uint64_t *wval = (uint64_t *)(f.outbuf + ((outdw + 1) & ~1u));
    uint32_t *wkey = (uint32_t *)(wval + nunits);

    uint32_t oldrung = 0;
    BestUnit<T> u;
    u.writer = false; u.cf = 1; u.szN = u.szBase = u.szCf = 0; u.idx = 0xffffffffu; u.trung = 0;
    if (payload) {
        oldrung = (gblk == 0) ? a0.st.rung[c] : f.rungs[tid - bands];
        if (used > 1) best_analyse<T>(g, rung, oldrung, u);
    }
    // who wrote the band's factor last, up to and including each unit
    last_writer_scan(wkey, wval, nunits, bands, (payload && u.writer) ? 1u : 0u, (uint64_t)(T)(u.cf - 2));
    if (PASS == 0) {
        // chunk summary: the entry of the last payload slot of each band
        const uint32_t last = (slots - 1) * bands + tid;
        if (tid < bands) { a.cw_has[(uint64_t)chunk * bands + tid] = (uint8_t)(wkey[last] != 0); a.cw_val[(uint64_t)chunk * bands + tid] = wval[last]; }
        return;
    }
    // factor state entering this unit: previous unit of the band in the chunk, else the chunk's entry state
    T pcf = (T)a.centry[(uint64_t)chunk * bands + c];
    if (payload && tid >= bands && wkey[tid - bands]) pcf = (T)wval[tid - bands];
    __syncthreads();

    // ---- choose the coding and its length (QB3encode.h:679-713)
    uint32_t len = 0, kind = 0;     // kind: 0 low (used <= 1), 1 plain, 2 common factor, 3 index
    bool same = false;
    if (payload) {
        if (used <= 1) len = cs_len<UB>((rung - oldrung) & UMASK) + 1 + (used ? 16 : 0);
        else {
            const uint32_t thr = 36 + 3 * UB + 2 * rung;
            uint32_t size;
            if (u.cf >= 2) { same = (T)(u.cf - 2) == pcf; size = u.szBase + (same ? 0 : u.szCf); kind = 2; }
            else { size = u.szN; kind = 1; }
            if (size >= thr && u.idx < size) { size = u.idx; kind = 3; }
            len = size;
        }
    }
    uint32_t total;
    const uint32_t pos = block_exscan(len, f.wsum, &total);

    if (payload) {
        LdsWriter w;
        w.init(f.outbuf, pos);
        if (kind == 0) {
            w.put(cs_code<UB>((rung - oldrung) & UMASK), cs_len<UB>((rung - oldrung) & UMASK));
            w.put((uint32_t)used, 1);
            if (used) {
                uint32_t bits = 0;
#pragma unroll
                for (uint32_t i = 0; i < 16; i++) bits |= (uint32_t)(g[i] & 1) << i;
                w.put(bits, 16);
            }
        } else if (kind == 1) {
            w.put(cs_code<UB>((rung - oldrung) & UMASK), cs_len<UB>((rung - oldrung) & UMASK));
            apply_step<T>(g, rung);
            put_group<T>(w, g, rung);
        } else if (kind == 2) {     // cfgenc, QB3encode.h:283-361
            T d[16];
#pragma unroll
            for (uint32_t i = 0; i < 16; i++) d[i] = mdiv_t<T>(g[i], u.cf);
            const T cfm = (T)(u.cf - 2);
            const uint32_t trung = u.trung, cfrung = topbit_t<T>(cfm);
            put_signal<UB>(w);
            put_sw_noflag<UB>(w, trung - oldrung);
            if (!same) {
                w.put(1, 1);
                if (trung >= cfrung && (trung < cfrung + UB || cfrung == 0)) { w.put(0, 1); put_single<T>(w, cfm, trung); }
                else {
                    const uint32_t dl = (cfrung - trung) & UMASK;
                    w.put(cs_code<UB>(dl), cs_len<UB>(dl));         // its change flag doubles as the "own rung" marker
                    put_single<T>(w, (T)(cfm ^ (T)((T)1 << cfrung)), cfrung - 1);
                }
            } else w.put(0, 1);
            if (trung == 0) {
                uint32_t bits = 0;
#pragma unroll
                for (uint32_t i = 0; i < 16; i++) bits |= (uint32_t)(d[i] & 1) << i;
                w.put(bits, 16);
            } else { apply_step<T>(d, trung); put_group<T>(w, d, trung); }
        } else {                    // ienc, QB3encode.h:557-613
            T val[8]; uint32_t cnt[8], n = 0;
#pragma unroll 1
            for (uint32_t i = 0; i < 16; i++) {
                uint32_t j = 0;
                while (j < n && val[j] != g[i]) j++;
                if (j == n) { val[n] = g[i]; cnt[n++] = 1; } else cnt[j]++;
            }
#pragma unroll 1
            for (uint32_t i = 1; i < n; i++)
                for (uint32_t j = i; j > 0 && cnt[j] > cnt[j - 1]; j--) {
                    const T tv = val[j]; val[j] = val[j - 1]; val[j - 1] = tv;
                    const uint32_t tc = cnt[j]; cnt[j] = cnt[j - 1]; cnt[j - 1] = tc;
                }
            put_signal<UB>(w);
            put_sw_noflag<UB>(w, UMASK - oldrung);
            put_sw_noflag<UB>(w, rung - oldrung);
#pragma unroll 1
            for (uint32_t i = 0; i < 16; i++) {
                uint32_t j = 0;
                while (val[j] != g[i]) j++;
                // plain rung-2 code of j (0..7): {0,2,1,5,3,7,11,15} with lengths {2,2,3,3,4,4,4,4}
                const uint32_t code = j < 2 ? (j << 1) : j < 4 ? (((j - 2) << 2) | 1) : (((j - 4) << 2) | 3);
                w.put(code, 2 + (j >= 2) + (j >= 4));
            }
#pragma unroll 1
            for (uint32_t j = 0; j < n; j++) put_single<T>(w, val[j], rung);
        }
        w.finish();
        // coder state on leaving the image (reference QB3encode.h:718-722)
        if (gblk == nblocks - 1) {
            a.res->prev[c] = (uint64_t)f.lastv; a.res->rung[c] = rung;
            a.res->cf[c] = (uint64_t)(kind == 2 ? (T)(u.cf - 2) : pcf);     // only a kept common-factor coding moves pcf
        }
        if (a.have_idx) {
            const uint32_t seg = gblk / a.g.seg_blocks;
            if (seg * a.g.seg_blocks == gblk) {
                ((T *)a.idx.prev)[(uint64_t)seg * bands + c] = f.pv;
                ((T *)a.idx.cf)[(uint64_t)seg * bands + c] = pcf;
                a.idx.rung[(uint64_t)seg * bands + c] = (uint8_t)oldrung;
                if (c == 0) a.idx.bitpos[seg] = ((uint64_t)chunk << 32) | pos;
            }
        }
    }
    __syncthreads();
    const uint32_t nd = (total + 31) >> 5;
    uint32_t *slot = a.scratch + (uint64_t)chunk * a.slot_dw;
    for (uint32_t d = tid; d < nd; d += nthr) slot[d] = f.outbuf[d];
    if (tid == 0) a.chunk_bits[chunk] = total;
}

// Carries the last factor writer across chunks: centry[k][c] = factor state on entering chunk k.  One workgroup;
// "last non-empty" is a max-scan over (chunk index + 1).
__global__ void best_scan_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    const uint32_t bands = a.g.bands, tid = threadIdx.x;
    for (uint32_t c = 0; c < bands; c++) {
        if (tid == 0) carry = 0;
        __syncthreads();
        for (uint32_t base = 0; base < a.nchunks; base += blockDim.x) {
            const uint32_t k = base + tid;
            uint32_t x = (k < a.nchunks && a.cw_has[(uint64_t)k * bands + c]) ? k + 1 : 0;
            // inclusive max-scan within the workgroup
            const uint32_t lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
            uint32_t m = x;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(m, d, 64); if (lane >= (uint32_t)d) m = max(m, y); }
            if (lane == 63) wsum[wave] = m;
            __syncthreads();
            uint32_t before = carry;
            for (uint32_t i = 0; i < wave && i < nw; i++) before = max(before, wsum[i]);
            const uint32_t incl = max(before, m);
            // exclusive: the last writer strictly before chunk k
            const uint32_t up = __shfl_up(m, 1, 64);
            const uint32_t excl = max(before, lane ? up : 0u);
            if (k < a.nchunks) a.centry[(uint64_t)k * bands + c] = excl ? a.cw_val[(uint64_t)(excl - 1) * bands + c] : a0.st.cf[c];
            __syncthreads();
            if (tid == blockDim.x - 1) carry = incl;
            __syncthreads();
        }
    }
}

// Exclusive scan of the chunk bit counts, one workgroup per SCAN_GROUP chunks (4 per thread); the per-group sums
// are folded in by the consumers.  64-bit offsets: a 16384^2 x 3 stream exceeds 2^32 bits.
__global__ void enc_scan_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    __shared__ uint32_t wsum[16];
    const uint32_t tid = threadIdx.x, i0 = blockIdx.x * SCAN_GROUP + 4 * tid;
    uint32_t v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = (i0 + k < a.nchunks) ? a.chunk_bits[i0 + k] : 0; sum += v[k]; }
    uint32_t total;
    uint64_t off = block_exscan(sum, wsum, &total);
#pragma unroll
    for (int k = 0; k < 4; k++) if (i0 + k < a.nchunks) { a.chunk_off[i0 + k] = off; off += v[k]; }
    if (tid == 0) a.group_sum[blockIdx.x] = total;
}

// Second level: exclusive scan of the group sums in place (one workgroup); entry [ngroups] gets the total.
__global__ void enc_scan2_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    __shared__ uint64_t wsum64[16];
    __shared__ uint64_t carry;
    const uint32_t ngroups = (a.nchunks + SCAN_GROUP - 1) / SCAN_GROUP;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < ngroups; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t v = i < ngroups ? a.group_sum[i] : 0ull;
        const uint64_t ex = block_exscan_v<uint64_t>(v, wsum64);
        const uint64_t c0 = carry;
        if (i < ngroups) a.group_sum[i] = c0 + ex;
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) carry = c0 + ex + v;
        __syncthreads();
    }
    if (threadIdx.x == 0) a.group_sum[ngroups] = carry;
}

// start of chunk k in the stream, in bits (k == nchunks: the stream length)
__device__ __forceinline__ uint64_t chunk_start(const EncArgs &a, uint32_t k) {
    if (k >= a.nchunks) return a.group_sum[(a.nchunks + SCAN_GROUP - 1) / SCAN_GROUP];
    return a.group_sum[k / SCAN_GROUP] + a.chunk_off[k];
}

// Concatenate: one WAVE per chunk reads the chunk's slot, funnel-shifts it to its bit position and stores the
// dwords that lie wholly inside the chunk; the first and last shifted dword go to the seam table.
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));     // four dwords at any dword address
__global__ void __launch_bounds__(256) enc_concat_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    const uint32_t chunk = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (chunk >= a.nchunks) return;
    const uint64_t G = (uint64_t)a.out_bit0 + chunk_start(a, chunk);
    const uint32_t total = a.chunk_bits[chunk];
    const uint32_t phase = (uint32_t)(G & 31), nsrc = (total + 31) >> 5;
    const uint32_t nd = (phase + total + 31) >> 5, tailbits = (phase + total) & 31;
    const uint32_t *slot = a.scratch + (uint64_t)chunk * a.slot_dw;    // 16-byte aligned (slot_dw is a multiple of 4)
    const uint4 *slot4 = (const uint4 *)slot;
    uint32_t *gout = a.out32 + (G >> 5);
    // a lane moves four dwords per step (one 16-byte load, the next one already in flight): memory-level parallelism
    // is what this copy needs.  Output dword d = source dwords d-1, d funnel-shifted by the chunk's bit phase.
    const uint32_t ng = (nd + 3) >> 2, sh = (32 - phase) & 31;
    constexpr int NQ = 4;                                               // 16-byte loads in flight per lane
    for (uint32_t gb = 0; gb < ng; gb += 64 * NQ) {
        uint4 cur[NQ];
        uint32_t before[NQ];                                            // lane 0: the dword before its group
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const uint32_t g = gb + 64 * q + lane;
            cur[q] = make_uint4(0, 0, 0, 0); before[q] = 0;
            if (g < ng && 4 * g < nsrc) cur[q] = slot4[g];
            if (lane == 0 && g && g < ng && 4 * g - 1 < nsrc) before[q] = slot[4 * g - 1];
        }
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const uint32_t g = gb + 64 * q + lane, d = 4 * g;
            uint32_t s[4] = { cur[q].x, cur[q].y, cur[q].z, cur[q].w };
#pragma unroll
            for (int k = 0; k < 4; k++) if (d + k >= nsrc) s[k] = 0;    // the slot is only defined up to nsrc
            uint32_t prv = __shfl_up(s[3], 1, 64);
            if (lane == 0) prv = before[q];
            if (g >= ng) continue;
            uint32_t v[4];
            if (phase) {
                v[0] = __builtin_amdgcn_alignbit(s[0], prv, sh);
#pragma unroll
                for (int k = 1; k < 4; k++) v[k] = __builtin_amdgcn_alignbit(s[k], s[k - 1], sh);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = s[k];
            }
            if (d > 0 && d + 4 < nd) {                                  // no seam dword in the group
                u32x4_a4 o = { v[0], v[1], v[2], v[3] };
                *(u32x4_a4 *)(gout + d) = o;
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t dd = d + k;
                    if (dd < nd) {
                        const bool shared = (dd == 0 && phase) || (dd == nd - 1 && tailbits);
                        if (!shared) gout[dd] = v[k];
                        if (dd == 0) a.seams[2 * chunk] = v[k];
                        if (dd == nd - 1) a.seams[2 * chunk + 1] = v[k];
                    }
                }
            }
        }
    }
}

// One thread per chunk boundary: a dword that holds the end of one chunk and the start of the next is the OR of
// their edge dwords; the thread of the FIRST boundary inside a dword assembles it.  The same launch turns the
// chunk-relative index positions into stream positions and publishes the stream length.
__global__ void enc_seam_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
    if (a.have_idx)
        for (uint64_t sgi = k; sgi < a.g.nseg; sgi += nthreads) {
            const uint64_t v = a.idx.bitpos[sgi];
            a.idx.bitpos[sgi] = chunk_start(a, (uint32_t)(v >> 32)) + (v & 0xffffffffu);
        }
    if (k > a.nchunks) return;
    const uint64_t Ek = (uint64_t)a.out_bit0 + chunk_start(a, k);
    if (k == a.nchunks) a.res->total_bits = Ek - a.out_bit0;
    if ((Ek & 31) == 0) return;
    const uint64_t d = Ek >> 5;
    if (k > 0) { const uint64_t Ep = (uint64_t)a.out_bit0 + chunk_start(a, k - 1); if ((Ep >> 5) == d && (Ep & 31)) return; }
    uint32_t v = k > 0 ? a.seams[2 * (k - 1) + 1] : 0u;
    for (uint32_t j = k; j < a.nchunks; j++) {
        v |= a.seams[2 * j];
        const uint64_t En = (uint64_t)a.out_bit0 + chunk_start(a, j + 1);
        if ((En >> 5) != d || (En & 31) == 0) break;       // chunk j reaches the end of the dword
    }
    a.out32[d] = v;
}

// Writes the container header in front of every tile's stream (after enc_seam_kernel, which owns the first dword).
__global__ void write_header_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    // the stream starts at out32 + out_bit0/8; the header ends there
    // (with a coarse index chunk the prepared bytes are followed by its entries and "DT": hdr_back > hdr_len)
    uint8_t *start = (uint8_t *)a.out32 + (a.out_bit0 >> 3) - a.hdr_back;
    for (uint32_t i = threadIdx.x; i < a.hdr_len; i += blockDim.x) start[i] = a0.hdr[i];
}

// The coarse index chunk: every ix_spe-th segment entry of the (finished) index, packed little endian, then "DT"
__global__ void ix_fill_kernel(const EncArgs a) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x, B = a.g.bands, tsz = a.g.tsz;
    if (k == 0) { uint8_t *dt = a.ix_dst + (uint64_t)a.ix_K * a.ix_E; dt[0] = 'D'; dt[1] = 'T'; }
    if (k >= a.ix_K) return;
    const uint64_t s = (uint64_t)k * a.ix_spe;
    uint8_t *e = a.ix_dst + (uint64_t)k * a.ix_E;
    const uint64_t bp = a.idx.bitpos[s];
    for (uint32_t i = 0; i < 6; i++) e[i] = (uint8_t)(bp >> (8 * i));
    e += 6;
    for (uint32_t c = 0; c < B; c++) e[c] = a.idx.rung[s * B + c];
    e += B;
    const uint8_t *pv = (const uint8_t *)a.idx.prev + s * B * tsz;
    for (uint32_t i = 0; i < B * tsz; i++) e[i] = pv[i];
    if (a.g.mode == CM_BEST) {
        e += B * tsz;
        const uint8_t *cf = (const uint8_t *)a.idx.cf + s * B * tsz;
        for (uint32_t i = 0; i < B * tsz; i++) e[i] = cf[i];
    }
}

// ------------------------------------------------------------------ decode
struct DecArgs {
    Geometry g;
    const uint32_t *in32;
    uint32_t in_bit0;
    uint64_t in_bits;           // stream length in bits
    void *img;
    IndexView idx;
    uint32_t *status;
    uint32_t lane_dw;           // LDS dwords per lane (odd)
    uint32_t dpr;
    // unit-parallel kernel (dec3_kernel)
    uint32_t bpp, passes, in_cap_dw, magic_bpp, magic_dpr;
    uint32_t px_ng, px_magic_ng;    // 16-bit lane-per-block kernel: band groups (lanes) per block
    uint32_t totals_only;           // lane-per-block kernels: write the segments' per-band sums to idx.prev, no pixels
    uint32_t px_aligned;            // lane-per-block kernels: every row of every block is dword aligned (plain dword stores)
    uint64_t *stamps;               // debugging: per-wave phase time stamps (null: off), see dbg_set_stamps
    uint32_t stamps_n;
    const uint8_t *ix;              // coarse index chunk found in the container (null: none): restart points for the walk
    uint32_t ix_K, ix_blocks, ix_E;
    // batched tiles (blockIdx.y = tile): byte strides, and each tile's stream length in bits (null: in_bits for all)
    uint32_t ntiles;
    uint64_t ts_in, ts_img, ts_idx;
    const uint64_t *tile_bits;
};


__device__ __forceinline__ DecArgs dec_for_tile(DecArgs a, uint32_t t) {
    if (a.tile_bits) a.in_bits = a.tile_bits[t];
    if (t) {
        a.in32 = shift_ptr(a.in32, t * a.ts_in);
        a.img = shift_ptr((uint8_t *)a.img, t * a.ts_img);
        const uint64_t x = t * a.ts_idx;
        a.idx.bitpos = shift_ptr(a.idx.bitpos, x); a.idx.prev = shift_ptr(a.idx.prev, x); a.idx.cf = shift_ptr(a.idx.cf, x);
        a.idx.rung = shift_ptr(a.idx.rung, x); a.idx.ulen = shift_ptr(a.idx.ulen, x);
        a.status += t;
    }
    return a;
}

// LSB-first bit reader over aligned dword loads; reads past the stream end return zeros, like the
// reference's iBits::peek (bitstream.h:39-50)
template <typename PTR>
struct ReaderT {
    PTR in;
    uint64_t buf, wp, endw;
    uint32_t n;
    __device__ __forceinline__ uint32_t load(uint64_t i) const { return i < endw ? in[i] : 0u; }
    __device__ __forceinline__ void init(PTR p, uint64_t bitpos, uint64_t endbit) {
        in = p; endw = (endbit + 31) >> 5; wp = bitpos >> 5;
        const uint32_t sh = (uint32_t)(bitpos & 31);
        buf = (uint64_t)(load(wp++) >> sh); n = 32 - sh;
    }
    __device__ __forceinline__ void ensure(uint32_t k) {    // k <= 32
        if (n < k) { buf |= (uint64_t)load(wp++) << n; n += 32; }
    }
    __device__ __forceinline__ void skip(uint32_t k) { buf >>= k; n -= k; }
    __device__ __forceinline__ uint32_t get(uint32_t k) {   // k <= 32
        if (k == 0) return 0;
        ensure(k);
        const uint32_t v = (uint32_t)(buf & (0xffffffffull >> (32 - k)));
        skip(k);
        return v;
    }
    __device__ __forceinline__ uint64_t get64(uint32_t k) { // k <= 64
        const uint64_t lo = get(k < 32 ? k : 32);
        return k > 32 ? lo | ((uint64_t)get(k - 32) << 32) : lo;
    }
    __device__ __forceinline__ uint64_t position() const { return wp * 32 - n; }   // bits consumed, from `in`
};
typedef ReaderT<const uint32_t *> Reader;

// one value at rung r >= 1, not yet unswapped (reference QB3decode.h:119-129)
template <typename T, typename RD> __device__ __forceinline__ T get_value(RD &rd, uint32_t r) {
    if (sizeof(T) <= 2) {       // r + 2 <= 17 bits
        rd.ensure(r + 2);
        const uint32_t x = (uint32_t)rd.buf, half = 1u << (r - 1), top = 1u << r;
        uint32_t v, len;
        if (!(x & 1)) { v = (x & (top - 1)) >> 1; len = r; }
        else if (!(x & 2)) { v = ((x >> 2) & (half - 1)) | half; len = r + 1; }
        else { v = ((x >> 2) & (top - 1)) | top; len = r + 2; }
        rd.skip(len);
        return (T)v;
    } else {
        rd.ensure(2);
        const uint32_t x = (uint32_t)rd.buf;
        if (!(x & 1)) { rd.skip(1); return (T)rd.get64(r - 1); }
        rd.skip(2);
        if (!(x & 2)) return (T)(rd.get64(r - 1) | (1ull << (r - 1)));
        return (T)(rd.get64(r) | (1ull << r));
    }
}
template <typename T> __device__ __forceinline__ T unswap(T v, uint32_t r) {
    const T top = (T)((T)1 << r);
    return (r < 8 && (v == top || v == (T)(top - 1))) ? (T)(v ^ (T)(2 * top - 1)) : v;
}

// rung switch: returns delta in [0, 2^UB), sets signal when the unused code is met (reference QB3decode.h:97-116)
template <uint32_t UB, typename RD> __device__ __forceinline__ uint32_t get_switch_noflag(RD &rd, bool &signal) {
    constexpr uint32_t n = 1u << UB, r = UB - 1, half = 1u << (r - 1), top = 1u << r;
    rd.ensure(r + 2);
    const uint32_t x = (uint32_t)rd.buf;
    uint32_t m, len;
    if (!(x & 1)) { m = (x & (top - 1)) >> 1; len = r; }
    else if (!(x & 2)) { m = ((x >> 2) & (half - 1)) | half; len = r + 1; }
    else { m = ((x >> 2) & (top - 1)) | top; len = r + 2; }
    rd.skip(len);
    signal = (m == n - 2);
    if (signal) return 0;
    return (m & 1) ? (n - (m + 1) / 2) & (n - 1) : m / 2 + 1;
}


// 16 values at `rung` into g (mag-sign), with the step undone when STEP (reference QB3decode.h:142-290)
template <typename T, bool STEP, typename RD> __device__ __forceinline__ void get_group(RD &rd, uint32_t rung, T (&g)[16]) {
    if (rung == 0) {
        const uint32_t bits = rd.get(1) ? rd.get(16) : 0;
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) g[i] = (T)((bits >> i) & 1);
        return;
    }
    uint32_t rb = 0;
#pragma unroll
    for (uint32_t i = 0; i < 16; i++) {
        g[i] = unswap<T>(get_value<T, RD>(rd, rung), rung);
        rb |= (uint32_t)((g[i] >> rung) & 1) << i;
    }
    if (STEP && (rb & (rb + 1)) == 0) {
        const uint32_t m = __popc(rb);
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) if (i == m) g[i] ^= (T)((T)1 << rung);
    }
}

// Parse one unit.  rung / pcf are the running state of this band.  Returns false on a corrupt stream.
// MODE: CM_FTL (no step, signal is an ordinary "no change"), CM_BASE / CM_BEST (step; signal opens the
// common-factor and index forms, reference QB3decode.h:619-716).
template <typename T, int MODE, typename RD> __device__ __forceinline__ bool parse_unit(RD &rd, uint32_t &rung, T &pcf, T (&g)[16]) {
    constexpr uint32_t UB = UBits<T>::v, UMASK = (1u << UB) - 1;
    bool signal = false;
    uint32_t delta = 0;
    if (rd.get(1)) delta = get_switch_noflag<UB, RD>(rd, signal);
    if (MODE == CM_FTL || !signal) {
        rung = (rung + delta) & UMASK;
        get_group<T, MODE != CM_FTL, RD>(rd, rung, g);
        return true;
    }
    bool sig2;
    uint32_t r = (rung + get_switch_noflag<UB, RD>(rd, sig2)) & UMASK;
    if (r != UMASK) {       // common factor
        uint32_t cfrung = r;
        T cf = pcf;
        if (rd.get(1)) {
            const uint32_t own = rd.get(1);
            if (own) {
                cfrung = (r + get_switch_noflag<UB, RD>(rd, sig2)) & UMASK;
                if (cfrung == r || cfrung == 0) return false;
            }
            const uint32_t vr = cfrung - own;
            uint64_t v;
            if (vr == 0) v = rd.get(1);
            else { T t = get_value<T, RD>(rd, vr); v = (uint64_t)((vr >= 3) ? unswap<T>(t, vr) : t); }   // cf values: rungs 1,2 unswapped (QB3encode.h:144-150)
            pcf = cf = (T)(v + ((uint64_t)own << cfrung));
        }
        cf = (T)(cf + 2);
        if (r) {
            get_group<T, true, RD>(rd, r, g);
            T used = 0;
#pragma unroll
            for (uint32_t i = 0; i < 16; i++) { g[i] = mmul_t<T>(g[i], cf); used |= g[i]; }
            rung = topbit_t<T>(used);
            return !(cf > used);
        }
        const uint32_t bits = rd.get(16);
        const T v = (T)((T)((T)(cf - 1) << 1) | 1);
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) g[i] = ((bits >> i) & 1) ? v : (T)0;
        rung = topbit_t<T>(v);
        return true;
    }
    // index coding
    rung = r = (rung + get_switch_noflag<UB, RD>(rd, sig2)) & UMASK;
    if (r == 63 || r == 0) return false;
    uint64_t ix = 0;                            // 16 x 3 bit indices packed
    uint32_t maxidx = 0, ibits = 0;
#pragma unroll
    for (uint32_t i = 0; i < 16; i++) {
        rd.ensure(4);
        const uint32_t x = (uint32_t)rd.buf;
        uint32_t v, len;                         // plain rung 2 code
        if (!(x & 1)) { v = (x & 3) >> 1; len = 2; }
        else if (!(x & 2)) { v = ((x >> 2) & 1) | 2; len = 3; }
        else { v = ((x >> 2) & 3) | 4; len = 4; }
        rd.skip(len);
        ibits += len;
        ix |= (uint64_t)v << (3 * i);
        maxidx = v > maxidx ? v : maxidx;
    }
    if (ibits > 52) return false;
    T tab[8];
#pragma unroll
    for (uint32_t i = 0; i < 8; i++) {
        tab[i] = 0;
        if (i <= maxidx) { T t = get_value<T, RD>(rd, r); tab[i] = (r >= 3) ? unswap<T>(t, r) : t; }
    }
#pragma unroll
    for (uint32_t i = 0; i < 16; i++) {
        const uint32_t j = (uint32_t)(ix >> (3 * i)) & 7;
        T v = tab[0];
#pragma unroll
        for (uint32_t k = 1; k < 8; k++) v = (j == k) ? tab[k] : v;
        g[i] = v;
    }
    return true;
}

// Lane per index segment.
template <typename T, int MODE>
__global__ void dec_kernel(const DecArgs a0) {
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t bands = a.g.bands, S = a.g.seg_blocks, nbx = a.g.nbx;
    const uint64_t seg = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (seg >= a.g.nseg) return;
    // per-lane LDS: scratch block [y][x][band] then band state
    uint32_t *lane_mem = (uint32_t *)smem + (size_t)threadIdx.x * a.lane_dw;
    T *blk = (T *)lane_mem;
    T *prev = blk + 16 * bands;
    T *pcf = prev + bands;
    uint8_t *rungs = (uint8_t *)(pcf + bands);
    for (uint32_t c = 0; c < bands; c++) {
        prev[c] = ((const T *)a.idx.prev)[seg * bands + c];
        pcf[c] = (MODE == CM_BEST) ? ((const T *)a.idx.cf)[seg * bands + c] : (T)0;
        rungs[c] = a.idx.rung[seg * bands + c];
    }
    Reader rd;
    rd.init(a.in32, a.in_bit0 + a.idx.bitpos[seg], a.in_bit0 + a.in_bits);
    const uint64_t order = a.g.order;
    const uint32_t gend = (uint32_t)(((seg + 1) * S < a.g.nblocks) ? (seg + 1) * S : a.g.nblocks);
    bool ok = true;
    T g[16];
    for (uint32_t gb = (uint32_t)(seg * S); gb < gend && ok; gb++) {
        for (uint32_t c = 0; c < bands; c++) {
            uint32_t rung = rungs[c];
            T cf = pcf[c];
            ok = parse_unit<T, MODE, Reader>(rd, rung, cf, g) && ok;
            rungs[c] = (uint8_t)rung;
            pcf[c] = cf;
            T prv = prev[c];
#pragma unroll
            for (uint32_t i = 0; i < 16; i++) {
                const uint32_t nib = curve_nib(order, i);
                prv = (T)(prv + smag_t<T>(g[i]));
                blk[nib * bands + c] = prv;
            }
            prev[c] = prv;
        }
        // add the core band back, sequentially in place like the strip epilogue (reference QB3decode.h:560-567)
        for (uint32_t c = 0; c < bands; c++) {
            const uint32_t cb = a0.g.cband[c];
            if (cb != c)
                for (uint32_t i = 0; i < 16; i++) blk[i * bands + c] = (T)(blk[i * bands + c] + blk[i * bands + cb]);
        }
        const uint32_t by = gb / nbx, bx = gb - by * nbx;
        const uint32_t x0 = (4 * bx + 4 > a.g.w) ? a.g.w - 4 : 4 * bx;
        const uint32_t y0 = (4 * by + 4 > a.g.h) ? a.g.h - 4 : 4 * by;
        for (uint32_t y = 0; y < 4; y++) {
            uint8_t *dst = (uint8_t *)a.img + ((uint64_t)(y0 + y) * a.g.stride + (uint64_t)x0 * bands) * sizeof(T);
            const uint32_t *srow = lane_mem + y * a.dpr;
            if (((uintptr_t)dst & 3) == 0)
                for (uint32_t d = 0; d < a.dpr; d++) ((uint32_t *)dst)[d] = srow[d];
            else
                for (uint32_t d = 0; d < 4 * a.dpr; d++) dst[d] = ((const uint8_t *)srow)[d];
        }
    }
    if (!ok) atomicOr(a.status, 1u);
    if (seg == a.g.nseg - 1) {
        // reference: fails when more than 7 bits are left (QB3decode.h:411,569,740); also flag overruns
        const uint64_t used = rd.position() - a.in_bit0;
        if (used > a.in_bits) atomicOr(a.status, 4u);
        else if (a.in_bits - used > 7) atomicOr(a.status, 2u);
    }
}

// ---- unit-parallel decode (FTL / BASE, band maps whose core bands are themselves core) ------------------
// One workgroup per index segment (= NB blocks).  Nothing in it is serial: the index carries the bit length of
// every unit, so
//   positions   exclusive scan of block lengths, plus the unit lengths inside the block
//   rungs       each lane reads its own rung-switch code; the rung is the entry rung of the band plus the
//               per-band scan of the switch deltas (mod 2^UB)
//   values      lane per unit decodes its 16 codes, undoes step and mag-sign; the value entering the unit is
//               the band's entry value plus the per-band scan of the unit totals
// Lanes are ordered band-major inside a pass (lane = band*BPP + block) so that a per-band scan is a plain
// workgroup scan minus its value at the band's first lane.  The compressed range is staged in LDS with
// coalesced loads, pixels are assembled in an LDS tile laid out like the image and stored as coalesced dwords.
// same, with ONE barrier: the scratch must not be rewritten before the caller's next barrier (use distinct areas)
template <typename V>
__device__ __forceinline__ V block_exscan_1b(V v, V *wsum) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    V x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        V y = __shfl_up(x, d, 64);
        if (lane >= (uint32_t)d) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    V base = 0;
    for (uint32_t i = 0; i < wave; i++) base += wsum[i];
    return (V)(base + x - v);
}

// reads the rung-switch code at bit `pos`: returns the delta (mod 2^UB), sets *gpos to the first value code
template <typename T, typename PTR>
__device__ __forceinline__ uint32_t dec3_switch(PTR src, uint32_t endw, uint32_t pos, uint32_t *gpos, bool *signal) {
    constexpr uint32_t UB = UBits<T>::v;
    ReaderT<PTR> rd;
    rd.in = src; rd.endw = endw; rd.wp = pos >> 5;
    const uint32_t sh = pos & 31;
    rd.buf = (uint64_t)(rd.load(rd.wp++) >> sh); rd.n = 32 - sh;
    uint32_t delta = 0;
    *signal = false;
    if (rd.get(1)) delta = get_switch_noflag<UB, ReaderT<PTR>>(rd, *signal);
    *gpos = (uint32_t)rd.position();
    return delta;
}

// decodes the 16 values at bit `gpos`; run[i] = sum of the first i+1 deltas in curve order.
// Rungs 1..7 go through the LDS table (one read per value, three values per refill of the bit buffer).
template <typename T, bool STEP, typename PTR>
__device__ __forceinline__ void dec3_group(PTR src, uint32_t endw, uint32_t gpos, uint32_t rung, const uint16_t *dtab, T (&run)[16]) {
    ReaderT<PTR> rd;
    rd.in = src; rd.endw = endw; rd.wp = gpos >> 5;
    const uint32_t sh = gpos & 31;
    rd.buf = (uint64_t)(rd.load(rd.wp++) >> sh); rd.n = 32 - sh;
    if (rung >= 1 && rung < 8) {
        const uint16_t *tab = dtab + dec_tab_off(rung);
        const uint32_t mask = (4u << rung) - 1;
        uint32_t rb = 0;
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) {
            if (i % 3 == 0) rd.ensure(32);          // three codes are at most 27 bits
            const uint32_t x = (uint32_t)rd.buf & mask;
            const uint32_t e = tab[x];              // value: off the critical path, the reads pipeline
            rd.skip(rung + (x & 1) + ((x & 3) == 3));   // length from the two flag bits alone (QB3decode.h:119-129)
            run[i] = (T)(e & 0xfff);
            rb |= ((e >> rung) & 1) << i;
        }
        if (STEP && (rb & (rb + 1)) == 0) {         // undo the step (reference QB3decode.h:285-289)
            const uint32_t m = __popc(rb);
#pragma unroll
            for (uint32_t i = 0; i < 16; i++) if (i == m) run[i] ^= (T)((T)1 << rung);
        }
    } else
        get_group<T, STEP, ReaderT<PTR>>(rd, rung, run);
    T acc = 0;
#pragma unroll
    for (uint32_t i = 0; i < 16; i++) { acc = (T)(acc + smag_t<T>(run[i])); run[i] = acc; }
}

template <typename T, bool STEP>
__global__ void dec3_kernel(const DecArgs a0) {
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    constexpr uint32_t UB = UBits<T>::v, UMASK = (1u << UB) - 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    const uint32_t bands = a.g.bands, NB = a.g.seg_blocks, dpr = a.dpr, nbx = a.g.nbx, BPP = a.bpp;
    const uint64_t seg = blockIdx.x;
    const uint32_t g0 = (uint32_t)(seg * NB);
    const uint32_t nblocks = (uint32_t)a.g.nblocks;
    const uint32_t nb_here = (nblocks - g0 < NB) ? nblocks - g0 : NB;
    const uint64_t stride = a.g.stride;

    // LDS carve (8-byte aligned pieces)
    uint64_t *slot_base = (uint64_t *)smem;
    uint64_t *wsum = slot_base + NB;                   // 16: scan scratch
    uint64_t *cprev = wsum + 16;                       // MAXBANDS: value entering the pass, per band
    uint64_t *ebaseT = cprev + MAXBANDS;               // MAXBANDS: scan value at the band's first lane (totals)
    uint32_t *crung = (uint32_t *)(ebaseT + MAXBANDS); // MAXBANDS
    uint32_t *ebase = crung + MAXBANDS;                // MAXBANDS (deltas)
    uint32_t *bpos = ebase + MAXBANDS;                 // BPP (+pad)
    uint32_t *stage = bpos + ((BPP + 1) & ~1u);
    uint32_t *tile = stage + a.in_cap_dw;
    uint16_t *ulen_s = (uint16_t *)(tile + 4 * NB * dpr);   // BPP*bands (padded to 8 bytes)
    uint16_t *dtab = ulen_s + ((BPP * bands + 3) & ~3u);    // DEC_TAB_SIZE + pad
    fill_dec_tab(dtab);

    // the compressed range of this segment, in bits from a.in32
    const uint64_t P0 = a.idx.bitpos[seg];
    const uint64_t P1 = (seg + 1 < a.g.nseg) ? a.idx.bitpos[seg + 1] : a.in_bits;
    const uint64_t w0 = (a.in_bit0 + P0) >> 5;
    const uint64_t endw_abs = (a.in_bit0 + a.in_bits + 31) >> 5;
    const uint64_t ndw64 = ((a.in_bit0 + P1 + 31) >> 5) - w0 + 2;
    const bool staged = ndw64 <= a.in_cap_dw;          // workgroup uniform
    const uint32_t ndw = (uint32_t)ndw64;
    if (staged)
        for (uint32_t base = 0; base < ndw; base += 4 * nthr) {        // four loads in flight per thread, then four LDS stores
            uint32_t sw[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { const uint32_t i = base + tid + q * nthr; sw[q] = (i < ndw && w0 + i < endw_abs) ? a.in32[w0 + i] : 0u; }
#pragma unroll
            for (int q = 0; q < 4; q++) { const uint32_t i = base + tid + q * nthr; if (i < ndw) stage[i] = sw[q]; }
        }
    const uint32_t endw_g = (uint32_t)((endw_abs - w0 < 0xffffffffull) ? endw_abs - w0 : 0xffffffffull);
    for (uint32_t sl = tid; sl < nb_here; sl += nthr) {
        const uint32_t g = g0 + sl, by = g / nbx, bx = g - by * nbx;
        const uint32_t x0 = (4 * bx + 4 > a.g.w) ? a.g.w - 4 : 4 * bx;
        const uint32_t y0 = (4 * by + 4 > a.g.h) ? a.g.h - 4 : 4 * by;
        slot_base[sl] = (uint64_t)y0 * stride + (uint64_t)x0 * bands;
    }
    if (tid < bands) {
        cprev[tid] = (uint64_t)((const T *)a.idx.prev)[seg * bands + tid];
        crung[tid] = a.idx.rung[seg * bands + tid];
    }
    uint32_t cpos = (uint32_t)(a.in_bit0 + P0 - 32 * w0);     // bit position of the pass, relative to word w0
    const uint32_t c = fastdiv(tid, BPP, a.magic_bpp), b = tid - c * BPP;
    const uint32_t cb = a0.g.cband[c < MAXBANDS ? c : 0];
    const uint64_t order = a.g.order;
    T *tt = (T *)tile;
    const uint32_t rowel = NB * 4 * bands;       // tile elements per pixel row
    bool bad = false;
    __syncthreads();

    for (uint32_t p = 0; p < a.passes; p++) {
        const uint32_t pb0 = p * BPP;
        const uint32_t nbp = pb0 >= nb_here ? 0 : ((nb_here - pb0 < BPP) ? nb_here - pb0 : BPP);
        // unit lengths of this pass (contiguous in the table)
        const uint64_t ubase = ((uint64_t)g0 + pb0) * bands;
        if (a.g.ulen_sz == 1) for (uint32_t i = tid; i < nbp * bands; i += nthr) ulen_s[i] = ((const uint8_t *)a.idx.ulen)[ubase + i];
        else for (uint32_t i = tid; i < nbp * bands; i += nthr) ulen_s[i] = ((const uint16_t *)a.idx.ulen)[ubase + i];
        __syncthreads();
        uint32_t blen = 0;
        if (tid < nbp) for (uint32_t k = 0; k < bands; k++) blen += ulen_s[tid * bands + k];
        const uint32_t bex = (uint32_t)block_exscan_v<uint64_t>(blen, wsum);
        if (tid < nbp) bpos[tid] = cpos + bex;
        if (tid == nthr - 1) wsum[15] = bex + blen;        // pass total (lane nthr-1 holds the inclusive sum)
        __syncthreads();
        const uint32_t ptotal = (uint32_t)wsum[15];
        const bool act = c < bands && b < nbp;
        const uint32_t sl = pb0 + b;
        uint32_t pos = 0, gpos = 0, delta = 0;
        if (act) {
            pos = bpos[b];
            for (uint32_t k = 0; k < c; k++) pos += ulen_s[b * bands + k];
            bool sig;
            delta = staged ? dec3_switch<T, LdsWords>((LdsWords)stage, ndw, pos, &gpos, &sig)
                           : dec3_switch<T, const uint32_t *>(a.in32 + w0, endw_g, pos, &gpos, &sig);
            if (sig && STEP) bad = true;       // common-factor / index unit in a BASE stream: not handled here
        }
        // per-band scan of the rung deltas
        const uint32_t dex = (uint32_t)block_exscan_v<uint64_t>(act ? delta : 0u, wsum);
        if (act && b == 0) ebase[c] = dex;
        __syncthreads();
        T run[16];
        T usum = 0;
        uint32_t rung = 0;
        if (act) {
            rung = (crung[c] + dex + delta - ebase[c]) & UMASK;
            if (staged) dec3_group<T, STEP, LdsWords>((LdsWords)stage, ndw, gpos, rung, dtab, run);
            else dec3_group<T, STEP, const uint32_t *>(a.in32 + w0, endw_g, gpos, rung, dtab, run);
            usum = run[15];
        }
        // per-band scan of the unit totals -> value entering each unit
        const uint64_t sex = block_exscan_v<uint64_t>(act ? (uint64_t)usum : 0ull, wsum);
        if (act && b == 0) ebaseT[c] = sex;
        __syncthreads();
        T pv = 0;
        if (act) {
            pv = (T)(cprev[c] + sex - ebaseT[c]);
            if (cb == c) {
#pragma unroll
                for (uint32_t i = 0; i < 16; i++) {
                    const uint32_t nib = curve_nib(order, i);
                    tt[sl * 4 * bands + c + ((nib >> 2) * rowel + (nib & 3) * bands)] = (T)(run[i] + pv);
                }
            }
        }
        __syncthreads();            // core bands are in the tile; every read of crung/cprev is done
        if (act) {
            if (cb != c) {
#pragma unroll
                for (uint32_t i = 0; i < 16; i++) {
                    const uint32_t nib = curve_nib(order, i);
                    const uint32_t e = sl * 4 * bands + ((nib >> 2) * rowel + (nib & 3) * bands);
                    tt[e + c] = (T)(run[i] + pv + tt[e + cb]);
                }
            }
            if (b == nbp - 1) { crung[c] = rung; cprev[c] = (uint64_t)(T)(pv + usum); }
        }
        cpos += ptotal;
        __syncthreads();
    }
    if (bad) atomicOr(a.status, 1u);
    if (tid == 0 && seg == a.g.nseg - 1) {      // reference: more than 7 unused bits at the end is a failure
        const uint64_t used = (uint64_t)cpos + 32 * w0 - a.in_bit0;
        if (used > a.in_bits) atomicOr(a.status, 4u);
        else if (a.in_bits - used > 7) atomicOr(a.status, 2u);
    }

    // ---- store the tile rows, coalesced dwords
    const uint32_t rowdw = nb_here * dpr;
    for (uint32_t r = 0; r < 4; r++)
        for (uint32_t j = tid; j < rowdw; j += nthr) {
            const uint32_t sl = fastdiv(j, dpr, a.magic_dpr), d = j - sl * dpr;
            uint8_t *dst = (uint8_t *)a.img + (slot_base[sl] + (uint64_t)r * stride) * sizeof(T) + 4 * d;
            const uint32_t v = tile[r * NB * dpr + j];
            if (((uintptr_t)dst & 3) == 0) *(uint32_t *)dst = v;
            else { dst[0] = (uint8_t)v; dst[1] = (uint8_t)(v >> 8); dst[2] = (uint8_t)(v >> 16); dst[3] = (uint8_t)(v >> 24); }
        }
}

// ---- 8-bit, 1/3/4 bands: lane per BLOCK decode, in registers (counterpart of enc_px_kernel) -------------
// WAVE per index segment (64 blocks), lane per block; the waves of a workgroup share the code table and nothing
// else, so there is one barrier and every wave hides the others' memory latency.  The kernel is bound by memory
// latency, the LDS pipe and instruction issue, not by HBM bandwidth, so everything here is about instructions and LDS
// accesses per value:
//   * the segment's bits are staged in LDS (padded with zero words: no bounds checks on the decode path) and all
//     bit positions are kept relative to LDS address 0, so a refill is  lshr, and, ds_read2_b32, v_alignbit;
//   * the code table holds the mag-sign-undone delta (and the step flag) as 32-bit entries in rung regions aligned
//     to their size, so the entry address is ONE v_and_or of the bit buffer; the code length is ONE v_bfe_u32
//     of a per-rung constant; the table itself is a compile-time constant copied from L2 with one 16-byte load;
//   * running sums are kept two to a register as 16-bit lanes: entering values and core bands are added with
//     v_pk_add_u16, bytes are gathered into pixel order with v_perm_b32 (3 per output dword);
//   * the three wave scans (bit positions, rung deltas and unit totals, the last two packed 16 bits per
//     band) use DPP row shifts/broadcasts, no LDS.
// The four rows go straight to HBM (B dwords per lane and row: 64 lanes write one contiguous run).
__device__ __forceinline__ uint32_t pk_add16(uint32_t x, uint32_t y) {       // two independent 16-bit adds (v_pk_add_u16)
    return __builtin_bit_cast(uint32_t, (u16x2_t)(__builtin_bit_cast(u16x2_t, x) + __builtin_bit_cast(u16x2_t, y)));
}
constexpr int curve_pos_of(uint64_t order, int x, int y) {      // inverse of the curve: visit index of pixel (x, y)
    for (int i = 0; i < 16; i++) if ((int)order_nib(order, i) == ((y << 2) | x)) return i;
    return 0;
}

// Decode table of the 8-bit lane-per-block kernel, built at compile time.  Region of rung r (1..7): entries
// [4<<r, 8<<r), i.e. byte offset 16<<r, aligned to its own size.  Entry: bits 0..15 the value with mag-sign undone
// (two's complement), bit 16 the top (rung) bit of the mag-sign value, bit 17 its low bit (the sign) -- the two
// flags the step needs (reference QB3decode.h:285-289).
struct PxDecTab { alignas(16) uint32_t e[1024]; };
constexpr PxDecTab make_px_dec_tab() {
    PxDecTab t{};
    for (uint32_t r = 1; r < 8; r++) {
        const uint32_t top = 1u << r, half = top >> 1;
        for (uint32_t x = 0; x < (4u << r); x++) {
            uint32_t v = 0;
            if (!(x & 1)) v = (x & (top - 1)) >> 1;
            else if (!(x & 2)) v = ((x >> 2) & (half - 1)) | half;
            else v = ((x >> 2) & (top - 1)) | top;
            if (v == top || v == top - 1) v ^= 2 * top - 1;
            const uint32_t d = ((v >> 1) ^ (0u - (v & 1u))) & 0xffffu;
            t.e[(4u << r) + x] = d | (((v >> r) & 1u) << 16) | ((v & 1u) << 17);
        }
    }
    return t;
}
__device__ const PxDecTab px_dec_tab = make_px_dec_tab();

// rung switch of an 8-bit unit at bit `pos`: delta (mod 8); *cslen = bits consumed
__device__ __forceinline__ uint32_t px_switch(uint32_t pos, uint32_t *cslen, bool *signal) {
    uint32_t x = lds_bits(pos);
    *signal = false;
    if (!(x & 1)) { *cslen = 1; return 0; }
    x >>= 1;                                            // code at rung 2 (reference QB3decode.h:97-116)
    uint32_t m, len;
    if (!(x & 1)) { m = (x & 3) >> 1; len = 2; }
    else if (!(x & 2)) { m = ((x >> 2) & 1) | 2; len = 3; }
    else { m = ((x >> 2) & 3) | 4; len = 4; }
    *cslen = 1 + len;
    if (m == 6) { *signal = true; return 0; }
    return (m & 1) ? (8 - (m + 1) / 2) & 7 : m / 2 + 1;
}

// the 16 values of an 8-bit unit whose codes start at bit `gpos`: rp[k] = running sums of values 2k, 2k+1 as two
// 16-bit lanes (low byte = the sum mod 256); returns the unit total (garbage above bit 7)
template <bool STEP>
__device__ __forceinline__ uint32_t px_group(uint32_t gpos, uint32_t rung, uint32_t (&rp)[8]) {
    uint32_t acc = 0;
    if (rung == 0) {
        const uint32_t x = lds_bits(gpos);
        const uint32_t bits = (x & 1) ? (x >> 1) & 0xffffu : 0u;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            acc -= (bits >> i) & 1u;                    // mag-sign 1 is -1
            if (i & 1) rp[i >> 1] |= acc << 16; else rp[i >> 1] = acc & 0xffffu;
        }
        return acc;
    }
    const uint32_t base = 16u << rung, m2 = base - 4;   // table region and the mask of (rung+2 bits) << 2
    const uint32_t K = rung * 0x11111111u + 0x20102010u; // code length by the low three bits, 4 bits each
    uint32_t pos = gpos, buf = 0, fl = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (i % 3 == 0) buf = lds_bits(pos);            // three codes are at most 27 bits
        const uint32_t t = buf << 2;
        const uint32_t e = *lds_at((t & m2) | base);
        const uint32_t len = __builtin_amdgcn_ubfe(K, t, 4);
        buf >>= len; pos += len;
        acc += e;
        if (STEP) fl |= ((e >> 16) & 3u) << (2 * i);
        if (i & 1) rp[i >> 1] |= acc << 16; else rp[i >> 1] = acc & 0xffffu;
    }
    if (STEP) {                                         // undo the step (reference QB3decode.h:285-289)
        const uint32_t tb = fl & 0x55555555u, u = tb | (tb << 1);
        const uint32_t m = __popc(tb);
        if ((u & (u + 1)) == 0 && m < 16) {
            // value m regains its rung bit: its delta moves by half a rung, away from zero; sums m.. follow
            const uint32_t half = base >> 5;            // 1 << (rung - 1)
            const uint32_t c16 = ((fl >> (2 * m + 1)) & 1u) ? (0u - half) & 0xffffu : half;
            const uint32_t ge = 0xffff0000u >> (16 - m);// bit i set: value i >= m   (as a 16-bit mask in the low half)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t pair = (ge >> (2 * k)) & 3u;
                rp[k] = pk_add16(rp[k], ((pair | (pair << 15)) & 0x00010001u) * c16);
            }
            acc += c16;
        }
    }
    return acc;
}

template <int B, bool RGB, uint64_t ORDER, bool STEP>
__global__ void __launch_bounds__(256) dec_px_kernel(const DecArgs a0) {
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint64_t t_start = a0.stamps ? clock64() : 0;
    constexpr int NW = (B + 1) / 2;                     // 32-bit words of a scan packed 16 bits per band
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const uint32_t NB = a.g.seg_blocks, nbx = a.g.nbx;  // NB <= 64: a WAVE owns a segment, nothing is shared but the table
    const uint64_t stride = a.g.stride;

    uint32_t *tab = (uint32_t *)smem;                   // 4 KB, at LDS address 0 (the table addressing relies on it)
    uint32_t *stage = tab + 1024 + wave * (a.in_cap_dw + 8);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)smem;
    const uint32_t stage_bit0 = 8 * (lds0 + (uint32_t)((uint8_t *)stage - smem));
    // Loads that depend on nothing but the segment number go out first -- positions, unit lengths, entering rungs and
    // values -- so that their round trips overlap the table copy and its barrier (in-kernel time stamps: a wave spends
    // 45 % of its life waiting for memory before it can start, the two dependent trips "position, then stream words").
    const uint64_t seg = (uint64_t)blockIdx.x * nwaves + wave;
    const bool live = seg < a.g.nseg;
    const uint64_t segc = live ? seg : 0;
    const uint32_t g0 = (uint32_t)(segc * NB), nblocks = (uint32_t)a.g.nblocks;
    const uint32_t nb_here = (nblocks - g0 < NB) ? nblocks - g0 : NB;
    const bool act = live && lane < nb_here;
    const uint64_t P0 = a.idx.bitpos[segc];
    const uint64_t P1 = (segc + 1 < a.g.nseg) ? a.idx.bitpos[segc + 1] : a.in_bits;
    uint32_t ul_[B], rg0[B], pv0[B], blen = 0;
    {
        const uint8_t *ul = (const uint8_t *)a.idx.ulen + ((uint64_t)g0 + lane) * B;
#pragma unroll
        for (int c = 0; c < B; c++) {
            ul_[c] = act ? ul[c] : 0u;
            rg0[c] = a.idx.rung[segc * B + c];
            pv0[c] = ((const uint8_t *)a.idx.prev)[segc * B + c];
        }
    }
    for (uint32_t i = tid; i < 256; i += blockDim.x) ((uint4 *)tab)[i] = ((const uint4 *)px_dec_tab.e)[i];
    __syncthreads();                                    // the only workgroup barrier
    if (!live) return;
    const bool stamp = a0.stamps && seg < a0.stamps_n && lane == 0;
    if (stamp) { a0.stamps[seg * 8 + 0] = t_start; a0.stamps[seg * 8 + 1] = clock64(); }
    const uint64_t w0 = (a.in_bit0 + P0) >> 5;
    const uint64_t endw_abs = (a.in_bit0 + a.in_bits + 31) >> 5;
    const uint64_t ndw64 = ((a.in_bit0 + P1 + 31) >> 5) - w0;
    // the staging area holds the longest valid segment; an index that says otherwise is not ours
    const bool fits = ndw64 <= a.in_cap_dw && lds0 == 0;
    const uint32_t ndw = fits ? (uint32_t)ndw64 : 0;
    for (uint32_t base = 0; base < ndw + 8; base += 512) {          // eight loads in flight per lane, then eight LDS stores
        uint32_t sw[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t i = base + lane + 64 * k;
            sw[k] = (i < ndw && w0 + i < endw_abs) ? a.in32[w0 + i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t i = base + lane + 64 * k;
            if (i < ndw + 8) stage[i] = sw[k];
        }
    }
#pragma unroll
    for (int c = 0; c < B; c++) blen += ul_[c];
    // the wave reads what its own lanes staged: LDS operations of a wave execute in order, the fence is for the compiler
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    if (stamp) a0.stamps[seg * 8 + 2] = clock64();
    const uint32_t limit = stage_bit0 + 32 * ndw;       // no unit starts beyond the staged bits (8 zero words follow)
    const uint32_t cpos = stage_bit0 + (uint32_t)(a.in_bit0 + P0 - 32 * w0);
    bool bad = !fits;
    const uint32_t binc = wave_iscan32(blen);           // inclusive: lane 63 holds the bits of the segment
    // rung switches of the lane's units
    uint32_t gpos[B], pos = cpos + binc - blen, dpk[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) dpk[k] = 0;
#pragma unroll
    for (int c = 0; c < B; c++) {
        pos = pos < limit ? pos : limit;
        bool sig; uint32_t csl;
        const uint32_t d = px_switch(pos, &csl, &sig);
        gpos[c] = pos + csl;
        if (act && sig && STEP) bad = true;             // common-factor / index unit: not handled here
        dpk[c >> 1] |= (act ? d : 0u) << (16 * (c & 1));
        pos += ul_[c];
    }
#pragma unroll
    for (int k = 0; k < NW; k++) dpk[k] = wave_iscan32(dpk[k]);                 // inclusive, 16 bits per band
    if (stamp) a0.stamps[seg * 8 + 3] = clock64();
    // decode the units; running sums in curve order, two 16-bit lanes per register
    uint32_t rp[B][8], spk[NW], sinc[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) spk[k] = 0;
#pragma unroll
    for (int c = 0; c < B; c++) {
        const uint32_t rung = (rg0[c] + ((dpk[c >> 1] >> (16 * (c & 1))) & 0xffffu)) & 7u;
        const uint32_t tot = px_group<STEP>(gpos[c], rung, rp[c]) & 0xffu;
        spk[c >> 1] |= (act ? tot : 0u) << (16 * (c & 1));
    }
#pragma unroll
    for (int k = 0; k < NW; k++) sinc[k] = wave_iscan32(spk[k]);
    if (stamp) a0.stamps[seg * 8 + 4] = clock64();
    if (a.totals_only) { // foreign stream, first pass: leave the segment's per-band sums where the entering values go
        if (lane == 63)
#pragma unroll
            for (int c = 0; c < B; c++) ((uint8_t *)a.idx.prev)[seg * B + c] = (uint8_t)(sinc[c >> 1] >> (16 * (c & 1)));
        if (bad) atomicOr(a.status, fits ? 1u : 8u);
        return;
    }
    if (act) {
        // entering value, then the core band (reference QB3decode.h:560-567)
#pragma unroll
        for (int c = 0; c < B; c++) {
            const uint32_t pv = pv0[c] + (((sinc[c >> 1] - spk[c >> 1]) >> (16 * (c & 1))) & 0xffffu);
#pragma unroll
            for (int k = 0; k < 8; k++) rp[c][k] = pk_add16(rp[c][k], (pv & 0xffu) * 0x00010001u);
        }
#pragma unroll
        for (int c = 0; c < B; c++) {
            const int cb = core_of<B, RGB>(c);
            if (cb != c)
#pragma unroll
                for (int k = 0; k < 8; k++) rp[c][k] = pk_add16(rp[c][k], rp[cb][k]);
        }
        // curve order, band planar -> pixel order, band interleaved; store the four rows
        const uint32_t g = g0 + lane, by = g / nbx, bx = g - by * nbx;
        const uint32_t x0 = (4 * bx + 4 > a.g.w) ? a.g.w - 4 : 4 * bx;     // last column / row is shifted, not padded
        const uint32_t y0 = (4 * by + 4 > a.g.h) ? a.g.h - 4 : 4 * by;
        uint8_t *p0 = (uint8_t *)a.img + (uint64_t)y0 * stride + (uint64_t)x0 * B;
#pragma unroll
        for (int y = 0; y < 4; y++) {
            uint32_t ow[B];
#pragma unroll
            for (int k = 0; k < B; k++) {
                // byte j of output dword k is band (4k+j)%B of pixel x = (4k+j)/B: low byte of a 16-bit lane
                uint32_t half2[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int b0 = 4 * k + 2 * h, b1 = b0 + 1;
                    const int i0 = curve_pos_of(ORDER, b0 / B, y), i1 = curve_pos_of(ORDER, b1 / B, y);
                    // v_perm_b32: selector bytes 0..3 pick from the second operand, 4..7 from the first
                    half2[h] = __builtin_amdgcn_perm(rp[b1 % B][i1 >> 1], rp[b0 % B][i0 >> 1],
                                                     (uint32_t)((4 + 2 * (i1 & 1)) << 8 | (2 * (i0 & 1))));
                }
                ow[k] = __builtin_amdgcn_perm(half2[1], half2[0], 0x05040100u);
            }
            uint8_t *row = p0 + (uint64_t)y * stride;
            const uint32_t al = a.px_aligned ? 0u : (uint32_t)(uintptr_t)row & 3;     // px_aligned: wave uniform
            if (al == 0) {
#pragma unroll
                for (int k = 0; k < B; k++) ((uint32_t *)row)[k] = ow[k];
            } else {        // unaligned row: head bytes, the aligned dwords inside it, tail bytes -- only the row's own 4*B bytes
                const uint32_t head = 4 - al, sh = 8 * head;            // bytes before the first aligned dword
#pragma unroll
                for (uint32_t t = 0; t < 3; t++) if (t < head) row[t] = (uint8_t)(ow[0] >> (8 * t));
                uint32_t *mid = (uint32_t *)(row + head);
#pragma unroll
                for (int k = 0; k + 1 < B; k++) mid[k] = __builtin_amdgcn_alignbit(ow[k + 1], ow[k], sh);
                uint8_t *tail = row + head + 4 * (B - 1);               // the last `al` bytes
                const uint32_t last = ow[B - 1] >> sh;
#pragma unroll
                for (uint32_t t = 0; t < 3; t++) if (t < al) tail[t] = (uint8_t)(last >> (8 * t));
            }
        }
    }
    if (stamp) a0.stamps[seg * 8 + 5] = clock64();
    if (bad) atomicOr(a.status, fits ? 1u : 8u);
    if (lane == 63 && seg == a.g.nseg - 1 && fits) {    // reference: more than 7 unused bits at the end is a failure
        const uint64_t used = (uint64_t)(cpos + binc - stage_bit0) + 32 * w0 - a.in_bit0;
        if (used > a.in_bits) atomicOr(a.status, 4u);
        else if (a.in_bits - used > 7) atomicOr(a.status, 2u);
    }
}
// ---- 16-bit: wave per index segment, lane per (block, band group) -- counterpart of enc_px16_kernel -------
// Same organisation as dec_px_kernel; a lane decodes the BG <= 4 units of its band group.  Rungs up to 7 go through
// the same table (values below 256), higher rungs decode by the code rule from a 64-bit buffer (three codes of at
// most 17 bits per refill).  Lane = block * NG + group, i.e. stream order, so bit positions are one DPP scan; the
// per-band scans (rung deltas, unit totals) run over the lanes of one group: DPP when NG = 1, a strided shuffle
// scan otherwise.
__device__ __forceinline__ uint32_t px16_switch(uint32_t pos, uint32_t *cslen, bool *signal) {
    uint32_t x = lds_bits(pos);
    *signal = false;
    if (!(x & 1)) { *cslen = 1; return 0; }
    x >>= 1;                                            // code at rung 3 (reference QB3decode.h:97-116)
    uint32_t m, len;
    if (!(x & 1)) { m = (x & 7) >> 1; len = 3; }
    else if (!(x & 2)) { m = ((x >> 2) & 3) | 4; len = 4; }
    else { m = ((x >> 2) & 7) | 8; len = 5; }
    *cslen = 1 + len;
    if (m == 14) { *signal = true; return 0; }
    return (m & 1) ? (16 - (m + 1) / 2) & 15 : m / 2 + 1;
}

// 16 values of a 16-bit unit at bit `gpos`: rp[k] = running sums of values 2k, 2k+1 (16-bit lanes); returns the total
template <bool STEP>
__device__ __forceinline__ uint32_t px16_group(uint32_t gpos, uint32_t rung, uint32_t (&rp)[8]) {
    if (rung < 8) return px_group<STEP>(gpos, rung, rp);     // values below 256: the table path of the 8-bit kernel
    const uint32_t top = 1u << rung, half = top >> 1;
    uint32_t pos = gpos, acc = 0, fl = 0;
    uint64_t buf = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (i % 3 == 0) {                               // three codes are at most 51 bits
            LdsWords p = lds_at((pos >> 3) & ~3u);
            const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
            buf = ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, pos) << 32) | __builtin_amdgcn_alignbit(d1, d0, pos);
        }
        const uint32_t x = (uint32_t)buf;
        const bool c1 = x & 1, c2 = (x & 3) == 3;
        const uint32_t len = rung + c1 + c2;
        const uint32_t v = c2 ? (((x >> 2) & (top - 1)) | top) : c1 ? (((x >> 2) & (half - 1)) | half) : ((x & (top - 1)) >> 1);
        buf >>= len; pos += len;
        acc += (v >> 1) ^ (0u - (v & 1u));              // undo mag-sign, accumulate (mod 2^16 in the packed lanes)
        if (STEP) fl |= ((uint32_t)c2 | ((v & 1u) << 1)) << (2 * i);
        if (i & 1) rp[i >> 1] |= acc << 16; else rp[i >> 1] = acc & 0xffffu;
    }
    if (STEP) {                                         // undo the step (reference QB3decode.h:285-289), as in px_group
        const uint32_t tb = fl & 0x55555555u, u = tb | (tb << 1);
        const uint32_t m = __popc(tb);
        if ((u & (u + 1)) == 0 && m < 16) {
            const uint32_t c16 = ((fl >> (2 * m + 1)) & 1u) ? (0u - half) & 0xffffu : half;
            const uint32_t ge = 0xffff0000u >> (16 - m);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t pair = (ge >> (2 * k)) & 3u;
                rp[k] = pk_add16(rp[k], ((pair | (pair << 15)) & 0x00010001u) * c16);
            }
            acc += c16;
        }
    }
    return acc;
}

// inclusive scan over the lanes of the same band group (stride NG), NW words per lane
template <int NW>
__device__ __forceinline__ void group_iscan(uint32_t (&v)[NW], uint32_t NG) {
    if (NG == 1) {
#pragma unroll
        for (int k = 0; k < NW; k++) v[k] = wave_iscan32(v[k]);
        return;
    }
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t d = NG; d < 64; d <<= 1) {
#pragma unroll
        for (int k = 0; k < NW; k++) {
            const uint32_t y = __shfl_up(v[k], d, 64);
            if (lane >= d) v[k] += y;
        }
    }
}

template <int BG, bool RGB, uint64_t ORDER, bool STEP>
__global__ void __launch_bounds__(256) dec_px16_kernel(const DecArgs a0) {
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int NW = (BG + 1) / 2;                    // 32-bit words of a scan packed 16 bits per band
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const uint32_t NB = a.g.seg_blocks, nbx = a.g.nbx, B = a.g.bands, NG = a.px_ng;    // NB * NG <= 64
    const uint64_t stride = a.g.stride;                 // in values
    const uint32_t slot = fastdiv(lane, NG, a.px_magic_ng), grp = lane - slot * NG, band0 = grp * BG;

    uint32_t *tab = (uint32_t *)smem;                   // 4 KB, at LDS address 0 (the table addressing relies on it)
    uint32_t *stage = tab + 1024 + wave * (a.in_cap_dw + 16);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)smem;
    const uint32_t stage_bit0 = 8 * (lds0 + (uint32_t)((uint8_t *)stage - smem));
    // loads that depend on nothing but the segment number go out first; their round trips overlap the table copy and
    // its barrier (see dec_px_kernel)
    const uint64_t seg = (uint64_t)blockIdx.x * nwaves + wave;
    const bool live = seg < a.g.nseg;
    const uint64_t segc = live ? seg : 0;
    const uint32_t g0 = (uint32_t)(segc * NB), nblocks = (uint32_t)a.g.nblocks;
    const uint32_t nb_here = (nblocks - g0 < NB) ? nblocks - g0 : NB;
    const bool act = live && slot < nb_here;
    const uint64_t P0 = a.idx.bitpos[segc];
    const uint64_t P1 = (segc + 1 < a.g.nseg) ? a.idx.bitpos[segc + 1] : a.in_bits;
    uint32_t ul_[BG], rg0[BG], pv0[BG], blen = 0;
    {
        const uint16_t *ul = (const uint16_t *)a.idx.ulen + ((uint64_t)g0 + slot) * B + band0;
#pragma unroll
        for (int c = 0; c < BG; c++) {
            ul_[c] = act ? ul[c] : 0u;
            rg0[c] = a.idx.rung[segc * B + band0 + c];
            pv0[c] = ((const uint16_t *)a.idx.prev)[segc * B + band0 + c];
        }
    }
    for (uint32_t i = tid; i < 256; i += blockDim.x) ((uint4 *)tab)[i] = ((const uint4 *)px_dec_tab.e)[i];
    __syncthreads();                                    // the only workgroup barrier
    if (!live) return;
    const uint64_t w0 = (a.in_bit0 + P0) >> 5;
    const uint64_t endw_abs = (a.in_bit0 + a.in_bits + 31) >> 5;
    const uint64_t ndw64 = ((a.in_bit0 + P1 + 31) >> 5) - w0;
    const bool fits = ndw64 <= a.in_cap_dw && lds0 == 0;
    const uint32_t ndw = fits ? (uint32_t)ndw64 : 0;
    for (uint32_t base = 0; base < ndw + 16; base += 512) {         // eight loads in flight per lane, then eight LDS stores; 16 zero words follow
        uint32_t sw[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t i = base + lane + 64 * q;
            sw[q] = (i < ndw && w0 + i < endw_abs) ? a.in32[w0 + i] : 0u;
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t i = base + lane + 64 * q;
            if (i < ndw + 16) stage[i] = sw[q];
        }
    }
#pragma unroll
    for (int c = 0; c < BG; c++) blen += ul_[c];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const uint32_t limit = stage_bit0 + 32 * ndw;
    const uint32_t cpos = stage_bit0 + (uint32_t)(a.in_bit0 + P0 - 32 * w0);
    bool bad = !fits;
    const uint32_t binc = wave_iscan32(blen);           // lanes are in stream order
    uint32_t gpos[BG], pos = cpos + binc - blen, dpk[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) dpk[k] = 0;
#pragma unroll
    for (int c = 0; c < BG; c++) {
        pos = pos < limit ? pos : limit;
        bool sig; uint32_t csl;
        const uint32_t d = px16_switch(pos, &csl, &sig);
        gpos[c] = pos + csl;
        if (act && sig && STEP) bad = true;             // common-factor / index unit: not handled here
        dpk[c >> 1] |= (act ? d : 0u) << (16 * (c & 1));
        pos += ul_[c];
    }
    group_iscan<NW>(dpk, NG);                           // inclusive, 16 bits per band
    uint32_t rp[BG][8], spk[NW], sinc[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) spk[k] = 0;
#pragma unroll
    for (int c = 0; c < BG; c++) {
        const uint32_t rung = (rg0[c] + ((dpk[c >> 1] >> (16 * (c & 1))) & 0xffffu)) & 15u;
        const uint32_t tot = px16_group<STEP>(gpos[c], rung, rp[c]) & 0xffffu;
        spk[c >> 1] |= (act ? tot : 0u) << (16 * (c & 1));
    }
    {   // per-band scan of the unit totals modulo 2^16: the two halves of a word must not carry into each other
        uint32_t lo[NW], hi[NW];
#pragma unroll
        for (int k = 0; k < NW; k++) { lo[k] = spk[k] & 0xffffu; hi[k] = spk[k] >> 16; }
        group_iscan<NW>(lo, NG);
        group_iscan<NW>(hi, NG);
#pragma unroll
        for (int k = 0; k < NW; k++) sinc[k] = (lo[k] & 0xffffu) | (hi[k] << 16);
    }
    if (a.totals_only) { // foreign stream, first pass: the last lane of every band group holds the group's sums
        if (lane >= 64 - NG)
#pragma unroll
            for (int c = 0; c < BG; c++) ((uint16_t *)a.idx.prev)[seg * B + band0 + c] = (uint16_t)(sinc[c >> 1] >> (16 * (c & 1)));
        if (bad) atomicOr(a.status, fits ? 1u : 8u);
        return;
    }
    if (act) {
#pragma unroll
        for (int c = 0; c < BG; c++) {
            const uint32_t excl = ((sinc[c >> 1] >> (16 * (c & 1))) - (spk[c >> 1] >> (16 * (c & 1)))) & 0xffffu;
            const uint32_t pv = (pv0[c] + excl) & 0xffffu;
#pragma unroll
            for (int k = 0; k < 8; k++) rp[c][k] = pk_add16(rp[c][k], pv * 0x00010001u);
        }
#pragma unroll
        for (int c = 0; c < BG; c++) {
            const int cb = core_of<BG, RGB>(c);
            if (cb != c)        // the R-G, G, B-G map applies to the first three bands of the image: group 0 only
#pragma unroll
                for (int k = 0; k < 8; k++) rp[c][k] = pk_add16(rp[c][k], grp == 0 ? rp[cb][k] : 0u);
        }
        const uint32_t g = g0 + slot, by = g / nbx, bx = g - by * nbx;
        const uint32_t x0 = (4 * bx + 4 > a.g.w) ? a.g.w - 4 : 4 * bx;     // last column / row is shifted, not padded
        const uint32_t y0 = (4 * by + 4 > a.g.h) ? a.g.h - 4 : 4 * by;
        uint16_t *p0 = (uint16_t *)a.img + (uint64_t)y0 * stride + (uint64_t)x0 * B + band0;
        // N dwords to a halfword address: aligned dwords when it is dword aligned, else a head halfword, the aligned dwords
        // inside and a tail halfword -- never a byte outside the N dwords' own place
        auto store_dw = [&](uint16_t *p, const uint32_t *src, auto nconst) {
            constexpr int N = decltype(nconst)::value;
            if (a.px_aligned || !((uintptr_t)p & 2)) {
#pragma unroll
                for (int t = 0; t < N; t++) ((uint32_t *)p)[t] = src[t];
            } else {
                p[0] = (uint16_t)src[0];
                uint32_t *mid = (uint32_t *)(p + 1);
#pragma unroll
                for (int t = 0; t + 1 < N; t++) mid[t] = __builtin_amdgcn_alignbit(src[t + 1], src[t], 16);
                p[2 * N - 1] = (uint16_t)(src[N - 1] >> 16);
            }
        };
#pragma unroll
        for (int y = 0; y < 4; y++) {
            uint16_t *rowp = p0 + (uint64_t)y * stride;
            uint32_t ow[2 * BG];
#pragma unroll
            for (int j = 0; j < 2 * BG; j++) {          // halfwords 2j, 2j+1 of the lane's row: band h % BG of pixel h / BG
                const int h0 = 2 * j, h1 = 2 * j + 1;
                const int i0 = curve_pos_of(ORDER, h0 / BG, y), i1 = curve_pos_of(ORDER, h1 / BG, y);
                const uint32_t sel = (uint32_t)(2 * (i0 & 1)) | (uint32_t)(2 * (i0 & 1) + 1) << 8 |
                                     (uint32_t)(4 + 2 * (i1 & 1)) << 16 | (uint32_t)(4 + 2 * (i1 & 1) + 1) << 24;
                ow[j] = __builtin_amdgcn_perm(rp[h1 % BG][i1 >> 1], rp[h0 % BG][i0 >> 1], sel);
            }
            if (BG % 2 == 0) {
#pragma unroll
                for (int x = 0; x < 4; x++) store_dw(rowp + (uint64_t)x * B, &ow[x * (BG / 2)], std::integral_constant<int, (BG / 2 ? BG / 2 : 1)>());
            } else
                store_dw(rowp, &ow[0], std::integral_constant<int, 2 * BG>());
        }
    }
    if (bad) atomicOr(a.status, fits ? 1u : 8u);
    if (lane == 63 && seg == a.g.nseg - 1 && fits) {
        const uint64_t used = (uint64_t)(cpos + binc - stage_bit0) + 32 * w0 - a.in_bit0;
        if (used > a.in_bits) atomicOr(a.status, 4u);
        else if (a.in_bits - used > 7) atomicOr(a.status, 2u);
    }
}

// ---- foreign streams, 8/16-bit FTL/BASE: rebuild the index without decoding values --------------------------
// The stream has no restart points, so unit positions can only be found by walking it; what CAN be parallel is
// everything else.  dec_walk_kernel walks unit LENGTHS only (a code's length is its rung plus what its low two bits
// say, reference QB3decode.h:119-129): one wave per tile, the stream staged through LDS in windows by all lanes,
// then every lane runs the same walk (uniform control flow and LDS broadcast reads; the next stream word is always
// already in a register).  It writes the per-unit lengths and each segment's bit position and rungs.  The values
// entering the segments then come from the parallel decoder itself: one pass in TOTALS mode leaves every segment's
// per-band sum in idx.prev, prev_scan_kernel turns the sums into exclusive prefixes, the normal pass follows.
template <uint32_t UB>
__global__ void __launch_bounds__(64) dec_walk_kernel(const DecArgs a0) {
    const DecArgs a = dec_for_tile(a0, blockIdx.x);
    constexpr uint32_t WIN = 4096, UMASK = (1u << UB) - 1, NRUNG = 1u << UB;
    constexpr uint32_t MAXU = UB + 2 + 16 * ((8u << (UB - 3)) + 1);    // longest unit: 149 bits (8-bit), 278 (16-bit)
    static_assert(MAXU == (UB == 3 ? 149u : 278u), "unit length bound");
    __shared__ uint32_t win[WIN + 4];
    const uint32_t lane = threadIdx.x, B = a.g.bands, NB = a.g.seg_blocks, nblocks = (uint32_t)a.g.nblocks;
    const uint64_t endw_abs = (a.in_bit0 + a.in_bits + 31) >> 5;
    uint64_t R = 0;                         // current rungs, 4 bits per band
    uint64_t P = a.in_bit0;                 // bit position, from a.in32
    uint32_t gb = 0, gb_end = nblocks, inseg = 0;
    uint64_t seg = 0;
    if (a.ix) {                             // restart point blockIdx.y of the container's coarse table
        const uint8_t *e = a.ix + (uint64_t)blockIdx.y * a.ix_E;
        uint64_t bp = 0;
        for (uint32_t i = 0; i < 6; i++) bp |= (uint64_t)e[i] << (8 * i);
        P += bp;
        for (uint32_t c = 0; c < B; c++) R |= (uint64_t)(e[6 + c] & 15u) << (4 * c);
        gb = blockIdx.y * a.ix_blocks;
        gb_end = (nblocks - gb < a.ix_blocks) ? nblocks : gb + a.ix_blocks;
        seg = gb / NB;
    }
    P = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(P >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)P);
    bool bad = false;
    while (gb < gb_end) {
        const uint64_t w0 = P >> 5;         // stage the window that starts in the word of P
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t base = 0; base < WIN + 4; base += 1024) {        // sixteen loads in flight per lane
            uint32_t sw[16];
#pragma unroll
            for (int q = 0; q < 16; q++) { const uint32_t i = base + lane + 64 * q; sw[q] = (i < WIN + 4 && w0 + i < endw_abs) ? a.in32[w0 + i] : 0u; }
#pragma unroll
            for (int q = 0; q < 16; q++) { const uint32_t i = base + lane + 64 * q; if (i < WIN + 4) win[i] = sw[q]; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // bit reader over the window: 64-bit buffer, the next word prefetched
        // (readfirstlane: the words are the same in every lane -- keep the whole walk in scalar registers, a dependent
        // scalar instruction issues twice as fast as a dependent vector one)
        uint32_t wp = (uint32_t)(P - 32 * w0) >> 5;
        const uint32_t sh = (uint32_t)P & 31;
        uint64_t buf = (uint64_t)((uint32_t)__builtin_amdgcn_readfirstlane(win[wp]) >> sh);     // the builtin returns int
        uint32_t n = 32 - sh;
        // the next word is requested one refill ahead and only moved to a scalar register when it is consumed, so the
        // LDS latency is off the walk
        uint32_t nxt_v = win[++wp];
        auto refill = [&]() {
            if (n <= 32) { buf |= (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(nxt_v) << n; n += 32; nxt_v = win[++wp]; }   // wp <= WIN + 3 by the loop bound
        };
        // walk whole blocks while the longest possible block still fits in the window
        while (gb < gb_end && 32 * wp + B * MAXU + 64 <= 32 * WIN) {
            if (inseg == 0) {
                if (lane == 0) {
                    a.idx.bitpos[seg] = 32 * (w0 + wp) - n - a.in_bit0;
                    for (uint32_t c = 0; c < B; c++) a.idx.rung[seg * B + c] = (uint8_t)((R >> (4 * c)) & 15u);
                }
                seg++;
            }
            if (++inseg == NB) inseg = 0;
            for (uint32_t c = 0; c < B; c++) {
                refill();                                   // >= 33 bits: the switch code is at most UB + 2
                uint32_t x = (uint32_t)buf, ulen;
                uint32_t rung = (uint32_t)(R >> (4 * c)) & 15u;
                if (!(x & 1)) ulen = 1;
                else {                                      // code at rung UB - 1 (reference QB3decode.h:97-116)
                    constexpr uint32_t r = UB - 1, half = 1u << (r - 1), top = 1u << r;
                    x >>= 1;
                    uint32_t m, len;
                    if (!(x & 1)) { m = (x & (top - 1)) >> 1; len = r; }
                    else if (!(x & 2)) { m = ((x >> 2) & (half - 1)) | half; len = r + 1; }
                    else { m = ((x >> 2) & (top - 1)) | top; len = r + 2; }
                    ulen = 1 + len;
                    if (m == NRUNG - 2) bad = true;         // signal: a common-factor stream, not for this walker
                    const uint32_t delta = (m & 1) ? (NRUNG - (m + 1) / 2) & UMASK : m / 2 + 1;
                    rung = (rung + delta) & UMASK;
                    R = (R & ~(15ull << (4 * c))) | ((uint64_t)rung << (4 * c));
                }
                buf >>= ulen; n -= ulen;
                if (rung == 0) {                            // one flag, then 16 raw bits
                    refill();
                    const uint32_t l = ((uint32_t)buf & 1) ? 17 : 1;
                    buf >>= l; n -= l; ulen += l;
                } else {
                    uint32_t glen = 0;
                    const uint32_t kr = rung * 0x01010101u + 0x02000100u;    // code length by the low two bits: r, r+1, r, r+2
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        if (UB == 3 ? (i % 3 == 0) : true) refill();     // 3 x 9 bits, or one code of up to 17
                        const uint32_t len = (kr >> (((uint32_t)buf & 3u) << 3)) & 0xffu;
                        buf >>= len; n -= len; glen += len;
                    }
                    ulen += glen;
                }
                if (lane == 0) {
                    if (UB == 3) ((uint8_t *)a.idx.ulen)[(uint64_t)gb * B + c] = (uint8_t)ulen;
                    else ((uint16_t *)a.idx.ulen)[(uint64_t)gb * B + c] = (uint16_t)ulen;
                }
            }
            gb++;
        }
        P = 32 * (w0 + wp) - n;
    }
    if (bad && lane == 0) atomicOr(a.status, 1u);
}

// idx.prev holds every segment's per-band sum of values: make it the value entering the segment (exclusive prefix,
// modulo the value width; a stream starts from zero).  One workgroup per tile.
template <typename T>
__global__ void __launch_bounds__(1024) prev_scan_kernel(const DecArgs a0) {
    const DecArgs a = dec_for_tile(a0, blockIdx.x);
    __shared__ uint32_t part[1024];
    const uint32_t tid = threadIdx.x, B = a.g.bands, c = blockIdx.y;       // one workgroup per tile and band
    const uint64_t nseg = a.g.nseg, per = (nseg + 1023) / 1024;
    const uint64_t s0 = (uint64_t)tid * per, s1 = (s0 + per < nseg) ? s0 + per : nseg;
    T *prev = (T *)a.idx.prev;
    uint32_t sum = 0;
    for (uint64_t s = s0; s < s1; s++) sum += prev[s * B + c];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {               // inclusive scan of the partial sums
        const uint32_t y = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += y;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;
    for (uint64_t s = s0; s < s1; s++) { const uint32_t t = prev[s * B + c]; prev[s * B + c] = (T)run; run += t; }
}

// Foreign stream: ONE lane walks the stream and rebuilds the index (bit position + band state at every
// segment start).  Latency bound by construction.
template <typename T, int MODE>
__global__ void dec_index_serial(const DecArgs a0) {
    if (threadIdx.x) return;
    const DecArgs a = dec_for_tile(a0, blockIdx.x);       // one lane per tile
    __shared__ uint64_t st_prev[MAXBANDS], st_cf[MAXBANDS];
    __shared__ uint32_t st_rung[MAXBANDS];
    const uint32_t bands = a.g.bands, S = a.g.seg_blocks;
    for (uint32_t c = 0; c < bands; c++) { st_prev[c] = 0; st_cf[c] = 0; st_rung[c] = 0; }
    const uint32_t nblocks = (uint32_t)a.g.nblocks;
    uint32_t gb0 = 0, gb_end = nblocks;
    uint64_t seg = 0, bp = 0;
    if (a.ix) {                             // restart point blockIdx.y of the container's coarse table
        const uint8_t *e = a.ix + (uint64_t)blockIdx.y * a.ix_E;
        for (uint32_t i = 0; i < 6; i++) bp |= (uint64_t)e[i] << (8 * i);
        const uint8_t *pv = e + 6 + bands, *cf = pv + bands * sizeof(T);
        for (uint32_t c = 0; c < bands; c++) {
            st_rung[c] = e[6 + c];
            uint64_t v = 0, f = 0;
            for (uint32_t i = 0; i < sizeof(T); i++) { v |= (uint64_t)pv[c * sizeof(T) + i] << (8 * i); if (MODE == CM_BEST) f |= (uint64_t)cf[c * sizeof(T) + i] << (8 * i); }
            st_prev[c] = v; st_cf[c] = f;
        }
        gb0 = blockIdx.y * a.ix_blocks;
        gb_end = (nblocks - gb0 < a.ix_blocks) ? nblocks : gb0 + a.ix_blocks;
        seg = gb0 / S;
    }
    Reader rd;
    rd.init(a.in32, a.in_bit0 + bp, a.in_bit0 + a.in_bits);
    T g[16];
    bool ok = true;
    uint32_t inseg = 0;
    for (uint32_t gb = gb0; gb < gb_end && ok; gb++) {
        if (inseg == 0) {
            a.idx.bitpos[seg] = rd.position() - a.in_bit0;
            for (uint32_t c = 0; c < bands; c++) {
                ((T *)a.idx.prev)[seg * bands + c] = (T)st_prev[c];
                if (MODE == CM_BEST) ((T *)a.idx.cf)[seg * bands + c] = (T)st_cf[c];
                a.idx.rung[seg * bands + c] = (uint8_t)st_rung[c];
            }
            seg++;
        }
        if (++inseg == S) inseg = 0;
        for (uint32_t c = 0; c < bands; c++) {
            uint32_t rung = st_rung[c];
            T cf = (T)st_cf[c];
            const uint64_t ustart = rd.position();
            ok = parse_unit<T, MODE, Reader>(rd, rung, cf, g) && ok;
            if (a.g.ulen_sz == 1) ((uint8_t *)a.idx.ulen)[(uint64_t)gb * bands + c] = (uint8_t)(rd.position() - ustart);
            else if (a.g.ulen_sz == 2) ((uint16_t *)a.idx.ulen)[(uint64_t)gb * bands + c] = (uint16_t)(rd.position() - ustart);
            T sum = 0;
#pragma unroll
            for (uint32_t i = 0; i < 16; i++) sum = (T)(sum + smag_t<T>(g[i]));
            st_prev[c] = (T)((T)st_prev[c] + sum);
            st_cf[c] = cf;
            st_rung[c] = rung;
        }
    }
    if (!ok) atomicOr(a.status, 1u);
}

// RLE0 (reference QB3encode.cpp:536-565) can only shorten a stream that holds a run of four zero bytes; looking for one
// on the device spares the host pass (a copy of the whole stream over PCIe and a byte loop) whenever there is none.
__global__ void zero_run_probe_kernel(const uint32_t *buf, uint64_t first_byte, uint64_t end_byte, uint32_t *flag) {
    const uint64_t ndw = (end_byte + 3) >> 2;
    bool found = false;
    for (uint64_t d = (first_byte >> 2) + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; d < ndw; d += (uint64_t)gridDim.x * blockDim.x) {
        // bytes outside [first_byte, end_byte) count as non-zero
        auto dw = [&](uint64_t i) -> uint32_t {
            if (i >= ndw) return 0xffffffffu;
            uint32_t v = buf[i];
            if (4 * i < first_byte) v |= 0xffffffffu >> (8 * (4 - (uint32_t)(first_byte - 4 * i)));
            if (4 * i + 4 > end_byte) v |= 0xffffffffu << (8 * (uint32_t)(end_byte - 4 * i));
            return v;
        };
        const uint32_t cur = dw(d), nxt = dw(d + 1);
        found = found || cur == 0 || __builtin_amdgcn_alignbit(nxt, cur, 8) == 0 || __builtin_amdgcn_alignbit(nxt, cur, 16) == 0 ||
                __builtin_amdgcn_alignbit(nxt, cur, 24) == 0;
    }
    if (__any(found) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

// ------------------------------------------------------------------ host side of the kernels
static thread_local char g_err[256] = "";
static uint64_t *g_stamps = nullptr;        // debugging aid (qb3x_debug_set_stamps): phase time stamps of the first waves
static uint32_t g_stamps_n = 0;
void dbg_set_stamps(void *d_buf, uint32_t nwaves) { g_stamps = (uint64_t *)d_buf; g_stamps_n = nwaves; }
const char *last_error() { return g_err; }
void set_error(const char *what, int e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, e ? hipGetErrorString((hipError_t)e) : "failed");
}
// ---- per-kernel timing: hipEvents recorded on the launch stream, resolved after the caller's sync
}  // namespace qb3dev
#include <map>
#include <mutex>
#include <string>
#include <vector>
namespace qb3dev {
struct ProfPending { const char *name; hipEvent_t a, b; };
static std::mutex g_prof_mu;
static int g_prof_level = 0;
static std::vector<ProfPending> g_prof_pending;
static std::vector<hipEvent_t> g_prof_pool;
static std::map<std::string, std::pair<double, uint64_t>> g_prof_acc;
void prof_enable(int level) { std::lock_guard<std::mutex> l(g_prof_mu); g_prof_level = level; }
// level 2 skips the microsecond kernels: two events per kernel cost more than those kernels take
static bool prof_minor(const char *n) { const std::string s(n); return s == "enc_scan" || s == "enc_seams" || s == "enc_best_scan"; }
void prof_reset() { std::lock_guard<std::mutex> l(g_prof_mu); g_prof_acc.clear(); }
static hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
struct ProfScope {      // records an event before and after the launches made in its lifetime
    hipStream_t st; hipEvent_t a = nullptr, b = nullptr; const char *name; bool on;
    ProfScope(const char *n, hipStream_t s) : st(s), name(n) {
        std::lock_guard<std::mutex> l(g_prof_mu);
        on = g_prof_level == 1 || (g_prof_level >= 2 && !prof_minor(n));
        if (on) { a = prof_event(); b = prof_event(); (void)hipEventRecord(a, st); }
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(b, st);
        std::lock_guard<std::mutex> l(g_prof_mu);
        g_prof_pending.push_back({name, a, b});
    }
};
void prof_collect() {
    std::lock_guard<std::mutex> l(g_prof_mu);
    for (auto &p : g_prof_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { auto &acc = g_prof_acc[p.name]; acc.first += ms; acc.second++; }
        g_prof_pool.push_back(p.a); g_prof_pool.push_back(p.b);
    }
    g_prof_pending.clear();
}
bool prof_get(const char *name, double *total_ms, uint64_t *count) {
    std::lock_guard<std::mutex> l(g_prof_mu);
    auto it = g_prof_acc.find(name);
    if (it == g_prof_acc.end()) return false;
    *total_ms = it->second.first; *count = it->second.second;
    return true;
}
int prof_names(char *buf, size_t n) {
    std::lock_guard<std::mutex> l(g_prof_mu);
    std::string s;
    for (auto &kv : g_prof_acc) { if (!s.empty()) s += ","; s += kv.first; }
    snprintf(buf, n, "%s", s.c_str());
    return (int)g_prof_acc.size();
}

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(#x, (int)e_); return (int)e_; } } while (0)

// Decoder workgroup geometry of the unit-parallel kernel: threads, blocks per pass, passes
static void fast_geometry(uint32_t bands, uint32_t tsz, uint32_t *threads, uint32_t *bpp, uint32_t *passes) {
    *threads = tsz == 8 ? 128 : 256;
    *bpp = *threads / bands;
    uint32_t k = (uint32_t)(16384 / ((size_t)*bpp * bands * tsz * 16));    // keep the pixel tile near 16 KB
    *passes = k < 1 ? 1 : (k > 3 ? 3 : k);
    static const int knob = [] { const char *e = getenv("QB3_DEC_PASSES"); return e ? atoi(e) : 0; }();   // tuning knob
    if (knob > 0 && (uint32_t)knob < *passes) *passes = (uint32_t)knob;
}
// Blocks per index segment.  A function of stream-intrinsic properties only (bands, value size, mode): encoder
// and decoder must agree on it whatever their strides, band maps or buffer alignments are.
// 16-bit lane-per-block kernels: bands = ng x bg, bg <= 4 bands per lane (0: no such split)
static void px16_split(uint32_t B, uint32_t *bg, uint32_t *ng) {
    *bg = B <= 4 ? B : (B % 4 == 0) ? 4 : (B % 2 == 0) ? 2 : 0;
    *ng = *bg ? B / *bg : 0;
}

uint32_t seg_blocks_for(const Geometry &g) {
    if (g.mode != CM_BEST) {      // one segment = the blocks of one decoder workgroup, at most 256
        static const uint32_t knob = [] { const char *e = getenv("QB3_SEG_BLOCKS"); int k = e ? atoi(e) : 0; return (uint32_t)(k < 0 ? 0 : k); }();
        if (knob) return knob;                                                  // tuning knob, process wide
        // 8-bit grey/RGB/RGBA: the lane-per-block decoder gives a segment to a WAVE (a function of type and band
        // count only: encoder and decoder must agree whatever kernel either of them ends up using)
        if (g.tsz == 1 && (g.bands == 1 || g.bands == 3 || g.bands == 4)) return 64;
        if (g.tsz == 2) {       // 16-bit: a wave = 64 lanes of (block, band group)
            uint32_t bg, ng;
            px16_split(g.bands, &bg, &ng);
            if (bg) return 64 / ng;
        }
        uint32_t threads, bpp, passes;
        fast_geometry(g.bands, g.tsz, &threads, &bpp, &passes);
        while (passes > 1 && bpp * passes > 256) passes--;
        return bpp * passes;
    }
    // common-factor modes: a lane walks the segment serially, keep it short (QB3_SEG_UNITS: tuning knob)
    static const uint32_t units = [] { const char *e = getenv("QB3_SEG_UNITS"); int v = e ? atoi(e) : 12; return (uint32_t)(v < 1 ? 1 : v); }();
    uint32_t s = units / g.bands;
    return s ? s : 1;
}
uint32_t ix_entry_bytes(const Geometry &g) { return 6 + g.bands * (1 + g.tsz * (g.mode == CM_BEST ? 2 : 1)); }
uint32_t ulen_size_for(uint32_t tsz, uint32_t mode) { return mode == CM_BEST ? 0 : (tsz == 1 ? 1 : 2); }

static size_t align8(size_t v) { return (v + 7) & ~(size_t)7; }
size_t index_bytes(const Geometry &g) {
    const size_t n = (size_t)g.nseg * g.bands;
    return align8(8 * (size_t)g.nseg) + 2 * align8(n * g.tsz) + align8(n) + align8((size_t)g.nblocks * g.bands * g.ulen_sz);
}
IndexView index_view(const Geometry &g, void *base) {
    IndexView v;
    uint8_t *p = (uint8_t *)base;
    const size_t n = (size_t)g.nseg * g.bands;
    v.bitpos = (uint64_t *)p; p += align8(8 * (size_t)g.nseg);
    v.prev = p; p += align8(n * g.tsz);
    v.cf = p; p += align8(n * g.tsz);
    v.rung = p; p += align8(n);
    v.ulen = g.ulen_sz ? p : nullptr;
    return v;
}

static uint32_t magic_div(uint32_t d) { return d == 1 ? 0u : (uint32_t)(((1ull << 32) + d - 1) / d); }   // exact for n*d < 2^32/d-ish, n small

static uint32_t max_unit_bits(uint32_t tsz, uint32_t mode = CM_FTL) {
    const uint32_t ub = tsz == 1 ? 3 : tsz == 2 ? 4 : tsz == 4 ? 5 : 6;
    const uint32_t plain = ub + 2 + 16 * (8 * tsz + 1);
    // common factor: signal + switch + 2 flags + own-rung switch + factor code + group
    return mode == CM_BEST ? plain + 3 * ub + 8 + 8 * tsz + 2 : plain;
}

// encoder workspace layout (all 8-byte aligned), EncResult last
struct EncWs { size_t bits, off, gsum, seams, scratch, cwhas, cwval, centry, res, total; uint32_t slot_dw, ngroups; };
static EncWs enc_ws_layout(const Geometry &g, uint32_t nchunks, uint32_t nbp) {
    EncWs w;
    // a multiple of 4 dwords: slots are 16-byte aligned (the px kernel copies them out as uint4)
    w.slot_dw = (uint32_t)(((31 + (size_t)nbp * g.bands * max_unit_bits(g.tsz, g.mode)) / 32 + 1 + 3) & ~(size_t)3);
    w.ngroups = (nchunks + SCAN_GROUP - 1) / SCAN_GROUP;
    size_t o = 0;
    w.bits = o; o += align8(4 * (size_t)nchunks);
    w.off = o; o += 8 * (size_t)nchunks;
    w.gsum = o; o += 8 * ((size_t)w.ngroups + 1);
    w.seams = o; o += 8 * (size_t)nchunks;
    o = (o + 15) & ~(size_t)15;
    w.scratch = o; o += align8(4 * (size_t)nchunks * w.slot_dw);
    const size_t nb = g.mode == CM_BEST ? (size_t)nchunks * g.bands : 0;
    w.cwhas = o; o += align8(nb);
    w.cwval = o; o += 8 * nb;
    w.centry = o; o += 8 * nb;
    w.res = o; o += sizeof(EncResult);
    w.total = o;
    return w;
}

// the 8-bit lane-per-block kernels need: uint8, 1/3/4 bands, rows of whole blocks at dword-aligned addresses,
// Hilbert or Z curve, identity or default RGB(A) band map
static bool px_eligible(const Geometry &g, bool *rgb) {
    if (g.tsz != 1 || !(g.bands == 1 || g.bands == 3 || g.bands == 4) || g.mode == CM_BEST) return false;
    if (g.w < 4 || g.h < 4) return false;                  // any width, stride and pointer: rows are read and written unaligned
    if (g.order != HILBERT && g.order != ZCURVE) return false;
    bool ident = true, def = g.bands >= 3;
    for (uint32_t c = 0; c < g.bands; c++) {
        ident = ident && g.cband[c] == c;
        def = def && g.cband[c] == ((c == 0 || c == 2) ? 1u : c);
    }
    *rgb = def && !ident;
    return (ident || def) && !getenv("QB3_NO_PX");
}

// 16-bit lane-per-(block, band group) kernels: bands = NG x BG with BG <= 4; a group must be whole dwords per
// pixel unless it is the whole pixel (BG = bands = 1 or 3)
static bool px16_eligible(const Geometry &g, bool *rgb, uint32_t *bg, uint32_t *ng) {
    if (g.tsz != 2 || g.mode == CM_BEST || getenv("QB3_NO_PX")) return false;
    if (g.w < 4 || g.h < 4) return false;                  // any width and stride: rows are read and written at halfword alignment
    if (g.order != HILBERT && g.order != ZCURVE) return false;
    const uint32_t B = g.bands;
    px16_split(B, bg, ng);
    if (!*bg) return false;
    // identity, or R-G, G, B-G on the first three bands (they must sit in one lane: 3 or 4 bands per group)
    bool ident = true, def = *bg >= 3;
    for (uint32_t c = 0; c < B; c++) {
        ident = ident && g.cband[c] == c;
        def = def && g.cband[c] == ((c == 0 || c == 2) ? 1u : c);
    }
    *rgb = def && !ident;
    return ident || def;
}

EncPlan plan_encode(const Geometry &g) {
    EncPlan p;
    const uint32_t dpr = g.bands * g.tsz;
    p.px = px_eligible(g, &p.px_rgb);
    p.px16 = false; p.px16_bg = p.px16_ng = 0;
    if (p.px) {
        p.threads = 256; p.slots = 256; p.nbp = 255;
        p.nchunks = (uint32_t)((g.nblocks + 254) / 255);
        const EncWs L = enc_ws_layout(g, p.nchunks, p.nbp);
        p.lds_bytes = 2048 + 256 + 4 * (size_t)L.slot_dw;
        p.ws_bytes = L.total;
        return p;
    }
    bool rgb16 = false;
    if (px16_eligible(g, &rgb16, &p.px16_bg, &p.px16_ng)) {
        p.px16 = true; p.px_rgb = rgb16;
        p.threads = 256; p.slots = 256 / p.px16_ng; p.nbp = p.slots - 1;
        p.nchunks = (uint32_t)((g.nblocks + p.nbp - 1) / p.nbp);
        const EncWs L = enc_ws_layout(g, p.nchunks, p.nbp);
        p.lds_bytes = 2048 + 256 + 1024 + 4 * (size_t)L.slot_dw;
        p.ws_bytes = L.total;
        return p;
    }
    p.threads = g.tsz == 8 ? 128 : 256;
    if (const char *e = getenv("QB3_ENC_THREADS")) p.threads = (uint32_t)atoi(e);      // tuning knob
    p.slots = p.threads / g.bands;
    const uint32_t nbp = p.slots - 1;
    p.nbp = nbp;
    p.nchunks = (uint32_t)((g.nblocks + nbp - 1) / nbp);
    const size_t outdw = (31 + (size_t)nbp * g.bands * max_unit_bits(g.tsz, g.mode)) / 32 + 1;
    p.lds_bytes = 8 * (size_t)p.slots + 4 * (size_t)(4 * p.slots * dpr) + 256 + 1024 + (((size_t)p.slots * g.bands + 7) & ~(size_t)7) + 4 * ((outdw + 1) & ~(size_t)1);
    if (g.mode == CM_BEST) p.lds_bytes += 12 * (size_t)p.slots * g.bands + 8;
    p.ws_bytes = enc_ws_layout(g, p.nchunks, nbp).total;
    return p;
}

// dispatch of the 8-bit lane-per-block encoder over its compile-time parameters
template <int B, bool RGB>
static void launch_enc_px_b(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    const bool step = a.g.mode != CM_FTL, z = a.g.order == ZCURVE;
    dim3 grid(plan.nchunks, a.ntiles), block(256);
    if (!z && !step) hipLaunchKernelGGL((enc_px_kernel<B, RGB, HILBERT, false>), grid, block, plan.lds_bytes, st, a);
    else if (!z && step) hipLaunchKernelGGL((enc_px_kernel<B, RGB, HILBERT, true>), grid, block, plan.lds_bytes, st, a);
    else if (z && !step) hipLaunchKernelGGL((enc_px_kernel<B, RGB, ZCURVE, false>), grid, block, plan.lds_bytes, st, a);
    else hipLaunchKernelGGL((enc_px_kernel<B, RGB, ZCURVE, true>), grid, block, plan.lds_bytes, st, a);
}
template <int BG, bool RGB>
static void launch_enc_px16_b(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    const bool step = a.g.mode != CM_FTL, z = a.g.order == ZCURVE;
    dim3 grid(plan.nchunks, a.ntiles), block(256);
    if (!z && !step) hipLaunchKernelGGL((enc_px16_kernel<BG, RGB, HILBERT, false>), grid, block, plan.lds_bytes, st, a);
    else if (!z && step) hipLaunchKernelGGL((enc_px16_kernel<BG, RGB, HILBERT, true>), grid, block, plan.lds_bytes, st, a);
    else if (z && !step) hipLaunchKernelGGL((enc_px16_kernel<BG, RGB, ZCURVE, false>), grid, block, plan.lds_bytes, st, a);
    else hipLaunchKernelGGL((enc_px16_kernel<BG, RGB, ZCURVE, true>), grid, block, plan.lds_bytes, st, a);
}
static void launch_enc_px16(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    switch (plan.px16_bg) {
    case 1: launch_enc_px16_b<1, false>(a, plan, st); break;
    case 2: launch_enc_px16_b<2, false>(a, plan, st); break;
    case 3: if (plan.px_rgb) launch_enc_px16_b<3, true>(a, plan, st); else launch_enc_px16_b<3, false>(a, plan, st); break;
    default: if (plan.px_rgb) launch_enc_px16_b<4, true>(a, plan, st); else launch_enc_px16_b<4, false>(a, plan, st); break;
    }
}
static void launch_enc_px(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    if (a.g.bands == 1) launch_enc_px_b<1, false>(a, plan, st);
    else if (a.g.bands == 3) { if (plan.px_rgb) launch_enc_px_b<3, true>(a, plan, st); else launch_enc_px_b<3, false>(a, plan, st); }
    else { if (plan.px_rgb) launch_enc_px_b<4, true>(a, plan, st); else launch_enc_px_b<4, false>(a, plan, st); }
}

template <typename T>
static int launch_encode_t(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    const bool step = a.g.mode != CM_FTL;
    const uint32_t nt = a.ntiles;
    dim3 grid(plan.nchunks, nt), block(plan.threads);
    if (a.g.mode == CM_BEST) {
        {
            ProfScope ps("enc_best_pass0", st);
            hipLaunchKernelGGL((enc_best_kernel<T, 0>), grid, block, plan.lds_bytes, st, a);
        }
        {
            ProfScope ps("enc_best_scan", st);
            hipLaunchKernelGGL(best_scan_kernel, dim3(1, nt), dim3(1024), 0, st, a);
        }
        {
            ProfScope ps("enc_best_units", st);
            hipLaunchKernelGGL((enc_best_kernel<T, 1>), grid, block, plan.lds_bytes, st, a);
        }
    } else if (plan.px && sizeof(T) == 1) {
        ProfScope ps("enc_units", st);
        launch_enc_px(a, plan, st);
    } else if (plan.px16 && sizeof(T) == 2 && ((uintptr_t)a.img & 1) == 0) {
        ProfScope ps("enc_units", st);
        launch_enc_px16(a, plan, st);
    } else {
        ProfScope ps("enc_units", st);
        if (step) hipLaunchKernelGGL((enc_kernel<T, true>), grid, block, plan.lds_bytes, st, a);
        else hipLaunchKernelGGL((enc_kernel<T, false>), grid, block, plan.lds_bytes, st, a);
    }
    {
        ProfScope ps("enc_scan", st);
        hipLaunchKernelGGL(enc_scan_kernel, dim3((plan.nchunks + SCAN_GROUP - 1) / SCAN_GROUP, nt), dim3(SCAN_GROUP / 4), 0, st, a);
    }
    {
        ProfScope ps("enc_scan", st);
        hipLaunchKernelGGL(enc_scan2_kernel, dim3(1, nt), dim3(1024), 0, st, a);
    }
    {
        ProfScope ps("enc_concat", st);
        hipLaunchKernelGGL(enc_concat_kernel, dim3((plan.nchunks + 3) / 4, nt), dim3(256), 0, st, a);
    }
    {
        ProfScope ps("enc_seams", st);
        hipLaunchKernelGGL(enc_seam_kernel, dim3((plan.nchunks + 1 + 255) / 256, nt), dim3(256), 0, st, a);
        if (a.hdr_len) hipLaunchKernelGGL(write_header_kernel, dim3(1, nt), dim3(64), 0, st, a);
        if (a.ix_dst && a.have_idx && nt == 1) hipLaunchKernelGGL(ix_fill_kernel, dim3((a.ix_K + 255) / 256), dim3(256), 0, st, a);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_encode(const Geometry &g, const EncPlan &plan, const void *img, uint32_t *out32, uint32_t out_bit0,
                  const BandState &st_in, void *ws, void *index, void *stream, const TileBatch &tb,
                  const uint8_t *hdr, uint32_t hdr_len, const IxTable &ix) {
    EncArgs a;
    a.ix_dst = ix.entries; a.ix_K = ix.K; a.ix_E = ix.entry_bytes;
    a.ix_spe = g.seg_blocks ? ix.blocks / g.seg_blocks : 0;
    a.ntiles = tb.n ? tb.n : 1; a.ts_img = tb.src_pitch; a.ts_out = tb.dst_pitch; a.ts_ws = tb.ws_pitch; a.ts_idx = tb.idx_pitch;
    a.hdr_len = hdr_len <= sizeof(a.hdr) ? hdr_len : 0;
    a.hdr_back = a.hdr_len + (ix.entries ? ix.K * ix.entry_bytes + 2 : 0);
    for (uint32_t i = 0; i < a.hdr_len; i++) a.hdr[i] = hdr[i];
    a.g = g; a.img = img; a.out32 = out32; a.out_bit0 = out_bit0;
    a.slots = plan.slots; a.nchunks = plan.nchunks; a.dpr = g.bands * g.tsz;
    a.magic_dpr = magic_div(a.dpr); a.magic_bands = magic_div(g.bands);
    uint8_t *w = (uint8_t *)ws;
    const EncWs L = enc_ws_layout(g, plan.nchunks, plan.nbp);
    a.chunk_bits = (uint32_t *)(w + L.bits);
    a.chunk_off = (uint64_t *)(w + L.off);
    a.group_sum = (uint64_t *)(w + L.gsum);
    a.seams = (uint32_t *)(w + L.seams);
    a.scratch = (uint32_t *)(w + L.scratch);
    a.cw_has = w + L.cwhas; a.cw_val = (uint64_t *)(w + L.cwval); a.centry = (uint64_t *)(w + L.centry);
    a.slot_dw = L.slot_dw;
    a.px_ng = plan.px16 ? plan.px16_ng : 1; a.px_magic_ng = magic_div(a.px_ng);
    a.stamps = g_stamps; a.stamps_n = g_stamps_n;
    a.px_aligned = !(g.w & 3) && !((g.stride * g.tsz) & 3) && !((uintptr_t)img & 3) && !(tb.src_pitch & 3);
    a.res = (EncResult *)(w + L.res);
    a.st = st_in;
    a.have_idx = index != nullptr;
    { const char *e = getenv("QB3_ENC_FLAGS"); a.flags = e ? (uint32_t)atoi(e) : 1; }
    a.idx = index ? index_view(g, index) : IndexView{nullptr, nullptr, nullptr, nullptr, nullptr};
    hipStream_t st = (hipStream_t)stream;
    switch (g.tsz) {
    case 1: return launch_encode_t<uint8_t>(a, plan, st);
    case 2: return launch_encode_t<uint16_t>(a, plan, st);
    case 4: return launch_encode_t<uint32_t>(a, plan, st);
    case 8: return launch_encode_t<uint64_t>(a, plan, st);
    }
    set_error("encode: bad value size", 0);
    return -1;
}

// LDS dwords per decoder lane: 16*bands values + 2*bands state values + bands rung bytes.  The count is made
// odd (conflict-free lane stride) for <= 4 byte values; 8-byte values need an 8-byte aligned lane base, so
// there it is made 2 mod 4.
static uint32_t dec_lane_dwords(const Geometry &g) {
    uint32_t dw = (uint32_t)((16 * g.bands * g.tsz + 2 * g.bands * g.tsz + g.bands + 3) / 4);
    if (g.tsz == 8) { dw = (dw + 1) & ~1u; if ((dw & 3) == 0) dw += 2; }
    else dw |= 1;
    return dw;
}

DecPlan plan_decode(const Geometry &g) {
    DecPlan p;
    // per lane: 16*bands values + 2*bands state values + bands rung bytes, rounded to an odd dword count
    const uint32_t lane_dw = dec_lane_dwords(g);
    uint32_t threads = 64;
    while (threads > 1 && (size_t)threads * lane_dw * 4 > 48 * 1024) threads >>= 1;
    p.threads = threads;
    p.nwg = (uint32_t)((g.nseg + threads - 1) / threads);
    p.lds_bytes = (size_t)threads * lane_dw * 4;
    p.ws_bytes = align8(index_bytes(g)) + 64;          // per tile: rebuilt index + its share of the status words
    // unit-parallel kernel: FTL/BASE with a per-unit length table, and every core band must itself be core
    // (true for every map the encoder's setter can produce, reference QB3encode.cpp:70-72); anything else keeps
    // the lane-per-segment kernel
    bool simple = g.mode != CM_BEST && g.ulen_sz != 0;
    for (uint32_t c = 0; c < g.bands; c++) simple = simple && g.cband[g.cband[c]] == g.cband[c];
    fast_geometry(g.bands, g.tsz, &p.threads2, &p.bpp, &p.passes);
    const uint32_t dpr = g.bands * g.tsz, NB = g.seg_blocks;
    p.passes = (NB + p.bpp - 1) / p.bpp;
    p.in_cap_dw = (NB * dpr * 4 + 8 + 1) & ~1u;         // room for a stream as large as the raw blocks
    p.lds2_bytes = 8 * (size_t)NB + 8 * 16 + 8 * 2 * MAXBANDS + 4 * 2 * MAXBANDS + 4 * (size_t)((p.bpp + 1) & ~1u)
                 + 4 * (size_t)p.in_cap_dw + 16 * (size_t)NB * dpr + align8(2 * (size_t)p.bpp * g.bands) + 2048;
    p.fast = simple && p.lds2_bytes <= 64 * 1024;
    // 8-bit lane-per-block kernel
    bool rgb = false;
    p.px = p.fast && px_eligible(g, &rgb);
    p.px_rgb = rgb;
    // staging of the px kernel: the longest valid segment (every unit at its maximum) + the word the first unit
    // starts in + 8 zero words, after the 4 KB table, the scan scratch and the unit lengths
    p.px_cap_dw = (uint32_t)(((size_t)NB * g.bands * max_unit_bits(g.tsz, g.mode) + 31) / 32 + 2);
    p.px = p.px && NB <= 64;
    p.lds_px = 4096 + 4 * 4 * ((size_t)p.px_cap_dw + 8);       // table + four waves' staging
    p.px16 = false; p.px16_bg = p.px16_ng = 0;
    bool rgb16 = false;
    if (!p.px && p.fast && px16_eligible(g, &rgb16, &p.px16_bg, &p.px16_ng) && NB * p.px16_ng <= 64) {
        p.px16 = true; p.px_rgb = rgb16;
        p.lds_px = 4096 + 4 * 4 * ((size_t)p.px_cap_dw + 16);
    }
    return p;
}

template <int B, bool RGB>
static void launch_dec_px_b(const DecArgs &a, const DecPlan &plan, hipStream_t st) {
    const bool step = a.g.mode != CM_FTL, z = a.g.order == ZCURVE;
    dim3 grid((uint32_t)((a.g.nseg + 3) / 4), a.ntiles), block(256);
    if (!z && !step) hipLaunchKernelGGL((dec_px_kernel<B, RGB, HILBERT, false>), grid, block, plan.lds_px, st, a);
    else if (!z && step) hipLaunchKernelGGL((dec_px_kernel<B, RGB, HILBERT, true>), grid, block, plan.lds_px, st, a);
    else if (z && !step) hipLaunchKernelGGL((dec_px_kernel<B, RGB, ZCURVE, false>), grid, block, plan.lds_px, st, a);
    else hipLaunchKernelGGL((dec_px_kernel<B, RGB, ZCURVE, true>), grid, block, plan.lds_px, st, a);
}
template <int BG, bool RGB>
static void launch_dec_px16_b(const DecArgs &a, const DecPlan &plan, hipStream_t st) {
    const bool step = a.g.mode != CM_FTL, z = a.g.order == ZCURVE;
    dim3 grid((uint32_t)((a.g.nseg + 3) / 4), a.ntiles), block(256);
    if (!z && !step) hipLaunchKernelGGL((dec_px16_kernel<BG, RGB, HILBERT, false>), grid, block, plan.lds_px, st, a);
    else if (!z && step) hipLaunchKernelGGL((dec_px16_kernel<BG, RGB, HILBERT, true>), grid, block, plan.lds_px, st, a);
    else if (z && !step) hipLaunchKernelGGL((dec_px16_kernel<BG, RGB, ZCURVE, false>), grid, block, plan.lds_px, st, a);
    else hipLaunchKernelGGL((dec_px16_kernel<BG, RGB, ZCURVE, true>), grid, block, plan.lds_px, st, a);
}
static void launch_dec_px16(const DecArgs &a, const DecPlan &plan, hipStream_t st) {
    switch (plan.px16_bg) {
    case 1: launch_dec_px16_b<1, false>(a, plan, st); break;
    case 2: launch_dec_px16_b<2, false>(a, plan, st); break;
    case 3: if (plan.px_rgb) launch_dec_px16_b<3, true>(a, plan, st); else launch_dec_px16_b<3, false>(a, plan, st); break;
    default: if (plan.px_rgb) launch_dec_px16_b<4, true>(a, plan, st); else launch_dec_px16_b<4, false>(a, plan, st); break;
    }
}
static void launch_dec_px(const DecArgs &a, const DecPlan &plan, hipStream_t st) {
    if (a.g.bands == 1) launch_dec_px_b<1, false>(a, plan, st);
    else if (a.g.bands == 3) { if (plan.px_rgb) launch_dec_px_b<3, true>(a, plan, st); else launch_dec_px_b<3, false>(a, plan, st); }
    else { if (plan.px_rgb) launch_dec_px_b<4, true>(a, plan, st); else launch_dec_px_b<4, false>(a, plan, st); }
}

template <typename T, int MODE>
static int launch_decode_tm(const DecArgs &a, const DecPlan &plan, bool rebuild, hipStream_t st) {
    const bool use_px = plan.px && MODE != CM_BEST && sizeof(T) == 1;
    const bool use_px16 = plan.px16 && MODE != CM_BEST && sizeof(T) == 2 && ((uintptr_t)a.img & 1) == 0;
    if (rebuild && (use_px || use_px16) && !getenv("QB3_SLOW_INDEX")) {
        // foreign stream through the lane-per-block kernels: walk the lengths, then let the parallel decoder itself
        // produce the values entering the segments (totals pass + scan)
        {
            ProfScope ps("dec_index_serial", st);
            const dim3 wg(a.ntiles, a.ix ? a.ix_K : 1);
            if (sizeof(T) == 1) hipLaunchKernelGGL(dec_walk_kernel<3>, wg, dim3(64), 0, st, a);
            else hipLaunchKernelGGL(dec_walk_kernel<4>, wg, dim3(64), 0, st, a);
        }
        {
            ProfScope ps("dec_index_prev", st);
            DecArgs t = a;
            t.totals_only = 1;
            if (use_px) launch_dec_px(t, plan, st); else launch_dec_px16(t, plan, st);
        }
        ProfScope ps("dec_index_scan", st);
        if (sizeof(T) == 1) hipLaunchKernelGGL(prev_scan_kernel<uint8_t>, dim3(a.ntiles, a.g.bands), dim3(1024), 0, st, a);
        else hipLaunchKernelGGL(prev_scan_kernel<uint16_t>, dim3(a.ntiles, a.g.bands), dim3(1024), 0, st, a);
    } else if (rebuild) {
        ProfScope ps("dec_index_serial", st);
        hipLaunchKernelGGL((dec_index_serial<T, MODE>), dim3(a.ntiles, a.ix ? a.ix_K : 1), dim3(64), 0, st, a);
    }
    if (plan.px && MODE != CM_BEST && sizeof(T) == 1) {
        ProfScope ps("dec_units", st);
        launch_dec_px(a, plan, st);
    } else if (plan.px16 && MODE != CM_BEST && sizeof(T) == 2 && ((uintptr_t)a.img & 1) == 0) {
        ProfScope ps("dec_units", st);
        launch_dec_px16(a, plan, st);
    } else if (plan.fast && MODE != CM_BEST) {
        ProfScope ps("dec_units", st);
        hipLaunchKernelGGL((dec3_kernel<T, MODE == CM_BASE>), dim3((uint32_t)a.g.nseg, a.ntiles), dim3(plan.threads2), plan.lds2_bytes, st, a);
    } else {
        ProfScope ps("dec_segments", st);
        hipLaunchKernelGGL((dec_kernel<T, MODE>), dim3(plan.nwg, a.ntiles), dim3(plan.threads), plan.lds_bytes, st, a);
    }
    HIPCHK(hipGetLastError());
    return 0;
}
template <typename T>
static int launch_decode_t(const DecArgs &a, const DecPlan &plan, bool rebuild, hipStream_t st) {
    switch (a.g.mode) {
    case CM_FTL: return launch_decode_tm<T, CM_FTL>(a, plan, rebuild, st);
    case CM_BASE: return launch_decode_tm<T, CM_BASE>(a, plan, rebuild, st);
    default: return launch_decode_tm<T, CM_BEST>(a, plan, rebuild, st);
    }
}

// *has_run = 1 when bytes [off, off + nbytes) of the dword-aligned device buffer hold four consecutive zero bytes.
// d_flag: one device word of scratch.  Synchronises the stream.
int zero_run_probe(const void *d_buf, size_t off, size_t nbytes, void *d_flag, int *has_run, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(d_flag, 0, 4, st));
    const uint64_t ndw = (off + nbytes + 3) / 4 - off / 4;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(4096, (ndw + 255) / 256 ? (ndw + 255) / 256 : 1);
    hipLaunchKernelGGL(zero_run_probe_kernel, dim3(blocks), dim3(256), 0, st, (const uint32_t *)d_buf, (uint64_t)off, (uint64_t)(off + nbytes), (uint32_t *)d_flag);
    uint32_t f = 0;
    HIPCHK(hipMemcpyAsync(&f, d_flag, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *has_run = (int)f;
    return 0;
}

int launch_decode(const Geometry &g, const DecPlan &plan, const uint32_t *in32, uint32_t in_bit0, uint64_t in_bits,
                  void *img, const void *index, void *ws, uint32_t **status_out, void *stream, const TileBatch &tb,
                  const uint64_t *tile_bits, const IxTable &ix) {
    hipStream_t st = (hipStream_t)stream;
    DecArgs a;
    // the container's coarse restart table is usable when it matches this geometry and this library's segments
    a.ix = nullptr; a.ix_K = a.ix_blocks = a.ix_E = 0;
    if (ix.entries && !tb.n && ix.blocks && ix.blocks % g.seg_blocks == 0 && ix.entry_bytes == ix_entry_bytes(g) &&
        ix.K == (g.nblocks + ix.blocks - 1) / ix.blocks) {
        a.ix = ix.entries; a.ix_K = ix.K; a.ix_blocks = ix.blocks; a.ix_E = ix.entry_bytes;
    }
    a.g = g; a.in32 = in32; a.in_bit0 = in_bit0; a.in_bits = in_bits; a.img = img;
    a.ntiles = tb.n ? tb.n : 1; a.ts_in = tb.src_pitch; a.ts_img = tb.dst_pitch; a.tile_bits = tile_bits;
    uint8_t *w = (uint8_t *)ws;
    const bool rebuild = index == nullptr;
    // workspace: [status words, 64 bytes per 16 tiles][rebuilt indices, one per tile]
    const size_t status_bytes = ((4 * (size_t)a.ntiles + 63) / 64) * 64;
    a.status = (uint32_t *)w;
    a.idx = index_view(g, rebuild ? (void *)(w + status_bytes) : const_cast<void *>(index));
    a.ts_idx = rebuild ? align8(index_bytes(g)) : tb.idx_pitch;
    HIPCHK(hipMemsetAsync(a.status, 0, status_bytes, st));
    a.lane_dw = dec_lane_dwords(g);
    a.dpr = g.bands * g.tsz;
    a.bpp = plan.bpp; a.passes = plan.passes; a.in_cap_dw = (plan.px || plan.px16) ? plan.px_cap_dw : plan.in_cap_dw;
    a.px_ng = plan.px16 ? plan.px16_ng : 1; a.px_magic_ng = magic_div(a.px_ng);
    a.totals_only = 0;
    a.stamps = g_stamps; a.stamps_n = g_stamps_n;
    a.px_aligned = !(g.w & 3) && !((g.stride * g.tsz) & 3) && !((uintptr_t)img & 3) && !(tb.dst_pitch & 3);
    a.magic_bpp = magic_div(plan.bpp); a.magic_dpr = magic_div(a.dpr);
    *status_out = a.status;
    switch (g.tsz) {
    case 1: return launch_decode_t<uint8_t>(a, plan, rebuild, st);
    case 2: return launch_decode_t<uint16_t>(a, plan, rebuild, st);
    case 4: return launch_decode_t<uint32_t>(a, plan, rebuild, st);
    case 8: return launch_decode_t<uint64_t>(a, plan, rebuild, st);
    }
    set_error("decode: bad value size", 0);
    return -1;
}

}  // namespace qb3dev

// ------------------------------------------------------------------ quantisation (elementwise, HBM bound)
namespace qb3dev {

// reference QB3encode.cpp:137-186: round to nearest; ties toward zero, or away from zero when `away`
template <typename TS>
__global__ void quantize_kernel(TS *dst, const TS *src, uint32_t rowvals, uint32_t rows, uint64_t stride, uint64_t quanta, int away) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)rowvals * rows) return;
    const uint64_t y = i / rowvals, x = i - y * rowvals;
    const TS v = src[y * stride + x], q = (TS)quanta;
    TS r;
    if (q == 2) r = away ? (TS)(v / 2 + v % 2) : (TS)(v / 2);
    else if (q == 3) r = (TS)(v / 3 + (v % 3) / 2);
    else if (q == 4) r = away ? (TS)(v / 4 + (v % 4) / 2) : (TS)(v / 4 + (v % 4) / 3);
    else {
        const TS m = (TS)(v % q);
        const bool neg = v < (TS)0;
        if (away) { const TS h = (TS)(q / 2 + q % 2); r = (TS)(v / q + (!neg & (m >= h)) - (neg & ((TS)(m + h) <= (TS)0))); }
        else { const TS h = (TS)(q / 2); r = (TS)(v / q + (!neg & (m > h)) - (neg & ((TS)(m + h) < (TS)0))); }
    }
    dst[i] = r;
}

// reference QB3decode.cpp:77-107: multiply back, saturating
template <typename TS>
__global__ void dequantize_kernel(TS *img, uint32_t rowvals, uint32_t rows, uint64_t stride, uint64_t quanta) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)rowvals * rows) return;
    const uint64_t y = i / rowvals, x = i - y * rowvals;
    constexpr bool is_signed = (TS)-1 < (TS)0;
    constexpr TS tmax = is_signed ? (TS)(((uint64_t)1 << (8 * sizeof(TS) - 1)) - 1) : (TS)~(TS)0;
    constexpr TS tmin = is_signed ? (TS)((uint64_t)1 << (8 * sizeof(TS) - 1)) : (TS)0;
    const TS q = (TS)quanta, mai = (TS)(tmax / q), mii = (TS)(tmin / q);
    const TS v = img[y * stride + x];
    TS r = (v <= mai) ? (TS)(v * q) : tmax;
    if (is_signed && q > 2 && v < mii) r = tmin;
    img[y * stride + x] = r;
}

template <typename TS> static int launch_q(void *dst, const void *src, const Geometry &g, uint64_t q, bool away, hipStream_t st) {
    const uint64_t n = (uint64_t)g.w * g.bands * g.h;
    hipLaunchKernelGGL(quantize_kernel<TS>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, (TS *)dst, (const TS *)src,
                       g.w * g.bands, g.h, g.stride, q, away ? 1 : 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("quantize", (int)e); return (int)e; }
    return 0;
}
template <typename TS> static int launch_dq(void *img, const Geometry &g, uint64_t q, hipStream_t st) {
    const uint64_t n = (uint64_t)g.w * g.bands * g.h;
    hipLaunchKernelGGL(dequantize_kernel<TS>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, (TS *)img, g.w * g.bands, g.h, g.stride, q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dequantize", (int)e); return (int)e; }
    return 0;
}

int launch_quantize(void *dst, const void *src, const Geometry &g, int dtype, uint64_t q, bool away, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
    case 0: return launch_q<uint8_t>(dst, src, g, q, away, st);   case 1: return launch_q<int8_t>(dst, src, g, q, away, st);
    case 2: return launch_q<uint16_t>(dst, src, g, q, away, st);  case 3: return launch_q<int16_t>(dst, src, g, q, away, st);
    case 4: return launch_q<uint32_t>(dst, src, g, q, away, st);  case 5: return launch_q<int32_t>(dst, src, g, q, away, st);
    case 6: return launch_q<uint64_t>(dst, src, g, q, away, st);  case 7: return launch_q<int64_t>(dst, src, g, q, away, st);
    }
    return -1;
}
int launch_dequantize(void *img, const Geometry &g, int dtype, uint64_t q, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
    case 0: return launch_dq<uint8_t>(img, g, q, st);   case 1: return launch_dq<int8_t>(img, g, q, st);
    case 2: return launch_dq<uint16_t>(img, g, q, st);  case 3: return launch_dq<int16_t>(img, g, q, st);
    case 4: return launch_dq<uint32_t>(img, g, q, st);  case 5: return launch_dq<int32_t>(img, g, q, st);
    case 6: return launch_dq<uint64_t>(img, g, q, st);  case 7: return launch_dq<int64_t>(img, g, q, st);
    }
    return -1;
}

}  // namespace qb3dev
