// qb3_amd/csrc/qb3_kernels.h -- shared by the kernel translation units (k_*.hip) of the MI355X-native QB3 block codec
// (gfx950, wave64): device helpers, kernel argument blocks, the bit writer / reader, and the host-side launcher
// entry points each translation unit provides.  Nothing here is exported from the shared library.
//
// What the kernels compute is the reference's per-band 4x4 micro-block code (reference QB3lib/QB3encode.h:155-280,
// 376-724; QB3decode.h:142-741) -- bit-identical streams -- organised for the GPU:
//
// ENCODE  A workgroup owns a CHUNK of consecutive blocks.  The running predictor of the reference
//   (`prv += g -= prv`, QB3encode.h:434) makes the value entering a unit the last pixel the curve visited in the
//   previous block, and the rung-switch code needs only the previous block's rung, so a chunk is self-contained
//   once it also looks at ONE halo block.  Units are coded once, into an LDS bit buffer starting at bit 0 of the
//   chunk; the only global dependency is the chunk's bit offset (64-bit: a 16384^2 x 3 stream exceeds 2^32 bits).
//   The chunk's bits go to a private slot in the workspace, a two-level scan of the chunk totals gives the offsets,
//   enc_concat_kernel funnel-shifts every slot into place (single-pass variants with a decoupled look-back were built
//   and measured slower on this part).  Dwords shared by two chunks are assembled by
//   enc_finish_kernel from a two-entry-per-chunk seam table: no memset of the output, no global atomics on it.
//   8-bit grey/RGB/RGBA and 16-bit rasters use lane-per-block kernels that keep the block in registers
//   (k_enc_px.hip, k_enc_px16.hip); everything else the unit-per-lane kernels (k_enc_generic.hip, k_enc_best.hip).
//
// DECODE  A unit's bit position and rung depend on every earlier unit (QB3decode.h:445-454); the stream has no
//   restart points.  With an index (bit position + band state per SEGMENT of blocks, and for FTL/BASE the bit
//   length of every unit) everything is a scan: positions, rungs and entering values (k_dec_px.hip,
//   k_dec_px16.hip, dec3_kernel).  Common-factor streams keep short segments walked by one lane each (dec_kernel).
//   Without an index the unit lengths are found by walking the stream (k_dec_walk.hip): from the restart points of
//   the container's own "ix" chunks when it has them (one lane per restart point), else serially.
//
// No MFMA anywhere: this is integer bit packing, bounded by VALU/LDS issue and memory latency, nominally by HBM.
#pragma once
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


typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));     // four dwords at any dword address
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
    uint32_t *centry_parts; //   ... best_scan_kernel: the last writer of every part of the chunk range, per band
    uint8_t *cw_used;       //   ... per chunk and band: something in the chunk depended on the factor on entry
    uint8_t *seg_from_entry;    //   ... per index segment and band: its factor entry is the chunk's entering factor
    uint32_t *recode_need, *recode_list, *recode_n;    //   ... chunks to code again: flag per chunk, list, count
    uint32_t ntiles;
    uint64_t ts_img, ts_out, ts_ws, ts_idx;     // batched tiles: byte strides from tile to tile (blockIdx.y = tile)
    uint32_t hdr_len;       // container header bytes to put in front of the stream (enc_finish_kernel)
    uint32_t hdr_back;      // distance from the container start to the stream start (= hdr_len without an index chunk)
    uint8_t hdr[80];        // at most 11 + 20 (CB) + 12 (QV) + 12 (SC) + 12 (ix head) bytes
    uint32_t px_ng, px_magic_ng;    // 16-bit lane-per-block kernel: band groups per block (lanes per block), magic of it
    uint32_t px16_bg;               // ... bands a lane owns (what the table's fields are counted by, whichever kernel codes)
    uint32_t px_aligned;            // lane-per-block kernels: every row of every block is dword aligned (plain dword loads)
    uint8_t *ix_dst;                // restart table: where its first chunk goes (null: none), "DT" right after the last
    uint32_t ix_K, ix_spe, ix_E;    //   ... entries, fine segments per entry, bytes per entry
    uint32_t ix_bl;                 //   ... entries end with their segment's block lengths (IX_BL_BITS bits each)
    uint32_t ix_per_chunk, ix_blocks;   // ... entries per chunk, blocks per entry
    EncResult *res;
    BandState st;
    IndexView idx;
    uint32_t have_idx;
    uint32_t zrun_probe;    // enc_concat_kernel also looks for four zero bytes in a row (res->zero_run; RLE0 can only win on such a stream)
    uint32_t idx_no_ulen;   // the index is the library's own, only sampled for the restart table: segment entries, no unit lengths
    uint32_t chunk0, chunk_end;     // the chunks this launch codes / moves ([0, nchunks) but for the strips of a pipelined host call)
    uint32_t finish_what;           // enc_finish_kernel: 1 = seams and stream length, 2 = index positions, table entries, header; 3 = both
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
        a.cw_has = shift_ptr(a.cw_has, w); a.cw_val = shift_ptr(a.cw_val, w); a.centry = shift_ptr(a.centry, w); a.cw_used = shift_ptr(a.cw_used, w); a.seg_from_entry = shift_ptr(a.seg_from_entry, w); a.recode_need = shift_ptr(a.recode_need, w); a.recode_list = shift_ptr(a.recode_list, w); a.recode_n = shift_ptr(a.recode_n, w); a.centry_parts = shift_ptr(a.centry_parts, w);
        a.idx.bitpos = shift_ptr(a.idx.bitpos, x); a.idx.prev = shift_ptr(a.idx.prev, x); a.idx.cf = shift_ptr(a.idx.cf, x);
        a.idx.rung = shift_ptr(a.idx.rung, x); a.idx.ulen = shift_ptr(a.idx.ulen, x);
        a.ix_dst = shift_ptr(a.ix_dst, t * a.ts_out);     // (the restart table sits at the same place in every tile's container)
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
// chunks the common-factor encoders analyse first, to choose between one coding pass and two: an eighth of them, 64 at least, 1024 at most
// (a raster of fewer than BEST_SAMPLE_MIN chunks is not sampled at all: coding it once and again where the assumption was wrong costs at
// most a second pass, which is what the sample could save -- and the sample is a launch of its own, a fifth of such a raster's coding time)
constexpr uint32_t BEST_SAMPLE_MIN = 32768;
__host__ __device__ constexpr uint32_t best_sample_count(uint32_t nchunks) { return nchunks <= 64 ? nchunks : (nchunks / 8 < 64 ? 64u : nchunks / 8 > 1024 ? 1024u : nchunks / 8); }
// every coding kernel's first workgroup: the counter by which enc_scan_kernel's last workgroup knows itself (behind the group sums)
__device__ __forceinline__ void enc_scan_counter_reset(const EncArgs &a) {
    if (blockIdx.x == 0 && threadIdx.x == 0) a.group_sum[(a.nchunks + SCAN_GROUP - 1) / SCAN_GROUP + 1] = 0;
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
    uint32_t magic_bands;           // lane-per-unit kernels (k_dec_pxu.hip): lane / bands
    uint32_t from_ix;        // lane-per-segment decoder: the segments are the pieces between the entries of the container's restart table
    uint32_t seg_cap_dw;     // lane-per-segment decoder: stream words a workgroup may stage in LDS (0: lanes read global memory)
    uint32_t in_cap_full;    // 16-bit lane-per-block decoder: the worst-case staging (in_cap_dw may be sized for this stream's average)
    uint32_t px_ng, px_magic_ng;    // 16-bit lane-per-block kernel: band groups (lanes) per block
    uint32_t totals_only;           // lane-per-block kernels: write the segments' per-band sums to idx.prev, no pixels
    uint32_t px_aligned;            // lane-per-block kernels: every row of every block is dword aligned (plain dword stores)
    const uint8_t *ix;              // coarse index chunk found in the container (null: none): restart points for the walk
    uint32_t ix_K, ix_blocks, ix_E, ix_per_chunk, ix_pad;     // ix_pad: bytes of the pad chunk behind every table chunk
    uint32_t ix_bl;                 // the entries carry block lengths: the lane-per-block decoder needs no walk and no index
    uint32_t ix_ver, ix_check_heads;    // version of the table's chunks (3: with checks); the host has not seen the chunk heads behind the first
    uint32_t chk_wgs;                   // dec_px_kernel from the entries alone: the launch's first chk_wgs workgroups check a chunk of the table each (ix_check_chunk)
    uint32_t bl_mode;               // ... and this launch decodes from them
    uint64_t seg0, seg_end;         // lane-per-block decoders: the segments this launch decodes ([0, nseg) but for the strips of a pipelined host call)
    uint32_t wide_band;             // plain 32/64-bit streams: rungs in the band the walk's table covers (16; 8: QB3_WIDE_BAND, a test hook)
    // batched tiles (blockIdx.y = tile): byte strides, and each tile's stream length in bits (null: in_bits for all)
    uint32_t ntiles;
    uint64_t ts_in, ts_img, ts_idx;
    const uint64_t *tile_bits;
};


// The check a version 3 "ix" chunk carries in the two reserved bytes of its head: a position-weighted sum of the chunk's
// entry bytes folded to 16 bits (a table sits in an ignorable chunk the format does not protect; its positions, rungs and
// values are taken as truth by the decoder, so a damaged table must be told from a good one: the decoder then falls back
// to the plain walk).  part = this thread's share of the sum over bytes [0, n) of `e`, thread t of nthr taking i = t, t + nthr, ...
__device__ __forceinline__ uint32_t ix_sum_part(const uint8_t *e, uint32_t n, uint32_t t, uint32_t nthr) {
    // sum over i in [0, n) of (e[i] + 1) * (i * K + 1), K = 0x9e3779b1 (mod 2^32).  The bytes are read sixteen at a time from
    // the aligned address at or below e (a table starts wherever the header ends); bytes outside [0, n) do not count.
    const uint32_t off = (uint32_t)((uintptr_t)e & 15);
    const uint4 *base = (const uint4 *)(e - off);
    const uint32_t nvec = (off + n + 15) >> 4;
    uint32_t s = 0;
    for (uint32_t v = t; v < nvec; v += nthr) {
        const uint4 q = base[v];
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) {
            const uint32_t i = 16 * v + k - off;            // (wraps for the bytes in front of e: then i >= n)
            const uint32_t b = (w[k >> 2] >> (8 * (k & 3))) & 0xffu;
            s += i < n ? (b + 1u) * (i * 0x9e3779b1u + 1u) : 0u;
        }
    }
    return s;
}
__device__ __forceinline__ uint32_t ix_sum_fold(uint32_t s) { return (s ^ (s >> 16)) & 0xffffu; }

// Checks chunk c of the container's restart table: its head where the host has not read it (signature, length, version, flags, blocks
// per entry, the pad chunk behind) and, for version 3 tables, the 16-bit check of its entries.  A workgroup of 256; a mismatch raises
// status bit 5.  part: four words of LDS (the caller's: a static array here would take LDS address 0 from dec_px_kernel's table)
__device__ __forceinline__ void ix_check_chunk(const DecArgs &a, uint32_t c, uint32_t *part) {
    const uint32_t nch = (a.ix_K + a.ix_per_chunk - 1) / a.ix_per_chunk;
    const uint32_t here = (c + 1 < nch) ? a.ix_per_chunk : a.ix_K - c * a.ix_per_chunk;
    const uint8_t *chunk = a.ix + (uint64_t)c * (IX_HEAD + a.ix_pad + (uint64_t)a.ix_per_chunk * a.ix_E);
    bool bad = false;
    if (threadIdx.x == 0 && (a.ix_check_heads || a.ix_ver >= 3)) {
        const uint32_t len = IX_HEAD + here * a.ix_E;
        const uint32_t blocks = chunk[8] | (chunk[9] << 8) | (chunk[10] << 16) | ((uint32_t)chunk[11] << 24);
        bad = chunk[0] != 'i' || chunk[1] != 'x' || (chunk[2] | (chunk[3] << 8)) != (int)len || chunk[4] != a.ix_ver ||
              (chunk[5] & 3) != ((a.g.mode == CM_BEST ? 1u : 0u) | (a.ix_bl ? 2u : 0u)) || blocks != a.ix_blocks;
        if (a.ix_pad) bad = bad || chunk[len] != 'z' || chunk[len + 1] != 'z' || chunk[len + 2] != 4 || chunk[len + 3] != 0;
        if (c + 1 == nch) bad = bad || chunk[len + a.ix_pad] != 'D' || chunk[len + a.ix_pad + 1] != 'T';
    }
    if (a.ix_ver >= 3) {
        uint32_t s = ix_sum_part(chunk + IX_HEAD, here * a.ix_E, threadIdx.x, 256);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) s += (uint32_t)__shfl_xor((int)s, d, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) bad = bad || ix_sum_fold(part[0] + part[1] + part[2] + part[3]) != (uint32_t)(chunk[6] | (chunk[7] << 8));
    }
    if (bad) atomicOr(a.status, 32u);
}

// entry k of a restart table whose first chunk starts at `base` (layout: qb3_dev.h, IxTable)
__device__ __forceinline__ const uint8_t *ix_entry_at(const uint8_t *base, uint32_t per_chunk, uint32_t E, uint32_t pad, uint32_t k) {
    const uint32_t c = k / per_chunk, j = k - c * per_chunk;
    return base + (uint64_t)c * (IX_HEAD + pad + (uint64_t)per_chunk * E) + IX_HEAD + (uint64_t)j * E;
}

__device__ __forceinline__ DecArgs dec_for_tile(DecArgs a, uint32_t t) {
    if (a.tile_bits) a.in_bits = a.tile_bits[t];
    if (t) {
        a.in32 = shift_ptr(a.in32, t * a.ts_in);
        a.img = shift_ptr((uint8_t *)a.img, t * a.ts_img);
        const uint64_t x = t * a.ts_idx;
        a.idx.bitpos = shift_ptr(a.idx.bitpos, x); a.idx.prev = shift_ptr(a.idx.prev, x); a.idx.cf = shift_ptr(a.idx.cf, x);
        a.idx.rung = shift_ptr(a.idx.rung, x); a.idx.ulen = shift_ptr(a.idx.ulen, x);
        a.ix = shift_ptr(a.ix, t * a.ts_in);               // (the restart table sits at the same place in every tile's container)
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
// flags (optional; the plain-stream walks of k_dec_walk.hip): bit 0 = a common-factor unit that takes the factor of the band's
// last such unit (pcf as handed in), bit 1 = one that brings its own (pcf is that factor afterwards)
template <typename T, int MODE, typename RD> __device__ __forceinline__ bool parse_unit(RD &rd, uint32_t &rung, T &pcf, T (&g)[16], uint32_t *flags = nullptr) {
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
            if (flags) *flags |= 2u;
        } else if (flags) *flags |= 1u;
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

// ------------------------------------------------------------------ host side shared by the translation units
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(#x, (int)e_); return (int)e_; } } while (0)

// records a HIP event before and after the launches made in its lifetime (k_host.hip; off unless profiling is enabled)
struct ProfScope {
    hipStream_t st; hipEvent_t a = nullptr, b = nullptr; const char *name; bool on;
    ProfScope(const char *n, hipStream_t s);
    ~ProfScope();
};

// process-wide debugging switches, read once from the environment (k_host.hip)
struct Tuning { bool no_px; bool slow_index; bool slow_walk; bool no_bl; size_t walk_tab_kb; int wide_band; uint32_t best_sample_min; int64_t exits_from; };
const Tuning &tuning();

uint32_t magic_div(uint32_t d);
uint32_t max_unit_bits(uint32_t tsz, uint32_t mode = CM_FTL);

// launchers, one per translation unit; they enqueue on `st` and return nothing (errors surface through hipGetLastError)
void launch_enc_generic(const EncArgs &a, const EncPlan &plan, hipStream_t st);    // k_enc_generic.hip
void launch_enc_best(const EncArgs &a, const EncPlan &plan, hipStream_t st);       // k_enc_best.hip
void launch_enc_px(const EncArgs &a, const EncPlan &plan, hipStream_t st);         // k_enc_px.hip
void launch_enc_px_best(const EncArgs &a, const EncPlan &plan, hipStream_t st);    // k_enc_px_best.hip
void launch_enc_px16(const EncArgs &a, const EncPlan &plan, hipStream_t st);       // k_enc_px16.hip
void launch_enc_pxw(const EncArgs &a, const EncPlan &plan, hipStream_t st);        // k_enc_pxw.hip
void launch_enc_post(const EncArgs &a, const EncPlan &plan, hipStream_t st);       // k_enc_post.hip: scan, concat, seams, header, ix
void launch_enc_post_strip(const EncArgs &a, const EncPlan &plan, hipStream_t st, uint32_t strip);     // ... of one strip (a scan group of chunks): scan, concat, seams
void launch_enc_post_tail(const EncArgs &a, const EncPlan &plan, hipStream_t st);  // ... behind the last strip: index positions, table, header
void launch_dec_generic(const DecArgs &a, const DecPlan &plan, hipStream_t st);    // k_dec_generic.hip: dec3_kernel / dec_kernel
void launch_dec_index_serial(const DecArgs &a, hipStream_t st);                    // k_dec_generic.hip
void launch_dec_index_walk_best(const DecArgs &a, hipStream_t st);                 // k_dec_generic.hip: plain common-factor streams of several bands: lengths by one wave, values by the parallel decoder
bool dec_index_walk_best_ok(const DecArgs &a);
void launch_dec_px(const DecArgs &a, const DecPlan &plan, hipStream_t st);         // k_dec_px.hip
void launch_dec_px16(const DecArgs &a, const DecPlan &plan, hipStream_t st);       // k_dec_px16.hip
void launch_dec_pxw(const DecArgs &a, const DecPlan &plan, hipStream_t st);        // k_dec_pxw.hip
void launch_dec_pxw_best(const DecArgs &a, const DecPlan &plan, hipStream_t st);   // k_dec_pxw.hip
void launch_dec_px_best(const DecArgs &a, const DecPlan &plan, hipStream_t st);    // k_dec_px_best.hip
void launch_dec_pxu(const DecArgs &a, const DecPlan &plan, hipStream_t st);        // k_dec_pxu.hip: lane per unit, any band count
void launch_dec_pxu_best(const DecArgs &a, const DecPlan &plan, hipStream_t st);   // k_dec_pxu.hip
size_t pxu_lds_bytes(uint32_t in_cap_dw, uint32_t tsz, bool best);
void launch_dec_walk(const DecArgs &a, hipStream_t st);                            // k_dec_walk.hip: unit lengths of an index-less 8/16-bit stream
void launch_prev_scan(const DecArgs &a, hipStream_t st);                           // k_dec_walk.hip
bool launch_dec_walk_best(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits);    // k_dec_walk.hip: plain single-band 32/64-bit common-factor streams
void launch_dec_walk_table(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits);   // k_dec_walk.hip: plain 8- and 16-bit streams

}  // namespace qb3dev
