// qb3_amd/csrc/qb3_wide.h -- pieces shared by the unit-parallel decoder (dec3_kernel, k_dec_generic.hip) and the 32/64-bit
// lane-per-block decoder (k_dec_pxw.hip): a unit's switch and its sixteen values read at a known bit position.
#pragma once
#include "qb3_kernels.h"

namespace qb3dev {

// The decode table of rungs 1..7 as a compile-time constant (layout and entries of fill_dec_tab, qb3_kernels.h: entry =
// len << 12 | value with the swap undone, rung r at [4 << r) - 8, indexed by the next r + 2 stream bits; reference
// QB3decode.h:24-95,119-129) -- copied from L2 with 16-byte loads instead of being computed by every workgroup.
struct WideDecTab { alignas(16) uint16_t e[1024]; };
constexpr WideDecTab make_wide_dec_tab() {
    WideDecTab t{};
    for (uint32_t r = 1; r < 8; r++) {
        const uint32_t top = 1u << r, half = top >> 1;
        for (uint32_t x = 0; x < (4u << r); x++) {
            uint32_t v = 0, len = 0;
            if (!(x & 1)) { v = (x & (top - 1)) >> 1; len = r; }
            else if (!(x & 2)) { v = ((x >> 2) & (half - 1)) | half; len = r + 1; }
            else { v = ((x >> 2) & (top - 1)) | top; len = r + 2; }
            if (v == top || v == top - 1) v ^= 2 * top - 1;
            t.e[(4u << r) - 8 + x] = (uint16_t)((len << 12) | v);
        }
    }
    return t;
}
static __device__ const WideDecTab wide_dec_tab = make_wide_dec_tab();

template <typename T> struct WideVec;
template <> struct WideVec<uint32_t> { typedef uint32_t v4 __attribute__((ext_vector_type(4), aligned(4))); };
template <> struct WideVec<uint64_t> { typedef uint64_t v2 __attribute__((ext_vector_type(2), aligned(8))); };
template <> struct WideVec<uint16_t> { typedef uint16_t v4 __attribute__((ext_vector_type(4), aligned(2))); };

// four values of a row at any T-aligned address
__device__ __forceinline__ void pxw_load_row(const uint32_t *p, uint32_t (&r)[4]) {
    const WideVec<uint32_t>::v4 v = *(const WideVec<uint32_t>::v4 *)p;
    r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
}
__device__ __forceinline__ void pxw_load_row(const uint16_t *p, uint16_t (&r)[4]) {
    const WideVec<uint16_t>::v4 v = *(const WideVec<uint16_t>::v4 *)p;
    r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
}
__device__ __forceinline__ void pxw_load_row(const uint64_t *p, uint64_t (&r)[4]) {
    const WideVec<uint64_t>::v2 a = *(const WideVec<uint64_t>::v2 *)p, b = *(const WideVec<uint64_t>::v2 *)(p + 2);
    r[0] = a.x; r[1] = a.y; r[2] = b.x; r[3] = b.y;
}

// The encode table of rungs 1..7 as a compile-time constant (layout and entries of fill_enc_tab, qb3_kernels.h: entry =
// len << 12 | code with the middle swap applied, rung r at [2 << r) - 4, indexed by the mag-sign value)
struct WideEncTab { alignas(16) uint16_t e[512]; };
constexpr WideEncTab make_wide_enc_tab() {
    WideEncTab t{};
    for (uint32_t r = 1; r < 8; r++) {
        const uint32_t top = 1u << r, half = top >> 1;
        for (uint32_t m = 0; m < (2u << r); m++) {
            uint32_t v = m;
            if (v == top || v == top - 1) v ^= 2 * top - 1;
            const uint32_t code = (v < half) ? (v << 1) : (v < top) ? (((v - half) << 2) | 1) : (((v - top) << 2) | 3);
            t.e[(2u << r) - 4 + m] = (uint16_t)(((r + (v >= half) + (v >= top)) << 12) | code);
        }
    }
    return t;
}
static __device__ const WideEncTab wide_enc_tab = make_wide_enc_tab();

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

// Sixteen values at a rung of 8 and above out of staged words in LDS, 32/64-bit data, WITHOUT A BRANCH: every code is cut out of a
// window of 64 (96) bits read at its bit position -- three (four) LDS reads a value in place of a bit buffer with its refills, and
// nothing a lane does depends on the form of its code (reference QB3decode.h:119-129: short r bits, middle r + 1, long r + 2).
// The staged words must be followed by WIDE_PAD_DW zero words: sixteen codes of the longest kind (65 bits) and the last window
// reach that far beyond a position inside the staged bits, whatever the stream holds.  (*end: the bit behind the unit)
template <typename T, bool STEP>
__device__ __forceinline__ void wide_values_lds(LdsWords src, uint32_t gpos, uint32_t rung, T (&g)[16], uint32_t *end) {
    static_assert(sizeof(T) >= 4, "the 8- and 16-bit kernels read their codes through the tables");
    uint32_t p = gpos, rb = 0;
#pragma unroll
    for (uint32_t i = 0; i < 16; i++) {
        const uint32_t w = p >> 5, sh = p & 31;
        const uint32_t w0 = src[w], w1 = src[w + 1], w2 = src[w + 2];
        const uint32_t x0 = __builtin_amdgcn_alignbit(w1, w0, sh), x1 = __builtin_amdgcn_alignbit(w2, w1, sh);
        const uint32_t b0 = x0 & 1, b1 = (x0 >> 1) & b0;           // b0: middle or long form, b1: long
        const uint32_t plen = 1 + b0, k = rung - 1 + b1;            // prefix bits, value bits
        T v;
        if constexpr (sizeof(T) == 8) {
            const uint32_t x2 = __builtin_amdgcn_alignbit(src[w + 3], w2, sh);
            const uint64_t f = (uint64_t)__builtin_amdgcn_alignbit(x1, x0, plen) | (uint64_t)__builtin_amdgcn_alignbit(x2, x1, plen) << 32;
            v = (f & ((1ull << k) - 1)) | ((uint64_t)b0 << k);      // k <= 63
        } else {
            v = (__builtin_amdgcn_alignbit(x1, x0, plen) & ((1u << k) - 1)) | (b0 << k);     // k <= 31
        }
        g[i] = v;
        rb |= (uint32_t)((v >> rung) & 1) << i;
        p += plen + k;
    }
    if (STEP && (rb & (rb + 1)) == 0) {             // undo the step (reference QB3decode.h:285-289)
        const uint32_t m = __popc(rb);
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) if (i == m) g[i] ^= (T)((T)1 << rung);
    }
    *end = p;
}

// decodes the 16 values at bit `gpos`; run[i] = sum of the first i+1 deltas in curve order.
// Rungs 1..7 go through the LDS table (one read per value, three values per refill of the bit buffer).
// (*end, when asked for: the bit behind the unit)
// WINDOW (32/64-bit data staged in LDS with WIDE_PAD_DW zero words behind): rungs 8 and above by wide_values_lds
template <typename T, bool STEP, typename PTR, bool WINDOW = false>
__device__ __forceinline__ void dec3_group(PTR src, uint32_t endw, uint32_t gpos, uint32_t rung, const uint16_t *dtab, T (&run)[16], uint32_t *end = nullptr) {
    ReaderT<PTR> rd;
    rd.in = src; rd.endw = endw; rd.wp = gpos >> 5;
    const uint32_t sh = gpos & 31;
    uint32_t endbit = 0;
    if (rung >= 1 && rung < 8) {
        rd.buf = (uint64_t)(rd.load(rd.wp++) >> sh); rd.n = 32 - sh;
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
        endbit = (uint32_t)rd.position();
    } else if constexpr (WINDOW) {
        if (rung) wide_values_lds<T, STEP>(src, gpos, rung, run, &endbit);
        else {                                      // rung 0: a flag, then sixteen bits if it is set (reference QB3decode.h:150-160)
            rd.buf = (uint64_t)(rd.load(rd.wp++) >> sh); rd.n = 32 - sh;
            const uint32_t bits = rd.get(1) ? rd.get(16) : 0;
#pragma unroll
            for (uint32_t i = 0; i < 16; i++) run[i] = (T)((bits >> i) & 1);
            endbit = (uint32_t)rd.position();
        }
    } else {
        rd.buf = (uint64_t)(rd.load(rd.wp++) >> sh); rd.n = 32 - sh;
        get_group<T, STEP, ReaderT<PTR>>(rd, rung, run);
        endbit = (uint32_t)rd.position();
    }
    if (end) *end = endbit;
    T acc = 0;
#pragma unroll
    for (uint32_t i = 0; i < 16; i++) { acc = (T)(acc + smag_t<T>(run[i])); run[i] = acc; }
}

}  // namespace qb3dev
