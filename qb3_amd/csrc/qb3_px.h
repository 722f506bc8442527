// qb3_amd/csrc/qb3_px.h -- pieces shared by the lane-per-block kernels (k_enc_px.hip, k_enc_px16.hip, k_dec_px.hip,
// k_dec_px16.hip): compile-time code tables, curve helpers, SWAR arithmetic.
#pragma once
#include "qb3_kernels.h"

namespace qb3dev {

// packed 16-bit arithmetic (two values per register)
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

// ------------------------------------------------------------------ 8-bit, 1/3/4 bands: lane per BLOCK, in registers
// Specialisation of enc_kernel for the common rasters (uint8, grey / RGB / RGBA, width a multiple of 4, identity
// or default R-G,G,B-G band map, Hilbert or Z curve).  Same bit stream, different organisation.  The kernel is
// bound by instruction issue and memory latency, not by HBM bandwidth, so it is written for instruction count:
//   * a lane owns a whole block: it loads the four rows of the block straight from HBM (B dwords per row: 64
//     lanes x 4*B bytes are one contiguous run, so the loads are coalesced without an LDS tile) plus the one
//     dword that holds the previous block's last visited pixel;
//   * band count and curve are template parameters: v_perm_b32 gathers each band's bytes in curve order, four to
//     a register, and band difference, running delta and mag-sign are byte-parallel (SWAR) on those registers;
//   * the code table is a compile-time constant copied from L2; a unit's bit string is six pieces of at most 27 bits,
//     each built BACKWARDS with one v_alignbit_b32 per value (the entry holds the code left-aligned and the shift count
//     in its low five bits) and its length comes from the low byte of the sum of the entries;
//   * rungs of the neighbouring block come from the neighbouring lane (DPP wave shift, LDS only across waves);
//     lane 0 of the workgroup is the halo block (computes rungs only), so a chunk is 255 blocks;
//   * one workgroup scan per chunk (block bit lengths, DPP), one 32-bit LDS bit writer per lane.
constexpr uint32_t order_nib(uint64_t order, int i) { return (uint32_t)(order >> (60 - 4 * i)) & 15u; }
// core band of band c under the default map: R-G, G, B-G (, A)   (reference QB3encode.cpp:41-45)
template <int B, bool RGB> constexpr int core_of(int c) { return (RGB && (c == 0 || c == 2)) ? 1 : c; }

// Encode table of the px kernels, built at compile time: rung r (1..7) at entries [2<<r, 4<<r), indexed by the mag-sign
// value, middle swap applied (reference QB3encode.h:30-33, 132-141).  An entry is made for ONE instruction per value: the
// code left-aligned in the dword, and in the low five bits 32 - length -- v_alignbit_b32(acc, e, e) shifts (acc : e) right
// by 32 - length, which is (acc << length) | code (a unit's bit string is built backwards, last value first); the sum of
// the entries' low bytes gives the piece's length: 32 * values - sum (a code is 1 to 9 bits, a piece at most three values).
struct PxEncTab { alignas(16) uint32_t e[512]; };
constexpr PxEncTab make_px_enc_tab() {
    PxEncTab t{};
    for (uint32_t r = 1; r < 8; r++) {
        const uint32_t top = 1u << r, half = top >> 1;
        for (uint32_t m = 0; m < (2u << r); m++) {
            uint32_t v = m;
            if (v == top || v == top - 1) v ^= 2 * top - 1;
            const uint32_t code = (v < half) ? (v << 1) : (v < top) ? (((v - half) << 2) | 1) : (((v - top) << 2) | 3);
            const uint32_t len = r + (v >= half) + (v >= top);
            t.e[(2u << r) + m] = (code << (32 - len)) | (32 - len);
        }
    }
    return t;
}
static __device__ const PxEncTab px_enc_tab = make_px_enc_tab();
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
static __device__ const PxDecTab px_dec_tab = make_px_dec_tab();

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
// (*end, when asked for: the bit behind the unit)
template <bool STEP>
__device__ __forceinline__ uint32_t px_group(uint32_t gpos, uint32_t rung, uint32_t (&rp)[8], uint32_t *end = nullptr) {
    uint32_t acc = 0;
    if (rung == 0) {
        const uint32_t x = lds_bits(gpos);
        const uint32_t bits = (x & 1) ? (x >> 1) & 0xffffu : 0u;
        if (end) *end = gpos + ((x & 1) ? 17 : 1);
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
    if (end) *end = pos;
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

}  // namespace qb3dev
