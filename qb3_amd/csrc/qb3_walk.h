// qb3_amd/csrc/qb3_walk.h -- shared by the plain-stream walks (k_dec_walk.hip: dispatch, the walk from a container's table;
// k_dec_walk_chain.hip: one lane follows a table of "where a unit starting here would end"; k_dec_walk_exit.hip: exits of
// super-windows composed): reading code lengths by position, the walk's state between slabs, the probe of a stream's first segment
#pragma once
#include "qb3_kernels.h"
#include <type_traits>

namespace qb3dev {

// window of a lane in dwords (a multiple of 4: 16-byte LDS stores): 144 bytes for 8- and 16-bit data (measured: 176 bytes
// 0.336 ms on config 2, 144 bytes 0.326, 128 bytes 0.68 -- the window must leave room to walk after the longest step),
// 176 for 32- and 64-bit data, whose longest unit alone is 131 bytes
__host__ __device__ constexpr uint32_t walk_winp(uint32_t ub) { return ub <= 4 ? 36 : 44; }

// 64 stream bits at bit position `pos` (counted from LDS address 0): lo = bits 0..31, hi = bits 32..63
__device__ __forceinline__ void lds_bits64(uint32_t pos, uint32_t &lo, uint32_t &hi) {
    LdsWords p = lds_at((pos >> 3) & ~3u);
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
    lo = __builtin_amdgcn_alignbit(d1, d0, pos);
    hi = __builtin_amdgcn_alignbit(d2, d1, pos);
}
// n codes at the low end of b (8-bit data: three codes are at most 27 bits); returns the bits they take
template <int N> __device__ __forceinline__ uint32_t walk_codes(uint32_t b, uint32_t K) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const uint32_t len = __builtin_amdgcn_ubfe(K, b << 2, 4);     // code length by the low three bits
        b >>= len; acc += len;
    }
    return acc;
}
// rung switch at the low end of x: bits taken; rung updated; bad set on the signal code.  Branch free: lanes differ
// from unit to unit in which of the four forms they meet (reference QB3decode.h:97-116, the code at rung UB - 1).
template <uint32_t UB> __device__ __forceinline__ uint32_t walk_switch(uint32_t x, uint32_t &rung, bool &bad) {
    constexpr uint32_t UMASK = (1u << UB) - 1, NRUNG = 1u << UB, r = UB - 1, half = 1u << (r - 1), top = 1u << r;
    const uint32_t b0 = x & 1, y = x >> 1, c1 = y & 1, c2 = (y >> 1) & 1, t = y >> 2;
    const uint32_t m0 = (y & (top - 1)) >> 1, m1 = (t & (half - 1)) | half, m2 = (t & (top - 1)) | top;
    const uint32_t m = c1 ? (c2 ? m2 : m1) : m0;
    const uint32_t len = r + c1 + (c1 & c2);
    const uint32_t dpos = (m >> 1) + 1, dneg = (NRUNG - ((m + 1) >> 1)) & UMASK;
    const uint32_t delta = (m & 1) ? dneg : dpos;
    bad = bad || (b0 && m == NRUNG - 2);        // signal: a common-factor stream, not for this walker
    rung = (rung + (b0 ? delta : 0u)) & UMASK;
    return b0 ? 1 + len : 1u;
}
// length of the unit that starts at LDS bit position rp
template <uint32_t UB> __device__ __forceinline__ uint32_t walk_unit(uint32_t rp, uint32_t &rung, bool &bad) {
    uint32_t lo, hi;
    lds_bits64(rp, lo, hi);
    const uint32_t cs = walk_switch<UB>(lo, rung, bad);
    // rung 0: one flag, then 16 raw bits.  (Taken by select, not by branch: the code walk below then runs over the same
    // bits with lengths of at most two and its result is dropped.)
    const uint32_t len0 = cs + ((__builtin_amdgcn_alignbit(hi, lo, cs) & 1) ? 17 : 1);
    if (UB == 3) {
        const uint32_t K = rung * 0x11111111u + 0x20102010u;    // 4-bit fields by the low three bits: r, r+1, r, r+2, ...
        // read 1: switch + 2 codes (at most 5 + 18 bits) from lo, 3 codes from the next 32 bits; reads 2, 3: 3 + 3, 3 + 2
        uint32_t used = cs + walk_codes<2>(lo >> cs, K);
        used += walk_codes<3>(__builtin_amdgcn_alignbit(hi, lo, used), K);
        uint32_t q = rp + used;
        lds_bits64(q, lo, hi);
        used = walk_codes<3>(lo, K);
        used += walk_codes<3>(__builtin_amdgcn_alignbit(hi, lo, used), K);
        q += used;
        lds_bits64(q, lo, hi);
        used = walk_codes<3>(lo, K);
        used += walk_codes<2>(__builtin_amdgcn_alignbit(hi, lo, used), K);
        return rung ? q + used - rp : len0;
    } else if (UB >= 5) {
        // 32- and 64-bit data: a code is up to 65 bits long but its length is still in its two low bits: one read a code
        uint32_t q = rp + cs;
#pragma unroll 4
        for (int i = 0; i < 16; i++) {
            const uint32_t b = lds_bits(q);
            q += rung + (b & 1) + ((b & 3) == 3);
        }
        return rung ? q - rp : len0;
    } else {
        // 16-bit data: a code is at most 17 bits, three fit a 64-bit read (51 bits; the first read also holds the switch)
        // (lengths up to 17 do not fit the 4-bit fields of K: byte fields by the low two bits: r, r+1, r, r+2)
        const uint32_t kr = rung * 0x01010101u + 0x02000100u;
        uint64_t b = (((uint64_t)hi << 32) | lo) >> cs;
        uint32_t q = rp + cs;
#pragma unroll
        for (int g = 0; g < 6; g++) {
            if (g) { lds_bits64(q, lo, hi); b = ((uint64_t)hi << 32) | lo; }
            uint32_t acc = 0;
#pragma unroll
            for (int i = 0; i < (g == 5 ? 1 : 3); i++) {
                const uint32_t len = __builtin_amdgcn_ubfe(kr, ((uint32_t)b & 3u) << 3, 8);
                b >>= len; acc += len;
            }
            q += acc;
        }
        return rung ? q - rp : len0;
    }
}

struct WalkState { uint64_t P; uint32_t gb, rungs, bad, pad; };           // a tile's walk between two slabs
struct WalkState16 { uint64_t P, unit, rungs; uint32_t bad, pad; uint64_t cf; uint32_t cfs[16], cf_any, pad2; };   // a tile's walk between two slabs (rungs: 4 bits a band; cf: the exit walk of common-factor streams, the factor in force behind the first segment)

// The first index segment of every tile, parsed outright by one lane: unit lengths, the segment's entry, the band of rungs
// [R0, R0 + 16) for the table (WalkState16::pad) and the walk's entry state behind the segment.
template <typename T, int MODE>
__global__ void __launch_bounds__(64) walk_probe_kernel(const DecArgs a0, WalkState16 *states, uint32_t nr, uint32_t few = 0) {
    const DecArgs a = dec_for_tile(a0, blockIdx.x);
    constexpr uint32_t UB = UBits<T>::v, NRUNG = 1u << UB, MAXU = UB + 2 + 16 * (NRUNG + 1), STAGE = 2048;     // (dwords of the stream's head staged in LDS)
    __shared__ uint32_t stage[STAGE + 4], s_rung[MAXBANDS];
    __shared__ uint64_t s_pcf[MAXBANDS], s_tot[MAXBANDS];                  // (per-band state: indexed at run time, so not in registers)
    const uint32_t B = a.g.bands, NB = a.g.seg_blocks, lane = threadIdx.x;
    // (few: the exit walks only want the band of rungs and a state to start from -- their unit lanes parse the rest of the segment)
    const uint64_t nblocks = a.g.nblocks, nb0 = nblocks < NB ? nblocks : NB, nb = few && few < nb0 ? few : nb0;
    // the first segment is parsed from LDS when it is sure to fit (a lane reading global memory waits a round trip per word)
    const bool staged = nb * B * MAXU + 64 <= 32ull * STAGE;
    const uint64_t w0 = a.in_bit0 >> 5, endw = (a.in_bit0 + a.in_bits + 31) >> 5;
    if (staged) for (uint32_t i = lane; i < STAGE + 4; i += 64) stage[i] = w0 + i < endw ? a.in32[w0 + i] : 0u;
    if (lane < MAXBANDS) { s_rung[lane] = 0; s_pcf[lane] = 0; s_tot[lane] = 0; }
    __syncthreads();
    if (lane) return;
    uint32_t minr = NRUNG, maxr = 0;
    bool ok = true;
    a.idx.bitpos[0] = 0;
    for (uint32_t c = 0; c < B; c++) { a.idx.rung[c] = 0; if (MODE == CM_BEST) ((T *)a.idx.cf)[c] = 0; }
    uint64_t P_end = 0;
    auto run = [&](auto &rd, uint64_t origin) {                            // origin: position() of the stream's first bit
        T g[16];
        uint32_t bt = 0;
        uint64_t b0 = 0;
        for (uint64_t gb = 0; gb < nb && ok; gb++)
            for (uint32_t c = 0; c < B; c++) {
                const uint64_t u0 = rd.position();
                uint32_t rg = s_rung[c];
                const uint32_t rg_in = rg;
                T pc = (T)s_pcf[c];
                ok = parse_unit<T, MODE>(rd, rg, pc, g) && ok;             // (FTL / BASE: lengths and rungs are the same with and without the step)
                s_rung[c] = rg; s_pcf[c] = (uint64_t)pc;
                if (a.g.ulen_sz == 2) ((uint16_t *)a.idx.ulen)[gb * B + c] = (uint16_t)(rd.position() - u0);
                else if (a.g.ulen_sz == 1) ((uint8_t *)a.idx.ulen)[gb * B + c] = (uint8_t)(rd.position() - u0);
                else if (a.g.ulen_sz == 4) {                               // (block table of the 8-bit common-factor decoder: the block's bits | its units' entering rungs)
                    if (c == 0) { bt = 0; b0 = u0; }
                    if (c < 4) bt |= (rg_in & (sizeof(T) >= 4 ? 63u : 15u)) << (16 + 4 * c);
                    if (c + 1 == B) ((uint32_t *)a.idx.ulen)[gb] = bt | (uint32_t)((rd.position() - b0) & 0xffffu);
                }
                if (MODE == CM_BEST) {                                     // (common-factor streams: the segment's sum, for the scan that gives every segment its entering value)
                    T t = (T)s_tot[c];
                    for (uint32_t i = 0; i < 16; i++) t = (T)(t + smag_t<T>(g[i]));
                    s_tot[c] = (uint64_t)t;
                }
                if (gb || nb == 1) { minr = rg < minr ? rg : minr; maxr = rg > maxr ? rg : maxr; }
            }
        P_end = rd.position() - origin;
    };
    if (staged) { ReaderT<LdsWords> rd; rd.init((LdsWords)stage, a.in_bit0 & 31, 32ull * (STAGE + 4)); run(rd, (uint64_t)(a.in_bit0 & 31)); }
    else { Reader rd; rd.init(a.in32, a.in_bit0, a.in_bit0 + a.in_bits); run(rd, (uint64_t)a.in_bit0); }
    if (MODE == CM_BEST) for (uint32_t c = 0; c < B; c++) ((T *)a.idx.prev)[c] = (T)s_tot[c];
    WalkState16 *S = states + blockIdx.x;
    // the band: nr rungs from a little below the smallest rung the first segment saw.  What lies ABOVE the typical rung matters
    // more than what lies below: the first unit of every block row is entered from the far end of the row before and sits
    // log2(row length) rungs above its neighbours.  (The stream's very first units, entered from zero, are not looked at.)
    uint32_t R0 = minr >= 3 ? minr - 3 : 0;
    if (R0 > NRUNG - nr) R0 = NRUNG - nr;
    uint64_t rel = 0;
    for (uint32_t c = 0; c < B; c++) { const uint32_t d = s_rung[c] - R0; ok = ok && d < nr; rel |= (uint64_t)(d & 15u) << (4 * c); }
    uint64_t cfs = s_pcf[0];                                               // (several bands, 8-bit data: a byte a band)
    if (B > 1) { cfs = 0; for (uint32_t c = 0; c < B && c < 8; c++) cfs |= (s_pcf[c] & 0xffull) << (8 * c); }
    S->P = P_end; S->unit = nb * B; S->rungs = rel; S->pad = R0; S->bad = ok ? 0u : 1u; S->cf = cfs;
    if (!ok) atomicOr(a.status, 1u);
}

// ---- what the dispatcher (launch_dec_walk_table, k_dec_walk.hip) calls
// k_dec_walk_chain.hip: the table + chain walks; walk_cw: positions of a chain window (what the table memory is counted in)
bool walk_chain_lds_ok();
uint32_t walk_cw(uint32_t tsz);
uint32_t walk_win_bytes(uint32_t tsz);          // ... and the table bytes of one window
void walk_chain_8bit(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits);
void walk_chain_16bit(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits);      // (also common-factor streams of several bands)
void walk_chain_8bit_any(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits);   // 8-bit streams of any band count, FTL / BASE / common factor: the 16-bit chain's kernels with eight rungs
void walk_chain_wide(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits);      // (probes the first segment itself)
// k_dec_walk_exit.hip: the walks by exits; false: not taken (no table memory for them)
bool walk_exit_lds_ok();
size_t walk_exit_bytes(uint32_t tsz, uint32_t bands, bool best, uint32_t nt, uint64_t max_bits);
bool walk_exits_one_band(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits);  // FTL / BASE, one band of any width
bool walk_exits_rgb(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits);       // FTL / BASE, 8-bit RGB

}  // namespace qb3dev
