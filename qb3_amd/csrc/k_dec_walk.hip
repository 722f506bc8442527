// qb3_amd/csrc/k_dec_walk.hip -- index-less streams: find the unit lengths (and segment entries) by walking
#include "qb3_kernels.h"
#include <type_traits>

namespace qb3dev {

// ---- foreign streams: rebuild the index without decoding values ---------------------------------------------
// The stream has no restart points, so unit positions can only be found by walking it; what CAN be parallel is
// everything else.  The walks here find unit LENGTHS only (a code's length is its rung plus what its low two bits say,
// reference QB3decode.h:119-129) and each segment's bit position and rungs: from the restart points of the container's
// own table (dec_walk_lanes_kernel, a lane per entry), else through tables of "where would a unit starting at this bit
// with this rung end" made by the whole chip -- followed by one lane (the chains) or composed as exits of super-windows.
// The values entering the segments then come from the parallel decoder itself: one pass in TOTALS mode leaves every
// segment's per-band sum in idx.prev, prev_scan_kernel turns the sums into exclusive prefixes, the normal pass follows.
// Without memory for a table (or under QB3_SLOW_WALK) a stream goes to the one-lane parser of k_dec_generic.hip
// (dec_index_staged): the one last-resort walk of every width and mode (round 4 retired the one-wave length walk that
// 8/16-bit streams had beside it).
// idx.prev holds every segment's per-band sum of values: make it the value entering the segment (exclusive prefix,
// modulo the value width; a stream starts from zero).  One workgroup per tile.
template <typename T>
__global__ void __launch_bounds__(1024) prev_scan_kernel(const DecArgs a0) {
    const DecArgs a = dec_for_tile(a0, blockIdx.x);
    typedef typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type S;
    __shared__ S part[1024];
    const uint32_t tid = threadIdx.x, B = a.g.bands, c = blockIdx.y;       // one workgroup per tile and band
    const uint64_t nseg = a.g.nseg, per = (nseg + 1023) / 1024;
    const uint64_t s0 = (uint64_t)tid * per, s1 = (s0 + per < nseg) ? s0 + per : nseg;
    T *prev = (T *)a.idx.prev;
    S sum = 0;
    for (uint64_t s = s0; s < s1; s++) sum += prev[s * B + c];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {               // inclusive scan of the partial sums
        const S y = tid >= d ? part[tid - d] : (S)0;
        __syncthreads();
        part[tid] += y;
        __syncthreads();
    }
    S run = part[tid] - sum;
    for (uint64_t s = s0; s < s1; s++) { const S t = prev[s * B + c]; prev[s * B + c] = (T)run; run += t; }
}

// ---- the same walk from the container's restart table: ONE LANE per restart point ---------------------------
// With the "ix" chunks in the container (include/qb3x.h) the stream is cut into K independent walks, one index
// segment each where the lane-per-block decoders apply.  A lane walks its own piece; lanes of a wave run in lockstep
// because a unit is always a switch and sixteen codes whatever its rung (only rung 0 takes a short side path).
// The kernel is bound by instruction issue, so it is written for instructions per unit: every lane keeps a 256-byte
// WINDOW of its piece in LDS (sixteen-byte loads, re-centred for all lanes whenever one of them runs low) and reads
// bits by POSITION -- three dependent 64-bit reads per 8-bit unit, no bit buffer to maintain, no divergent refills;
// a code costs four vector instructions (shift, length from a per-rung constant, shift, add).
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

template <uint32_t UB, int BT>     // BT: bands at compile time (8-bit data), 0: run time
__global__ void __launch_bounds__(64) dec_walk_lanes_kernel(const DecArgs a0, const uint32_t stage_bytes) {
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    // LDS: the 64 windows, then (stage_bytes per lane, 0: none) the unit lengths the wave finds, laid out like the
    // part of the length table they belong to: a lane's lengths are scattered bytes, the wave's are one contiguous run
    // that leaves as whole cache lines at the end
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr uint32_t WALK_WINP = walk_winp(UB);
    uint32_t *win = (uint32_t *)smem;
    uint8_t *ul_s = smem + 64 * WALK_WINP * 4;
    constexpr uint32_t USZ = UB == 3 ? 1 : 2;                               // bytes per unit length
    constexpr uint32_t MAXU = UB + 2 + 16 * ((8u << (UB - 3)) + 1);        // longest unit: 149 bits (8-bit), 278 (16-bit), 535, 1048
    const uint32_t lane = threadIdx.x, k = blockIdx.x * 64 + lane;
    const uint32_t B = BT ? (uint32_t)BT : a.g.bands, NB = a.g.seg_blocks, nblocks = (uint32_t)a.g.nblocks;
    const bool live = k < a.ix_K;
    const uint32_t kk = live ? k : a.ix_K - 1;
    const uint8_t *e = ix_entry_at(a.ix, a.ix_per_chunk, a.ix_E, a.ix_pad, kk);
    uint64_t bp = 0;
#pragma unroll
    for (uint32_t i = 0; i < 6; i++) bp |= (uint64_t)e[i] << (8 * i);
    const uint32_t gb0 = kk * a.ix_blocks;
    const uint32_t nb = !live ? 0u : (nblocks - gb0 < a.ix_blocks ? nblocks - gb0 : a.ix_blocks);
    const uint32_t nu = nb * B;                                             // units to walk
    uint64_t seg = gb0 / NB;
    const uint64_t endw = (a.in_bit0 + a.in_bits + 31) >> 5;
    uint32_t *mywin = win + lane * WALK_WINP;
    const uint32_t lbit = 8 * (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)mywin;   // LDS bit address of the window
    // a step walks a block (8-bit, bands known) or a unit; it may start while the window still holds its longest case
    constexpr uint32_t STEP_UNITS = BT ? (uint32_t)BT : 1u;
    constexpr uint32_t STEP_BITS = STEP_UNITS * MAXU + 96;                   // + the three dwords of the last read
    const uint32_t rp_limit = lbit + 32 * WALK_WINP - STEP_BITS;
    uint64_t P = a.in_bit0 + bp;                                            // bit position, from a.in32
    bool bad = false;
    uint32_t u = 0;                                                         // units done
    uint32_t rungs[BT ? BT : 1];
    uint64_t R = 0;                                                         // run-time bands: rungs, 4 bits per band
    uint8_t *wide_rung = ul_s + (stage_bytes ? 64 * 64 : 0) + lane * MAXBANDS;         // ... 32/64-bit data (rungs up to 63): a byte per band, in LDS
    if (BT) {
#pragma unroll
        for (int c = 0; c < (BT ? BT : 1); c++) rungs[c] = e[6 + c] & 15u;
    } else if (UB <= 4) {
        for (uint32_t c = 0; c < B; c++) R |= (uint64_t)(e[6 + c] & 15u) << (4 * c);
    } else
        for (uint32_t c = 0; c < B; c++) wide_rung[c] = e[6 + c];
    uint32_t band = 0, blk = 0;                                             // run-time bands: position inside the block
    while (__any(u < nu)) {
        // ---- (re)centre every lane's window on its position: sixteen-byte loads, nothing read beyond the stream
        const uint64_t wb = P >> 5;
        if (__all(wb + WALK_WINP <= endw)) {        // (all but the last wave of the stream)
#pragma unroll
            for (uint32_t q = 0; q < WALK_WINP / 4; q++) {
                const u32x4_a4 t = *(const u32x4_a4 *)(a.in32 + wb + 4 * q);
                *(uint4 *)(mywin + 4 * q) = make_uint4(t.x, t.y, t.z, t.w);
            }
        } else {
#pragma unroll 1
            for (uint32_t q = 0; q < WALK_WINP; q++) mywin[q] = wb + q < endw ? a.in32[wb + q] : 0u;
        }
        // a lane reads what it wrote itself: LDS operations of a wave execute in order, the fences are for the compiler
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t rp = lbit + (uint32_t)(P - 32 * wb);
        // ---- walk while every lane that still has units has room for a step
        while (true) {
            const bool left = u < nu;
            if (!__any(left) || __any(left && rp > rp_limit)) break;
            if (left) {
                if (BT) {
                    constexpr int BB = BT ? BT : 1;
                    const uint32_t b = u / BB;                              // block of the entry
                    if (b % NB == 0) {
                        a.idx.bitpos[seg] = 32 * wb + (rp - lbit) - a.in_bit0;
#pragma unroll
                        for (int c = 0; c < BB; c++) a.idx.rung[seg * BB + c] = (uint8_t)rungs[c];
                        if (b == 0)                 // the entering values of the entry's first segment are in the entry
#pragma unroll
                            for (int c = 0; c < BB; c++) ((uint8_t *)a.idx.prev)[seg * BB + c] = e[6 + BB + c];
                        seg++;
                    }
                    uint32_t pk = 0;
#pragma unroll
                    for (int c = 0; c < BB; c++) {
                        const uint32_t ulen = walk_unit<UB>(rp, rungs[c], bad);
                        rp += ulen;
                        pk |= ulen << (8 * c);
                    }
                    if (stage_bytes) {
                        // sixteen blocks of lengths collect in the lane's 16 * BB bytes of LDS, then leave as BB sixteen-byte
                        // stores to the lane's place in the length table (a whole segment staged per lane cost 12 KB a wave
                        // and with it half the resident waves)
                        uint8_t *ul = ul_s + lane * (16 * BB) + (b & 15u) * BB;
                        if (BB == 4) *(uint32_t *)ul = pk;
                        else
#pragma unroll
                            for (int c = 0; c < BB; c++) ul[c] = (uint8_t)(pk >> (8 * c));
                        if ((b & 15u) == 15u) {
                            uint8_t *g = (uint8_t *)a.idx.ulen + ((uint64_t)gb0 + (b & ~15u)) * BB;
#pragma unroll
                            for (int q = 0; q < BB; q++) {
                                const uint4 v = *(const uint4 *)(ul_s + lane * (16 * BB) + 16 * q);
                                const u32x4_a4 t = { v.x, v.y, v.z, v.w };
                                *(u32x4_a4 *)(g + 16 * q) = t;
                            }
                        }
                    } else {
                        uint8_t *ul = (uint8_t *)a.idx.ulen + ((uint64_t)gb0 + b) * BB;
#pragma unroll
                        for (int c = 0; c < BB; c++) ul[c] = (uint8_t)(pk >> (8 * c));
                    }
                    u += BB;
                } else {
                    if (band == 0 && blk % NB == 0) {
                        a.idx.bitpos[seg] = 32 * wb + (rp - lbit) - a.in_bit0;
                        for (uint32_t c = 0; c < B; c++) a.idx.rung[seg * B + c] = UB <= 4 ? (uint8_t)((R >> (4 * c)) & 15u) : wide_rung[c];
                        if (blk == 0) {             // the entering values of the entry's first segment are in the entry
                            constexpr uint32_t TSZ = 1u << (UB - 3);
                            const uint8_t *pv = e + 6 + B;
                            uint8_t *dst = (uint8_t *)a.idx.prev + seg * B * TSZ;
                            for (uint32_t i = 0; i < B * TSZ; i++) dst[i] = pv[i];
                        }
                        seg++;
                    }
                    uint32_t rung;
                    if (UB <= 4) rung = (uint32_t)(R >> (4 * band)) & 15u;
                    else rung = wide_rung[band];
                    const uint32_t ulen = walk_unit<UB>(rp, rung, bad);
                    rp += ulen;
                    if (UB <= 4) R = (R & ~(15ull << (4 * band))) | ((uint64_t)rung << (4 * band));
                    else wide_rung[band] = (uint8_t)rung;
                    if (stage_bytes) {          // a ring of 64 bytes of lengths per lane; a full ring leaves as four 16-byte stores
                        const uint32_t off = u * USZ;
                        uint8_t *ul = ul_s + lane * 64 + (off & 63u);
                        if (UB == 3) *ul = (uint8_t)ulen;
                        else *(uint16_t *)ul = (uint16_t)ulen;
                        if (((off + USZ) & 63u) == 0) {
                            uint8_t *g = (uint8_t *)a.idx.ulen + (uint64_t)gb0 * B * USZ + (off & ~63u);
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                const uint4 v = *(const uint4 *)(ul_s + lane * 64 + 16 * q);
                                const u32x4_a4 t = { v.x, v.y, v.z, v.w };
                                *(u32x4_a4 *)(g + 16 * q) = t;
                            }
                        }
                    } else {
                        uint8_t *ul = (uint8_t *)a.idx.ulen + (((uint64_t)gb0 + blk) * B + band) * USZ;
                        if (UB == 3) *ul = (uint8_t)ulen;
                        else *(uint16_t *)ul = (uint16_t)ulen;
                    }
                    if (++band == B) { band = 0; blk++; }
                    u++;
                }
            }
        }
        P = 32 * wb + (rp - lbit);
    }
    if (stage_bytes && BT) {            // a lane's last blocks when their count is not a multiple of sixteen
        constexpr int BB = BT ? BT : 1;
        const uint32_t done = nb & ~15u;
        for (uint32_t b = done; b < nb; b++)
            for (int c = 0; c < BB; c++) ((uint8_t *)a.idx.ulen)[((uint64_t)gb0 + b) * BB + c] = ul_s[lane * (16 * BB) + (b & 15u) * BB + c];
    } else if (stage_bytes) {           // what is left in the lane's ring
        const uint32_t total = nu * USZ, done = total & ~63u;
        for (uint32_t i = done; i < total; i++) ((uint8_t *)a.idx.ulen)[(uint64_t)gb0 * B * USZ + i] = ul_s[lane * 64 + (i & 63u)];
    }
    if (bad && live) atomicOr(a.status, 1u);
}

// ---- plain 8-bit streams (no index, no restart table): walk through a TABLE of unit lengths by position -----------
// Where a unit starts depends on every unit before it, but how long a unit WOULD be if it started at bit p with rung r
// depends on the bits alone.  walk_table_kernel computes that for every bit position of a slab of the stream and every
// rung, the whole chip at once (a code's length is its rung plus what its two low bits say, so the sixteen codes of a
// unit are four rounds of pointer doubling over "length of the next code"); walk_chain_kernel then follows the one
// chain that is real.  That walk is a pointer chase (measured on this chip: 48 cycles for a dependent LDS read, and
// about 8 more for every instruction between the value read and the next address), so the table is written in the
// form that makes the value read BE the next address: in windows of CW positions, a row of eight 16-bit entries per
// position, entry[o][r_in] = 16 * (o + unit length) | 2 * (rung after the unit's switch) | signal, o counted from
// the window's start.  16 * o' is the LDS offset of row o' in the window's buffer: one AND-OR with the buffer's base
// and the rung of the band that comes next gives the address of the next look-up.
namespace chain {
constexpr uint32_t CW = 3072;                           // positions at which the blocks of a window start
constexpr uint32_t ROWS = CW + 576;                     // ... and those their later units can start at (3 x 149 bits), in 3 x 64 rows for the loaders
constexpr uint32_t WIN_BYTES = ROWS * 16, SLOT = 65536; // a window in LDS: its rows, in a slot whose base has no bit below 2^16
constexpr uint32_t NSLOT = 2;                           // windows in LDS: one walked, one on its way
constexpr uint32_t TR_ENTRIES = CW / 2 + 16, TR_BYTES = TR_ENTRIES * 2;     // trail of a window: a unit is at least two bits
constexpr uint32_t TR0 = NSLOT * SLOT, META = TR0 + NSLOT * TR_BYTES, LDS_BYTES = META + 128;
static_assert(ROWS + 149 <= 4096 && WIN_BYTES <= SLOT && ROWS % 192 == 0 && ROWS >= CW + 448 && TR_BYTES % 16 == 0 && CW == 0xc00, "window layout (the walk tests position >= CW by its two top bits)");
constexpr uint32_t NP = ROWS + 160;                     // positions a table workgroup looks at: sixteen codes beyond the last switch
static_assert(NP % 32 == 0, "whole words");
}  // namespace chain
struct WalkState { uint64_t P; uint32_t gb, rungs, bad, pad; };           // a tile's walk between two slabs

__global__ void __launch_bounds__(256) walk_table_kernel(const DecArgs a0, uint4 *tab, uint64_t slab0, uint32_t nwin, uint64_t tab_pitch) {
    using namespace chain;
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    const uint64_t p0 = slab0 + (uint64_t)blockIdx.x * CW;
    if (p0 >= a.in_bits + 2 * CW) return;                                   // (uniform) far beyond the stream: no walk comes here
    __shared__ uint32_t words[NP / 32 + 3];
    __shared__ uint8_t nA[7][NP], nB[7][NP];
    const uint32_t tid = threadIdx.x;
    const uint64_t q0 = a.in_bit0 + p0, w0 = q0 >> 5, endw = (a.in_bit0 + a.in_bits + 31) >> 5;
    const uint32_t sh = (uint32_t)q0 & 31;
    for (uint32_t i = tid; i < NP / 32 + 3; i += 256) words[i] = w0 + i < endw ? a.in32[w0 + i] : 0u;
    __syncthreads();
    auto bits = [&](uint32_t i) { const uint32_t b = sh + i, k = b >> 5; return __builtin_amdgcn_alignbit(words[k + 1], words[k], b & 31); };
    for (uint32_t i = tid; i < NP; i += 256) {                              // one code (reference QB3decode.h:119-129: r, r + 1 or r + 2 bits)
        const uint32_t x = bits(i), e = (x & 1) + ((x & 3) == 3);
#pragma unroll
        for (uint32_t r = 1; r < 8; r++) nA[r - 1][i] = (uint8_t)(r + e);
    }
    __syncthreads();
    uint8_t (*src)[NP] = nA, (*dst)[NP] = nB;
    uint32_t valid = NP;
#pragma unroll 1
    for (uint32_t lvl = 0; lvl < 4; lvl++) {                                // 2, 4, 8, 16 codes
        valid -= 9u << lvl;                                                 // (a code is at most nine bits)
        for (uint32_t r = 0; r < 7; r++)
            for (uint32_t i = tid; i < valid; i += 256) { const uint32_t n = src[r][i]; dst[r][i] = (uint8_t)(n + src[r][i + n]); }
        __syncthreads();
        uint8_t (*t)[NP] = src; src = dst; dst = t;
    }
    // valid = NP - 135 >= ROWS + 5: sixteen codes from every position a switch in this window can end on
    uint4 *out = tab + ((uint64_t)blockIdx.y * tab_pitch + (uint64_t)blockIdx.x * ROWS);
    for (uint32_t o = tid; o < ROWS; o += 256) {
        const uint32_t x = bits(o);
        uint32_t delta = 0; bool sig = false;
        const uint32_t cs = walk_switch<3>(x, delta, sig);                  // from rung 0: the step itself
        const uint32_t len0 = cs + (((x >> cs) & 1) ? 17 : 1);              // rung 0: one flag, then 16 raw bits
        uint32_t e[8];
#pragma unroll
        for (uint32_t rin = 0; rin < 8; rin++) {
            const uint32_t r = (rin + delta) & 7u;
            const uint32_t u = r ? cs + src[r ? r - 1 : 0][o + cs] : len0;
            e[rin] = ((o + u) << 4) | (r << 1) | (sig ? 1u : 0u);
        }
        out[o] = make_uint4(e[0] | e[1] << 16, e[2] | e[3] << 16, e[4] | e[5] << 16, e[6] | e[7] << 16);
    }
}

namespace chain {
typedef volatile __attribute__((address_space(3))) uint32_t *LdsFlag;
__device__ __forceinline__ uint32_t flag_get(uint32_t addr) { return *(LdsFlag)(uintptr_t)addr; }
__device__ __forceinline__ void flag_set(uint32_t addr, uint32_t v) { *(LdsFlag)(uintptr_t)addr = v; }
// words at META: what the waves of a workgroup tell each other (all counts of windows)
// F_READY[slot]: four words the walk reads at once: a window's three parts in LDS, and its trail slot written out
constexpr uint32_t F_READY = META /* [NSLOT][4] */, F_TRAILED = META + 32, F_GB0 = META + 40, F_NUNITS = META + 48,
                   F_WALKED = META + 56, F_STOP = META + 64;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bool ready(uint32_t slot, uint32_t want) {
    const u32x4_t f = *(volatile __attribute__((address_space(3))) u32x4_t *)(uintptr_t)(F_READY + 16 * slot);
    return f.x == want && f.y == want && f.z == want && f.w == want;
}
constexpr uint32_t SPIN_MAX = 1u << 22;                                     // (a wait that long is a defect: give up, flag the tile)

// One window: follow the chain from address A (slot base | 16 * position | 2 * rung of band 0) until a block starts
// beyond the window or the blocks run out.  T: LDS address of the trail (the address of every unit).  R[c]: slot
// base | 2 * rung of band c.  The loop is written out because the ORDER is the point -- every instruction between a
// read's value and the next read's issue costs its full latency (measured: 48 cycles the read, 8-10 each other), so
// between them stands only the AND-OR that makes the address; the trail write, the bookkeeping of the unit BEFORE
// (its band's new rung, the signal bit) and the loop's own tests all issue while a read is in flight.
#define CH_READ(E) "ds_read_u16 %[" #E "], %[A]\n"
#define CH_BOOK(E, Rc) "v_and_or_b32 %[" #Rc "], %[" #E "], 14, %[base]\n v_or_b32 %[bad], %[bad], %[" #E "]\n"
#define CH_STEP(E, Rn, off) "ds_write_b16 %[T], %[A] offset:" #off "\n s_waitcnt lgkmcnt(1)\n v_and_or_b32 %[A], %[" #E "], %[M], %[" #Rn "]\n"
#define CH_TOP "v_and_b32 %[t], 0xc000, %[A]\n v_cmp_eq_u32 vcc, 0xc000, %[t]\n"
#define CH_EXIT "s_cbranch_vccnz 2f\n s_cmp_eq_u32 %[left], 0\n s_cbranch_scc1 2f\n"
template <int B>
__device__ __forceinline__ void walk_asm(uint32_t &A, uint32_t &T, uint32_t (&R)[B], uint32_t &bad, uint32_t &left, uint32_t base) {
    uint32_t t;
    const uint32_t M = 0xfff0u;
    if constexpr (B == 3) {
        uint32_t e0, e1, e2 = R[2] & 14u;           // (the first turn books "the unit before": nothing changes)
        asm volatile(
            CH_READ(e0)
            "1:\n" CH_TOP CH_BOOK(e2, R2) CH_EXIT
            CH_STEP(e0, R1, 0)
            CH_READ(e1) CH_BOOK(e0, R0) CH_STEP(e1, R2, 2)
            CH_READ(e2) CH_BOOK(e1, R1) CH_STEP(e2, R0, 4)
            CH_READ(e0)
            "v_add_u32 %[T], 6, %[T]\n s_sub_u32 %[left], %[left], 1\n s_branch 1b\n"
            "2:\n s_waitcnt lgkmcnt(0)\n"
            : [A] "+v"(A), [T] "+v"(T), [R0] "+v"(R[0]), [R1] "+v"(R[1]), [R2] "+v"(R[2]), [bad] "+v"(bad), [left] "+s"(left),
              [e0] "=&v"(e0), [e1] "=&v"(e1), [e2] "+v"(e2), [t] "=&v"(t)
            : [M] "s"(M), [base] "v"(base)
            : "vcc", "scc", "memory");
    } else if constexpr (B == 4) {
        uint32_t e0, e1, e2, e3 = R[3] & 14u;
        asm volatile(
            CH_READ(e0)
            "1:\n" CH_TOP CH_BOOK(e3, R3) CH_EXIT
            CH_STEP(e0, R1, 0)
            CH_READ(e1) CH_BOOK(e0, R0) CH_STEP(e1, R2, 2)
            CH_READ(e2) CH_BOOK(e1, R1) CH_STEP(e2, R3, 4)
            CH_READ(e3) CH_BOOK(e2, R2) CH_STEP(e3, R0, 6)
            CH_READ(e0)
            "v_add_u32 %[T], 8, %[T]\n s_sub_u32 %[left], %[left], 1\n s_branch 1b\n"
            "2:\n s_waitcnt lgkmcnt(0)\n"
            : [A] "+v"(A), [T] "+v"(T), [R0] "+v"(R[0]), [R1] "+v"(R[1]), [R2] "+v"(R[2]), [R3] "+v"(R[3]), [bad] "+v"(bad), [left] "+s"(left),
              [e0] "=&v"(e0), [e1] "=&v"(e1), [e2] "=&v"(e2), [e3] "+v"(e3), [t] "=&v"(t)
            : [M] "s"(M), [base] "v"(base)
            : "vcc", "scc", "memory");
    } else {
        // one band: the rung that comes next is the one just read: the next address is the entry without its signal bit
        uint32_t e0;
        const uint32_t M1 = 0xfffeu;
        asm volatile(
            CH_READ(e0)
            "1:\n" CH_TOP CH_EXIT
            "ds_write_b16 %[T], %[A]\n s_waitcnt lgkmcnt(1)\n v_or_b32 %[bad], %[bad], %[e0]\n v_and_or_b32 %[A], %[e0], %[M], %[base]\n"
            CH_READ(e0)
            "v_add_u32 %[T], 2, %[T]\n s_sub_u32 %[left], %[left], 1\n s_branch 1b\n"
            "2:\n s_waitcnt lgkmcnt(0)\n"
            : [A] "+v"(A), [T] "+v"(T), [bad] "+v"(bad), [left] "+s"(left), [e0] "=&v"(e0), [t] "=&v"(t)
            : [M] "s"(M1), [base] "v"(base)
            : "vcc", "scc", "memory");
        R[0] = base | (A & 14u);
    }
}
#undef CH_READ
#undef CH_BOOK
#undef CH_STEP
#undef CH_TOP
#undef CH_EXIT
}  // namespace chain

// A workgroup per tile, eight waves: wave 0 (one lane) walks; waves 1-6 bring windows of the table into LDS -- three
// waves a window, one group the even windows and one the odd, so that a group's loads from HBM are in flight while the
// window before theirs is walked; wave 7 turns the trail of a walked window into unit lengths and segment entries.
template <int B>
__global__ void __launch_bounds__(512) walk_chain_kernel(const DecArgs a0, const uint4 *tab, uint64_t slab0, uint32_t nwin, uint64_t tab_pitch,
                                                         WalkState *states, uint32_t first_round) {
    using namespace chain;
    const DecArgs a = dec_for_tile(a0, blockIdx.x);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t NB = a.g.seg_blocks, nblocks = (uint32_t)a.g.nblocks;
    WalkState *S = states + blockIdx.x;
    const uint64_t P0 = first_round ? 0 : S->P;                             // in stream bits
    const uint32_t gb_in = first_round ? 0 : S->gb;
    const uint32_t R_in = first_round ? 0 : S->rungs;
    const uint64_t slab_end = slab0 + (uint64_t)nwin * CW;
    if (P0 < slab0 || P0 >= slab_end || P0 >= a.in_bits || gb_in >= nblocks) return;       // (uniform) nothing of this tile in this slab
    const uint32_t k0 = (uint32_t)((P0 - slab0) / CW);                      // the window the walk starts in
    if (tid < 32) {     // (the first two windows find their trail slots free)
        uint32_t v = tid == (F_STOP - META) / 4 ? 0xffffffffu : 0u;
        if (tid == (k0 % NSLOT) * 4 + 3) v = k0 + 1;
        if (tid == ((k0 + 1) % NSLOT) * 4 + 3) v = k0 + 2;
        ((uint32_t *)(smem + META))[tid] = v;
    }
    __syncthreads();
    const uint4 *wt = tab + (uint64_t)blockIdx.x * tab_pitch;

    if (wave == 0) {
        if (lane) return;
        uint32_t R[B], bad = first_round ? 0 : S->bad;
        uint32_t left = nblocks - gb_in, k = k0, o = (uint32_t)((P0 - slab0) % CW);
        for (int c = 0; c < B; c++) R[c] = ((R_in >> (4 * c)) & 7u) << 1;
        bool stuck = false;
        uint64_t Pn = P0;                                                   // where the next block starts
        while (true) {
            const uint32_t s = k % NSLOT, base = s * SLOT;
            uint32_t spin = 0;                                              // the window in LDS (three parts), and this trail slot written out
            while (!ready(s, k + 1) && ++spin < SPIN_MAX) __builtin_amdgcn_s_sleep(1);
            if (spin >= SPIN_MAX) { stuck = true; break; }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            for (int c = 0; c < B; c++) R[c] = (R[c] & 14u) | base;
            uint32_t A = base | (o << 4) | (R[0] & 14u), T = TR0 + s * TR_BYTES;
            const uint32_t T0 = T, left0 = left;
            walk_asm<B>(A, T, R, bad, left, base);
            *(volatile __attribute__((address_space(3))) uint16_t *)(uintptr_t)T = (uint16_t)A;        // where the next block starts: the last unit's end
            flag_set(F_GB0 + 4 * s, nblocks - left0);
            flag_set(F_NUNITS + 4 * s, (T - T0) >> 1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            flag_set(F_TRAILED + 4 * s, k + 1);
            const uint32_t oe = (A & 0xfff0u) >> 4;
            Pn = slab0 + (uint64_t)k * CW + oe;
            k++;
            flag_set(F_WALKED, k - k0);
            if (!left) break;                                               // the blocks ran out
            o = oe - CW;                                                    // (the walk left the window: oe >= CW)
            if (k >= nwin || Pn >= a.in_bits) break;                        // the slab ends here, or the stream does (a damaged one)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        flag_set(F_STOP, k);                                                // k windows were walked
        uint32_t Rn = 0;
        for (int c = 0; c < B; c++) Rn |= ((R[c] >> 1) & 7u) << (4 * c);
        S->P = stuck ? ~0ull : Pn; S->gb = nblocks - left; S->rungs = Rn; S->bad = (bad & 1u) | (stuck ? 1u : 0u);
        if ((bad & 1u) || stuck) atomicOr(a.status, 1u);
        return;
    }
    if (wave <= 6) {
        // loaders: group g (three waves, a third of the rows each) takes the windows of parity g, into slot g
        const uint32_t g = (wave - 1) / 3, part = (wave - 1) % 3;
        constexpr uint32_t NV = ROWS / 192;                                 // sixty-four rows a load: loads of a wave per window
        for (uint32_t k = k0 + ((k0 ^ g) & 1u); k < nwin; k += 2) {
            const uint4 *src = wt + (uint64_t)k * ROWS;
            // (named values, not an array: the array went to scratch memory)
#define CH_REP19(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18)
            static_assert(NV == 19, "CH_REP19");
#define CH_LD(i) const uint4 v##i = src[lane + 64 * (part + 3 * i)];
            CH_REP19(CH_LD)
            uint32_t spin = 0;
            bool stop = false;
            while (true) {                                                  // the slot is free when the window two back has been walked
                if (flag_get(F_STOP) != 0xffffffffu) { stop = true; break; }
                if (flag_get(F_WALKED) + NSLOT > k - k0) break;
                if (++spin >= SPIN_MAX) { stop = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (stop) break;
            uint4 *slot = (uint4 *)(smem + g * SLOT);
            // (a pause after every store: nineteen 1 KB stores back to back hold the LDS long enough to stall the walk's reads)
#define CH_ST(i) slot[lane + 64 * (part + 3 * i)] = v##i; __builtin_amdgcn_s_sleep(3);
            CH_REP19(CH_ST)
#undef CH_ST
#undef CH_LD
#undef CH_REP19
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) flag_set(F_READY + 16 * g + 4 * part, k + 1);
        }
        return;
    }
    // writer: the trail of a walked window gives the position and rung of every unit: lengths by difference
    for (uint32_t k = k0;; k++) {
        const uint32_t s = k % NSLOT;
        uint32_t spin = 0;
        bool stop = false;
        while (flag_get(F_TRAILED + 4 * s) != k + 1) {
            const uint32_t st = flag_get(F_STOP);
            if ((st != 0xffffffffu && k >= st) || ++spin >= SPIN_MAX) { stop = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (stop) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint32_t gb0 = flag_get(F_GB0 + 4 * s), n = flag_get(F_NUNITS + 4 * s);
        const uint16_t *tr = (const uint16_t *)(smem + TR0 + s * TR_BYTES);
        const uint64_t wpos = slab0 + (uint64_t)k * CW;
        uint8_t *ul = (uint8_t *)a.idx.ulen + (uint64_t)gb0 * B;
        for (uint32_t j = lane; j < n; j += 64) {
            const uint32_t o0 = tr[j] >> 4, o1 = tr[j + 1] >> 4;
            ul[j] = (uint8_t)(o1 - o0);
            if (j % B == 0 && (gb0 + j / B) % NB == 0) {
                const uint64_t seg = (gb0 + j / B) / NB;
                a.idx.bitpos[seg] = wpos + o0;
#pragma unroll
                for (int c = 0; c < B; c++) a.idx.rung[seg * B + c] = (uint8_t)((tr[j + c] >> 1) & 7u);
            }
        }
        // (no release fence: the trail has been READ -- LDS operations of a wave are in order -- and the index stores may still be
        // on their way; a fence would hold the slot for a memory round trip per window)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) flag_set(F_READY + 16 * s + 12, k + NSLOT + 1);      // the slot is free for the window that takes it next
    }
}

// ---- the same for plain 16-bit streams ---------------------------------------------------------------------------
// Sixteen rungs, codes of up to 17 bits, units of up to 278: a row is sixteen 16-bit entries (32 bytes), a window 1536
// positions (48 KB, two of them side by side in LDS: slot 1 is reached through the read's immediate offset, so the
// addresses the walk carries stay window-relative).  A block has up to 16 bands here and can be longer than a window,
// so the walk changes windows between any two UNITS (every look-up position is inside the window: no margin rows,
// 32 table bytes per stream bit); the rungs of the bands live in a small LDS array, and the trail holds the entries
// read (next position | rung out), from which a writer wave derives unit lengths and segment entries.  The walk loop is
// plain C++ here (about 1.5 x the cycles per unit of the hand-ordered 8-bit loop).
namespace chain16 {
constexpr uint32_t NR = 16, ROWB = 32, CW = 1536, WIN_BYTES = CW * ROWB, WIN_U4 = WIN_BYTES / 16;      // 49152 bytes, 3072 sixteen-byte pieces
constexpr uint32_t MAXC = 17, NP = CW + 288;            // longest code; positions a table workgroup looks at
static_assert(NP % 32 == 0 && NP >= CW + 6 + 15 * MAXC + 2 && WIN_BYTES == 0xc000 && (CW + 278) * ROWB < 65536 && WIN_U4 % 192 == 0, "16-bit window layout");
constexpr uint32_t TR_BYTES = ((CW / 2 + 8) * 2 + 15) & ~15u;              // trail of a window: a unit is at least two bits
constexpr uint32_t TR0 = 2 * WIN_BYTES, RS0 = TR0 + 2 * TR_BYTES, WR0 = RS0 + 64, META = WR0 + 64, LDS_BYTES = META + 128;
constexpr uint32_t F_READY = META /* [2][4] */, F_TRAILED = META + 32, F_NUNITS = META + 40, F_O0 = META + 48,
                   F_WALKED = META + 56, F_STOP = META + 60, F_U0 = META + 64 /* u64[2] */;
}  // namespace chain16
struct WalkState16 { uint64_t P, unit, rungs; uint32_t bad, pad; uint64_t cf; };   // a tile's walk between two slabs (rungs: 4 bits a band; cf: the exit walk of common-factor streams, the factor in force behind the first segment)

__global__ void __launch_bounds__(256) walk_table16_kernel(const DecArgs a0, uint4 *tab, uint64_t slab0, uint32_t nwin, uint64_t tab_pitch) {
    using namespace chain16;
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    const uint64_t p0 = slab0 + (uint64_t)blockIdx.x * CW;
    if (p0 >= a.in_bits + 2 * CW) return;                                   // (uniform) far beyond the stream: no walk comes here
    __shared__ uint32_t words[NP / 32 + 3];
    __shared__ uint8_t nA[NR - 1][NP], nB[NR - 1][NP];
    const uint32_t tid = threadIdx.x;
    const uint64_t q0 = a.in_bit0 + p0, w0 = q0 >> 5, endw = (a.in_bit0 + a.in_bits + 31) >> 5;
    const uint32_t sh = (uint32_t)q0 & 31;
    for (uint32_t i = tid; i < NP / 32 + 3; i += 256) words[i] = w0 + i < endw ? a.in32[w0 + i] : 0u;
    __syncthreads();
    auto bits = [&](uint32_t i) { const uint32_t b = sh + i, k = b >> 5; return __builtin_amdgcn_alignbit(words[k + 1], words[k], b & 31); };
    for (uint32_t i = tid; i < NP; i += 256) {                              // one code: r, r + 1 or r + 2 bits
        const uint32_t x = bits(i), e = (x & 1) + ((x & 3) == 3);
#pragma unroll
        for (uint32_t r = 1; r < NR; r++) nA[r - 1][i] = (uint8_t)(r + e);
    }
    __syncthreads();
    uint8_t (*src)[NP] = nA, (*dst)[NP] = nB;
    uint32_t valid = NP;
#pragma unroll 1
    for (uint32_t lvl = 0; lvl < 3; lvl++) {                                // 2, 4, 8 codes (eight codes are at most 136 bits: a byte)
        valid -= MAXC << lvl;
        for (uint32_t r = 0; r < NR - 1; r++)
            for (uint32_t i = tid; i < valid; i += 256) { const uint32_t n = src[r][i]; dst[r][i] = (uint8_t)(n + src[r][i + n]); }
        __syncthreads();
        uint8_t (*t)[NP] = src; src = dst; dst = t;
    }
    // src = eight codes, valid for i < NP - 7 * 17; sixteen = eight + eight, formed here (up to 272: not a byte)
    uint4 *out = tab + ((uint64_t)blockIdx.y * tab_pitch + (uint64_t)blockIdx.x * WIN_U4);
    for (uint32_t o = tid; o < CW; o += 256) {
        const uint32_t x = bits(o);
        uint32_t delta = 0; bool sig = false;
        const uint32_t cs = walk_switch<4>(x, delta, sig);                  // from rung 0: the step itself
        const uint32_t len0 = cs + (((x >> cs) & 1) ? 17 : 1);              // rung 0: one flag, then 16 raw bits
        uint32_t e[NR];
#pragma unroll
        for (uint32_t rin = 0; rin < NR; rin++) {
            const uint32_t r = (rin + delta) & (NR - 1);
            uint32_t u = len0;
            if (r) { const uint32_t n8 = src[r - 1][o + cs]; u = cs + n8 + src[r - 1][o + cs + n8]; }
            e[rin] = ((o + u) * ROWB) | (r << 1) | (sig ? 1u : 0u);
        }
        out[2 * o] = make_uint4(e[0] | e[1] << 16, e[2] | e[3] << 16, e[4] | e[5] << 16, e[6] | e[7] << 16);
        out[2 * o + 1] = make_uint4(e[8] | e[9] << 16, e[10] | e[11] << 16, e[12] | e[13] << 16, e[14] | e[15] << 16);
    }
}

// A workgroup per tile, eight waves: wave 0 (one lane) walks; waves 1-6 load windows (two groups of three, as in the
// 8-bit kernel); wave 7 writes unit lengths and segment entries from the trail.
__global__ void __launch_bounds__(512) walk_chain16_kernel(const DecArgs a0, const uint4 *tab, uint64_t slab0, uint32_t nwin, uint64_t tab_pitch,
                                                           WalkState16 *states, uint32_t first_round) {
    using namespace chain16;
    using chain::flag_get; using chain::flag_set; using chain::SPIN_MAX;
    const DecArgs a = dec_for_tile(a0, blockIdx.x);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t B = a.g.bands, NB = a.g.seg_blocks;
    const uint64_t nunits = a.g.nblocks * B;
    WalkState16 *S = states + blockIdx.x;
    const uint64_t P0 = first_round ? 0 : S->P, U_in = first_round ? 0 : S->unit, R_in = first_round ? 0 : S->rungs;
    const uint64_t slab_end = slab0 + (uint64_t)nwin * CW;
    if (P0 < slab0 || P0 >= slab_end || P0 >= a.in_bits || U_in >= nunits) return;         // (uniform) nothing of this tile in this slab
    const uint32_t k0 = (uint32_t)((P0 - slab0) / CW);                      // the window the walk starts in
    volatile uint32_t *rs = (volatile uint32_t *)(smem + RS0), *wr = (volatile uint32_t *)(smem + WR0);     // rung * 2 per band: the walk's, the writer's
    if (tid < 32) {     // (the first two windows find their trail slots free)
        uint32_t v = tid == (F_STOP - META) / 4 ? 0xffffffffu : 0u;
        if (tid == (k0 & 1) * 4 + 3) v = k0 + 1;
        if (tid == ((k0 + 1) & 1) * 4 + 3) v = k0 + 2;
        ((uint32_t *)(smem + META))[tid] = v;
    }
    if (tid < 16) { const uint32_t r2 = (uint32_t)((R_in >> (4 * tid)) & 15u) << 1; rs[tid] = r2; wr[tid] = r2; }
    __syncthreads();
    const uint4 *wt = tab + (uint64_t)blockIdx.x * tab_pitch;
    auto ready = [&](uint32_t slot, uint32_t want) {
        const chain::u32x4_t f = *(volatile __attribute__((address_space(3))) chain::u32x4_t *)(uintptr_t)(F_READY + 16 * slot);
        return f.x == want && f.y == want && f.z == want && f.w == want;
    };

    if (wave == 0) {
        if (lane) return;
        uint32_t bad = first_round ? 0 : S->bad, k = k0, o = (uint32_t)((P0 - slab0) % CW);
        uint64_t U = U_in, Pn = P0;
        uint32_t c = (uint32_t)(U % B);
        bool stuck = false;
        while (true) {
            const uint32_t s = k & 1;
            uint32_t spin = 0;
            while (!ready(s, k + 1) && ++spin < SPIN_MAX) __builtin_amdgcn_s_sleep(1);
            if (spin >= SPIN_MAX) { stuck = true; break; }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            uint32_t A = o * ROWB + rs[c], n = 0;
            const uint64_t left64 = nunits - U;
            uint32_t left = left64 > 0xffffffffull ? 0xffffffffu : (uint32_t)left64;
            typedef const __attribute__((address_space(3))) uint16_t *LdsHalf;
            typedef __attribute__((address_space(3))) uint16_t *LdsHalfW;
            typedef __attribute__((address_space(3))) uint32_t *LdsWordW;
            const uint32_t wbase = s * WIN_BYTES;
            LdsHalfW trw = (LdsHalfW)(uintptr_t)(TR0 + s * TR_BYTES);
            LdsWordW rsw = (LdsWordW)(uintptr_t)RS0;
            constexpr uint32_t M = 0xffe0u, RM = (NR - 1) << 1;
            // a unit per turn, until one starts beyond the window: ONE dependent LDS read a unit -- the rung of the band
            // that comes next is fetched a unit ahead (from registers for one or two bands, else from the LDS array)
            if (B == 1) {
                while (A < CW * ROWB && left) {
                    const uint32_t e = *(LdsHalf)(uintptr_t)(wbase + A);
                    trw[n++] = (uint16_t)e; bad |= e;
                    A = e & (M | RM);
                    left--;
                }
                rsw[0] = A & RM;
            } else if (B == 2) {
                uint32_t rn = rsw[c ^ 1];
                while (A < CW * ROWB && left) {
                    const uint32_t e = *(LdsHalf)(uintptr_t)(wbase + A);
                    trw[n++] = (uint16_t)e; bad |= e;
                    A = (e & M) | rn;
                    rn = e & RM;                // this band comes again after the next unit
                    c ^= 1; left--;
                }
                rsw[c] = A & RM; rsw[c ^ 1] = rn;
            } else {
                uint32_t cn = c + 1 == B ? 0 : c + 1;
                uint32_t rn = rsw[cn];
                while (A < CW * ROWB && left) {
                    const uint32_t cn2 = cn + 1 == B ? 0 : cn + 1;
                    const uint32_t e = *(LdsHalf)(uintptr_t)(wbase + A);
                    const uint32_t r2 = rsw[cn2];                           // (three bands or more: not the band being written below)
                    trw[n++] = (uint16_t)e; bad |= e;
                    rsw[c] = e & RM;
                    A = (e & M) | rn;
                    c = cn; cn = cn2; rn = r2; left--;
                }
            }
            *(volatile uint64_t *)(smem + F_U0 + 8 * s) = U;
            flag_set(F_NUNITS + 4 * s, n);
            flag_set(F_O0 + 4 * s, o);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            flag_set(F_TRAILED + 4 * s, k + 1);
            U += n;
            const uint32_t oe = A / ROWB;
            Pn = slab0 + (uint64_t)k * CW + oe;
            k++;
            flag_set(F_WALKED, k - k0);
            if (U >= nunits) break;                                         // the units ran out
            o = oe - CW;                                                    // (the walk left the window: oe >= CW)
            if (k >= nwin || Pn >= a.in_bits) break;                        // the slab ends here, or the stream does (a damaged one)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        flag_set(F_STOP, k);
        uint64_t Rn = 0;
        for (uint32_t i = 0; i < B; i++) Rn |= (uint64_t)((rs[i] >> 1) & 15u) << (4 * i);
        S->P = stuck ? ~0ull : Pn; S->unit = U; S->rungs = Rn; S->bad = (bad & 1u) | (stuck ? 1u : 0u);
        if ((bad & 1u) || stuck) atomicOr(a.status, 1u);
        return;
    }
    if (wave <= 6) {
        const uint32_t g = (wave - 1) / 3, part = (wave - 1) % 3;
        constexpr uint32_t NV = WIN_U4 / 192;
        static_assert(NV == 16, "CH16_REP");
        for (uint32_t k = k0 + ((k0 ^ g) & 1u); k < nwin; k += 2) {
            const uint4 *src = wt + (uint64_t)k * WIN_U4;
#define CH16_REP(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define CH16_LD(i) const uint4 v##i = src[lane + 64 * (part + 3 * i)];
            CH16_REP(CH16_LD)
            uint32_t spin = 0;
            bool stop = false;
            while (true) {                                                  // the slot is free when the window two back has been walked
                if (flag_get(F_STOP) != 0xffffffffu) { stop = true; break; }
                if (flag_get(F_WALKED) + 2 > k - k0) break;
                if (++spin >= SPIN_MAX) { stop = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (stop) break;
            uint4 *slot = (uint4 *)(smem + g * WIN_BYTES);
#define CH16_ST(i) slot[lane + 64 * (part + 3 * i)] = v##i;
            CH16_REP(CH16_ST)
#undef CH16_ST
#undef CH16_LD
#undef CH16_REP
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) flag_set(F_READY + 16 * g + 4 * part, k + 1);
        }
        return;
    }
    // writer: entry j of the trail = (16 * position the unit ENDS at | rung of its band after it): lengths by difference
    for (uint32_t k = k0;; k++) {
        const uint32_t s = k & 1;
        uint32_t spin = 0;
        bool stop = false;
        while (flag_get(F_TRAILED + 4 * s) != k + 1) {
            const uint32_t st = flag_get(F_STOP);
            if ((st != 0xffffffffu && k >= st) || ++spin >= SPIN_MAX) { stop = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (stop) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint64_t U0 = *(volatile uint64_t *)(smem + F_U0 + 8 * s);
        const uint32_t n = flag_get(F_NUNITS + 4 * s), o_first = flag_get(F_O0 + 4 * s);
        const uint16_t *tr = (const uint16_t *)(smem + TR0 + s * TR_BYTES);
        const uint64_t wpos = slab0 + (uint64_t)k * CW;
        uint16_t *ul = (uint16_t *)a.idx.ulen + U0;
        for (uint32_t j = lane; j < n; j += 64) {
            const uint32_t o0 = j ? tr[j - 1] / ROWB : o_first, o1 = tr[j] / ROWB;
            ul[j] = (uint16_t)(o1 - o0);
            const uint64_t Uj = U0 + j;
            if (Uj % B == 0 && (Uj / B) % NB == 0) {        // a segment starts here: position, and every band's rung as the block finds it
                const uint64_t seg = Uj / B / NB;
                a.idx.bitpos[seg] = wpos + o0;
                for (uint32_t cc = 0; cc < B; cc++) {       // band cc's unit before this one: B - cc units back
                    const int32_t jj = (int32_t)j - (int32_t)(B - cc);
                    a.idx.rung[seg * B + cc] = (uint8_t)(((jj >= 0 ? (uint32_t)tr[jj] : wr[cc]) >> 1) & 15u);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // the rung every band has after this window: the last unit of each band in it
        if (lane < B) {
            const uint32_t cl = (uint32_t)((U0 + n - 1) % B);               // band of the window's last unit
            const uint32_t back = (cl + B - lane) % B;                      // band `lane` last came `back` units before it
            if (n > back) wr[lane] = tr[n - 1 - back] & ((NR - 1) << 1);
        }
        // (no release fence: the trail has been READ -- LDS operations of a wave are in order -- and the index stores may still be
        // on their way; a fence would hold the slot for a memory round trip per window)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) flag_set(F_READY + 16 * s + 12, k + 3);              // the trail slot is free for the window that takes it next
    }
}

// ---- the same for plain 32- and 64-bit streams (FTL / BASE) -----------------------------------------------------------
// Thirty-two or sixty-four rungs would make a row 64-128 bytes and a window a few units long.  But the rungs a stream
// visits keep to a narrow band (a band's rung moves with the local range of the data): the table is built for SIXTEEN
// CONSECUTIVE RUNGS [R0, R0 + 16), in the 16-bit layout (a row of sixteen 16-bit entries per position, rungs relative to
// R0), and an entry whose unit leaves the band carries the stop bit -- the walk then gives up on the table and the
// call falls back to the one-lane parser (a stream that ranges over more than sixteen rungs: rare, and no worse off than
// before).  R0 comes from the stream's first index segment, which walk_probe_kernel parses outright (one lane; the
// stream starts at rung 0, outside any band that fits real data): it leaves the walk's entry state behind that segment.
// Code lengths: a code at rung r takes r, r + 1 or r + 2 bits by its two low bits whatever r is, so the table workgroup
// keeps the EXTRA bits of 2, 4 and 8 codes (at most 16: a byte) per rung and position and adds the multiples of r.
constexpr uint32_t WIDE_NG = 4, WIDE_NWR = 3, WIDE_NT = 8;   // loader groups of three waves, writer waves, trail slots
constexpr uint32_t WIDE_THREADS = 64 * (1 + 3 * WIDE_NG + WIDE_NWR);
// NR_: rungs in the band.  8 or 16: a row of 16-bit entries (16 or 32 bytes a position), entry = (position the unit ends at) *
// ROWB | 2 * rung | stop -- one dependent read a unit.  14: BYTE entries, a row of sixteen bytes -- what the walk costs is the
// table bytes ONE CU can stream (15.8 GB/s measured, whatever the slab size or the number of loader waves: a CU keeps about
// 128 cache lines in flight), so half the bytes is half the time: bytes 0 .. 13 hold, per rung of the band, the EXTRA bits of
// the sixteen codes that start behind the switch (0 .. 32; rung 0: 1 or 17, the flag and the raw bits), bytes 14, 15 the
// switch (bits 0-3 its length, 4-9 the rung step, 10 the signal).  Two dependent reads a unit: the switch, then the extras
// of the rung it leads to; unit length = switch + 16 * rung + extras.
template <uint32_t UB, uint32_t NR_> struct chainW {
    static constexpr bool BYTE = NR_ == 14;
    static constexpr uint32_t NRUNG = 1u << UB, NR = NR_, ROWB = BYTE ? 16 : 2 * NR_, MAXC = NRUNG + 1, MAXU = UB + 2 + 16 * MAXC;
    // a window of the walk: as many positions as the 16-bit entries can address ((CW + MAXU) * ROWB < 65536), a multiple of 96 (the
    // loaders' 192 sixteen-byte pieces a turn) and of TCW, the positions ONE table workgroup tabulates (its LDS holds 32 bytes a position)
    static constexpr uint32_t CW = ROWB == 16 ? 2880 : (UB == 5 ? 1440 : 960), WIN_BYTES = CW * ROWB, WIN_U4 = WIN_BYTES / 16, TCW = 480;
    static constexpr uint32_t NP = (TCW + UB + 2 + 15 * MAXC + 2 + 31) & ~31u;     // positions a table workgroup looks at
    // One lane walks; what it waits for must never be one memory round trip per window.  A window's load takes about three
    // times as long as its walk: four loader groups (three waves each) keep four windows in flight for the two slots.  A
    // writer wave ends its turn waiting for its index stores (the compiler drains the store counter before the next spin
    // loop): about as long again -- so three writers take the windows in turn, and the trails wait for them in a ring of
    // eight slots, each with the walk's state at the window's start (units done, every band's rung).
    static constexpr uint32_t NG = WIDE_NG, NWR = WIDE_NWR, NT = WIDE_NT;
    static constexpr uint32_t TR_BYTES = ((CW / 2 + 8) * 2 + 15) & ~15u;
    static constexpr uint32_t TR0 = 2 * WIN_BYTES, TM0 = TR0 + NT * TR_BYTES /* [NT] x 32 bytes: units done (u64), rungs (u64), units, first position */,
                              RS0 = TM0 + NT * 32, META = RS0 + 64, LDS_BYTES = META + 128;
    static constexpr uint32_t F_READY = META /* [2][4] */, F_TRAILED = META + 32 /* [NT] */, F_TFREE = META + 64 /* [NT] */, F_WALKED = META + 96, F_STOP = META + 100;
    static_assert(((CW + MAXU) * ROWB < 65536 || (BYTE && CW + MAXU < 4096)) && WIN_U4 % 192 == 0 && CW % TCW == 0 && (UB == 5 || UB == 6) && (NR_ == 8 || NR_ == 14 || NR_ == 16), "window layout of the wide types");
};

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

template <uint32_t UB, uint32_t NRB>
__global__ void __launch_bounds__(256) walk_tableW_kernel(const DecArgs a0, uint4 *tab, uint64_t slab0, uint32_t nwin, uint64_t tab_pitch, const WalkState16 *states) {
    typedef chainW<UB, NRB> W;
    constexpr uint32_t NP = W::NP, CW = W::CW, TCW = W::TCW, NR = W::NR, MAXC = W::MAXC, ROWB = W::ROWB, NRUNG = W::NRUNG;
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    const uint64_t p0 = slab0 + (uint64_t)blockIdx.x * TCW;                // blockIdx.x: a piece of TCW positions; CW / TCW pieces a window
    const uint32_t ow = (uint32_t)(((uint64_t)blockIdx.x * TCW) % CW);     // the piece's place in its window: entries count positions from the window's start
    if (p0 >= a.in_bits + 2 * CW || states[blockIdx.y].bad) return;        // (uniform) far beyond the stream, or no walk will come
    const uint32_t R0 = states[blockIdx.y].pad;
    __shared__ uint32_t words[NP / 32 + 3];
    __shared__ uint8_t t1[NP], eA[NR][NP], eB[NR][NP];
    const uint32_t tid = threadIdx.x;
    const uint64_t q0 = a.in_bit0 + p0, w0 = q0 >> 5, endw = (a.in_bit0 + a.in_bits + 31) >> 5;
    const uint32_t sh = (uint32_t)q0 & 31;
    for (uint32_t i = tid; i < NP / 32 + 3; i += 256) words[i] = w0 + i < endw ? a.in32[w0 + i] : 0u;
    __syncthreads();
    auto bits = [&](uint32_t i) { const uint32_t b = sh + i, k = b >> 5; return __builtin_amdgcn_alignbit(words[k + 1], words[k], b & 31); };
    for (uint32_t i = tid; i < NP; i += 256) { const uint32_t x = bits(i); t1[i] = (uint8_t)((x & 1) + ((x & 3) == 3)); }   // a code's extra bits
    __syncthreads();
    uint32_t valid = NP - MAXC;
    for (uint32_t b = 0; b < NR; b++) {                                     // two codes
        const uint32_t r = R0 + b;
        if (r) for (uint32_t i = tid; i < valid; i += 256) { const uint32_t e = t1[i]; eA[b][i] = (uint8_t)(e + t1[i + r + e]); }
    }
    __syncthreads();
    uint8_t (*src)[NP] = eA, (*dst)[NP] = eB;
#pragma unroll 1
    for (uint32_t lvl = 1; lvl < 3; lvl++) {                                // four, eight codes: extras add, positions move by k * r + extras
        valid -= MAXC << lvl;
        for (uint32_t b = 0; b < NR; b++) {
            const uint32_t r = R0 + b, kr = r << lvl;
            if (r) for (uint32_t i = tid; i < valid; i += 256) { const uint32_t e = src[b][i]; dst[b][i] = (uint8_t)(e + src[b][i + kr + e]); }
        }
        __syncthreads();
        uint8_t (*t)[NP] = src; src = dst; dst = t;
    }
    // src = the extras of eight codes; sixteen = eight + eight, formed here
    uint4 *out = tab + ((uint64_t)blockIdx.y * tab_pitch + (uint64_t)blockIdx.x * (ROWB / 16 * TCW));      // (rows are consecutive: ROWB / 16 sixteen-byte pieces a position)
    for (uint32_t o = tid; o < TCW; o += 256) {
        const uint32_t x = bits(o);
        uint32_t delta = 0; bool sig = false;
        const uint32_t cs = walk_switch<UB>(x, delta, sig);                 // from rung 0: the step itself
        const uint32_t len0 = cs + (((bits(o + cs)) & 1) ? 17 : 1);         // rung 0: one flag, then 16 raw bits
        if constexpr (W::BYTE) {
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t b = 0; b < NR; b++) {                             // the rung the switch LEADS to: the extras of the sixteen codes behind it
                const uint32_t r = R0 + b;
                uint32_t ex = len0 - cs;
                if (r) { const uint32_t e8 = src[b][o + cs]; ex = e8 + src[b][o + cs + 8 * r + e8]; }
                w[b >> 2] |= ex << (8 * (b & 3));
            }
            w[3] |= (cs | (delta << 4) | ((sig ? 1u : 0u) << 10)) << 16;
            out[o] = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            uint32_t e[NR];
#pragma unroll
            for (uint32_t bin = 0; bin < NR; bin++) {
                const uint32_t r = (R0 + bin + delta) & (NRUNG - 1), rb = r - R0;
                const bool out_of_band = rb >= NR;
                const uint32_t bb = out_of_band ? 0u : rb;
                uint32_t u = len0;
                if (r && !out_of_band) { const uint32_t n8 = 8 * r + src[bb][o + cs]; u = cs + n8 + 8 * r + src[bb][o + cs + n8]; }
                e[bin] = ((ow + o + (out_of_band ? 1u : u)) * ROWB) | (bb << 1) | ((sig || out_of_band) ? 1u : 0u);
            }
            if (NR == 16) {
                out[2 * o] = make_uint4(e[0] | e[1] << 16, e[2] | e[3] << 16, e[4] | e[5] << 16, e[6] | e[7] << 16);
                out[2 * o + 1] = make_uint4(e[8 % NR] | e[9 % NR] << 16, e[10 % NR] | e[11 % NR] << 16, e[12 % NR] | e[13 % NR] << 16, e[14 % NR] | e[15 % NR] << 16);
            } else out[o] = make_uint4(e[0] | e[1] << 16, e[2] | e[3] << 16, e[4] | e[5] << 16, e[6] | e[7] << 16);
        }
    }
}

// The walk: walk_chain16_kernel's organisation (a lane chases, six waves load windows, one writes the index) with the
// wide types' window size; rungs are relative to the band's R0 on the way, absolute in the index.
template <uint32_t UB, uint32_t NRB>
__global__ void __launch_bounds__(WIDE_THREADS) walk_chainW_kernel(const DecArgs a0, const uint4 *tab, uint64_t slab0, uint32_t nwin, uint64_t tab_pitch, WalkState16 *states) {
    typedef chainW<UB, NRB> W;
    constexpr uint32_t CW = W::CW, ROWB = W::ROWB, NR = W::NR, WIN_BYTES = W::WIN_BYTES, WIN_U4 = W::WIN_U4, TR0 = W::TR0, TR_BYTES = W::TR_BYTES, TM0 = W::TM0, RS0 = W::RS0,
                       META = W::META, F_READY = W::F_READY, F_TRAILED = W::F_TRAILED, F_TFREE = W::F_TFREE, F_WALKED = W::F_WALKED, F_STOP = W::F_STOP,
                       NG = W::NG, NWR = W::NWR, NT = W::NT;
    using chain::flag_get; using chain::flag_set; using chain::SPIN_MAX;
    const DecArgs a = dec_for_tile(a0, blockIdx.x);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t B = a.g.bands, NB = a.g.seg_blocks;
    const uint64_t nunits = a.g.nblocks * B;
    WalkState16 *S = states + blockIdx.x;
    const uint64_t P0 = S->P, U_in = S->unit, R_in = S->rungs;
    const uint32_t R0 = S->pad;
    const uint64_t slab_end = slab0 + (uint64_t)nwin * CW;
    if (S->bad || P0 < slab0 || P0 >= slab_end || P0 >= a.in_bits || U_in >= nunits) return;  // (uniform) nothing of this tile in this slab
    const uint32_t k0 = (uint32_t)((P0 - slab0) / CW);                      // the window the walk starts in
    constexpr uint32_t RSH = W::BYTE ? 0 : 1;                               // rs[]: the bands' rungs (relative to R0), times two for the entry tables
    volatile uint32_t *rs = (volatile uint32_t *)(smem + RS0);
    if (tid < 32) ((uint32_t *)(smem + META))[tid] = tid == (F_STOP - META) / 4 ? 0xffffffffu : 0u;
    if (tid < 16) rs[tid] = (uint32_t)((R_in >> (4 * tid)) & 15u) << RSH;
    __syncthreads();
    const uint4 *wt = tab + (uint64_t)blockIdx.x * tab_pitch;

    if (wave == 0) {
        if (lane) return;
        uint32_t bad = 0, k = k0, o = (uint32_t)((P0 - slab0) % CW);
        uint64_t U = U_in, Pn = P0;
        uint32_t c = (uint32_t)(U % B);
        bool stuck = false;
        while (true) {
            const uint32_t s = k & 1, ts = k % NT;
            uint32_t spin = 0;
            while (true) {      // the window's three parts are in LDS, and the trail slot has been read out (it held window k - NT)
                const bool here = flag_get(F_READY + 16 * s) == k + 1 && flag_get(F_READY + 16 * s + 4) == k + 1 && flag_get(F_READY + 16 * s + 8) == k + 1;
                const bool slot = k - k0 < NT || flag_get(F_TFREE + 4 * ts) == k - NT + 1;
                if (here && slot) break;
                if (++spin >= SPIN_MAX) { stuck = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (stuck) break;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            uint64_t Rw = 0;                                                // every band's rung as the window finds it
            for (uint32_t i = 0; i < B; i++) Rw |= (uint64_t)((rs[i] >> RSH) & 15u) << (4 * i);
            volatile uint64_t *tm = (volatile uint64_t *)(smem + TM0 + 32 * ts);
            tm[0] = U; tm[1] = Rw;
            uint32_t A = o * ROWB + (W::BYTE ? 0u : rs[c]), n = 0;
            const uint64_t left64 = nunits - U;
            uint32_t left = left64 > 0xffffffffull ? 0xffffffffu : (uint32_t)left64;
            typedef const __attribute__((address_space(3))) uint16_t *LdsHalf;
            typedef __attribute__((address_space(3))) uint16_t *LdsHalfW;
            typedef __attribute__((address_space(3))) uint32_t *LdsWordW;
            const uint32_t wbase = s * WIN_BYTES;
            LdsHalfW trw = (LdsHalfW)(uintptr_t)(TR0 + ts * TR_BYTES);
            LdsWordW rsw = (LdsWordW)(uintptr_t)RS0;
            constexpr uint32_t M = 0xffffu & ~(ROWB - 1), RM = (NR - 1) << 1;
            // a unit per turn, until one starts beyond the window or an entry carries the stop bit (a unit that leaves the band of
            // rungs, or the signal code): ONE dependent LDS read a unit (byte tables: two)
            if constexpr (W::BYTE) {
                typedef const __attribute__((address_space(3))) uint8_t *LdsByte;
                uint32_t rb = rsw[c];                                       // the band's rung, relative to R0
                while (A < CW * ROWB && left && !(bad & 1u)) {              // (A: the row of the unit's first bit)
                    const uint32_t sw = *(LdsHalf)(uintptr_t)(wbase + A + 14);
                    const uint32_t rabs = (R0 + rb + ((sw >> 4) & 63u)) & (W::NRUNG - 1), rnew = rabs - R0;
                    const bool stop = ((sw >> 10) & 1u) || rnew >= NR;
                    const uint32_t ex = *(LdsByte)(uintptr_t)(wbase + A + (stop ? 0u : rnew));
                    const uint32_t oe = (A >> 4) + (sw & 15u) + 16 * rabs + ex;         // where the unit ends: the next one's first bit
                    trw[n++] = (uint16_t)((oe << 4) | (rnew & 15u)); bad |= stop ? 1u : 0u;
                    rsw[c] = rnew;
                    A = oe << 4;
                    c = c + 1 == B ? 0 : c + 1;
                    rb = B == 1 ? rnew : rsw[c];
                    left--;
                }
            } else if (B == 1) {
                while (A < CW * ROWB && left && !(bad & 1u)) {
                    const uint32_t e = *(LdsHalf)(uintptr_t)(wbase + A);
                    trw[n++] = (uint16_t)e; bad |= e;
                    A = e & (M | RM);
                    left--;
                }
                rsw[0] = A & RM;
            } else {
                uint32_t cn = c + 1 == B ? 0 : c + 1;
                uint32_t rn = rsw[cn];
                while (A < CW * ROWB && left && !(bad & 1u)) {
                    const uint32_t cn2 = cn + 1 == B ? 0 : cn + 1;
                    const uint32_t e = *(LdsHalf)(uintptr_t)(wbase + A);
                    const uint32_t r2 = B > 2 ? rsw[cn2] : 0u;
                    trw[n++] = (uint16_t)e; bad |= e;
                    rsw[c] = e & RM;
                    A = (e & M) | rn;
                    if (B == 2) rn = e & RM; else rn = r2;                  // (two bands: this band comes again after the next unit)
                    c = cn; cn = cn2; left--;
                }
            }
            ((volatile uint32_t *)tm)[4] = n; ((volatile uint32_t *)tm)[5] = o;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            flag_set(F_TRAILED + 4 * ts, k + 1);
            U += n;
            const uint32_t oe = A / ROWB;
            Pn = slab0 + (uint64_t)k * CW + oe;
            k++;
            flag_set(F_WALKED, k - k0);
            if (U >= nunits || (bad & 1u)) break;                           // the units ran out, or the table does not carry this stream
            o = oe - CW;                                                    // (the walk left the window: oe >= CW)
            if (k >= nwin || Pn >= a.in_bits) break;                        // the slab ends here, or the stream does (a damaged one)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        flag_set(F_STOP, k);
        uint64_t Rn = 0;
        for (uint32_t i = 0; i < B; i++) Rn |= (uint64_t)((rs[i] >> RSH) & 15u) << (4 * i);
        S->P = stuck ? ~0ull : Pn; S->unit = U; S->rungs = Rn; S->bad = (bad & 1u) | (stuck ? 1u : 0u);
        if ((bad & 1u) || stuck) atomicOr(a.status, 1u);
        return;
    }
    if (wave <= 3 * NG) {
        const uint32_t g = (wave - 1) / 3, part = (wave - 1) % 3;
        constexpr uint32_t NV = WIN_U4 / 192;                               // sixteen-byte pieces a lane moves per window
        for (uint32_t k = k0 + ((g + NG - k0 % NG) % NG); k < nwin; k += NG) {
            const uint4 *src = wt + (uint64_t)k * WIN_U4;
            uint4 v[NV];
#pragma unroll
            for (uint32_t i = 0; i < NV; i++) v[i] = src[lane + 64 * (part + 3 * i)];
            uint32_t spin = 0;
            bool stop = false;
            while (true) {                                                  // the slot is free when the window two back has been walked
                if (flag_get(F_STOP) != 0xffffffffu) { stop = true; break; }
                if (flag_get(F_WALKED) + 2 > k - k0) break;
                if (++spin >= SPIN_MAX) { stop = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (stop) break;
            uint4 *slot = (uint4 *)(smem + (k & 1) * WIN_BYTES);
#pragma unroll
            for (uint32_t i = 0; i < NV; i++) slot[lane + 64 * (part + 3 * i)] = v[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) flag_set(F_READY + 16 * (k & 1) + 4 * part, k + 1);
        }
        return;
    }
    // writers: entry j of a window's trail = (position the unit ENDS at | rung of its band after it): lengths by difference
    const uint32_t wtr = wave - 1 - 3 * NG;
    for (uint32_t k = k0 + wtr;; k += NWR) {
        const uint32_t ts = k % NT;
        uint32_t spin = 0;
        bool stop = false;
        while (flag_get(F_TRAILED + 4 * ts) != k + 1) {
            const uint32_t st = flag_get(F_STOP);
            if ((st != 0xffffffffu && k >= st) || ++spin >= SPIN_MAX) { stop = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (stop) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const volatile uint64_t *tm = (const volatile uint64_t *)(smem + TM0 + 32 * ts);
        const uint64_t U0 = tm[0], Rw = tm[1];
        const uint32_t n = ((const volatile uint32_t *)tm)[4], o_first = ((const volatile uint32_t *)tm)[5];
        const uint16_t *tr = (const uint16_t *)(smem + TR0 + ts * TR_BYTES);
        const uint64_t wpos = slab0 + (uint64_t)k * CW;
        uint16_t *ul = (uint16_t *)a.idx.ulen + U0;
        for (uint32_t j = lane; j < n; j += 64) {
            constexpr uint32_t PSH = W::BYTE ? 4 : 0, PDIV = W::BYTE ? 1 : ROWB, RMASK = W::BYTE ? 15u : NR - 1;      // a trail entry: position << 4 | rung (byte tables), position * ROWB | rung << 1
            const uint32_t o0 = j ? (tr[j - 1] >> PSH) / PDIV : o_first, o1 = (tr[j] >> PSH) / PDIV;
            ul[j] = (uint16_t)(o1 - o0);
            const uint64_t Uj = U0 + j;
            if (Uj % B == 0 && (Uj / B) % NB == 0) {        // a segment starts here: position, and every band's rung as the block finds it
                const uint64_t seg = Uj / B / NB;
                a.idx.bitpos[seg] = wpos + o0;
                for (uint32_t cc = 0; cc < B; cc++) {       // band cc's unit before this one: B - cc units back, or the window's entering state
                    const int32_t jj = (int32_t)j - (int32_t)(B - cc);
                    const uint32_t rb = jj >= 0 ? ((uint32_t)tr[jj] >> RSH) & RMASK : (uint32_t)(Rw >> (4 * cc)) & 15u;
                    a.idx.rung[seg * B + cc] = (uint8_t)(R0 + rb);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the trail has been READ: LDS operations of a wave are in order; the index stores may still be on their way)
        if (lane == 0) flag_set(F_TFREE + 4 * ts, k + 1);
    }
}

// ---- Single-band 32/64-bit plain streams: EXITS instead of a chain through the table.
// The chain above reads the whole table through one CU (32 bytes a stream bit at 15.8 GB/s).  With ONE band the state of the
// walk is (position, rung) and nothing else, so a function "state entering a stretch of the stream -> state leaving it" can be
// tabulated and functions of consecutive stretches composed -- the serial part then takes one step per STRETCH, not per unit.
// walk_exitW_kernel: a workgroup per super-window of K windows of W positions.  Per window it builds the table T[position][rung
// in] -> (position the unit ends at, rung behind its switch) in LDS, the way walk_tableW_kernel does (one target rung at a time:
// extras of 2, 4, 8 codes by doubling), and moves every state of X -- all (position < MAXU, rung) a walk can enter the
// super-window with -- through T until it leaves the window; after K windows X holds, per entering state, the state the walk
// leaves the super-window with and the units it took: 4 bytes x MAXU x 16 per 32768 stream bits, about a byte a bit.
// walk_exit_chain_kernel: one lane hops from super-window to super-window (one dependent load each) and notes where each is
// entered.  walk_exit_units_kernel: a lane per super-window parses its units from there: unit lengths, segment entries.
// A unit that leaves the band of rungs or carries the signal code stops the walk: status bit 0, and the caller falls back.
template <uint32_t UB> struct WalkValue { typedef typename std::conditional<UB == 3, uint8_t, typename std::conditional<UB == 4, uint16_t, typename std::conditional<UB == 5, uint32_t, uint64_t>::type>::type>::type type; };
template <uint32_t UB> struct exitW {
    static constexpr uint32_t NRUNG = 1u << UB, NR = 16, NRB = NRUNG < NR ? NRUNG : NR, MAXC = NRUNG + 1, MAXU = UB + 2 + 16 * MAXC;
    static constexpr uint32_t W = UB == 6 ? 1024 : 2048, K = 65536 / W, SW = W * K, THREADS = 1024;      // (a super-window's cost is its first window's: long ones)
    // A walk leaves a window at the first unit that starts behind it AND is entered with a rung of the band: units entered
    // out of the band (the one behind a unit whose switch jumped out: the first unit of a block row of a wide raster) are
    // walked on the spot from the code lengths, so a window can be entered up to PE bits in.  (8- and 16-bit data: the band
    // is all the rungs there are.)
    static constexpr uint32_t PE = MAXU + (UB >= 5 ? 512 : 0), NX = PE * NR;                // states a window can be entered with
    static constexpr uint32_t NPT = (W + UB + 2 + 15 * MAXC + 2 + 31) & ~31u;               // positions the table of a window looks at
    static constexpr uint32_t NPS = W + PE, NP1 = (NPS + MAXU + 2 + 31) & ~31u;             // positions with a switch entry; with a code length
    static constexpr uint32_t X_DEP = 1u << 30, X_SLOW = 1u << 31, X_CNT = 0x7fffu;          // (common-factor streams) a unit took the factor in force when the super-window was entered; a unit brought its own
    static constexpr uint32_t X_STOP = 0x7fffu;                                             // X: (position - W) * 16 + rung (15 bits: the entering state of the next window) | units << 15; stop: the low 15 bits all set
    static constexpr uint32_t BMW = (NX + 31) / 32, DCAP = UB == 6 ? NX : (NX < 4096 ? NX : 4096);   // words of the bitmap of first-window exits; distinct exits carried (64-bit data, 1024-bit windows: thousands; else a few hundred)
    static constexpr uint32_t T0 = 0, X0 = T0 + W * NR * 2, PF0 = X0 + ((BMW * 4 + 15) & ~15u), XD0 = PF0 + ((BMW * 2 + 15) & ~15u), S0 = XD0 + DCAP * 4,
                              E1 = S0 + ((NPS * 2 + 15) & ~15u), EA = E1 + NP1, EB = EA + NPT, WORDS = EB + NPT, LDS_BYTES = WORDS + (NP1 / 32 + 3) * 4;
    static_assert(W + MAXU < 4095 && PE * NR + NR <= 0x7fff && UB >= 3 && UB <= 6 && LDS_BYTES <= 160 * 1024, "entry layouts of the exit walk");
};

template <uint32_t UB, bool CF>
__global__ void __launch_bounds__(1024) walk_exitW_kernel(const DecArgs a0, uint32_t *xg, uint32_t s_begin, uint32_t s_count, const WalkState16 *states) {
    typedef exitW<UB> E;
    constexpr uint32_t W = E::W, NR = E::NR, NPT = E::NPT, NP1 = E::NP1, NPS = E::NPS, MAXC = E::MAXC, NRUNG = E::NRUNG, NX = E::NX, NT = E::THREADS;
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    const WalkState16 &S = states[blockIdx.y];
    if (S.bad) return;
    const uint32_t R0 = S.pad;
    const uint64_t base = S.P + (uint64_t)(s_begin + blockIdx.x) * E::SW;                   // the super-window's first bit (the walk enters the first one at its bit 0)
    if (base >= a.in_bits) return;                                                          // (uniform) no walk comes here
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *T = (uint16_t *)(smem + E::T0), *sw = (uint16_t *)(smem + E::S0);
    uint32_t *bm = (uint32_t *)(smem + E::X0), *Xd = (uint32_t *)(smem + E::XD0), *words = (uint32_t *)(smem + E::WORDS);
    uint16_t *pf = (uint16_t *)(smem + E::PF0);
    uint32_t *out = xg + ((uint64_t)blockIdx.y * s_count + blockIdx.x) * NX;          // X: the super-window's exits, entering state by entering state
    __shared__ uint32_t s_D;
    uint32_t D = 0;
    uint8_t *t1 = smem + E::E1, *eA = smem + E::EA, *eB = smem + E::EB;
    const uint32_t tid = threadIdx.x;
    const uint64_t endw = (a.in_bit0 + a.in_bits + 31) >> 5;
    // a first-window exit (or a list entry) a, then b: flags add up -- a factor brought anywhere, a factor taken before one was
    // brought, or after: X_SLOW then
    auto compose = [](uint32_t a, uint32_t b) -> uint32_t {
        const uint32_t n = ((a >> 15) & E::X_CNT) + ((b >> 15) & E::X_CNT);               // (more units than the field holds -- two bits a unit: flat data -- stop the walk)
        return (n > E::X_CNT ? E::X_STOP : (b & 0x7fffu)) | ((n & E::X_CNT) << 15) | ((a | b) & (E::X_DEP | E::X_SLOW));
    };
#pragma unroll 1
    for (uint32_t k = 0; k < E::K; k++) {
        const uint64_t q0 = a.in_bit0 + base + (uint64_t)k * W, w0 = q0 >> 5;
        const uint32_t sh = (uint32_t)q0 & 31;
        for (uint32_t i = tid; i < NP1 / 32 + 3; i += NT) words[i] = w0 + i < endw ? a.in32[w0 + i] : 0u;
        __syncthreads();
        auto bits = [&](uint32_t i) { const uint32_t b = sh + i, j = b >> 5; return __builtin_amdgcn_alignbit(words[j + 1], words[j], b & 31); };
        for (uint32_t i = tid; i < NP1; i += NT) { const uint32_t x = bits(i); t1[i] = (uint8_t)((x & 1) + ((x & 3) == 3)); }   // a code's extra bits
        for (uint32_t o = tid; o < NPS; o += NT) {                                          // the switch in front of a unit that starts at o
            uint32_t delta = 0; bool sig = false;
            const uint32_t cs = walk_switch<UB>(bits(o), delta, sig);
            sw[o] = (uint16_t)(cs | (delta << 4) | ((sig ? 1u : 0u) << 10) | ((bits(o + cs) & 1u) << 11));
        }
        // The table is made for the super-window's FIRST window, where thousands of states walk; behind it a few hundred
        // distinct states are usually left, and walking those from the code lengths (sixteen dependent byte reads a unit)
        // costs a quarter of what tabulating sixteen rungs of the window does.
        const bool tabled = k == 0 || D > 1024;                                             // (uniform.  Many distinct states: the table pays in every window)
        if (tabled) for (uint32_t i = tid; i < W * NR / 2; i += NT) ((uint32_t *)T)[i] = 0xffffffffu;  // (an entry no target rung fills: the unit leaves the band, or is the signal)
        __syncthreads();
#pragma unroll 1
        for (uint32_t rb = 0; tabled && rb < E::NRB; rb++) {                                // the rung the switch leads to
            const uint32_t r = R0 + rb;
            if (r) {                                                                        // extras of two, four, eight codes at rung r
                for (uint32_t i = tid; i < NPT - MAXC; i += NT) { const uint32_t e = t1[i]; eA[i] = (uint8_t)(e + t1[i + r + e]); }
                __syncthreads();
                for (uint32_t i = tid; i < NPT - 3 * MAXC; i += NT) { const uint32_t e = eA[i]; eB[i] = (uint8_t)(e + eA[i + 2 * r + e]); }
                __syncthreads();
                for (uint32_t i = tid; i < NPT - 7 * MAXC; i += NT) { const uint32_t e = eB[i]; eA[i] = (uint8_t)(e + eB[i + 4 * r + e]); }
                __syncthreads();
            }
            for (uint32_t o = tid; o < W; o += NT) {
                const uint32_t s = sw[o], cs = s & 15u, delta = (s >> 4) & 63u;
                const uint32_t bin = ((r - delta) & (NRUNG - 1)) - R0;                      // the rung the unit is entered with, in the band
                if (bin >= NR || ((s >> 10) & 1u)) continue;
                uint32_t u = cs + (((s >> 11) & 1u) ? 17u : 1u);                            // rung 0: one flag, then 16 raw bits
                if (r) { const uint32_t n8 = 8 * r + eA[o + cs]; u = cs + n8 + 8 * r + eA[o + cs + n8]; }
                T[o * NR + bin] = (uint16_t)((o + u) | (rb << 12));
            }
            __syncthreads();
        }
        // Every entering state of the super-window through its first window (exit and unit count to global memory, X); walks merge
        // -- behind the first window the thousands of states stand at a few hundred distinct (position, rung) -- so the distinct
        // exits are ranked through a bitmap and only those (Xd, LDS) are carried through the other windows; at the end every
        // state composes its first-window exit with what became of it.
        auto walk = [&](uint32_t key, auto with_table) -> uint32_t {                        // (with_table: a compile-time flag -- the loop without the table look-up is the tighter one)
            constexpr bool TB = decltype(with_table)::value;
            uint32_t pos = key / NR, r = key % NR, cnt = 0;                                 // (a state's low 15 bits: position * 16 + rung: the key itself)
            bool stop = false;
            typedef typename WalkValue<UB>::type TT;
            TT cfv = (TT)S.cf; uint32_t xfl = 0;                                            // (common-factor streams) the factor in force: the one behind the first segment until a unit brings its own
            while (true) {
                if (r < NR) {
                    if (pos >= W) break;                                                    // behind the window, in the band: the next window's
                    if (TB) {
                        const uint32_t e = T[pos * NR + r];
                        if (e != 0xffffu) { pos = e & 0xfffu; r = e >> 12; cnt++; continue; }
                    }
                }
                // a unit the table does not hold (it leaves the band, is entered from outside it, or there is no table): by the code lengths
                if (pos >= NPS) { stop = true; break; }
                const uint32_t s = sw[pos], cs = s & 15u;
                if ((s >> 10) & 1u) {                                                       // the signal code
                    if (!CF) { stop = true; break; }                                        // ... in a stream that should have none
                    // a common-factor or index unit: parsed outright (its values decide the rung it leaves).  The factor in force
                    // is not part of the state: a unit that takes it is walked with the factor the stream had behind its first
                    // segment and says so (X_DEP: right as long as no unit in between brought another, which the hop checks); a
                    // unit that brings its own marks the walk X_SLOW: the hop parses that super-window outright.
                    ReaderT<LdsWords> rd;
                    rd.init((LdsWords)words, sh + pos, 32ull * (NP1 / 32 + 3));
                    uint32_t rg = (R0 + r) & (NRUNG - 1), fl = 0;
                    TT pc = cfv, g[16];
                    const bool ok = parse_unit<TT, CM_BEST>(rd, rg, pc, g, &fl);
                    if (!ok) { stop = true; break; }
                    if ((fl & 1u) && !(xfl & E::X_SLOW)) xfl |= E::X_DEP;
                    if (fl & 2u) { xfl |= E::X_SLOW; cfv = pc; }
                    pos = (uint32_t)rd.position() - sh; r = (rg - R0) & (NRUNG - 1); cnt++;
                    continue;
                }
                const uint32_t rabs = (R0 + r + ((s >> 4) & 63u)) & (NRUNG - 1);
                uint32_t q = pos + cs;
                if (rabs) { for (uint32_t i = 0; i < 16; i++) q += rabs + t1[q]; }
                else q += ((s >> 11) & 1u) ? 17u : 1u;
                pos = q; r = (rabs - R0) & (NRUNG - 1); cnt++;
            }
            if (pos - W >= E::PE) stop = true;                                              // (only behind a unit entered out of the band)
            return (stop ? (E::X_STOP | (cnt << 15)) : ((pos - W) * NR + r) | (cnt << 15)) | xfl;
        };
        if (k == 0) {
            for (uint32_t i = tid; i < E::BMW; i += NT) bm[i] = 0;
            __syncthreads();
            for (uint32_t key = tid; key < NX; key += NT) {
                const uint32_t x = walk(key, std::true_type());
                out[key] = x;
                if ((x & E::X_STOP) != E::X_STOP) atomicOr(&bm[(x & 0x7fffu) >> 5], 1u << (x & 31u));
            }
            __syncthreads();
            if (tid < 64) {                                                                 // rank of every distinct exit: one wave scans the bitmap words' bit counts
                uint32_t run = 0;
                for (uint32_t w0 = 0; w0 < E::BMW; w0 += 64) {
                    const uint32_t w = w0 + tid, c = w < E::BMW ? __popc(bm[w]) : 0u;
                    uint32_t x = c;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if ((int)tid >= d) x += y; }
                    if (w < E::BMW) pf[w] = (uint16_t)(run + x - c);
                    run += __shfl(x, 63, 64);
                }
                if (tid == 0) s_D = run;
            }
            __syncthreads();
            D = s_D;
            if (D > E::DCAP) {                                                              // (uniform) more distinct exits than are carried: the hop parses this super-window outright
                for (uint32_t key = tid; key < NX; key += NT) out[key] = E::X_STOP;
                return;
            }
            for (uint32_t w = tid; w < E::BMW; w += NT) {                                   // the distinct exits, in rank order
                uint32_t m = bm[w], j = pf[w];
                while (m) { const uint32_t b = __ffs(m) - 1; Xd[j++] = w * 32 + b; m &= m - 1; }
            }
            __syncthreads();
        } else {
            for (uint32_t j = tid; j < D; j += NT) {                                        // the distinct walks through this window
                const uint32_t x = Xd[j];
                if ((x & E::X_STOP) == E::X_STOP) continue;
                Xd[j] = compose(x, tabled ? walk(x & 0x7fffu, std::true_type()) : walk(x & 0x7fffu, std::false_type()));
            }
            __syncthreads();
        }
    }
    for (uint32_t key = tid; key < NX; key += NT) {                                         // every state: its first-window exit, then what became of that
        const uint32_t e = out[key];
        if ((e & E::X_STOP) == E::X_STOP) continue;
        const uint32_t k1 = e & 0x7fffu, w = k1 >> 5;
        out[key] = compose(e, Xd[pf[w] + __popc(bm[w] & ((1u << (k1 & 31u)) - 1u))]);
    }
}

// entries: per tile nsuper + 2 pairs of {position lo, hi, unit, rung in the band} {factor in force lo, hi}: where and how the walk
// enters super-window s; the last pair is {the super-window the walk stands in front of, 1 when every unit has been found}.  The
// stream is taken s_count super-windows at a time (the memory for their exits is reused): a call takes up where the one before
// stopped.  A super-window whose exit cannot be taken from the table -- a unit in it brought a common factor of its own, or took
// the one in force when that is no longer the one the table was made with, or the walk stopped -- is parsed outright by this
// lane (about 260 units): the stream still decodes, at the one-lane parser's pace for that stretch.
template <uint32_t UB, int MODE>
__global__ void __launch_bounds__(64) walk_exit_chain_kernel(const DecArgs a0, const uint32_t *xg, uint32_t nsuper, uint32_t s_begin, uint32_t s_count, WalkState16 *states, uint4 *entries) {
    typedef exitW<UB> E;
    typedef typename WalkValue<UB>::type T;
    const DecArgs a = dec_for_tile(a0, blockIdx.x);
    if (threadIdx.x) return;
    WalkState16 *S = states + blockIdx.x;
    if (S->bad) return;
    const uint64_t nunits = a.g.nblocks, P0 = S->P, spec = S->cf;
    const uint32_t R0 = S->pad;
    uint4 *en = entries + (uint64_t)blockIdx.x * 2 * (nsuper + 2), *hd = en + 2 * (nsuper + 1);
    uint64_t P = P0, U = S->unit, cf = spec;
    uint32_t r = (uint32_t)S->rungs & 15u, s = 0;
    bool bad = false, done = false;
    if (s_begin) {
        const uint4 h = *hd;
        if (h.y) return;                                                                    // all units found in an earlier call
        const uint4 e = en[2 * s_begin], f = en[2 * s_begin + 1];
        P = (uint64_t)e.x | (uint64_t)e.y << 32; U = e.z; r = e.w; s = s_begin; cf = (uint64_t)f.x | (uint64_t)f.y << 32;
        bad = h.x != s_begin;
    }
    const uint32_t s_end = s_begin + s_count < nsuper ? s_begin + s_count : nsuper;
    const uint32_t *x0 = xg + (uint64_t)blockIdx.x * s_count * E::NX;
    while (!bad) {
        en[2 * s] = make_uint4((uint32_t)P, (uint32_t)(P >> 32), (uint32_t)U, r);
        en[2 * s + 1] = make_uint4((uint32_t)cf, (uint32_t)(cf >> 32), 0u, 0u);
        if (U >= nunits) { done = true; break; }
        if (s >= s_end) { bad = s >= nsuper; break; }                                       // the next call's; or units left and no stream (a damaged one)
        if (P >= a.in_bits) { bad = true; break; }
        const uint64_t base = P0 + (uint64_t)s * E::SW;
        const uint32_t x = x0[(uint64_t)(s - s_begin) * E::NX + (uint32_t)(P - base) * E::NR + r];
        const bool stopped = (x & E::X_STOP) == E::X_STOP;
        const bool slow = stopped || (MODE == CM_BEST && ((x & E::X_SLOW) || ((x & E::X_DEP) && cf != spec)));
        s++;
        if (!slow) {
            U += (x >> 15) & E::X_CNT;
            r = x & 15u;
            P = base + E::SW + ((x & 0x7fffu) >> 4);
            continue;
        }
        if (stopped && U + ((x >> 15) & E::X_CNT) >= nunits) {                              // the stream's units end before the stop
            done = true;
            en[2 * s] = make_uint4(0u, 0u, (uint32_t)nunits, 0u); en[2 * s + 1] = make_uint4(0u, 0u, 0u, 0u);
            break;
        }
        // this super-window by the units themselves: up to the first unit that starts behind it and is entered with a rung of the band
        atomicOr(a.status, 64u);                                                            // (not an error: says that the walk was handed to this lane)
        Reader rd;
        rd.init(a.in32, a.in_bit0 + P, a.in_bit0 + a.in_bits);
        uint32_t rung = R0 + r;
        T pc = (T)cf, g[16];
        bool ok = true;
        const uint64_t end = base + E::SW;
        while (ok && U < nunits) {
            const uint64_t pos = rd.position() - a.in_bit0;
            if (pos >= a.in_bits || (pos >= end && ((rung - R0) & (E::NRUNG - 1)) < E::NR)) break;
            ok = parse_unit<T, MODE>(rd, rung, pc, g);
            U++;
        }
        P = rd.position() - a.in_bit0; r = (rung - R0) & (E::NRUNG - 1); cf = (uint64_t)pc;
        if (!ok || (U < nunits && (P < end || P - end >= E::PE || r >= E::NR))) { bad = true; break; }
    }
    *hd = make_uint4(s, done ? 1u : 0u, 0u, 0u);                                            // done: super-windows 0 .. s - 1 have units to parse, entries 0 .. s stand
    if (bad) { S->bad = 1u; atomicOr(a.status, 1u); }
}

// idx.prev = 0 behind the first segment: the sums of the segments' values are added up there by the lanes that parse them
template <typename T>
__global__ void __launch_bounds__(256) walk_exit_zero_kernel(const DecArgs a0) {
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    const uint64_t seg = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (seg && seg < a.g.nseg) for (uint32_t c = 0; c < a.g.bands; c++) ((T *)a.idx.prev)[seg * a.g.bands + c] = 0;
}
// A WAVE per super-window: its stretch of the stream staged in LDS by all lanes (a lane parsing straight from global memory waits
// a round trip per word: 2.5 ms for 4096^2 int32 against 0.3 staged), then lane 0 parses the units: unit lengths, segment entries,
// for common-factor streams the segments' sums.  sw_bits / pe_bits: the super-window's size and how far in it can be entered.
template <typename T, int MODE>
__global__ void __launch_bounds__(64) walk_exit_units_kernel(const DecArgs a0, const WalkState16 *states, const uint4 *entries, uint32_t nsuper, uint32_t sw_bits, uint32_t pe_bits) {
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    const WalkState16 &S = states[blockIdx.y];
    const uint32_t s = blockIdx.x, lane = threadIdx.x;
    const uint4 *en = entries + (uint64_t)blockIdx.y * 2 * (nsuper + 2);
    const uint4 hd = en[2 * (nsuper + 1)];
    if (S.bad || !hd.y || s >= hd.x) return;                                                // (uniform)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *stage = (uint32_t *)smem;
    const uint4 e = en[2 * s], f = en[2 * s + 1];
    const uint32_t B = a.g.bands;
    const uint64_t nblocks = a.g.nblocks, NB = a.g.seg_blocks;
    uint64_t U = e.z, Uend = en[2 * s + 2].z;                                               // (in blocks)
    if (Uend > nblocks) Uend = nblocks;
    const uint64_t P = (uint64_t)e.x | (uint64_t)e.y << 32;
    const uint64_t q0 = a.in_bit0 + P, w0 = q0 >> 5, endw = (a.in_bit0 + a.in_bits + 31) >> 5;
    const uint32_t nw = (sw_bits + pe_bits) / 32 + 4;                                       // (the units of this super-window end where the next is entered)
    for (uint32_t i = lane; i < nw; i += 64) stage[i] = w0 + i < endw ? a.in32[w0 + i] : 0u;
    __syncthreads();
    if (lane) return;
    uint32_t rung[4];
    for (uint32_t c = 0; c < 4; c++) rung[c] = (B == 1 ? S.pad : 0u) + ((e.w >> (4 * c)) & 15u);
    ReaderT<LdsWords> rd;
    rd.init((LdsWords)stage, (uint32_t)q0 & 31, 32ull * nw);
    const uint64_t rel = w0 * 32 - a.in_bit0;                                               // stream position of the stage's first bit
    uint32_t lpos = (uint32_t)q0 & 31;                                                      // (FTL / BASE) bit position in the stage
    const uint32_t lds0 = 8u * (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)stage;    // ... whose first bit is LDS bit lds0
    // (the factors in force where the super-window is entered: one band -- the whole value; several -- a byte a band, 8-bit data)
    T g[16], pcf[4], tot[4];
    for (uint32_t c = 0; c < 4; c++) { tot[c] = 0; pcf[c] = B == 1 ? (T)((uint64_t)f.x | (uint64_t)f.y << 32) : (T)(f.x >> (8 * c)); }
    bool ok = true;
    typedef typename std::conditional<sizeof(T) == 8, unsigned long long, unsigned int>::type AT;
    for (; U < Uend; U++) {
        if (U % NB == 0) {
            const uint64_t seg = U / NB;
            a.idx.bitpos[seg] = rel + (MODE != CM_BEST ? (uint64_t)lpos : rd.position());
            for (uint32_t c = 0; c < B; c++) {
                a.idx.rung[seg * B + c] = (uint8_t)rung[c];
                if (MODE == CM_BEST) ((T *)a.idx.cf)[seg * B + c] = pcf[c];
            }
        }
        if (MODE != CM_BEST) {      // FTL / BASE: lengths only, by position in the staged words (walk_unit: three dependent reads an 8-bit unit; a full parse costs ten times that)
            for (uint32_t c = 0; c < B; c++) {
                bool bad = false;
                const uint32_t len = walk_unit<UBits<T>::v>(lds0 + lpos, rung[c], bad);
                ok = ok && !bad;
                if (sizeof(T) == 1) ((uint8_t *)a.idx.ulen)[U * B + c] = (uint8_t)len; else ((uint16_t *)a.idx.ulen)[U * B + c] = (uint16_t)len;
                lpos += len;
            }
            continue;
        }
        const uint64_t b0 = rd.position();
        uint32_t bt = 0;                                                                    // (8-bit common-factor streams: the block's entry of the lane-per-block decoder's table)
        for (uint32_t c = 0; c < B; c++) {
            const uint64_t u0 = rd.position();
            if (c < 4) bt |= (rung[c] & (sizeof(T) >= 4 ? 63u : 15u)) << (16 + 4 * c);
            ok = parse_unit<T, MODE>(rd, rung[c], pcf[c], g) && ok;
            if (MODE != CM_BEST) {
                if (sizeof(T) == 1) ((uint8_t *)a.idx.ulen)[U * B + c] = (uint8_t)(rd.position() - u0); else ((uint16_t *)a.idx.ulen)[U * B + c] = (uint16_t)(rd.position() - u0);
            } else {                                                                        // the segment's sum of values: the scan makes entering values of them
#pragma unroll
                for (uint32_t i = 0; i < 16; i++) tot[c] = (T)(tot[c] + smag_t<T>(g[i]));
                if ((U + 1) % NB == 0 || U + 1 == Uend) {
                    const uint64_t slot = (U / NB) * B + c;
                    if (sizeof(T) >= 4) atomicAdd((AT *)a.idx.prev + slot, (AT)tot[c]);
                    else {                                                                  // narrow values: the slot's lane of its dword, by compare and swap (the neighbours may be added to meanwhile)
                        constexpr uint32_t BITS = sizeof(T) < 4 ? 8 * sizeof(T) : 16, PER = sizeof(T) < 4 ? 4 / sizeof(T) : 1, MASK = (1u << BITS) - 1;       // (instantiated, not run, for wide values)
                        uint32_t *wp = (uint32_t *)a.idx.prev + slot / PER;
                        const uint32_t sh = (uint32_t)(slot % PER) * BITS;
                        uint32_t old = *(volatile uint32_t *)wp, assumed;
                        do {
                            assumed = old;
                            const uint32_t nv = (assumed & ~(MASK << sh)) | ((((assumed >> sh) + (uint32_t)tot[c]) & MASK) << sh);
                            old = atomicCAS(wp, assumed, nv);
                        } while (old != assumed);
                    }
                    tot[c] = 0;
                }
            }
        }
        if (MODE == CM_BEST && a.g.ulen_sz == 4) ((uint32_t *)a.idx.ulen)[U] = bt | (uint32_t)((rd.position() - b0) & 0xffffu);
    }
    if (!ok) atomicOr(a.status, 1u);
}
// ---- The same for 8-bit rasters of THREE bands (RGB).  The walk's state at a block boundary is (position, a rung per band):
// 447 positions x 512 rung combinations = 228 864 states a super-window can be entered with -- too many to carry through every
// window, but rungs aside the walks merge within one window (positions do, rung offsets never: about ten positions survive per
// combination).  So the first window is walked by every state (its exit and block count go to global memory, G), the distinct
// exits are ranked through a bitmap (D, a few thousand), only those are carried through the other windows of the super-window
// (Xd, LDS), and at the end every state composes its first-window exit with what became of it.  One hop per super-window of
// 65 536 bits as before; 915 KB of exits per super-window, so the stream is taken in rounds of what the table memory holds.
template <uint32_t B, bool CF = false> struct exitB {
    static constexpr uint32_t UB = 3, NRUNG = 8, NR = 8, MAXC = NRUNG + 1, MAXU = UB + 2 + 16 * MAXC;        // 149
    static constexpr uint32_t W = 2048, K = 64, SW = W * K, THREADS = 1024;             // (a super-window's cost is its first window's, where every state walks: long ones -- twice this: 4 % more, and the lanes that parse the units become the long pole)
    static constexpr uint32_t PE = B * MAXU, NC = 1u << (3 * B), NKEY = PE * NC;                             // entering positions, rung combinations, states
    static constexpr uint32_t TP = W + (B - 1) * MAXU;                                                        // positions with a table row: the later units of a block that starts in the window
    static constexpr uint32_t NPT = (TP + UB + 2 + 15 * MAXC + 2 + 31) & ~31u, NP1 = (TP + MAXU + 2 + 63) & ~31u;
    static constexpr uint32_t KEYB = 18, KEYM = (1u << KEYB) - 1, X_STOP = KEYM, DCAP = 8192;                 // X: state | blocks << 18; stop: the state field all set
    // blocks of a super-window: 14 bits, or 13 beside the bit that says (common-factor streams) "a unit took the factor in force when
    // the super-window was entered": more blocks than that -- eight or sixteen bits a block: flat data -- stop the walk, the hop parses it
    static constexpr uint32_t X_DEP = CF ? 1u << 31 : 0u, CNTM = CF ? 0x1fffu : 0x3fffu;
    static constexpr uint32_t BMW = (NKEY + 31) / 32;                                                         // words of the bitmap of first-window exits
    static constexpr uint32_t NSIG = 128;                                                                     // (common-factor streams) positions of a window whose unit carries the signal code, at most
    static constexpr uint32_t T0 = 0, BM0 = T0 + ((TP * NR * 2 + 15) & ~15u), PF0 = BM0 + BMW * 4, XD0 = PF0 + ((BMW * 2 + 15) & ~15u), S0 = XD0 + DCAP * 4,
                              E1 = S0 + ((TP * 2 + 15) & ~15u), EA = E1 + NP1, EB = EA + NPT, WORDS = EB + NPT, SG0 = (WORDS + (NP1 / 32 + 3) * 4 + 15) & ~15u,
                              SL0 = SG0 + (CF ? NSIG * B * NR * 4 : 0), SP0 = SL0 + (CF ? (TP + 15) & ~15u : 0), LDS_BYTES = SP0 + (CF ? NSIG * 2 + 16 : 0);
    static_assert(B == 3 && NKEY <= KEYM && TP + MAXU < 4095 && LDS_BYTES <= 160 * 1024, "entry layouts of the exit walk of RGB rasters");
};

template <uint32_t B, bool CF>
__global__ void __launch_bounds__(1024) walk_exitB_kernel(const DecArgs a0, uint32_t *xg, uint32_t s_begin, uint32_t s_count, const WalkState16 *states, uint32_t dcap) {
    typedef exitB<B, CF> E;
    constexpr uint32_t W = E::W, NR = E::NR, NPT = E::NPT, NP1 = E::NP1, TP = E::TP, MAXC = E::MAXC, NRUNG = E::NRUNG, NKEY = E::NKEY, NT = E::THREADS, UB = E::UB;
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    const WalkState16 &S = states[blockIdx.y];
    if (S.bad) return;
    const uint64_t base = S.P + (uint64_t)(s_begin + blockIdx.x) * E::SW;
    if (base >= a.in_bits) return;                                                          // (uniform) no walk comes here
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *T = (uint16_t *)(smem + E::T0), *sw = (uint16_t *)(smem + E::S0), *pf = (uint16_t *)(smem + E::PF0);
    uint32_t *bm = (uint32_t *)(smem + E::BM0), *Xd = (uint32_t *)(smem + E::XD0), *words = (uint32_t *)(smem + E::WORDS);
    uint8_t *t1 = smem + E::E1, *eA = smem + E::EA, *eB = smem + E::EB;
    uint32_t *G = xg + ((uint64_t)blockIdx.y * s_count + blockIdx.x) * NKEY;               // the super-window's exits, entering state by entering state
    __shared__ uint32_t s_D;
    const uint32_t tid = threadIdx.x;
    const uint64_t endw = (a.in_bit0 + a.in_bits + 31) >> 5;
    uint32_t *side = (uint32_t *)(smem + E::SG0);                                           // (common-factor streams: see below)
    uint8_t *sig_slot = smem + E::SL0;
    uint16_t *sigpos = (uint16_t *)(smem + E::SP0);
    __shared__ uint32_t s_nsig;
    // a state through the window: whole blocks until one starts behind it.  Returns the state behind | blocks << 18 (| X_DEP), or the
    // stop.  Common-factor streams: the units with the signal code are tabulated apart, per window, by band and entering rung (a
    // dense pass: every lane parses one -- parsed inside the walks, one lane of a wave at a time, they made the kernel 25 times
    // slower); one that takes its band's factor in force is parsed with the factor the stream had behind its first segment and
    // the exit says so (X_DEP); one that brings its own ends the walk -- the hop parses such a super-window itself.
    auto walk = [&](uint32_t key) -> uint32_t {
        uint32_t pos = key >> (3 * B), r[B], cnt = 0, dep = 0;
#pragma unroll
        for (uint32_t c = 0; c < B; c++) r[c] = (key >> (3 * c)) & 7u;
        while (pos < W) {
#pragma unroll
            for (uint32_t c = 0; c < B; c++) {
                const uint32_t e = T[pos * NR + r[c]];
                if (e != 0xffffu) { pos = e & 0xfffu; r[c] = e >> 12; continue; }
                if (!CF) return E::X_STOP;                                                  // the signal code: not a stream for this walk
                const uint32_t j = sig_slot[pos];                                           // ... tabulated apart: by band and entering rung
                if (j == 0xffu) return E::X_STOP;
                const uint32_t v = side[(j * B + c) * NR + r[c]];
                if (v & 0x10000u) return E::X_STOP;
                if (v & 0x8000u) dep = E::X_DEP;
                pos = v & 0xfffu; r[c] = (v >> 12) & 7u;
                if (c + 1 < B && pos >= TP) return E::X_STOP;                               // (the block's next unit would start behind the table)
            }
            cnt++;
        }
        if (pos - W >= E::PE) return E::X_STOP;
        uint32_t k2 = (pos - W) << (3 * B);
#pragma unroll
        for (uint32_t c = 0; c < B; c++) k2 |= r[c] << (3 * c);
        return k2 | (cnt << E::KEYB) | dep;
    };
    // an exit a, then b
    auto compose = [](uint32_t a, uint32_t b) -> uint32_t {
        if ((b & E::KEYM) == E::X_STOP) return E::X_STOP;
        const uint32_t n = ((a >> E::KEYB) & E::CNTM) + ((b >> E::KEYB) & E::CNTM);
        if (n > E::CNTM) return E::X_STOP;
        return (b & E::KEYM) | (n << E::KEYB) | ((a | b) & E::X_DEP);
    };
    uint32_t D = 0;
#pragma unroll 1
    for (uint32_t k = 0; k < E::K; k++) {
        const uint64_t q0 = a.in_bit0 + base + (uint64_t)k * W, w0 = q0 >> 5;
        const uint32_t sh = (uint32_t)q0 & 31;
        for (uint32_t i = tid; i < NP1 / 32 + 3; i += NT) words[i] = w0 + i < endw ? a.in32[w0 + i] : 0u;
        __syncthreads();
        auto bits = [&](uint32_t i) { const uint32_t b = sh + i, j = b >> 5; return __builtin_amdgcn_alignbit(words[j + 1], words[j], b & 31); };
        for (uint32_t i = tid; i < NP1; i += NT) { const uint32_t x = bits(i); t1[i] = (uint8_t)((x & 1) + ((x & 3) == 3)); }
        for (uint32_t o = tid; o < TP; o += NT) {
            uint32_t delta = 0; bool sig = false;
            const uint32_t cs = walk_switch<UB>(bits(o), delta, sig);
            sw[o] = (uint16_t)(cs | (delta << 4) | ((sig ? 1u : 0u) << 10) | ((bits(o + cs) & 1u) << 11));
        }
        for (uint32_t i = tid; i < TP * NR / 2; i += NT) ((uint32_t *)T)[i] = 0xffffffffu;
        __syncthreads();
#pragma unroll 1
        for (uint32_t r = 0; r < NRUNG; r++) {                                              // the rung the switch leads to
            if (r) {
                for (uint32_t i = tid; i < NPT - MAXC; i += NT) { const uint32_t e = t1[i]; eA[i] = (uint8_t)(e + t1[i + r + e]); }
                __syncthreads();
                for (uint32_t i = tid; i < NPT - 3 * MAXC; i += NT) { const uint32_t e = eA[i]; eB[i] = (uint8_t)(e + eA[i + 2 * r + e]); }
                __syncthreads();
                for (uint32_t i = tid; i < NPT - 7 * MAXC; i += NT) { const uint32_t e = eB[i]; eA[i] = (uint8_t)(e + eB[i + 4 * r + e]); }
                __syncthreads();
            }
            for (uint32_t o = tid; o < TP; o += NT) {
                const uint32_t s = sw[o], cs = s & 15u, delta = (s >> 4) & 63u;
                if ((s >> 10) & 1u) continue;
                const uint32_t bin = (r - delta) & (NRUNG - 1);
                uint32_t u = cs + (((s >> 11) & 1u) ? 17u : 1u);
                if (r) { const uint32_t n8 = 8 * r + eA[o + cs]; u = cs + n8 + 8 * r + eA[o + cs + n8]; }
                T[o * NR + bin] = (uint16_t)((o + u) | (r << 12));
            }
            __syncthreads();
        }
        if (CF) {       // the units with the signal code: their places, then every (place, band, entering rung) parsed by a lane of its own
            if (tid == 0) s_nsig = 0;
            for (uint32_t o = tid; o < TP; o += NT) sig_slot[o] = 0xffu;
            __syncthreads();
            for (uint32_t o = tid; o < TP; o += NT)
                if ((sw[o] >> 10) & 1u) { const uint32_t j = atomicAdd(&s_nsig, 1u); if (j < E::NSIG) { sigpos[j] = (uint16_t)o; sig_slot[o] = (uint8_t)j; } }
            __syncthreads();
            const uint32_t nsig = s_nsig < E::NSIG ? s_nsig : E::NSIG;
            for (uint32_t i = tid; i < nsig * NR; i += NT) {                               // (a lane per place and entering rung: the bands differ only where the unit takes a factor in force, and only if theirs differ)
                const uint32_t j = i / NR, rin = i % NR, o = sigpos[j];
                uint32_t first = 0;
#pragma unroll
                for (uint32_t c = 0; c < B; c++) {
                    const uint8_t spec_c = (uint8_t)(S.cf >> (8 * c));
                    uint32_t v = first;
                    if (c == 0 || ((first & 0x8000u) && spec_c != (uint8_t)S.cf)) {
                        ReaderT<LdsWords> rd;
                        rd.init((LdsWords)words, sh + o, 32ull * (NP1 / 32 + 3));
                        uint32_t rg = rin, fl = 0;
                        uint8_t pc = spec_c, g[16];
                        const bool ok = parse_unit<uint8_t, CM_BEST>(rd, rg, pc, g, &fl);
                        const uint32_t end = (uint32_t)rd.position() - sh;
                        v = (!ok || (fl & 2u) || end >= 4096u) ? 0x10000u : end | ((rg & 7u) << 12) | ((fl & 1u) ? 0x8000u : 0u);
                        if (c == 0) first = v;
                    }
                    side[(j * B + c) * NR + rin] = v;
                }
            }
            __syncthreads();
        }
        if (k == 0) {
            for (uint32_t i = tid; i < E::BMW; i += NT) bm[i] = 0;
            __syncthreads();
            for (uint32_t key = tid; key < NKEY; key += NT) {                               // every state through the first window
                const uint32_t x = walk(key);
                G[key] = x;
                if ((x & E::KEYM) != E::X_STOP) atomicOr(&bm[(x & E::KEYM) >> 5], 1u << (x & 31u));
            }
            __syncthreads();
            // rank of every distinct exit: exclusive prefix of the bitmap words' bit counts (the scan's scratch: Xd, not yet in use)
            constexpr uint32_t PER = (E::BMW + NT - 1) / NT;
            uint32_t mine = 0;
            for (uint32_t i = 0; i < PER; i++) { const uint32_t w = tid * PER + i; if (w < E::BMW) mine += __popc(bm[w]); }
            Xd[tid] = mine;
            __syncthreads();
            for (uint32_t d = 1; d < NT; d <<= 1) {
                const uint32_t y = tid >= d ? Xd[tid - d] : 0u;
                __syncthreads();
                Xd[tid] += y;
                __syncthreads();
            }
            uint32_t run = Xd[tid] - mine;
            if (tid == NT - 1) s_D = Xd[tid];
            __syncthreads();
            D = s_D;
            if (D > dcap) {                                                                 // (uniform) more distinct exits than are carried (dcap <= DCAP; less: a test hook): the hop parses this super-window outright
                for (uint32_t key = tid; key < NKEY; key += NT) G[key] = E::X_STOP;
                return;
            }
            for (uint32_t i = 0; i < PER; i++) { const uint32_t w = tid * PER + i; if (w < E::BMW) { pf[w] = (uint16_t)run; run += __popc(bm[w]); } }
            __syncthreads();
            for (uint32_t w = tid; w < E::BMW; w += NT) {                                   // the distinct exits, in rank order
                uint32_t m = bm[w], j = pf[w];
                while (m) { const uint32_t b = __ffs(m) - 1; Xd[j++] = w * 32 + b; m &= m - 1; }
            }
            __syncthreads();
        } else {
            for (uint32_t j = tid; j < D; j += NT) {                                        // the distinct walks through this window
                const uint32_t x = Xd[j];
                if ((x & E::KEYM) == E::X_STOP) continue;
                const uint32_t y = walk(x & E::KEYM);
                Xd[j] = compose(x, y);
            }
            __syncthreads();
        }
    }
    for (uint32_t key = tid; key < NKEY; key += NT) {                                       // every state: its first-window exit, then what became of that
        const uint32_t e = G[key];
        if ((e & E::KEYM) == E::X_STOP) continue;
        const uint32_t k1 = e & E::KEYM, w = k1 >> 5;
        const uint32_t x = Xd[pf[w] + __popc(bm[w] & ((1u << (k1 & 31u)) - 1u))];
        G[key] = compose(e, x);
    }
    __syncthreads();
}

// the hop for rasters of B bands: entries {position lo, hi, block, rungs (4 bits a band)} {factors in force (a byte a band)}
template <uint32_t B, int MODE>
__global__ void __launch_bounds__(64) walk_exitB_chain_kernel(const DecArgs a0, const uint32_t *xg, uint32_t nsuper, uint32_t s_begin, uint32_t s_count, WalkState16 *states, uint4 *entries) {
    typedef exitB<B, MODE == CM_BEST> E;
    const DecArgs a = dec_for_tile(a0, blockIdx.x);
    if (threadIdx.x) return;
    WalkState16 *S = states + blockIdx.x;
    if (S->bad) return;
    const uint64_t nblocks = a.g.nblocks, P0 = S->P;
    const uint32_t spec = (uint32_t)S->cf;
    uint4 *en = entries + (uint64_t)blockIdx.x * 2 * (nsuper + 2), *hd = en + 2 * (nsuper + 1);
    uint64_t P = P0, U = S->unit / B;
    uint32_t rr = (uint32_t)S->rungs & ((1u << (4 * B)) - 1), s = 0, cf = spec;
    bool bad = false, done = false;
    if (s_begin) {
        const uint4 h = *hd;
        if (h.y) return;
        const uint4 e = en[2 * s_begin];
        P = (uint64_t)e.x | (uint64_t)e.y << 32; U = e.z; rr = e.w; s = s_begin; cf = en[2 * s_begin + 1].x;
        bad = h.x != s_begin;
    }
    const uint32_t s_end = s_begin + s_count < nsuper ? s_begin + s_count : nsuper;
    const uint32_t *x0 = xg + (uint64_t)blockIdx.x * s_count * E::NKEY;
    while (!bad) {
        en[2 * s] = make_uint4((uint32_t)P, (uint32_t)(P >> 32), (uint32_t)U, rr);
        en[2 * s + 1] = make_uint4(cf, 0u, 0u, 0u);
        if (U >= nblocks) { done = true; break; }
        if (s >= s_end) { bad = s >= nsuper; break; }
        if (P >= a.in_bits) { bad = true; break; }
        const uint64_t base = P0 + (uint64_t)s * E::SW;
        uint32_t key = (uint32_t)(P - base) << (3 * B);
        for (uint32_t c = 0; c < B; c++) key |= ((rr >> (4 * c)) & 7u) << (3 * c);
        const uint32_t x = x0[(uint64_t)(s - s_begin) * E::NKEY + key];
        s++;
        if ((x & E::KEYM) != E::X_STOP && !((x & E::X_DEP) && cf != spec)) {
            U += (x >> E::KEYB) & E::CNTM;
            const uint32_t k2 = x & E::KEYM;
            P = base + E::SW + (k2 >> (3 * B));
            rr = 0;
            for (uint32_t c = 0; c < B; c++) rr |= ((k2 >> (3 * c)) & 7u) << (4 * c);
            continue;
        }
        // this super-window by the units themselves: whole blocks up to the first that starts behind it
        atomicOr(a.status, 64u);                                                            // (not an error: says that the walk was handed to this lane)
        Reader rd;
        rd.init(a.in32, a.in_bit0 + P, a.in_bit0 + a.in_bits);
        uint32_t rung[B];
        uint8_t pc[B], g[16];
        for (uint32_t c = 0; c < B; c++) { rung[c] = (rr >> (4 * c)) & 15u; pc[c] = (uint8_t)(cf >> (8 * c)); }
        bool ok = true;
        const uint64_t end = base + E::SW;
        while (ok && U < nblocks) {
            const uint64_t pos = rd.position() - a.in_bit0;
            if (pos >= a.in_bits || pos >= end) break;
#pragma unroll
            for (uint32_t c = 0; c < B; c++) ok = parse_unit<uint8_t, MODE>(rd, rung[c], pc[c], g) && ok;
            U++;
        }
        P = rd.position() - a.in_bit0;
        rr = 0; cf = 0;
        for (uint32_t c = 0; c < B; c++) { rr |= rung[c] << (4 * c); cf |= (uint32_t)pc[c] << (8 * c); }
        if (!ok || (U < nblocks && (P < end || P - end >= E::PE))) { bad = true; break; }
    }
    *hd = make_uint4(s, done ? 1u : 0u, 0u, 0u);
    if (bad) { S->bad = 1u; atomicOr(a.status, 1u); }
}

template <uint32_t B, int MODE>
static bool launch_walk_exitB(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits) {
    typedef exitB<B, MODE == CM_BEST> E;
    const uint32_t nt = a.ntiles;
    const uint64_t ns = (max_bits + E::SW - 1) / E::SW;
    const size_t fixed = (((size_t)nt * sizeof(WalkState16) + 255) & ~(size_t)255) + (((size_t)nt * (ns + 2) * 32 + 255) & ~(size_t)255);
    if (ns == 0 || ns > 0x7fffffffu || tab_bytes < fixed + (size_t)nt * E::NKEY * 4) return false;
    const uint64_t fit = (tab_bytes - fixed) / ((size_t)nt * E::NKEY * 4);
    const uint32_t nsuper = (uint32_t)ns, slab = (uint32_t)(fit < ns ? fit : ns);
    WalkState16 *states = (WalkState16 *)tab;
    uint4 *entries = (uint4 *)((uint8_t *)tab + (((size_t)nt * sizeof(WalkState16) + 255) & ~(size_t)255));
    uint32_t *xg = (uint32_t *)((uint8_t *)tab + fixed);
    for (uint32_t s0 = 0; s0 < nsuper; s0 += slab) {
        const uint32_t cnt = nsuper - s0 < slab ? nsuper - s0 : slab;
        { ProfScope ps("dec_index_table", st);
          hipLaunchKernelGGL((walk_exitB_kernel<B, MODE == CM_BEST>), dim3(cnt, nt), dim3(E::THREADS), E::LDS_BYTES, st, a, xg, s0, cnt, (const WalkState16 *)states, a.wide_band == 18 ? 64u : E::DCAP); }
        ProfScope ps("dec_index_serial", st);
        hipLaunchKernelGGL((walk_exitB_chain_kernel<B, MODE>), dim3(nt), dim3(64), 0, st, a, (const uint32_t *)xg, nsuper, s0, cnt, states, entries);
    }
    ProfScope ps("dec_index_serial", st);
    hipLaunchKernelGGL((walk_exit_units_kernel<uint8_t, MODE>), dim3(nsuper, nt), dim3(64), ((E::SW + E::PE) / 32 + 4) * 4, st, a, (const WalkState16 *)states, (const uint4 *)entries, nsuper, E::SW, E::PE);
    return true;
}

// memory of the exit walk: states, entries, and the exits of as many super-windows as fit (at least one a tile)
template <uint32_t UB> static bool walk_exit_layout(uint32_t nt, uint64_t max_bits, size_t tab_bytes, uint32_t *nsuper, uint32_t *slab, size_t *x_off) {
    typedef exitW<UB> E;
    const uint64_t ns = (max_bits + E::SW - 1) / E::SW;
    const size_t fixed = (((size_t)nt * sizeof(WalkState16) + 255) & ~(size_t)255) + (((size_t)nt * (ns + 2) * 32 + 255) & ~(size_t)255);
    if (ns == 0 || ns > 0x7fffffffu || tab_bytes < fixed + (size_t)nt * E::NX * 4) return false;
    const uint64_t fit = (tab_bytes - fixed) / ((size_t)nt * E::NX * 4);
    *nsuper = (uint32_t)ns; *slab = (uint32_t)(fit < ns ? fit : ns); *x_off = fixed;
    return true;
}
template <uint32_t UB, typename T, int MODE>
static bool launch_walk_exit(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits) {
    typedef exitW<UB> E;
    uint32_t nsuper = 0, slab = 0;
    size_t x_off = 0;
    const uint32_t nt = a.ntiles;
    if (!walk_exit_layout<UB>(nt, max_bits, tab_bytes, &nsuper, &slab, &x_off)) return false;
    WalkState16 *states = (WalkState16 *)tab;
    uint4 *entries = (uint4 *)((uint8_t *)tab + (((size_t)nt * sizeof(WalkState16) + 255) & ~(size_t)255));
    uint32_t *xg = (uint32_t *)((uint8_t *)tab + x_off);
    for (uint32_t s0 = 0; s0 < nsuper; s0 += slab) {
        const uint32_t cnt = nsuper - s0 < slab ? nsuper - s0 : slab;
        { ProfScope ps("dec_index_table", st);
          hipLaunchKernelGGL((walk_exitW_kernel<UB, MODE == CM_BEST>), dim3(cnt, nt), dim3(E::THREADS), E::LDS_BYTES, st, a, xg, s0, cnt, (const WalkState16 *)states); }
        ProfScope ps("dec_index_serial", st);
        hipLaunchKernelGGL((walk_exit_chain_kernel<UB, MODE>), dim3(nt), dim3(64), 0, st, a, (const uint32_t *)xg, nsuper, s0, cnt, states, entries);
    }
    ProfScope ps("dec_index_serial", st);
    hipLaunchKernelGGL((walk_exit_units_kernel<T, MODE>), dim3(nsuper, nt), dim3(64), ((E::SW + E::PE) / 32 + 4) * 4, st, a, (const WalkState16 *)states, (const uint4 *)entries, nsuper, E::SW, E::PE);
    return true;
}

// Slabs of the streams are tabulated by the whole chip, then walked by a workgroup per tile, slab after slab; the
// table of the next slab is built (on a stream of its own, in the other half of the memory) while this one is walked.
// tab: [walk state per tile][windows of table rows per tile] x 2; max_bits: the longest stream of the call.
template <typename TABLE, typename CHAIN>
static void walk_in_slabs(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits, uint32_t cw, uint32_t win_u4, size_t state_size,
                          TABLE &&launch_table, CHAIN &&launch_chain) {
    const uint32_t nt = a.ntiles;
    const size_t state_bytes = ((size_t)nt * state_size + 255) & ~(size_t)255;
    uint8_t *base = (uint8_t *)tab;
    const uint64_t need = (max_bits + cw - 1) / cw;                         // windows of the longest stream
    const uint64_t cap = (tab_bytes - state_bytes) / ((uint64_t)win_u4 * 16 * nt);  // windows per tile the memory holds
    // one round when the streams are short (nothing to overlap, and a stream costs more to create than it saves)
    if (need <= cap && need * cw <= (8u << 20)) {
        const uint64_t pitch = need * win_u4;
        uint4 *rows = (uint4 *)(base + state_bytes);
        { ProfScope ps("dec_index_table", st); launch_table(st, rows, (uint64_t)0, (uint32_t)need, pitch); }
        ProfScope ps("dec_index_serial", st);
        launch_chain(st, rows, (uint64_t)0, (uint32_t)need, pitch, base, 1u);
        return;
    }
    // rounds of at most half the memory, and at least four of them
    uint64_t nwin = cap / 2;
    if (nwin > (need + 3) / 4) nwin = (need + 3) / 4;
    if (nwin < 16) nwin = 16;                                               // (walk_table_min_bytes holds 2 x 16)
    if (nwin > 0x7fffffffu / win_u4) nwin = 0x7fffffffu / win_u4;
    const uint64_t pitch = nwin * win_u4;                                   // in rows of sixteen bytes
    uint4 *rows[2] = {(uint4 *)(base + state_bytes), (uint4 *)(base + state_bytes) + pitch * nt};
    hipStream_t aux = nullptr;
    hipEvent_t ev_tab[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr}, ev_start = nullptr;
    bool ok = hipStreamCreateWithFlags(&aux, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < 2 && ok; i++)
        ok = hipEventCreateWithFlags(&ev_tab[i], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ev_done[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&ev_start, hipEventDisableTiming) == hipSuccess;
    if (ok) { (void)hipEventRecord(ev_start, st); (void)hipStreamWaitEvent(aux, ev_start, 0); }
    hipStream_t tst = ok ? aux : st;                                        // (no second stream: everything in order on the caller's)
    uint32_t first = 1, j = 0;
    for (uint64_t s0 = 0; s0 < max_bits; s0 += nwin * cw, first = 0, j++) {
        const int h = j & 1;
        if (ok && j >= 2) (void)hipStreamWaitEvent(aux, ev_done[h], 0);     // the walk of two rounds ago has left this half
        { ProfScope ps("dec_index_table", tst); launch_table(tst, rows[h], s0, (uint32_t)nwin, pitch); }
        if (ok) { (void)hipEventRecord(ev_tab[h], aux); (void)hipStreamWaitEvent(st, ev_tab[h], 0); }
        { ProfScope ps("dec_index_serial", st); launch_chain(st, rows[h], s0, (uint32_t)nwin, pitch, base, first); }
        if (ok) (void)hipEventRecord(ev_done[h], st);
    }
    // (destroying a stream or an event with work pending is deferred by the runtime until that work is done)
    for (int i = 0; i < 2; i++) { if (ev_tab[i]) (void)hipEventDestroy(ev_tab[i]); if (ev_done[i]) (void)hipEventDestroy(ev_done[i]); }
    if (ev_start) (void)hipEventDestroy(ev_start);
    if (aux) (void)hipStreamDestroy(aux);
}
// Plain single-band 32/64-bit COMMON-FACTOR streams through the same exits (units with the signal code are parsed outright
// inside the walk).  False: not taken (no memory for it) -- the caller parses the stream with one lane.
static bool walk_exit_lds_ok();
bool launch_dec_walk_best(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits) {
    if (a.g.mode != CM_BEST || !walk_exit_lds_ok() || (a.g.tsz == 1 && a.g.ulen_sz != 4)) return false;    // (8-bit: the lane-per-block decoder's block table)
    if (a.g.bands == 3 && a.g.tsz == 1) {          // 8-bit RGB
        WalkState16 *states = (WalkState16 *)tab;
        { ProfScope ps("dec_index_serial", st);
          hipLaunchKernelGGL(walk_exit_zero_kernel<uint8_t>, dim3((uint32_t)((a.g.nseg + 255) / 256), a.ntiles), dim3(256), 0, st, a);
          hipLaunchKernelGGL((walk_probe_kernel<uint8_t, CM_BEST>), dim3(a.ntiles), dim3(64), 0, st, a, states, 8u); }
        return launch_walk_exitB<3, CM_BEST>(a, st, tab, tab_bytes, max_bits);
    }
    if (a.g.bands != 1) return false;
    uint32_t ns = 0, slab = 0; size_t xo = 0;
    const uint32_t nt = a.ntiles;
    if (!(a.g.tsz == 1 ? walk_exit_layout<3>(nt, max_bits, tab_bytes, &ns, &slab, &xo) : a.g.tsz == 2 ? walk_exit_layout<4>(nt, max_bits, tab_bytes, &ns, &slab, &xo) : a.g.tsz == 4 ? walk_exit_layout<5>(nt, max_bits, tab_bytes, &ns, &slab, &xo) : walk_exit_layout<6>(nt, max_bits, tab_bytes, &ns, &slab, &xo))) return false;
    WalkState16 *states = (WalkState16 *)tab;
    { ProfScope ps("dec_index_serial", st);
      const dim3 zg((uint32_t)((a.g.nseg + 255) / 256), nt);
      if (a.g.tsz == 1) { hipLaunchKernelGGL(walk_exit_zero_kernel<uint8_t>, zg, dim3(256), 0, st, a); hipLaunchKernelGGL((walk_probe_kernel<uint8_t, CM_BEST>), dim3(nt), dim3(64), 0, st, a, states, 8u); }
      else if (a.g.tsz == 2) { hipLaunchKernelGGL(walk_exit_zero_kernel<uint16_t>, zg, dim3(256), 0, st, a); hipLaunchKernelGGL((walk_probe_kernel<uint16_t, CM_BEST>), dim3(nt), dim3(64), 0, st, a, states, 16u); }
      else if (a.g.tsz == 4) { hipLaunchKernelGGL(walk_exit_zero_kernel<uint32_t>, zg, dim3(256), 0, st, a); hipLaunchKernelGGL((walk_probe_kernel<uint32_t, CM_BEST>), dim3(nt), dim3(64), 0, st, a, states, 16u); }
      else { hipLaunchKernelGGL(walk_exit_zero_kernel<uint64_t>, zg, dim3(256), 0, st, a); hipLaunchKernelGGL((walk_probe_kernel<uint64_t, CM_BEST>), dim3(nt), dim3(64), 0, st, a, states, 16u); } }
    return a.g.tsz == 1 ? launch_walk_exit<3, uint8_t, CM_BEST>(a, st, tab, tab_bytes, max_bits) : a.g.tsz == 2 ? launch_walk_exit<4, uint16_t, CM_BEST>(a, st, tab, tab_bytes, max_bits)
         : a.g.tsz == 4 ? launch_walk_exit<5, uint32_t, CM_BEST>(a, st, tab, tab_bytes, max_bits) : launch_walk_exit<6, uint64_t, CM_BEST>(a, st, tab, tab_bytes, max_bits);
}
template <uint32_t U, uint32_t N> struct WideTag { static constexpr uint32_t UB_ = U, NR_ = N; };
static bool walk_lds_attributes() {
    static const bool lds_ok = [] {
        bool ok = true;
        ok = ok && hipFuncSetAttribute((const void *)walk_chain_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, chain::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_chain_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, chain::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_chain_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, chain::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_chain16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, chain16::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_chainW_kernel<5, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, chainW<5, 8>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_chainW_kernel<6, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, chainW<6, 8>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_chainW_kernel<5, 14>, hipFuncAttributeMaxDynamicSharedMemorySize, chainW<5, 14>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_chainW_kernel<6, 14>, hipFuncAttributeMaxDynamicSharedMemorySize, chainW<6, 14>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_chainW_kernel<5, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, chainW<5, 16>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_chainW_kernel<6, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, chainW<6, 16>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitB_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, exitB<3, false>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitB_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, exitB<3, true>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitW_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, exitW<3>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitW_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, exitW<4>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitW_kernel<5, false>, hipFuncAttributeMaxDynamicSharedMemorySize, exitW<5>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitW_kernel<6, false>, hipFuncAttributeMaxDynamicSharedMemorySize, exitW<6>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitW_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, exitW<3>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitW_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, exitW<4>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitW_kernel<5, true>, hipFuncAttributeMaxDynamicSharedMemorySize, exitW<5>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitW_kernel<6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, exitW<6>::LDS_BYTES) == hipSuccess;
        return ok;
    }();
    return lds_ok;
}
static bool walk_exit_lds_ok() { return walk_lds_attributes(); }
void launch_dec_walk_table(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits) {
    const bool lds_ok = walk_lds_attributes();
    const uint32_t nt = a.ntiles;
    // (Exits are the whole chip's work for one stream, a chain is one workgroup's: a batch of many tiles is walked sooner by a
    // chain a tile, side by side -- 32 tiles of 4096^2 x 3: 0.30 s by chains, 0.80 s by exits.)
    if (a.g.tsz <= 2 && a.g.bands == 1 && a.wide_band == 16 && lds_ok && nt <= 16) {       // one band of 8- or 16-bit data: the exits too (the band is all the rungs)
        WalkState16 *states = (WalkState16 *)tab;
        { ProfScope ps("dec_index_serial", st);
          if (a.g.tsz == 1) hipLaunchKernelGGL((walk_probe_kernel<uint8_t, CM_FTL>), dim3(nt), dim3(64), 0, st, a, states, 8u, 16u);
          else hipLaunchKernelGGL((walk_probe_kernel<uint16_t, CM_FTL>), dim3(nt), dim3(64), 0, st, a, states, 16u, 16u); }
        if (a.g.tsz == 1 ? launch_walk_exit<3, uint8_t, CM_FTL>(a, st, tab, tab_bytes, max_bits) : launch_walk_exit<4, uint16_t, CM_FTL>(a, st, tab, tab_bytes, max_bits)) return;
    }
    if (a.g.tsz == 1 && a.g.bands == 3 && (a.wide_band == 16 || a.wide_band == 18) && lds_ok && nt <= 4) {    // 8-bit RGB: exits with the rung of every band in the state (18: a test hook, see dcap)
        WalkState16 *states = (WalkState16 *)tab;
        { ProfScope ps("dec_index_serial", st);
          hipLaunchKernelGGL((walk_probe_kernel<uint8_t, CM_FTL>), dim3(nt), dim3(64), 0, st, a, states, 8u, 8u); }
        if (launch_walk_exitB<3, CM_FTL>(a, st, tab, tab_bytes, max_bits)) return;
    }
    if (a.g.tsz >= 4) {         // 32/64-bit FTL/BASE: the first segment parsed outright (band of rungs, entry state), then table + chain
        WalkState16 *states = (WalkState16 *)tab;
        const uint32_t nr = 16u;    // (a band of eight rungs and a byte-entry table of fourteen were built and measured: DESIGN.md section 4, "Tried and measured")
        const bool exits = a.g.bands == 1 && a.wide_band == 16 && lds_ok && nt <= 16;      // one band: exits of super-windows composed, a hop per 32768 bits (wide_band 17: the chain, a test hook)
        auto probe = [&](uint32_t few) {
            ProfScope ps("dec_index_serial", st);
            if (a.g.tsz == 4) hipLaunchKernelGGL((walk_probe_kernel<uint32_t, CM_FTL>), dim3(nt), dim3(64), 0, st, a, states, nr, few);
            else hipLaunchKernelGGL((walk_probe_kernel<uint64_t, CM_FTL>), dim3(nt), dim3(64), 0, st, a, states, nr, few);
        };
        probe(exits ? 16u : 0u);
        if (exits) {
            if (a.g.tsz == 4 ? launch_walk_exit<5, uint32_t, CM_FTL>(a, st, tab, tab_bytes, max_bits) : launch_walk_exit<6, uint64_t, CM_FTL>(a, st, tab, tab_bytes, max_bits)) return;
            probe(0u);                                                                      // (no memory for the exits: the chain wants the whole first segment parsed)
        }
        auto run = [&](auto tag) {
            constexpr uint32_t UB = decltype(tag)::UB_, NRB = decltype(tag)::NR_;
            typedef chainW<UB, NRB> W;
            walk_in_slabs(a, st, tab, tab_bytes, max_bits, W::CW, W::WIN_U4, sizeof(WalkState16),
                [&](hipStream_t s, uint4 *rows, uint64_t s0, uint32_t nwin, uint64_t pitch) {
                    hipLaunchKernelGGL((walk_tableW_kernel<UB, NRB>), dim3(nwin * (W::CW / W::TCW), nt), dim3(256), 0, s, a, rows, s0, nwin, pitch, (const WalkState16 *)states); },
                [&](hipStream_t s, const uint4 *rows, uint64_t s0, uint32_t nwin, uint64_t pitch, uint8_t *sts, uint32_t) {
                    hipLaunchKernelGGL((walk_chainW_kernel<UB, NRB>), dim3(nt), dim3(WIDE_THREADS), W::LDS_BYTES, s, a, rows, s0, nwin, pitch, (WalkState16 *)sts); });
        };
        if (a.g.tsz == 4) run(WideTag<5, 16>()); else run(WideTag<6, 16>());
        return;
    }
    if (a.g.tsz == 2) {
        walk_in_slabs(a, st, tab, tab_bytes, max_bits, chain16::CW, chain16::WIN_U4, sizeof(WalkState16),
            [&](hipStream_t s, uint4 *rows, uint64_t s0, uint32_t nwin, uint64_t pitch) {
                hipLaunchKernelGGL(walk_table16_kernel, dim3(nwin, nt), dim3(256), 0, s, a, rows, s0, nwin, pitch); },
            [&](hipStream_t s, const uint4 *rows, uint64_t s0, uint32_t nwin, uint64_t pitch, uint8_t *states, uint32_t first) {
                hipLaunchKernelGGL(walk_chain16_kernel, dim3(nt), dim3(512), chain16::LDS_BYTES, s, a, rows, s0, nwin, pitch, (WalkState16 *)states, first); });
        return;
    }
    walk_in_slabs(a, st, tab, tab_bytes, max_bits, chain::CW, chain::ROWS, sizeof(WalkState),
        [&](hipStream_t s, uint4 *rows, uint64_t s0, uint32_t nwin, uint64_t pitch) {
            hipLaunchKernelGGL(walk_table_kernel, dim3(nwin, nt), dim3(256), 0, s, a, rows, s0, nwin, pitch); },
        [&](hipStream_t s, const uint4 *rows, uint64_t s0, uint32_t nwin, uint64_t pitch, uint8_t *states, uint32_t first) {
            using namespace chain;
            WalkState *ws = (WalkState *)states;
            if (a.g.bands == 1) hipLaunchKernelGGL(walk_chain_kernel<1>, dim3(nt), dim3(512), LDS_BYTES, s, a, rows, s0, nwin, pitch, ws, first);
            else if (a.g.bands == 3) hipLaunchKernelGGL(walk_chain_kernel<3>, dim3(nt), dim3(512), LDS_BYTES, s, a, rows, s0, nwin, pitch, ws, first);
            else hipLaunchKernelGGL(walk_chain_kernel<4>, dim3(nt), dim3(512), LDS_BYTES, s, a, rows, s0, nwin, pitch, ws, first); });
}
// bytes of table memory that take `max_bits` of every stream in one round (16-bit data: 32 bytes a stream bit)
// (32/64-bit data: sized for the table of sixteen rungs, 32 bytes a stream bit in windows of 1440 / 960 positions; the table of eight is half of it)
static uint32_t walk_cw(uint32_t tsz) { return tsz == 2 ? chain16::CW : tsz == 4 ? chainW<5, 16>::CW : tsz == 8 ? chainW<6, 16>::CW : chain::CW; }
size_t walk_table_bytes(uint32_t ntiles, uint64_t max_bits, uint32_t tsz) {
    const uint32_t cw = walk_cw(tsz), win_bytes = tsz == 1 ? chain::WIN_BYTES : cw * 32;
    const uint64_t need = (max_bits + cw - 1) / cw;
    return (((size_t)ntiles * sizeof(WalkState16) + 255) & ~(size_t)255) + (size_t)win_bytes * ntiles * need + 4096;
}
size_t walk_table_min_bytes(uint32_t ntiles, uint32_t tsz) { return walk_table_bytes(ntiles, 2 * 16 * walk_cw(tsz), tsz); }

void launch_dec_walk(const DecArgs &a, hipStream_t st) {
    if (a.ix) {                             // the containers' own restart tables: a lane per entry
        const dim3 grid((a.ix_K + 63) / 64, a.ntiles), block(64);
        // unit lengths staged in LDS when a lane's share is small enough (it is when an entry is one index segment)
        // unit lengths leave through a small per-lane ring in LDS (sixteen blocks for 8-bit data, 64 bytes otherwise)
        const uint32_t stage = 1;
        const bool bt = a.g.tsz == 1;                                       // bands at compile time: a ring of sixteen blocks per lane instead of the whole piece
        const size_t lds = 64 * walk_winp(a.g.tsz == 1 ? 3 : a.g.tsz == 2 ? 4 : 5) * 4 + (bt && stage ? 64 * 16 * (size_t)a.g.bands : stage ? 64 * 64 : 0) + (a.g.tsz >= 4 ? 64 * MAXBANDS : 0);
        if (a.g.tsz == 1 && a.g.bands == 1) hipLaunchKernelGGL((dec_walk_lanes_kernel<3, 1>), grid, block, lds, st, a, stage);
        else if (a.g.tsz == 1 && a.g.bands == 3) hipLaunchKernelGGL((dec_walk_lanes_kernel<3, 3>), grid, block, lds, st, a, stage);
        else if (a.g.tsz == 1) hipLaunchKernelGGL((dec_walk_lanes_kernel<3, 4>), grid, block, lds, st, a, stage);
        else if (a.g.tsz == 2) hipLaunchKernelGGL((dec_walk_lanes_kernel<4, 0>), grid, block, lds, st, a, stage);
        else if (a.g.tsz == 4) hipLaunchKernelGGL((dec_walk_lanes_kernel<5, 0>), grid, block, lds, st, a, stage);
        else hipLaunchKernelGGL((dec_walk_lanes_kernel<6, 0>), grid, block, lds, st, a, stage);
        return;
    }

    set_error("launch_dec_walk: no restart table (the caller parses such a stream with one lane)", 0);
}
void launch_prev_scan(const DecArgs &a, hipStream_t st) {
    if (a.g.tsz == 1) hipLaunchKernelGGL(prev_scan_kernel<uint8_t>, dim3(a.ntiles, a.g.bands), dim3(1024), 0, st, a);
    else if (a.g.tsz == 4) hipLaunchKernelGGL(prev_scan_kernel<uint32_t>, dim3(a.ntiles, a.g.bands), dim3(1024), 0, st, a);
    else if (a.g.tsz == 8) hipLaunchKernelGGL(prev_scan_kernel<uint64_t>, dim3(a.ntiles, a.g.bands), dim3(1024), 0, st, a);
    else hipLaunchKernelGGL(prev_scan_kernel<uint16_t>, dim3(a.ntiles, a.g.bands), dim3(1024), 0, st, a);
}

}  // namespace qb3dev
