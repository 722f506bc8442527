// qb3_amd/csrc/k_dec_walk.hip -- index-less streams: find the unit lengths (and segment entries) by walking
#include "qb3_walk.h"

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

void launch_dec_walk_table(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits) {
    const bool lds_ok = walk_chain_lds_ok() && walk_exit_lds_ok();
    const uint32_t nt = a.ntiles;
    // (Exits are the whole chip's work for one stream, a chain is one workgroup's: a batch of many tiles is walked sooner by a
    // chain a tile, side by side -- 32 tiles of 4096^2 x 3: 0.30 s by chains, 0.80 s by exits.)
    // one band of any width: the exits (8- and 16-bit data: the band is all the rungs; wide_band 17: the chain, a test hook)
    if (a.g.bands == 1 && a.wide_band == 16 && lds_ok && nt <= 16 && walk_exits_one_band(a, st, tab, tab_bytes, max_bits)) return;
    // 8-bit RGB, and two bands: exits with the rung of every band in the state (18: a test hook, see dcap)
    // (a super-window costs what its first window's walks cost -- 2.2 to 3.5 ms of one CU whatever the stream's length -- so a SHORT stream, a
    // tile of 512 x 512 and less, is walked sooner by the chain: measured crossovers, tools/small_plain.sh)
    const uint64_t exits_from = tuning().exits_from >= 0 ? (uint64_t)tuning().exits_from : a.g.tsz == 2 ? 2500000 : a.g.bands == 2 ? 1200000 : 4500000;
    if (((a.g.tsz == 1 && (a.g.bands == 3 || a.g.bands == 2)) || (a.g.tsz == 2 && a.g.bands == 2 && a.g.mode != CM_BEST)) && (a.wide_band == 16 || a.wide_band == 18) && lds_ok && nt <= 4 &&
        (max_bits >= exits_from || a.wide_band == 18) && walk_exits_rgb(a, st, tab, tab_bytes, max_bits)) return;
    if (a.g.tsz >= 4) walk_chain_wide(a, st, tab, tab_bytes, max_bits);
    else if (a.g.tsz == 2) walk_chain_16bit(a, st, tab, tab_bytes, max_bits);
    else if (a.g.bands <= 4 && a.g.mode != CM_BEST) walk_chain_8bit(a, st, tab, tab_bytes, max_bits);        // (the hand-ordered loops: one to four bands)
    else walk_chain_8bit_any(a, st, tab, tab_bytes, max_bits);          // more than 4 bands: the 16-bit chain's kernels with eight rungs
}

// bytes of table memory that take `max_bits` of every stream in one round (16-bit data: 32 bytes a stream bit)
// (32/64-bit data: sized for the table of sixteen rungs, 32 bytes a stream bit in windows of 1440 / 960 positions; the table of eight is half of it)
size_t walk_table_bytes(uint32_t ntiles, uint64_t max_bits, uint32_t tsz) {
    const uint32_t cw = walk_cw(tsz), win_bytes = walk_win_bytes(tsz);
    const uint64_t need = (max_bits + cw - 1) / cw;
    return (((size_t)ntiles * sizeof(WalkState16) + 255) & ~(size_t)255) + (size_t)win_bytes * ntiles * need + 4096;
}
size_t walk_table_min_bytes(uint32_t ntiles, uint32_t tsz) { return walk_table_bytes(ntiles, 2 * 16 * walk_cw(tsz), tsz); }
// ... and what the walk that will actually run wants: a raster that walks by exits needs 2-14 bytes a stream bit, not the
// chains' 16-32 (a decoder handle keeps this memory, and its pool after it)
size_t walk_memory_bytes(const Geometry &g, uint32_t ntiles, uint64_t max_bits) {
    const size_t chain = walk_table_bytes(ntiles, max_bits, g.tsz), least = walk_table_min_bytes(ntiles, g.tsz);
    const size_t ex = walk_exit_bytes(g.tsz, g.bands, g.mode == CM_BEST, ntiles, max_bits);
    if (!ex) return chain;
    return ex < least ? least : ex;       // (the chain must still fit as the fallback's first rung)
}

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
