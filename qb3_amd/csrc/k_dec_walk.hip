// qb3_amd/csrc/k_dec_walk.hip -- index-less 8/16-bit FTL/BASE streams: find the unit lengths by walking
#include "qb3_kernels.h"

namespace qb3dev {

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

// ---- the same walk from the container's restart table: ONE LANE per restart point ---------------------------
// With the "ix" chunks in the container (include/qb3x.h) the stream is cut into K independent walks of about a
// thousand units.  A lane walks its own piece straight from global memory (a 64-bit bit buffer, the next word always
// requested one refill ahead); lanes of a wave run in lockstep because a unit is always a switch and sixteen codes
// whatever its rung (only rung 0 takes a short side path).  8-bit data: B is a template parameter, the unit lengths
// of four blocks leave as B dwords.
template <uint32_t UB> struct WalkBits {
    const uint32_t *in;
    uint64_t endw, wp, buf;         // wp: index of the word held in nxt
    uint32_t n, nxt;
    __device__ __forceinline__ uint32_t ld(uint64_t w) const { return w < endw ? in[w] : 0u; }
    __device__ __forceinline__ void init(const uint32_t *p, uint64_t bitpos, uint64_t endbit) {
        in = p; endw = (endbit + 31) >> 5;
        const uint64_t w = bitpos >> 5;
        const uint32_t sh = (uint32_t)bitpos & 31;
        buf = (uint64_t)(ld(w) >> sh); n = 32 - sh; wp = w + 1; nxt = ld(wp);
    }
    __device__ __forceinline__ void refill() {      // afterwards n >= 33
        if (n <= 32) { buf |= (uint64_t)nxt << n; n += 32; nxt = ld(++wp); }
    }
    __device__ __forceinline__ void skip(uint32_t k) { buf >>= k; n -= k; }
    __device__ __forceinline__ uint64_t position() const { return 32 * wp - n; }
    // length of one unit of a band whose rung is `rung` (updated); sets bad on a signal code
    __device__ __forceinline__ uint32_t unit(uint32_t &rung, bool &bad) {
        constexpr uint32_t UMASK = (1u << UB) - 1, NRUNG = 1u << UB;
        refill();                                   // >= 33 bits: the switch code is at most UB + 2
        uint32_t x = (uint32_t)buf, ulen = 1;
        if (x & 1) {                                // code at rung UB - 1 (reference QB3decode.h:97-116)
            constexpr uint32_t r = UB - 1, half = 1u << (r - 1), top = 1u << r;
            x >>= 1;
            uint32_t m, len;
            if (!(x & 1)) { m = (x & (top - 1)) >> 1; len = r; }
            else if (!(x & 2)) { m = ((x >> 2) & (half - 1)) | half; len = r + 1; }
            else { m = ((x >> 2) & (top - 1)) | top; len = r + 2; }
            ulen = 1 + len;
            bad = bad || m == NRUNG - 2;            // signal: a common-factor stream, not for this walker
            const uint32_t delta = (m & 1) ? (NRUNG - (m + 1) / 2) & UMASK : m / 2 + 1;
            rung = (rung + delta) & UMASK;
        }
        skip(ulen);
        if (rung == 0) {                            // one flag, then 16 raw bits
            refill();
            const uint32_t l = ((uint32_t)buf & 1) ? 17 : 1;
            skip(l);
            return ulen + l;
        }
        const uint32_t kr = rung * 0x01010101u + 0x02000100u;      // code length by the low two bits: r, r+1, r, r+2
        if (UB == 3) {                              // three codes are at most 27 bits: one refill, one 64-bit shift
#pragma unroll
            for (int g = 0; g < 6; g++) {
                refill();
                uint32_t b = (uint32_t)buf, acc = 0;
#pragma unroll
                for (int i = 0; i < (g == 5 ? 1 : 3); i++) {
                    const uint32_t len = __builtin_amdgcn_ubfe(kr, (b & 3u) << 3, 8);
                    b >>= len; acc += len;
                }
                skip(acc); ulen += acc;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) {          // one code of up to 17 bits per refill
                refill();
                const uint32_t len = __builtin_amdgcn_ubfe(kr, ((uint32_t)buf & 3u) << 3, 8);
                skip(len); ulen += len;
            }
        }
        return ulen;
    }
};

template <uint32_t UB, int BT>     // BT: bands at compile time (8-bit data), 0: run time
__global__ void __launch_bounds__(64) dec_walk_lanes_kernel(const DecArgs a) {
    const uint32_t k = blockIdx.x * 64 + threadIdx.x;
    const uint32_t B = BT ? (uint32_t)BT : a.g.bands, NB = a.g.seg_blocks, nblocks = (uint32_t)a.g.nblocks;
    const bool live = k < a.ix_K;
    const uint32_t kk = live ? k : a.ix_K - 1;
    const uint8_t *e = ix_entry_at(a.ix, a.ix_per_chunk, a.ix_E, a.ix_pad, kk);
    uint64_t bp = 0;
#pragma unroll
    for (uint32_t i = 0; i < 6; i++) bp |= (uint64_t)e[i] << (8 * i);
    const uint32_t gb0 = kk * a.ix_blocks;
    const uint32_t nb = !live ? 0u : (nblocks - gb0 < a.ix_blocks ? nblocks - gb0 : a.ix_blocks);
    uint64_t seg = gb0 / NB;
    WalkBits<UB> rd;
    rd.init(a.in32, a.in_bit0 + bp, a.in_bit0 + a.in_bits);
    bool bad = false;
    if (BT) {
        constexpr int BB = BT ? BT : 1;
        uint32_t rung[BB];
#pragma unroll
        for (int c = 0; c < BB; c++) rung[c] = e[6 + c] & 15u;
        uint8_t *ul8 = (uint8_t *)a.idx.ulen + (uint64_t)gb0 * BB;
        for (uint32_t b4 = 0; b4 < a.ix_blocks; b4 += 4) {     // ix_blocks is a multiple of the segment size, itself of 4
            uint32_t pk[BB];
#pragma unroll
            for (int c = 0; c < BB; c++) pk[c] = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t blk = b4 + q;
                if (blk < nb && blk % NB == 0) {
                    a.idx.bitpos[seg] = rd.position() - a.in_bit0;
#pragma unroll
                    for (int c = 0; c < BB; c++) a.idx.rung[seg * BB + c] = (uint8_t)rung[c];
                    seg++;
                }
#pragma unroll
                for (int c = 0; c < BB; c++) {
                    const uint32_t ulen = rd.unit(rung[c], bad);
                    const int j = q * BB + c;               // byte of the group
                    pk[j >> 2] |= ulen << (8 * (j & 3));
                }
            }
            if (b4 + 4 <= nb) {
#pragma unroll
                for (int c = 0; c < BB; c++) ((uint32_t *)(ul8 + (uint64_t)b4 * BB))[c] = pk[c];
            } else {
#pragma unroll
                for (int j = 0; j < 4 * BB; j++)
                    if (b4 + j / BB < nb) ul8[(uint64_t)b4 * BB + j] = (uint8_t)(pk[j >> 2] >> (8 * (j & 3)));
            }
        }
    } else {
        uint64_t R = 0;                                         // rungs, 4 bits per band
        for (uint32_t c = 0; c < B; c++) R |= (uint64_t)(e[6 + c] & 15u) << (4 * c);
        for (uint32_t blk = 0; blk < a.ix_blocks; blk++) {
            const bool act = blk < nb;
            if (act && blk % NB == 0) {
                a.idx.bitpos[seg] = rd.position() - a.in_bit0;
                for (uint32_t c = 0; c < B; c++) a.idx.rung[seg * B + c] = (uint8_t)((R >> (4 * c)) & 15u);
                seg++;
            }
            for (uint32_t c = 0; c < B; c++) {
                uint32_t rung = (uint32_t)(R >> (4 * c)) & 15u;
                const uint32_t ulen = rd.unit(rung, bad);
                R = (R & ~(15ull << (4 * c))) | ((uint64_t)rung << (4 * c));
                if (act) {
                    if (UB == 3) ((uint8_t *)a.idx.ulen)[((uint64_t)gb0 + blk) * B + c] = (uint8_t)ulen;
                    else ((uint16_t *)a.idx.ulen)[((uint64_t)gb0 + blk) * B + c] = (uint16_t)ulen;
                }
            }
        }
    }
    if (bad && live) atomicOr(a.status, 1u);
}

void launch_dec_walk(const DecArgs &a, hipStream_t st) {
    if (a.ix && a.ntiles == 1) {            // the container's own restart table: a lane per entry
        const dim3 grid((a.ix_K + 63) / 64), block(64);
        if (a.g.tsz == 1 && a.g.bands == 1) hipLaunchKernelGGL((dec_walk_lanes_kernel<3, 1>), grid, block, 0, st, a);
        else if (a.g.tsz == 1 && a.g.bands == 3) hipLaunchKernelGGL((dec_walk_lanes_kernel<3, 3>), grid, block, 0, st, a);
        else if (a.g.tsz == 1) hipLaunchKernelGGL((dec_walk_lanes_kernel<3, 4>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((dec_walk_lanes_kernel<4, 0>), grid, block, 0, st, a);
        return;
    }

    const dim3 wg(a.ntiles, 1);
    if (a.g.tsz == 1) hipLaunchKernelGGL(dec_walk_kernel<3>, wg, dim3(64), 0, st, a);
    else hipLaunchKernelGGL(dec_walk_kernel<4>, wg, dim3(64), 0, st, a);
}
void launch_prev_scan(const DecArgs &a, hipStream_t st) {
    if (a.g.tsz == 1) hipLaunchKernelGGL(prev_scan_kernel<uint8_t>, dim3(a.ntiles, a.g.bands), dim3(1024), 0, st, a);
    else hipLaunchKernelGGL(prev_scan_kernel<uint16_t>, dim3(a.ntiles, a.g.bands), dim3(1024), 0, st, a);
}

}  // namespace qb3dev
