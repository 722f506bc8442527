// qb3_amd/csrc/k_dec_generic.hip -- generic decoders: unit-parallel (dec3_kernel), lane per segment (dec_kernel), serial index rebuild
#include "qb3_wide.h"
#include "qb3_walk.h"

namespace qb3dev {

// Lane per index segment.  The stream words of the workgroup's segments (they are consecutive) are staged in LDS with
// coalesced 16-byte loads when they fit -- a lane refilling its bit buffer from global memory waits a memory round
// trip per word, 64 lanes on 64 different lines: the kernel spent 70 % of its wave cycles waiting -- and the lanes
// read LDS; a span that does not fit (seg_cap_dw is sized for half as much again as the stream's average) is read
// from global memory as before.
template <typename T, int MODE, typename RD>
__device__ __forceinline__ void dec_segment(const DecArgs &a, const DecArgs &a0, RD &rd, uint32_t *lane_mem, uint64_t seg, uint64_t pos_bias) {
    const uint32_t bands = a.g.bands, S = a.from_ix ? a.ix_blocks : a.g.seg_blocks, nbx = a.g.nbx;
    const uint64_t nseg = a.from_ix ? a.ix_K : a.g.nseg;
    T *blk = (T *)lane_mem;
    T *prev = blk + 16 * bands;
    T *pcf = prev + bands;
    uint8_t *rungs = (uint8_t *)(pcf + bands);
    if (a.from_ix) {            // the state entering the piece comes from the container's own table (layout: include/qb3x.h)
        const uint8_t *e = ix_entry_at(a.ix, a.ix_per_chunk, a.ix_E, a.ix_pad, (uint32_t)seg);
        const uint8_t *pv = e + 6 + bands, *cf = pv + bands * sizeof(T);
        for (uint32_t c = 0; c < bands; c++) {
            rungs[c] = e[6 + c];
            uint64_t v = 0, f = 0;
            for (uint32_t i = 0; i < sizeof(T); i++) { v |= (uint64_t)pv[c * sizeof(T) + i] << (8 * i); if (MODE == CM_BEST) f |= (uint64_t)cf[c * sizeof(T) + i] << (8 * i); }
            prev[c] = (T)v; pcf[c] = (T)f;
        }
    } else
        for (uint32_t c = 0; c < bands; c++) {
            prev[c] = a.totals_only ? (T)0 : ((const T *)a.idx.prev)[seg * bands + c];      // (totals_only: the segment's sums, not pixels)
            pcf[c] = (MODE == CM_BEST) ? ((const T *)a.idx.cf)[seg * bands + c] : (T)0;
            rungs[c] = a.idx.rung[seg * bands + c];
        }
    const uint64_t order = a.g.order;
    const uint32_t gend = (uint32_t)(((seg + 1) * S < a.g.nblocks) ? (seg + 1) * S : a.g.nblocks);
    bool ok = true;
    T g[16];
    for (uint32_t gb = (uint32_t)(seg * S); gb < gend && ok; gb++) {
        for (uint32_t c = 0; c < bands; c++) {
            uint32_t rung = rungs[c];
            T cf = pcf[c];
            ok = parse_unit<T, MODE, RD>(rd, rung, cf, g) && ok;
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
        if (a.totals_only) continue;        // (a plain stream, first pass: only what the segment's values add up to is wanted)
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
    if (a.totals_only) {        // leave the per-band sums where the entering values go (prev_scan_kernel turns them into those)
        for (uint32_t c = 0; c < bands; c++) ((T *)a.idx.prev)[seg * bands + c] = prev[c];
        return;
    }
    if (seg == nseg - 1) {
        // reference: fails when more than 7 bits are left (QB3decode.h:411,569,740); also flag overruns
        const uint64_t used = rd.position() + pos_bias - a.in_bit0;
        if (used > a.in_bits) atomicOr(a.status, 4u);
        else if (a.in_bits - used > 7) atomicOr(a.status, 2u);
    }
}

template <typename T, int MODE>
__global__ void dec_kernel(const DecArgs a0) {
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    const uint64_t seg0 = (uint64_t)blockIdx.x * nthr, seg = seg0 + tid;
    // per-lane LDS: scratch block [y][x][band] then band state; the staged stream words follow the lanes' areas
    uint32_t *lane_mem = (uint32_t *)smem + (size_t)tid * a.lane_dw;
    uint32_t *stage = (uint32_t *)smem + (((size_t)nthr * a.lane_dw + 3) & ~(size_t)3);
    // a "segment" is an index segment -- or, decoding straight from the container's restart table (from_ix), the piece
    // between two of its entries: no index is rebuilt, the lane decodes its piece from the entry's state
    const uint64_t nseg = a.from_ix ? a.ix_K : a.g.nseg;
    auto seg_pos = [&](uint64_t sidx) -> uint64_t {
        if (!a.from_ix) return a.idx.bitpos[sidx];
        const uint8_t *e = ix_entry_at(a.ix, a.ix_per_chunk, a.ix_E, a.ix_pad, (uint32_t)sidx);
        uint64_t v = 0;
        for (uint32_t i = 0; i < 6; i++) v |= (uint64_t)e[i] << (8 * i);
        return v;
    };
    const uint64_t segl = seg0 + nthr < nseg ? seg0 + nthr : nseg;                    // (workgroup uniform) one past the last segment
    const uint64_t P0 = seg_pos(seg0), P1 = segl < nseg ? seg_pos(segl) : a.in_bits;
    const uint64_t w0 = ((a.in_bit0 + P0) >> 5) & ~(uint64_t)3;
    const uint64_t endw_abs = (a.in_bit0 + a.in_bits + 31) >> 5;
    const uint64_t ndw64 = ((a.in_bit0 + P1 + 31) >> 5) - w0 + 2;                      // (+2: a refill may look one word ahead)
    const bool staged = a.seg_cap_dw && P1 >= P0 && ndw64 <= a.seg_cap_dw;
    if (staged) {
        const uint32_t ndw = (uint32_t)ndw64;
        const bool in16 = !((uintptr_t)a.in32 & 15);
        for (uint32_t base = 0; base < ndw; base += 16 * nthr) {            // four 16-byte loads in flight per lane
            uint4 sw[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t i = base + 4 * (tid + q * nthr);
                if (in16 && i + 4 <= ndw && w0 + i + 4 <= endw_abs) sw[q] = *(const uint4 *)(a.in32 + w0 + i);
                else {
                    sw[q].x = (i + 0 < ndw && w0 + i + 0 < endw_abs) ? a.in32[w0 + i + 0] : 0u;
                    sw[q].y = (i + 1 < ndw && w0 + i + 1 < endw_abs) ? a.in32[w0 + i + 1] : 0u;
                    sw[q].z = (i + 2 < ndw && w0 + i + 2 < endw_abs) ? a.in32[w0 + i + 2] : 0u;
                    sw[q].w = (i + 3 < ndw && w0 + i + 3 < endw_abs) ? a.in32[w0 + i + 3] : 0u;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++) { const uint32_t i = base + 4 * (tid + q * nthr); if (i < ndw) *(uint4 *)(stage + i) = sw[q]; }
        }
        __syncthreads();
    }
    if (seg >= nseg) return;
    if (staged) {
        ReaderT<LdsWords> rd;
        rd.init((LdsWords)stage, a.in_bit0 + seg_pos(seg) - 32 * w0, 32 * (uint64_t)(((uint32_t)ndw64 + 3) & ~3u));
        dec_segment<T, MODE, ReaderT<LdsWords>>(a, a0, rd, lane_mem, seg, 32 * w0);
    } else {
        Reader rd;
        rd.init(a.in32, a.in_bit0 + seg_pos(seg), a.in_bit0 + a.in_bits);
        dec_segment<T, MODE, Reader>(a, a0, rd, lane_mem, seg, 0);
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

// BL (32/64-bit data): no index -- position, entering rungs and values and a twelve-bit length per unit come from the segment's
// entry of the container's restart table (qb3x_set_encoder_index_chunk level 2)
template <typename T, bool STEP, bool BL = false>
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
    const uint8_t *ent = BL ? ix_entry_at(a.ix, a.ix_per_chunk, a.ix_E, a.ix_pad, (uint32_t)seg) : nullptr;
    auto pos6 = [](const uint8_t *q) { uint64_t v = 0;
        for (uint32_t i = 0; i < 6; i++) v |= (uint64_t)q[i] << (8 * i);
        return v; };
    const uint64_t P0 = BL ? pos6(ent) : a.idx.bitpos[seg];
    const uint64_t P1 = (seg + 1 < a.g.nseg) ? (BL ? pos6(ix_entry_at(a.ix, a.ix_per_chunk, a.ix_E, a.ix_pad, (uint32_t)seg + 1)) : a.idx.bitpos[seg + 1]) : a.in_bits;
    const uint64_t w0 = (a.in_bit0 + P0) >> 5;
    const uint64_t endw_abs = (a.in_bit0 + a.in_bits + 31) >> 5;
    const uint64_t ndw64 = ((a.in_bit0 + P1 + 31) >> 5) - w0 + 2;
    // (positions out of a container's table are not to be trusted: a segment that does not lie inside the stream reads nothing)
    const bool sane = !BL || (P0 <= P1 && P1 <= a.in_bits);
    const bool staged = sane && ndw64 <= a.in_cap_dw;  // workgroup uniform
    const uint32_t ndw = (uint32_t)ndw64;
    if (staged)
        for (uint32_t base = 0; base < ndw; base += 4 * nthr) {        // four loads in flight per thread, then four LDS stores
            uint32_t sw[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { const uint32_t i = base + tid + q * nthr; sw[q] = (i < ndw && w0 + i < endw_abs) ? a.in32[w0 + i] : 0u; }
#pragma unroll
            for (int q = 0; q < 4; q++) { const uint32_t i = base + tid + q * nthr; if (i < ndw) stage[i] = sw[q]; }
        }
    // words readable from w0 on (none when the segment is said to start behind the stream's end: a truncated stream, a table that lies)
    const uint32_t endw_g = (!sane || w0 >= endw_abs) ? 0u : (uint32_t)((endw_abs - w0 < 0xffffffffull) ? endw_abs - w0 : 0xffffffffull);
    for (uint32_t sl = tid; sl < nb_here; sl += nthr) {
        const uint32_t g = g0 + sl, by = g / nbx, bx = g - by * nbx;
        const uint32_t x0 = (4 * bx + 4 > a.g.w) ? a.g.w - 4 : 4 * bx;
        const uint32_t y0 = (4 * by + 4 > a.g.h) ? a.g.h - 4 : 4 * by;
        slot_base[sl] = (uint64_t)y0 * stride + (uint64_t)x0 * bands;
    }
    if (tid < bands) {
        if (BL) {
            const uint8_t *pv = ent + 6 + bands + tid * sizeof(T);
            uint64_t v = 0;
            for (uint32_t i = 0; i < sizeof(T); i++) v |= (uint64_t)pv[i] << (8 * i);
            cprev[tid] = v;
            crung[tid] = ent[6 + tid] & UMASK;
        } else {
            cprev[tid] = a.totals_only ? 0ull : (uint64_t)((const T *)a.idx.prev)[seg * bands + tid];       // (totals_only: the segment's sums, not pixels)
            crung[tid] = a.idx.rung[seg * bands + tid];
        }
    }
    uint32_t cpos = (uint32_t)(a.in_bit0 + P0 - 32 * w0);     // bit position of the pass, relative to word w0
    const uint32_t c = fastdiv(tid, BPP, a.magic_bpp), b = tid - c * BPP;
    const uint32_t cb = a0.g.cband[c < MAXBANDS ? c : 0];
    const uint64_t order = a.g.order;
    T *tt = (T *)tile;
    const uint32_t rowel = NB * 4 * bands;       // tile elements per pixel row
    bool bad = !sane;
    __syncthreads();

    for (uint32_t p = 0; p < a.passes; p++) {
        const uint32_t pb0 = p * BPP;
        const uint32_t nbp = pb0 >= nb_here ? 0 : ((nb_here - pb0 < BPP) ? nb_here - pb0 : BPP);
        // unit lengths of this pass (contiguous in the table)
        const uint64_t ubase = ((uint64_t)g0 + pb0) * bands;
        if (BL) {           // twelve-bit fields behind the entry's fixed part, unit by unit of the segment
            const uint8_t *fl = ent + 6 + bands * (1 + sizeof(T));
            for (uint32_t i = tid; i < nbp * bands; i += nthr) {
                const uint32_t f = pb0 * bands + i, bit = 12 * f;
                const uint8_t *q = fl + (bit >> 3);
                ulen_s[i] = (uint16_t)((((uint32_t)q[0] | (uint32_t)q[1] << 8) >> (bit & 7)) & 0xfffu);
            }
        } else if (a.g.ulen_sz == 1) for (uint32_t i = tid; i < nbp * bands; i += nthr) ulen_s[i] = ((const uint8_t *)a.idx.ulen)[ubase + i];
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
    if (a.totals_only) {        // a plain stream, first pass: leave the segment's per-band sums where the entering values go (prev_scan_kernel turns them into those)
        if (tid < bands) ((T *)a.idx.prev)[seg * bands + tid] = (T)cprev[tid];
        return;
    }
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

// Foreign stream: ONE lane walks the stream and rebuilds the index (bit position + band state at every
// segment start).  Latency bound by construction.
template <typename T, int MODE>
__global__ void __launch_bounds__(64) dec_index_serial(const DecArgs a0) {
    // without a restart table: lane 0 walks the whole tile; with one: a LANE per entry (blockIdx.y * 64 + lane)
    const DecArgs a = dec_for_tile(a0, blockIdx.x);
    __shared__ uint64_t s_prev[64][MAXBANDS + 1], s_cf[64][MAXBANDS + 1];      // (+1: lanes on different banks)
    __shared__ uint32_t s_rung[64][MAXBANDS + 1];
    const uint32_t bands = a.g.bands, S = a.g.seg_blocks, lane = threadIdx.x;
    const uint32_t nblocks = (uint32_t)a.g.nblocks;
    const uint32_t k = blockIdx.y * 64 + lane;
    if (a.ix ? k >= a.ix_K : lane != 0) return;
    uint64_t *st_prev = s_prev[lane], *st_cf = s_cf[lane];
    uint32_t *st_rung = s_rung[lane];
    for (uint32_t c = 0; c < bands; c++) { st_prev[c] = 0; st_cf[c] = 0; st_rung[c] = 0; }
    uint32_t gb0 = 0, gb_end = nblocks;
    uint64_t seg = 0, bp = 0;
    if (a.ix) {                             // restart point k of the container's table
        const uint8_t *e = ix_entry_at(a.ix, a.ix_per_chunk, a.ix_E, a.ix_pad, k);
        for (uint32_t i = 0; i < 6; i++) bp |= (uint64_t)e[i] << (8 * i);
        const uint8_t *pv = e + 6 + bands, *cf = pv + bands * sizeof(T);
        for (uint32_t c = 0; c < bands; c++) {
            st_rung[c] = e[6 + c];
            uint64_t v = 0, f = 0;
            for (uint32_t i = 0; i < sizeof(T); i++) { v |= (uint64_t)pv[c * sizeof(T) + i] << (8 * i); if (MODE == CM_BEST) f |= (uint64_t)cf[c * sizeof(T) + i] << (8 * i); }
            st_prev[c] = v; st_cf[c] = f;
        }
        gb0 = k * a.ix_blocks;
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
        uint32_t bt = 0;                     // block table entry (ulen_sz == 4): the block's bits | entering rungs << 16
        const uint64_t bstart = rd.position();
        for (uint32_t c = 0; c < bands; c++) {
            uint32_t rung = st_rung[c];
            T cf = (T)st_cf[c];
            const uint64_t ustart = rd.position();
            if (c < 4) bt |= (rung & (sizeof(T) >= 4 ? 63u : 15u)) << (16 + 4 * c);     // (32/64-bit data: one band, the whole rung)
            const uint32_t rung_in = rung;
            ok = parse_unit<T, MODE, Reader>(rd, rung, cf, g) && ok;
            if (a.g.ulen_sz == 1) ((uint8_t *)a.idx.ulen)[(uint64_t)gb * bands + c] = (uint8_t)(rd.position() - ustart);
            else if (a.g.ulen_sz == 2) ((uint16_t *)a.idx.ulen)[(uint64_t)gb * bands + c] = (uint16_t)(rd.position() - ustart);
            else if (a.g.ulen_sz == ULEN_UNIT) ((uint32_t *)a.idx.ulen)[(uint64_t)gb * bands + c] = (uint32_t)((rd.position() - ustart) & 0xffffu) | (rung_in << 16);
            T sum = 0;
#pragma unroll
            for (uint32_t i = 0; i < 16; i++) sum = (T)(sum + smag_t<T>(g[i]));
            st_prev[c] = (T)((T)st_prev[c] + sum);
            st_cf[c] = cf;
            st_rung[c] = rung;
        }
        if (a.g.ulen_sz == 4) ((uint32_t *)a.idx.ulen)[gb] = bt | (uint32_t)((rd.position() - bstart) & 0xffffu);
    }
    if (!ok) atomicOr(a.status, 1u);
}

// The same walk for a stream WITHOUT a table, one wave per tile: the wave stages a window of the stream in LDS with
// coalesced loads, lane 0 parses whole blocks out of it while the longest possible block still fits, the wave moves the
// window on.  (A lane refilling its bit buffer straight from global memory waits a memory round trip for every word.)
constexpr uint32_t SERIAL_WIN = 4096;       // dwords of stream per window
template <typename T, int MODE>
__global__ void __launch_bounds__(64) dec_index_staged(const DecArgs a0, uint32_t block_bits) {
    const DecArgs a = dec_for_tile(a0, blockIdx.x);
    __shared__ __attribute__((aligned(16))) uint32_t stage[SERIAL_WIN + 4];
    __shared__ uint64_t st_prev[MAXBANDS], st_cf[MAXBANDS], s_P;
    __shared__ uint32_t st_rung[MAXBANDS], s_gb, s_bad;
    const uint32_t bands = a.g.bands, S = a.g.seg_blocks, lane = threadIdx.x, nblocks = (uint32_t)a.g.nblocks;
    if (lane < bands) { st_prev[lane] = 0; st_cf[lane] = 0; st_rung[lane] = 0; }
    if (lane == 0) { s_P = a.in_bit0; s_gb = 0; s_bad = 0; }
    const uint64_t endw_abs = (a.in_bit0 + a.in_bits + 31) >> 5;
    __syncthreads();
    while (true) {
        const uint64_t P = s_P;
        const uint32_t gb0 = s_gb;
        if (gb0 >= nblocks || s_bad) break;
        const uint64_t w0 = (P >> 5) & ~(uint64_t)3;
        for (uint32_t i = lane; i < SERIAL_WIN + 4; i += 64) stage[i] = w0 + i < endw_abs ? a.in32[w0 + i] : 0u;
        __syncthreads();
        if (lane == 0) {
            ReaderT<LdsWords> rd;
            rd.init((LdsWords)stage, P - 32 * w0, 32ull * (SERIAL_WIN + 4));
            T g[16];
            bool ok = true;
            uint32_t gb = gb0;
            // a block is parsed only while the longest possible one still ends inside the window
            while (gb < nblocks && ok && rd.position() + block_bits + 64 <= 32ull * SERIAL_WIN) {
                if (gb % S == 0) {
                    const uint64_t seg = gb / S;
                    a.idx.bitpos[seg] = 32 * w0 + rd.position() - a.in_bit0;
                    for (uint32_t c = 0; c < bands; c++) {
                        ((T *)a.idx.prev)[seg * bands + c] = (T)st_prev[c];
                        if (MODE == CM_BEST) ((T *)a.idx.cf)[seg * bands + c] = (T)st_cf[c];
                        a.idx.rung[seg * bands + c] = (uint8_t)st_rung[c];
                    }
                }
                uint32_t bt = 0;
                const uint64_t bstart = rd.position();
                for (uint32_t c = 0; c < bands; c++) {
                    uint32_t rung = st_rung[c];
                    T cf = (T)st_cf[c];
                    const uint64_t ustart = rd.position();
                    if (c < 4) bt |= (rung & (sizeof(T) >= 4 ? 63u : 15u)) << (16 + 4 * c);     // (32/64-bit data: one band, the whole rung)
                    const uint32_t rung_in = rung;
                    ok = parse_unit<T, MODE, ReaderT<LdsWords>>(rd, rung, cf, g) && ok;
                    if (a.g.ulen_sz == 1) ((uint8_t *)a.idx.ulen)[(uint64_t)gb * bands + c] = (uint8_t)(rd.position() - ustart);
                    else if (a.g.ulen_sz == 2) ((uint16_t *)a.idx.ulen)[(uint64_t)gb * bands + c] = (uint16_t)(rd.position() - ustart);
                    else if (a.g.ulen_sz == ULEN_UNIT) ((uint32_t *)a.idx.ulen)[(uint64_t)gb * bands + c] = (uint32_t)((rd.position() - ustart) & 0xffffu) | (rung_in << 16);
                    T sum = 0;
#pragma unroll
                    for (uint32_t i = 0; i < 16; i++) sum = (T)(sum + smag_t<T>(g[i]));
                    st_prev[c] = (T)((T)st_prev[c] + sum);
                    st_cf[c] = cf;
                    st_rung[c] = rung;
                }
                if (a.g.ulen_sz == 4) ((uint32_t *)a.idx.ulen)[gb] = bt | (uint32_t)((rd.position() - bstart) & 0xffffu);
                gb++;
            }
            s_P = 32 * w0 + rd.position();
            if (gb == gb0 && ok) ok = false;        // (no block fits the window: cannot happen for a valid geometry; do not spin)
            s_gb = gb;
            if (!ok) s_bad = 1;
        }
        __syncthreads();
    }
    if (lane == 0 && s_bad) atomicOr(a.status, 1u);
}

// Plain COMMON-FACTOR streams of several bands (no table memory helps them: the factor in force is part of the walk's state): the
// same one-wave walk, but a unit WITHOUT the signal code -- nearly all of them -- is walked by LENGTH only (walk_unit, qb3_walk.h:
// a code's length is its rung plus what its two low bits say; an eighth of what a full parse costs one lane), a unit with it is
// parsed outright (its values decide the rung it leaves, QB3decode.h:619-716).  Leaves the segment entries (position, rungs,
// factors); the entering VALUES come from the parallel decoder: dec_kernel in totals mode, then prev_scan_kernel.
template <typename T>
__global__ void __launch_bounds__(64) dec_index_walk_best(const DecArgs a0, uint32_t block_bits) {
    const DecArgs a = dec_for_tile(a0, blockIdx.x);
    constexpr uint32_t UB = UBits<T>::v;
    __shared__ __attribute__((aligned(16))) uint32_t stage[SERIAL_WIN + 8];
    __shared__ uint64_t st_cf[MAXBANDS], s_P;
    __shared__ uint32_t st_rung[MAXBANDS], s_gb, s_bad;
    const uint32_t bands = a.g.bands, S = a.g.seg_blocks, lane = threadIdx.x, nblocks = (uint32_t)a.g.nblocks;
    const uint32_t stage_bit0 = 8 * (uint32_t)(uintptr_t)(LdsWords)stage;
    if (lane < bands) { st_cf[lane] = 0; st_rung[lane] = 0; }
    if (lane == 0) { s_P = a.in_bit0; s_gb = 0; s_bad = 0; }
    const uint64_t endw_abs = (a.in_bit0 + a.in_bits + 31) >> 5;
    __syncthreads();
    while (true) {
        const uint64_t P = s_P;
        const uint32_t gb0 = s_gb;
        if (gb0 >= nblocks || s_bad) break;
        const uint64_t w0 = (P >> 5) & ~(uint64_t)3;
        for (uint32_t i = lane; i < SERIAL_WIN + 8; i += 64) stage[i] = w0 + i < endw_abs ? a.in32[w0 + i] : 0u;
        __syncthreads();
        if (lane == 0) {
            uint32_t pos = (uint32_t)(P - 32 * w0);                 // bit position inside the window
            bool ok = true;
            uint32_t gb = gb0;
            T g[16];
            // a block is walked only while the longest possible one still ends inside the window
            while (gb < nblocks && ok && pos + block_bits + 64 <= 32u * SERIAL_WIN) {
                if (gb % S == 0) {
                    const uint64_t seg = gb / S;
                    a.idx.bitpos[seg] = 32 * w0 + pos - a.in_bit0;
                    for (uint32_t c = 0; c < bands; c++) {
                        ((T *)a.idx.cf)[seg * bands + c] = (T)st_cf[c];
                        a.idx.rung[seg * bands + c] = (uint8_t)st_rung[c];
                    }
                }
                const uint32_t b0 = pos;
                uint32_t bt = 0;                                    // (block-table shapes: the block's bits | its units' entering rungs)
                for (uint32_t c = 0; c < bands; c++) {
                    uint32_t rung = st_rung[c];
                    if (c < 4) bt |= (rung & (sizeof(T) >= 4 ? 63u : 15u)) << (16 + 4 * c);
                    const uint32_t u0 = pos;
                    bool sig = false;
                    const uint32_t len = walk_unit<UB>(stage_bit0 + pos, rung, sig);
                    if (!sig) pos += len;
                    else {                                          // common-factor or index form: the whole unit
                        ReaderT<LdsWords> rd;
                        rd.init((LdsWords)stage, pos, 32ull * (SERIAL_WIN + 8));
                        rung = st_rung[c];
                        T cf = (T)st_cf[c];
                        ok = parse_unit<T, CM_BEST, ReaderT<LdsWords>>(rd, rung, cf, g) && ok;
                        st_cf[c] = (uint64_t)cf;
                        pos = (uint32_t)rd.position();
                    }
                    if (a.g.ulen_sz == ULEN_UNIT) ((uint32_t *)a.idx.ulen)[(uint64_t)gb * bands + c] = ((pos - u0) & 0xffffu) | (st_rung[c] << 16);     // (the lane-per-unit decoder: bits | entering rung)
                    st_rung[c] = rung;
                }
                if (a.g.ulen_sz == 4) ((uint32_t *)a.idx.ulen)[gb] = bt | ((pos - b0) & 0xffffu);
                gb++;
            }
            s_P = 32 * w0 + pos;
            if (gb == gb0 && ok) ok = false;        // (no block fits the window: cannot happen for a valid geometry; do not spin)
            s_gb = gb;
            if (!ok) s_bad = 1;
        }
        __syncthreads();
    }
    if (lane == 0 && s_bad) atomicOr(a.status, 1u);
}
void launch_dec_index_walk_best(const DecArgs &a, hipStream_t st) {
    const uint32_t block_bits = a.g.bands * max_unit_bits(a.g.tsz, a.g.mode);
    const dim3 grid(a.ntiles), block(64);
    switch (a.g.tsz) {
    case 1: hipLaunchKernelGGL((dec_index_walk_best<uint8_t>), grid, block, 0, st, a, block_bits); break;
    case 2: hipLaunchKernelGGL((dec_index_walk_best<uint16_t>), grid, block, 0, st, a, block_bits); break;
    case 4: hipLaunchKernelGGL((dec_index_walk_best<uint32_t>), grid, block, 0, st, a, block_bits); break;
    default: hipLaunchKernelGGL((dec_index_walk_best<uint64_t>), grid, block, 0, st, a, block_bits); break;
    }
}
bool dec_index_walk_best_ok(const DecArgs &a) { return a.g.mode == CM_BEST && (a.g.ulen_sz == 0 || a.g.ulen_sz == 4 || a.g.ulen_sz == ULEN_UNIT) && a.g.bands * max_unit_bits(a.g.tsz, a.g.mode) + 64 + 64 <= 32 * SERIAL_WIN; }

template <typename T>
static void launch_dec_generic_t(const DecArgs &a, const DecPlan &plan, hipStream_t st) {
    if (plan.fast && a.g.mode != CM_BEST) {
        const dim3 grid((uint32_t)a.g.nseg, a.ntiles), block(plan.threads2);
        if (a.bl_mode && sizeof(T) >= 4) {
            if (a.g.mode == CM_BASE) hipLaunchKernelGGL((dec3_kernel<T, true, sizeof(T) >= 4>), grid, block, plan.lds2_bytes, st, a);
            else hipLaunchKernelGGL((dec3_kernel<T, false, sizeof(T) >= 4>), grid, block, plan.lds2_bytes, st, a);
            return;
        }
        if (a.g.mode == CM_BASE) hipLaunchKernelGGL((dec3_kernel<T, true>), grid, block, plan.lds2_bytes, st, a);
        else hipLaunchKernelGGL((dec3_kernel<T, false>), grid, block, plan.lds2_bytes, st, a);
        return;
    }
    const dim3 grid(a.from_ix ? (a.ix_K + plan.threads - 1) / plan.threads : plan.nwg, a.ntiles), block(plan.threads);
    const size_t lds = a.seg_cap_dw ? (((size_t)plan.threads * a.lane_dw + 3) & ~(size_t)3) * 4 + 4 * (size_t)a.seg_cap_dw : plan.lds_bytes;
    switch (a.g.mode) {
    case CM_FTL: hipLaunchKernelGGL((dec_kernel<T, CM_FTL>), grid, block, lds, st, a); break;
    case CM_BASE: hipLaunchKernelGGL((dec_kernel<T, CM_BASE>), grid, block, lds, st, a); break;
    default: hipLaunchKernelGGL((dec_kernel<T, CM_BEST>), grid, block, lds, st, a); break;
    }
}
void launch_dec_generic(const DecArgs &a, const DecPlan &plan, hipStream_t st) {
    switch (a.g.tsz) {
    case 1: launch_dec_generic_t<uint8_t>(a, plan, st); break;
    case 2: launch_dec_generic_t<uint16_t>(a, plan, st); break;
    case 4: launch_dec_generic_t<uint32_t>(a, plan, st); break;
    default: launch_dec_generic_t<uint64_t>(a, plan, st); break;
    }
}
template <typename T>
static void launch_dec_index_serial_t(const DecArgs &a, hipStream_t st) {
    if (!a.ix) {        // no table: a wave per tile, the stream staged through LDS
        const uint32_t block_bits = a.g.bands * max_unit_bits(a.g.tsz, a.g.mode);
        if (block_bits + 64 + 64 <= 32 * SERIAL_WIN) {
            const dim3 grid(a.ntiles), block(64);
            switch (a.g.mode) {
            case CM_FTL: hipLaunchKernelGGL((dec_index_staged<T, CM_FTL>), grid, block, 0, st, a, block_bits); break;
            case CM_BASE: hipLaunchKernelGGL((dec_index_staged<T, CM_BASE>), grid, block, 0, st, a, block_bits); break;
            default: hipLaunchKernelGGL((dec_index_staged<T, CM_BEST>), grid, block, 0, st, a, block_bits); break;
            }
            return;
        }
    }
    const dim3 grid(a.ntiles, a.ix ? (a.ix_K + 63) / 64 : 1), block(64);
    switch (a.g.mode) {
    case CM_FTL: hipLaunchKernelGGL((dec_index_serial<T, CM_FTL>), grid, block, 0, st, a); break;
    case CM_BASE: hipLaunchKernelGGL((dec_index_serial<T, CM_BASE>), grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL((dec_index_serial<T, CM_BEST>), grid, block, 0, st, a); break;
    }
}
void launch_dec_index_serial(const DecArgs &a, hipStream_t st) {
    switch (a.g.tsz) {
    case 1: launch_dec_index_serial_t<uint8_t>(a, st); break;
    case 2: launch_dec_index_serial_t<uint16_t>(a, st); break;
    case 4: launch_dec_index_serial_t<uint32_t>(a, st); break;
    default: launch_dec_index_serial_t<uint64_t>(a, st); break;
    }
}

}  // namespace qb3dev
