// qb3_amd/csrc/k_dec_px16.hip -- 16-bit decoder, wave per segment, lane per (block, band group)
#include "qb3_px.h"

namespace qb3dev {

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

// The units of a lane's BG bands whose rung is 8 or more, by the code rule (values do not fit the 8-bit tables): 16
// values each from bit gpos[c]; rp[c][k] = running sums of values 2k, 2k+1 (16-bit halves); tot[c] = the unit's total.
// The BG walks are independent chains of data-dependent LDS reads and shifts, so they advance in LOCKSTEP, code by
// code: several reads in flight instead of one (the kernel is bound by that latency, not by issue: SQ_INSTS_VALU x 2 /
// SIMD = 28 % of its duration when the bands were walked one after the other).  A band whose rung is below 8 walks
// along with a harmless result (its reads stay inside the staged words and their zero margin); the caller overwrites it.
// RS: the N bands are RS apart in the caller's arrays; endp (when given): the bit behind each unit.
template <bool STEP, int N, int RS = 1>         // N bands at a time: two is what the registers hold without spilling
__device__ __forceinline__ void px16_groups_hi(const uint32_t *gpos, const uint32_t *rung, uint32_t (*rp)[8], uint32_t *tot, uint32_t *endp = nullptr) {
    constexpr int BG = N;
    uint32_t pos[BG], acc[BG], fl[BG], top[BG], half[BG];
    uint64_t buf[BG];
#pragma unroll
    for (int c = 0; c < BG; c++) { pos[c] = gpos[c * RS]; acc[c] = 0; fl[c] = 0; top[c] = 1u << rung[c * RS]; half[c] = top[c] >> 1; buf[c] = 0; }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (i % 3 == 0) {                               // three codes are at most 51 bits
#pragma unroll
            for (int c = 0; c < BG; c++) {
                LdsWords p = lds_at((pos[c] >> 3) & ~3u);
                const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
                buf[c] = ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, pos[c]) << 32) | __builtin_amdgcn_alignbit(d1, d0, pos[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < BG; c++) {
            const uint32_t x = (uint32_t)buf[c];
            const bool c1 = x & 1, c2 = (x & 3) == 3;
            const uint32_t len = rung[c * RS] + c1 + c2;
            const uint32_t v = c2 ? (((x >> 2) & (top[c] - 1)) | top[c]) : c1 ? (((x >> 2) & (half[c] - 1)) | half[c]) : ((x & (top[c] - 1)) >> 1);
            buf[c] >>= len; pos[c] += len;
            acc[c] += (v >> 1) ^ (0u - (v & 1u));       // undo mag-sign, accumulate (mod 2^16 in the packed halves)
            if (STEP) fl[c] |= ((uint32_t)c2 | ((v & 1u) << 1)) << (2 * i);
            if (i & 1) rp[c * RS][i >> 1] |= acc[c] << 16; else rp[c * RS][i >> 1] = acc[c] & 0xffffu;
        }
    }
#pragma unroll
    for (int c = 0; c < BG; c++) {
        if (endp) endp[c * RS] = pos[c];
        if (STEP) {                                     // undo the step (reference QB3decode.h:285-289), as in px_group
            const uint32_t tb = fl[c] & 0x55555555u, u = tb | (tb << 1);
            const uint32_t m = __popc(tb);
            if ((u & (u + 1)) == 0 && m < 16) {
                const uint32_t c16 = ((fl[c] >> (2 * m + 1)) & 1u) ? (0u - half[c]) & 0xffffu : half[c];
                const uint32_t ge = 0xffff0000u >> (16 - m);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const uint32_t pair = (ge >> (2 * k)) & 3u;
                    rp[c * RS][k] = pk_add16(rp[c * RS][k], ((pair | (pair << 15)) & 0x00010001u) * c16);
                }
                acc[c] += c16;
            }
        }
        tot[c * RS] = acc[c];
    }
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

// BL (four bands a lane): no index -- the container's table has an entry per segment that ends with two band-pair lengths per
// lane (qb3x_set_encoder_index_chunk level 2).  Bands 0 and 2 of a lane start where the pair lengths say and walk in
// lockstep; bands 1 and 3 start where those ended.  No walk, no index.
template <int BG, bool RGB, uint64_t ORDER, bool STEP, bool BL = false>
__global__ void __launch_bounds__(256, 4) dec_px16_kernel(const DecArgs a0) {
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t chk = BL ? a0.chk_wgs : 0u;          // the launch's first workgroups check a chunk of the container's table each (ix_check_chunk)
    if (blockIdx.x < chk) { ix_check_chunk(a, blockIdx.x, (uint32_t *)smem); return; }
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
    const uint64_t seg = a.seg0 + (uint64_t)(blockIdx.x - chk) * nwaves + wave;       // (seg0, seg_end: this launch's range of segments)
    const bool live = seg < a.seg_end;
    const uint64_t segc = live ? seg : 0;
    const uint32_t g0 = (uint32_t)(segc * NB), nblocks = (uint32_t)a.g.nblocks;
    const uint32_t nb_here = (nblocks - g0 < NB) ? nblocks - g0 : NB;
    const bool act = live && slot < nb_here;
    uint64_t P0, P1;
    uint32_t ul_[BG], rg0[BG], pv0[BG], blen = 0, f0 = 0, f1 = 0;
    if (BL) {
        const uint8_t *e = ix_entry_at(a.ix, a.ix_per_chunk, a.ix_E, a.ix_pad, (uint32_t)segc);
        auto pos6 = [](const uint8_t *q) { uint64_t v = 0;
#pragma unroll
            for (uint32_t i = 0; i < 6; i++) v |= (uint64_t)q[i] << (8 * i);
            return v; };
        P0 = pos6(e);
        P1 = (segc + 1 < a.g.nseg) ? pos6(ix_entry_at(a.ix, a.ix_per_chunk, a.ix_E, a.ix_pad, (uint32_t)segc + 1)) : a.in_bits;
#pragma unroll
        for (int c = 0; c < BG; c++) {
            ul_[c] = 0;
            rg0[c] = e[6 + band0 + c] & 15u;
            const uint8_t *pv = e + 6 + B + 2 * (band0 + c);
            pv0[c] = (uint32_t)pv[0] | (uint32_t)pv[1] << 8;
        }
        constexpr uint32_t FPL = BG == 1 ? 1 : 2;                   // fields a lane: two (band pairs; pair and band; two bands), or the one unit of a single band
        const uint32_t bit = FPL * IX_BL_BITS * lane;
        const uint8_t *fp = e + 6 + 3 * B + (bit >> 3);
        const uint32_t v = ((uint32_t)fp[0] | (uint32_t)fp[1] << 8 | (FPL == 2 ? (uint32_t)fp[2] << 16 : 0u)) >> (bit & 7);
        f0 = act ? v & ((1u << IX_BL_BITS) - 1) : 0u;
        f1 = act && FPL == 2 ? (v >> IX_BL_BITS) & ((1u << IX_BL_BITS) - 1) : 0u;
        blen = f0 + f1;
    } else {
        P0 = a.idx.bitpos[segc];
        P1 = (segc + 1 < a.g.nseg) ? a.idx.bitpos[segc + 1] : a.in_bits;
        if (P1 < P0) P1 = P0;       // (the last segment of a truncated stream starts behind its end: it reads zeros, like the reference's reader, bitstream.h:36)
        // a lane's BG lengths, rungs and entering values are contiguous: one load each where the address allows
        const uint16_t *ul = (const uint16_t *)a.idx.ulen + ((uint64_t)g0 + slot) * B + band0;
        const uint16_t *pvp = (const uint16_t *)a.idx.prev + segc * B + band0;
        const uint8_t *rgp = a.idx.rung + segc * B + band0;
        if (BG == 4 && !(((uintptr_t)ul | (uintptr_t)pvp) & 7) && !((uintptr_t)rgp & 3)) {
            const uint2 u = act ? *(const uint2 *)ul : make_uint2(0, 0), v = *(const uint2 *)pvp;
            const uint32_t r = *(const uint32_t *)rgp;
            ul_[0] = u.x & 0xffffu; ul_[1 % BG] = u.x >> 16; ul_[2 % BG] = u.y & 0xffffu; ul_[3 % BG] = u.y >> 16;
            pv0[0] = v.x & 0xffffu; pv0[1 % BG] = v.x >> 16; pv0[2 % BG] = v.y & 0xffffu; pv0[3 % BG] = v.y >> 16;
#pragma unroll
            for (int c = 0; c < BG; c++) rg0[c] = (r >> (8 * c)) & 0xffu;
        } else {
#pragma unroll
            for (int c = 0; c < BG; c++) {
                ul_[c] = act ? ul[c] : 0u;
                rg0[c] = rgp[c];
                pv0[c] = pvp[c];
            }
        }
    }
    for (uint32_t i = tid; i < 256; i += blockDim.x) ((uint4 *)tab)[i] = ((const uint4 *)px_dec_tab.e)[i];
    __syncthreads();                                    // the only workgroup barrier
    if (!live) return;
    // the segment's words, from the 16-byte aligned word at or before its first: sixteen bytes a load, four loads in
    // flight per lane (1024 words: a typical segment in ONE round trip), then the LDS stores; 16 zero words follow
    const uint64_t w0 = ((a.in_bit0 + P0) >> 5) & ~(uint64_t)3;
    const uint64_t endw_abs = (a.in_bit0 + a.in_bits + 31) >> 5;
    const uint64_t ndw64 = ((a.in_bit0 + P1 + 31) >> 5) - w0;
    const bool fits = ndw64 <= a.in_cap_dw && lds0 == 0;
    const uint32_t ndw = fits ? (uint32_t)ndw64 : 0;
    const bool in16 = !((uintptr_t)a.in32 & 15);         // (the stream's base decides whether whole 16-byte loads are aligned)
    for (uint32_t base = 0; base < ndw + 16; base += 1024) {
        uint4 sw[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t i = base + 4 * (lane + 64 * q);
            if (in16 && i + 4 <= ndw && w0 + i + 4 <= endw_abs) sw[q] = *(const uint4 *)(a.in32 + w0 + i);
            else {
                sw[q].x = (i + 0 < ndw && w0 + i + 0 < endw_abs) ? a.in32[w0 + i + 0] : 0u;
                sw[q].y = (i + 1 < ndw && w0 + i + 1 < endw_abs) ? a.in32[w0 + i + 1] : 0u;
                sw[q].z = (i + 2 < ndw && w0 + i + 2 < endw_abs) ? a.in32[w0 + i + 2] : 0u;
                sw[q].w = (i + 3 < ndw && w0 + i + 3 < endw_abs) ? a.in32[w0 + i + 3] : 0u;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t i = base + 4 * (lane + 64 * q);
            if (i < ndw + 16) *(uint4 *)(stage + i) = sw[q];
        }
    }
    if (!BL)
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
    uint32_t rp[BG][8], spk[NW], sinc[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) { dpk[k] = 0; spk[k] = 0; }
    if (BL && BG == 1) {
        // a single band: the lane's unit starts where the scan of the lengths says
        uint32_t rungs[1], tots[1] = { 0 }, ends[1] = { 0 };
        const uint32_t lane0 = pos;
        pos = pos < limit ? pos : limit;
        bool sig; uint32_t csl;
        const uint32_t d = px16_switch(pos, &csl, &sig);
        gpos[0] = pos + csl;
        if (act && sig && STEP) bad = true;             // common-factor / index unit: not handled here
        uint32_t dsc[1] = { act ? d : 0u };
        group_iscan<1>(dsc, NG);
        rungs[0] = (rg0[0] + dsc[0]) & 15u;
        if (__any(rungs[0] >= 8)) px16_groups_hi<STEP, 1>(&gpos[0], &rungs[0], &rp[0], &tots[0], &ends[0]);
        if (rungs[0] < 8) tots[0] = px_group<STEP>(gpos[0], rungs[0], rp[0], &ends[0]);
        if (act && ends[0] != lane0 + f0) bad = true;   // the table's length is not this unit's
        spk[0] |= act ? tots[0] & 0xffffu : 0u;
    } else if (BL && BG >= 2) {
        // The lane's fields give two starts.  Four bands: the pairs (0,1) and (2,3) -- bands 0 and 2 walk in lockstep, then bands 1
        // and 3 from where those ended.  Three: (0,1) and band 2 -- bands 0 and 2, then band 1.  Two: a field per band, one round.
        uint32_t rungs[BG], tots[BG], ends[BG];
#pragma unroll
        for (int c = 0; c < BG; c++) { rungs[c] = 0; tots[c] = 0; ends[c] = 0; }
        const uint32_t lane0 = pos;
        auto round = [&](auto nc, auto rsc, auto c0c, uint32_t s0, uint32_t s1) {
            constexpr int N = decltype(nc)::value, RS = decltype(rsc)::value, C0 = decltype(c0c)::value;
            const uint32_t st[2] = { s0, s1 };
            uint32_t dd = 0;
#pragma unroll
            for (int h = 0; h < N; h++) {
                const int c = C0 + h * RS;
                uint32_t p0 = st[h];
                p0 = p0 < limit ? p0 : limit;
                bool sig; uint32_t csl;
                const uint32_t d = px16_switch(p0, &csl, &sig);
                gpos[c] = p0 + csl;
                if (act && sig && STEP) bad = true;     // common-factor / index unit: not handled here
                dd |= (act ? d : 0u) << (16 * h);
            }
            uint32_t dsc[1] = { dd };
            group_iscan<1>(dsc, NG);                    // inclusive, the bands' rung changes 16 bits each
            bool lane_hi = false, lane_lo = false;
#pragma unroll
            for (int h = 0; h < N; h++) {
                const int c = C0 + h * RS;
                rungs[c] = (rg0[c] + ((dsc[0] >> (16 * h)) & 0xffffu)) & 15u;
                lane_hi = lane_hi || rungs[c] >= 8; lane_lo = lane_lo || rungs[c] < 8;
            }
            if (__any(lane_hi)) px16_groups_hi<STEP, N, RS>(&gpos[C0], &rungs[C0], &rp[C0], &tots[C0], &ends[C0]);
            if (__any(lane_lo)) {
#pragma unroll
                for (int h = 0; h < N; h++) {
                    const int c = C0 + h * RS;
                    if (rungs[c] < 8) tots[c] = px_group<STEP>(gpos[c], rungs[c], rp[c], &ends[c]);
                }
            }
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        bool ok;
        if constexpr (BG == 4) {
            round(I2(), I2(), I0(), pos, pos + f0);
            round(I2(), I2(), I1(), ends[0], ends[2]);
            ok = ends[1] == lane0 + f0 && ends[3] == lane0 + f0 + f1;
        } else if constexpr (BG == 3) {
            round(I2(), I2(), I0(), pos, pos + f0);
            round(I1(), I1(), I1(), ends[0], 0u);
            ok = ends[1] == lane0 + f0 && ends[2] == lane0 + f0 + f1;
        } else {
            round(I2(), I1(), I0(), pos, pos + f0);
            ok = ends[0] == lane0 + f0 && ends[1] == lane0 + f0 + f1;
        }
        if (act && !ok) bad = true;                     // the table's lengths are not this stream's
#pragma unroll
        for (int c = 0; c < BG; c++) spk[c >> 1] |= (act ? tots[c] & 0xffffu : 0u) << (16 * (c & 1));
    } else {
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
        {
            uint32_t rungs[BG], tots[BG];
            bool lane_hi = false, lane_lo = false;
#pragma unroll
            for (int c = 0; c < BG; c++) {
                rungs[c] = (rg0[c] + ((dpk[c >> 1] >> (16 * (c & 1))) & 0xffffu)) & 15u;
                lane_hi = lane_hi || rungs[c] >= 8; lane_lo = lane_lo || rungs[c] < 8;
                tots[c] = 0;
            }
            if (__any(lane_hi)) {                           // the lane's bands in lockstep, two at a time
                if (BG >= 2) px16_groups_hi<STEP, 2>(&gpos[0], &rungs[0], &rp[0], &tots[0]);
                if (BG == 4) px16_groups_hi<STEP, 2>(&gpos[2], &rungs[2], &rp[2], &tots[2]);
                if (BG & 1) px16_groups_hi<STEP, 1>(&gpos[BG - 1], &rungs[BG - 1], &rp[BG - 1], &tots[BG - 1]);
            }
            if (__any(lane_lo)) {                           // values below 256: the table path of the 8-bit kernel
#pragma unroll
                for (int c = 0; c < BG; c++) if (rungs[c] < 8) tots[c] = px_group<STEP>(gpos[c], rungs[c], rp[c]);
            }
#pragma unroll
            for (int c = 0; c < BG; c++) spk[c >> 1] |= (act ? tots[c] & 0xffffu : 0u) << (16 * (c & 1));
        }
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
        if (bad) atomicOr(a.status, fits ? 1u : (lds0 == 0 && ndw64 <= a.in_cap_full ? 16u : 8u));     // 16: staging sized for this stream's average was too small -- the host runs the call again with the worst case
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
        // (wave uniform) eight bands, pixels on 16-byte addresses
        const bool pair16 = BG == 4 && NG == 2 && !((uintptr_t)a.img & 15) && !((stride * 2) & 15);
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
            if (BG == 4 && pair16) {
                // eight bands: the two lanes of a block hold half a pixel each (8 bytes at a 16-byte stride: every store would
                // touch 32 lines for 16 bytes each).  They swap halves -- the even lane takes pixels 0 and 1 whole, the odd
                // lane pixels 2 and 3 -- and store 32 contiguous bytes each.
                uint32_t rcv[4];
#pragma unroll
                for (int j = 0; j < 4; j++) rcv[j] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(grp ? ow[j] : ow[4 + j]), 0xb1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
                const uint4 pa = grp ? make_uint4(rcv[0], rcv[1], ow[4], ow[5]) : make_uint4(ow[0], ow[1], rcv[0], rcv[1]);
                const uint4 pb = grp ? make_uint4(rcv[2], rcv[3], ow[6], ow[7]) : make_uint4(ow[2], ow[3], rcv[2], rcv[3]);
                uint4 *dst = (uint4 *)(rowp - band0 + (grp ? 2 * B : 0));
                dst[0] = pa; dst[1] = pb;
            } else if (BG % 2 == 0) {
#pragma unroll
                for (int x = 0; x < 4; x++) store_dw(rowp + (uint64_t)x * B, &ow[x * (BG / 2)], std::integral_constant<int, (BG / 2 ? BG / 2 : 1)>());
            } else
                store_dw(rowp, &ow[0], std::integral_constant<int, 2 * BG>());
        }
    }
    if (bad) atomicOr(a.status, fits ? 1u : (lds0 == 0 && ndw64 <= a.in_cap_full ? 16u : 8u));     // 16: staging sized for this stream's average was too small -- the host runs the call again with the worst case
    if (lane == 63 && seg == a.g.nseg - 1 && fits) {
        const uint64_t used = (uint64_t)(cpos + binc - stage_bit0) + 32 * w0 - a.in_bit0;
        if (used > a.in_bits) atomicOr(a.status, 4u);
        else if (a.in_bits - used > 7) atomicOr(a.status, 2u);
    }
}

template <int BG, bool RGB>
static void launch_dec_px16_b(const DecArgs &a, const DecPlan &plan, hipStream_t st) {
    const bool step = a.g.mode != CM_FTL, z = a.g.order == ZCURVE;
    dim3 grid((uint32_t)((a.seg_end - a.seg0 + 3) / 4) + (a.bl_mode ? a.chk_wgs : 0u), a.ntiles), block(256);
    if (a.bl_mode) {
        constexpr bool bl = true;
        if (!z && !step) hipLaunchKernelGGL((dec_px16_kernel<BG, RGB, HILBERT, false, bl>), grid, block, plan.lds_px, st, a);
        else if (!z && step) hipLaunchKernelGGL((dec_px16_kernel<BG, RGB, HILBERT, true, bl>), grid, block, plan.lds_px, st, a);
        else if (z && !step) hipLaunchKernelGGL((dec_px16_kernel<BG, RGB, ZCURVE, false, bl>), grid, block, plan.lds_px, st, a);
        else hipLaunchKernelGGL((dec_px16_kernel<BG, RGB, ZCURVE, true, bl>), grid, block, plan.lds_px, st, a);
        return;
    }
    if (!z && !step) hipLaunchKernelGGL((dec_px16_kernel<BG, RGB, HILBERT, false>), grid, block, plan.lds_px, st, a);
    else if (!z && step) hipLaunchKernelGGL((dec_px16_kernel<BG, RGB, HILBERT, true>), grid, block, plan.lds_px, st, a);
    else if (z && !step) hipLaunchKernelGGL((dec_px16_kernel<BG, RGB, ZCURVE, false>), grid, block, plan.lds_px, st, a);
    else hipLaunchKernelGGL((dec_px16_kernel<BG, RGB, ZCURVE, true>), grid, block, plan.lds_px, st, a);
}
void launch_dec_px16(const DecArgs &a, const DecPlan &plan, hipStream_t st) {
    switch (plan.px16_bg) {
    case 1: launch_dec_px16_b<1, false>(a, plan, st); break;
    case 2: launch_dec_px16_b<2, false>(a, plan, st); break;
    case 3: if (plan.px_rgb) launch_dec_px16_b<3, true>(a, plan, st); else launch_dec_px16_b<3, false>(a, plan, st); break;
    default: if (plan.px_rgb) launch_dec_px16_b<4, true>(a, plan, st); else launch_dec_px16_b<4, false>(a, plan, st); break;
    }
}

}  // namespace qb3dev
