// qb3_amd/csrc/k_dec_px.hip -- 8-bit grey / RGB / RGBA decoder, wave per segment, lane per block
#include "qb3_px.h"

namespace qb3dev {

// BL: no index -- the container's restart table has an entry per segment that ends with the bit lengths of the segment's
// blocks (qb3x_set_encoder_index_chunk level 2): bit position, entering rungs and values come from the entry, a block's
// place from the scan of the lengths, and a unit's place from where the band before it ended (the bands of a block are
// decoded one after the other anyway).  No walk, no index: the decode is this one kernel.
template <int B, bool RGB, uint64_t ORDER, bool STEP, bool BL>
__global__ void __launch_bounds__(256) dec_px_kernel(const DecArgs a0) {
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t chk = BL ? a0.chk_wgs : 0u;
    if (blockIdx.x < chk) {                             // the launch's first workgroups: a chunk of the container's table each (beside the decoding, not behind it)
        ix_check_chunk(a, blockIdx.x, (uint32_t *)smem);
        return;
    }
    constexpr int NW = (B + 1) / 2;                     // 32-bit words of a scan packed 16 bits per band
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const uint32_t NB = a.g.seg_blocks, nbx = a.g.nbx;  // NB <= 64: a WAVE owns a segment, nothing is shared but the table
    const uint64_t stride = a.g.stride;

    uint32_t *tab = (uint32_t *)smem;                   // 4 KB, at LDS address 0 (the table addressing relies on it)
    uint32_t *stage = tab + 1024 + wave * (a.in_cap_dw + 8);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)smem;
    const uint32_t stage_bit0 = 8 * (lds0 + (uint32_t)((uint8_t *)stage - smem));
    // Loads that depend on nothing but the segment number go out first -- positions, unit lengths, entering rungs and
    // values -- so that their round trips overlap the table copy and its barrier (a wave spends 45 % of its life waiting
    // for memory before it can start: the two dependent trips "position, then stream words").
    const uint64_t seg = a.seg0 + (uint64_t)(blockIdx.x - chk) * nwaves + wave;       // (seg0, seg_end: this launch's range of segments)
    const bool live = seg < a.seg_end;
    const uint64_t segc = live ? seg : 0;
    const uint32_t g0 = (uint32_t)(segc * NB), nblocks = (uint32_t)a.g.nblocks;
    const uint32_t nb_here = (nblocks - g0 < NB) ? nblocks - g0 : NB;
    const bool act = live && lane < nb_here;
    uint64_t P0, P1;
    uint32_t ul_[B], rg0[B], pv0[B], blen = 0;
    if (BL) {
        const uint8_t *e = ix_entry_at(a.ix, a.ix_per_chunk, a.ix_E, a.ix_pad, (uint32_t)segc);
        auto pos6 = [](const uint8_t *q) { uint64_t v = 0;
#pragma unroll
            for (uint32_t i = 0; i < 6; i++) v |= (uint64_t)q[i] << (8 * i);
            return v; };
        P0 = pos6(e);
        P1 = (segc + 1 < a.g.nseg) ? pos6(ix_entry_at(a.ix, a.ix_per_chunk, a.ix_E, a.ix_pad, (uint32_t)segc + 1)) : a.in_bits;
#pragma unroll
        for (int c = 0; c < B; c++) { ul_[c] = 0; rg0[c] = e[6 + c] & 7u; pv0[c] = e[6 + B + c]; }
        const uint8_t *bl = e + 6 + 2 * B + ((IX_BL_BITS * lane) >> 3);
        blen = act ? (((uint32_t)bl[0] | (uint32_t)bl[1] << 8) >> ((IX_BL_BITS * lane) & 7)) & ((1u << IX_BL_BITS) - 1) : 0u;
    } else {
        P0 = a.idx.bitpos[segc];
        P1 = (segc + 1 < a.g.nseg) ? a.idx.bitpos[segc + 1] : a.in_bits;
        if (P1 < P0) P1 = P0;       // (the last segment of a truncated stream starts behind its end: it reads zeros, like the reference's reader, bitstream.h:36)
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
    if (!BL)
#pragma unroll
        for (int c = 0; c < B; c++) blen += ul_[c];
    // the wave reads what its own lanes staged: LDS operations of a wave execute in order, the fence is for the compiler
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const uint32_t limit = stage_bit0 + 32 * ndw;       // no unit starts beyond the staged bits (8 zero words follow)
    const uint32_t cpos = stage_bit0 + (uint32_t)(a.in_bit0 + P0 - 32 * w0);
    bool bad = !fits;
    const uint32_t binc = wave_iscan32(blen);           // inclusive: lane 63 holds the bits of the segment
    // rung switches of the lane's units
    uint32_t gpos[B], pos = cpos + binc - blen, dpk[NW];
    uint32_t rp[B][8], spk[NW], sinc[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) { dpk[k] = 0; spk[k] = 0; }
    if (BL) {
        // band after band: the switch, the band's rungs across the segment (a scan of the switches), the unit, and
        // where it ended is where the next band's unit starts
        const uint32_t blk_end = pos + blen;
#pragma unroll
        for (int c = 0; c < B; c++) {
            pos = pos < limit ? pos : limit;
            bool sig; uint32_t csl;
            const uint32_t d = px_switch(pos, &csl, &sig);
            if (act && sig && STEP) bad = true;         // common-factor / index unit: not handled here
            const uint32_t rung = (rg0[c] + wave_iscan32(act ? d : 0u)) & 7u;
            uint32_t end;
            const uint32_t tot = px_group<STEP>(pos + csl, rung, rp[c], &end) & 0xffu;
            spk[c >> 1] |= (act ? tot : 0u) << (16 * (c & 1));
            pos = end;
        }
        if (act && pos != blk_end) bad = true;          // the table's lengths are not this stream's
    } else {
#pragma unroll
        for (int c = 0; c < B; c++) {
            pos = pos < limit ? pos : limit;
            bool sig; uint32_t csl;
            const uint32_t d = px_switch(pos, &csl, &sig);
            gpos[c] = pos + csl;
            if (act && sig && STEP) bad = true;         // common-factor / index unit: not handled here
            dpk[c >> 1] |= (act ? d : 0u) << (16 * (c & 1));
            pos += ul_[c];
        }
#pragma unroll
        for (int k = 0; k < NW; k++) dpk[k] = wave_iscan32(dpk[k]);             // inclusive, 16 bits per band
        // decode the units; running sums in curve order, two 16-bit lanes per register
#pragma unroll
        for (int c = 0; c < B; c++) {
            const uint32_t rung = (rg0[c] + ((dpk[c >> 1] >> (16 * (c & 1))) & 0xffffu)) & 7u;
            const uint32_t tot = px_group<STEP>(gpos[c], rung, rp[c]) & 0xffu;
            spk[c >> 1] |= (act ? tot : 0u) << (16 * (c & 1));
        }
    }
#pragma unroll
    for (int k = 0; k < NW; k++) sinc[k] = wave_iscan32(spk[k]);
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
    if (bad) atomicOr(a.status, fits ? 1u : 8u);
    if (lane == 63 && seg == a.g.nseg - 1 && fits) {    // reference: more than 7 unused bits at the end is a failure
        const uint64_t used = (uint64_t)(cpos + binc - stage_bit0) + 32 * w0 - a.in_bit0;
        if (used > a.in_bits) atomicOr(a.status, 4u);
        else if (a.in_bits - used > 7) atomicOr(a.status, 2u);
    }
}

template <int B, bool RGB>
static void launch_dec_px_b(const DecArgs &a, const DecPlan &plan, hipStream_t st) {
    const bool step = a.g.mode != CM_FTL, z = a.g.order == ZCURVE;
    dim3 grid((uint32_t)((a.seg_end - a.seg0 + 3) / 4), a.ntiles), block(256);
    if (a.bl_mode) {
        grid.x += a.chk_wgs;                            // (workgroups that check the container's table instead of decoding a segment)
        if (!z && !step) hipLaunchKernelGGL((dec_px_kernel<B, RGB, HILBERT, false, true>), grid, block, plan.lds_px, st, a);
        else if (!z && step) hipLaunchKernelGGL((dec_px_kernel<B, RGB, HILBERT, true, true>), grid, block, plan.lds_px, st, a);
        else if (z && !step) hipLaunchKernelGGL((dec_px_kernel<B, RGB, ZCURVE, false, true>), grid, block, plan.lds_px, st, a);
        else hipLaunchKernelGGL((dec_px_kernel<B, RGB, ZCURVE, true, true>), grid, block, plan.lds_px, st, a);
        return;
    }
    if (!z && !step) hipLaunchKernelGGL((dec_px_kernel<B, RGB, HILBERT, false, false>), grid, block, plan.lds_px, st, a);
    else if (!z && step) hipLaunchKernelGGL((dec_px_kernel<B, RGB, HILBERT, true, false>), grid, block, plan.lds_px, st, a);
    else if (z && !step) hipLaunchKernelGGL((dec_px_kernel<B, RGB, ZCURVE, false, false>), grid, block, plan.lds_px, st, a);
    else hipLaunchKernelGGL((dec_px_kernel<B, RGB, ZCURVE, true, false>), grid, block, plan.lds_px, st, a);
}
void launch_dec_px(const DecArgs &a, const DecPlan &plan, hipStream_t st) {
    if (a.g.bands == 1) launch_dec_px_b<1, false>(a, plan, st);
    else if (a.g.bands == 3) { if (plan.px_rgb) launch_dec_px_b<3, true>(a, plan, st); else launch_dec_px_b<3, false>(a, plan, st); }
    else { if (plan.px_rgb) launch_dec_px_b<4, true>(a, plan, st); else launch_dec_px_b<4, false>(a, plan, st); }
}

}  // namespace qb3dev
