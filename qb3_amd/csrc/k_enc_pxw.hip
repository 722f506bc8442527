// qb3_amd/csrc/k_enc_pxw.hip -- 32/64-bit single-band encoder (FTL / BASE), lane per BLOCK, the block in registers
//
// The counterpart of enc_px_kernel (k_enc_px.hip) for wide types -- elevation and count rasters: int32 / int64 / their
// unsigned twins, one band (BASELINE configs[3]).  Same bit stream as the reference's encode_fast<T> (QB3encode.h:376-451)
// and groupencode<T> (:155-280), organised for the GPU:
//   * a lane owns a block: its four rows are four 16-byte loads (two per row for 64-bit data) straight from HBM -- 64 lanes
//     read one contiguous kilobyte per row, no LDS tile -- plus the one value the curve left the previous block with;
//   * the curve is a template parameter, so "gather in curve order" is register renaming; delta, mag-sign, rung in registers;
//   * the rung of the block before comes from the neighbouring lane (DPP wave shift; LDS only across waves); lane 0 of the
//     workgroup is the halo block (rung only), so a chunk is one block less than the workgroup has lanes;
//   * a unit below rung 8 (smooth data: its deltas fit a byte) is coded the way the 8-bit kernel codes: the sixteen values packed
//     four to a register, six pieces of at most 27 bits, each built backwards with one v_alignbit per value out of the 2 KB code
//     table in LDS (px_unit_pieces, qb3_px_enc.h); from rung 8 on the codes come from the code RULE (three lengths: rung,
//     rung + 1, rung + 2; no swap up there) through a 64-bit bit writer;
//   * one DPP workgroup scan of the unit lengths, the chunk's bits to its slot.
#include "qb3_px_enc.h"
#include "qb3_wide.h"

namespace qb3dev {

template <typename T, uint64_t ORDER, bool STEP, int NT>
__global__ void __launch_bounds__(NT) enc_pxw_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    enc_scan_counter_reset(a);
    constexpr uint32_t UB = UBits<T>::v, UMASK = (1u << UB) - 1, NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nblocks = (uint32_t)a.g.nblocks, nbx = a.g.nbx;
    const uint64_t stride = a.g.stride;
    uint32_t *etab = (uint32_t *)smem;                      // 512 entries: the code table of rungs 1..7 (qb3_px.h)
    uint32_t *wsum = etab + 512;                            // 16 dwords: [0..3] scan, [8..11] rung of each wave's last lane
    uint32_t *outbuf = wsum + 16;                           // slot_dw dwords (a multiple of 4), 16-byte aligned
    // the table is asked for now and written to LDS before the first barrier: its round trip runs beside the pixel loads
    const uint4 tabv = ((const uint4 *)px_enc_tab.e)[tid & 127];
    for (uint32_t i = tid; i < a.slot_dw / 4; i += NT) ((uint4 *)outbuf)[i] = make_uint4(0, 0, 0, 0);
    const uint32_t etab_off = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)smem;

    const uint32_t chunk = a.chunk0 + blockIdx.x;      // (chunk0: the first chunk of this launch -- 0 but for the strips of a pipelined host call)
    const int64_t gs = (int64_t)chunk * (NT - 1) - 1 + tid; // lane 0 is the halo block
    const bool valid = gs >= 0 && gs < (int64_t)nblocks, payload = valid && tid >= 1;
    const uint32_t gblk = valid ? (uint32_t)gs : 0u;

    // ---- the block and the value entering it
    T w[4][4], pv = 0;
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int x = 0; x < 4; x++) w[r][x] = 0;
    constexpr uint32_t n15 = (uint32_t)(ORDER & 15);        // the pixel the curve visits last
    if (valid) {
        const uint32_t by = gblk / nbx, bx = gblk - by * nbx;
        const uint32_t x0 = (4 * bx + 4 > a.g.w) ? a.g.w - 4 : 4 * bx;     // last column / row is shifted, not padded (QB3encode.h:410-416)
        const uint32_t y0 = (4 * by + 4 > a.g.h) ? a.g.h - 4 : 4 * by;
        const T *p0 = (const T *)a.img + (uint64_t)y0 * stride + x0;
#pragma unroll
        for (int r = 0; r < 4; r++) pxw_load_row(p0 + (uint64_t)r * stride, w[r]);
        if (gblk) {
            const uint32_t pb = gblk - 1, pby = pb / nbx, pbx = pb - pby * nbx;
            const uint32_t px0 = (4 * pbx + 4 > a.g.w) ? a.g.w - 4 : 4 * pbx;
            const uint32_t py0 = (4 * pby + 4 > a.g.h) ? a.g.h - 4 : 4 * pby;
            pv = ((const T *)a.img)[(uint64_t)(py0 + (n15 >> 2)) * stride + px0 + (n15 & 3)];
        } else pv = (T)a0.st.prev[0];
    }
    // ---- curve order, running delta, mag-sign (QB3encode.h:423-437)
    T g[16], used = 0, prv = pv;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        constexpr uint64_t O = ORDER;
        const uint32_t nib = (uint32_t)(O >> (60 - 4 * i)) & 15u;
        const T v = w[nib >> 2][nib & 3];
        g[i] = mags_t<T>((T)(v - prv));
        used |= g[i];
        prv = v;
    }
    const T lastv = prv;
    const uint32_t rung = valid ? topbit_t<T>(used) : 0u;
    // rung of the block before: the neighbouring lane, the last lane of the wave before through LDS, the handle's state
    uint32_t prung = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rung, 0x138, 0xf, 0xf, false);      // wave_shr:1
    if (lane == 63) wsum[8 + wave] = rung;
    if (tid < 128) ((uint4 *)etab)[tid] = tabv;
    __syncthreads();                                        // (also: the bit buffer is zero)
    if (lane == 0 && wave) prung = wsum[8 + wave - 1];
    if (gblk == 0) prung = (uint32_t)a0.st.rung[0] & UMASK;

    // ---- the unit's length (QB3encode.h:155-280): switch, then rung 0: flag (+ 16 bits), else sixteen three-length codes
    uint32_t len = 0, delta = 0;
    uint32_t pc[6] = {0, 0, 0, 0, 0, 0}, pl[6] = {0, 0, 0, 0, 0, 0};
    const bool narrow = used > 1 && rung < 8;               // the sixteen values fit a byte each
    if (payload) {
        delta = (rung - prung) & UMASK;
        len = cs_len<UB>(delta);
        if (used <= 1) len += 1 + (used ? 16 : 0);
        else if (narrow) {
            uint32_t gq[4];
#pragma unroll
            for (int q = 0; q < 4; q++)
                gq[q] = (uint32_t)g[4 * q] | (uint32_t)g[4 * q + 1] << 8 | (uint32_t)g[4 * q + 2] << 16 | (uint32_t)g[4 * q + 3] << 24;
            len = px_unit_pieces<STEP>(gq, rung, len, cs_code<UB>(delta), etab_off + (8u << rung), pc, pl);
        } else {
            const T top = (T)((T)1 << rung), half = (T)(top >> 1);
            if (STEP) {     // clear the rung bit of the last value of a 1..10..0 rung-bit run (QB3encode.h:169-176)
                uint32_t bits = 0;
#pragma unroll
                for (int i = 0; i < 16; i++) bits |= (uint32_t)((g[i] >> rung) & 1) << i;
                if ((bits & (bits + 1)) == 0) {
                    const uint32_t n = __popc(bits);        // >= 1 here
#pragma unroll
                    for (int i = 0; i < 16; i++) if ((uint32_t)i + 1 == n) g[i] ^= top;
                }
            }
            uint32_t extra = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) extra += (uint32_t)(g[i] >= half) + (uint32_t)(g[i] >= top);
            len += 16 * rung + extra;
        }
    }
    // ---- one scan of the lengths (DPP inside the waves, the waves' sums through LDS)
    const uint32_t inc = wave_iscan32(len);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t pos = inc - len, total = 0;
#pragma unroll
    for (uint32_t i = 0; i < NW; i++) { const uint32_t s = wsum[i]; if (i < wave) pos += s; total += s; }

    if (payload && narrow) {
        LdsWriter32 wr;
        wr.init(outbuf, pos);
#pragma unroll
        for (int k = 0; k < 6; k++) wr.put(pc[k], pl[k]);
        wr.finish();
    } else if (payload) {
        LdsWriter wr;
        wr.init(outbuf, pos);
        const uint32_t csl = cs_len<UB>(delta), csc = cs_code<UB>(delta);
        if (used <= 1) {
            uint32_t bits = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) bits |= (uint32_t)(g[i] & 1) << i;
            wr.put(csc | ((uint32_t)used << csl) | (used ? bits << (csl + 1) : 0u), csl + 1 + (used ? 16 : 0));       // <= 8 + 17 bits
        } else {
            wr.put(csc, csl);
#pragma unroll
            for (int i = 0; i < 16; i++) put_value<T>(wr, g[i], rung);
        }
        wr.finish();
    }
    if (payload) {
        // coder state on leaving the image, for handle statefulness (QB3encode.h:446-449)
        if (gblk == nblocks - 1) { a.res->prev[0] = (uint64_t)lastv; a.res->rung[0] = rung; a.res->cf[0] = a0.st.cf[0]; }
        if (a.have_idx) {
            if (!a.idx_no_ulen) ((uint16_t *)a.idx.ulen)[gblk] = (uint16_t)len;
            const uint32_t seg = gblk / a.g.seg_blocks;
            if (seg * a.g.seg_blocks == gblk) {
                ((T *)a.idx.prev)[seg] = pv;
                a.idx.rung[seg] = (uint8_t)prung;
                a.idx.bitpos[seg] = ((uint64_t)chunk << 32) | pos;      // chunk-relative; enc_finish_kernel makes it a stream position
            }
        }
    }
    // the chunk's bits go to its slot; enc_concat_kernel moves them into place once every chunk is counted
    __syncthreads();
    const uint32_t nd4 = (total + 127) >> 7;
    uint4 *slot = (uint4 *)(a.scratch + (uint64_t)chunk * a.slot_dw);
    for (uint32_t d = tid; d < nd4; d += NT) slot[d] = ((const uint4 *)outbuf)[d];
    if (tid == 0) a.chunk_bits[chunk] = total;
}

template <typename T, int NT>
static void launch_enc_pxw_t(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    const bool step = a.g.mode != CM_FTL, z = a.g.order == ZCURVE;
    dim3 grid(a.chunk_end - a.chunk0, a.ntiles), block(NT);
    if (!z && !step) hipLaunchKernelGGL((enc_pxw_kernel<T, HILBERT, false, NT>), grid, block, plan.lds_bytes, st, a);
    else if (!z && step) hipLaunchKernelGGL((enc_pxw_kernel<T, HILBERT, true, NT>), grid, block, plan.lds_bytes, st, a);
    else if (z && !step) hipLaunchKernelGGL((enc_pxw_kernel<T, ZCURVE, false, NT>), grid, block, plan.lds_bytes, st, a);
    else hipLaunchKernelGGL((enc_pxw_kernel<T, ZCURVE, true, NT>), grid, block, plan.lds_bytes, st, a);
}
void launch_enc_pxw(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    if (a.g.tsz == 4) launch_enc_pxw_t<uint32_t, 256>(a, plan, st);
    else launch_enc_pxw_t<uint64_t, 128>(a, plan, st);
}

}  // namespace qb3dev
