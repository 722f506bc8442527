// qb3_amd/csrc/k_dec_pxw.hip -- 32/64-bit single-band decoder (FTL / BASE), wave per segment, lane per BLOCK
//
// The counterpart of dec_px_kernel (k_dec_px.hip) for wide types (reference decodeFTL<T> / decode<T>, QB3decode.h:293-412,
// 578-741, gdecode :142-290).  A WAVE owns an index segment of 64 blocks; with one band a block is a unit, so a lane decodes one
// unit and nothing in the wave is serial:
//   positions   a DPP wave scan of the unit lengths (from the out-of-band index, or -- BL -- the twelve-bit fields of the
//               segment's entry in the container's own table);
//   rungs       every lane reads its own switch code; rung = the segment's entry rung + the wave scan of the switches;
//   values      sixteen codes per lane out of the wave's staged words (LDS), rungs 1..7 through the code table, above
//               that by the code rule; the value entering a unit = the entry value + the wave scan of the unit totals.
// The waves of a workgroup share the 2 KB code table and nothing else: one barrier.  The block's rows leave as 16-byte
// stores (two for 64-bit data) straight to HBM: 64 lanes write one contiguous kilobyte per row, no LDS tile.
#include "qb3_wide.h"

namespace qb3dev {

__device__ __forceinline__ uint32_t wave_exscan_t(uint32_t v) { return wave_iscan32(v) - v; }
__device__ __forceinline__ uint16_t wave_exscan_t(uint16_t v) { return (uint16_t)(wave_iscan32((uint32_t)v) - (uint32_t)v); }
__device__ __forceinline__ uint64_t wave_exscan_t(uint64_t v) {
    const uint32_t lane = threadIdx.x & 63;
    uint64_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t y = __shfl_up(x, d, 64);
        if (lane >= (uint32_t)d) x += y;
    }
    return x - v;
}
__device__ __forceinline__ void pxw_store_row(uint32_t *p, const uint32_t (&r)[4]) {
    typedef uint32_t v4 __attribute__((ext_vector_type(4), aligned(4)));
    const v4 v = { r[0], r[1], r[2], r[3] };
    *(v4 *)p = v;
}
__device__ __forceinline__ void pxw_store_row(uint16_t *p, const uint16_t (&r)[4]) {
    typedef uint16_t v4 __attribute__((ext_vector_type(4), aligned(2)));
    const v4 v = { r[0], r[1], r[2], r[3] };
    *(v4 *)p = v;
}
__device__ __forceinline__ void pxw_store_row(uint64_t *p, const uint64_t (&r)[4]) {
    typedef uint64_t v2 __attribute__((ext_vector_type(2), aligned(8)));
    const v2 a = { r[0], r[1] }, b = { r[2], r[3] };
    *(v2 *)p = a; *(v2 *)(p + 2) = b;
}

template <typename T, uint64_t ORDER, bool STEP, bool BL>
__global__ void __launch_bounds__(256) dec_pxw_kernel(const DecArgs a0) {
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    constexpr uint32_t UB = UBits<T>::v, UMASK = (1u << UB) - 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t chk = BL ? a0.chk_wgs : 0u;          // the launch's first workgroups check a chunk of the container's table each (ix_check_chunk)
    if (blockIdx.x < chk) { ix_check_chunk(a, blockIdx.x, (uint32_t *)smem); return; }
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const uint32_t NB = a.g.seg_blocks, nbx = a.g.nbx;      // NB == 64
    const uint64_t stride = a.g.stride;
    uint16_t *dtab = (uint16_t *)smem;                      // 2 KB
    uint32_t *stage = (uint32_t *)(smem + 2048) + wave * (a.in_cap_dw + WIDE_PAD_DW);

    // loads that depend on nothing but the segment number go out first: their round trips overlap the table copy
    const uint64_t seg = a.seg0 + (uint64_t)(blockIdx.x - chk) * nwaves + wave;       // (seg0, seg_end: this launch's range of segments)
    const bool live = seg < a.seg_end;
    const uint64_t segc = live ? seg : 0;
    const uint32_t g0 = (uint32_t)(segc * NB), nblocks = (uint32_t)a.g.nblocks;
    const uint32_t nb_here = (nblocks - g0 < NB) ? nblocks - g0 : NB;
    const bool act = live && lane < nb_here;
    uint64_t P0, P1;
    uint32_t blen, rg0;
    T pv0;
    if (BL) {
        const uint8_t *e = ix_entry_at(a.ix, a.ix_per_chunk, a.ix_E, a.ix_pad, (uint32_t)segc);
        auto pos6 = [](const uint8_t *q) { uint64_t v = 0;
#pragma unroll
            for (uint32_t i = 0; i < 6; i++) v |= (uint64_t)q[i] << (8 * i);
            return v; };
        P0 = pos6(e);
        P1 = (segc + 1 < a.g.nseg) ? pos6(ix_entry_at(a.ix, a.ix_per_chunk, a.ix_E, a.ix_pad, (uint32_t)segc + 1)) : a.in_bits;
        rg0 = e[6] & UMASK;
        uint64_t v = 0;
#pragma unroll
        for (uint32_t i = 0; i < sizeof(T); i++) v |= (uint64_t)e[7 + i] << (8 * i);
        pv0 = (T)v;
        const uint8_t *fl = e + 7 + sizeof(T) + ((IX_BL_BITS_WIDE * lane) >> 3);
        blen = act ? (((uint32_t)fl[0] | (uint32_t)fl[1] << 8) >> ((IX_BL_BITS_WIDE * lane) & 7)) & ((1u << IX_BL_BITS_WIDE) - 1) : 0u;
    } else {
        P0 = a.idx.bitpos[segc];
        P1 = (segc + 1 < a.g.nseg) ? a.idx.bitpos[segc + 1] : a.in_bits;
        blen = act ? ((const uint16_t *)a.idx.ulen)[(uint64_t)g0 + lane] : 0u;
        rg0 = a.idx.rung[segc];
        pv0 = a.totals_only ? (T)0 : ((const T *)a.idx.prev)[segc];
    }
    for (uint32_t i = tid; i < 128; i += blockDim.x) ((uint4 *)dtab)[i] = ((const uint4 *)wide_dec_tab.e)[i];
    __syncthreads();                                        // the only workgroup barrier
    if (!live) return;
    if (!BL && P1 < P0) P1 = P0;        // (the last segment of a truncated stream starts behind its end: it reads zeros)
    const uint64_t w0 = (a.in_bit0 + P0) >> 5;
    const uint64_t endw_abs = (a.in_bit0 + a.in_bits + 31) >> 5;
    const uint64_t ndw64 = ((a.in_bit0 + P1 + 31) >> 5) - w0;
    // positions out of a container's table are untrusted: a segment that does not lie inside the stream, or is longer than
    // the longest valid one, reads nothing (status bit 3: "this index does not describe this stream")
    const bool sane = P0 <= P1 && (!BL || P1 <= a.in_bits);       // (a truncated stream reads as zeros behind its end, like the reference's: bitstream.h:36)
    const bool fits = sane && ndw64 <= a.in_cap_dw;
    // (the staging may be sized for this stream's average segment: a longer -- but valid -- one raises status bit 4 and the
    // host runs the call again with the worst case)
    const uint32_t misfit = (sane && ndw64 <= a.in_cap_full) ? 16u : 8u;
    const uint32_t ndw = fits ? (uint32_t)ndw64 : 0;
    for (uint32_t base = 0; base < ndw + WIDE_PAD_DW; base += 512) { // eight loads in flight per lane, then eight LDS stores
        uint32_t sw[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t i = base + lane + 64 * k;
            sw[k] = (i < ndw && w0 + i < endw_abs) ? a.in32[w0 + i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t i = base + lane + 64 * k;
            if (i < ndw + WIDE_PAD_DW) stage[i] = sw[k];
        }
    }
    // the wave reads what its own lanes staged: LDS operations of a wave execute in order, the fence is for the compiler
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const uint32_t limit = 32 * ndw;                        // no unit starts beyond the staged bits (WIDE_PAD_DW zero words follow)
    const uint32_t cpos = (uint32_t)(a.in_bit0 + P0 - 32 * w0);
    bool bad = !fits;
    const uint32_t binc = wave_iscan32(blen);               // inclusive: lane 63 holds the bits of the segment
    uint32_t pos = cpos + binc - blen, gpos = 0;
    pos = pos < limit ? pos : limit;
    bool sig = false;
    const LdsWords sw = (LdsWords)stage;
    const uint32_t d = dec3_switch<T, LdsWords>(sw, ndw + WIDE_PAD_DW, pos, &gpos, &sig);
    if (act && sig && STEP) bad = true;                     // common-factor / index unit in a BASE stream: not handled here
    const uint32_t rung = (rg0 + wave_iscan32(act ? d : 0u)) & UMASK;
    T run[16];
    uint32_t end = 0;
    dec3_group<T, STEP, LdsWords, sizeof(T) >= 4>(sw, ndw + WIDE_PAD_DW, gpos, rung, dtab, run, &end);
    if (BL && act && end != pos + blen) bad = true;         // the table's lengths are not this stream's
    const T usum = act ? run[15] : (T)0;
    const T sex = wave_exscan_t(usum);
    if (a.totals_only) {    // a plain stream, first pass: leave the segment's sum where the entering value goes (prev_scan_kernel)
        if (lane == 63) ((T *)a.idx.prev)[seg] = (T)(sex + usum);
        if (bad) atomicOr(a.status, fits ? 1u : misfit);
        return;
    }
    if (act) {
        const T pv = (T)(pv0 + sex);
        T o[4][4];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            constexpr uint64_t O = ORDER;
            const uint32_t nib = (uint32_t)(O >> (60 - 4 * i)) & 15u;
            o[nib >> 2][nib & 3] = (T)(run[i] + pv);
        }
        const uint32_t g = g0 + lane, by = g / nbx, bx = g - by * nbx;
        const uint32_t x0 = (4 * bx + 4 > a.g.w) ? a.g.w - 4 : 4 * bx;     // last column / row is shifted, not padded
        const uint32_t y0 = (4 * by + 4 > a.g.h) ? a.g.h - 4 : 4 * by;
        T *p0 = (T *)a.img + (uint64_t)y0 * stride + x0;
#pragma unroll
        for (int y = 0; y < 4; y++) pxw_store_row(p0 + (uint64_t)y * stride, o[y]);
    }
    if (bad) atomicOr(a.status, fits ? 1u : misfit);
    if (lane == 63 && seg == a.g.nseg - 1 && fits) {        // reference: more than 7 unused bits at the end is a failure
        const uint64_t used = (uint64_t)(cpos + binc) + 32 * w0 - a.in_bit0;
        if (used > a.in_bits) atomicOr(a.status, 4u);
        else if (a.in_bits - used > 7) atomicOr(a.status, 2u);
    }
}

// ---- the common-factor modes (QB3M_BEST family; reference decode<T>, QB3decode.h:578-741: normal units :619-623, common-factor
// units :629-679, index units :680-715).  The index (or, BL, the container's own table) holds ONE DWORD PER BLOCK: the unit's
// bits | the rung it is entered with << 16 -- a common-factor unit leaves its band at the rung of the MULTIPLIED values
// (:664), so rungs are not a scan of the switch codes here, and the encoder knows them -- and per segment the factor in force.
// A lane parses its unit (parse_unit, qb3_kernels.h).  A unit that says "same factor as before" needs the band's last
// writer: a ballot of the lanes whose unit brought a factor, the nearest one below the lane, its value by a lane permute --
// else the segment entry's, which is what every lane assumed; only a lane that assumed wrongly parses its unit again.
// Every unit is checked against its length field, its leaving rung against the next block's entering rung.
template <typename T, uint64_t ORDER, bool BL>
__global__ void __launch_bounds__(256) dec_pxw_best_kernel(const DecArgs a0) {
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    constexpr uint32_t UB = UBits<T>::v, UMASK = (1u << UB) - 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t chk = BL ? a0.chk_wgs : 0u;          // the launch's first workgroups check a chunk of the container's table each (ix_check_chunk)
    if (blockIdx.x < chk) { ix_check_chunk(a, blockIdx.x, (uint32_t *)smem); return; }
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const uint32_t NB = 64, nbx = a.g.nbx;
    const uint64_t stride = a.g.stride;
    uint32_t *stage = (uint32_t *)smem + wave * (a.in_cap_dw + WIDE_PAD_DW);         // nothing is shared between the waves: no barrier
    const uint64_t seg = a.seg0 + (uint64_t)(blockIdx.x - chk) * nwaves + wave;
    if (seg >= a.seg_end) return;
    const uint32_t g0 = (uint32_t)(seg * NB), nblocks = (uint32_t)a.g.nblocks;
    const uint32_t nb_here = (nblocks - g0 < NB) ? nblocks - g0 : NB;
    const bool act = lane < nb_here;
    uint64_t P0, P1;
    uint32_t bt;                                        // the unit's bits | entering rung << 16
    T pv0, cf0;
    if (BL) {
        const uint8_t *e = ix_entry_at(a.ix, a.ix_per_chunk, a.ix_E, a.ix_pad, (uint32_t)seg);
        auto pos6 = [](const uint8_t *q) { uint64_t v = 0;
#pragma unroll
            for (uint32_t i = 0; i < 6; i++) v |= (uint64_t)q[i] << (8 * i);
            return v; };
        P0 = pos6(e);
        P1 = (seg + 1 < a.g.nseg) ? pos6(ix_entry_at(a.ix, a.ix_per_chunk, a.ix_E, a.ix_pad, (uint32_t)seg + 1)) : a.in_bits;
        uint64_t v = 0, f = 0;
#pragma unroll
        for (uint32_t i = 0; i < sizeof(T); i++) { v |= (uint64_t)e[7 + i] << (8 * i); f |= (uint64_t)e[7 + sizeof(T) + i] << (8 * i); }
        pv0 = (T)v; cf0 = (T)f;
        const uint8_t *fp = e + 7 + 2 * sizeof(T) + IX_BL_BEST_BYTES * lane;
        const uint32_t fld = act ? (uint32_t)fp[0] | (uint32_t)fp[1] << 8 | (uint32_t)fp[2] << 16 : 0u;
        bt = (fld & 0xfffu) | ((fld >> 12) & UMASK) << 16;          // (the entry's own rung byte repeats block 0's field; the field is what is used)
    } else {
        P0 = a.idx.bitpos[seg];
        P1 = (seg + 1 < a.g.nseg) ? a.idx.bitpos[seg + 1] : a.in_bits;
        bt = act ? ((const uint32_t *)a.idx.ulen)[(uint64_t)g0 + lane] : 0u;
        pv0 = ((const T *)a.idx.prev)[seg]; cf0 = ((const T *)a.idx.cf)[seg];
    }
    if (!BL && P1 < P0) P1 = P0;        // (the last segment of a truncated stream starts behind its end: it reads zeros)
    const uint64_t w0 = (a.in_bit0 + P0) >> 5;
    const uint64_t endw_abs = (a.in_bit0 + a.in_bits + 31) >> 5;
    const uint64_t ndw64 = ((a.in_bit0 + P1 + 31) >> 5) - w0;
    const bool sane = P0 <= P1 && (!BL || P1 <= a.in_bits);       // (a truncated stream reads as zeros behind its end, like the reference's: bitstream.h:36)
    const bool fits = sane && ndw64 <= a.in_cap_dw;
    const uint32_t misfit = (sane && ndw64 <= a.in_cap_full) ? 16u : 8u;      // 16: the staging was sized for the stream's average; the host calls again with the worst case
    const uint32_t ndw = fits ? (uint32_t)ndw64 : 0;
    for (uint32_t base = 0; base < ndw + WIDE_PAD_DW; base += 512) {
        uint32_t sw[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t i = base + lane + 64 * k;
            sw[k] = (i < ndw && w0 + i < endw_abs) ? a.in32[w0 + i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t i = base + lane + 64 * k;
            if (i < ndw + WIDE_PAD_DW) stage[i] = sw[k];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const uint32_t limit = 32 * ndw;
    const uint32_t cpos = (uint32_t)(a.in_bit0 + P0 - 32 * w0);
    bool bad = !fits;
    const uint32_t blen = bt & 0xffffu, oldrung = (bt >> 16) & UMASK;
    const uint32_t binc = wave_iscan32(blen);
    uint32_t pos = cpos + binc - blen;
    pos = pos < limit ? pos : limit;
    // the rung the NEXT block is entered with is the rung this block's unit must leave: checked, not trusted
    const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(bt >> 16), 0x130, 0xf, 0xf, false);      // wave_shl:1
    T g[16], pcf = cf0, cf_in = cf0;
#pragma unroll
    for (int i = 0; i < 16; i++) g[i] = 0;
    uint32_t rung = oldrung, flags = 0, end = pos;
    bool ok = true, need = act;
#pragma nounroll
    for (int pass = 0; pass < 2; pass++) {
        if (need) {
            ReaderT<LdsWords> rd;
            rd.init((LdsWords)stage, pos, 32ull * (ndw + WIDE_PAD_DW));
            rung = oldrung; pcf = cf_in; flags = 0;
            ok = parse_unit<T, CM_BEST, ReaderT<LdsWords>>(rd, rung, pcf, g, &flags);
            end = (uint32_t)rd.position();
        }
        if (pass) break;
        // the factor in force for a unit that takes the band's: the nearest lane below whose unit brought one
        const uint64_t wm = __ballot(act && (flags & 2u));
        if (!wm) break;                                             // (no writer in the segment: every lane assumed right)
        const uint64_t below = wm & ((1ull << lane) - 1);
        const uint32_t src = below ? 63u - (uint32_t)__clzll((long long)below) : lane;
        const T got = (T)__shfl((unsigned long long)pcf, (int)src, 64);
        need = act && (flags & 1u) && below && got != cf0;
        cf_in = got;
        if (!__any(need)) break;
    }
    if (act && (!ok || end != pos + blen)) bad = true;              // malformed unit, or the lengths are not this stream's
    if (act && lane + 1 < nb_here && rung != (nxt & UMASK)) bad = true;
    T run[16], acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { acc = (T)(acc + smag_t<T>(g[i])); run[i] = acc; }
    const T usum = act ? acc : (T)0;
    const T sex = wave_exscan_t(usum);
    if (act) {
        const T pv = (T)(pv0 + sex);
        T o[4][4];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            constexpr uint64_t O = ORDER;
            const uint32_t nib = (uint32_t)(O >> (60 - 4 * i)) & 15u;
            o[nib >> 2][nib & 3] = (T)(run[i] + pv);
        }
        const uint32_t gb = g0 + lane, by = gb / nbx, bx = gb - by * nbx;
        const uint32_t x0 = (4 * bx + 4 > a.g.w) ? a.g.w - 4 : 4 * bx;
        const uint32_t y0 = (4 * by + 4 > a.g.h) ? a.g.h - 4 : 4 * by;
        T *p0 = (T *)a.img + (uint64_t)y0 * stride + x0;
#pragma unroll
        for (int y = 0; y < 4; y++) pxw_store_row(p0 + (uint64_t)y * stride, o[y]);
    }
    if (bad) atomicOr(a.status, fits ? 1u : misfit);
    if (lane == 63 && seg == a.g.nseg - 1 && fits) {
        const uint64_t used = (uint64_t)(cpos + binc) + 32 * w0 - a.in_bit0;
        if (used > a.in_bits) atomicOr(a.status, 4u);
        else if (a.in_bits - used > 7) atomicOr(a.status, 2u);
    }
}

template <typename T>
static void launch_dec_pxw_best_t(const DecArgs &a, const DecPlan &plan, hipStream_t st) {
    dim3 grid((uint32_t)((a.seg_end - a.seg0 + 3) / 4) + (a.bl_mode ? a.chk_wgs : 0u), a.ntiles), block(256);
    const size_t lds = plan.lds_pxw;
    const bool z = a.g.order == ZCURVE;
    if (a.bl_mode) {
        if (z) hipLaunchKernelGGL((dec_pxw_best_kernel<T, ZCURVE, true>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((dec_pxw_best_kernel<T, HILBERT, true>), grid, block, lds, st, a);
        return;
    }
    if (z) hipLaunchKernelGGL((dec_pxw_best_kernel<T, ZCURVE, false>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((dec_pxw_best_kernel<T, HILBERT, false>), grid, block, lds, st, a);
}
void launch_dec_pxw_best(const DecArgs &a, const DecPlan &plan, hipStream_t st) {
    if (a.g.tsz == 2) launch_dec_pxw_best_t<uint16_t>(a, plan, st);
    else if (a.g.tsz == 4) launch_dec_pxw_best_t<uint32_t>(a, plan, st);
    else launch_dec_pxw_best_t<uint64_t>(a, plan, st);
}

template <typename T>
static void launch_dec_pxw_t(const DecArgs &a, const DecPlan &plan, hipStream_t st) {
    const bool step = a.g.mode != CM_FTL, z = a.g.order == ZCURVE;
    dim3 grid((uint32_t)((a.seg_end - a.seg0 + 3) / 4) + (a.bl_mode ? a.chk_wgs : 0u), a.ntiles), block(256);
    const size_t lds = plan.lds_pxw;
    if (a.bl_mode) {
        if (!z && !step) hipLaunchKernelGGL((dec_pxw_kernel<T, HILBERT, false, true>), grid, block, lds, st, a);
        else if (!z && step) hipLaunchKernelGGL((dec_pxw_kernel<T, HILBERT, true, true>), grid, block, lds, st, a);
        else if (z && !step) hipLaunchKernelGGL((dec_pxw_kernel<T, ZCURVE, false, true>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((dec_pxw_kernel<T, ZCURVE, true, true>), grid, block, lds, st, a);
        return;
    }
    if (!z && !step) hipLaunchKernelGGL((dec_pxw_kernel<T, HILBERT, false, false>), grid, block, lds, st, a);
    else if (!z && step) hipLaunchKernelGGL((dec_pxw_kernel<T, HILBERT, true, false>), grid, block, lds, st, a);
    else if (z && !step) hipLaunchKernelGGL((dec_pxw_kernel<T, ZCURVE, false, false>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((dec_pxw_kernel<T, ZCURVE, true, false>), grid, block, lds, st, a);
}
void launch_dec_pxw(const DecArgs &a, const DecPlan &plan, hipStream_t st) {
    if (a.g.tsz == 4) launch_dec_pxw_t<uint32_t>(a, plan, st);
    else launch_dec_pxw_t<uint64_t>(a, plan, st);
}

}  // namespace qb3dev
