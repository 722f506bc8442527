// qb3_amd/csrc/k_dec_px_best.hip -- 8-bit grey / RGB / RGBA decoder for the common-factor modes (QB3M_BEST family): wave per
// segment, lane per block
//
// Reference: decode<T> (QB3decode.h:578-741): normal units (:619-623), common-factor units (:629-679), index units (:680-715).
// dec_px_kernel's organisation (k_dec_px.hip) with what the common-factor modes add.  The index holds, besides the segment
// entries (bit position, entering value, factor and rung per band), ONE DWORD PER BLOCK: the block's bits and the rungs its
// units are entered with -- a common-factor unit leaves its band at the rung of the MULTIPLIED values (QB3decode.h:664), so
// rungs are not a scan of the switch codes here, and the encoder knows them.  A lane finds its block by a wave scan of the
// lengths and decodes its bands one after the other (a unit starts where the one before ended).  A unit without the signal
// is QB3M_BASE's and goes through px_group.  A unit with it is parsed by the lane on its own: the divided group or the
// indexed values into sixteen bytes, and for a common-factor unit whether it brings its own factor.  The factor in force
// for a unit that says "same as before" is the band's last writer: a ballot of the writers, the nearest one below the lane,
// its value by a lane permute -- else the segment entry's.  Then multiply, running sums, and on as dec_px_kernel.
#include "qb3_px.h"

namespace qb3dev {

// A unit that started with the signal, read from LDS at bit `pos` (just behind the signal): g = the sixteen mag-sign values
// (common factor: of the DIVIDED group, not yet multiplied).  Returns false on a malformed unit.
// kind: 0 common factor with its own factor (*cfv = cf - 2), 1 common factor with the band's factor in force, 2 index form
__device__ __forceinline__ bool best_slow_unit(uint32_t pos, uint32_t oldrung, uint8_t (&g)[16], uint32_t *kind, uint32_t *cfv, uint32_t *rung_out, uint32_t *end) {
    typedef uint8_t T;
    constexpr uint32_t UB = 3, UMASK = 7;
    ReaderT<LdsWords> rd;
    rd.init(lds_at(0), pos, ~0ull >> 8);
    bool sig2;
    bool ok = true;
    const uint32_t r = (oldrung + get_switch_noflag<UB>(rd, sig2)) & UMASK;
    *cfv = 0; *rung_out = r;
    if (r != UMASK) {           // common factor (QB3decode.h:629-679)
        *kind = 1;
        uint32_t cfrung = r;
        if (rd.get(1)) {
            const uint32_t own = rd.get(1);
            if (own) {
                cfrung = (r + get_switch_noflag<UB>(rd, sig2)) & UMASK;
                if (cfrung == r || cfrung == 0) ok = false;
            }
            const uint32_t vr = (cfrung - own) & UMASK;
            uint32_t v;
            if (vr == 0) v = rd.get(1);
            else { const T t = get_value<T>(rd, vr); v = (vr >= 3) ? unswap<T>(t, vr) : t; }    // factor values: rungs 1, 2 unswapped (QB3encode.h:144-150)
            *cfv = (v + (own << cfrung)) & 0xffu;
            *kind = 0;
        }
        if (r) get_group<T, true>(rd, r, g);
        else {
            const uint32_t bits = rd.get(16);
#pragma unroll
            for (uint32_t i = 0; i < 16; i++) g[i] = (T)((bits >> i) & 1);
        }
    } else {                    // index form (QB3decode.h:680-715)
        *kind = 2;
        const uint32_t r2 = (oldrung + get_switch_noflag<UB>(rd, sig2)) & UMASK;
        *rung_out = r2;
        if (r2 == 0) ok = false;
        uint64_t ix = 0;        // 16 x 3 bit indices packed
        uint32_t maxidx = 0, ibits = 0;
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) {
            rd.ensure(4);
            const uint32_t x = (uint32_t)rd.buf;
            uint32_t v, len;    // plain rung 2 code
            if (!(x & 1)) { v = (x & 3) >> 1; len = 2; }
            else if (!(x & 2)) { v = ((x >> 2) & 1) | 2; len = 3; }
            else { v = ((x >> 2) & 3) | 4; len = 4; }
            rd.skip(len);
            ibits += len;
            ix |= (uint64_t)v << (3 * i);
            maxidx = v > maxidx ? v : maxidx;
        }
        if (ibits > 52) ok = false;
        T tab[8];
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) {
            tab[i] = 0;
            if (i <= maxidx && ok) { const T t = get_value<T>(rd, r2 ? r2 : 1); tab[i] = (r2 >= 3) ? unswap<T>(t, r2) : t; }
        }
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) {
            const uint32_t j = (uint32_t)(ix >> (3 * i)) & 7;
            T v = tab[0];
#pragma unroll
            for (uint32_t k = 1; k < 8; k++) v = (j == k) ? tab[k] : v;
            g[i] = v;
        }
    }
    *end = (uint32_t)rd.position();
    return ok;
}

// BL: no index -- position, entering rungs, values and factors and the per-block fields come from the segment's entry of the
// container's restart table (layout: ix_bl_best_fill_kernel, k_enc_post.hip).  A table is untrusted input: positions are
// checked against the stream, every block against its length, every unit's rung against the next block's entering rung.
template <int B, bool RGB, uint64_t ORDER, bool BL>
__global__ void __launch_bounds__(256) dec_px_best_kernel(const DecArgs a0) {
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t chk = BL ? a0.chk_wgs : 0u;          // the launch's first workgroups check a chunk of the container's table each (ix_check_chunk)
    if (blockIdx.x < chk) { ix_check_chunk(a, blockIdx.x, (uint32_t *)smem); return; }
    constexpr int NW = (B + 1) / 2;                     // 32-bit words of a scan packed 16 bits per band
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const uint32_t NB = 64, nbx = a.g.nbx;              // a WAVE owns a segment of 64 blocks, nothing is shared but the table
    const uint64_t stride = a.g.stride;

    uint32_t *tab = (uint32_t *)smem;                   // 4 KB, at LDS address 0 (the table addressing relies on it)
    uint32_t *stage = tab + 1024 + wave * (a.in_cap_dw + 8);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)smem;
    const uint32_t stage_bit0 = 8 * (lds0 + (uint32_t)((uint8_t *)stage - smem));
    const uint64_t seg = a.seg0 + (uint64_t)(blockIdx.x - chk) * nwaves + wave;       // (seg0, seg_end: this launch's range of segments)
    const bool live = seg < a.seg_end;
    const uint64_t segc = live ? seg : 0;
    const uint32_t g0 = (uint32_t)(segc * NB), nblocks = (uint32_t)a.g.nblocks;
    const uint32_t nb_here = (nblocks - g0 < NB) ? nblocks - g0 : NB;
    const bool act = live && lane < nb_here;
    uint64_t P0, P1;
    uint32_t bt = 0, pv0[B], cf0[B];                    // bt: the block's bits | entering rungs << 16 (four bits a band)
    if (BL) {
        const uint8_t *e = ix_entry_at(a.ix, a.ix_per_chunk, a.ix_E, a.ix_pad, (uint32_t)segc);
        auto pos6 = [](const uint8_t *q) { uint64_t v = 0;
#pragma unroll
            for (uint32_t i = 0; i < 6; i++) v |= (uint64_t)q[i] << (8 * i);
            return v; };
        P0 = pos6(e);
        P1 = (segc + 1 < a.g.nseg) ? pos6(ix_entry_at(a.ix, a.ix_per_chunk, a.ix_E, a.ix_pad, (uint32_t)segc + 1)) : a.in_bits;
        if (P1 > a.in_bits) P1 = 0;                     // (a position behind the stream: the segment will not fit)
#pragma unroll
        for (int c = 0; c < B; c++) { pv0[c] = e[6 + B + c]; cf0[c] = e[6 + 2 * B + c]; }
        const uint8_t *fp = e + 6 + 3 * B + IX_BL_BEST_BYTES * lane;
        const uint32_t f = act ? (uint32_t)fp[0] | (uint32_t)fp[1] << 8 | (uint32_t)fp[2] << 16 : 0u;
        bt = f & 0xfffu;
#pragma unroll
        for (int c = 0; c < 4; c++) bt |= ((f >> (12 + 3 * c)) & 7u) << (16 + 4 * c);
        // (the entry's own rung bytes repeat block 0's field; the field is what is used)
    } else {
        P0 = a.idx.bitpos[segc];
        P1 = (segc + 1 < a.g.nseg) ? a.idx.bitpos[segc + 1] : a.in_bits;
        if (P1 < P0) P1 = P0;       // (the last segment of a truncated stream starts behind its end: it reads zeros, like the reference's reader, bitstream.h:36)
        bt = act ? ((const uint32_t *)a.idx.ulen)[(uint64_t)g0 + lane] : 0u;
#pragma unroll
        for (int c = 0; c < B; c++) { pv0[c] = ((const uint8_t *)a.idx.prev)[segc * B + c]; cf0[c] = ((const uint8_t *)a.idx.cf)[segc * B + c]; }
    }
    for (uint32_t i = tid; i < 256; i += blockDim.x) ((uint4 *)tab)[i] = ((const uint4 *)px_dec_tab.e)[i];
    __syncthreads();                                    // the only workgroup barrier
    if (!live) return;
    const uint64_t w0 = (a.in_bit0 + P0) >> 5;
    const uint64_t endw_abs = (a.in_bit0 + a.in_bits + 31) >> 5;
    const uint64_t ndw64 = ((a.in_bit0 + P1 + 31) >> 5) - w0;
    const bool fits = P1 >= P0 && ndw64 <= a.in_cap_dw && lds0 == 0;      // the staging area holds the longest valid segment
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
    // the wave reads what its own lanes staged: LDS operations of a wave execute in order, the fence is for the compiler
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const uint32_t limit = stage_bit0 + 32 * ndw;       // no unit starts beyond the staged bits (8 zero words follow)
    const uint32_t cpos = stage_bit0 + (uint32_t)(a.in_bit0 + P0 - 32 * w0);
    bool bad = !fits;
    const uint32_t blen = bt & 0xffffu;
    const uint32_t binc = wave_iscan32(blen);           // inclusive: lane 63 holds the bits of the segment
    uint32_t pos = cpos + binc - blen;
    const uint32_t blk_end = pos + blen;
    // the rungs the NEXT block is entered with are the rungs this block's units must leave: checked, not trusted
    const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(bt >> 16), 0x130, 0xf, 0xf, false);      // wave_shl:1
    uint32_t rp[B][8], spk[NW], sinc[NW];
    bool clamped = false;
#pragma unroll
    for (int k = 0; k < NW; k++) spk[k] = 0;
#pragma unroll
    for (int c = 0; c < B; c++) {
        if (pos > limit) { pos = limit; clamped = true; }       // (a stream cut short: the unit reads zeros wherever it starts behind the end)
        const uint32_t oldrung = (bt >> (16 + 4 * c)) & 7u;
        bool sig; uint32_t csl;
        const uint32_t d = px_switch(pos, &csl, &sig);
        uint32_t rung = (oldrung + d) & 7u, end = pos, tot = 0;
        // a unit in common-factor or index form: the lane parses it; `same`: it waits for the band's factor in force
        uint8_t g[16];
        uint32_t kind = 3, cfv = 0;
        const bool slow = act && sig;
        if (slow) { if (!best_slow_unit(pos + csl, oldrung, g, &kind, &cfv, &rung, &end)) bad = true; }
        else tot = px_group<true>(pos + csl, rung, rp[c], &end) & 0xffu;
        if (__any(slow)) {
            // the factor in force: the nearest lane below with a unit that brought its own, else the segment entry's
            const uint64_t wm = __ballot(slow && kind == 0);
            const uint64_t below = wm & ((1ull << lane) - 1);
            const uint32_t src = below ? 63u - (uint32_t)__clzll((long long)below) : lane;
            const uint32_t got = (uint32_t)__shfl((int)cfv, (int)src, 64);
            if (slow) {
                if (kind == 1) cfv = below ? got : cf0[c];
                uint32_t acc = 0, used = 0;
                const uint32_t cf = (cfv + 2) & 0xffu;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    uint32_t v = g[i];
                    if (kind < 2) v = ((((v >> 1) + (v & 1)) * (cf << 1)) - (v & 1)) & 0xffu;     // magsmul (QB3decode.h:575)
                    used |= v;
                    acc += (v >> 1) ^ (0u - (v & 1u));                                          // mag-sign undone
                    if (i & 1) rp[c][i >> 1] |= acc << 16; else rp[c][i >> 1] = acc & 0xffffu;
                }
                tot = acc & 0xffu;
                if (kind < 2) {
                    // the band's rung is that of the multiplied values (QB3decode.h:664); a factor above them: malformed (:665)
                    if (rung == 0) rung = topbit32(((cf - 1) << 1) | 1);
                    else { rung = topbit32(used | 1); if (cf > used) bad = true; }
                }
            }
        }
        if (act && lane + 1 < nb_here && rung != ((nxt >> (4 * c)) & 7u)) bad = true;
        spk[c >> 1] |= (act ? tot : 0u) << (16 * (c & 1));
        pos = end;
    }
    if (act && !clamped && pos != blk_end) bad = true;  // the index's lengths are not this stream's
#pragma unroll
    for (int k = 0; k < NW; k++) sinc[k] = wave_iscan32(spk[k]);
    if (act) {
        // entering value, then the core band (reference QB3decode.h:730-737)
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
        const uint32_t gb = g0 + lane, by = gb / nbx, bx = gb - by * nbx;
        const uint32_t x0 = (4 * bx + 4 > a.g.w) ? a.g.w - 4 : 4 * bx;     // last column / row is shifted, not padded
        const uint32_t y0 = (4 * by + 4 > a.g.h) ? a.g.h - 4 : 4 * by;
        uint8_t *p0 = (uint8_t *)a.img + (uint64_t)y0 * stride + (uint64_t)x0 * B;
#pragma unroll
        for (int y = 0; y < 4; y++) {
            uint32_t ow[B];
#pragma unroll
            for (int k = 0; k < B; k++) {
                uint32_t half2[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int b0 = 4 * k + 2 * h, b1 = b0 + 1;
                    const int i0 = curve_pos_of(ORDER, b0 / B, y), i1 = curve_pos_of(ORDER, b1 / B, y);
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
                const uint32_t head = 4 - al, sh = 8 * head;
#pragma unroll
                for (uint32_t t = 0; t < 3; t++) if (t < head) row[t] = (uint8_t)(ow[0] >> (8 * t));
                uint32_t *mid = (uint32_t *)(row + head);
#pragma unroll
                for (int k = 0; k + 1 < B; k++) mid[k] = __builtin_amdgcn_alignbit(ow[k + 1], ow[k], sh);
                uint8_t *tail = row + head + 4 * (B - 1);
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
static void launch_dec_px_best_b(const DecArgs &a, const DecPlan &plan, hipStream_t st) {
    dim3 grid((uint32_t)((a.seg_end - a.seg0 + 3) / 4) + (a.bl_mode ? a.chk_wgs : 0u), a.ntiles), block(256);
    if (a.bl_mode) {
        if (a.g.order == ZCURVE) hipLaunchKernelGGL((dec_px_best_kernel<B, RGB, ZCURVE, true>), grid, block, plan.lds_px, st, a);
        else hipLaunchKernelGGL((dec_px_best_kernel<B, RGB, HILBERT, true>), grid, block, plan.lds_px, st, a);
        return;
    }
    if (a.g.order == ZCURVE) hipLaunchKernelGGL((dec_px_best_kernel<B, RGB, ZCURVE, false>), grid, block, plan.lds_px, st, a);
    else hipLaunchKernelGGL((dec_px_best_kernel<B, RGB, HILBERT, false>), grid, block, plan.lds_px, st, a);
}
void launch_dec_px_best(const DecArgs &a, const DecPlan &plan, hipStream_t st) {
    if (a.g.bands == 1) launch_dec_px_best_b<1, false>(a, plan, st);
    else if (a.g.bands == 3) { if (plan.px_rgb) launch_dec_px_best_b<3, true>(a, plan, st); else launch_dec_px_best_b<3, false>(a, plan, st); }
    else { if (plan.px_rgb) launch_dec_px_best_b<4, true>(a, plan, st); else launch_dec_px_best_b<4, false>(a, plan, st); }
}

}  // namespace qb3dev
