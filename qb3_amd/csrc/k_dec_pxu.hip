// qb3_amd/csrc/k_dec_pxu.hip -- decoders for every raster no lane-per-block kernel takes: wave per segment, lane per UNIT
//
// 8-bit rasters of 2 or more than 4 bands, 16-bit rasters of an odd band count above 4, 32/64-bit rasters of several bands, and the
// common-factor streams of every raster of several bands but 8-bit RGB / RGBA (lane_per_unit_shape, qb3_dev.h).  The organisation is
// dec_pxw_kernel's (k_dec_pxw.hip; reference decodeFTL<T> / decode<T>, QB3decode.h:293-412, 578-741, gdecode :142-290) with the
// band as one more coordinate of a lane: a WAVE owns an index segment of 64 / bands blocks, lane = block * bands + band, and
// nothing in the wave is serial:
//   positions   a DPP wave scan of the unit lengths (units follow one another in the stream in lane order);
//   rungs       FTL / BASE: every lane reads its own switch code, rung = the band's entry rung + a scan of the switches ALONG THE
//               BAND (lanes `bands` apart); common factor: the index holds the rung every unit is entered with;
//   values      sixteen codes per lane out of the wave's staged words; the value entering a unit = the band's entry value + a
//               scan along the band of the unit totals; the core band is added back by a lane permute (reference :560-567).
// Pixels: the lanes write their values into a tile in LDS laid out like the image ([row][block][x][band]; the tile takes the place of
// the staged words, which are dead by then) and the wave stores the four rows of its blocks in pieces of up to sixteen bytes -- whole
// runs of 4 x bands values a block, whatever the band count.
#include "qb3_wide.h"

namespace qb3dev {

template <typename T> __device__ __forceinline__ T pxu_shfl(T v, uint32_t src) {
    if constexpr (sizeof(T) == 8) return (T)__shfl((unsigned long long)v, (int)src, 64);
    else return (T)__shfl((unsigned)v, (int)src, 64);
}
// exclusive scan along the lanes `stride` apart (a band's units)
template <typename T> __device__ __forceinline__ T pxu_exscan_band(T v, uint32_t stride) {
    const uint32_t lane = threadIdx.x & 63;
    T x = v;
    for (uint32_t d = stride; d < 64; d <<= 1) {
        T y;
        if constexpr (sizeof(T) == 8) y = (T)__shfl_up((unsigned long long)x, d, 64);
        else y = (T)__shfl_up((unsigned)x, d, 64);
        if (lane >= d) x = (T)(x + y);
    }
    return (T)(x - v);
}
// the band map, four bits a band (a lane indexes it by its band: kernel arguments live in scalar registers)
__device__ __forceinline__ uint64_t pxu_band_nibbles(const DecArgs &a0) {
    uint64_t n = 0;
    for (uint32_t c = 0; c < (uint32_t)MAXBANDS; c++) n |= (uint64_t)(a0.g.cband[c] & 15u) << (4 * c);
    return n;
}
template <int N> struct PxuPiece;
template <> struct PxuPiece<16> { typedef uint32_t v __attribute__((ext_vector_type(4))); };
template <> struct PxuPiece<8> { typedef uint32_t v __attribute__((ext_vector_type(2))); };
template <> struct PxuPiece<4> { typedef uint32_t v; };
template <int N, typename T>
__device__ __forceinline__ void pxu_store_rows(uint8_t *img, const uint8_t *tile, const uint64_t *origin, uint64_t stride, uint32_t n, uint32_t ppb, uint32_t row_bytes, uint32_t row_pitch_bytes) {
    typedef typename PxuPiece<N>::v V;
    typedef V VU __attribute__((aligned(sizeof(T))));        // rows start at any value-aligned address
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t j = lane; j < n; j += 64) {
        const uint32_t blk = j / ppb, part = j - blk * ppb;
        uint8_t *dst = img + origin[blk] * sizeof(T) + (uint64_t)part * N;
        const uint8_t *src = tile + (uint64_t)blk * row_bytes + (uint64_t)part * N;
#pragma unroll
        for (uint32_t y = 0; y < 4; y++) *(VU *)(dst + (uint64_t)y * stride * sizeof(T)) = *(const V *)(src + (uint64_t)y * row_pitch_bytes);
    }
}
// The wave's pixels: o[i] is the value of curve position i of the lane's unit.  tile: 16 x 64 values; origin: a slot per block.
template <typename T>
__device__ __forceinline__ void pxu_pixels(const DecArgs &a, uint8_t *tile8, uint64_t *origin, const T (&o)[16], bool act, uint32_t blk, uint32_t c,
                                           uint32_t g0, uint32_t nb_here, uint32_t SB) {
    const uint32_t lane = threadIdx.x & 63, B = a.g.bands, nbx = a.g.nbx;
    const uint64_t stride = a.g.stride, order = a.g.order;
    T *tile = (T *)tile8;
    const uint32_t RP = SB * 4 * B;                        // tile elements per pixel row
    if (lane < nb_here) {
        const uint32_t g = g0 + lane, by = g / nbx, bx = g - by * nbx;
        const uint32_t x0 = (4 * bx + 4 > a.g.w) ? a.g.w - 4 : 4 * bx;     // last column / row is shifted, not padded
        const uint32_t y0 = (4 * by + 4 > a.g.h) ? a.g.h - 4 : 4 * by;
        origin[lane] = (uint64_t)y0 * stride + (uint64_t)x0 * B;
    }
    if (act) {
        const uint32_t e0 = blk * 4 * B + c;
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) {
            const uint32_t nib = curve_nib(order, i);
            tile[e0 + (nib >> 2) * RP + (nib & 3) * B] = o[i];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t RB = 4 * B * (uint32_t)sizeof(T);        // bytes of a block's row: a multiple of 4
    uint8_t *img = (uint8_t *)a.img;
    if ((RB & 15) == 0) pxu_store_rows<16, T>(img, tile8, origin, stride, nb_here * (RB >> 4), RB >> 4, RB, RP * (uint32_t)sizeof(T));
    else if ((RB & 7) == 0) pxu_store_rows<8, T>(img, tile8, origin, stride, nb_here * (RB >> 3), RB >> 3, RB, RP * (uint32_t)sizeof(T));
    else pxu_store_rows<4, T>(img, tile8, origin, stride, nb_here * (RB >> 2), RB >> 2, RB, RP * (uint32_t)sizeof(T));
}
// the 8-bit rows of a raster whose rows are not dword aligned: value-aligned means byte-aligned there, which the vector stores take
// as they are (the hardware's unaligned access mode); nothing to do.

// bytes of LDS a wave needs: the staged words and, in their place afterwards, the tile; then a slot per block
// (PXU_PAD zero words behind the staged ones: what wide_values_lds reads beyond a position inside the staged bits, qb3_wide.h)
constexpr uint32_t PXU_PAD = WIDE_PAD_DW;
__host__ __device__ inline uint32_t pxu_wave_bytes(uint32_t in_cap_dw, uint32_t tsz) {
    const uint32_t st = 4 * (in_cap_dw + PXU_PAD), tl = 16 * 64 * tsz;
    return (((st > tl ? st : tl) + 15u) & ~15u) + 8 * 32;
}

template <typename T> __device__ __forceinline__ uint64_t pxu_le(const uint8_t *q, uint32_t n) {
    uint64_t v = 0;
    for (uint32_t i = 0; i < n; i++) v |= (uint64_t)q[i] << (8 * i);
    return v;
}

// ---- FTL / BASE.  BL: no index -- position, entering rungs and values and a twelve-bit length per unit come from the segment's
// entry of the container's restart table (level 2)
template <typename T, bool STEP, bool BL>
__global__ void __launch_bounds__(256) dec_pxu_kernel(const DecArgs a0) {
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    constexpr uint32_t UB = UBits<T>::v, UMASK = (1u << UB) - 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t chk = BL ? a0.chk_wgs : 0u;          // the launch's first workgroups check a chunk of the container's table each (ix_check_chunk)
    if (blockIdx.x < chk) { ix_check_chunk(a, blockIdx.x, (uint32_t *)smem); return; }
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const uint32_t B = a.g.bands, SB = a.g.seg_blocks;
    uint16_t *dtab = (uint16_t *)smem;                      // 2 KB
    uint8_t *wmem = smem + 2048 + (size_t)wave * pxu_wave_bytes(a.in_cap_dw, (uint32_t)sizeof(T));
    uint32_t *stage = (uint32_t *)wmem;
    uint64_t *origin = (uint64_t *)(wmem + pxu_wave_bytes(a.in_cap_dw, (uint32_t)sizeof(T)) - 8 * 32);

    const uint64_t seg = a.seg0 + (uint64_t)(blockIdx.x - chk) * nwaves + wave;
    const bool live = seg < a.seg_end;
    const uint64_t segc = live ? seg : 0;
    const uint32_t g0 = (uint32_t)(segc * SB), nblocks = (uint32_t)a.g.nblocks;
    const uint32_t nb_here = (nblocks - g0 < SB) ? nblocks - g0 : SB;
    const uint32_t blk = fastdiv(lane, B, a.magic_bands), c = lane - blk * B;
    const bool act = live && blk < nb_here;
    uint64_t P0, P1;
    uint32_t blen = 0, rg0 = 0;
    T pv0 = 0;
    if (BL) {
        const uint8_t *e = ix_entry_at(a.ix, a.ix_per_chunk, a.ix_E, a.ix_pad, (uint32_t)segc);
        P0 = pxu_le<T>(e, 6);
        P1 = (segc + 1 < a.g.nseg) ? pxu_le<T>(ix_entry_at(a.ix, a.ix_per_chunk, a.ix_E, a.ix_pad, (uint32_t)segc + 1), 6) : a.in_bits;
        if (act) {
            rg0 = e[6 + c] & UMASK;
            pv0 = (T)pxu_le<T>(e + 6 + B + c * sizeof(T), sizeof(T));
            const uint8_t *fl = e + 6 + B * (1 + sizeof(T)) + ((IX_BL_BITS_WIDE * lane) >> 3);
            blen = (((uint32_t)fl[0] | (uint32_t)fl[1] << 8) >> ((IX_BL_BITS_WIDE * lane) & 7)) & ((1u << IX_BL_BITS_WIDE) - 1);
        }
    } else {
        P0 = a.idx.bitpos[segc];
        P1 = (segc + 1 < a.g.nseg) ? a.idx.bitpos[segc + 1] : a.in_bits;
        if (act) {
            const uint64_t u = (uint64_t)g0 * B + lane;
            blen = sizeof(T) == 1 ? (uint32_t)((const uint8_t *)a.idx.ulen)[u] : (uint32_t)((const uint16_t *)a.idx.ulen)[u];
            rg0 = a.idx.rung[segc * B + c];
            pv0 = a.totals_only ? (T)0 : ((const T *)a.idx.prev)[segc * B + c];
        }
    }
    for (uint32_t i = tid; i < 128; i += blockDim.x) ((uint4 *)dtab)[i] = ((const uint4 *)wide_dec_tab.e)[i];
    __syncthreads();                                        // the only workgroup barrier
    if (!live) return;
    if (!BL && P1 < P0) P1 = P0;        // (the last segment of a truncated stream starts behind its end: it reads zeros)
    const uint64_t w0 = (a.in_bit0 + P0) >> 5;
    const uint64_t endw_abs = (a.in_bit0 + a.in_bits + 31) >> 5;
    const uint64_t ndw64 = ((a.in_bit0 + P1 + 31) >> 5) - w0;
    // positions out of a container's table are untrusted: a segment that does not lie inside the stream, or is longer than the
    // longest valid one, reads nothing (the staging may be sized for the stream's average segment: a longer -- but valid -- one
    // raises status bit 4 and the host runs the call again with the worst case)
    const bool sane = P0 <= P1 && (!BL || P1 <= a.in_bits);       // (a truncated stream reads as zeros behind its end, like the reference's: bitstream.h:36)
    const bool fits = sane && ndw64 <= a.in_cap_dw;
    const uint32_t misfit = (sane && ndw64 <= a.in_cap_full) ? 16u : 8u;
    const uint32_t ndw = fits ? (uint32_t)ndw64 : 0;
    for (uint32_t base = 0; base < ndw + PXU_PAD; base += 512) {  // eight loads in flight per lane, then eight LDS stores
        uint32_t sw[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t i = base + lane + 64 * k;
            sw[k] = (i < ndw && w0 + i < endw_abs) ? a.in32[w0 + i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t i = base + lane + 64 * k;
            if (i < ndw + PXU_PAD) stage[i] = sw[k];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const uint32_t limit = 32 * ndw;                        // no unit starts beyond the staged bits (PXU_PAD zero words follow)
    const uint32_t cpos = (uint32_t)(a.in_bit0 + P0 - 32 * w0);
    bool bad = !fits;
    const uint32_t binc = wave_iscan32(blen);               // inclusive: the last lane holds the bits of the segment
    uint32_t pos = cpos + binc - blen, gpos = 0;
    pos = pos < limit ? pos : limit;
    bool sig = false;
    const LdsWords sw = (LdsWords)stage;
    const uint32_t d = dec3_switch<T, LdsWords>(sw, ndw + PXU_PAD, pos, &gpos, &sig);
    if (act && sig && STEP) bad = true;                     // common-factor / index unit in a BASE stream: not handled here
    const uint32_t dd = act ? d : 0u;
    const uint32_t rung = (rg0 + pxu_exscan_band<uint32_t>(dd, B) + dd) & UMASK;
    T run[16];
    uint32_t end = 0;
    dec3_group<T, STEP, LdsWords, sizeof(T) >= 4>(sw, ndw + PXU_PAD, gpos, rung, dtab, run, &end);
    if (BL && act && end != pos + blen) bad = true;         // the table's lengths are not this stream's
    const T usum = act ? run[15] : (T)0;
    const T sex = pxu_exscan_band<T>(usum, B);
    if (bad) atomicOr(a.status, fits ? 1u : misfit);
    if (a.totals_only) {    // a plain stream, first pass: leave the segment's per-band sums where the entering values go (prev_scan_kernel)
        if (act && blk == nb_here - 1) ((T *)a.idx.prev)[seg * B + c] = (T)(sex + usum);
        return;
    }
    if (lane == 63 && seg == a.g.nseg - 1 && fits) {        // reference: more than 7 unused bits at the end is a failure
        const uint64_t used = (uint64_t)(cpos + binc) + 32 * w0 - a.in_bit0;
        if (used > a.in_bits) atomicOr(a.status, 4u);
        else if (a.in_bits - used > 7) atomicOr(a.status, 2u);
    }
    const T pv = (T)(pv0 + sex);
    T o[16];
#pragma unroll
    for (int i = 0; i < 16; i++) o[i] = (T)(run[i] + pv);
    const uint64_t nibs = pxu_band_nibbles(a0);
    bool mapped = false;
    for (uint32_t k = 0; k < B; k++) mapped = mapped || ((nibs >> (4 * k)) & 15u) != k;
    if (mapped) {           // the core band back on (core bands are themselves core: their lanes hold final values)
        const uint32_t cb = (uint32_t)(nibs >> (4 * c)) & 15u, src = lane - c + cb;
#pragma unroll
        for (int i = 0; i < 16; i++) { const T v = pxu_shfl<T>(o[i], src); if (cb != c) o[i] = (T)(o[i] + v); }
    }
    pxu_pixels<T>(a, wmem, origin, o, act, blk, c, g0, nb_here, SB);
}

// ---- the common-factor modes (reference decode<T>, QB3decode.h:578-741: normal units :619-623, common-factor units :629-679, index
// units :680-715).  As dec_pxw_best_kernel: the index (or, BL, the container's table: three bytes a unit) holds a DWORD PER UNIT -- its
// bits | the rung it is entered with << 16 -- and per segment and band the factor in force; a lane parses its unit assuming the
// segment's entering factor of its band; the band's units that brought a factor are found by ballot, the nearest one below the lane
// gives the factor in force, and only a lane that assumed wrongly parses again.
template <typename T, bool BL>
__global__ void __launch_bounds__(256) dec_pxu_best_kernel(const DecArgs a0) {
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    constexpr uint32_t UB = UBits<T>::v, UMASK = (1u << UB) - 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t chk = BL ? a0.chk_wgs : 0u;          // the launch's first workgroups check a chunk of the container's table each (ix_check_chunk)
    if (blockIdx.x < chk) { ix_check_chunk(a, blockIdx.x, (uint32_t *)smem); return; }
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const uint32_t B = a.g.bands, SB = a.g.seg_blocks;
    uint8_t *wmem = smem + (size_t)wave * pxu_wave_bytes(a.in_cap_dw, (uint32_t)sizeof(T));      // nothing is shared between the waves: no barrier
    uint32_t *stage = (uint32_t *)wmem;
    uint64_t *origin = (uint64_t *)(wmem + pxu_wave_bytes(a.in_cap_dw, (uint32_t)sizeof(T)) - 8 * 32);
    const uint64_t seg = a.seg0 + (uint64_t)(blockIdx.x - chk) * nwaves + wave;
    if (seg >= a.seg_end) return;
    const uint32_t g0 = (uint32_t)(seg * SB), nblocks = (uint32_t)a.g.nblocks;
    const uint32_t nb_here = (nblocks - g0 < SB) ? nblocks - g0 : SB;
    const uint32_t blk = fastdiv(lane, B, a.magic_bands), c = lane - blk * B;
    const bool act = blk < nb_here;
    uint64_t P0, P1;
    uint32_t bt = 0;                                    // the unit's bits | entering rung << 16
    T pv0 = 0, cf0 = 0;
    if (BL) {
        const uint8_t *e = ix_entry_at(a.ix, a.ix_per_chunk, a.ix_E, a.ix_pad, (uint32_t)seg);
        P0 = pxu_le<T>(e, 6);
        P1 = (seg + 1 < a.g.nseg) ? pxu_le<T>(ix_entry_at(a.ix, a.ix_per_chunk, a.ix_E, a.ix_pad, (uint32_t)seg + 1), 6) : a.in_bits;
        if (act) {
            pv0 = (T)pxu_le<T>(e + 6 + B + c * sizeof(T), sizeof(T));
            cf0 = (T)pxu_le<T>(e + 6 + B + (B + c) * sizeof(T), sizeof(T));
            const uint8_t *fp = e + 6 + B * (1 + 2 * sizeof(T)) + IX_BL_BEST_BYTES * lane;
            const uint32_t fld = (uint32_t)fp[0] | (uint32_t)fp[1] << 8 | (uint32_t)fp[2] << 16;
            bt = (fld & 0xfffu) | ((fld >> 12) & UMASK) << 16;
        }
    } else {
        P0 = a.idx.bitpos[seg];
        P1 = (seg + 1 < a.g.nseg) ? a.idx.bitpos[seg + 1] : a.in_bits;
        if (act) {
            bt = ((const uint32_t *)a.idx.ulen)[(uint64_t)g0 * B + lane];
            pv0 = a.totals_only ? (T)0 : ((const T *)a.idx.prev)[seg * B + c];
            cf0 = ((const T *)a.idx.cf)[seg * B + c];
        }
    }
    if (!BL && P1 < P0) P1 = P0;        // (the last segment of a truncated stream starts behind its end: it reads zeros)
    const uint64_t w0 = (a.in_bit0 + P0) >> 5;
    const uint64_t endw_abs = (a.in_bit0 + a.in_bits + 31) >> 5;
    const uint64_t ndw64 = ((a.in_bit0 + P1 + 31) >> 5) - w0;
    const bool sane = P0 <= P1 && (!BL || P1 <= a.in_bits);       // (a truncated stream reads as zeros behind its end, like the reference's: bitstream.h:36)
    const bool fits = sane && ndw64 <= a.in_cap_dw;
    const uint32_t misfit = (sane && ndw64 <= a.in_cap_full) ? 16u : 8u;      // 16: the staging was sized for the stream's average; the host calls again with the worst case
    const uint32_t ndw = fits ? (uint32_t)ndw64 : 0;
    for (uint32_t base = 0; base < ndw + PXU_PAD; base += 512) {
        uint32_t sw[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t i = base + lane + 64 * k;
            sw[k] = (i < ndw && w0 + i < endw_abs) ? a.in32[w0 + i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t i = base + lane + 64 * k;
            if (i < ndw + PXU_PAD) stage[i] = sw[k];
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
    // the rung the band's NEXT unit is entered with is the rung this unit must leave: checked, not trusted
    const uint32_t nxt = (uint32_t)__shfl_down((unsigned)(bt >> 16), B, 64);
    uint64_t period = 0;                                            // lanes 0, B, 2B, ...: shifted by its band, a lane's band
    for (uint32_t k = 0; k < 64; k += B) period |= 1ull << k;
    T g[16], pcf = cf0, cf_in = cf0;
#pragma unroll
    for (int i = 0; i < 16; i++) g[i] = 0;
    uint32_t rung = oldrung, flags = 0, end = pos;
    bool ok = true, need = act;
#pragma nounroll
    for (int pass = 0; pass < 2; pass++) {
        if (need) {
            ReaderT<LdsWords> rd;
            rd.init((LdsWords)stage, pos, 32ull * (ndw + PXU_PAD));
            rung = oldrung; pcf = cf_in; flags = 0;
            ok = parse_unit<T, CM_BEST, ReaderT<LdsWords>>(rd, rung, pcf, g, &flags);
            end = (uint32_t)rd.position();
        }
        if (pass) break;
        // the factor in force for a unit that takes the band's: the nearest lane of the band below whose unit brought one
        const uint64_t wm = __ballot(act && (flags & 2u));
        if (!wm) break;                                             // (no writer in the segment: every lane assumed right)
        const uint64_t below = wm & (period << c) & ((1ull << lane) - 1);
        const uint32_t src = below ? 63u - (uint32_t)__clzll((long long)below) : lane;
        const T got = pxu_shfl<T>(pcf, src);
        need = act && (flags & 1u) && below && got != cf0;
        cf_in = got;
        if (!__any(need)) break;
    }
    if (act && (!ok || end != pos + blen)) bad = true;              // malformed unit, or the lengths are not this stream's
    if (act && blk + 1 < nb_here && rung != (nxt & UMASK)) bad = true;
    T run[16], acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { acc = (T)(acc + smag_t<T>(g[i])); run[i] = acc; }
    const T usum = act ? acc : (T)0;
    const T sex = pxu_exscan_band<T>(usum, B);
    if (bad) atomicOr(a.status, fits ? 1u : misfit);
    if (a.totals_only) {    // a plain stream, first pass: the segment's per-band sums (prev_scan_kernel makes entering values of them)
        if (act && blk == nb_here - 1) ((T *)a.idx.prev)[seg * B + c] = (T)(sex + usum);
        return;
    }
    if (lane == 63 && seg == a.g.nseg - 1 && fits) {
        const uint64_t used = (uint64_t)(cpos + binc) + 32 * w0 - a.in_bit0;
        if (used > a.in_bits) atomicOr(a.status, 4u);
        else if (a.in_bits - used > 7) atomicOr(a.status, 2u);
    }
    const T pv = (T)(pv0 + sex);
    T o[16];
#pragma unroll
    for (int i = 0; i < 16; i++) o[i] = (T)(run[i] + pv);
    const uint64_t nibs = pxu_band_nibbles(a0);
    bool mapped = false;
    for (uint32_t k = 0; k < B; k++) mapped = mapped || ((nibs >> (4 * k)) & 15u) != k;
    if (mapped) {
        const uint32_t cb = (uint32_t)(nibs >> (4 * c)) & 15u, src = lane - c + cb;
#pragma unroll
        for (int i = 0; i < 16; i++) { const T v = pxu_shfl<T>(o[i], src); if (cb != c) o[i] = (T)(o[i] + v); }
    }
    pxu_pixels<T>(a, wmem, origin, o, act, blk, c, g0, nb_here, SB);
}

size_t pxu_lds_bytes(uint32_t in_cap_dw, uint32_t tsz, bool best) { return (best ? 0 : 2048) + 4 * (size_t)pxu_wave_bytes(in_cap_dw, tsz); }

template <typename T>
static void launch_dec_pxu_t(const DecArgs &a, hipStream_t st) {
    const bool step = a.g.mode != CM_FTL;
    dim3 grid((uint32_t)((a.seg_end - a.seg0 + 3) / 4) + (a.bl_mode ? a.chk_wgs : 0u), a.ntiles), block(256);
    const size_t lds = pxu_lds_bytes(a.in_cap_dw, a.g.tsz, false);
    if (a.bl_mode) {
        if (step) hipLaunchKernelGGL((dec_pxu_kernel<T, true, true>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((dec_pxu_kernel<T, false, true>), grid, block, lds, st, a);
        return;
    }
    if (step) hipLaunchKernelGGL((dec_pxu_kernel<T, true, false>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((dec_pxu_kernel<T, false, false>), grid, block, lds, st, a);
}
void launch_dec_pxu(const DecArgs &a, const DecPlan &plan, hipStream_t st) {
    (void)plan;
    switch (a.g.tsz) {
    case 1: launch_dec_pxu_t<uint8_t>(a, st); break;
    case 2: launch_dec_pxu_t<uint16_t>(a, st); break;
    case 4: launch_dec_pxu_t<uint32_t>(a, st); break;
    default: launch_dec_pxu_t<uint64_t>(a, st); break;
    }
}
template <typename T>
static void launch_dec_pxu_best_t(const DecArgs &a, hipStream_t st) {
    dim3 grid((uint32_t)((a.seg_end - a.seg0 + 3) / 4) + (a.bl_mode ? a.chk_wgs : 0u), a.ntiles), block(256);
    const size_t lds = pxu_lds_bytes(a.in_cap_dw, a.g.tsz, true);
    if (a.bl_mode) hipLaunchKernelGGL((dec_pxu_best_kernel<T, true>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((dec_pxu_best_kernel<T, false>), grid, block, lds, st, a);
}
void launch_dec_pxu_best(const DecArgs &a, const DecPlan &plan, hipStream_t st) {
    (void)plan;
    switch (a.g.tsz) {
    case 1: launch_dec_pxu_best_t<uint8_t>(a, st); break;
    case 2: launch_dec_pxu_best_t<uint16_t>(a, st); break;
    case 4: launch_dec_pxu_best_t<uint32_t>(a, st); break;
    default: launch_dec_pxu_best_t<uint64_t>(a, st); break;
    }
}

}  // namespace qb3dev
