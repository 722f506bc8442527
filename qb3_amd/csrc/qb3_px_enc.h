// qb3_amd/csrc/qb3_px_enc.h -- front end of the 8-bit lane-per-block encoders (k_enc_px.hip: FTL/BASE, k_enc_px_best.hip:
// the common-factor modes): block load, gather, band difference, running delta, mag-sign, rungs, and the plain
// (reference groupencode, QB3encode.h:155-280) bit string of a unit.
#pragma once
#include "qb3_px.h"

namespace qb3dev {

// The block of lane `tid` of a chunk (4 rows x B dwords) and the dword holding the previous block's last visited pixel.
template <int B, uint64_t ORDER>
__device__ __forceinline__ void px_load_block(const EncArgs &a, bool valid, uint32_t gblk, uint32_t (&w)[4][B], uint32_t &pd) {
    const uint32_t nbx = a.g.nbx;
    const uint64_t stride = a.g.stride;
    pd = 0;
    constexpr uint32_t n15 = order_nib(ORDER, 15);
    // Rows need not be dword aligned (odd widths and strides, the shifted last column, any pointer): a row is read as the
    // aligned dwords that cover it -- one more than it has when it is not aligned -- and funnel-shifted into place.
    // Nothing is read beyond the aligned dword that holds the row's last byte.
    auto load_row = [&](const uint8_t *p, uint32_t (&row)[B]) {
        const uint32_t sh = 8 * ((uint32_t)(uintptr_t)p & 3);
        const uint32_t *q = (const uint32_t *)((uintptr_t)p & ~(uintptr_t)3);
        uint32_t d[B + 1];
#pragma unroll
        for (int t = 0; t < B; t++) d[t] = q[t];
        d[B] = sh ? q[B] : 0u;
#pragma unroll
        for (int t = 0; t < B; t++) row[t] = __builtin_amdgcn_alignbit(d[t + 1], d[t], sh);
    };
    if (valid) {
        const uint32_t by = gblk / nbx, bx = gblk - by * nbx;
        const uint32_t x0 = (4 * bx + 4 > a.g.w) ? a.g.w - 4 : 4 * bx;     // last column / row is shifted, not padded
        const uint32_t y0 = (4 * by + 4 > a.g.h) ? a.g.h - 4 : 4 * by;
        const uint8_t *p0 = (const uint8_t *)a.img + (uint64_t)y0 * stride + (uint64_t)x0 * B;
        const uint8_t *pp = nullptr;      // the four bytes that end the previous block's row holding its last visited pixel
        if (gblk) {
            const uint32_t pb = gblk - 1, pby = pb / nbx, pbx = pb - pby * nbx;
            const uint32_t px0 = (4 * pbx + 4 > a.g.w) ? a.g.w - 4 : 4 * pbx;
            const uint32_t py0 = (4 * pby + 4 > a.g.h) ? a.g.h - 4 : 4 * pby;
            pp = (const uint8_t *)a.img + (uint64_t)(py0 + (n15 >> 2)) * stride + (uint64_t)px0 * B + 4 * (B - 1);
        }
        if (a.px_aligned) {             // workgroup uniform: width, stride and pointer are multiples of 4
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t *rp = (const uint32_t *)(p0 + (uint64_t)r * stride);
#pragma unroll
                for (int t = 0; t < B; t++) w[r][t] = rp[t];
            }
            if (gblk) pd = *(const uint32_t *)pp;
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) load_row(p0 + (uint64_t)r * stride, w[r]);
            if (gblk) {
                const uint32_t sh = 8 * ((uint32_t)(uintptr_t)pp & 3);
                const uint32_t *q = (const uint32_t *)((uintptr_t)pp & ~(uintptr_t)3);
                pd = __builtin_amdgcn_alignbit(sh ? q[1] : 0u, q[0], sh);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int k = 0; k < B; k++) w[r][k] = 0;
    }

}

// What a lane knows about its block before any coding decision: per band the sixteen mag-sign deltas in curve order (four
// to a register), their OR, the value the curve leaves the block with, the value it entered with; the rungs of this block and of
// the block before it, four bits a band.  Contains ONE workgroup barrier, under which the code table goes to LDS.
template <int B> struct PxFront {
    uint32_t gp[B][4], usedv[B], lastv[B], pvv[B];
    uint32_t rp_packed, prp;
};
template <int B, bool RGB, uint64_t ORDER>
__device__ __forceinline__ void px_front(const EncArgs &a0, uint32_t gblk, const uint32_t (&w)[4][B], uint32_t pd, uint32_t *etab, uint32_t *wsum,
                                         const uint4 &tabv, PxFront<B> &f) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- per band: bytes in curve order, band difference, running delta, mag-sign -- four values per register
    uint32_t cur[B][4];
#pragma unroll
    for (int c = 0; c < B; c++)
#pragma unroll
        for (int q = 0; q < 4; q++) cur[c][q] = gather_quad<B, ORDER>(w, q, c);
    uint32_t rp_packed = 0;
#pragma unroll
    for (int c = 0; c < B; c++) {
        const int cb = core_of<B, RGB>(c);
        uint32_t prv;
        if (gblk == 0) prv = (uint32_t)a0.st.prev[c] & 0xffu;
        else {      // pixel x = 3 of the previous block sits in the last dword of its row: byte c + 4 - B
            prv = (pd >> (8 * (c + 4 - B))) & 0xffu;
            if (cb != c) prv = (prv - ((pd >> (8 * (cb + 4 - B))) & 0xffu)) & 0xffu;
        }
        f.pvv[c] = prv;
        uint32_t x[4];
#pragma unroll
        for (int q = 0; q < 4; q++) x[q] = (cb != c) ? swar_sub8(cur[c][q], cur[cb][q]) : cur[c][q];
        uint32_t u = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t before = q ? __builtin_amdgcn_alignbit(x[q], x[q - 1], 24) : ((x[0] << 8) | prv);
            f.gp[c][q] = swar_mags8(swar_sub8(x[q], before));
            u |= f.gp[c][q];
        }
        u |= u >> 16; u |= u >> 8; u &= 0xffu;
        f.usedv[c] = u; f.lastv[c] = x[3] >> 24;
        rp_packed |= topbit32(u | 1) << (4 * c);
    }
    // rungs of the previous block: neighbouring lane, or the last lane of the previous wave through LDS
    uint32_t prp = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rp_packed, 0x138, 0xf, 0xf, false);      // wave_shr:1
    if (lane == 63) wsum[32 + wave] = rp_packed;
    if (tid < 128) ((uint4 *)etab)[tid] = tabv;
    __syncthreads();
    if (lane == 0 && wave) prp = wsum[32 + wave - 1];
    if (gblk == 0) { prp = 0;
#pragma unroll
        for (int c = 0; c < B; c++) prp |= ((uint32_t)a0.st.rung[c] & 15u) << (4 * c); }
    f.rp_packed = rp_packed; f.prp = prp;
}

// The plain bit string of a unit with used > 1 (rung >= 1): the switch code (csc, csl bits) and the sixteen value codes as six
// pieces of at most 27 bits, each built backwards with one shift-or per value; pl = piece length.  Returns the unit's bits.
// tb: byte address of the rung's region of the code table in LDS.
template <bool STEP>
__device__ __forceinline__ uint32_t px_unit_pieces(const uint32_t (&gq)[4], uint32_t rung, uint32_t csl, uint32_t csc, uint32_t tb,
                                                   uint32_t (&pc)[6], uint32_t (&pl)[6]) {
    uint32_t g4[4] = {gq[0], gq[1], gq[2], gq[3]};
    if (STEP) {     // clear the rung bit of the last value of a 1..10..0 rung-bit run (reference QB3encode.h:169-176)
        uint32_t bits = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) bits |= ((g4[i >> 2] >> (8 * (i & 3) + rung)) & 1u) << i;
        if ((bits & (bits + 1)) == 0) {
            const uint32_t n = __popc(bits) - 1;        // index of the value to change
#pragma unroll
            for (int q = 0; q < 4; q++) if ((n >> 2) == (uint32_t)q) g4[q] ^= (1u << rung) << (8 * (n & 3));
        }
    }
    constexpr int first[7] = {0, 2, 5, 8, 11, 14, 16};  // piece k holds values first[k] .. first[k+1]-1
    uint32_t lsum = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        uint32_t acc = 0, s = 0;
#pragma unroll
        for (int i = first[k + 1] - 1; i >= first[k]; i--) {
            const uint32_t m = (g4[i >> 2] >> (8 * (i & 3))) & 0xffu;
            const uint32_t e = *lds_at((m << 2) + tb);
            acc = __builtin_amdgcn_alignbit(acc, e, e);      // (acc << length) | code: the entry's low five bits are 32 - length
            s += e;
        }
        uint32_t len = 32u * (uint32_t)(first[k + 1] - first[k]) - (s & 0xffu);
        if (k == 0) { acc = (acc << csl) | csc; len += csl; }
        pc[k] = acc; pl[k] = len; lsum += len;
    }
    return lsum;
}
// ... and of a unit with used <= 1: switch, the "not all zero" flag, then the 16 bits (split so that no piece exceeds 27 bits)
__device__ __forceinline__ uint32_t px_unit_low(const uint32_t (&gq)[4], uint32_t used, uint32_t csl, uint32_t csc, uint32_t (&pc)[6], uint32_t (&pl)[6]) {
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) bits |= ((gq[i >> 2] >> (8 * (i & 3))) & 1u) << i;
    pc[0] = csc | (used << csl); pl[0] = csl + 1;
    pc[1] = bits; pl[1] = used ? 16 : 0;
    return pl[0] + pl[1];
}

}  // namespace qb3dev
