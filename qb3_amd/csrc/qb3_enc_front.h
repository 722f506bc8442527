// qb3_amd/csrc/qb3_enc_front.h -- front end of the unit-per-lane encoder kernels (k_enc_generic.hip, k_enc_best.hip)
#pragma once
#include "qb3_kernels.h"
#include "qb3_wide.h"

namespace qb3dev {

// Everything a unit-per-lane encoder kernel needs before coding: LDS carve, tile staging, gather, deltas.
template <typename T> struct EncFront {
    uint64_t *slot_base; uint32_t *tile, *wsum; uint8_t *rungs; uint16_t *etab; uint32_t *outbuf;
    uint8_t *board;         // the LDS behind the front end's own (the common-factor kernels' writer board)
    uint32_t s, c, cb, gblk, rung, nbp, chunk;
    bool valid, payload;
    T used, pv, lastv;
    uint32_t prung;         // rung of the block before (the lane-per-block front end; the generic one leaves the rungs in LDS)
};

// dwords the tile (4 rows of rowdw) and the bit buffer (outdw) share
__host__ __device__ inline uint32_t enc_front_union_dw(uint32_t rowdw, uint32_t outdw) {
    const uint32_t t = 4 * rowdw, o = (outdw + 1) & ~1u;
    return t > o ? t : o;
}
template <typename T>
__device__ __forceinline__ void enc_front(const EncArgs &a, const EncArgs &a0, uint8_t *smem, uint32_t outdw, EncFront<T> &f, T (&g)[16], uint32_t chunk) {
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    const uint32_t bands = a.g.bands, slots = a.slots, dpr = a.dpr, nbp = slots - 1;
    const uint32_t nblocks = (uint32_t)a.g.nblocks, nbx = a.g.nbx;
    const uint64_t stride = a.g.stride;
    const uint32_t rowdw = slots * dpr;

    // LDS carve (all offsets multiples of 8).  The pixel tile and the bit buffer share their memory: the tile is dead once the units are
    // gathered, the buffer is zeroed behind that barrier and written after the callers' scan -- 16 + 17 KB a workgroup of two int32
    // bands became 17, twice the workgroups a CU holds (enc_front_lds_dw: the plan's arithmetic)
    f.slot_base = (uint64_t *)smem;
    f.tile = (uint32_t *)(f.slot_base + slots);
    f.outbuf = f.tile;
    f.wsum = f.tile + enc_front_union_dw(rowdw, outdw);   // 64 dwords of scan scratch
    f.rungs = (uint8_t *)(f.wsum + 64);                   // slots*bands bytes, padded to 8
    f.etab = (uint16_t *)(f.rungs + ((slots * bands + 7) & ~7u));     // ENC_TAB_SIZE + pad
    f.board = (uint8_t *)(f.etab + 512);
    fill_enc_tab(f.etab);
    uint64_t *slot_base = f.slot_base; uint32_t *tile = f.tile;

    const uint32_t g0 = chunk * nbp;                      // first payload block of this chunk
    f.nbp = nbp; f.chunk = chunk;

    // block index of slot s is g0 - 1 + s; invalid slots are clamped to a valid block so loads stay in bounds
    auto slot_block = [&](uint32_t s, bool &valid) -> uint32_t {
        const int64_t g = (int64_t)g0 - 1 + s;
        valid = g >= 0 && g < (int64_t)nblocks;
        return g < 0 ? 0u : (g >= (int64_t)nblocks ? nblocks - 1 : (uint32_t)g);
    };
    auto block_origin = [&](uint32_t g, uint32_t &x0, uint32_t &y0) {
        const uint32_t by = g / nbx, bx = g - by * nbx;
        x0 = (4 * bx + 4 > a.g.w) ? a.g.w - 4 : 4 * bx;     // last column / row is shifted, not padded
        y0 = (4 * by + 4 > a.g.h) ? a.g.h - 4 : 4 * by;     // (reference QB3encode.h:410-416)
    };

    if (tid < slots) {
        bool valid; uint32_t x0, y0;
        block_origin(slot_block(tid, valid), x0, y0);
        slot_base[tid] = (uint64_t)y0 * stride + (uint64_t)x0 * bands;
    }
    __syncthreads();

    // ---- stage the 4-row tile: coalesced dword loads, [row][slot][pixel][band] as in memory
    const uint8_t *imgb = (const uint8_t *)a.img;
    // (four loads in flight per thread before the LDS stores: a load-store-load-store loop pays one memory round trip
    // per element)
    for (uint32_t j0 = 0; j0 < rowdw; j0 += nthr) {
        const uint32_t j = j0 + tid;
        const bool in = j < rowdw;
        const uint32_t s = fastdiv(in ? j : 0, dpr, a.magic_dpr), d = (in ? j : 0) - s * dpr;
        const uint8_t *p0 = imgb + slot_base[s] * sizeof(T) + 4 * d;
        uint32_t v[4];
#pragma unroll
        for (uint32_t r = 0; r < 4; r++) {
            const uint8_t *p = p0 + (uint64_t)r * stride * sizeof(T);
            v[r] = 0;
            if (in) {
                if (((uintptr_t)p & 3) == 0) v[r] = *(const uint32_t *)p;
                else v[r] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
            }
        }
#pragma unroll
        for (uint32_t r = 0; r < 4; r++) if (in) tile[r * rowdw + j] = v[r];
    }
    __syncthreads();

    // ---- per unit: gather in curve order, band difference, delta, mag-sign, rung
    const uint32_t s = fastdiv(tid, bands, a.magic_bands), c = tid - s * bands;
    bool valid = false;
    uint32_t gblk = 0;
    if (s < slots) gblk = slot_block(s, valid);
    const uint32_t cb = a0.g.cband[c < MAXBANDS ? c : 0];
    const T *tt = (const T *)tile;
    const uint64_t order = a.g.order;
    T used = 0, pv = 0, lastv = 0;
    uint32_t rung = 0;
    if (valid) {
        // value entering the unit: last visited pixel of the previous block, or the carried state
        const uint32_t n15 = curve_nib(order, 15);
        if (gblk == 0) pv = (T)a0.st.prev[c];
        else if (s >= 1) {
            const uint32_t e = (((n15 >> 2) * slots + (s - 1)) * 4 + (n15 & 3)) * bands;
            pv = tt[e + c];
            if (cb != c) pv = (T)(pv - tt[e + cb]);
        } else {            // halo unit: its predecessor block is not in the tile
            uint32_t x0, y0;
            block_origin(gblk - 1, x0, y0);
            const T *ip = (const T *)a.img + (uint64_t)(y0 + (n15 >> 2)) * stride + (uint64_t)(x0 + (n15 & 3)) * bands;
            pv = ip[c];
            if (cb != c) pv = (T)(pv - ip[cb]);
        }
        T prv = pv;
        const T cbmask = (cb != c) ? (T)~(T)0 : (T)0;
        // element index = lane part (slot, band) + a wave-uniform part per curve position (scalar arithmetic)
        const uint32_t ebase = s * 4 * bands, rowel = slots * 4 * bands;
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) {
            const uint32_t nib = curve_nib(order, i);
            const uint32_t e = ebase + ((nib >> 2) * rowel + (nib & 3) * bands);
            const T v = (T)(tt[e + c] - (tt[e + cb] & cbmask));    // core band read always: no branch, same-address LDS reads broadcast
            g[i] = mags_t<T>((T)(v - prv));
            used |= g[i];
            prv = v;
        }
        lastv = prv;
        rung = topbit_t<T>(used);
        f.rungs[tid] = (uint8_t)rung;
    }
    __syncthreads();
    // the tile has been read: its memory becomes the (zeroed) bit buffer; the callers' scans put a barrier in front of the first write
    for (uint32_t i = tid; i < outdw; i += nthr) f.outbuf[i] = 0;
    f.s = s; f.c = c; f.cb = cb; f.gblk = gblk; f.rung = rung;
    f.valid = valid; f.payload = valid && s >= 1;
    f.used = used; f.pv = pv; f.lastv = lastv;
}

// The same for 32/64-bit rasters of ONE band, lane per block, the block in registers (the front half of enc_pxw_kernel,
// k_enc_pxw.hip): four 16-byte row loads straight from HBM, the curve as register renaming, the rung of the block before by a
// DPP wave shift.  A chunk is the same run of blocks the generic front end gives a workgroup (slots = threads for one band),
// so everything behind the front end -- scans across chunks, index, table -- is shared.  No LDS tile: the carve is scan
// scratch, the code table (a compile-time constant copied from L2), the bit buffer.
template <typename T, uint64_t ORDER>
__device__ __forceinline__ void pxw_front(const EncArgs &a, const EncArgs &a0, uint8_t *smem, uint32_t outdw, EncFront<T> &f, T (&g)[16], uint32_t chunk) {
    const uint32_t tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nblocks = (uint32_t)a.g.nblocks, nbx = a.g.nbx;
    const uint64_t stride = a.g.stride;
    f.slot_base = nullptr; f.tile = nullptr; f.rungs = nullptr;
    f.wsum = (uint32_t *)smem;                              // 64 dwords of scan scratch ([32 ..]: rung of each wave's last lane)
    f.etab = (uint16_t *)(f.wsum + 64);                     // 512 entries
    f.outbuf = (uint32_t *)(f.etab + 512);
    f.board = (uint8_t *)(f.outbuf + ((outdw + 1) & ~1u));
    const uint4 tabv = ((const uint4 *)wide_enc_tab.e)[tid & 63];
    for (uint32_t i = tid; i < outdw; i += nthr) f.outbuf[i] = 0;
    const int64_t gs = (int64_t)chunk * (nthr - 1) - 1 + tid;       // lane 0 is the halo block
    const bool valid = gs >= 0 && gs < (int64_t)nblocks;
    const uint32_t gblk = valid ? (uint32_t)gs : 0u;
    T w[4][4], pv = 0;
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int x = 0; x < 4; x++) w[r][x] = 0;
    constexpr uint32_t n15 = (uint32_t)(ORDER & 15);
    if (valid) {
        const uint32_t by = gblk / nbx, bx = gblk - by * nbx;
        const uint32_t x0 = (4 * bx + 4 > a.g.w) ? a.g.w - 4 : 4 * bx;
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
    T used = 0, prv = pv;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        constexpr uint64_t O = ORDER;
        const uint32_t nib = (uint32_t)(O >> (60 - 4 * i)) & 15u;
        const T v = w[nib >> 2][nib & 3];
        g[i] = mags_t<T>((T)(v - prv));
        used |= g[i];
        prv = v;
    }
    const uint32_t rung = valid ? topbit_t<T>(used) : 0u;
    uint32_t prung = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rung, 0x138, 0xf, 0xf, false);      // wave_shr:1
    if (lane == 63) f.wsum[32 + wave] = rung;
    if (tid < 64) ((uint4 *)f.etab)[tid] = tabv;
    __syncthreads();
    if (lane == 0 && wave) prung = f.wsum[32 + wave - 1];
    f.s = tid; f.c = 0; f.cb = 0; f.gblk = gblk; f.rung = rung; f.nbp = nthr - 1; f.chunk = chunk;
    f.valid = valid; f.payload = valid && tid >= 1;
    f.used = used; f.pv = pv; f.lastv = prv; f.prung = prung;
}

}  // namespace qb3dev
