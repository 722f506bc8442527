// qb3_amd/csrc/k_enc_px.hip -- 8-bit grey / RGB / RGBA encoder, lane per block
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

// Codes the chunk whose blocks the lanes hold (lane 0: the halo block) into the LDS bit buffer, from bit 0; the
// buffer must be zero.  total: bits of the chunk; pos: where the lane's block starts.
template <int B, bool RGB, uint64_t ORDER, bool STEP>
__device__ __forceinline__ void px_code_chunk(const EncArgs &a, const EncArgs &a0, uint32_t chunk, bool valid, bool payload, uint32_t gblk,
                                              const uint32_t (&w)[4][B], uint32_t pd, uint32_t *etab, uint32_t *wsum, uint32_t *outbuf,
                                              uint32_t etab_off, const uint4 &tabv, uint32_t &total, uint32_t &pos) {
    constexpr uint32_t UMASK = 7;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nblocks = (uint32_t)a.g.nblocks;
    // ---- per band: bytes in curve order, band difference, running delta, mag-sign -- four values per register
    uint32_t cur[B][4];
#pragma unroll
    for (int c = 0; c < B; c++)
#pragma unroll
        for (int q = 0; q < 4; q++) cur[c][q] = gather_quad<B, ORDER>(w, q, c);
    uint32_t gp[B][4], usedv[B], lastv[B], pvv[B];
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
        pvv[c] = prv;
        uint32_t x[4];
#pragma unroll
        for (int q = 0; q < 4; q++) x[q] = (cb != c) ? swar_sub8(cur[c][q], cur[cb][q]) : cur[c][q];
        uint32_t u = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t before = q ? __builtin_amdgcn_alignbit(x[q], x[q - 1], 24) : ((x[0] << 8) | prv);
            gp[c][q] = swar_mags8(swar_sub8(x[q], before));
            u |= gp[c][q];
        }
        u |= u >> 16; u |= u >> 8; u &= 0xffu;
        usedv[c] = u; lastv[c] = x[3] >> 24;
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

    // ---- per band: the unit's bit string as six pieces of at most 27 bits; pl = piece length (low byte)
    uint32_t pc[B][6], pl[B][6], lens[B], blen[1] = { 0 };
#pragma unroll
    for (int c = 0; c < B; c++) {
#pragma unroll
        for (int k = 0; k < 6; k++) { pc[c][k] = 0; pl[c][k] = 0; }
        lens[c] = 0;
        if (payload) {
            const uint32_t rung = (rp_packed >> (4 * c)) & 15u, prung = (prp >> (4 * c)) & 15u, used = usedv[c];
            const uint32_t delta = (rung - prung) & UMASK;
            const uint32_t csl = __builtin_amdgcn_ubfe(cs3_lens(), 4 * delta, 4), csc = (uint32_t)(cs3_codes() >> (8 * delta)) & 0xffu;
            if (used <= 1) {
                uint32_t bits = 0;
#pragma unroll
                for (int i = 0; i < 16; i++) bits |= ((gp[c][i >> 2] >> (8 * (i & 3))) & 1u) << i;
                // switch, the "not all zero" flag, then the 16 bits: split so that no piece exceeds 27 bits
                pc[c][0] = csc | (used << csl); pl[c][0] = csl + 1;
                pc[c][1] = bits; pl[c][1] = used ? 16 : 0;
                lens[c] = pl[c][0] + pl[c][1];
            } else {
                uint32_t g4[4] = {gp[c][0], gp[c][1], gp[c][2], gp[c][3]};
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
                const uint32_t tb = etab_off + (8u << rung);         // byte address of the rung's table region
                constexpr int first[7] = {0, 2, 5, 8, 11, 14, 16};  // piece k holds values first[k] .. first[k+1]-1
                uint32_t lsum = 0;
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    uint32_t acc = 0, s = 0;
#pragma unroll
                    for (int i = first[k + 1] - 1; i >= first[k]; i--) {
                        const uint32_t m = (g4[i >> 2] >> (8 * (i & 3))) & 0xffu;
                        const uint32_t e = *lds_at((m << 2) + tb);
                        acc = (acc << (e & 31u)) | (e >> 8);
                        s += e;
                    }
                    if (k == 0) { acc = (acc << csl) | csc; s += csl; }
                    pc[c][k] = acc; pl[c][k] = s & 0xffu; lsum += s & 0xffu;
                }
                lens[c] = lsum;
            }
            blen[0] += lens[c];
        }
    }
    block_exscan_dpp<1>(blen, wsum);
    pos = blen[0]; total = wsum[0] + wsum[1] + wsum[2] + wsum[3];

    if (payload) {
        LdsWriter32 wr;
        wr.init(outbuf, pos);
#pragma unroll
        for (int c = 0; c < B; c++)
#pragma unroll
            for (int k = 0; k < 6; k++) wr.put(pc[c][k], pl[c][k]);
        wr.finish();
        if (gblk == nblocks - 1) {
#pragma unroll
            for (int c = 0; c < B; c++) { a.res->prev[c] = lastv[c]; a.res->rung[c] = (rp_packed >> (4 * c)) & 15u; a.res->cf[c] = a0.st.cf[c]; }
        }
        if (a.have_idx) {
            if (!a.idx_no_ulen) {
                uint8_t *ul = (uint8_t *)a.idx.ulen + (uint64_t)gblk * B;
#pragma unroll
                for (int c = 0; c < B; c++) ul[c] = (uint8_t)lens[c];
            }
            const uint32_t seg = gblk / a.g.seg_blocks;
            if (seg * a.g.seg_blocks == gblk) {
#pragma unroll
                for (int c = 0; c < B; c++) {
                    ((uint8_t *)a.idx.prev)[(uint64_t)seg * B + c] = (uint8_t)pvv[c];
                    a.idx.rung[(uint64_t)seg * B + c] = (uint8_t)((prp >> (4 * c)) & 15u);
                }
                a.idx.bitpos[seg] = ((uint64_t)chunk << 32) | pos;
            }
        }
    }
}

template <int B, bool RGB, uint64_t ORDER, bool STEP>
__global__ void __launch_bounds__(256, 4) enc_px_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t nblocks = (uint32_t)a.g.nblocks;

    uint32_t *etab = (uint32_t *)smem;                      // 512 entries
    uint32_t *wsum = etab + 512;                            // 64 dwords: scan scratch, [32..35] rungs of each wave's last lane
    uint32_t *outbuf = wsum + 64;                           // slot_dw dwords (a multiple of 4)
    // the code table is asked for now and written to LDS only before the first barrier: its round trip runs beside the
    // pixel loads instead of in front of them
    const uint4 tabv = ((const uint4 *)px_enc_tab.e)[tid & 127];
    for (uint32_t i = tid; i < a.slot_dw / 4; i += 256) ((uint4 *)outbuf)[i] = make_uint4(0, 0, 0, 0);
    const uint32_t etab_off = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)smem;

    const uint32_t chunk = blockIdx.x;
    const int64_t gs = (int64_t)chunk * 255 - 1 + tid;     // lane 0 is the halo block
    const bool valid = gs >= 0 && gs < (int64_t)nblocks, payload = valid && tid >= 1;
    const uint32_t gblk = valid ? (uint32_t)gs : 0u;
    uint32_t w[4][B], pd, total, pos;
    px_load_block<B, ORDER>(a, valid, gblk, w, pd);
    px_code_chunk<B, RGB, ORDER, STEP>(a, a0, chunk, valid, payload, gblk, w, pd, etab, wsum, outbuf, etab_off, tabv, total, pos);
    // the chunk's bits go to its slot; enc_concat_kernel moves them into place once every chunk is counted
    __syncthreads();
    const uint32_t nd4 = (total + 127) >> 7;
    uint4 *slot = (uint4 *)(a.scratch + (uint64_t)chunk * a.slot_dw);
    for (uint32_t d = tid; d < nd4; d += 256) slot[d] = ((const uint4 *)outbuf)[d];
    if (tid == 0) a.chunk_bits[chunk] = total;
}

// dispatch over the compile-time parameters
template <int B, bool RGB>
static void launch_enc_px_b(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    const bool step = a.g.mode != CM_FTL, z = a.g.order == ZCURVE;
    dim3 grid(plan.nchunks, a.ntiles), block(256);
    if (!z && !step) hipLaunchKernelGGL((enc_px_kernel<B, RGB, HILBERT, false>), grid, block, plan.lds_bytes, st, a);
    else if (!z && step) hipLaunchKernelGGL((enc_px_kernel<B, RGB, HILBERT, true>), grid, block, plan.lds_bytes, st, a);
    else if (z && !step) hipLaunchKernelGGL((enc_px_kernel<B, RGB, ZCURVE, false>), grid, block, plan.lds_bytes, st, a);
    else hipLaunchKernelGGL((enc_px_kernel<B, RGB, ZCURVE, true>), grid, block, plan.lds_bytes, st, a);
}
void launch_enc_px(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    if (a.g.bands == 1) launch_enc_px_b<1, false>(a, plan, st);
    else if (a.g.bands == 3) { if (plan.px_rgb) launch_enc_px_b<3, true>(a, plan, st); else launch_enc_px_b<3, false>(a, plan, st); }
    else { if (plan.px_rgb) launch_enc_px_b<4, true>(a, plan, st); else launch_enc_px_b<4, false>(a, plan, st); }
}

}  // namespace qb3dev
