// qb3_amd/csrc/k_enc_px16.hip -- 16-bit encoder, lane per (block, band group)
#include "qb3_px.h"

namespace qb3dev {

// ------------------------------------------------------------------ 16-bit: lane per (block, band group), in registers
// The 16-bit counterpart of enc_px_kernel.  A lane owns BG <= 4 bands of one block (bands = NG x BG: 8-band data
// is two lanes per block); units of a block are consecutive in the stream, so lane order is still stream order.
// Two values per register: v_perm_b32 gathers curve-ordered pairs, band difference / running delta / mag-sign are
// packed 16-bit operations (v_pk_sub_u16, v_pk_lshlrev_b16, v_pk_ashrrev_i16).  Rungs up to 7 use the same
// compile-time code table as the 8-bit kernel, higher rungs the code rule in ALU (no middle swap above rung 7);
// pieces are 64 bits wide (three codes of at most 17 bits).  Slot 0 of a workgroup (its first NG lanes) is the halo
// block, so a chunk is 256/NG - 1 blocks.
// values 2k, 2k+1 (curve order) of band c of the lane's group; w[y][j] = dword j of the lane's row y, halfword
// x*BG + c of it is band c of pixel x
template <int BG, uint64_t ORDER>
__device__ __forceinline__ uint32_t gather_pair16(const uint32_t (&w)[4][2 * BG], int k, int c) {
    const int n0 = (int)order_nib(ORDER, 2 * k), n1 = (int)order_nib(ORDER, 2 * k + 1);
    const int h0 = (n0 & 3) * BG + c, h1 = (n1 & 3) * BG + c;
    const uint32_t sel = (uint32_t)(2 * (h0 & 1)) | (uint32_t)(2 * (h0 & 1) + 1) << 8 |
                         (uint32_t)(4 + 2 * (h1 & 1)) << 16 | (uint32_t)(4 + 2 * (h1 & 1) + 1) << 24;
    return __builtin_amdgcn_perm(w[n1 >> 2][h1 >> 1], w[n0 >> 2][h0 >> 1], sel);
}

template <int BG, bool RGB, uint64_t ORDER, bool STEP>
__global__ void __launch_bounds__(256, 2) enc_px16_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    enc_scan_counter_reset(a);
    constexpr uint32_t UB = 4, UMASK = 15;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t nblocks = (uint32_t)a.g.nblocks, nbx = a.g.nbx, B = a.g.bands, NG = a.px_ng, S = 256 / NG;
    const uint64_t stride = a.g.stride;                     // in values
    const uint32_t slot = fastdiv(tid, NG, a.px_magic_ng), grp = tid - slot * NG, band0 = grp * BG;

    uint32_t *etab = (uint32_t *)smem;                      // 512 entries
    uint32_t *wsum = etab + 512;                            // 64 dwords of scan scratch
    uint32_t *rp_s = wsum + 64;                             // 256: every lane's packed rungs
    uint32_t *outbuf = rp_s + 256;                          // slot_dw dwords (a multiple of 4)
    const uint4 tabv = ((const uint4 *)px_enc_tab.e)[tid & 127];     // written to LDS before the first barrier, see enc_px_kernel
    for (uint32_t i = tid; i < a.slot_dw / 4; i += 256) ((uint4 *)outbuf)[i] = make_uint4(0, 0, 0, 0);
    const uint32_t etab_off = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)smem;

    const uint32_t chunk = a.chunk0 + blockIdx.x;      // (chunk0: the first chunk of this launch -- 0 but for the strips of a pipelined host call)
    const int64_t gs = (int64_t)chunk * (S - 1) - 1 + slot; // slot 0 is the halo block
    const bool valid = slot < S && gs >= 0 && gs < (int64_t)nblocks, payload = valid && slot >= 1;
    const uint32_t gblk = valid ? (uint32_t)gs : 0u;

    // ---- load the lane's bands of the block (4 rows) and of the previous block's last visited pixel
    uint32_t w[4][2 * BG], pvals[BG];
    constexpr uint32_t n15 = order_nib(ORDER, 15);
#pragma unroll
    for (int c = 0; c < BG; c++) pvals[c] = 0;
    // N dwords starting at a halfword address: when it is not dword aligned (odd strides, the shifted last column, odd
    // widths) the N+1 aligned dwords covering them are read and funnel-shifted; nothing is read beyond the aligned dword
    // holding the last halfword
    auto load_dw = [&](const uint16_t *p, uint32_t *dst, auto nconst) {
        constexpr int N = decltype(nconst)::value;
        if (a.px_aligned) {                 // workgroup uniform
            const uint32_t *q = (const uint32_t *)p;
#pragma unroll
            for (int t = 0; t < N; t++) dst[t] = q[t];
        } else {
            const uint32_t sh = 8 * ((uint32_t)(uintptr_t)p & 2);
            const uint32_t *q = (const uint32_t *)((uintptr_t)p & ~(uintptr_t)3);
            uint32_t d[N + 1];
#pragma unroll
            for (int t = 0; t < N; t++) d[t] = q[t];
            d[N] = sh ? q[N] : 0u;
#pragma unroll
            for (int t = 0; t < N; t++) dst[t] = __builtin_amdgcn_alignbit(d[t + 1], d[t], sh);
        }
    };
    if (valid) {
        const uint32_t by = gblk / nbx, bx = gblk - by * nbx;
        const uint32_t x0 = (4 * bx + 4 > a.g.w) ? a.g.w - 4 : 4 * bx;     // last column / row is shifted, not padded
        const uint32_t y0 = (4 * by + 4 > a.g.h) ? a.g.h - 4 : 4 * by;
        const uint16_t *p0 = (const uint16_t *)a.img + (uint64_t)y0 * stride + (uint64_t)x0 * B + band0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint16_t *rowp = p0 + (uint64_t)r * stride;
            if (BG % 2 == 0) {      // a pixel's BG values are whole dwords; pixels are B values apart
#pragma unroll
                for (int x = 0; x < 4; x++) load_dw(rowp + (uint64_t)x * B, &w[r][x * (BG / 2)], std::integral_constant<int, (BG / 2 ? BG / 2 : 1)>());
            } else                  // BG == bands: the row of the block is contiguous
                load_dw(rowp, &w[r][0], std::integral_constant<int, 2 * BG>());
        }
        if (gblk) {
            const uint32_t pb = gblk - 1, pby = pb / nbx, pbx = pb - pby * nbx;
            const uint32_t px0 = (4 * pbx + 4 > a.g.w) ? a.g.w - 4 : 4 * pbx;
            const uint32_t py0 = (4 * pby + 4 > a.g.h) ? a.g.h - 4 : 4 * pby;
            const uint16_t *q = (const uint16_t *)a.img + (uint64_t)(py0 + (n15 >> 2)) * stride + (uint64_t)(px0 + (n15 & 3)) * B + band0;
#pragma unroll
            for (int c = 0; c < BG; c++) pvals[c] = q[c];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int k = 0; k < 2 * BG; k++) w[r][k] = 0;
    }

    // ---- per band: values in curve order, band difference, running delta, mag-sign -- two values per register
    uint32_t cur[BG][8];
#pragma unroll
    for (int c = 0; c < BG; c++)
#pragma unroll
        for (int k = 0; k < 8; k++) cur[c][k] = gather_pair16<BG, ORDER>(w, k, c);
    uint32_t gp[BG][8], usedv[BG], lastv[BG], pvv[BG];
    uint32_t rp_packed = 0;
#pragma unroll
    for (int c = 0; c < BG; c++) {
        const int cb = core_of<BG, RGB>(c);
        uint32_t prv;
        // the R-G, G, B-G map applies to the first three bands of the image: group 0 only
        const bool diff = cb != c && grp == 0;
        if (gblk == 0) prv = (uint32_t)a0.st.prev[band0 + c] & 0xffffu;
        else prv = diff ? (pvals[c] - pvals[cb]) & 0xffffu : pvals[c];
        pvv[c] = prv;
        uint32_t x[8], u = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = (cb != c) ? pk_sub16(cur[c][k], diff ? cur[cb][k] : 0u) : cur[c][k];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t before = k ? __builtin_amdgcn_alignbit(x[k], x[k - 1], 16) : ((x[0] << 16) | prv);
            gp[c][k] = pk_mags16(pk_sub16(x[k], before));
            u |= gp[c][k];
        }
        u = (u | (u >> 16)) & 0xffffu;
        usedv[c] = u; lastv[c] = x[7] >> 16;
        rp_packed |= topbit32(u | 1) << (4 * c);
    }
    // rungs of the same bands of the previous block: NG lanes back
    rp_s[tid] = rp_packed;
    if (tid < 128) ((uint4 *)etab)[tid] = tabv;
    __syncthreads();
    uint32_t prp = tid >= NG ? rp_s[tid - NG] : 0u;
    if (gblk == 0) { prp = 0;
#pragma unroll
        for (int c = 0; c < BG; c++) prp |= ((uint32_t)a0.st.rung[band0 + c] & 15u) << (4 * c); }

    // ---- per band: the unit's bit string as six pieces (64-bit), pl = piece length
    uint64_t pc[BG][6];
    uint32_t pl[BG][6], lens[BG], blen[1] = { 0 };
#pragma unroll
    for (int c = 0; c < BG; c++) {
#pragma unroll
        for (int k = 0; k < 6; k++) { pc[c][k] = 0; pl[c][k] = 0; }
        lens[c] = 0;
        if (payload) {
            const uint32_t rung = (rp_packed >> (4 * c)) & 15u, prung = (prp >> (4 * c)) & 15u, used = usedv[c];
            const uint32_t delta = (rung - prung) & UMASK;
            const uint32_t csl = cs_len<UB>(delta), csc = cs_code<UB>(delta);
            if (used <= 1) {
                uint32_t bits = 0;
#pragma unroll
                for (int i = 0; i < 16; i++) bits |= ((gp[c][i >> 1] >> (16 * (i & 1))) & 1u) << i;
                pc[c][0] = csc | (used << csl); pl[c][0] = csl + 1;
                pc[c][1] = bits; pl[c][1] = used ? 16 : 0;
                lens[c] = pl[c][0] + pl[c][1];
            } else {
                uint32_t g8[8];
#pragma unroll
                for (int k = 0; k < 8; k++) g8[k] = gp[c][k];
                if (STEP) {     // clear the rung bit of the last value of a 1..10..0 rung-bit run (reference QB3encode.h:169-176)
                    uint32_t bits = 0;
#pragma unroll
                    for (int i = 0; i < 16; i++) bits |= ((g8[i >> 1] >> (16 * (i & 1) + rung)) & 1u) << i;
                    if ((bits & (bits + 1)) == 0) {
                        const uint32_t n = __popc(bits) - 1;        // index of the value to change
#pragma unroll
                        for (int k = 0; k < 8; k++) if ((n >> 1) == (uint32_t)k) g8[k] ^= (1u << rung) << (16 * (n & 1));
                    }
                }
                constexpr int first[7] = {0, 2, 5, 8, 11, 14, 16};  // piece k holds values first[k] .. first[k+1]-1
                uint32_t lsum = 0;
                if (rung <= 7) {                                     // all values below 256: the code table
                    const uint32_t tb = etab_off + (8u << rung);
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        uint32_t acc = 0, s = 0;
#pragma unroll
                        for (int i = first[k + 1] - 1; i >= first[k]; i--) {
                            const uint32_t m = (g8[i >> 1] >> (16 * (i & 1))) & 0xffffu;
                            const uint32_t e = *lds_at((m << 2) + tb);
                            acc = __builtin_amdgcn_alignbit(acc, e, e);      // (acc << length) | code (qb3_px.h, PxEncTab)
                            s += e;
                        }
                        s = 32u * (uint32_t)(first[k + 1] - first[k]) - (s & 0xffu);
                        uint64_t a64 = acc;
                        if (k == 0) { a64 = (a64 << csl) | csc; s += csl; }
                        pc[c][k] = a64; pl[c][k] = s; lsum += s;
                    }
                } else {                                             // the code rule (reference QB3encode.h:30-33), no swap above rung 7
                    const uint32_t top = 1u << rung, half = top >> 1;
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        uint64_t acc = 0;
                        uint32_t s = 0;
#pragma unroll
                        for (int i = first[k + 1] - 1; i >= first[k]; i--) {
                            const uint32_t m = (g8[i >> 1] >> (16 * (i & 1))) & 0xffffu;
                            const bool c1 = m >= half, c2 = m >= top;
                            const uint32_t code = c2 ? (((m - top) << 2) | 3u) : c1 ? (((m - half) << 2) | 1u) : (m << 1);
                            const uint32_t len = rung + c1 + c2;
                            acc = (acc << len) | code;
                            s += len;
                        }
                        if (k == 0) { acc = (acc << csl) | csc; s += csl; }
                        pc[c][k] = acc; pl[c][k] = s; lsum += s;
                    }
                }
                lens[c] = lsum;
            }
            blen[0] += lens[c];
        }
    }
    block_exscan_dpp<1>(blen, wsum);
    const uint32_t pos = blen[0], total = wsum[0] + wsum[1] + wsum[2] + wsum[3];

    if (payload) {
        LdsWriter wr;
        wr.init(outbuf, pos);
#pragma unroll
        for (int c = 0; c < BG; c++)
#pragma unroll
            for (int k = 0; k < 6; k++) wr.put64(pc[c][k], pl[c][k]);
        wr.finish();
        if (gblk == nblocks - 1) {
#pragma unroll
            for (int c = 0; c < BG; c++) { a.res->prev[band0 + c] = lastv[c]; a.res->rung[band0 + c] = (rp_packed >> (4 * c)) & 15u; a.res->cf[band0 + c] = a0.st.cf[band0 + c]; }
        }
        if (a.have_idx) {
            if (!a.idx_no_ulen) {
                uint16_t *ul = (uint16_t *)a.idx.ulen + (uint64_t)gblk * B + band0;
#pragma unroll
                for (int c = 0; c < BG; c++) ul[c] = (uint16_t)lens[c];
            }
            const uint32_t seg = gblk / a.g.seg_blocks;
            if (seg * a.g.seg_blocks == gblk) {
#pragma unroll
                for (int c = 0; c < BG; c++) {
                    ((uint16_t *)a.idx.prev)[(uint64_t)seg * B + band0 + c] = (uint16_t)pvv[c];
                    a.idx.rung[(uint64_t)seg * B + band0 + c] = (uint8_t)((prp >> (4 * c)) & 15u);
                }
                if (grp == 0) a.idx.bitpos[seg] = ((uint64_t)chunk << 32) | pos;
            }
        }
    }
    __syncthreads();
    const uint32_t nd4 = (total + 127) >> 7;
    uint4 *slotp = (uint4 *)(a.scratch + (uint64_t)chunk * a.slot_dw);
    for (uint32_t d = tid; d < nd4; d += 256) slotp[d] = ((const uint4 *)outbuf)[d];
    if (tid == 0) a.chunk_bits[chunk] = total;
}

template <int BG, bool RGB>
static void launch_enc_px16_b(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    const bool step = a.g.mode != CM_FTL, z = a.g.order == ZCURVE;
    dim3 grid(a.chunk_end - a.chunk0, a.ntiles), block(256);
    if (!z && !step) hipLaunchKernelGGL((enc_px16_kernel<BG, RGB, HILBERT, false>), grid, block, plan.lds_bytes, st, a);
    else if (!z && step) hipLaunchKernelGGL((enc_px16_kernel<BG, RGB, HILBERT, true>), grid, block, plan.lds_bytes, st, a);
    else if (z && !step) hipLaunchKernelGGL((enc_px16_kernel<BG, RGB, ZCURVE, false>), grid, block, plan.lds_bytes, st, a);
    else hipLaunchKernelGGL((enc_px16_kernel<BG, RGB, ZCURVE, true>), grid, block, plan.lds_bytes, st, a);
}
void launch_enc_px16(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    switch (plan.px16_bg) {
    case 1: launch_enc_px16_b<1, false>(a, plan, st); break;
    case 2: launch_enc_px16_b<2, false>(a, plan, st); break;
    case 3: if (plan.px_rgb) launch_enc_px16_b<3, true>(a, plan, st); else launch_enc_px16_b<3, false>(a, plan, st); break;
    default: if (plan.px_rgb) launch_enc_px16_b<4, true>(a, plan, st); else launch_enc_px16_b<4, false>(a, plan, st); break;
    }
}

}  // namespace qb3dev
