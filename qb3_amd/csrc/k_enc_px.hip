// qb3_amd/csrc/k_enc_px.hip -- 8-bit grey / RGB / RGBA encoder, lane per block
#include "qb3_px_enc.h"

namespace qb3dev {

// Codes the chunk whose blocks the lanes hold (lane 0: the halo block) into the LDS bit buffer, from bit 0; the
// buffer must be zero.  total: bits of the chunk; pos: where the lane's block starts.
template <int B, bool RGB, uint64_t ORDER, bool STEP>
__device__ __forceinline__ void px_code_chunk(const EncArgs &a, const EncArgs &a0, uint32_t chunk, bool valid, bool payload, uint32_t gblk,
                                              const uint32_t (&w)[4][B], uint32_t pd, uint32_t *etab, uint32_t *wsum, uint32_t *outbuf,
                                              uint32_t etab_off, const uint4 &tabv, uint32_t &total, uint32_t &pos) {
    constexpr uint32_t UMASK = 7;
    const uint32_t nblocks = (uint32_t)a.g.nblocks;
    PxFront<B> f;
    px_front<B, RGB, ORDER>(a0, gblk, w, pd, etab, wsum, tabv, f);
    const uint32_t rp_packed = f.rp_packed, prp = f.prp;

    // ---- per band: the unit's bit string as six pieces of at most 27 bits; pl = piece length (low byte)
    uint32_t pc[B][6], pl[B][6], lens[B], blen[1] = { 0 };
#pragma unroll
    for (int c = 0; c < B; c++) {
#pragma unroll
        for (int k = 0; k < 6; k++) { pc[c][k] = 0; pl[c][k] = 0; }
        lens[c] = 0;
        if (payload) {
            const uint32_t rung = (rp_packed >> (4 * c)) & 15u, prung = (prp >> (4 * c)) & 15u, used = f.usedv[c];
            const uint32_t delta = (rung - prung) & UMASK;
            const uint32_t csl = __builtin_amdgcn_ubfe(cs3_lens(), 4 * delta, 4), csc = (uint32_t)(cs3_codes() >> (8 * delta)) & 0xffu;
            if (used <= 1) lens[c] = px_unit_low(f.gp[c], used, csl, csc, pc[c], pl[c]);
            else lens[c] = px_unit_pieces<STEP>(f.gp[c], rung, csl, csc, etab_off + (8u << rung), pc[c], pl[c]);
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
            for (int c = 0; c < B; c++) { a.res->prev[c] = f.lastv[c]; a.res->rung[c] = (rp_packed >> (4 * c)) & 15u; a.res->cf[c] = a0.st.cf[c]; }
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
                    ((uint8_t *)a.idx.prev)[(uint64_t)seg * B + c] = (uint8_t)f.pvv[c];
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
    enc_scan_counter_reset(a);
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

    const uint32_t chunk = a.chunk0 + blockIdx.x;      // (chunk0: the first chunk of this launch -- 0 but for the strips of a pipelined host call)
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
    dim3 grid(a.chunk_end - a.chunk0, a.ntiles), block(256);
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
