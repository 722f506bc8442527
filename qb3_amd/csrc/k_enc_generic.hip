// qb3_amd/csrc/k_enc_generic.hip -- unit-per-lane FTL/BASE encoder for every type, band count, curve and band map
#include "qb3_enc_front.h"

namespace qb3dev {

template <typename T, bool STEP>
__global__ void enc_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    enc_scan_counter_reset(a);
    constexpr uint32_t UB = UBits<T>::v, UMASK = (1u << UB) - 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    const uint32_t bands = a.g.bands, nblocks = (uint32_t)a.g.nblocks;
    T g[16];
    EncFront<T> f;
    enc_front<T>(a, a0, smem, (31 + (a.slots - 1) * bands * (UB + 2 + 16 * (8 * (uint32_t)sizeof(T) + 1))) / 32 + 1, f, g, a.chunk0 + blockIdx.x);
    const uint32_t c = f.c, gblk = f.gblk, rung = f.rung, chunk = f.chunk;
    const bool payload = f.payload;
    const T used = f.used, pv = f.pv, lastv = f.lastv;
    uint8_t *rungs = f.rungs; uint16_t *etab = f.etab; uint32_t *outbuf = f.outbuf, *wsum = f.wsum;

    // ---- unit bit string.  Short units (rung < 8, always the case for 8-bit data) are assembled BEFORE the scan
    // into six pieces of at most 27 bits -- [switch, c0, c1] [c2..c4] [c5..c7] [c8..c10] [c11..c13] [c14, c15] --
    // so that only six words and their packed lengths stay live across the scan (the 16 values die here).
    // Wider units keep their values and are coded from the rule after the scan.
    uint32_t len = 0, prung = 0, delta = 0;
    uint32_t pc[6] = {0, 0, 0, 0, 0, 0}, plens = 0;        // pieces and their lengths (5 bits each)
    bool pieces = false;
    if (payload) {
        prung = (gblk == 0) ? a0.st.rung[c] : rungs[tid - bands];
        delta = (rung - prung) & UMASK;
        const uint32_t csl = cs_len<UB>(delta), csc = cs_code<UB>(delta);
        len = csl;
        if (used <= 1) {            // flag, then the sixteen one-bit values if any is set (reference QB3encode.h:159-166)
            uint32_t bits = 0;
#pragma unroll
            for (uint32_t i = 0; i < 16; i++) bits |= (uint32_t)(g[i] & 1) << i;
            const uint32_t l = 1 + (used ? 16 : 0);
            pc[0] = csc | ((uint32_t)used << csl) | (bits << (csl + 1));
            plens = csl + l;        // <= 8 + 17
            len += l;
            pieces = true;
        } else {
            const T top = (T)((T)1 << rung);
            if (STEP) {     // clear the rung bit of the last value of a 1..10..0 rung-bit run (reference QB3encode.h:169-176)
                uint32_t bits = 0;
#pragma unroll
                for (uint32_t i = 0; i < 16; i++) bits |= (uint32_t)((g[i] >> rung) & 1) << i;
                if ((bits & (bits + 1)) == 0) {
                    const uint32_t n = __popc(bits);    // >= 1 here
#pragma unroll
                    for (uint32_t i = 0; i < 16; i++) if (i + 1 == n) g[i] ^= top;
                }
            }
            if (sizeof(T) == 1 || rung < 8) {
                // code and length per value from the LDS table: the code rule with the middle swap (QB3encode.h:30-33,132-141)
                const uint16_t *tab = etab + enc_tab_off(rung);
                uint32_t acc = csc, al = csl, k = 0, lsum = 0;
#pragma unroll
                for (uint32_t i = 0; i < 16; i++) {
                    const uint32_t e = tab[(uint32_t)g[i]];
                    const uint32_t code = e & 0xfff, l = e >> 12;
                    acc |= code << al; al += l; lsum += l;
                    if (i == 1 || i == 4 || i == 7 || i == 10 || i == 13 || i == 15) {   // piece boundary (static)
                        pc[k] = acc; plens |= al << (5 * k); k++; acc = 0; al = 0;
                    }
                }
                len += lsum;
                pieces = true;
            } else {                                 // computed three-length code, no swap above rung 7
                uint32_t extra = 0;
                const T half = (T)(top >> 1);
#pragma unroll
                for (uint32_t i = 0; i < 16; i++) extra += (g[i] >= half) + (g[i] >= top);
                len += 16 * rung + extra;
            }
        }
    }
    uint32_t total;
    const uint32_t pos = block_exscan(len, wsum, &total);

    // ---- emit: the chunk's bits are assembled in LDS starting at bit 0 and go to the chunk's private slot.
    // Where they land in the stream is only known after all chunks are counted; enc_concat_kernel moves them.
    // (A single pass with a decoupled look-back was measured slower here: at ~160 chunks/us the prefix frontier
    // cannot keep up with L2 polling latency, and a ticket counter alone caps the kernel at ~88 chunks/us.)
    if (payload) {
        LdsWriter w;
        w.init(outbuf, pos);
        if (pieces) {
#pragma unroll
            for (uint32_t k = 0; k < 6; k++) w.put(pc[k], (plens >> (5 * k)) & 31);
        } else {
            w.put(cs_code<UB>(delta), cs_len<UB>(delta));
#pragma unroll
            for (uint32_t i = 0; i < 16; i++) put_value<T>(w, g[i], rung);
        }
        w.finish();
        // coder state on leaving the image, for handle statefulness (reference QB3encode.h:446-449)
        if (gblk == nblocks - 1) { a.res->prev[c] = (uint64_t)lastv; a.res->rung[c] = rung; a.res->cf[c] = a0.st.cf[c]; }
        if (a.have_idx) {
            if (!a.idx_no_ulen && a.g.ulen_sz == 1) ((uint8_t *)a.idx.ulen)[(uint64_t)gblk * bands + c] = (uint8_t)len;
            else if (!a.idx_no_ulen && a.g.ulen_sz == 2) ((uint16_t *)a.idx.ulen)[(uint64_t)gblk * bands + c] = (uint16_t)len;
            const uint32_t seg = gblk / a.g.seg_blocks;
            if (seg * a.g.seg_blocks == gblk) {
                ((T *)a.idx.prev)[(uint64_t)seg * bands + c] = pv;
                a.idx.rung[(uint64_t)seg * bands + c] = (uint8_t)prung;
                if (c == 0) a.idx.bitpos[seg] = ((uint64_t)chunk << 32) | pos;     // chunk-relative; fixed up by enc_finish_kernel
            }
        }
    }
    __syncthreads();
    const uint32_t nd = (total + 31) >> 5;
    uint32_t *slot = a.scratch + (uint64_t)chunk * a.slot_dw;
    for (uint32_t d = tid; d < nd; d += nthr) slot[d] = outbuf[d];
    if (tid == 0) a.chunk_bits[chunk] = total;
}

template <typename T>
static void launch_enc_generic_t(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    dim3 grid(a.chunk_end - a.chunk0, a.ntiles), block(plan.threads);
    if (a.g.mode != CM_FTL) hipLaunchKernelGGL((enc_kernel<T, true>), grid, block, plan.lds_bytes, st, a);
    else hipLaunchKernelGGL((enc_kernel<T, false>), grid, block, plan.lds_bytes, st, a);
}
void launch_enc_generic(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    switch (a.g.tsz) {
    case 1: launch_enc_generic_t<uint8_t>(a, plan, st); break;
    case 2: launch_enc_generic_t<uint16_t>(a, plan, st); break;
    case 4: launch_enc_generic_t<uint32_t>(a, plan, st); break;
    default: launch_enc_generic_t<uint64_t>(a, plan, st); break;
    }
}

}  // namespace qb3dev
