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
// buffer must be zero.  total: bits of the chunk; pos: where the lane's block starts.  DEFER_POS: the caller writes
// the index position of a segment that starts at this lane (seg_out, else ~0) once it knows the chunk's offset.
// place(total), called by every thread once the chunk's size is known and before any of its bits is written, returns the
// bit of the buffer the chunk starts at (the super-chunk encoder appends chunks and may empty the buffer there).
struct PlaceAtZero { __device__ __forceinline__ uint32_t operator()(uint32_t) const { return 0; } };
template <int B, bool RGB, uint64_t ORDER, bool STEP, bool DEFER_POS, class PLACE = PlaceAtZero>
__device__ __forceinline__ void px_code_chunk(const EncArgs &a, const EncArgs &a0, uint32_t chunk, bool valid, bool payload, uint32_t gblk,
                                              const uint32_t (&w)[4][B], uint32_t pd, uint32_t *etab, uint32_t *wsum, uint32_t *outbuf,
                                              uint32_t etab_off, bool put_tab, const uint4 &tabv, uint32_t &total, uint32_t &pos, uint32_t &seg_out,
                                              PLACE place = PLACE()) {
    constexpr uint32_t UMASK = 7;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nblocks = (uint32_t)a.g.nblocks;
    seg_out = ~0u;
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
    if (put_tab && tid < 128) ((uint4 *)etab)[tid] = tabv;
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
    const uint32_t mybits = blen[0];
    block_exscan_dpp<1>(blen, wsum);
    pos = blen[0]; total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    (void)mybits;
    const uint32_t bit0 = place(total);

    if (payload) {
        LdsWriter32 wr;
        wr.init(outbuf, bit0 + pos);
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
                if (DEFER_POS) seg_out = seg; else a.idx.bitpos[seg] = ((uint64_t)chunk << 32) | pos;
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
    uint32_t w[4][B], pd, total, pos, seg;
    px_load_block<B, ORDER>(a, valid, gblk, w, pd);
    px_code_chunk<B, RGB, ORDER, STEP, false>(a, a0, chunk, valid, payload, gblk, w, pd, etab, wsum, outbuf, etab_off, true, tabv, total, pos, seg);
    // the chunk's bits go to its slot; enc_concat_kernel moves them into place once every chunk is counted
    __syncthreads();
    const uint32_t nd4 = (total + 127) >> 7;
    uint4 *slot = (uint4 *)(a.scratch + (uint64_t)chunk * a.slot_dw);
    for (uint32_t d = tid; d < nd4; d += 256) slot[d] = ((const uint4 *)outbuf)[d];
    if (tid == 0) a.chunk_bits[chunk] = total;
}

// ---- single pass: persistent workgroups, decoupled look-back ------------------------------------------------
// The same coding, but a chunk's bits go from LDS straight to their place in the stream: no slot, no concatenate
// pass, half the HBM traffic.  What a chunk needs for that is the sum of the bit counts of all chunks before it.
// Workgroup g of G (all resident: G is sized from the occupancy query, and every wait is bounded) codes chunks g,
// g + G, ...  After coding a chunk it publishes its count in lookback[chunk] as ONE 8-byte word {state, value} --
// state 1: the chunk's own count, 2: the count of everything up to and including the chunk -- with an agent-scope
// store, and wave 0 reads the words of its predecessors 64 at a time (agent-scope loads: the XCDs' L2s are not coherent
// with each other) until it meets a state-2 word behind only state-1 words; their sum is the chunk's offset.  The
// next chunk's pixel loads are issued before that wait, so their round trip runs beside it.  Dwords a chunk shares
// with its neighbours go to the seam table as in the slot path (enc_seam_kernel assembles them).
// A wait that exceeds its bound (a predecessor that is not resident: never seen, by construction) publishes state 3,
// which every later chunk passes on; the host then codes the image again through the slots.
constexpr uint64_t LB_VAL = (1ull << 62) - 1;
__device__ __forceinline__ uint64_t lb_load(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lb_store(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Wave 0: the sum of the counts of every word in front of word `idx` (idx >= 1) -- words are added until one with a running
// total (state 2) is met.  Returns 2, or 3 when a predecessor gave up or a wait exceeded its bound.
__device__ __forceinline__ uint32_t lb_look(const uint64_t *lb, uint32_t idx0, uint32_t lane, uint64_t &excl_out) {
    // a lane reads LBW words a round -- the window is 64 * LBW words: the running total moves down the words by one
    // window per memory round trip, and that, not the coding, would set the pace with a narrow one
    constexpr int LBW = 8;
    uint64_t excl = 0;
    uint32_t state = 0;          // 2: done, 3: abort
    int64_t base = (int64_t)idx0 - 1;
    uint32_t spins = 0;
    while (state == 0) {
        // wait for the nearest word with ONE load per try (a whole window polled by every waiting workgroup costs
        // the memory system more than the coding), then read the window behind it once
        {
            uint32_t st = 0;
            while (true) {
                st = (uint32_t)(lb_load(&lb[base]) >> 62);          // (every lane: the same word, one request)
                if (st != 0) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 22)) { st = 3; break; }
            }
            if (st == 3) { state = 3; break; }
        }
        uint64_t v[LBW];
#pragma unroll
        for (int j = 0; j < LBW; j++) {             // word p = 64 * j + lane of the window is word base - p (a load: 512 bytes in a row)
            const int64_t idx = base - (int64_t)(64 * j + lane);
            v[j] = idx >= 0 ? lb_load(&lb[idx]) : (2ull << 62);          // in front of word 0: a total of zero
        }
        uint32_t nr = 64 * LBW, ni = 64 * LBW, nd = 64 * LBW;         // first word not there yet / with a running total / dead
#pragma unroll
        for (int j = 0; j < LBW; j++) {
            const uint32_t st = (uint32_t)(v[j] >> 62);
            const uint64_t notready = __ballot(st == 0), incl = __ballot(st >= 2), dead = __ballot(st == 3);
            if (notready) nr = min(nr, (uint32_t)(64 * j + __builtin_ctzll(notready)));
            if (incl) ni = min(ni, (uint32_t)(64 * j + __builtin_ctzll(incl)));
            if (dead) nd = min(nd, (uint32_t)(64 * j + __builtin_ctzll(dead)));
        }
        const uint32_t take = ni < nr ? ni + 1 : nr;                   // words that can be added now
        if (nd < take) { state = 3; break; }
        uint32_t part = 0;                                             // the counts: each below 2^24
#pragma unroll
        for (int j = 0; j < LBW; j++) {
            const uint32_t p = 64 * j + lane;
            part += (p < take && p != ni) ? (uint32_t)(v[j] & LB_VAL) : 0u;
        }
        part = wave_iscan32(part);
        excl += (uint32_t)__builtin_amdgcn_readlane((int)part, 63);
        if (ni < nr) {
            uint64_t t = 0;
#pragma unroll
            for (int j = 0; j < LBW; j++) t = (64 * j + lane == ni) ? v[j] : t;
            const int src = __builtin_amdgcn_readfirstlane((int)(ni & 63));
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)t, src);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(t >> 32), src);
            excl += (((uint64_t)hi << 32) | lo) & LB_VAL;
            state = 2;
        } else {
            base -= take;             // (take >= 1: the nearest word was there)
        }
    }
    excl_out = excl;
    return state;
}

template <int B, bool RGB, uint64_t ORDER, bool STEP, bool LOOKBACK>
__global__ void __launch_bounds__(256, 3) enc_px_sp_kernel(const EncArgs a0) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *etab = (uint32_t *)smem;                      // 512 entries
    uint32_t *wsum = etab + 512;                            // 64 dwords: scan scratch; [40..43]: the chunk's offset (lo, hi), abort flag
    uint32_t *outbuf = wsum + 64;                           // slot_dw dwords (a multiple of 4)
    const uint4 tabv = ((const uint4 *)px_enc_tab.e)[tid & 127];
    for (uint32_t i = tid; i < a0.slot_dw / 4; i += 256) ((uint4 *)outbuf)[i] = make_uint4(0, 0, 0, 0);
    const uint32_t etab_off = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)smem;
    const uint32_t nchunks = a0.nchunks, G = gridDim.x;
    const uint64_t nq = (uint64_t)nchunks * a0.ntiles;     // chunks of all tiles, tile-major

    auto chunk_lanes = [&](uint64_t q, uint32_t &tile, uint32_t &chunk, bool &valid, bool &payload, uint32_t &gblk) {
        tile = (uint32_t)(q / nchunks); chunk = (uint32_t)(q - (uint64_t)tile * nchunks);
        const int64_t gs = (int64_t)chunk * 255 - 1 + tid;  // lane 0 is the halo block
        valid = gs >= 0 && gs < (int64_t)a0.g.nblocks; payload = valid && tid >= 1;
        gblk = valid ? (uint32_t)gs : 0u;
    };
    uint64_t q = blockIdx.x;
    uint32_t w[4][B], pd = 0;
    uint32_t tile = 0, chunk = 0, gblk = 0;
    bool valid = false, payload = false, first = true;
    if (q < nq) {
        chunk_lanes(q, tile, chunk, valid, payload, gblk);
        px_load_block<B, ORDER>(enc_for_tile(a0, tile), valid, gblk, w, pd);
    }
    while (q < nq) {
        const EncArgs a = enc_for_tile(a0, tile);
        uint32_t total, pos, seg;
        px_code_chunk<B, RGB, ORDER, STEP, LOOKBACK>(a, a0, chunk, valid, payload, gblk, w, pd, etab, wsum, outbuf, etab_off, first, tabv, total, pos, seg);
        first = false;
        uint64_t *lb = a.lookback;
        if (LOOKBACK && tid == 0) lb_store(&lb[chunk], ((chunk == 0 ? 2ull : 1ull) << 62) | total);
        // the next chunk's pixels: requested now, used after the wait and the write-out
        const uint64_t qn = q + G;
        uint32_t ntile = 0, nchunk = 0, ngblk = 0;
        bool nvalid = false, npayload = false;
        if (qn < nq) {
            chunk_lanes(qn, ntile, nchunk, nvalid, npayload, ngblk);
            px_load_block<B, ORDER>(enc_for_tile(a0, ntile), nvalid, ngblk, w, pd);
        }
        if (!LOOKBACK) {
            // persistent workgroups, slots: the chunk's bits go to its slot (enc_concat_kernel moves them into place); what the
            // persistence buys is the next chunk's pixels arriving while these bits leave
            __syncthreads();
            const uint32_t nd4 = (total + 127) >> 7;
            uint4 *slot = (uint4 *)(a.scratch + (uint64_t)chunk * a.slot_dw);
            for (uint32_t d = tid; d < nd4; d += 256) slot[d] = ((const uint4 *)outbuf)[d];
            if (tid == 0) a.chunk_bits[chunk] = total;
            __syncthreads();
            for (uint32_t d = tid; d < nd4; d += 256) ((uint4 *)outbuf)[d] = make_uint4(0, 0, 0, 0);
            q = qn; tile = ntile; chunk = nchunk; valid = nvalid; payload = npayload; gblk = ngblk;
            continue;
        }
        // ---- the chunk's offset: look back over the predecessors' words
        if (wave == 0) {
            uint64_t excl = 0;
            const uint32_t state = chunk == 0 ? 2u : lb_look(lb, chunk, lane, excl);
            if (lane == 0) {
                lb_store(&lb[chunk], (state == 3 ? 3ull << 62 : 2ull << 62) | ((excl + total) & LB_VAL));
                wsum[40] = (uint32_t)excl; wsum[41] = (uint32_t)(excl >> 32); wsum[42] = state;
            }
        }
        __syncthreads();
        const uint64_t excl = ((uint64_t)wsum[41] << 32) | wsum[40];
        if (wsum[42] == 3 && tid == 0) lb[nchunks] = 1;          // abort flag: enc_seam_kernel hands it to the host
        // ---- what waited for the offset: index positions, the offset table of the seam pass, the stream itself
        if (seg != ~0u) a.idx.bitpos[seg] = excl + pos;
        if (tid == 0) {
            a.chunk_off[chunk] = excl;
            if (chunk == nchunks - 1) a.group_sum[(nchunks + SCAN_GROUP - 1) / SCAN_GROUP] = excl + total;
        }
        {
            const uint64_t G0 = (uint64_t)a.out_bit0 + excl;
            const uint32_t phase = (uint32_t)G0 & 31, sh = (32 - phase) & 31;
            const uint32_t nd = (phase + total + 31) >> 5, tailbits = (phase + total) & 31;
            uint32_t *gout = a.out32 + (G0 >> 5);
            // four output dwords a lane (one 16-byte store); dwords shared with a neighbouring chunk go to the seam table
            for (uint32_t d0 = 4 * tid; d0 < nd; d0 += 1024) {
                const uint4 c4 = *(const uint4 *)(outbuf + d0);            // (the buffer is zero behind the chunk's last dword)
                const uint32_t prv = d0 ? outbuf[d0 - 1] : 0u;
                uint32_t v[4] = { c4.x, c4.y, c4.z, c4.w };
                if (phase) {
                    v[3] = __builtin_amdgcn_alignbit(c4.w, c4.z, sh); v[2] = __builtin_amdgcn_alignbit(c4.z, c4.y, sh);
                    v[1] = __builtin_amdgcn_alignbit(c4.y, c4.x, sh); v[0] = __builtin_amdgcn_alignbit(c4.x, prv, sh);
                }
                if (d0 > 0 && d0 + 4 < nd) {
                    const u32x4_a4 t = { v[0], v[1], v[2], v[3] };
                    *(u32x4_a4 *)(gout + d0) = t;
                } else {
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) {
                        const uint32_t d = d0 + k;
                        if (d < nd) {
                            const bool shared = (d == 0 && phase) || (d == nd - 1 && tailbits);
                            if (!shared) gout[d] = v[k];
                            if (d == 0) a.seams[2 * chunk] = v[k];
                            if (d == nd - 1) a.seams[2 * chunk + 1] = v[k];
                        }
                    }
                }
            }
            __syncthreads();
            // the buffer of the next chunk (its first bits are written two barriers from here)
            for (uint32_t d = tid; d < (nd + 4) / 4 && d < a0.slot_dw / 4; d += 256) ((uint4 *)outbuf)[d] = make_uint4(0, 0, 0, 0);
        }
        q = qn; tile = ntile; chunk = nchunk; valid = nvalid; payload = npayload; gblk = ngblk;
    }
}

// ---- single pass over SUPER-CHUNKS ----------------------------------------------------------------------------
// The look-back above costs a chunk more than its coding does (a few agent-scope round trips against 3-4 us of work).
// Here a persistent workgroup codes SC_K chunks in a row into one LDS buffer -- chunk after chunk at the bit the
// previous one ended: the buffer then holds that piece of the stream as it will stand in memory but for one shift --
// and looks back ONCE per super-chunk, over one word per super-chunk.  Super-chunks are taken from a counter.  The stream leaves LDS for its final place: no
// slots, no concatenate pass.  A super-chunk that outgrows the buffer (data that hardly compresses) leaves it in pieces:
// the first piece looks back without having published a count (the successors wait for the last piece, which knows
// the running total and publishes it directly).  Seam and offset tables are kept per chunk, as the other paths keep them:
// a boundary inside a piece gets the finished dword on both sides, so enc_seam_kernel rewrites what is already there.
// Measured (16384 x 16384 RGB): 0.87 ms against 0.32 + 0.16 ms for slots + concatenate, and 0.57 ms with the look-back
// taken out (wrong offsets, timing only): a workgroup that codes, waits and writes in turn keeps fewer chunks in flight
// than a grid of independent ones, whatever the look-back costs.  Kept behind QB3_SINGLE_PASS=2; not the default.
constexpr uint32_t SC_K = 6;
template <int B, bool RGB, uint64_t ORDER, bool STEP>
__global__ void __launch_bounds__(256, 3) enc_px_sc_kernel(const EncArgs a0) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *etab = (uint32_t *)smem;                      // 512 entries
    uint32_t *wsum = etab + 512;                            // 64 dwords: scan scratch; [40..42]: the piece's offset (lo, hi), state
    uint32_t *rel = wsum + 64;                              // 16 dwords: where each chunk of the piece starts in the buffer
    uint32_t *segrel = rel + 16;                            // 48 dwords: ... and each index segment that starts in the piece
    uint32_t *outbuf = segrel + 48;                         // sc_cap_dw (+4) dwords
    const uint32_t cap_dw = a0.sc_cap_dw, cap_bits = cap_dw * 32;
    const uint4 tabv = ((const uint4 *)px_enc_tab.e)[tid & 127];
    for (uint32_t i = tid; i < cap_dw / 4 + 1; i += 256) ((uint4 *)outbuf)[i] = make_uint4(0, 0, 0, 0);
    const uint32_t etab_off = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)smem;
    const uint32_t nchunks = a0.nchunks, nsc = (nchunks + SC_K - 1) / SC_K;
    const uint64_t nsq = (uint64_t)nsc * a0.ntiles;        // super-chunks of all tiles, tile-major
    const uint32_t sb = a0.g.seg_blocks, nblocks = (uint32_t)a0.g.nblocks;

    auto chunk_lanes = [&](uint32_t chunk, bool &valid, bool &payload, uint32_t &gblk) {
        const int64_t gs = (int64_t)chunk * 255 - 1 + tid;  // lane 0 is the halo block
        valid = gs >= 0 && gs < (int64_t)nblocks; payload = valid && tid >= 1;
        gblk = valid ? (uint32_t)gs : 0u;
    };
    // super-chunks are handed out by a counter, in order: whoever holds one is running, and waits only for lower ones --
    // no assumption about how many workgroups of the grid are resident
    uint32_t *ticket = (uint32_t *)&a0.lookback[nchunks + 1];
    if (tid == 0) wsum[44] = atomicAdd(ticket, 1u);
    __syncthreads();
    uint64_t s = wsum[44];
    uint32_t w[4][B], pd = 0;
    uint32_t tile = 0, sc = 0, gblk = 0;
    bool valid = false, payload = false, first = true;
    if (s < nsq) {
        tile = (uint32_t)(s / nsc); sc = (uint32_t)(s - (uint64_t)tile * nsc);
        chunk_lanes(sc * SC_K, valid, payload, gblk);
        px_load_block<B, ORDER>(enc_for_tile(a0, tile), valid, gblk, w, pd);
    }
    while (s < nsq) {
        const EncArgs a = enc_for_tile(a0, tile);
        uint64_t *lb = a.lookback;
        const uint32_t c0 = sc * SC_K, c1 = min(c0 + SC_K, nchunks);
        uint32_t acc = 0, done = 0, p0 = c0;               // bits in the buffer, bits of earlier pieces, the piece's first chunk
        uint64_t excl = 0;
        bool have_excl = false;

        // the piece [p0, pend) leaves the buffer
        auto flush = [&](uint32_t pend, bool final) {
            __syncthreads();                               // its bits, rel[] and segrel[] are in LDS
            if (!have_excl) {
                if (wave == 0) {
                    if (final && lane == 0 && sc) lb_store(&lb[sc], (1ull << 62) | acc);
                    uint64_t e = 0;
                    const uint32_t state = sc == 0 ? 2u : lb_look(lb, sc, lane, e);
                    if (lane == 0) {
                        if (final) lb_store(&lb[sc], (state == 3 ? 3ull << 62 : 2ull << 62) | ((e + acc) & LB_VAL));
                        wsum[40] = (uint32_t)e; wsum[41] = (uint32_t)(e >> 32); wsum[42] = state;
                    }
                }
                __syncthreads();
                excl = ((uint64_t)wsum[41] << 32) | wsum[40];
                if (wsum[42] == 3 && tid == 0) lb[nchunks] = 1;          // abort flag: enc_seam_kernel hands it to the host
                have_excl = true;
            } else if (final && tid == 0)
                lb_store(&lb[sc], (wsum[42] == 3 ? 3ull << 62 : 2ull << 62) | ((excl + done + acc) & LB_VAL));
            const uint64_t E0 = excl + done, G0 = (uint64_t)a.out_bit0 + E0;
            const uint32_t phase = (uint32_t)G0 & 31, sh = (32 - phase) & 31;
            const uint32_t nd = (phase + acc + 31) >> 5, tailbits = (phase + acc) & 31;
            uint32_t *gout = a.out32 + (G0 >> 5);
            auto shifted = [&](uint32_t d) {
                const uint32_t cur = outbuf[d], prv = d ? outbuf[d - 1] : 0u;
                return phase ? __builtin_amdgcn_alignbit(cur, prv, sh) : cur;
            };
            // four output dwords a lane (one 16-byte store); the dwords the piece shares with its neighbours are left to the seam pass
            for (uint32_t d0 = 4 * tid; d0 < nd; d0 += 1024) {
                const uint4 c4 = *(const uint4 *)(outbuf + d0);            // (the buffer is zero behind the piece's last dword)
                const uint32_t prv = d0 ? outbuf[d0 - 1] : 0u;
                uint32_t v[4] = { c4.x, c4.y, c4.z, c4.w };
                if (phase) {
                    v[3] = __builtin_amdgcn_alignbit(c4.w, c4.z, sh); v[2] = __builtin_amdgcn_alignbit(c4.z, c4.y, sh);
                    v[1] = __builtin_amdgcn_alignbit(c4.y, c4.x, sh); v[0] = __builtin_amdgcn_alignbit(c4.x, prv, sh);
                }
                if (d0 > 0 && d0 + 4 < nd) {
                    const u32x4_a4 t = { v[0], v[1], v[2], v[3] };
                    *(u32x4_a4 *)(gout + d0) = t;
                } else {
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) {
                        const uint32_t d = d0 + k;
                        if (d < nd && !((d == 0 && phase) || (d == nd - 1 && tailbits))) gout[d] = v[k];
                    }
                }
            }
            // per chunk: its offset and the dwords at its two ends, as the seam pass expects them
            const uint32_t np = pend - p0;
            if (tid < np) {
                const uint32_t chunk = p0 + tid, r0 = rel[tid], r1 = tid + 1 < np ? rel[tid + 1] : acc;
                a.chunk_off[chunk] = E0 + r0;
                a.seams[2 * chunk] = shifted((phase + r0) >> 5);
                a.seams[2 * chunk + 1] = shifted((phase + r1 - 1) >> 5);
                if (chunk == nchunks - 1) a.group_sum[(nchunks + SCAN_GROUP - 1) / SCAN_GROUP] = E0 + r1;
            }
            if (a.have_idx) {       // the segments that start in the piece
                const uint32_t seg0 = (uint32_t)(((uint64_t)p0 * 255 + sb - 1) / sb);
                const uint32_t seg1 = (uint32_t)((min((uint64_t)pend * 255, (uint64_t)nblocks) + sb - 1) / sb);
                if (seg0 + tid < seg1) a.idx.bitpos[seg0 + tid] = E0 + segrel[tid];
            }
            __syncthreads();
            for (uint32_t d = tid; d < (nd + 4) / 4 && d < cap_dw / 4 + 1; d += 256) ((uint4 *)outbuf)[d] = make_uint4(0, 0, 0, 0);
            done += acc; acc = 0; p0 = pend;
            __syncthreads();
        };

        uint64_t sn = s;
        uint32_t ntile = tile, nsc_i = sc;
        for (uint32_t chunk = c0; chunk < c1; chunk++) {
            uint32_t total, pos, seg, bit0 = 0;
            const bool last = chunk + 1 == c1;
            if (last && tid == 0) wsum[44] = atomicAdd(ticket, 1u);      // the next super-chunk (read behind the coder's barriers)
            // the next chunk's pixels are asked for before this one is coded (inside a super-chunk nothing else would hide their
            // round trip); those of the next super-chunk's first chunk behind it, beside the look-back
            uint32_t wn[4][B], pdn = 0, ngblk = 0;
            bool nvalid = false, npayload = false;
            constexpr bool EARLY = B < 4;          // (four bands: the second set of pixel registers would spill)
            if (EARLY && !last) {
                chunk_lanes(chunk + 1, nvalid, npayload, ngblk);
                px_load_block<B, ORDER>(a, nvalid, ngblk, wn, pdn);
            }
            auto place = [&](uint32_t t) {
                if (acc + t > cap_bits) flush(chunk, false);
                bit0 = acc;
                if (tid == 0) rel[chunk - p0] = acc;
                acc += t;
                return bit0;
            };
            px_code_chunk<B, RGB, ORDER, STEP, true>(a, a0, chunk, valid, payload, gblk, w, pd, etab, wsum, outbuf, etab_off, first, tabv,
                                                     total, pos, seg, place);
            first = false;
            if (seg != ~0u) segrel[seg - (uint32_t)(((uint64_t)p0 * 255 + sb - 1) / sb)] = bit0 + pos;
            if (!EARLY && !last) {
                chunk_lanes(chunk + 1, valid, payload, gblk);
                px_load_block<B, ORDER>(a, valid, gblk, w, pd);
            } else if (!last) {
                valid = nvalid; payload = npayload; gblk = ngblk; pd = pdn;
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int k = 0; k < B; k++) w[r][k] = wn[r][k];
            } else {
                sn = wsum[44];
                if (sn < nsq) {
                    ntile = (uint32_t)(sn / nsc); nsc_i = (uint32_t)(sn - (uint64_t)ntile * nsc);
                    chunk_lanes(nsc_i * SC_K, valid, payload, gblk);
                    px_load_block<B, ORDER>(enc_for_tile(a0, ntile), valid, gblk, w, pd);
                }
            }
        }
        flush(c1, true);
        s = sn; tile = ntile; sc = nsc_i;
    }
}

// dispatch over the compile-time parameters
template <int B, bool RGB>
static void launch_enc_px_b(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    const bool step = a.g.mode != CM_FTL, z = a.g.order == ZCURVE;
    if (a.single_pass || plan.persistent) {
        // persistent grid: every workgroup must be resident (look-back waits on lower chunks only, and those belong to
        // workgroups of the same grid): what the occupancy query admits per CU, at most 8, times the CUs
        auto launch = [&](auto kernel) {
            static int per_cu = 0, cus = 0;
            if (!per_cu) {
                int dev = 0;
                hipDeviceProp_t prop;
                if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { cus = 64; per_cu = 1; }
                else {
                    cus = prop.multiProcessorCount;
                    int n = 0;
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 256, plan.lds_bytes) != hipSuccess || n < 1) n = 1;
                    per_cu = n > 8 ? 8 : n;
                }
            }
            const uint64_t units = a.single_pass == 2 ? (plan.nchunks + SC_K - 1) / SC_K : plan.nchunks;
            const uint64_t nq = units * a.ntiles, cap = (uint64_t)per_cu * cus;
            hipLaunchKernelGGL(kernel, dim3((uint32_t)(nq < cap ? nq : cap)), dim3(256), plan.lds_bytes, st, a);
        };
        if (a.single_pass == 2) {
            if (!z && !step) launch(enc_px_sc_kernel<B, RGB, HILBERT, false>);
            else if (!z && step) launch(enc_px_sc_kernel<B, RGB, HILBERT, true>);
            else if (z && !step) launch(enc_px_sc_kernel<B, RGB, ZCURVE, false>);
            else launch(enc_px_sc_kernel<B, RGB, ZCURVE, true>);
        } else if (a.single_pass) {
            if (!z && !step) launch(enc_px_sp_kernel<B, RGB, HILBERT, false, true>);
            else if (!z && step) launch(enc_px_sp_kernel<B, RGB, HILBERT, true, true>);
            else if (z && !step) launch(enc_px_sp_kernel<B, RGB, ZCURVE, false, true>);
            else launch(enc_px_sp_kernel<B, RGB, ZCURVE, true, true>);
        } else {
            if (!z && !step) launch(enc_px_sp_kernel<B, RGB, HILBERT, false, false>);
            else if (!z && step) launch(enc_px_sp_kernel<B, RGB, HILBERT, true, false>);
            else if (z && !step) launch(enc_px_sp_kernel<B, RGB, ZCURVE, false, false>);
            else launch(enc_px_sp_kernel<B, RGB, ZCURVE, true, false>);
        }
        return;
    }
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
