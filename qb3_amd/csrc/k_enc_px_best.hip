// qb3_amd/csrc/k_enc_px_best.hip -- 8-bit grey / RGB / RGBA encoder for the common-factor modes (QB3M_BEST family), lane per block
//
// Reference: encode_best (QB3encode.h:617-724), cfgenc (:283-361), ienc (:557-613), gcf (:98-126).
// The front end is enc_px_kernel's (qb3_px_enc.h): a lane owns a block, its bands' sixteen mag-sign deltas sit four to a
// register.  What the common-factor modes add per unit is a decision -- plain, common factor, or index form -- and on real
// data nearly every unit settles it at once: a magnitude of 1 among the sixteen means the factor is 1, more than eight
// distinct values (a lower bound from a 32-bit bitmap) means no index form, and the unit is coded exactly as QB3M_BASE
// codes it, with the pieces the lane has already built.  The few units left (2-3 % on noisy data: no magnitude of 1, so
// Euclid has to run; or few distinct values) are HARD.  A wave pays for its slowest lane, so hard units are not analysed
// where they are found: their lanes put them in a queue in LDS and the workgroup's lanes take one queue entry each -- the
// divergent analysis (qb3_best.h, shared with the unit-per-lane kernel) runs densely packed, once per 256 hard units
// instead of in nearly every wave.  The same again for emission: units that end up in common-factor or index form are
// queued with their bit position and written by dense lanes straight into the chunk's bit buffer (LDS atomic OR: a unit's
// bits can go in from any lane), the owners write the plain ones.
// The band's factor state (pcf) is a last-writer scan: per band one ballot per wave plus a byte per lane.  Across chunks:
// k_enc_best.hip's scheme unchanged (every chunk coded assuming the starting state, best_scan_kernel, the chunks where
// that mattered coded again).
#include "qb3_px_enc.h"
#include "qb3_best.h"

namespace qb3dev {

constexpr uint32_t PXB_CAP = 256;           // queue entries a round = lanes of the workgroup


// any byte of x equal to 1 or 2 (a mag-sign value of magnitude 1)
__device__ __forceinline__ uint32_t swar_has_mag1(uint32_t x) {
    const uint32_t u = swar_sub8(x, 0x01010101u) & 0xfefefefeu;     // byte - 1, low bit dropped: zero for 1 and 2 (0 gives fe)
    return (u - 0x01010101u) & ~u & 0x80808080u;
}
constexpr uint32_t odd_multiples_mask(uint32_t p) { uint32_t m = 0; for (uint32_t a = p; a <= 16; a += p) m |= 1u << (2 * a - 1); return m; }   // bit 2a - 1 for the multiples a of p
// wave-aggregated append: lanes with `want` get consecutive queue indices
__device__ __forceinline__ uint32_t queue_take(uint32_t *counter, bool want) {
    const uint64_t m = __ballot(want);
    if (!m) return ~0u;
    const uint32_t lane = threadIdx.x & 63, leader = (uint32_t)__builtin_ctzll(m);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
    const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    return want ? base + before : ~0u;
}

// ---- byte-parallel helpers for the dense lanes: a unit is four dwords of mag-sign bytes all the way
__device__ __forceinline__ uint32_t swar_eq8(uint32_t x, uint32_t y) {            // 0xff in every byte where x and y agree
    const uint32_t z = x ^ y, t = ~(((z & 0x7f7f7f7fu) + 0x7f7f7f7fu) | z | 0x7f7f7f7fu);        // 0x80 where the byte is zero
    return (t >> 7) * 0xffu;
}
__device__ __forceinline__ uint32_t swar_le7(uint32_t a, uint32_t b) {            // 0x80 in every byte where a <= b (bytes below 128)
    return ((b | 0x80808080u) - a) & 0x80808080u;
}
__device__ __forceinline__ uint32_t byte_of(const uint32_t (&G)[4], int i) { return (G[i >> 2] >> (8 * (i & 3))) & 0xffu; }
// ... with a run-time position (a select chain, no indexed register array): the dense lanes' loops are NOT unrolled, so that
// the rare forms cost the kernel few registers
__device__ __forceinline__ uint32_t byte_at(const uint32_t (&G)[4], uint32_t i) {
    const uint64_t lo = ((uint64_t)G[1] << 32) | G[0], hi = ((uint64_t)G[3] << 32) | G[2];      // (64-bit shifts: the compiler turns a select chain over G[] into a scratch array)
    return (uint32_t)((i < 8 ? lo : hi) >> (8 * (i & 7))) & 0xffu;
}

// The distinct values of a unit by descending count, first seen first among equals (the reference's stable insertion sort,
// QB3encode.h:546-554): per position, byte-parallel, the count of its value, the value's first position, the value's rank
__device__ __forceinline__ void pxb_counts(const uint32_t (&G)[4], uint32_t (&cnt)[4], uint32_t (&fp)[4]) {
#pragma unroll
    for (int q = 0; q < 4; q++) { cnt[q] = 0; fp[q] = 0; }
#pragma unroll 1
    for (int j = 15; j >= 0; j--) {
        const uint32_t bj = byte_at(G, (uint32_t)j) * 0x01010101u;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t m = swar_eq8(G[q], bj);
            cnt[q] += m & 0x01010101u;
            fp[q] = (fp[q] & ~m) | ((uint32_t)j * 0x01010101u & m);
        }
    }
}
__device__ __forceinline__ void pxb_ranks(const uint32_t (&G)[4], const uint32_t (&cnt)[4], const uint32_t (&fp)[4], uint32_t (&rank)[4]) {
#pragma unroll
    for (int q = 0; q < 4; q++) rank[q] = 0;
#pragma unroll 1
    for (uint32_t j = 0; j < 16; j++) {
        if (byte_at(fp, j) == j) {                  // position j holds the first occurrence of a value: it beats ...
            const uint32_t bj = byte_at(G, j) * 0x01010101u, cj = byte_at(cnt, j) * 0x01010101u, jj = j * 0x01010101u;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t ne = ~swar_eq8(G[q], bj);                                // ... other values
                const uint32_t gt = ~swar_le7(cj, cnt[q]) & 0x80808080u;               // with a smaller count,
                const uint32_t eqc = swar_eq8(cnt[q], cj) & ~swar_le7(fp[q], jj);      // or the same count and a later first position
                rank[q] += ((ne & (gt | eqc)) >> 7) & 0x01010101u;
            }
        }
    }
}
// the group divided by the factor, sign kept (magsdiv, QB3encode.h:95): exact through the float reciprocal
__device__ __forceinline__ void pxb_divide(const uint32_t (&G)[4], uint32_t cf, uint32_t (&D)[4]) {
    const float rcf = __builtin_amdgcn_rcpf((float)cf);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t d4 = 0;
#pragma unroll 1
        for (uint32_t i = 0; i < 4; i++) {
            const uint32_t g = (G[k] >> (8 * i)) & 0xffu, q = (uint32_t)((float)((g >> 1) + (g & 1)) * rcf + 0.5f);
            d4 |= (((q << 1) - (g & 1)) & 0xffu) << (8 * i);
        }
        D[k] = d4;
    }
}

// ---- cooperative analysis of the hard units: SIXTEEN LANES A UNIT, a lane a value.  What this phase costs the workgroup is
// the latency of one unit's analysis (everybody waits for it), so the unit is spread out: reductions over a row of sixteen
// lanes are four DPP instructions, comparing every value with every other is fifteen row rotations.
template <uint32_t CTRL> __device__ __forceinline__ uint32_t dpp_mov(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false); }
// all-reduce over the sixteen lanes of a DPP row: pairs, quads, eights (half mirror), the row (mirror)
#define ROW_ALLREDUCE(x, OP) do { x = OP(x, dpp_mov<0xB1>(x)); x = OP(x, dpp_mov<0x4E>(x)); x = OP(x, dpp_mov<0x141>(x)); x = OP(x, dpp_mov<0x140>(x)); } while (0)
__device__ __forceinline__ uint32_t op_min(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t op_or(uint32_t a, uint32_t b) { return a | b; }
__device__ __forceinline__ uint32_t op_add(uint32_t a, uint32_t b) { return a + b; }

// One unit per row: v = the lane's mag-sign value (lane i of the row: value i), meta = rung | oldrung << 4 | idx_may << 8.
// Results as best_analyse's (qb3_best.h; reference encode_best QB3encode.h:617-724), the same in every lane of the row:
// r0 = cf | trung << 8 | writer << 12, r1 = szBase | szCf << 16, r2 = size of the index form (all ones: none)
__device__ __forceinline__ void pxb_analyse_row(uint32_t v, uint32_t meta, bool on, uint32_t &r0, uint32_t &r1, uint32_t &r2) {
    constexpr uint32_t UB = 3, UMASK = 7;
    typedef uint8_t T;
    const uint32_t lane = threadIdx.x & 63, i = lane & 15, row = lane >> 4;
    const uint32_t rung = meta & 15u, oldrung = (meta >> 4) & 15u;
    const bool idx_may = (meta >> 8) & 1u;
    // gcd of the non-zero magnitudes (QB3encode.h:98-126): Euclid on the whole row at once -- x = the smallest, every value
    // modulo x, the smallest remainder is the next x (x itself stays in the set), until every remainder is zero
    uint32_t a = (v >> 1) + (v & 1), x = a ? a : 255u;
    ROW_ALLREDUCE(x, op_min);
    bool done = !on || x == 1;
    while (__any(!done)) {
        uint32_t r = mod_t<T>((T)a, (T)x), m = r ? r : 255u;
        ROW_ALLREDUCE(m, op_min);
        if (!done) {
            if (m == 255u) done = true;                 // every value is a multiple of x
            else { a = (r == 0 && a == x) ? x : r; x = m; done = x == 1; }
        }
    }
    const uint32_t cf = on ? x : 1u, thr = 45 + 2 * rung;
    uint32_t szBase = 0, szCf = 0, trung = 0, size = 0;
    if (__any(cf >= 2)) {
        uint32_t d = cf >= 2 ? (uint32_t)mdiv_t<T>((T)v, (T)cf) : 0u, usedd = d;
        ROW_ALLREDUCE(usedd, op_or);
        trung = topbit32(usedd | 1);
        // the step (QB3encode.h:169-176): the rung bits of the row's sixteen values are sixteen bits of a ballot
        const uint32_t rowbits = (uint32_t)(__ballot((d >> trung) & 1u) >> (16 * row)) & 0xffffu;
        if (trung && rowbits && (rowbits & (rowbits + 1)) == 0 && i + 1 == (uint32_t)__popc(rowbits)) d ^= 1u << trung;
        const uint32_t top = 1u << trung, half = top >> 1;
        uint32_t xx = d;
        if (xx == top || xx == top - 1) xx ^= 2 * top - 1;         // the middle swap (all rungs of 8-bit data)
        uint32_t grp = trung + (xx >= half) + (xx >= top);
        ROW_ALLREDUCE(grp, op_add);
        if (!trung) grp = 16;
        szBase = (UB + 2) + sw_noflag_len<UB>(trung - oldrung) + 1 + grp;
        const T cfm = (T)(cf - 2);
        const uint32_t cfrung = topbit_t<T>(cfm);
        if (trung >= cfrung && (trung < cfrung + UB || cfrung == 0)) szCf = 1 + vlen_t<T>(cfm, trung);
        else szCf = cs_len<UB>((cfrung - trung) & UMASK) + vlen_t<T>((T)(cfm ^ (T)((T)1 << cfrung)), cfrung - 1);
        if (cf < 2) { szBase = 0; szCf = 0; trung = 0; }
        size = szBase + szCf;
    }
    // index form (QB3encode.h:557-613, tried per :702; a plain unit above rung 3 always reaches the threshold)
    uint32_t idx = 0xffffffffu;
    const bool want = on && idx_may && rung > 3 && (cf < 2 || size >= thr);
    if (__any(want)) {
        // the count of the lane's value and its first position in the row: fifteen rotations, the value from i - j each
        uint32_t cnt = 1, fp = i;
        const uint32_t vi = v | (i << 8);           // (the position travels with the value: no per-rotation constants to keep)
        // (a scheduling barrier after every rotation: left alone the compiler issues all fifteen at once and keeps their results live)
#define PXB_ROT1(j) { const uint32_t w = dpp_mov<0x120 + j>(vi), src = w >> 8; const bool eq = (w & 0xffu) == v; cnt += eq; fp = (eq && src < fp) ? src : fp; __builtin_amdgcn_sched_barrier(0); }
        PXB_ROT1(1) PXB_ROT1(2) PXB_ROT1(3) PXB_ROT1(4) PXB_ROT1(5) PXB_ROT1(6) PXB_ROT1(7) PXB_ROT1(8)
        PXB_ROT1(9) PXB_ROT1(10) PXB_ROT1(11) PXB_ROT1(12) PXB_ROT1(13) PXB_ROT1(14) PXB_ROT1(15)
#undef PXB_ROT1
        const bool first = fp == i;
        const uint32_t ndist = (uint32_t)__popc((uint32_t)(__ballot(first) >> (16 * row)) & 0xffffu);
        // rank of the lane's value: the distinct values (at their first positions) with a larger count, or the same count and
        // an earlier first position (the reference's stable sort by descending count, QB3encode.h:546-554)
        const uint32_t P = v | (cnt << 8) | (i << 16) | ((uint32_t)first << 24);
        uint32_t rank = 0;
#define PXB_ROT2(j) { const uint32_t Q = dpp_mov<0x120 + j>(P), qc = (Q >> 8) & 0xffu; \
                      rank += (Q >> 24) && (Q & 0xffu) != v && (qc > cnt || (qc == cnt && ((Q >> 16) & 0xffu) < fp)); __builtin_amdgcn_sched_barrier(0); }
        PXB_ROT2(1) PXB_ROT2(2) PXB_ROT2(3) PXB_ROT2(4) PXB_ROT2(5) PXB_ROT2(6) PXB_ROT2(7) PXB_ROT2(8)
        PXB_ROT2(9) PXB_ROT2(10) PXB_ROT2(11) PXB_ROT2(12) PXB_ROT2(13) PXB_ROT2(14) PXB_ROT2(15)
#undef PXB_ROT2
        uint32_t ibits = 2 + (rank >= 2) + (rank >= 4), vbits = first ? vlen_t<T>((T)v, rung) : 0u;     // a rank's code: 2, 3 or 4 bits
        ROW_ALLREDUCE(ibits, op_add);
        ROW_ALLREDUCE(vbits, op_add);
        if (want && ndist <= 8) idx = (UB + 2) + sw_noflag_len<UB>(UMASK - oldrung) + sw_noflag_len<UB>(rung - oldrung) + ibits + vbits;
    }
    const bool writer = cf >= 2 && !(size >= thr && idx < size);
    r0 = cf | (trung << 8) | ((uint32_t)writer << 12);
    r1 = szBase | (szCf << 16);
    r2 = idx;
}

// Dense emission of one queued unit in common-factor (kind 2) or index (kind 3) form at bit `pos` of the chunk's buffer
// (reference cfgenc QB3encode.h:283-361, ienc :557-613).  tb0: byte address of the code table in LDS.
__device__ __forceinline__ void pxb_emit(uint32_t *outbuf, uint32_t pos, const uint32_t (&G)[4], uint32_t rung, uint32_t oldrung, uint32_t kind,
                                         bool same, uint32_t trung, uint32_t cf, uint32_t tb0) {
    constexpr uint32_t UB = 3, UMASK = 7;
    typedef uint8_t T;
    LdsWriter w;
    w.init(outbuf, pos);
    if (kind == 2) {
        uint32_t D[4];
        pxb_divide(G, cf, D);
        const T cfm = (T)(cf - 2);
        const uint32_t cfrung = topbit_t<T>(cfm);
        put_signal<UB>(w);
        put_sw_noflag<UB>(w, trung - oldrung);
        if (!same) {
            w.put(1, 1);
            if (trung >= cfrung && (trung < cfrung + UB || cfrung == 0)) { w.put(0, 1); put_single<T>(w, cfm, trung); }
            else {
                const uint32_t dl = (cfrung - trung) & UMASK;
                w.put(cs_code<UB>(dl), cs_len<UB>(dl));         // its change flag doubles as the "own rung" marker
                put_single<T>(w, (T)(cfm ^ (T)((T)1 << cfrung)), cfrung - 1);
            }
        } else w.put(0, 1);
        uint32_t pc[6] = {0, 0, 0, 0, 0, 0}, pl[6] = {0, 0, 0, 0, 0, 0};
        if (trung == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) pc[0] |= (byte_of(D, i) & 1u) << i;
            pl[0] = 16;
        } else px_unit_pieces<true>(D, trung, 0, 0, tb0 + (8u << trung), pc, pl);        // (step applied inside)
#pragma unroll
        for (int k = 0; k < 6; k++) w.put(pc[k], pl[k]);
    } else {
        uint32_t cnt[4], fp[4], rank[4];
        pxb_counts(G, cnt, fp);
        pxb_ranks(G, cnt, fp, rank);
        put_signal<UB>(w);
        put_sw_noflag<UB>(w, UMASK - oldrung);
        put_sw_noflag<UB>(w, rung - oldrung);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t acc = 0, al = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t j = (rank[q] >> (8 * i)) & 0xffu;         // plain rung-2 code of j (0..7): lengths {2,2,3,3,4,4,4,4}
                const uint32_t code = j < 2 ? (j << 1) : j < 4 ? (((j - 2) << 2) | 1) : (((j - 4) << 2) | 3);
                acc |= code << al; al += 2 + (j >= 2) + (j >= 4);
            }
            w.put(acc, al);
        }
        // the distinct values in rank order
#pragma unroll 1
        for (uint32_t r = 0; r < 8; r++) {
            uint32_t v = 0, have = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t m = swar_eq8(rank[q], r * 0x01010101u) & swar_eq8(fp[q], 0x03020100u + 0x04040404u * q);
                v |= G[q] & m; have |= m;
            }
            v |= v >> 16; v |= v >> 8;
            if (have) put_single<T>(w, (T)v, rung);
        }
    }
    w.finish();
}

// LDS carve of the kernel (bytes from the start of dynamic LDS)
constexpr uint32_t PXB_WSUM = 2048, PXB_WMASK = PXB_WSUM + 256, PXB_USED = PXB_WMASK + 128, PXB_WVALS = PXB_USED + 16,
                   PXB_HQ = PXB_WVALS + 1024, PXB_HR = PXB_HQ + 5 * 4 * PXB_CAP, PXB_OUT = PXB_HR + 3 * 4 * PXB_CAP;
static_assert(PXB_OUT == PXB_LDS_FIXED, "plan_encode sizes the kernel's LDS from PXB_LDS_FIXED");

template <int B, bool RGB, uint64_t ORDER, bool FIRST>
__device__ __forceinline__ void px_best_chunk(const EncArgs &a, const EncArgs &a0, uint8_t *smem, uint32_t chunk, bool summary_only) {
    constexpr uint32_t UMASK = 7;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nblocks = (uint32_t)a.g.nblocks;
    uint32_t *etab = (uint32_t *)smem;                          // 512 entries: px_enc_tab
    uint32_t *wsum = (uint32_t *)(smem + PXB_WSUM);             // 64 dwords: [0..3] scan, [32..35] last rungs, [40] hard units, [41] units to emit densely
    uint64_t *wmask = (uint64_t *)(smem + PXB_WMASK);           // [band][wave]: ballot of factor writers
    uint32_t *used_entry = (uint32_t *)(smem + PXB_USED);       // [band]: a unit looked at the factor entering the chunk
    uint8_t *wvals = smem + PXB_WVALS;                          // [band][lane of the workgroup]: the factor written (cf - 2)
    uint32_t *hq = (uint32_t *)(smem + PXB_HQ);                 // [5][PXB_CAP]: four dwords of mag-sign values, one of rungs and flags
    uint32_t *hr = (uint32_t *)(smem + PXB_HR);                 // [3][PXB_CAP]: analysis results / bit position
    uint32_t *outbuf = (uint32_t *)(smem + PXB_OUT);            // slot_dw dwords
    const uint4 tabv = ((const uint4 *)px_enc_tab.e)[tid & 127];
    for (uint32_t i = tid; i < a.slot_dw / 4; i += 256) ((uint4 *)outbuf)[i] = make_uint4(0, 0, 0, 0);
    if (tid == 0) { wsum[40] = 0; wsum[41] = 0; }
    if (FIRST && tid < B) used_entry[tid] = 0;
    const uint32_t etab_off = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)smem;

    const int64_t gs = (int64_t)chunk * 255 - 1 + tid;          // lane 0 is the halo block
    const bool valid = gs >= 0 && gs < (int64_t)nblocks, payload = valid && tid >= 1;
    const uint32_t gblk = valid ? (uint32_t)gs : 0u;
    uint32_t w[4][B], pd;
    px_load_block<B, ORDER>(a, valid, gblk, w, pd);
    PxFront<B> f;
    px_front<B, RGB, ORDER>(a0, gblk, w, pd, etab, wsum, tabv, f);      // (one barrier)
    const uint32_t rp_packed = f.rp_packed, prp = f.prp;
    uint32_t usedp = 0, lastp = 0, pvp = 0;     // the bands' used / leaving / entering values, a byte each
#pragma unroll
    for (int c = 0; c < B; c++) { usedp |= f.usedv[c] << (8 * c); lastp |= f.lastv[c] << (8 * c); pvp |= f.pvv[c] << (8 * c); }

    // ---- per band: the plain form of the unit (QB3M_BASE's, from the lane's registers), and is the unit settled by it.
    // One bitmap of the sixteen mag-sign values answers both questions while they are below 32 (rungs up to 4).
    // Common factor: magnitude a is values 2a - 1 and 2a, so bit 2a - 1 of bm | bm >> 1 says "a occurs"; a factor exists
    // exactly when some prime divides every magnitude that occurs: six mask tests (magnitudes up to 16).  From rung 5 up a
    // magnitude of 1 settles it, else the unit is hard.
    // Index form (QB3encode.h:557-613, tried per :702 for rungs above 3 -- where the plain size, at least 16 * rung + 1,
    // always reaches the threshold 45 + 2 * rung): n distinct values cost at least the head, 32 + max(n - 2, 0) +
    // max(n - 4, 0) bits of index codes (every count beyond the first value's is 1) and the values' own codes -- exact from the
    // bitmap up to rung 4 -- and only a unit whose plain size exceeds that goes to the analysis.
    // (state that lives across the analysis is kept small: the kernel's registers set how many workgroups a CU holds)
    uint32_t pc[B][6], plp[B];                  // plp: the six piece lengths, five bits each (a piece is at most 27 bits)
    uint32_t qip = 0, qip3 = 0, idxm = 0;       // qip: the places of bands 0 .. 2 in the queue of hard units, ten bits each (0x3ff: none); qip3: band 3's
    auto qi_of = [&](int c) -> uint32_t { const uint32_t q = c < 3 ? (qip >> (10 * c)) & 0x3ffu : qip3; return q == 0x3ffu ? ~0u : q; };
    auto plain_len = [&](int c) -> uint32_t { uint32_t n = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) n += (plp[c] >> (5 * k)) & 31u;
        return n; };
#pragma unroll
    for (int c = 0; c < B; c++) {
        uint32_t pl[6];
#pragma unroll
        for (int k = 0; k < 6; k++) { pc[c][k] = 0; pl[k] = 0; }
        uint32_t lenN = 0;
        bool hard = false;
        const uint32_t rung = (rp_packed >> (4 * c)) & 15u, prung = (prp >> (4 * c)) & 15u, used = (usedp >> (8 * c)) & 0xffu;
        if (payload) {
            const uint32_t delta = (rung - prung) & UMASK;
            const uint32_t csl = __builtin_amdgcn_ubfe(cs3_lens(), 4 * delta, 4), csc = (uint32_t)(cs3_codes() >> (8 * delta)) & 0xffu;
            if (used <= 1) lenN = px_unit_low(f.gp[c], used, csl, csc, pc[c], pl);
            else {
                lenN = px_unit_pieces<true>(f.gp[c], rung, csl, csc, etab_off + (8u << rung), pc[c], pl);
                uint32_t bm = 0;
#pragma unroll
                for (int i = 0; i < 16; i++) bm |= 1u << ((f.gp[c][i >> 2] >> (8 * (i & 3))) & 31u);
                const uint32_t occ = (bm | (bm >> 1)) & 0xaaaaaaaau;        // bit 2a - 1: magnitude a occurs (a = 1 .. 16)
                bool factor = false;
                constexpr uint32_t primes[6] = {2, 3, 5, 7, 11, 13};
#pragma unroll
                for (int k = 0; k < 6; k++) factor = factor || (occ & ~odd_multiples_mask(primes[k])) == 0;
                if (__any(rung > 4)) {
                    const uint32_t has1 = swar_has_mag1(f.gp[c][0]) | swar_has_mag1(f.gp[c][1]) | swar_has_mag1(f.gp[c][2]) | swar_has_mag1(f.gp[c][3]);
                    if (rung > 4) factor = !has1;
                }
                const uint32_t n = __popc(bm);             // distinct values: exact up to rung 4, else a lower bound
                uint32_t vb = n * rung;                     // their own codes: rung bits each, one more from half the range up, two more in the top half (middle swap: QB3encode.h:30-33)
                if (rung == 4) vb += __popc(bm & 0xffffff00u) + __popc(bm & 0xfffe8000u);
                const uint32_t head = 5 + sw_noflag_len<3>(UMASK - prung) + sw_noflag_len<3>(rung - prung);
                const uint32_t floor_ = head + 32 + (n > 2 ? n - 2 : 0) + (n > 4 ? n - 4 : 0) + vb;
                const bool idx_may = rung > 3 && n <= 8 && floor_ < lenN;
                hard = factor || idx_may;
                idxm |= (uint32_t)idx_may << c;
            }
        }
        plp[c] = pl[0] | (pl[1] << 5) | (pl[2] << 10) | (pl[3] << 15) | (pl[4] << 20) | (pl[5] << 25);
        const uint32_t qn = queue_take(&wsum[40], hard);       // (at most 765 hard units a chunk)
        if (c < 3) qip |= (qn & 0x3ffu) << (10 * c); else qip3 = qn & 0x3ffu;
        if (qn < PXB_CAP) {
#pragma unroll
            for (int q = 0; q < 4; q++) hq[q * PXB_CAP + qn] = f.gp[c][q];
            hq[4 * PXB_CAP + qn] = rung | (prung << 4) | (((idxm >> c) & 1u) << 8);
        }
    }
    __syncthreads();
    // ---- analysis of the hard units: cf | trung << 8 | writer << 12, szBase | szCf << 16, index size
    uint32_t res0[B], res1[B], res2[B];
#pragma unroll
    for (int c = 0; c < B; c++) { res0[c] = 1; res1[c] = 0; res2[c] = 0xffffffffu; }
    const uint32_t nh = wsum[40];
    for (uint32_t r0 = 0; r0 < nh; r0 += PXB_CAP) {
        if (r0) {
#pragma unroll
            for (int c = 0; c < B; c++)
                if (qi_of(c) != ~0u && qi_of(c) - r0 < PXB_CAP) {
                    const uint32_t j = qi_of(c) - r0;
#pragma unroll
                    for (int q = 0; q < 4; q++) hq[q * PXB_CAP + j] = f.gp[c][q];
                    hq[4 * PXB_CAP + j] = ((rp_packed >> (4 * c)) & 15u) | (((prp >> (4 * c)) & 15u) << 4) | (((idxm >> c) & 1u) << 8);
                }
            __syncthreads();
        }
        // sixteen units a pass: a row of sixteen lanes per unit, a lane per value (pxb_analyse_row)
        const uint32_t here = nh - r0 < PXB_CAP ? nh - r0 : PXB_CAP;
        for (uint32_t s0 = 0; s0 < here; s0 += 16) {
            const uint32_t j = s0 + (tid >> 4), i = tid & 15;
            const bool on = j < here;
            const uint32_t v = on ? (hq[(i >> 2) * PXB_CAP + j] >> (8 * (i & 3))) & 0xffu : 0u, meta = on ? hq[4 * PXB_CAP + j] : 0u;
            uint32_t x0, x1, x2;
            pxb_analyse_row(v, meta, on, x0, x1, x2);
            if (on && i == 0) { hr[j] = x0; hr[PXB_CAP + j] = x1; hr[2 * PXB_CAP + j] = x2; }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < B; c++)
            if (qi_of(c) != ~0u && qi_of(c) - r0 < PXB_CAP) {
                const uint32_t j = qi_of(c) - r0;
                res0[c] = hr[j]; res1[c] = hr[PXB_CAP + j]; res2[c] = hr[2 * PXB_CAP + j];
            }
        if (r0 + PXB_CAP < nh) __syncthreads();
    }

    // ---- who wrote the band's factor last: a ballot per band and wave, the value per lane
#pragma unroll
    for (int c = 0; c < B; c++) {
        const bool wr = payload && ((res0[c] >> 12) & 1u);
        const uint64_t m = __ballot(wr);
        if (lane == 0) wmask[c * 4 + wave] = m;
        if (wr) wvals[c * 256 + tid] = (uint8_t)((res0[c] & 0xffu) - 2);
    }
    __syncthreads();
    auto last_writer = [&](int c, int32_t upto, uint32_t &val) -> bool {      // among lanes 0..upto of the workgroup
        if (upto < 0) return false;
        for (int32_t wv = upto >> 6; wv >= 0; wv--) {
            uint64_t m = wmask[c * 4 + wv];
            if (wv == (upto >> 6)) m &= ~0ull >> (63 - (upto & 63));
            if (m) { val = wvals[c * 256 + wv * 64 + 63 - __clzll((long long)m)]; return true; }
        }
        return false;
    };
    if (FIRST && tid < B) {             // chunk summary: the last writer of each band among all the chunk's units
        uint32_t v = 0;
        const bool has = last_writer((int)tid, 255, v);
        a.cw_has[(uint64_t)chunk * B + tid] = (uint8_t)has; a.cw_val[(uint64_t)chunk * B + tid] = v;
    }
    if (FIRST && summary_only) return;  // (workgroup uniform)

    // ---- the coding of every unit and its length (QB3encode.h:679-713)
    const uint32_t seg = gblk / a.g.seg_blocks;
    const bool seg_start = payload && a.have_idx && seg * a.g.seg_blocks == gblk;
    uint32_t len[B], kind[B], pcfv[B], blen[1] = { 0 };       // kind: 0 low / 1 plain (the lane's pieces), 2 common factor, 3 index
    bool same[B];
#pragma unroll
    for (int c = 0; c < B; c++) {
        len[c] = 0; kind[c] = 0; same[c] = false; pcfv[c] = 0;
        if (payload) {
            const uint32_t rung = (rp_packed >> (4 * c)) & 15u, used = (usedp >> (8 * c)) & 0xffu, cf = res0[c] & 0xffu;
            if (cf >= 2 || seg_start) {
                // factor state entering this unit: the last writer before it in the chunk, else the chunk's entry state
                uint32_t v = 0;
                const bool mine = last_writer(c, (int32_t)tid - 1, v);
                pcfv[c] = mine ? v : (FIRST ? (uint32_t)a0.st.cf[c] & 0xffu : (uint32_t)a.centry[(uint64_t)chunk * B + c] & 0xffu);
                if (FIRST && !mine && cf >= 2) atomicOr(&used_entry[c], 1u);
                if (FIRST && seg_start) a.seg_from_entry[(uint64_t)seg * B + c] = (uint8_t)!mine;
            }
            uint32_t size = plain_len(c);
            if (used > 1) {
                const uint32_t thr = 45 + 2 * rung;
                kind[c] = 1;
                if (cf >= 2) { same[c] = (cf - 2) == pcfv[c]; size = (res1[c] & 0xffffu) + (same[c] ? 0u : res1[c] >> 16); kind[c] = 2; }
                if (size >= thr && res2[c] < size) { size = res2[c]; kind[c] = 3; }
            }
            len[c] = size;
            blen[0] += len[c];
        }
    }
    const uint32_t myblen = blen[0];
    block_exscan_dpp<1>(blen, wsum);            // (one barrier)
    const uint32_t pos = blen[0], total = wsum[0] + wsum[1] + wsum[2] + wsum[3];

    // ---- emission: plain units from the lane's pieces, the others queued for the dense lanes
    uint32_t qe[B], upos[B];
    {
        LdsWriter32 wr;
        uint32_t p = pos;
        wr.init(outbuf, p);
#pragma unroll
        for (int c = 0; c < B; c++) {
            upos[c] = p;
            const bool dense = payload && kind[c] >= 2;
            if (payload && !dense) {
#pragma unroll
                for (int k = 0; k < 6; k++) wr.put(pc[c][k], (plp[c] >> (5 * k)) & 31u);
            }
            p += len[c];
            if (dense) { wr.finish(); wr.init(outbuf, p); }     // (the unit's bits come from another lane)
            qe[c] = queue_take(&wsum[41], dense);
        }
        wr.finish();
    }
    if (payload) {
        if (gblk == nblocks - 1) {      // coder state on leaving the image (QB3encode.h:718-722; the band's final factor: best_scan_kernel)
#pragma unroll
            for (int c = 0; c < B; c++) { a.res->prev[c] = (lastp >> (8 * c)) & 0xffu; a.res->rung[c] = (rp_packed >> (4 * c)) & 15u; }
        }
        if (a.have_idx) {
            // the block table of the lane-per-block decoder: the block's bits, and the rungs its units are entered with
            if (a.g.ulen_sz == 4) ((uint32_t *)a.idx.ulen)[gblk] = myblen | ((prp & 0xffffu) << 16);
            if (seg_start) {
#pragma unroll
                for (int c = 0; c < B; c++) {
                    ((uint8_t *)a.idx.prev)[(uint64_t)seg * B + c] = (uint8_t)(pvp >> (8 * c));
                    ((uint8_t *)a.idx.cf)[(uint64_t)seg * B + c] = (uint8_t)pcfv[c];
                    a.idx.rung[(uint64_t)seg * B + c] = (uint8_t)((prp >> (4 * c)) & 15u);
                }
                a.idx.bitpos[seg] = ((uint64_t)chunk << 32) | pos;
            }
        }
    }
    __syncthreads();                    // every unit to emit densely is counted (and the analysis queue is free again)
    const uint32_t ne = wsum[41];
    for (uint32_t r0 = 0; r0 < ne; r0 += PXB_CAP) {
#pragma unroll
        for (int c = 0; c < B; c++)
            if (qe[c] != ~0u && qe[c] - r0 < PXB_CAP) {
                const uint32_t j = qe[c] - r0;
#pragma unroll
                for (int q = 0; q < 4; q++) hq[q * PXB_CAP + j] = f.gp[c][q];
                hq[4 * PXB_CAP + j] = ((rp_packed >> (4 * c)) & 15u) | (((prp >> (4 * c)) & 15u) << 4) | (kind[c] << 8) | ((uint32_t)same[c] << 10) |
                                      (((res0[c] >> 8) & 15u) << 12) | ((res0[c] & 0xffu) << 16);
                hr[j] = upos[c];
            }
        __syncthreads();
        if (r0 + tid < ne) {
            const uint32_t G[4] = {hq[tid], hq[PXB_CAP + tid], hq[2 * PXB_CAP + tid], hq[3 * PXB_CAP + tid]};
            const uint32_t meta = hq[4 * PXB_CAP + tid];
            pxb_emit(outbuf, hr[tid], G, meta & 15u, (meta >> 4) & 15u, (meta >> 8) & 3u, (meta >> 10) & 1u, (meta >> 12) & 15u, (meta >> 16) & 0xffu, etab_off);
        }
        __syncthreads();
    }
    // the chunk's bits go to its slot; enc_concat_kernel moves them into place once every chunk is counted
    const uint32_t nd4 = (total + 127) >> 7;
    uint4 *slot = (uint4 *)(a.scratch + (uint64_t)chunk * a.slot_dw);
    for (uint32_t d = tid; d < nd4; d += 256) slot[d] = ((const uint4 *)outbuf)[d];
    if (tid == 0) a.chunk_bits[chunk] = total;
    if (FIRST) {
        if (tid < B) a.cw_used[(uint64_t)chunk * B + tid] = (uint8_t)used_entry[tid];
        if (tid == 0) a.recode_need[chunk] = 0;
    }
}

// The driver kernels: as k_enc_best.hip's (a sample of the chunks decides between one coding pass plus repairs and two passes)
__device__ __forceinline__ bool pxb_two_pass(const EncArgs &a) {
    const uint32_t sampled = best_sample_count(a.nchunks);
    return a.recode_n[1] > ((sampled * a.g.bands) >> 4);
}
template <int B, bool RGB, uint64_t ORDER>
__global__ void __launch_bounds__(256) enc_px_best_sample_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t step = (a.nchunks + gridDim.x - 1) / gridDim.x, chunk = blockIdx.x * step;
    if (chunk >= a.nchunks) return;
    px_best_chunk<B, RGB, ORDER, true>(a, a0, smem, chunk, true);
    // what counts is a writer that moves the factor AWAY from the state the one-pass coding assumes
    if (threadIdx.x < B && a.cw_has[(uint64_t)chunk * B + threadIdx.x] &&
        a.cw_val[(uint64_t)chunk * B + threadIdx.x] != a0.st.cf[threadIdx.x]) atomicAdd(&a.recode_n[1], 1u);
}
template <int B, bool RGB, uint64_t ORDER, bool FIRST>
__global__ void __launch_bounds__(256) enc_px_best_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    enc_scan_counter_reset(a);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const bool two_pass = pxb_two_pass(a);
    if (FIRST) { px_best_chunk<B, RGB, ORDER, true>(a, a0, smem, blockIdx.x, two_pass); return; }
    const uint32_t n = two_pass ? a.nchunks : a.recode_n[0];
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        px_best_chunk<B, RGB, ORDER, false>(a, a0, smem, two_pass ? i : a.recode_list[i], false);
        __syncthreads();
    }
}

void launch_best_scan(const EncArgs &a, hipStream_t st);        // k_enc_best.hip: the scan across chunks and the index fix-up

template <int B, bool RGB, uint64_t ORDER>
static void launch_enc_px_best_o(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    const dim3 grid(plan.nchunks, a.ntiles), block(256);
    {
        ProfScope ps("enc_best_units", st);
        if (a.ntiles > 1) (void)hipMemset2DAsync(a.recode_n, a.ts_ws, 0, 8, a.ntiles, st);
        else (void)hipMemsetAsync(a.recode_n, 0, 8, st);
        if (plan.nchunks >= tuning().best_sample_min) hipLaunchKernelGGL((enc_px_best_sample_kernel<B, RGB, ORDER>), dim3(best_sample_count(plan.nchunks), a.ntiles), block, plan.lds_bytes, st, a);
        hipLaunchKernelGGL((enc_px_best_kernel<B, RGB, ORDER, true>), grid, block, plan.lds_bytes, st, a);
    }
    launch_best_scan(a, st);
    ProfScope ps("enc_best_recode", st);
    hipLaunchKernelGGL((enc_px_best_kernel<B, RGB, ORDER, false>), dim3(plan.nchunks < 4096 ? plan.nchunks : 4096, a.ntiles), block, plan.lds_bytes, st, a);
}
template <int B, bool RGB>
static void launch_enc_px_best_b(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    if (a.g.order == ZCURVE) launch_enc_px_best_o<B, RGB, ZCURVE>(a, plan, st);
    else launch_enc_px_best_o<B, RGB, HILBERT>(a, plan, st);
}
void launch_enc_px_best(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    if (a.g.bands == 1) launch_enc_px_best_b<1, false>(a, plan, st);
    else if (a.g.bands == 3) { if (plan.px_rgb) launch_enc_px_best_b<3, true>(a, plan, st); else launch_enc_px_best_b<3, false>(a, plan, st); }
    else { if (plan.px_rgb) launch_enc_px_best_b<4, true>(a, plan, st); else launch_enc_px_best_b<4, false>(a, plan, st); }
}

}  // namespace qb3dev

