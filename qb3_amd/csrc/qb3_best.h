// qb3_amd/csrc/qb3_best.h -- per-unit analysis and emission of the common-factor / index forms (QB3M_BEST family), shared by
// the unit-per-lane encoder (k_enc_best.hip) and the 8-bit lane-per-block encoder (k_enc_px_best.hip)
#pragma once
#include "qb3_kernels.h"

namespace qb3dev {

// ------------------------------------------------------------------ common-factor + index coding (BEST)
// Reference: encode_best (QB3encode.h:617-724), cfgenc (:283-361), ienc (:557-613).  A unit can be coded
// plainly, as common factor times a smaller group, or as up to eight distinct values plus indices.  The only
// state besides the rung is pcf, the previous factor of the band.  A unit overwrites pcf with cf-2 exactly when
// cf >= 2 and index coding does not beat the "factor differs" size -- a condition that does not involve pcf
// itself -- so pcf is a LAST-WRITER scan over units.  The chunks are coded ONCE assuming the starting state on entry,
// each leaving its last writer per band; best_scan_kernel carries the writers across chunks and lists the chunks whose
// assumption was wrong and mattered; those are coded again (on data without common factors: none).
// x mod y and x / y for magnitudes.  8- and 16-bit data: through the float reciprocal (exact after one correction step:
// both operands are below 2^16); wider data: the integer operations.
// x mod y through the float reciprocal: exact after one correction step while both operands are below 2^23 (they and their
// quotient are then exact in a float's mantissa but for the reciprocal's last bit)
__device__ __forceinline__ uint32_t mod_f23(uint32_t a, uint32_t b) {      // b != 0
    const uint32_t q = (uint32_t)((float)a * __builtin_amdgcn_rcpf((float)b));
    int32_t r = (int32_t)(a - q * b);
    r += r < 0 ? (int32_t)b : 0;
    r -= r >= (int32_t)b ? (int32_t)b : 0;
    return (uint32_t)r;
}
template <typename T> __device__ __forceinline__ T mod_t(T x, T y) {       // y != 0
    if (sizeof(T) <= 2) return (T)mod_f23((uint32_t)x, (uint32_t)y);
    return (T)(x % y);
}
template <typename T> __device__ __forceinline__ T div_exact_t(T x, T y) { // y divides x
    if (sizeof(T) <= 2) return (T)(uint32_t)((float)(uint32_t)x * __builtin_amdgcn_rcpf((float)(uint32_t)y) + 0.5f);
    return (T)(x / y);
}
template <typename T> __device__ __forceinline__ T mdiv_t(T v, T cf) { return (T)((T)(div_exact_t<T>(mabs_t<T>(v), cf) << 1) - (T)(v & 1)); }

// gcd of the non-zero magnitudes (QB3encode.h:98-126).  What a WAVE pays is its slowest lane, so: a magnitude of 1
// settles a lane at once (on noisy data nearly all of them); the others start from their smallest magnitude, which
// the gcd divides, so that one or two values usually bring it down to 1; and the walk over the sixteen values stops
// as soon as every lane of the wave is settled.
template <typename T> __device__ __forceinline__ T gcf_t(const T (&g)[16], bool active) {
    T m[16], mn = (T)~(T)0;
    bool one = false;
#pragma unroll
    for (uint32_t i = 0; i < 16; i++) {
        m[i] = mabs_t<T>(g[i]);
        one = one || m[i] == 1;
        mn = (m[i] != 0 && m[i] < mn) ? m[i] : mn;
    }
    T x = (one || !active) ? (T)1 : mn;         // (an active unit has a non-zero magnitude: its rung is at least 1)
    if (sizeof(T) >= 4) {
        // 32/64-bit data: an integer modulo is tens (64-bit: a hundred and more) of instructions.  Elevation and count rasters
        // keep their deltas small: when no magnitude of the WAVE's units reaches 2^23 the float reciprocal does it (mod_f23)
        T mx = 0;
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) mx |= m[i];
        if (!__any(active && (mx >> 23) != 0)) {
            uint32_t xs = (uint32_t)x;
#pragma unroll
            for (uint32_t i = 0; i < 16; i++) {
                if (!__any(xs != 1)) break;
                if (xs != 1) {
                    uint32_t y = mod_f23((uint32_t)m[i], xs);
                    while (y) { const uint32_t t = mod_f23(xs, y); xs = y; y = t; }
                }
            }
            return (T)xs;
        }
    }
#pragma unroll
    for (uint32_t i = 0; i < 16; i++) {
        if (!__any(x != 1)) break;
        if (x != 1) {
            T y = mod_t<T>(m[i], x);
            while (y) { const T t = mod_t<T>(x, y); x = y; y = t; }
        }
    }
    return x;
}
// ---- "is there a common factor" without Euclid, for mag-sign values below 512 (rungs up to 8: smooth and noisy rasters alike).  A
// table holds for every mag-sign value g, of magnitude m <= 256: the primes up to 13 that divide m (six bits) and what is left of m when
// they are divided out -- 1 or ONE prime (17 .. 251: two of them would exceed 256).  Some prime divides every non-zero magnitude of a unit exactly
// when the AND of the masks is not empty or all the cofactors are the same prime: sixteen look-ups, no loop, no divergence -- where gcf_t
// costs a wave its slowest lane's Euclid chains.  Entry: mask | cofactor << 8 | cofactor << 16; for m = 0 (a magnitude the gcd ignores):
// 0x3f | 0xff << 8 | 0.  A compile-time constant, copied to LDS from L2 with 16-byte loads.
struct GcfSig { alignas(16) uint32_t e[512]; };
constexpr GcfSig make_gcf_sig() {
    GcfSig t{};
    for (uint32_t g = 0; g < 512; g++) {
        const uint32_t m = (g >> 1) + (g & 1);
        uint32_t e = 0x3fu | 0xff00u;
        if (m) {
            uint32_t mask = 0, c = m;
            const uint32_t pr[6] = {2, 3, 5, 7, 11, 13};
            for (int k = 0; k < 6; k++)
                if (c % pr[k] == 0) { mask |= 1u << k; while (c % pr[k] == 0) c /= pr[k]; }
            e = mask | c << 8 | c << 16;
        }
        t.e[g] = e;
    }
    return t;
}
static __device__ const GcfSig gcf_sig_tab = make_gcf_sig();
constexpr uint32_t GCF_SIG_BYTES = 2048;
__device__ __forceinline__ void fill_gcf_sig(uint32_t *sig) {       // the caller's next barrier makes it visible
    for (uint32_t i = threadIdx.x; i < GCF_SIG_BYTES / 16; i += blockDim.x) ((uint4 *)sig)[i] = ((const uint4 *)gcf_sig_tab.e)[i];
}
// true exactly when the unit's non-zero magnitudes have a common factor above 1 (mag-sign values all below 512)
template <typename T> __device__ __forceinline__ bool gcf_sig_any(const T (&g)[16], const uint32_t *sig) {
    uint32_t a = 0xffffffffu, o = 0;
#pragma unroll
    for (uint32_t i = 0; i < 16; i++) { const uint32_t e = sig[(uint32_t)g[i] & 511u]; a &= e; o |= e; }
    const uint32_t ca = (a >> 8) & 0xffu, co = (o >> 16) & 0xffu;
    return (a & 0x3fu) != 0 || (ca == co && ca > 1);
}

// bit length of one value coded on its own at rung r (reference qb3csztbl, QB3encode.h:144-150): rung 0 is one raw
// bit, rungs 1-2 plain, rungs 3-7 with the middle swap, above that plain
template <typename T> __device__ __forceinline__ uint32_t vlen_t(T v, uint32_t r) {
    if (r == 0) return 1;
    const T top = (T)((T)1 << r), half = (T)(top >> 1);
    if (r >= 3 && r < 8 && (v == top || v == (T)(top - 1))) v ^= (T)(2 * top - 1);
    return r + (v >= half) + (v >= top);
}
template <typename T> __device__ __forceinline__ void put_single(LdsWriter &w, T v, uint32_t r) {
    if (r == 0) { w.put((uint32_t)v & 1, 1); return; }
    const T top = (T)((T)1 << r);
    if (r >= 3 && r < 8 && (v == top || v == (T)(top - 1))) v ^= (T)(2 * top - 1);
    put_value<T>(w, v, r);
}
// 16 group codes at rung >= 1, values already stepped: total length / emission
template <typename T> __device__ __forceinline__ uint32_t group_len(const T (&v)[16], uint32_t rung) {
    const T top = (T)((T)1 << rung), half = (T)(top >> 1);
    uint32_t n = 16 * rung;
#pragma unroll
    for (uint32_t i = 0; i < 16; i++) {
        T x = v[i];
        if (rung < 8 && (x == top || x == (T)(top - 1))) x ^= (T)(2 * top - 1);
        n += (x >= half) + (x >= top);
    }
    return n;
}
// etab: the LDS code table of rungs 1..7 (fill_enc_tab: length << 12 | code, middle swap applied), or null
template <typename T> __device__ __forceinline__ void put_group(LdsWriter &w, const T (&v)[16], uint32_t rung, const uint16_t *etab = nullptr) {
    if (etab && rung < 8) {
        // below rung 8 every value is under 256: one table read per value, and three codes (at most 27 bits) to a write
        // into the bit buffer instead of one each
        const uint16_t *tab = etab + enc_tab_off(rung);
        uint32_t acc = 0, al = 0;
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) {
            const uint32_t e = tab[(uint32_t)v[i]];
            acc |= (e & 0xfffu) << al; al += e >> 12;
            if (i % 3 == 2 || i == 15) { w.put(acc, al); acc = 0; al = 0; }
        }
        return;
    }
    const T top = (T)((T)1 << rung);
#pragma unroll
    for (uint32_t i = 0; i < 16; i++) {
        T x = v[i];
        if (rung < 8 && (x == top || x == (T)(top - 1))) x ^= (T)(2 * top - 1);
        put_value<T>(w, x, rung);
    }
}
template <uint32_t UB> __device__ __forceinline__ uint32_t sw_noflag_len(uint32_t delta) {       // switch without flag, signal for "no change"
    const uint32_t l = cs_len<UB>(delta & ((1u << UB) - 1));
    return (l == 1 ? UB + 2 : l) - 1;
}
template <uint32_t UB> __device__ __forceinline__ void put_sw_noflag(LdsWriter &w, uint32_t delta) {
    delta &= (1u << UB) - 1;
    constexpr uint32_t r = UB - 1, sig = ((((1u << UB) - 2 - (1u << r)) << 2) | 3);    // code of 2^UB-2 at rung UB-1 (long form)
    if (delta == 0) w.put(sig, UB + 1);
    else w.put(cs_code<UB>(delta) >> 1, cs_len<UB>(delta) - 1);
}
template <uint32_t UB> __device__ __forceinline__ void put_signal(LdsWriter &w) {
    constexpr uint32_t r = UB - 1, sig = ((((1u << UB) - 2 - (1u << r)) << 2) | 3);
    w.put((sig << 1) | 1, UB + 2);
}

// Everything pass 0 and pass 1 agree on for one unit (used > 1)
template <typename T> struct BestUnit {
    T cf;                   // common factor (>= 1)
    uint32_t szN;           // plain coding size
    uint32_t szBase;        // cf coding: signal + switch + same/diff flag + divided group
    uint32_t szCf;          // cf coding: extra bits when the factor has to be written
    uint32_t idx;           // index coding size, 0xffffffff if more than 8 distinct values
    uint32_t trung;
    bool writer;            // overwrites pcf with cf-2
};

// number of distinct values among the sixteen (exact up to 9: all the caller asks is "at most 8?")
template <typename T> __device__ __forceinline__ uint32_t distinct_t(const T (&g)[16], uint32_t rung, bool need) {
    // All the caller asks is "at most 8 distinct values, and then how many" (index coding holds no more).
    uint32_t distinct = 99;
    bool open = need;                   // lanes that still need the exact count
    {
        // the values' low five bits in a 32-bit bitmap: exact up to rung 4 (values below 32), else a LOWER bound (values
        // that differ there differ) -- more than 8 settles the lane: on noisy data nearly every lane, with 32-bit operations
        // whatever the value width
        if (__any(need)) {
            uint32_t bm = 0;
#pragma unroll
            for (uint32_t i = 0; i < 16; i++) bm |= 1u << ((uint32_t)g[i] & 31u);
            const uint32_t lb = (uint32_t)__popc(bm);
            if (rung <= 4 || lb > 8) { distinct = lb; open = false; }
        }
        if (__any(open && rung <= 5)) {     // values below 64: a 64-bit bitmap
            uint64_t bm = 0;
#pragma unroll
            for (uint32_t i = 0; i < 16; i++) bm |= 1ull << ((uint32_t)g[i] & 63u);
            if (open && rung <= 5) { distinct = (uint32_t)__popcll(bm); open = false; }
        }
    }
    if (__any(open)) {                  // plain comparisons, in registers and the same in every lane
        uint32_t d = 0;
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) {
            bool seen = false;
#pragma unroll
            for (uint32_t j = 0; j < i; j++) seen = seen || g[j] == g[i];
            d += !seen;
        }
        if (open) distinct = d;
    }
    return distinct;
}

// cf: the unit's common factor (gcf_t).  writer_only: all the caller wants is u.writer (pass 0).
template <typename T>
__device__ __forceinline__ void best_analyse(const T (&g)[16], uint32_t rung, uint32_t oldrung, T cf, bool writer_only, BestUnit<T> &u) {
    constexpr uint32_t UB = UBits<T>::v, UMASK = (1u << UB) - 1;
    u.cf = cf;
    u.szN = u.szBase = u.szCf = 0; u.trung = 0;
    u.idx = 0xffffffffu;
    u.writer = false;
    if (writer_only && cf < 2) return;          // only a unit with a common factor can overwrite the band's factor
    if (u.cf >= 2) {
        T d[16], usedd = 0;
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) { d[i] = mdiv_t<T>(g[i], u.cf); usedd |= d[i]; }
        const T cfm = (T)(u.cf - 2);
        const uint32_t trung = topbit_t<T>(usedd), cfrung = topbit_t<T>(cfm);
        u.trung = trung;
        uint32_t grp = 16;
        if (trung) { apply_step<T>(d, trung); grp = group_len<T>(d, trung); }
        u.szBase = (UB + 2) + sw_noflag_len<UB>(trung - oldrung) + 1 + grp;
        if (trung >= cfrung && (trung < cfrung + UB || cfrung == 0)) u.szCf = 1 + vlen_t<T>(cfm, trung);
        else u.szCf = cs_len<UB>((cfrung - trung) & UMASK) + vlen_t<T>((T)(cfm ^ (T)((T)1 << cfrung)), cfrung - 1);
    } else {
        T v[16];
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) v[i] = g[i];
        apply_step<T>(v, rung);
        u.szN = cs_len<UB>((rung - oldrung) & UMASK) + group_len<T>(v, rung);
    }
    // index coding (QB3encode.h:557-613): only tried for rungs 4..62 and when the size so far reaches the threshold
    // (:702); more than 8 distinct values means no index coding, and the search below -- small arrays indexed at run
    // time, divergent -- is skipped
    const uint32_t thr = 36 + 3 * UB + 2 * rung;
    const uint32_t szDiff = u.szBase + u.szCf, szSame = u.cf >= 2 ? u.szBase : u.szN;
    const bool try_idx = rung > 3 && rung < 63 && (u.cf >= 2 ? szDiff : szSame) >= thr;
    const uint32_t distinct = distinct_t<T>(g, rung, try_idx);
    // Can the index form win at all?  With n distinct values the index codes take at least 32 + max(n-2, 0) + max(n-4, 0)
    // bits (every count beyond the first value's is 1) and the values at least n * rung: where that bound already
    // reaches the size to beat, the exact size is not needed.  On noisy data this spares nearly every wave the sort.
    const uint32_t idx_head = (UB + 2) + sw_noflag_len<UB>(UMASK - oldrung) + sw_noflag_len<UB>(rung - oldrung);
    const uint32_t idx_floor = idx_head + 32 + (distinct > 2 ? distinct - 2 : 0) + (distinct > 4 ? distinct - 4 : 0) + distinct * rung;
    const bool idx_may_win = try_idx && distinct <= 8 && idx_floor < (u.cf >= 2 ? szDiff : szSame);
    if (__any(idx_may_win)) {
        // The size of the index form (QB3encode.h:557-613) without building it.  With the distinct values ranked by
        // descending count, a value of rank j costs cnt_j index codes of 2 + (j >= 2) + (j >= 4) bits (the plain rung-2
        // code) plus its own code at `rung`: the sum of the index codes is 64 - S2 - S4 with S2 / S4 the sum of the two /
        // four largest counts, and the values' own codes do not depend on the order.  Sort the sixteen values (a fixed
        // network: no run-time indexed arrays, the same instructions in every lane), read the counts off the runs.
        T v[16];
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) v[i] = g[i];
#pragma unroll
        for (uint32_t k = 2; k <= 16; k <<= 1)
#pragma unroll
            for (uint32_t j = k >> 1; j > 0; j >>= 1)
#pragma unroll
                for (uint32_t i = 0; i < 16; i++) {
                    const uint32_t l = i ^ j;
                    if (l > i) {
                        const bool up = (i & k) == 0;
                        const T lo = v[i] < v[l] ? v[i] : v[l], hi = v[i] < v[l] ? v[l] : v[i];
                        v[i] = up ? lo : hi; v[l] = up ? hi : lo;
                    }
                }
        uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, run = 0, vbits = 0;       // the four largest counts, the run in progress
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) {
            run++;
            const bool last = i == 15 || v[i + 1] != v[i];
            if (last) {
                vbits += vlen_t<T>(v[i], rung);
                uint32_t r = run;                   // insert into c0 >= c1 >= c2 >= c3
                uint32_t t = c0 < r ? c0 : r; c0 = c0 < r ? r : c0; r = t;
                t = c1 < r ? c1 : r; c1 = c1 < r ? r : c1; r = t;
                t = c2 < r ? c2 : r; c2 = c2 < r ? r : c2; r = t;
                c3 = c3 < r ? r : c3;
                run = 0;
            }
        }
        const uint32_t bits = idx_head + 64 - 2 * (c0 + c1) - (c2 + c3) + vbits;
        if (idx_may_win) u.idx = bits;
    }
    u.writer = u.cf >= 2 && !(szDiff >= thr && u.idx < szDiff);
}

}  // namespace qb3dev
