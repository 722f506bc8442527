// qb3_amd/csrc/k_enc_best.hip -- common-factor / index (QB3M_BEST family) encoder
#include "qb3_enc_front.h"

namespace qb3dev {

// ------------------------------------------------------------------ common-factor + index coding (BEST)
// Reference: encode_best (QB3encode.h:617-724), cfgenc (:283-361), ienc (:557-613).  A unit can be coded
// plainly, as common factor times a smaller group, or as up to eight distinct values plus indices.  The only
// state besides the rung is pcf, the previous factor of the band.  A unit overwrites pcf with cf-2 exactly when
// cf >= 2 and index coding does not beat the "factor differs" size -- a condition that does not involve pcf
// itself -- so pcf is a LAST-WRITER scan over units: pass 0 records each chunk's last writer per band,
// best_scan_kernel carries it across chunks, pass 1 codes with the right pcf.
template <typename T> __device__ __forceinline__ T mdiv_t(T v, T cf) { return (T)((T)((T)(mabs_t<T>(v) / cf) << 1) - (T)(v & 1)); }

template <typename T> __device__ __forceinline__ T gcf_t(const T (&g)[16]) {      // gcd of the non-zero magnitudes (QB3encode.h:98-126)
    // a magnitude of 1 settles it; so does an odd value next to an even one... only the first is cheap to see in every
    // lane at once, and on noisy data it spares most lanes the divergent Euclid loop
    bool one = false;
#pragma unroll
    for (uint32_t i = 0; i < 16; i++) one = one || mabs_t<T>(g[i]) == 1;
    if (one) return 1;
    T x = 0;
#pragma unroll 1
    for (uint32_t i = 0; i < 16 && x != 1; i++) {
        T y = mabs_t<T>(g[i]);
        while (y) { const T t = (T)(x % y); x = y; y = t; }
    }
    return x;
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
template <typename T> __device__ __forceinline__ void put_group(LdsWriter &w, const T (&v)[16], uint32_t rung) {
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

template <typename T>
__device__ __forceinline__ void best_analyse(const T (&g)[16], uint32_t rung, uint32_t oldrung, BestUnit<T> &u) {
    constexpr uint32_t UB = UBits<T>::v, UMASK = (1u << UB) - 1;
    u.cf = gcf_t<T>(g);
    u.szN = u.szBase = u.szCf = 0; u.trung = 0;
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
    // index coding (QB3encode.h:557-613)
    u.idx = 0xffffffffu;
    // (first count the distinct values with plain comparisons, in registers and the same in every lane: more than 8
    // means no index coding, and the search below -- small arrays indexed at run time, divergent -- is skipped)
    uint32_t distinct = 0;
    if (rung > 3 && rung < 63) {
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) {
            bool seen = false;
#pragma unroll
            for (uint32_t j = 0; j < i; j++) seen = seen || g[j] == g[i];
            distinct += !seen;
        }
    }
    if (rung > 3 && rung < 63 && distinct <= 8) {
        T val[8]; uint32_t cnt[8], n = 0;
        bool fits = true;
#pragma unroll 1
        for (uint32_t i = 0; i < 16 && fits; i++) {
            uint32_t j = 0;
            while (j < n && val[j] != g[i]) j++;
            if (j == n) { if (n == 8) fits = false; else { val[n] = g[i]; cnt[n++] = 1; } }
            else cnt[j]++;
        }
        if (fits) {
            // stable sort by descending count (QB3encode.h:546-554)
#pragma unroll 1
            for (uint32_t i = 1; i < n; i++)
                for (uint32_t j = i; j > 0 && cnt[j] > cnt[j - 1]; j--) {
                    const T tv = val[j]; val[j] = val[j - 1]; val[j - 1] = tv;
                    const uint32_t tc = cnt[j]; cnt[j] = cnt[j - 1]; cnt[j - 1] = tc;
                }
            uint32_t bits = (UB + 2) + sw_noflag_len<UB>(UMASK - oldrung) + sw_noflag_len<UB>(rung - oldrung);
#pragma unroll 1
            for (uint32_t j = 0; j < n; j++) bits += cnt[j] * (2 + (j >= 2) + (j >= 4)) + vlen_t<T>(val[j], rung);   // plain rung-2 index codes
            u.idx = bits;
        }
    }
    const uint32_t thr = 36 + 3 * UB + 2 * rung;
    const uint32_t szDiff = u.szBase + u.szCf;
    u.writer = u.cf >= 2 && !(szDiff >= thr && u.idx < szDiff);
}

// per-band "last writer" inclusive scan over the lanes of the workgroup (lanes are slot-major, band-minor, so the
// band's units are `bands` lanes apart): key = 0 for "no writer", else anything non-zero; doubling in LDS
__device__ __forceinline__ void last_writer_scan(uint32_t *key, uint64_t *val, uint32_t n, uint32_t bands, uint32_t mykey, uint64_t myval) {
    const uint32_t tid = threadIdx.x;
    if (tid < n) { key[tid] = mykey; val[tid] = myval; }
    __syncthreads();
    for (uint32_t d = bands; d < n; d <<= 1) {
        uint32_t k = 0; uint64_t v = 0;
        const bool take = tid < n && tid >= d && key[tid] == 0;
        if (take) { k = key[tid - d]; v = val[tid - d]; }
        __syncthreads();
        if (take && k) { key[tid] = k; val[tid] = v; }
        __syncthreads();
    }
}

template <typename T, int PASS>
__global__ void enc_best_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    constexpr uint32_t UB = UBits<T>::v, UMASK = (1u << UB) - 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    const uint32_t bands = a.g.bands, nblocks = (uint32_t)a.g.nblocks, slots = a.slots, nunits = slots * bands;
    T g[16];
    EncFront<T> f;
    const uint32_t outdw = a.slot_dw;
    enc_front<T>(a, a0, smem, PASS ? outdw : 0, f, g);
    const uint32_t c = f.c, gblk = f.gblk, rung = f.rung, chunk = f.chunk;
    const bool payload = f.payload;
    const T used = f.used;
    uint64_t *wval = (uint64_t *)(f.outbuf + ((outdw + 1) & ~1u));
    uint32_t *wkey = (uint32_t *)(wval + nunits);

    uint32_t oldrung = 0;
    BestUnit<T> u;
    u.writer = false; u.cf = 1; u.szN = u.szBase = u.szCf = 0; u.idx = 0xffffffffu; u.trung = 0;
    if (payload) {
        oldrung = (gblk == 0) ? a0.st.rung[c] : f.rungs[tid - bands];
        if (used > 1) best_analyse<T>(g, rung, oldrung, u);
    }
    // who wrote the band's factor last, up to and including each unit
    last_writer_scan(wkey, wval, nunits, bands, (payload && u.writer) ? 1u : 0u, (uint64_t)(T)(u.cf - 2));
    if (PASS == 0) {
        // chunk summary: the entry of the last payload slot of each band
        const uint32_t last = (slots - 1) * bands + tid;
        if (tid < bands) { a.cw_has[(uint64_t)chunk * bands + tid] = (uint8_t)(wkey[last] != 0); a.cw_val[(uint64_t)chunk * bands + tid] = wval[last]; }
        return;
    }
    // factor state entering this unit: previous unit of the band in the chunk, else the chunk's entry state
    T pcf = (T)a.centry[(uint64_t)chunk * bands + c];
    if (payload && tid >= bands && wkey[tid - bands]) pcf = (T)wval[tid - bands];
    __syncthreads();

    // ---- choose the coding and its length (QB3encode.h:679-713)
    uint32_t len = 0, kind = 0;     // kind: 0 low (used <= 1), 1 plain, 2 common factor, 3 index
    bool same = false;
    if (payload) {
        if (used <= 1) len = cs_len<UB>((rung - oldrung) & UMASK) + 1 + (used ? 16 : 0);
        else {
            const uint32_t thr = 36 + 3 * UB + 2 * rung;
            uint32_t size;
            if (u.cf >= 2) { same = (T)(u.cf - 2) == pcf; size = u.szBase + (same ? 0 : u.szCf); kind = 2; }
            else { size = u.szN; kind = 1; }
            if (size >= thr && u.idx < size) { size = u.idx; kind = 3; }
            len = size;
        }
    }
    uint32_t total;
    const uint32_t pos = block_exscan(len, f.wsum, &total);

    if (payload) {
        LdsWriter w;
        w.init(f.outbuf, pos);
        if (kind == 0) {
            w.put(cs_code<UB>((rung - oldrung) & UMASK), cs_len<UB>((rung - oldrung) & UMASK));
            w.put((uint32_t)used, 1);
            if (used) {
                uint32_t bits = 0;
#pragma unroll
                for (uint32_t i = 0; i < 16; i++) bits |= (uint32_t)(g[i] & 1) << i;
                w.put(bits, 16);
            }
        } else if (kind == 1) {
            w.put(cs_code<UB>((rung - oldrung) & UMASK), cs_len<UB>((rung - oldrung) & UMASK));
            apply_step<T>(g, rung);
            put_group<T>(w, g, rung);
        } else if (kind == 2) {     // cfgenc, QB3encode.h:283-361
            T d[16];
#pragma unroll
            for (uint32_t i = 0; i < 16; i++) d[i] = mdiv_t<T>(g[i], u.cf);
            const T cfm = (T)(u.cf - 2);
            const uint32_t trung = u.trung, cfrung = topbit_t<T>(cfm);
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
            if (trung == 0) {
                uint32_t bits = 0;
#pragma unroll
                for (uint32_t i = 0; i < 16; i++) bits |= (uint32_t)(d[i] & 1) << i;
                w.put(bits, 16);
            } else { apply_step<T>(d, trung); put_group<T>(w, d, trung); }
        } else {                    // ienc, QB3encode.h:557-613
            T val[8]; uint32_t cnt[8], n = 0;
#pragma unroll 1
            for (uint32_t i = 0; i < 16; i++) {
                uint32_t j = 0;
                while (j < n && val[j] != g[i]) j++;
                if (j == n) { val[n] = g[i]; cnt[n++] = 1; } else cnt[j]++;
            }
#pragma unroll 1
            for (uint32_t i = 1; i < n; i++)
                for (uint32_t j = i; j > 0 && cnt[j] > cnt[j - 1]; j--) {
                    const T tv = val[j]; val[j] = val[j - 1]; val[j - 1] = tv;
                    const uint32_t tc = cnt[j]; cnt[j] = cnt[j - 1]; cnt[j - 1] = tc;
                }
            put_signal<UB>(w);
            put_sw_noflag<UB>(w, UMASK - oldrung);
            put_sw_noflag<UB>(w, rung - oldrung);
#pragma unroll 1
            for (uint32_t i = 0; i < 16; i++) {
                uint32_t j = 0;
                while (val[j] != g[i]) j++;
                // plain rung-2 code of j (0..7): {0,2,1,5,3,7,11,15} with lengths {2,2,3,3,4,4,4,4}
                const uint32_t code = j < 2 ? (j << 1) : j < 4 ? (((j - 2) << 2) | 1) : (((j - 4) << 2) | 3);
                w.put(code, 2 + (j >= 2) + (j >= 4));
            }
#pragma unroll 1
            for (uint32_t j = 0; j < n; j++) put_single<T>(w, val[j], rung);
        }
        w.finish();
        // coder state on leaving the image (reference QB3encode.h:718-722)
        if (gblk == nblocks - 1) {
            a.res->prev[c] = (uint64_t)f.lastv; a.res->rung[c] = rung;
            a.res->cf[c] = (uint64_t)(kind == 2 ? (T)(u.cf - 2) : pcf);     // only a kept common-factor coding moves pcf
        }
        if (a.have_idx) {
            const uint32_t seg = gblk / a.g.seg_blocks;
            if (seg * a.g.seg_blocks == gblk) {
                ((T *)a.idx.prev)[(uint64_t)seg * bands + c] = f.pv;
                ((T *)a.idx.cf)[(uint64_t)seg * bands + c] = pcf;
                a.idx.rung[(uint64_t)seg * bands + c] = (uint8_t)oldrung;
                if (c == 0) a.idx.bitpos[seg] = ((uint64_t)chunk << 32) | pos;
            }
        }
    }
    __syncthreads();
    const uint32_t nd = (total + 31) >> 5;
    uint32_t *slot = a.scratch + (uint64_t)chunk * a.slot_dw;
    for (uint32_t d = tid; d < nd; d += nthr) slot[d] = f.outbuf[d];
    if (tid == 0) a.chunk_bits[chunk] = total;
}

// Carries the last factor writer across chunks: centry[k][c] = factor state on entering chunk k.  One workgroup;
// "last non-empty" is a max-scan over (chunk index + 1).
__global__ void best_scan_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    const uint32_t bands = a.g.bands, tid = threadIdx.x;
    for (uint32_t c = 0; c < bands; c++) {
        if (tid == 0) carry = 0;
        __syncthreads();
        for (uint32_t base = 0; base < a.nchunks; base += blockDim.x) {
            const uint32_t k = base + tid;
            uint32_t x = (k < a.nchunks && a.cw_has[(uint64_t)k * bands + c]) ? k + 1 : 0;
            // inclusive max-scan within the workgroup
            const uint32_t lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
            uint32_t m = x;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(m, d, 64); if (lane >= (uint32_t)d) m = max(m, y); }
            if (lane == 63) wsum[wave] = m;
            __syncthreads();
            uint32_t before = carry;
            for (uint32_t i = 0; i < wave && i < nw; i++) before = max(before, wsum[i]);
            const uint32_t incl = max(before, m);
            // exclusive: the last writer strictly before chunk k
            const uint32_t up = __shfl_up(m, 1, 64);
            const uint32_t excl = max(before, lane ? up : 0u);
            if (k < a.nchunks) a.centry[(uint64_t)k * bands + c] = excl ? a.cw_val[(uint64_t)(excl - 1) * bands + c] : a0.st.cf[c];
            __syncthreads();
            if (tid == blockDim.x - 1) carry = incl;
            __syncthreads();
        }
    }
}

template <typename T>
static void launch_enc_best_t(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    dim3 grid(plan.nchunks, a.ntiles), block(plan.threads);
    {
        ProfScope ps("enc_best_pass0", st);
        hipLaunchKernelGGL((enc_best_kernel<T, 0>), grid, block, plan.lds_bytes, st, a);
    }
    {
        ProfScope ps("enc_best_scan", st);
        hipLaunchKernelGGL(best_scan_kernel, dim3(1, a.ntiles), dim3(1024), 0, st, a);
    }
    ProfScope ps("enc_best_units", st);
    hipLaunchKernelGGL((enc_best_kernel<T, 1>), grid, block, plan.lds_bytes, st, a);
}
void launch_enc_best(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    switch (a.g.tsz) {
    case 1: launch_enc_best_t<uint8_t>(a, plan, st); break;
    case 2: launch_enc_best_t<uint16_t>(a, plan, st); break;
    case 4: launch_enc_best_t<uint32_t>(a, plan, st); break;
    default: launch_enc_best_t<uint64_t>(a, plan, st); break;
    }
}

}  // namespace qb3dev
