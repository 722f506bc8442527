// qb3_amd/csrc/k_enc_best.hip -- common-factor / index (QB3M_BEST family) encoder
#include "qb3_enc_front.h"
#include "qb3_best.h"

namespace qb3dev {

// Per-band "last writer" lookup over the lanes of the workgroup.  Lanes are slot-major, band-minor (unit t is band
// t % bands), so the lanes of one band are `bands` apart.  Every wave publishes the ballot of its writers (one word
// pair) and every lane its value; after ONE barrier a lane finds the last writer of its band at or before lane
// `upto` by masking the ballots with its band's lane pattern and counting leading zeros.
struct WriterBoard { uint64_t *masks; uint64_t *vals; };       // masks: one per wave; vals: one per lane
__device__ __forceinline__ void writers_publish(const WriterBoard &wb, bool writer, uint64_t val) {
    const uint64_t m = __ballot(writer);
    if ((threadIdx.x & 63) == 0) wb.masks[threadIdx.x >> 6] = m;
    wb.vals[threadIdx.x] = val;
    __syncthreads();
}
// last writer of band `c` among lanes 0..upto (upto < 0: none); false if there is none
__device__ __forceinline__ bool writers_find(const WriterBoard &wb, uint32_t bands, uint32_t c, int32_t upto, uint64_t *val) {
    if (upto < 0) return false;
    uint64_t period = 0;                                        // lanes 0, bands, 2*bands, ...
    for (uint32_t k = 0; k < 64; k += bands) period |= 1ull << k;
    for (int32_t w = upto >> 6; w >= 0; w--) {
        const uint32_t r = (c + bands - (uint32_t)(w * 64) % bands) % bands;    // lanes of wave w that belong to band c: l % bands == r
        uint64_t m = wb.masks[w] & (period << r);
        if (w == (upto >> 6)) m &= ~0ull >> (63 - (upto & 63));                 // lanes <= upto
        if (m) { *val = wb.vals[w * 64 + 63 - __clzll((long long)m)]; return true; }
    }
    return false;
}

// One chunk.  FIRST: the pass over all chunks, each assuming that the band's factor on entering the chunk is the state
// the image starts with; it also leaves the chunk's summary (its last factor writer per band, and whether anything in
// it depended on the entering factor).  !FIRST: a chunk coded again with its true entering factor (best_scan_kernel
// lists the chunks whose assumption was wrong AND mattered: on data without common factors, none).
// FRONT: 0 the unit-per-lane front end (an LDS tile, any band count); 1 / 2: 32/64-bit rasters of one band, lane per block, the
// block in registers (pxw_front; Hilbert / Z curve)
template <typename T, bool FIRST, int FRONT = 0>
__device__ __forceinline__ void best_chunk(const EncArgs &a, const EncArgs &a0, uint8_t *smem, uint32_t chunk, bool summary_only) {
    constexpr uint32_t UB = UBits<T>::v, UMASK = (1u << UB) - 1;
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    const uint32_t bands = a.g.bands, nblocks = (uint32_t)a.g.nblocks, slots = a.slots;
    T g[16];
    EncFront<T> f;
    const uint32_t outdw = a.slot_dw;
    // the first 2 KB of LDS: factor signatures of the mag-sign values below 512 (gcf_sig_any; visible behind the front end's barrier)
    uint32_t *gsig = (uint32_t *)smem;
    fill_gcf_sig(gsig);
    smem += GCF_SIG_BYTES;
    if constexpr (FRONT == 1) pxw_front<T, HILBERT>(a, a0, smem, outdw, f, g, chunk);
    else if constexpr (FRONT == 2) pxw_front<T, ZCURVE>(a, a0, smem, outdw, f, g, chunk);
    else enc_front<T>(a, a0, smem, outdw, f, g, chunk);
    const uint32_t c = f.c, gblk = f.gblk, rung = f.rung;
    const bool payload = f.payload;
    const T used = f.used;
    WriterBoard wb;
    wb.vals = (uint64_t *)f.board;                              // nthr values, then one mask per wave, then a "used the entry state" word per band
    wb.masks = wb.vals + nthr;
    uint32_t *used_entry = (uint32_t *)(wb.masks + 16);
    uint32_t *blk_tab = used_entry + MAXBANDS + 2;             // a word per block slot: the index's block table (ulen_sz == 4)
    if (FIRST && tid < bands) used_entry[tid] = 0;
    if (a.have_idx && a.g.ulen_sz == 4 && tid < slots) blk_tab[tid] = 0;

    uint32_t oldrung = 0;
    BestUnit<T> u;
    u.writer = false; u.cf = 1; u.szN = u.szBase = u.szCf = 0; u.idx = 0xffffffffu; u.trung = 0;
    const bool analyse = payload && used > 1;
    if (payload) oldrung = (gblk == 0) ? a0.st.rung[c] : (FRONT ? f.prung : f.rungs[tid - bands]);
    {
        T cf = 1;
        // units whose magnitudes all stay below 256 -- nearly every unit of 8- and 16-bit imagery and of smooth wide rasters --
        // ask the signature table whether there is a factor at all (rung <= 8); Euclid runs only in a wave that holds such a unit (or a larger magnitude)
        const bool small = rung <= 8;               // (mag-sign values below 512: magnitudes of at most 256)
        bool maybe = analyse;
        if (__any(analyse && small)) { const bool y = gcf_sig_any<T>(g, gsig); if (small) maybe = analyse && y; }
        if (__any(maybe)) cf = gcf_t<T>(g, maybe);
        if (analyse) best_analyse<T>(g, rung, oldrung, cf, summary_only, u);
    }
    // who wrote the band's factor last
    writers_publish(wb, payload && u.writer, (uint64_t)(T)(u.cf - 2));
    if (FIRST && tid < bands) {         // chunk summary: the last writer of each band among all the chunk's units
        uint64_t v = 0;
        const bool has = writers_find(wb, bands, tid, (int32_t)(slots * bands) - 1, &v);
        a.cw_has[(uint64_t)chunk * bands + tid] = (uint8_t)has; a.cw_val[(uint64_t)chunk * bands + tid] = v;
    }
    if (FIRST && summary_only) return;   // (workgroup uniform)
    // factor state entering this unit: the last writer before it in the chunk, else the chunk's entry state
    T pcf = FIRST ? (T)a0.st.cf[c] : (T)a.centry[(uint64_t)chunk * bands + c];
    {
        uint64_t v = 0;
        const bool mine = payload && writers_find(wb, bands, c, (int32_t)tid - 1, &v);
        if (mine) pcf = (T)v;
        // what depends on the entering factor: the coding of a unit with a common factor (the image's final state is
        // best_scan_kernel's business)
        if (FIRST && payload && !mine && u.cf >= 2) atomicOr(&used_entry[c], 1u);
        // ... and index entries that hold it are marked, best_idx_fix_kernel gives them the true value
        if (FIRST && payload && a.have_idx && gblk % a.g.seg_blocks == 0) a.seg_from_entry[(uint64_t)(gblk / a.g.seg_blocks) * bands + c] = (uint8_t)!mine;
    }

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
    if (payload && a.have_idx && a.g.ulen_sz == 4) atomicAdd(&blk_tab[f.s], len | (oldrung << (16 + 4 * c)));     // (block_exscan has the barrier)
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
            put_group<T>(w, g, rung, f.etab);
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
            } else { apply_step<T>(d, trung); put_group<T>(w, d, trung, f.etab); }
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
            // (the band's final factor: best_scan_kernel, from the chunk summaries)
        }
        if (a.have_idx) {
            if (a.g.ulen_sz == 4 && c == 0) ((uint32_t *)a.idx.ulen)[gblk] = blk_tab[f.s];
            if (a.g.ulen_sz == ULEN_UNIT) ((uint32_t *)a.idx.ulen)[(uint64_t)gblk * bands + c] = len | (oldrung << 16);      // (the lane-per-unit decoder: the unit's bits | the rung it is entered with)
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
    if (FIRST) {
        if (tid < bands) a.cw_used[(uint64_t)chunk * bands + tid] = (uint8_t)used_entry[tid];
        if (tid == 0) a.recode_need[chunk] = 0;
    }
}

// The image is coded in one pass when it has no common factors to speak of, in two when it has: a SAMPLE of the chunks
// (best_sample_count of them, evenly spread) is analysed first; if factor writers that change the state are common in it (recode_n[1]
// counts them), the first pass only collects the chunk summaries (cheap) while the last pass codes every chunk with its
// true entering factor;
// otherwise the first pass codes every chunk assuming the starting state and the last pass repairs the listed few.
template <typename T, int FRONT = 0>
__global__ void __launch_bounds__(256) enc_best_sample_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t step = (a.nchunks + gridDim.x - 1) / gridDim.x, chunk = blockIdx.x * step;
    if (chunk >= a.nchunks) return;
    best_chunk<T, true, FRONT>(a, a0, smem, chunk, true);
    // what counts is a writer that moves the factor AWAY from the state the one-pass coding assumes
    if (threadIdx.x < a.g.bands && a.cw_has[(uint64_t)chunk * a.g.bands + threadIdx.x] &&
        a.cw_val[(uint64_t)chunk * a.g.bands + threadIdx.x] != a0.st.cf[threadIdx.x]) atomicAdd(&a.recode_n[1], 1u);
}
// two passes when more than one in sixteen of the sampled (chunk, band) pairs had such a writer
__device__ __forceinline__ bool best_two_pass(const EncArgs &a) {
    const uint32_t sampled = best_sample_count(a.nchunks);
    return a.recode_n[1] > ((sampled * a.g.bands) >> 4);
}

template <typename T, bool FIRST, int FRONT = 0>
__global__ void __launch_bounds__(256) enc_best_kernel(const EncArgs a0) {      // (256 = the plan's largest block: without the bound the compiler budgets for 1024 threads and the recode loop spills)
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    enc_scan_counter_reset(a);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const bool two_pass = best_two_pass(a);
    if (FIRST) { best_chunk<T, true, FRONT>(a, a0, smem, blockIdx.x, two_pass); return; }
    const uint32_t n = two_pass ? a.nchunks : a.recode_n[0];
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        best_chunk<T, false, FRONT>(a, a0, smem, two_pass ? i : a.recode_list[i], false);
        __syncthreads();
    }
}

// Carries the last factor writer across chunks: centry[k][c] = factor state on entering chunk k.  "Last non-empty" is
// a max-scan over (chunk index + 1).  The chunks are cut into BS_PARTS parts; PHASE 0: the last writer inside every
// part (a max-reduction) -> part_last; PHASE 1: a part starts from the last writer of the parts before it and scans
// its own chunks, a thread 8 chunks in a row.  grid = (parts, bands, tiles).
constexpr uint32_t BS_PARTS = 32;
template <int PHASE>
__global__ void __launch_bounds__(256) best_scan_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.z);
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t carry;
    constexpr uint32_t PER = 8;
    const uint32_t bands = a.g.bands, tid = threadIdx.x, c = blockIdx.y, part = blockIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint32_t per_part = ((a.nchunks + BS_PARTS - 1) / BS_PARTS + 256 * PER - 1) / (256 * PER) * (256 * PER);
    const uint32_t k_begin = min(a.nchunks, part * per_part), k_end = min(a.nchunks, k_begin + per_part);
    uint32_t *part_last = a.centry_parts + (uint64_t)c * BS_PARTS;       // per band: BS_PARTS words
    if (PHASE == 0) {
        if (part == 0 && c == 0 && tid == 0) a.recode_n[0] = 0;
        uint32_t m = 0;
        for (uint32_t k = k_begin + tid; k < k_end; k += 256) m = a.cw_has[(uint64_t)k * bands + c] ? k + 1 : m;      // k ascends: the last one stays
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
        if (lane == 0) wsum[wave] = m;
        __syncthreads();
        if (tid == 0) part_last[part] = max(max(wsum[0], wsum[1]), max(wsum[2], wsum[3]));
        return;
    }
    if (tid == 0) {
        uint32_t m = 0;
        for (uint32_t p = 0; p < part; p++) m = max(m, part_last[p]);
        carry = m;
    }
    __syncthreads();
    for (uint32_t base = k_begin; base < k_end; base += 256 * PER) {
        const uint32_t k0 = base + tid * PER;
        uint32_t has[PER], last = 0;                    // last writer (chunk index + 1) among the thread's chunks
#pragma unroll
        for (uint32_t i = 0; i < PER; i++) {
            has[i] = (k0 + i < k_end && a.cw_has[(uint64_t)(k0 + i) * bands + c]) ? k0 + i + 1 : 0;
            last = has[i] ? has[i] : last;
        }
        uint32_t m = last;                              // inclusive max-scan over the threads
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(m, d, 64); if (lane >= (uint32_t)d) m = max(m, y); }
        if (lane == 63) wsum[wave] = m;
        __syncthreads();
        uint32_t before = carry;
        for (uint32_t i = 0; i < wave; i++) before = max(before, wsum[i]);
        const uint32_t up = __shfl_up(m, 1, 64);
        uint32_t run = max(before, lane ? up : 0u);     // last writer strictly before the thread's first chunk
#pragma unroll
        for (uint32_t i = 0; i < PER; i++) {
            const uint32_t k = k0 + i;
            if (k < k_end) {
                const uint64_t entry = run ? a.cw_val[(uint64_t)(run - 1) * bands + c] : a0.st.cf[c];
                a.centry[(uint64_t)k * bands + c] = entry;
                // the first pass assumed the image's starting state: where that was wrong and the chunk looked at it, code it again
                if (!best_two_pass(a) && entry != a0.st.cf[c] && a.cw_used[(uint64_t)k * bands + c] && atomicExch(&a.recode_need[k], 1u) == 0)
                    a.recode_list[atomicAdd(a.recode_n, 1u)] = k;
            }
            run = has[i] ? has[i] : run;
            if (k == a.nchunks - 1) a.res->cf[c] = run ? a.cw_val[(uint64_t)(run - 1) * bands + c] : a0.st.cf[c];   // the band's factor on leaving the image
        }
        __syncthreads();
        if (tid == 255) carry = max(before, m);
        __syncthreads();
    }
}

// One-pass coding: index entries that recorded the assumed entering factor get the true one.
template <typename T>
__global__ void best_idx_fix_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    if (best_two_pass(a)) return;
    const uint32_t bands = a.g.bands, nbp = a.slots - 1;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.g.nseg * bands || !a.seg_from_entry[i]) return;
    const uint64_t seg = i / bands;
    const uint32_t c = (uint32_t)(i - seg * bands), chunk = (uint32_t)(seg * a.g.seg_blocks / nbp);
    ((T *)a.idx.cf)[i] = (T)a.centry[(uint64_t)chunk * bands + c];
}

// the scan across chunks and the fix-up of the index entries, shared with the lane-per-block encoder (k_enc_px_best.hip)
template <typename T>
static void launch_best_scan_t(const EncArgs &a, hipStream_t st) {
    {
        ProfScope ps("enc_best_scan", st);
        hipLaunchKernelGGL(best_scan_kernel<0>, dim3(BS_PARTS, a.g.bands, a.ntiles), dim3(256), 0, st, a);
        hipLaunchKernelGGL(best_scan_kernel<1>, dim3(BS_PARTS, a.g.bands, a.ntiles), dim3(256), 0, st, a);
    }
    ProfScope ps("enc_best_recode", st);
    if (a.have_idx) hipLaunchKernelGGL((best_idx_fix_kernel<T>), dim3((uint32_t)((a.g.nseg * a.g.bands + 255) / 256), a.ntiles), dim3(256), 0, st, a);
}
void launch_best_scan(const EncArgs &a, hipStream_t st) {
    switch (a.g.tsz) {
    case 1: launch_best_scan_t<uint8_t>(a, st); break;
    case 2: launch_best_scan_t<uint16_t>(a, st); break;
    case 4: launch_best_scan_t<uint32_t>(a, st); break;
    default: launch_best_scan_t<uint64_t>(a, st); break;
    }
}

template <typename T, int FRONT>
static void launch_enc_best_f(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    dim3 grid(plan.nchunks, a.ntiles), block(plan.threads);
    {
        ProfScope ps("enc_best_units", st);
        if (a.ntiles > 1) (void)hipMemset2DAsync(a.recode_n, a.ts_ws, 0, 8, a.ntiles, st);
        else (void)hipMemsetAsync(a.recode_n, 0, 8, st);
        if (plan.nchunks >= tuning().best_sample_min) hipLaunchKernelGGL((enc_best_sample_kernel<T, FRONT>), dim3(best_sample_count(plan.nchunks), a.ntiles), block, plan.lds_bytes, st, a);
        hipLaunchKernelGGL((enc_best_kernel<T, true, FRONT>), grid, block, plan.lds_bytes, st, a);
    }
    launch_best_scan_t<T>(a, st);
    ProfScope ps("enc_best_recode", st);
    hipLaunchKernelGGL((enc_best_kernel<T, false, FRONT>), dim3(plan.nchunks < 4096 ? plan.nchunks : 4096, a.ntiles), block, plan.lds_bytes, st, a);
}
template <typename T>
static void launch_enc_best_t(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    if constexpr (sizeof(T) >= 2) {
        // one band of 16/32/64-bit data (value-aligned pointers): the lane-per-block front end
        if (plan.pxw_best && ((uintptr_t)a.img & (sizeof(T) - 1)) == 0 && !(a.ts_img & (sizeof(T) - 1))) {
            if (a.g.order == ZCURVE) launch_enc_best_f<T, 2>(a, plan, st); else launch_enc_best_f<T, 1>(a, plan, st);
            return;
        }
    }
    launch_enc_best_f<T, 0>(a, plan, st);
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
