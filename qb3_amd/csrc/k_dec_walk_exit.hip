// qb3_amd/csrc/k_dec_walk_exit.hip -- plain streams by EXITS: per super-window of the stream the state every entering state
// leaves it with (made by the whole chip), one hop a super-window, a wave a super-window parses its units.  One band of any
// width (FTL / BASE / common factor), 8-bit RGB (FTL / BASE / common factor).
#include "qb3_walk.h"

namespace qb3dev {

// ---- Single-band 32/64-bit plain streams: EXITS instead of a chain through the table.
// The chain above reads the whole table through one CU (32 bytes a stream bit at 15.8 GB/s).  With ONE band the state of the
// walk is (position, rung) and nothing else, so a function "state entering a stretch of the stream -> state leaving it" can be
// tabulated and functions of consecutive stretches composed -- the serial part then takes one step per STRETCH, not per unit.
// walk_exitW_kernel: a workgroup per super-window of K windows of W positions.  Per window it builds the table T[position][rung
// in] -> (position the unit ends at, rung behind its switch) in LDS, the way walk_tableW_kernel does (one target rung at a time:
// extras of 2, 4, 8 codes by doubling), and moves every state of X -- all (position < MAXU, rung) a walk can enter the
// super-window with -- through T until it leaves the window; after K windows X holds, per entering state, the state the walk
// leaves the super-window with and the units it took: 4 bytes x MAXU x 16 per 32768 stream bits, about a byte a bit.
// walk_exit_chain_kernel: one lane hops from super-window to super-window (one dependent load each) and notes where each is
// entered.  walk_exit_units_kernel: a lane per super-window parses its units from there: unit lengths, segment entries.
// A unit that leaves the band of rungs or carries the signal code stops the walk: status bit 0, and the caller falls back.
template <uint32_t UB> struct WalkValue { typedef typename std::conditional<UB == 3, uint8_t, typename std::conditional<UB == 4, uint16_t, typename std::conditional<UB == 5, uint32_t, uint64_t>::type>::type>::type type; };
template <uint32_t UB> struct exitW {
    static constexpr uint32_t NRUNG = 1u << UB, NR = 16, NRB = NRUNG < NR ? NRUNG : NR, MAXC = NRUNG + 1, MAXU = UB + 2 + 16 * MAXC;
    static constexpr uint32_t W = UB == 6 ? 1024 : 2048, K = 65536 / W, SW = W * K, THREADS = 1024;      // (a super-window's cost is its first window's: long ones)
    // A walk leaves a window at the first unit that starts behind it AND is entered with a rung of the band: units entered
    // out of the band (the one behind a unit whose switch jumped out: the first unit of a block row of a wide raster) are
    // walked on the spot from the code lengths, so a window can be entered up to PE bits in.  (8- and 16-bit data: the band
    // is all the rungs there are.)
    static constexpr uint32_t PE = MAXU + (UB >= 5 ? 512 : 0), NX = PE * NR;                // states a window can be entered with
    static constexpr uint32_t NPT = (W + UB + 2 + 15 * MAXC + 2 + 31) & ~31u;               // positions the table of a window looks at
    static constexpr uint32_t NPS = W + PE, NP1 = (NPS + MAXU + 2 + 31) & ~31u;             // positions with a switch entry; with a code length
    static constexpr uint32_t X_DEP = 1u << 30, X_SLOW = 1u << 31, X_CNT = 0x7fffu;          // (common-factor streams) a unit took the factor in force when the super-window was entered; a unit brought its own
    static constexpr uint32_t X_STOP = 0x7fffu;                                             // X: (position - W) * 16 + rung (15 bits: the entering state of the next window) | units << 15; stop: the low 15 bits all set
    static constexpr uint32_t BMW = (NX + 31) / 32, DCAP = UB == 6 ? NX : (NX < 4096 ? NX : 4096);   // words of the bitmap of first-window exits; distinct exits carried (64-bit data, 1024-bit windows: thousands; else a few hundred)
    static constexpr uint32_t T0 = 0, X0 = T0 + W * NR * 2, PF0 = X0 + ((BMW * 4 + 15) & ~15u), XD0 = PF0 + ((BMW * 2 + 15) & ~15u), S0 = XD0 + DCAP * 4,
                              E1 = S0 + ((NPS * 2 + 15) & ~15u), EA = E1 + NP1, EB = EA + NPT, WORDS = EB + NPT, LDS_BYTES = WORDS + (NP1 / 32 + 3) * 4;
    static_assert(W + MAXU < 4095 && PE * NR + NR <= 0x7fff && UB >= 3 && UB <= 6 && LDS_BYTES <= 160 * 1024, "entry layouts of the exit walk");
};

template <uint32_t UB, bool CF>
__global__ void __launch_bounds__(1024) walk_exitW_kernel(const DecArgs a0, uint32_t *xg, uint32_t s_begin, uint32_t s_count, const WalkState16 *states) {
    typedef exitW<UB> E;
    constexpr uint32_t W = E::W, NR = E::NR, NPT = E::NPT, NP1 = E::NP1, NPS = E::NPS, MAXC = E::MAXC, NRUNG = E::NRUNG, NX = E::NX, NT = E::THREADS;
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    const WalkState16 &S = states[blockIdx.y];
    if (S.bad) return;
    const uint32_t R0 = S.pad;
    const uint64_t base = S.P + (uint64_t)(s_begin + blockIdx.x) * E::SW;                   // the super-window's first bit (the walk enters the first one at its bit 0)
    if (base >= a.in_bits) return;                                                          // (uniform) no walk comes here
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *T = (uint16_t *)(smem + E::T0), *sw = (uint16_t *)(smem + E::S0);
    uint32_t *bm = (uint32_t *)(smem + E::X0), *Xd = (uint32_t *)(smem + E::XD0), *words = (uint32_t *)(smem + E::WORDS);
    uint16_t *pf = (uint16_t *)(smem + E::PF0);
    uint32_t *out = xg + ((uint64_t)blockIdx.y * s_count + blockIdx.x) * NX;          // X: the super-window's exits, entering state by entering state
    __shared__ uint32_t s_D;
    uint32_t D = 0;
    uint8_t *t1 = smem + E::E1, *eA = smem + E::EA, *eB = smem + E::EB;
    const uint32_t tid = threadIdx.x;
    const uint64_t endw = (a.in_bit0 + a.in_bits + 31) >> 5;
    // a first-window exit (or a list entry) a, then b: flags add up -- a factor brought anywhere, a factor taken before one was
    // brought, or after: X_SLOW then
    auto compose = [](uint32_t a, uint32_t b) -> uint32_t {
        const uint32_t n = ((a >> 15) & E::X_CNT) + ((b >> 15) & E::X_CNT);               // (more units than the field holds -- two bits a unit: flat data -- stop the walk)
        return (n > E::X_CNT ? E::X_STOP : (b & 0x7fffu)) | ((n & E::X_CNT) << 15) | ((a | b) & (E::X_DEP | E::X_SLOW));
    };
#pragma unroll 1
    for (uint32_t k = 0; k < E::K; k++) {
        const uint64_t q0 = a.in_bit0 + base + (uint64_t)k * W, w0 = q0 >> 5;
        const uint32_t sh = (uint32_t)q0 & 31;
        for (uint32_t i = tid; i < NP1 / 32 + 3; i += NT) words[i] = w0 + i < endw ? a.in32[w0 + i] : 0u;
        __syncthreads();
        auto bits = [&](uint32_t i) { const uint32_t b = sh + i, j = b >> 5; return __builtin_amdgcn_alignbit(words[j + 1], words[j], b & 31); };
        for (uint32_t i = tid; i < NP1; i += NT) { const uint32_t x = bits(i); t1[i] = (uint8_t)((x & 1) + ((x & 3) == 3)); }   // a code's extra bits
        for (uint32_t o = tid; o < NPS; o += NT) {                                          // the switch in front of a unit that starts at o
            uint32_t delta = 0; bool sig = false;
            const uint32_t cs = walk_switch<UB>(bits(o), delta, sig);
            sw[o] = (uint16_t)(cs | (delta << 4) | ((sig ? 1u : 0u) << 10) | ((bits(o + cs) & 1u) << 11));
        }
        // The table is made for the super-window's FIRST window, where thousands of states walk; behind it a few hundred
        // distinct states are usually left, and walking those from the code lengths (sixteen dependent byte reads a unit)
        // costs a quarter of what tabulating sixteen rungs of the window does.
        const bool tabled = k == 0 || D > 1024;                                             // (uniform.  Many distinct states: the table pays in every window)
        if (tabled) for (uint32_t i = tid; i < W * NR / 2; i += NT) ((uint32_t *)T)[i] = 0xffffffffu;  // (an entry no target rung fills: the unit leaves the band, or is the signal)
        __syncthreads();
#pragma unroll 1
        for (uint32_t rb = 0; tabled && rb < E::NRB; rb++) {                                // the rung the switch leads to
            const uint32_t r = R0 + rb;
            if (r) {                                                                        // extras of two, four, eight codes at rung r
                for (uint32_t i = tid; i < NPT - MAXC; i += NT) { const uint32_t e = t1[i]; eA[i] = (uint8_t)(e + t1[i + r + e]); }
                __syncthreads();
                for (uint32_t i = tid; i < NPT - 3 * MAXC; i += NT) { const uint32_t e = eA[i]; eB[i] = (uint8_t)(e + eA[i + 2 * r + e]); }
                __syncthreads();
                for (uint32_t i = tid; i < NPT - 7 * MAXC; i += NT) { const uint32_t e = eB[i]; eA[i] = (uint8_t)(e + eB[i + 4 * r + e]); }
                __syncthreads();
            }
            for (uint32_t o = tid; o < W; o += NT) {
                const uint32_t s = sw[o], cs = s & 15u, delta = (s >> 4) & 63u;
                const uint32_t bin = ((r - delta) & (NRUNG - 1)) - R0;                      // the rung the unit is entered with, in the band
                if (bin >= NR || ((s >> 10) & 1u)) continue;
                uint32_t u = cs + (((s >> 11) & 1u) ? 17u : 1u);                            // rung 0: one flag, then 16 raw bits
                if (r) { const uint32_t n8 = 8 * r + eA[o + cs]; u = cs + n8 + 8 * r + eA[o + cs + n8]; }
                T[o * NR + bin] = (uint16_t)((o + u) | (rb << 12));
            }
            __syncthreads();
        }
        // Every entering state of the super-window through its first window (exit and unit count to global memory, X); walks merge
        // -- behind the first window the thousands of states stand at a few hundred distinct (position, rung) -- so the distinct
        // exits are ranked through a bitmap and only those (Xd, LDS) are carried through the other windows; at the end every
        // state composes its first-window exit with what became of it.
        auto walk = [&](uint32_t key, auto with_table) -> uint32_t {                        // (with_table: a compile-time flag -- the loop without the table look-up is the tighter one)
            constexpr bool TB = decltype(with_table)::value;
            uint32_t pos = key / NR, r = key % NR, cnt = 0;                                 // (a state's low 15 bits: position * 16 + rung: the key itself)
            bool stop = false;
            typedef typename WalkValue<UB>::type TT;
            TT cfv = (TT)S.cf; uint32_t xfl = 0;                                            // (common-factor streams) the factor in force: the one behind the first segment until a unit brings its own
            while (true) {
                if (r < NR) {
                    if (pos >= W) break;                                                    // behind the window, in the band: the next window's
                    if (TB) {
                        const uint32_t e = T[pos * NR + r];
                        if (e != 0xffffu) { pos = e & 0xfffu; r = e >> 12; cnt++; continue; }
                    }
                }
                // a unit the table does not hold (it leaves the band, is entered from outside it, or there is no table): by the code lengths
                if (pos >= NPS) { stop = true; break; }
                const uint32_t s = sw[pos], cs = s & 15u;
                if ((s >> 10) & 1u) {                                                       // the signal code
                    if (!CF) { stop = true; break; }                                        // ... in a stream that should have none
                    // a common-factor or index unit: parsed outright (its values decide the rung it leaves).  The factor in force
                    // is not part of the state: a unit that takes it is walked with the factor the stream had behind its first
                    // segment and says so (X_DEP: right as long as no unit in between brought another, which the hop checks); a
                    // unit that brings its own marks the walk X_SLOW: the hop parses that super-window outright.
                    ReaderT<LdsWords> rd;
                    rd.init((LdsWords)words, sh + pos, 32ull * (NP1 / 32 + 3));
                    uint32_t rg = (R0 + r) & (NRUNG - 1), fl = 0;
                    TT pc = cfv, g[16];
                    const bool ok = parse_unit<TT, CM_BEST>(rd, rg, pc, g, &fl);
                    if (!ok) { stop = true; break; }
                    if ((fl & 1u) && !(xfl & E::X_SLOW)) xfl |= E::X_DEP;
                    if (fl & 2u) { xfl |= E::X_SLOW; cfv = pc; }
                    pos = (uint32_t)rd.position() - sh; r = (rg - R0) & (NRUNG - 1); cnt++;
                    continue;
                }
                const uint32_t rabs = (R0 + r + ((s >> 4) & 63u)) & (NRUNG - 1);
                uint32_t q = pos + cs;
                if (rabs) { for (uint32_t i = 0; i < 16; i++) q += rabs + t1[q]; }
                else q += ((s >> 11) & 1u) ? 17u : 1u;
                pos = q; r = (rabs - R0) & (NRUNG - 1); cnt++;
            }
            if (pos - W >= E::PE) stop = true;                                              // (only behind a unit entered out of the band)
            return (stop ? (E::X_STOP | (cnt << 15)) : ((pos - W) * NR + r) | (cnt << 15)) | xfl;
        };
        if (k == 0) {
            for (uint32_t i = tid; i < E::BMW; i += NT) bm[i] = 0;
            __syncthreads();
            for (uint32_t key = tid; key < NX; key += NT) {
                const uint32_t x = walk(key, std::true_type());
                out[key] = x;
                if ((x & E::X_STOP) != E::X_STOP) atomicOr(&bm[(x & 0x7fffu) >> 5], 1u << (x & 31u));
            }
            __syncthreads();
            if (tid < 64) {                                                                 // rank of every distinct exit: one wave scans the bitmap words' bit counts
                uint32_t run = 0;
                for (uint32_t w0 = 0; w0 < E::BMW; w0 += 64) {
                    const uint32_t w = w0 + tid, c = w < E::BMW ? __popc(bm[w]) : 0u;
                    uint32_t x = c;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if ((int)tid >= d) x += y; }
                    if (w < E::BMW) pf[w] = (uint16_t)(run + x - c);
                    run += __shfl(x, 63, 64);
                }
                if (tid == 0) s_D = run;
            }
            __syncthreads();
            D = s_D;
            if (D > E::DCAP) {                                                              // (uniform) more distinct exits than are carried: the hop parses this super-window outright
                for (uint32_t key = tid; key < NX; key += NT) out[key] = E::X_STOP;
                return;
            }
            for (uint32_t w = tid; w < E::BMW; w += NT) {                                   // the distinct exits, in rank order
                uint32_t m = bm[w], j = pf[w];
                while (m) { const uint32_t b = __ffs(m) - 1; Xd[j++] = w * 32 + b; m &= m - 1; }
            }
            __syncthreads();
        } else {
            for (uint32_t j = tid; j < D; j += NT) {                                        // the distinct walks through this window
                const uint32_t x = Xd[j];
                if ((x & E::X_STOP) == E::X_STOP) continue;
                Xd[j] = compose(x, tabled ? walk(x & 0x7fffu, std::true_type()) : walk(x & 0x7fffu, std::false_type()));
            }
            __syncthreads();
        }
    }
    for (uint32_t key = tid; key < NX; key += NT) {                                         // every state: its first-window exit, then what became of that
        const uint32_t e = out[key];
        if ((e & E::X_STOP) == E::X_STOP) continue;
        const uint32_t k1 = e & 0x7fffu, w = k1 >> 5;
        out[key] = compose(e, Xd[pf[w] + __popc(bm[w] & ((1u << (k1 & 31u)) - 1u))]);
    }
}

// entries: per tile nsuper + 2 pairs of {position lo, hi, unit, rung in the band} {factor in force lo, hi}: where and how the walk
// enters super-window s; the last pair is {the super-window the walk stands in front of, 1 when every unit has been found}.  The
// stream is taken s_count super-windows at a time (the memory for their exits is reused): a call takes up where the one before
// stopped.  A super-window whose exit cannot be taken from the table -- a unit in it brought a common factor of its own, or took
// the one in force when that is no longer the one the table was made with, or the walk stopped -- is parsed outright by this
// lane (about 260 units): the stream still decodes, at the one-lane parser's pace for that stretch.
template <uint32_t UB, int MODE>
__global__ void __launch_bounds__(64) walk_exit_chain_kernel(const DecArgs a0, const uint32_t *xg, uint32_t nsuper, uint32_t s_begin, uint32_t s_count, WalkState16 *states, uint4 *entries) {
    typedef exitW<UB> E;
    typedef typename WalkValue<UB>::type T;
    const DecArgs a = dec_for_tile(a0, blockIdx.x);
    if (threadIdx.x) return;
    WalkState16 *S = states + blockIdx.x;
    if (S->bad) return;
    const uint64_t nunits = a.g.nblocks, P0 = S->P, spec = S->cf;
    const uint32_t R0 = S->pad;
    uint4 *en = entries + (uint64_t)blockIdx.x * 2 * (nsuper + 2), *hd = en + 2 * (nsuper + 1);
    uint64_t P = P0, U = S->unit, cf = spec;
    uint32_t r = (uint32_t)S->rungs & 15u, s = 0;
    bool bad = false, done = false;
    if (s_begin) {
        const uint4 h = *hd;
        if (h.y) return;                                                                    // all units found in an earlier call
        const uint4 e = en[2 * s_begin], f = en[2 * s_begin + 1];
        P = (uint64_t)e.x | (uint64_t)e.y << 32; U = e.z; r = e.w; s = s_begin; cf = (uint64_t)f.x | (uint64_t)f.y << 32;
        bad = h.x != s_begin;
    }
    const uint32_t s_end = s_begin + s_count < nsuper ? s_begin + s_count : nsuper;
    const uint32_t *x0 = xg + (uint64_t)blockIdx.x * s_count * E::NX;
    while (!bad) {
        en[2 * s] = make_uint4((uint32_t)P, (uint32_t)(P >> 32), (uint32_t)U, r);
        en[2 * s + 1] = make_uint4((uint32_t)cf, (uint32_t)(cf >> 32), 0u, 0u);
        if (U >= nunits) { done = true; break; }
        if (s >= s_end) { bad = s >= nsuper; break; }                                       // the next call's; or units left and no stream (a damaged one)
        if (P >= a.in_bits) { bad = true; break; }
        const uint64_t base = P0 + (uint64_t)s * E::SW;
        const uint32_t x = x0[(uint64_t)(s - s_begin) * E::NX + (uint32_t)(P - base) * E::NR + r];
        const bool stopped = (x & E::X_STOP) == E::X_STOP;
        const bool slow = stopped || (MODE == CM_BEST && ((x & E::X_SLOW) || ((x & E::X_DEP) && cf != spec)));
        s++;
        if (!slow) {
            U += (x >> 15) & E::X_CNT;
            r = x & 15u;
            P = base + E::SW + ((x & 0x7fffu) >> 4);
            continue;
        }
        if (stopped && U + ((x >> 15) & E::X_CNT) >= nunits) {                              // the stream's units end before the stop
            done = true;
            en[2 * s] = make_uint4(0u, 0u, (uint32_t)nunits, 0u); en[2 * s + 1] = make_uint4(0u, 0u, 0u, 0u);
            break;
        }
        // this super-window by the units themselves: up to the first unit that starts behind it and is entered with a rung of the band
        atomicOr(a.status, 64u);                                                            // (not an error: says that the walk was handed to this lane)
        Reader rd;
        rd.init(a.in32, a.in_bit0 + P, a.in_bit0 + a.in_bits);
        uint32_t rung = R0 + r;
        T pc = (T)cf, g[16];
        bool ok = true;
        const uint64_t end = base + E::SW;
        while (ok && U < nunits) {
            const uint64_t pos = rd.position() - a.in_bit0;
            if (pos >= a.in_bits || (pos >= end && ((rung - R0) & (E::NRUNG - 1)) < E::NR)) break;
            ok = parse_unit<T, MODE>(rd, rung, pc, g);
            U++;
        }
        P = rd.position() - a.in_bit0; r = (rung - R0) & (E::NRUNG - 1); cf = (uint64_t)pc;
        if (!ok || (U < nunits && (P < end || P - end >= E::PE || r >= E::NR))) { bad = true; break; }
    }
    *hd = make_uint4(s, done ? 1u : 0u, 0u, 0u);                                            // done: super-windows 0 .. s - 1 have units to parse, entries 0 .. s stand
    if (bad) { S->bad = 1u; atomicOr(a.status, 1u); }
}

// idx.prev = 0 behind the first segment: the sums of the segments' values are added up there by the lanes that parse them
template <typename T>
__global__ void __launch_bounds__(256) walk_exit_zero_kernel(const DecArgs a0) {
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    const uint64_t seg = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (seg && seg < a.g.nseg) for (uint32_t c = 0; c < a.g.bands; c++) ((T *)a.idx.prev)[seg * a.g.bands + c] = 0;
}
// A WAVE per super-window: its stretch of the stream staged in LDS by all lanes (a lane parsing straight from global memory waits
// a round trip per word: 2.5 ms for 4096^2 int32 against 0.3 staged), then lane 0 parses the units: unit lengths, segment entries,
// for common-factor streams the segments' sums.  sw_bits / pe_bits: the super-window's size and how far in it can be entered.
template <typename T, int MODE>
__global__ void __launch_bounds__(64) walk_exit_units_kernel(const DecArgs a0, const WalkState16 *states, const uint4 *entries, uint32_t nsuper, uint32_t sw_bits, uint32_t pe_bits) {
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    const WalkState16 &S = states[blockIdx.y];
    const uint32_t s = blockIdx.x, lane = threadIdx.x;
    const uint4 *en = entries + (uint64_t)blockIdx.y * 2 * (nsuper + 2);
    const uint4 hd = en[2 * (nsuper + 1)];
    if (S.bad || !hd.y || s >= hd.x) return;                                                // (uniform)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *stage = (uint32_t *)smem;
    const uint4 e = en[2 * s], f = en[2 * s + 1];
    const uint32_t B = a.g.bands;
    const uint64_t nblocks = a.g.nblocks, NB = a.g.seg_blocks;
    uint64_t U = e.z, Uend = en[2 * s + 2].z;                                               // (in blocks)
    if (Uend > nblocks) Uend = nblocks;
    const uint64_t P = (uint64_t)e.x | (uint64_t)e.y << 32;
    const uint64_t q0 = a.in_bit0 + P, w0 = q0 >> 5, endw = (a.in_bit0 + a.in_bits + 31) >> 5;
    const uint32_t nw = (sw_bits + pe_bits) / 32 + 4;                                       // (the units of this super-window end where the next is entered)
    for (uint32_t i = lane; i < nw; i += 64) stage[i] = w0 + i < endw ? a.in32[w0 + i] : 0u;
    __syncthreads();
    if (lane) return;
    uint32_t rung[4];
    for (uint32_t c = 0; c < 4; c++) rung[c] = (B == 1 ? S.pad : 0u) + ((e.w >> (4 * c)) & 15u);
    ReaderT<LdsWords> rd;
    rd.init((LdsWords)stage, (uint32_t)q0 & 31, 32ull * nw);
    const uint64_t rel = w0 * 32 - a.in_bit0;                                               // stream position of the stage's first bit
    uint32_t lpos = (uint32_t)q0 & 31;                                                      // (FTL / BASE) bit position in the stage
    const uint32_t lds0 = 8u * (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)stage;    // ... whose first bit is LDS bit lds0
    // (the factors in force where the super-window is entered: one band -- the whole value; several -- a byte a band, 8-bit data)
    T g[16], pcf[4], tot[4];
    for (uint32_t c = 0; c < 4; c++) { tot[c] = 0; pcf[c] = B == 1 ? (T)((uint64_t)f.x | (uint64_t)f.y << 32) : (T)(f.x >> (8 * c)); }
    bool ok = true;
    typedef typename std::conditional<sizeof(T) == 8, unsigned long long, unsigned int>::type AT;
    for (; U < Uend; U++) {
        if (U % NB == 0) {
            const uint64_t seg = U / NB;
            a.idx.bitpos[seg] = rel + (MODE != CM_BEST ? (uint64_t)lpos : rd.position());
            for (uint32_t c = 0; c < B; c++) {
                a.idx.rung[seg * B + c] = (uint8_t)rung[c];
                if (MODE == CM_BEST) ((T *)a.idx.cf)[seg * B + c] = pcf[c];
            }
        }
        if (MODE != CM_BEST) {      // FTL / BASE: lengths only, by position in the staged words (walk_unit: three dependent reads an 8-bit unit; a full parse costs ten times that)
            for (uint32_t c = 0; c < B; c++) {
                bool bad = false;
                const uint32_t len = walk_unit<UBits<T>::v>(lds0 + lpos, rung[c], bad);
                ok = ok && !bad;
                if (sizeof(T) == 1) ((uint8_t *)a.idx.ulen)[U * B + c] = (uint8_t)len; else ((uint16_t *)a.idx.ulen)[U * B + c] = (uint16_t)len;
                lpos += len;
            }
            continue;
        }
        const uint64_t b0 = rd.position();
        uint32_t bt = 0;                                                                    // (8-bit common-factor streams: the block's entry of the lane-per-block decoder's table)
        for (uint32_t c = 0; c < B; c++) {
            const uint64_t u0 = rd.position();
            if (c < 4) bt |= (rung[c] & (sizeof(T) >= 4 ? 63u : 15u)) << (16 + 4 * c);
            const uint32_t rung_in = rung[c];
            ok = parse_unit<T, MODE>(rd, rung[c], pcf[c], g) && ok;
            if (MODE == CM_BEST && a.g.ulen_sz == ULEN_UNIT) ((uint32_t *)a.idx.ulen)[U * B + c] = (uint32_t)((rd.position() - u0) & 0xffffu) | rung_in << 16;      // (two bands: the lane-per-unit decoder's dword)
            if (MODE != CM_BEST) {
                if (sizeof(T) == 1) ((uint8_t *)a.idx.ulen)[U * B + c] = (uint8_t)(rd.position() - u0); else ((uint16_t *)a.idx.ulen)[U * B + c] = (uint16_t)(rd.position() - u0);
            } else {                                                                        // the segment's sum of values: the scan makes entering values of them
#pragma unroll
                for (uint32_t i = 0; i < 16; i++) tot[c] = (T)(tot[c] + smag_t<T>(g[i]));
                if ((U + 1) % NB == 0 || U + 1 == Uend) {
                    const uint64_t slot = (U / NB) * B + c;
                    if (sizeof(T) >= 4) atomicAdd((AT *)a.idx.prev + slot, (AT)tot[c]);
                    else {                                                                  // narrow values: the slot's lane of its dword, by compare and swap (the neighbours may be added to meanwhile)
                        constexpr uint32_t BITS = sizeof(T) < 4 ? 8 * sizeof(T) : 16, PER = sizeof(T) < 4 ? 4 / sizeof(T) : 1, MASK = (1u << BITS) - 1;       // (instantiated, not run, for wide values)
                        uint32_t *wp = (uint32_t *)a.idx.prev + slot / PER;
                        const uint32_t sh = (uint32_t)(slot % PER) * BITS;
                        uint32_t old = *(volatile uint32_t *)wp, assumed;
                        do {
                            assumed = old;
                            const uint32_t nv = (assumed & ~(MASK << sh)) | ((((assumed >> sh) + (uint32_t)tot[c]) & MASK) << sh);
                            old = atomicCAS(wp, assumed, nv);
                        } while (old != assumed);
                    }
                    tot[c] = 0;
                }
            }
        }
        if (MODE == CM_BEST && a.g.ulen_sz == 4) ((uint32_t *)a.idx.ulen)[U] = bt | (uint32_t)((rd.position() - b0) & 0xffffu);
    }
    if (!ok) atomicOr(a.status, 1u);
}
// ---- The same for 8-bit rasters of THREE bands (RGB).  The walk's state at a block boundary is (position, a rung per band):
// 447 positions x 512 rung combinations = 228 864 states a super-window can be entered with -- too many to carry through every
// window, but rungs aside the walks merge within one window (positions do, rung offsets never: about ten positions survive per
// combination).  So the first window is walked by every state (its exit and block count go to global memory, G), the distinct
// exits are ranked through a bitmap (D, a few thousand), only those are carried through the other windows of the super-window
// (Xd, LDS), and at the end every state composes its first-window exit with what became of it.  One hop per super-window of
// 65 536 bits as before; 915 KB of exits per super-window, so the stream is taken in rounds of what the table memory holds.
template <uint32_t B, bool CF = false, uint32_t UB_ = 3> struct exitB {       // UB_ = 4: 16-bit rasters of two bands (556 positions x 256 rung pairs), FTL / BASE
    static constexpr uint32_t UB = UB_, NRUNG = 1u << UB, NR = NRUNG, MAXC = NRUNG + 1, MAXU = UB + 2 + 16 * MAXC;        // 149 / 278
    static constexpr uint32_t W = 2048, K = 64, SW = W * K, THREADS = 1024;             // (a super-window's cost is its first window's, where every state walks: long ones -- twice this: 4 % more, and the lanes that parse the units become the long pole)
    static constexpr uint32_t PE = B * MAXU, NC = 1u << (UB * B), NKEY = PE * NC;                             // entering positions, rung combinations, states
    static constexpr uint32_t TP = W + (B - 1) * MAXU;                                                        // positions with a table row: the later units of a block that starts in the window
    static constexpr uint32_t NPT = (TP + UB + 2 + 15 * MAXC + 2 + 31) & ~31u, NP1 = (TP + MAXU + 2 + 63) & ~31u;
    static constexpr uint32_t KEYB = 18, KEYM = (1u << KEYB) - 1, X_STOP = KEYM, DCAP = 8192;                 // X: state | blocks << 18; stop: the state field all set
    // blocks of a super-window: 14 bits, or 13 beside the bit that says (common-factor streams) "a unit took the factor in force when
    // the super-window was entered": more blocks than that -- eight or sixteen bits a block: flat data -- stop the walk, the hop parses it
    static constexpr uint32_t X_DEP = CF ? 1u << 31 : 0u, CNTM = CF ? 0x1fffu : 0x3fffu;
    static constexpr uint32_t BMW = (NKEY + 31) / 32;                                                         // words of the bitmap of first-window exits
    static constexpr uint32_t NSIG = 128;                                                                     // (common-factor streams) positions of a window whose unit carries the signal code, at most
    // the code lengths of 2, 4, 8 codes for FG rungs in one pass each (FG arrays of each kind) instead of a pass -- and its barrier -- per rung:
    // all seven rungs at once for 8-bit FTL / BASE streams (30 KB more LDS), four at a time where the memory is shorter
    static constexpr uint32_t FG = (!CF && UB == 3) ? NRUNG - 1 : 4;
    static constexpr uint32_t T0 = 0, BM0 = T0 + ((TP * NR * 2 + 15) & ~15u), PF0 = BM0 + BMW * 4, XD0 = PF0 + ((BMW * 2 + 15) & ~15u), S0 = XD0 + DCAP * 4,
                              E1 = S0 + ((TP * 2 + 15) & ~15u), EA = E1 + NP1, EB = EA + NPT * FG, WORDS = EB + NPT * FG, SG0 = (WORDS + (NP1 / 32 + 3) * 4 + 15) & ~15u,
                              SL0 = SG0 + (CF ? NSIG * B * NR * 4 : 0), SP0 = SL0 + (CF ? (TP + 15) & ~15u : 0), LDS_BYTES = SP0 + (CF ? NSIG * 2 + 16 : 0);
    static_assert((B == 2 || (B == 3 && UB == 3)) && (UB == 3 || (UB == 4 && !CF)) && NKEY <= KEYM && TP + MAXU < 4095 && LDS_BYTES <= 160 * 1024, "entry layouts of the exit walk of two- and three-band rasters");
};

template <uint32_t B, bool CF, uint32_t UBW = 3>
__global__ void __launch_bounds__(1024) walk_exitB_kernel(const DecArgs a0, uint32_t *xg, uint32_t s_begin, uint32_t s_count, const WalkState16 *states, uint32_t dcap) {
    typedef exitB<B, CF, UBW> E;
    constexpr uint32_t W = E::W, NR = E::NR, NPT = E::NPT, NP1 = E::NP1, TP = E::TP, MAXC = E::MAXC, NRUNG = E::NRUNG, NKEY = E::NKEY, NT = E::THREADS, UB = E::UB;
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    const WalkState16 &S = states[blockIdx.y];
    if (S.bad) return;
    const uint64_t base = S.P + (uint64_t)(s_begin + blockIdx.x) * E::SW;
    if (base >= a.in_bits) return;                                                          // (uniform) no walk comes here
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *T = (uint16_t *)(smem + E::T0), *sw = (uint16_t *)(smem + E::S0), *pf = (uint16_t *)(smem + E::PF0);
    uint32_t *bm = (uint32_t *)(smem + E::BM0), *Xd = (uint32_t *)(smem + E::XD0), *words = (uint32_t *)(smem + E::WORDS);
    uint8_t *t1 = smem + E::E1, *eA = smem + E::EA, *eB = smem + E::EB;
    uint32_t *G = xg + ((uint64_t)blockIdx.y * s_count + blockIdx.x) * NKEY;               // the super-window's exits, entering state by entering state
    __shared__ uint32_t s_D;
    const uint32_t tid = threadIdx.x;
    const uint64_t endw = (a.in_bit0 + a.in_bits + 31) >> 5;
    uint32_t *side = (uint32_t *)(smem + E::SG0);                                           // (common-factor streams: see below)
    uint8_t *sig_slot = smem + E::SL0;
    uint16_t *sigpos = (uint16_t *)(smem + E::SP0);
    __shared__ uint32_t s_nsig;
    // a state through the window: whole blocks until one starts behind it.  Returns the state behind | blocks << 18 (| X_DEP), or the
    // stop.  Common-factor streams: the units with the signal code are tabulated apart, per window, by band and entering rung (a
    // dense pass: every lane parses one -- parsed inside the walks, one lane of a wave at a time, they made the kernel 25 times
    // slower); one that takes its band's factor in force is parsed with the factor the stream had behind its first segment and
    // the exit says so (X_DEP); one that brings its own ends the walk -- the hop parses such a super-window itself.
    auto walk = [&](uint32_t key) -> uint32_t {
        uint32_t pos = key >> (UB * B), r[B], cnt = 0, dep = 0;
#pragma unroll
        for (uint32_t c = 0; c < B; c++) r[c] = (key >> (UB * c)) & (NRUNG - 1);
        while (pos < W) {
#pragma unroll
            for (uint32_t c = 0; c < B; c++) {
                const uint32_t e = T[pos * NR + r[c]];
                if (e != 0xffffu) { pos = e & 0xfffu; r[c] = e >> 12; continue; }
                if (!CF) return E::X_STOP;                                                  // the signal code: not a stream for this walk
                const uint32_t j = sig_slot[pos];                                           // ... tabulated apart: by band and entering rung
                if (j == 0xffu) return E::X_STOP;
                const uint32_t v = side[(j * B + c) * NR + r[c]];
                if (v & 0x10000u) return E::X_STOP;
                if (v & 0x8000u) dep = E::X_DEP;
                pos = v & 0xfffu; r[c] = (v >> 12) & 7u;
                if (c + 1 < B && pos >= TP) return E::X_STOP;                               // (the block's next unit would start behind the table)
            }
            cnt++;
        }
        if (pos - W >= E::PE) return E::X_STOP;
        uint32_t k2 = (pos - W) << (UB * B);
#pragma unroll
        for (uint32_t c = 0; c < B; c++) k2 |= r[c] << (UB * c);
        return k2 | (cnt << E::KEYB) | dep;
    };
    // an exit a, then b
    auto compose = [](uint32_t a, uint32_t b) -> uint32_t {
        if ((b & E::KEYM) == E::X_STOP) return E::X_STOP;
        const uint32_t n = ((a >> E::KEYB) & E::CNTM) + ((b >> E::KEYB) & E::CNTM);
        if (n > E::CNTM) return E::X_STOP;
        return (b & E::KEYM) | (n << E::KEYB) | ((a | b) & E::X_DEP);
    };
    uint32_t D = 0;
#pragma unroll 1
    for (uint32_t k = 0; k < E::K; k++) {
        const uint64_t q0 = a.in_bit0 + base + (uint64_t)k * W, w0 = q0 >> 5;
        const uint32_t sh = (uint32_t)q0 & 31;
        for (uint32_t i = tid; i < NP1 / 32 + 3; i += NT) words[i] = w0 + i < endw ? a.in32[w0 + i] : 0u;
        __syncthreads();
        auto bits = [&](uint32_t i) { const uint32_t b = sh + i, j = b >> 5; return __builtin_amdgcn_alignbit(words[j + 1], words[j], b & 31); };
        for (uint32_t i = tid; i < NP1; i += NT) { const uint32_t x = bits(i); t1[i] = (uint8_t)((x & 1) + ((x & 3) == 3)); }
        for (uint32_t o = tid; o < TP; o += NT) {
            uint32_t delta = 0; bool sig = false;
            const uint32_t cs = walk_switch<UB>(bits(o), delta, sig);
            sw[o] = (uint16_t)(cs | (delta << 4) | ((sig ? 1u : 0u) << 10) | ((bits(o + cs) & 1u) << 11));
        }
        __syncthreads();
        {
            // FG rungs at a time: level by level (2, 4, 8 codes), an array per rung of the group; then the group's row entries (the first
            // group also rung 0 and, for positions with the signal code, the whole row of "no entry")
            constexpr uint32_t FG = E::FG;
            uint8_t (*A)[NPT] = (uint8_t (*)[NPT])eA, (*Bq)[NPT] = (uint8_t (*)[NPT])eB;
#pragma unroll 1
            for (uint32_t rb = 1; rb < NRUNG; rb += FG) {
                for (uint32_t i = tid; i < NPT - MAXC; i += NT) {
                    const uint32_t e = t1[i];
#pragma unroll
                    for (uint32_t q = 0; q < FG; q++) if (rb + q < NRUNG) A[q][i] = (uint8_t)(e + t1[i + (rb + q) + e]);
                }
                __syncthreads();
                for (uint32_t i = tid; i < NPT - 3 * MAXC; i += NT) {
#pragma unroll
                    for (uint32_t q = 0; q < FG; q++) if (rb + q < NRUNG) { const uint32_t e = A[q][i]; Bq[q][i] = (uint8_t)(e + A[q][i + 2 * (rb + q) + e]); }
                }
                __syncthreads();
                for (uint32_t i = tid; i < NPT - 7 * MAXC; i += NT) {
#pragma unroll
                    for (uint32_t q = 0; q < FG; q++) if (rb + q < NRUNG) { const uint32_t e = Bq[q][i]; A[q][i] = (uint8_t)(e + Bq[q][i + 4 * (rb + q) + e]); }
                }
                __syncthreads();
                for (uint32_t o = tid; o < TP; o += NT) {
                    const uint32_t s = sw[o], cs = s & 15u, delta = (s >> 4) & 63u;
                    if ((s >> 10) & 1u) {       // the signal code: no entry at any rung
                        if (rb == 1) for (uint32_t r = 0; r < NR; r++) T[o * NR + r] = 0xffffu;
                        continue;
                    }
                    if (rb == 1) T[o * NR + ((0u - delta) & (NRUNG - 1))] = (uint16_t)(o + cs + (((s >> 11) & 1u) ? 17u : 1u));      // rung 0
#pragma unroll
                    for (uint32_t q = 0; q < FG; q++) {
                        const uint32_t r = rb + q;
                        if (r < NRUNG) {
                            const uint32_t n8 = 8 * r + A[q][o + cs], u = cs + n8 + 8 * r + A[q][o + cs + n8];
                            T[o * NR + ((r - delta) & (NRUNG - 1))] = (uint16_t)((o + u) | (r << 12));
                        }
                    }
                }
                __syncthreads();
            }
        }
        if constexpr (CF) {       // the units with the signal code: their places, then every (place, band, entering rung) parsed by a lane of its own
            if (tid == 0) s_nsig = 0;
            for (uint32_t o = tid; o < TP; o += NT) sig_slot[o] = 0xffu;
            __syncthreads();
            for (uint32_t o = tid; o < TP; o += NT)
                if ((sw[o] >> 10) & 1u) { const uint32_t j = atomicAdd(&s_nsig, 1u); if (j < E::NSIG) { sigpos[j] = (uint16_t)o; sig_slot[o] = (uint8_t)j; } }
            __syncthreads();
            const uint32_t nsig = s_nsig < E::NSIG ? s_nsig : E::NSIG;
            for (uint32_t i = tid; i < nsig * NR; i += NT) {                               // (a lane per place and entering rung: the bands differ only where the unit takes a factor in force, and only if theirs differ)
                const uint32_t j = i / NR, rin = i % NR, o = sigpos[j];
                uint32_t first = 0;
#pragma unroll
                for (uint32_t c = 0; c < B; c++) {
                    const uint8_t spec_c = (uint8_t)(S.cf >> (8 * c));
                    uint32_t v = first;
                    if (c == 0 || ((first & 0x8000u) && spec_c != (uint8_t)S.cf)) {
                        ReaderT<LdsWords> rd;
                        rd.init((LdsWords)words, sh + o, 32ull * (NP1 / 32 + 3));
                        uint32_t rg = rin, fl = 0;
                        uint8_t pc = spec_c, g[16];
                        const bool ok = parse_unit<uint8_t, CM_BEST>(rd, rg, pc, g, &fl);
                        const uint32_t end = (uint32_t)rd.position() - sh;
                        v = (!ok || (fl & 2u) || end >= 4096u) ? 0x10000u : end | ((rg & 7u) << 12) | ((fl & 1u) ? 0x8000u : 0u);
                        if (c == 0) first = v;
                    }
                    side[(j * B + c) * NR + rin] = v;
                }
            }
            __syncthreads();
        }
        if (k == 0) {
            for (uint32_t i = tid; i < E::BMW; i += NT) bm[i] = 0;
            __syncthreads();
            for (uint32_t key = tid; key < NKEY; key += NT) {                               // every state through the first window
                const uint32_t x = walk(key);
                G[key] = x;
                if ((x & E::KEYM) != E::X_STOP) atomicOr(&bm[(x & E::KEYM) >> 5], 1u << (x & 31u));
            }
            __syncthreads();
            // rank of every distinct exit: exclusive prefix of the bitmap words' bit counts (the scan's scratch: Xd, not yet in use)
            constexpr uint32_t PER = (E::BMW + NT - 1) / NT;
            uint32_t mine = 0;
            for (uint32_t i = 0; i < PER; i++) { const uint32_t w = tid * PER + i; if (w < E::BMW) mine += __popc(bm[w]); }
            Xd[tid] = mine;
            __syncthreads();
            for (uint32_t d = 1; d < NT; d <<= 1) {
                const uint32_t y = tid >= d ? Xd[tid - d] : 0u;
                __syncthreads();
                Xd[tid] += y;
                __syncthreads();
            }
            uint32_t run = Xd[tid] - mine;
            if (tid == NT - 1) s_D = Xd[tid];
            __syncthreads();
            D = s_D;
            if (D > dcap) {                                                                 // (uniform) more distinct exits than are carried (dcap <= DCAP; less: a test hook): the hop parses this super-window outright
                for (uint32_t key = tid; key < NKEY; key += NT) G[key] = E::X_STOP;
                return;
            }
            for (uint32_t i = 0; i < PER; i++) { const uint32_t w = tid * PER + i; if (w < E::BMW) { pf[w] = (uint16_t)run; run += __popc(bm[w]); } }
            __syncthreads();
            for (uint32_t w = tid; w < E::BMW; w += NT) {                                   // the distinct exits, in rank order
                uint32_t m = bm[w], j = pf[w];
                while (m) { const uint32_t b = __ffs(m) - 1; Xd[j++] = w * 32 + b; m &= m - 1; }
            }
            __syncthreads();
        } else {
            for (uint32_t j = tid; j < D; j += NT) {                                        // the distinct walks through this window
                const uint32_t x = Xd[j];
                if ((x & E::KEYM) == E::X_STOP) continue;
                const uint32_t y = walk(x & E::KEYM);
                Xd[j] = compose(x, y);
            }
            __syncthreads();
        }
    }
    for (uint32_t key = tid; key < NKEY; key += NT) {                                       // every state: its first-window exit, then what became of that
        const uint32_t e = G[key];
        if ((e & E::KEYM) == E::X_STOP) continue;
        const uint32_t k1 = e & E::KEYM, w = k1 >> 5;
        const uint32_t x = Xd[pf[w] + __popc(bm[w] & ((1u << (k1 & 31u)) - 1u))];
        G[key] = compose(e, x);
    }
    __syncthreads();
}

// the hop for rasters of B bands: entries {position lo, hi, block, rungs (4 bits a band)} {factors in force (a byte a band)}
template <uint32_t B, int MODE, uint32_t UBW = 3>
__global__ void __launch_bounds__(64) walk_exitB_chain_kernel(const DecArgs a0, const uint32_t *xg, uint32_t nsuper, uint32_t s_begin, uint32_t s_count, WalkState16 *states, uint4 *entries) {
    typedef exitB<B, MODE == CM_BEST, UBW> E;
    typedef typename std::conditional<UBW == 3, uint8_t, uint16_t>::type TV;
    constexpr uint32_t UB = E::UB, RMASK = E::NRUNG - 1;
    const DecArgs a = dec_for_tile(a0, blockIdx.x);
    if (threadIdx.x) return;
    WalkState16 *S = states + blockIdx.x;
    if (S->bad) return;
    const uint64_t nblocks = a.g.nblocks, P0 = S->P;
    const uint32_t spec = (uint32_t)S->cf;
    uint4 *en = entries + (uint64_t)blockIdx.x * 2 * (nsuper + 2), *hd = en + 2 * (nsuper + 1);
    uint64_t P = P0, U = S->unit / B;
    uint32_t rr = (uint32_t)S->rungs & ((1u << (4 * B)) - 1), s = 0, cf = spec;
    bool bad = false, done = false;
    if (s_begin) {
        const uint4 h = *hd;
        if (h.y) return;
        const uint4 e = en[2 * s_begin];
        P = (uint64_t)e.x | (uint64_t)e.y << 32; U = e.z; rr = e.w; s = s_begin; cf = en[2 * s_begin + 1].x;
        bad = h.x != s_begin;
    }
    const uint32_t s_end = s_begin + s_count < nsuper ? s_begin + s_count : nsuper;
    const uint32_t *x0 = xg + (uint64_t)blockIdx.x * s_count * E::NKEY;
    while (!bad) {
        en[2 * s] = make_uint4((uint32_t)P, (uint32_t)(P >> 32), (uint32_t)U, rr);
        en[2 * s + 1] = make_uint4(cf, 0u, 0u, 0u);
        if (U >= nblocks) { done = true; break; }
        if (s >= s_end) { bad = s >= nsuper; break; }
        if (P >= a.in_bits) { bad = true; break; }
        const uint64_t base = P0 + (uint64_t)s * E::SW;
        uint32_t key = (uint32_t)(P - base) << (UB * B);
        for (uint32_t c = 0; c < B; c++) key |= ((rr >> (4 * c)) & RMASK) << (UB * c);
        const uint32_t x = x0[(uint64_t)(s - s_begin) * E::NKEY + key];
        s++;
        if ((x & E::KEYM) != E::X_STOP && !((x & E::X_DEP) && cf != spec)) {
            U += (x >> E::KEYB) & E::CNTM;
            const uint32_t k2 = x & E::KEYM;
            P = base + E::SW + (k2 >> (UB * B));
            rr = 0;
            for (uint32_t c = 0; c < B; c++) rr |= ((k2 >> (UB * c)) & RMASK) << (4 * c);
            continue;
        }
        // this super-window by the units themselves: whole blocks up to the first that starts behind it
        atomicOr(a.status, 64u);                                                            // (not an error: says that the walk was handed to this lane)
        Reader rd;
        rd.init(a.in32, a.in_bit0 + P, a.in_bit0 + a.in_bits);
        uint32_t rung[B];
        TV pc[B], g[16];
        for (uint32_t c = 0; c < B; c++) { rung[c] = (rr >> (4 * c)) & 15u; pc[c] = (TV)(cf >> (8 * sizeof(TV) * c)); }
        bool ok = true;
        const uint64_t end = base + E::SW;
        while (ok && U < nblocks) {
            const uint64_t pos = rd.position() - a.in_bit0;
            if (pos >= a.in_bits || pos >= end) break;
#pragma unroll
            for (uint32_t c = 0; c < B; c++) ok = parse_unit<TV, MODE>(rd, rung[c], pc[c], g) && ok;
            U++;
        }
        P = rd.position() - a.in_bit0;
        rr = 0; cf = 0;
        for (uint32_t c = 0; c < B; c++) { rr |= rung[c] << (4 * c); cf |= (uint32_t)pc[c] << (8 * sizeof(TV) * c); }
        if (!ok || (U < nblocks && (P < end || P - end >= E::PE))) { bad = true; break; }
    }
    *hd = make_uint4(s, done ? 1u : 0u, 0u, 0u);
    if (bad) { S->bad = 1u; atomicOr(a.status, 1u); }
}

template <uint32_t B, int MODE, uint32_t UBW = 3>
static bool launch_walk_exitB(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits) {
    typedef exitB<B, MODE == CM_BEST, UBW> E;
    typedef typename std::conditional<UBW == 3, uint8_t, uint16_t>::type TV;
    const uint32_t nt = a.ntiles;
    const uint64_t ns = (max_bits + E::SW - 1) / E::SW;
    const size_t fixed = (((size_t)nt * sizeof(WalkState16) + 255) & ~(size_t)255) + (((size_t)nt * (ns + 2) * 32 + 255) & ~(size_t)255);
    if (ns == 0 || ns > 0x7fffffffu || tab_bytes < fixed + (size_t)nt * E::NKEY * 4) return false;
    const uint64_t fit = (tab_bytes - fixed) / ((size_t)nt * E::NKEY * 4);
    const uint32_t nsuper = (uint32_t)ns, slab = (uint32_t)(fit < ns ? fit : ns);
    WalkState16 *states = (WalkState16 *)tab;
    uint4 *entries = (uint4 *)((uint8_t *)tab + (((size_t)nt * sizeof(WalkState16) + 255) & ~(size_t)255));
    uint32_t *xg = (uint32_t *)((uint8_t *)tab + fixed);
    for (uint32_t s0 = 0; s0 < nsuper; s0 += slab) {
        const uint32_t cnt = nsuper - s0 < slab ? nsuper - s0 : slab;
        { ProfScope ps("dec_index_table", st);
          hipLaunchKernelGGL((walk_exitB_kernel<B, MODE == CM_BEST, UBW>), dim3(cnt, nt), dim3(E::THREADS), E::LDS_BYTES, st, a, xg, s0, cnt, (const WalkState16 *)states, a.wide_band == 18 ? 64u : E::DCAP); }
        ProfScope ps("dec_index_serial", st);
        hipLaunchKernelGGL((walk_exitB_chain_kernel<B, MODE, UBW>), dim3(nt), dim3(64), 0, st, a, (const uint32_t *)xg, nsuper, s0, cnt, states, entries);
    }
    ProfScope ps("dec_index_serial", st);
    hipLaunchKernelGGL((walk_exit_units_kernel<TV, MODE>), dim3(nsuper, nt), dim3(64), ((E::SW + E::PE) / 32 + 4) * 4, st, a, (const WalkState16 *)states, (const uint4 *)entries, nsuper, E::SW, E::PE);
    return true;
}

// memory of the exit walk: states, entries, and the exits of as many super-windows as fit (at least one a tile)
template <uint32_t UB> static bool walk_exit_layout(uint32_t nt, uint64_t max_bits, size_t tab_bytes, uint32_t *nsuper, uint32_t *slab, size_t *x_off) {
    typedef exitW<UB> E;
    const uint64_t ns = (max_bits + E::SW - 1) / E::SW;
    const size_t fixed = (((size_t)nt * sizeof(WalkState16) + 255) & ~(size_t)255) + (((size_t)nt * (ns + 2) * 32 + 255) & ~(size_t)255);
    if (ns == 0 || ns > 0x7fffffffu || tab_bytes < fixed + (size_t)nt * E::NX * 4) return false;
    const uint64_t fit = (tab_bytes - fixed) / ((size_t)nt * E::NX * 4);
    *nsuper = (uint32_t)ns; *slab = (uint32_t)(fit < ns ? fit : ns); *x_off = fixed;
    return true;
}
template <uint32_t UB, typename T, int MODE>
static bool launch_walk_exit(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits) {
    typedef exitW<UB> E;
    uint32_t nsuper = 0, slab = 0;
    size_t x_off = 0;
    const uint32_t nt = a.ntiles;
    if (!walk_exit_layout<UB>(nt, max_bits, tab_bytes, &nsuper, &slab, &x_off)) return false;
    WalkState16 *states = (WalkState16 *)tab;
    uint4 *entries = (uint4 *)((uint8_t *)tab + (((size_t)nt * sizeof(WalkState16) + 255) & ~(size_t)255));
    uint32_t *xg = (uint32_t *)((uint8_t *)tab + x_off);
    for (uint32_t s0 = 0; s0 < nsuper; s0 += slab) {
        const uint32_t cnt = nsuper - s0 < slab ? nsuper - s0 : slab;
        { ProfScope ps("dec_index_table", st);
          hipLaunchKernelGGL((walk_exitW_kernel<UB, MODE == CM_BEST>), dim3(cnt, nt), dim3(E::THREADS), E::LDS_BYTES, st, a, xg, s0, cnt, (const WalkState16 *)states); }
        ProfScope ps("dec_index_serial", st);
        hipLaunchKernelGGL((walk_exit_chain_kernel<UB, MODE>), dim3(nt), dim3(64), 0, st, a, (const uint32_t *)xg, nsuper, s0, cnt, states, entries);
    }
    ProfScope ps("dec_index_serial", st);
    hipLaunchKernelGGL((walk_exit_units_kernel<T, MODE>), dim3(nsuper, nt), dim3(64), ((E::SW + E::PE) / 32 + 4) * 4, st, a, (const WalkState16 *)states, (const uint4 *)entries, nsuper, E::SW, E::PE);
    return true;
}

// Plain single-band 32/64-bit COMMON-FACTOR streams through the same exits (units with the signal code are parsed outright
// inside the walk).  False: not taken (no memory for it) -- the caller parses the stream with one lane.
bool launch_dec_walk_best(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits) {
    if (a.g.mode != CM_BEST || !walk_exit_lds_ok() || (a.g.tsz == 1 && a.g.ulen_sz != 4 && !(a.g.bands == 2 && a.g.ulen_sz == ULEN_UNIT))) return false;    // (8-bit: the lane-per-block decoder's block table; two bands: the lane-per-unit decoder's unit table)
    if (a.g.bands == 2 && a.g.tsz == 1) {          // 8-bit, two bands
        WalkState16 *states = (WalkState16 *)tab;
        { ProfScope ps("dec_index_serial", st);
          hipLaunchKernelGGL(walk_exit_zero_kernel<uint8_t>, dim3((uint32_t)((a.g.nseg + 255) / 256), a.ntiles), dim3(256), 0, st, a);
          hipLaunchKernelGGL((walk_probe_kernel<uint8_t, CM_BEST>), dim3(a.ntiles), dim3(64), 0, st, a, states, 8u); }
        return launch_walk_exitB<2, CM_BEST>(a, st, tab, tab_bytes, max_bits);
    }
    if (a.g.bands == 3 && a.g.tsz == 1) {          // 8-bit RGB
        WalkState16 *states = (WalkState16 *)tab;
        { ProfScope ps("dec_index_serial", st);
          hipLaunchKernelGGL(walk_exit_zero_kernel<uint8_t>, dim3((uint32_t)((a.g.nseg + 255) / 256), a.ntiles), dim3(256), 0, st, a);
          hipLaunchKernelGGL((walk_probe_kernel<uint8_t, CM_BEST>), dim3(a.ntiles), dim3(64), 0, st, a, states, 8u); }
        return launch_walk_exitB<3, CM_BEST>(a, st, tab, tab_bytes, max_bits);
    }
    if (a.g.bands != 1) return false;
    uint32_t ns = 0, slab = 0; size_t xo = 0;
    const uint32_t nt = a.ntiles;
    if (!(a.g.tsz == 1 ? walk_exit_layout<3>(nt, max_bits, tab_bytes, &ns, &slab, &xo) : a.g.tsz == 2 ? walk_exit_layout<4>(nt, max_bits, tab_bytes, &ns, &slab, &xo) : a.g.tsz == 4 ? walk_exit_layout<5>(nt, max_bits, tab_bytes, &ns, &slab, &xo) : walk_exit_layout<6>(nt, max_bits, tab_bytes, &ns, &slab, &xo))) return false;
    WalkState16 *states = (WalkState16 *)tab;
    { ProfScope ps("dec_index_serial", st);
      const dim3 zg((uint32_t)((a.g.nseg + 255) / 256), nt);
      if (a.g.tsz == 1) { hipLaunchKernelGGL(walk_exit_zero_kernel<uint8_t>, zg, dim3(256), 0, st, a); hipLaunchKernelGGL((walk_probe_kernel<uint8_t, CM_BEST>), dim3(nt), dim3(64), 0, st, a, states, 8u); }
      else if (a.g.tsz == 2) { hipLaunchKernelGGL(walk_exit_zero_kernel<uint16_t>, zg, dim3(256), 0, st, a); hipLaunchKernelGGL((walk_probe_kernel<uint16_t, CM_BEST>), dim3(nt), dim3(64), 0, st, a, states, 16u); }
      else if (a.g.tsz == 4) { hipLaunchKernelGGL(walk_exit_zero_kernel<uint32_t>, zg, dim3(256), 0, st, a); hipLaunchKernelGGL((walk_probe_kernel<uint32_t, CM_BEST>), dim3(nt), dim3(64), 0, st, a, states, 16u); }
      else { hipLaunchKernelGGL(walk_exit_zero_kernel<uint64_t>, zg, dim3(256), 0, st, a); hipLaunchKernelGGL((walk_probe_kernel<uint64_t, CM_BEST>), dim3(nt), dim3(64), 0, st, a, states, 16u); } }
    return a.g.tsz == 1 ? launch_walk_exit<3, uint8_t, CM_BEST>(a, st, tab, tab_bytes, max_bits) : a.g.tsz == 2 ? launch_walk_exit<4, uint16_t, CM_BEST>(a, st, tab, tab_bytes, max_bits)
         : a.g.tsz == 4 ? launch_walk_exit<5, uint32_t, CM_BEST>(a, st, tab, tab_bytes, max_bits) : launch_walk_exit<6, uint64_t, CM_BEST>(a, st, tab, tab_bytes, max_bits);
}
// table memory that takes `max_bits` of every stream through the exits in ONE round (states, entries, the exits of every
// super-window); 0: a raster of this shape, in a batch of this many tiles, does not walk by exits (launch_dec_walk_table's rules)
size_t walk_exit_bytes(uint32_t tsz, uint32_t bands, bool best, uint32_t nt, uint64_t max_bits) {
    uint64_t sw = 0, nx = 0;
    if (bands == 1 && nt <= 16) {
        switch (tsz) {
        case 1: sw = exitW<3>::SW; nx = exitW<3>::NX; break;
        case 2: sw = exitW<4>::SW; nx = exitW<4>::NX; break;
        case 4: sw = exitW<5>::SW; nx = exitW<5>::NX; break;
        default: sw = exitW<6>::SW; nx = exitW<6>::NX; break;
        }
    } else if (tsz == 1 && bands == 3 && nt <= 4) {
        if (best) { sw = exitB<3, true>::SW; nx = exitB<3, true>::NKEY; } else { sw = exitB<3, false>::SW; nx = exitB<3, false>::NKEY; }
    } else if (tsz == 1 && bands == 2 && nt <= 4) {
        if (best) { sw = exitB<2, true>::SW; nx = exitB<2, true>::NKEY; } else { sw = exitB<2, false>::SW; nx = exitB<2, false>::NKEY; }
    } else if (tsz == 2 && bands == 2 && !best && nt <= 4) { sw = exitB<2, false, 4>::SW; nx = exitB<2, false, 4>::NKEY;
    } else return 0;
    const uint64_t ns = (max_bits + sw - 1) / sw;
    const size_t fixed = (((size_t)nt * sizeof(WalkState16) + 255) & ~(size_t)255) + (((size_t)nt * (ns + 2) * 32 + 255) & ~(size_t)255);
    return fixed + (size_t)nt * nx * 4 * ns + 4096;
}

bool walk_exit_lds_ok() {
    static const bool lds_ok = [] {
        bool ok = true;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitB_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, exitB<3, false>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitB_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, exitB<3, true>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitB_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, exitB<2, false>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitB_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, exitB<2, true>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitB_kernel<2, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, exitB<2, false, 4>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitW_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, exitW<3>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitW_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, exitW<4>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitW_kernel<5, false>, hipFuncAttributeMaxDynamicSharedMemorySize, exitW<5>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitW_kernel<6, false>, hipFuncAttributeMaxDynamicSharedMemorySize, exitW<6>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitW_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, exitW<3>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitW_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, exitW<4>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitW_kernel<5, true>, hipFuncAttributeMaxDynamicSharedMemorySize, exitW<5>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_exitW_kernel<6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, exitW<6>::LDS_BYTES) == hipSuccess;
        return ok;
    }();
    return lds_ok;
}
// one band, FTL / BASE: the first segment parsed outright (8- and 16-bit data: the band is all the rungs there are), then the exits
bool walk_exits_one_band(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits) {
    const uint32_t nt = a.ntiles;
    WalkState16 *states = (WalkState16 *)tab;
    {
        ProfScope ps("dec_index_serial", st);
        if (a.g.tsz == 1) hipLaunchKernelGGL((walk_probe_kernel<uint8_t, CM_FTL>), dim3(nt), dim3(64), 0, st, a, states, 8u, 16u);
        else if (a.g.tsz == 2) hipLaunchKernelGGL((walk_probe_kernel<uint16_t, CM_FTL>), dim3(nt), dim3(64), 0, st, a, states, 16u, 16u);
        else if (a.g.tsz == 4) hipLaunchKernelGGL((walk_probe_kernel<uint32_t, CM_FTL>), dim3(nt), dim3(64), 0, st, a, states, 16u, 16u);
        else hipLaunchKernelGGL((walk_probe_kernel<uint64_t, CM_FTL>), dim3(nt), dim3(64), 0, st, a, states, 16u, 16u);
    }
    return a.g.tsz == 1 ? launch_walk_exit<3, uint8_t, CM_FTL>(a, st, tab, tab_bytes, max_bits) : a.g.tsz == 2 ? launch_walk_exit<4, uint16_t, CM_FTL>(a, st, tab, tab_bytes, max_bits)
         : a.g.tsz == 4 ? launch_walk_exit<5, uint32_t, CM_FTL>(a, st, tab, tab_bytes, max_bits) : launch_walk_exit<6, uint64_t, CM_FTL>(a, st, tab, tab_bytes, max_bits);
}
// 8-bit rasters of two or three bands, FTL / BASE: exits with the rung of every band in the state
bool walk_exits_rgb(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits) {
    WalkState16 *states = (WalkState16 *)tab;
    if (a.g.tsz == 2) {         // 16-bit rasters of two bands (dual-polarisation radar, complex samples): 556 positions x 256 rung pairs
        { ProfScope ps("dec_index_serial", st);
          hipLaunchKernelGGL((walk_probe_kernel<uint16_t, CM_FTL>), dim3(a.ntiles), dim3(64), 0, st, a, states, 16u, 16u); }
        return launch_walk_exitB<2, CM_FTL, 4>(a, st, tab, tab_bytes, max_bits);
    }
    { ProfScope ps("dec_index_serial", st);
      hipLaunchKernelGGL((walk_probe_kernel<uint8_t, CM_FTL>), dim3(a.ntiles), dim3(64), 0, st, a, states, 8u, 8u); }
    if (a.g.bands == 2) return launch_walk_exitB<2, CM_FTL>(a, st, tab, tab_bytes, max_bits);       // (two bands: 298 positions x 64 rung pairs: a twelfth of RGB's states)
    return launch_walk_exitB<3, CM_FTL>(a, st, tab, tab_bytes, max_bits);
}

}  // namespace qb3dev
