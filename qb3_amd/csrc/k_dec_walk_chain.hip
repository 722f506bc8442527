// qb3_amd/csrc/k_dec_walk_chain.hip -- plain streams through a TABLE of unit ends by position and ONE lane that follows it:
// 8-bit (any band count the lane-per-block decoder takes), 16-bit, 32/64-bit (a band of sixteen rungs)
#include "qb3_walk.h"

namespace qb3dev {

// ---- plain 8-bit streams (no index, no restart table): walk through a TABLE of unit lengths by position -----------
// Where a unit starts depends on every unit before it, but how long a unit WOULD be if it started at bit p with rung r
// depends on the bits alone.  walk_table_kernel computes that for every bit position of a slab of the stream and every
// rung, the whole chip at once (a code's length is its rung plus what its two low bits say, so the sixteen codes of a
// unit are four rounds of pointer doubling over "length of the next code"); walk_chain_kernel then follows the one
// chain that is real.  That walk is a pointer chase (measured on this chip: 48 cycles for a dependent LDS read, and
// about 8 more for every instruction between the value read and the next address), so the table is written in the
// form that makes the value read BE the next address: in windows of CW positions, a row of eight 16-bit entries per
// position, entry[o][r_in] = 16 * (o + unit length) | 2 * (rung after the unit's switch) | signal, o counted from
// the window's start.  16 * o' is the LDS offset of row o' in the window's buffer: one AND-OR with the buffer's base
// and the rung of the band that comes next gives the address of the next look-up.
namespace chain {
constexpr uint32_t CW = 3072;                           // positions at which the blocks of a window start
constexpr uint32_t ROWS = CW + 576;                     // ... and those their later units can start at (3 x 149 bits), in 3 x 64 rows for the loaders
constexpr uint32_t WIN_BYTES = ROWS * 16, SLOT = 65536; // a window in LDS: its rows, in a slot whose base has no bit below 2^16
constexpr uint32_t NSLOT = 2;                           // windows in LDS: one walked, one on its way
constexpr uint32_t TR_ENTRIES = CW / 2 + 16, TR_BYTES = TR_ENTRIES * 2;     // trail of a window: a unit is at least two bits
constexpr uint32_t TR0 = NSLOT * SLOT, META = TR0 + NSLOT * TR_BYTES, LDS_BYTES = META + 128;
static_assert(ROWS + 149 <= 4096 && WIN_BYTES <= SLOT && ROWS % 192 == 0 && ROWS >= CW + 448 && TR_BYTES % 16 == 0 && CW == 0xc00, "window layout (the walk tests position >= CW by its two top bits)");
constexpr uint32_t NP = ROWS + 160;                     // positions a table workgroup looks at: sixteen codes beyond the last switch
static_assert(NP % 32 == 0, "whole words");
}  // namespace chain

__global__ void __launch_bounds__(256) walk_table_kernel(const DecArgs a0, uint4 *tab, uint64_t slab0, uint32_t nwin, uint64_t tab_pitch) {
    using namespace chain;
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    const uint64_t p0 = slab0 + (uint64_t)blockIdx.x * CW;
    if (p0 >= a.in_bits + 2 * CW) return;                                   // (uniform) far beyond the stream: no walk comes here
    __shared__ uint32_t words[NP / 32 + 3];
    __shared__ uint8_t nA[7][NP], nB[7][NP];
    const uint32_t tid = threadIdx.x;
    const uint64_t q0 = a.in_bit0 + p0, w0 = q0 >> 5, endw = (a.in_bit0 + a.in_bits + 31) >> 5;
    const uint32_t sh = (uint32_t)q0 & 31;
    for (uint32_t i = tid; i < NP / 32 + 3; i += 256) words[i] = w0 + i < endw ? a.in32[w0 + i] : 0u;
    __syncthreads();
    auto bits = [&](uint32_t i) { const uint32_t b = sh + i, k = b >> 5; return __builtin_amdgcn_alignbit(words[k + 1], words[k], b & 31); };
    for (uint32_t i = tid; i < NP; i += 256) {                              // one code (reference QB3decode.h:119-129: r, r + 1 or r + 2 bits)
        const uint32_t x = bits(i), e = (x & 1) + ((x & 3) == 3);
#pragma unroll
        for (uint32_t r = 1; r < 8; r++) nA[r - 1][i] = (uint8_t)(r + e);
    }
    __syncthreads();
    uint8_t (*src)[NP] = nA, (*dst)[NP] = nB;
    uint32_t valid = NP;
#pragma unroll 1
    for (uint32_t lvl = 0; lvl < 4; lvl++) {                                // 2, 4, 8, 16 codes
        valid -= 9u << lvl;                                                 // (a code is at most nine bits)
        for (uint32_t r = 0; r < 7; r++)
            for (uint32_t i = tid; i < valid; i += 256) { const uint32_t n = src[r][i]; dst[r][i] = (uint8_t)(n + src[r][i + n]); }
        __syncthreads();
        uint8_t (*t)[NP] = src; src = dst; dst = t;
    }
    // valid = NP - 135 >= ROWS + 5: sixteen codes from every position a switch in this window can end on
    uint4 *out = tab + ((uint64_t)blockIdx.y * tab_pitch + (uint64_t)blockIdx.x * ROWS);
    for (uint32_t o = tid; o < ROWS; o += 256) {
        const uint32_t x = bits(o);
        uint32_t delta = 0; bool sig = false;
        const uint32_t cs = walk_switch<3>(x, delta, sig);                  // from rung 0: the step itself
        const uint32_t len0 = cs + (((x >> cs) & 1) ? 17 : 1);              // rung 0: one flag, then 16 raw bits
        uint32_t e[8];
#pragma unroll
        for (uint32_t rin = 0; rin < 8; rin++) {
            const uint32_t r = (rin + delta) & 7u;
            const uint32_t u = r ? cs + src[r ? r - 1 : 0][o + cs] : len0;
            e[rin] = ((o + u) << 4) | (r << 1) | (sig ? 1u : 0u);
        }
        out[o] = make_uint4(e[0] | e[1] << 16, e[2] | e[3] << 16, e[4] | e[5] << 16, e[6] | e[7] << 16);
    }
}

namespace chain {
typedef volatile __attribute__((address_space(3))) uint32_t *LdsFlag;
__device__ __forceinline__ uint32_t flag_get(uint32_t addr) { return *(LdsFlag)(uintptr_t)addr; }
__device__ __forceinline__ void flag_set(uint32_t addr, uint32_t v) { *(LdsFlag)(uintptr_t)addr = v; }
// words at META: what the waves of a workgroup tell each other (all counts of windows)
// F_READY[slot]: four words the walk reads at once: a window's three parts in LDS, and its trail slot written out
constexpr uint32_t F_READY = META /* [NSLOT][4] */, F_TRAILED = META + 32, F_GB0 = META + 40, F_NUNITS = META + 48,
                   F_WALKED = META + 56, F_STOP = META + 64;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bool ready(uint32_t slot, uint32_t want) {
    const u32x4_t f = *(volatile __attribute__((address_space(3))) u32x4_t *)(uintptr_t)(F_READY + 16 * slot);
    return f.x == want && f.y == want && f.z == want && f.w == want;
}
constexpr uint32_t SPIN_MAX = 1u << 22;                                     // (a wait that long is a defect: give up, flag the tile)

// One window: follow the chain from address A (slot base | 16 * position | 2 * rung of band 0) until a block starts
// beyond the window or the blocks run out.  T: LDS address of the trail (the address of every unit).  R[c]: slot
// base | 2 * rung of band c.  The loop is written out because the ORDER is the point -- every instruction between a
// read's value and the next read's issue costs its full latency (measured: 48 cycles the read, 8-10 each other), so
// between them stands only the AND-OR that makes the address; the trail write, the bookkeeping of the unit BEFORE
// (its band's new rung, the signal bit) and the loop's own tests all issue while a read is in flight.
#define CH_READ(E) "ds_read_u16 %[" #E "], %[A]\n"
#define CH_BOOK(E, Rc) "v_and_or_b32 %[" #Rc "], %[" #E "], 14, %[base]\n v_or_b32 %[bad], %[bad], %[" #E "]\n"
#define CH_STEP(E, Rn, off) "ds_write_b16 %[T], %[A] offset:" #off "\n s_waitcnt lgkmcnt(1)\n v_and_or_b32 %[A], %[" #E "], %[M], %[" #Rn "]\n"
#define CH_TOP "v_and_b32 %[t], 0xc000, %[A]\n v_cmp_eq_u32 vcc, 0xc000, %[t]\n"
#define CH_EXIT "s_cbranch_vccnz 2f\n s_cmp_eq_u32 %[left], 0\n s_cbranch_scc1 2f\n"
template <int B>
__device__ __forceinline__ void walk_asm(uint32_t &A, uint32_t &T, uint32_t (&R)[B], uint32_t &bad, uint32_t &left, uint32_t base) {
    uint32_t t;
    const uint32_t M = 0xfff0u;
    if constexpr (B == 3) {
        uint32_t e0, e1, e2 = R[2] & 14u;           // (the first turn books "the unit before": nothing changes)
        asm volatile(
            CH_READ(e0)
            "1:\n" CH_TOP CH_BOOK(e2, R2) CH_EXIT
            CH_STEP(e0, R1, 0)
            CH_READ(e1) CH_BOOK(e0, R0) CH_STEP(e1, R2, 2)
            CH_READ(e2) CH_BOOK(e1, R1) CH_STEP(e2, R0, 4)
            CH_READ(e0)
            "v_add_u32 %[T], 6, %[T]\n s_sub_u32 %[left], %[left], 1\n s_branch 1b\n"
            "2:\n s_waitcnt lgkmcnt(0)\n"
            : [A] "+v"(A), [T] "+v"(T), [R0] "+v"(R[0]), [R1] "+v"(R[1]), [R2] "+v"(R[2]), [bad] "+v"(bad), [left] "+s"(left),
              [e0] "=&v"(e0), [e1] "=&v"(e1), [e2] "+v"(e2), [t] "=&v"(t)
            : [M] "s"(M), [base] "v"(base)
            : "vcc", "scc", "memory");
    } else if constexpr (B == 2) {
        uint32_t e0, e1 = R[1] & 14u;
        asm volatile(
            CH_READ(e0)
            "1:\n" CH_TOP CH_BOOK(e1, R1) CH_EXIT
            CH_STEP(e0, R1, 0)
            CH_READ(e1) CH_BOOK(e0, R0) CH_STEP(e1, R0, 2)
            CH_READ(e0)
            "v_add_u32 %[T], 4, %[T]\n s_sub_u32 %[left], %[left], 1\n s_branch 1b\n"
            "2:\n s_waitcnt lgkmcnt(0)\n"
            : [A] "+v"(A), [T] "+v"(T), [R0] "+v"(R[0]), [R1] "+v"(R[1]), [bad] "+v"(bad), [left] "+s"(left),
              [e0] "=&v"(e0), [e1] "+v"(e1), [t] "=&v"(t)
            : [M] "s"(M), [base] "v"(base)
            : "vcc", "scc", "memory");
    } else if constexpr (B == 4) {
        uint32_t e0, e1, e2, e3 = R[3] & 14u;
        asm volatile(
            CH_READ(e0)
            "1:\n" CH_TOP CH_BOOK(e3, R3) CH_EXIT
            CH_STEP(e0, R1, 0)
            CH_READ(e1) CH_BOOK(e0, R0) CH_STEP(e1, R2, 2)
            CH_READ(e2) CH_BOOK(e1, R1) CH_STEP(e2, R3, 4)
            CH_READ(e3) CH_BOOK(e2, R2) CH_STEP(e3, R0, 6)
            CH_READ(e0)
            "v_add_u32 %[T], 8, %[T]\n s_sub_u32 %[left], %[left], 1\n s_branch 1b\n"
            "2:\n s_waitcnt lgkmcnt(0)\n"
            : [A] "+v"(A), [T] "+v"(T), [R0] "+v"(R[0]), [R1] "+v"(R[1]), [R2] "+v"(R[2]), [R3] "+v"(R[3]), [bad] "+v"(bad), [left] "+s"(left),
              [e0] "=&v"(e0), [e1] "=&v"(e1), [e2] "=&v"(e2), [e3] "+v"(e3), [t] "=&v"(t)
            : [M] "s"(M), [base] "v"(base)
            : "vcc", "scc", "memory");
    } else {
        // one band: the rung that comes next is the one just read: the next address is the entry without its signal bit
        uint32_t e0;
        const uint32_t M1 = 0xfffeu;
        asm volatile(
            CH_READ(e0)
            "1:\n" CH_TOP CH_EXIT
            "ds_write_b16 %[T], %[A]\n s_waitcnt lgkmcnt(1)\n v_or_b32 %[bad], %[bad], %[e0]\n v_and_or_b32 %[A], %[e0], %[M], %[base]\n"
            CH_READ(e0)
            "v_add_u32 %[T], 2, %[T]\n s_sub_u32 %[left], %[left], 1\n s_branch 1b\n"
            "2:\n s_waitcnt lgkmcnt(0)\n"
            : [A] "+v"(A), [T] "+v"(T), [bad] "+v"(bad), [left] "+s"(left), [e0] "=&v"(e0), [t] "=&v"(t)
            : [M] "s"(M1), [base] "v"(base)
            : "vcc", "scc", "memory");
        R[0] = base | (A & 14u);
    }
}
#undef CH_READ
#undef CH_BOOK
#undef CH_STEP
#undef CH_TOP
#undef CH_EXIT
}  // namespace chain

// A workgroup per tile, eight waves: wave 0 (one lane) walks; waves 1-6 bring windows of the table into LDS -- three
// waves a window, one group the even windows and one the odd, so that a group's loads from HBM are in flight while the
// window before theirs is walked; wave 7 turns the trail of a walked window into unit lengths and segment entries.
template <int B>
__global__ void __launch_bounds__(512) walk_chain_kernel(const DecArgs a0, const uint4 *tab, uint64_t slab0, uint32_t nwin, uint64_t tab_pitch,
                                                         WalkState *states, uint32_t first_round) {
    using namespace chain;
    const DecArgs a = dec_for_tile(a0, blockIdx.x);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t NB = a.g.seg_blocks, nblocks = (uint32_t)a.g.nblocks;
    WalkState *S = states + blockIdx.x;
    const uint64_t P0 = first_round ? 0 : S->P;                             // in stream bits
    const uint32_t gb_in = first_round ? 0 : S->gb;
    const uint32_t R_in = first_round ? 0 : S->rungs;
    const uint64_t slab_end = slab0 + (uint64_t)nwin * CW;
    if (P0 < slab0 || P0 >= slab_end || P0 >= a.in_bits || gb_in >= nblocks) return;       // (uniform) nothing of this tile in this slab
    const uint32_t k0 = (uint32_t)((P0 - slab0) / CW);                      // the window the walk starts in
    if (tid < 32) {     // (the first two windows find their trail slots free)
        uint32_t v = tid == (F_STOP - META) / 4 ? 0xffffffffu : 0u;
        if (tid == (k0 % NSLOT) * 4 + 3) v = k0 + 1;
        if (tid == ((k0 + 1) % NSLOT) * 4 + 3) v = k0 + 2;
        ((uint32_t *)(smem + META))[tid] = v;
    }
    __syncthreads();
    const uint4 *wt = tab + (uint64_t)blockIdx.x * tab_pitch;

    if (wave == 0) {
        if (lane) return;
        uint32_t R[B], bad = first_round ? 0 : S->bad;
        uint32_t left = nblocks - gb_in, k = k0, o = (uint32_t)((P0 - slab0) % CW);
        for (int c = 0; c < B; c++) R[c] = ((R_in >> (4 * c)) & 7u) << 1;
        bool stuck = false;
        uint64_t Pn = P0;                                                   // where the next block starts
        while (true) {
            const uint32_t s = k % NSLOT, base = s * SLOT;
            uint32_t spin = 0;                                              // the window in LDS (three parts), and this trail slot written out
            while (!ready(s, k + 1) && ++spin < SPIN_MAX) __builtin_amdgcn_s_sleep(1);
            if (spin >= SPIN_MAX) { stuck = true; break; }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            for (int c = 0; c < B; c++) R[c] = (R[c] & 14u) | base;
            uint32_t A = base | (o << 4) | (R[0] & 14u), T = TR0 + s * TR_BYTES;
            const uint32_t T0 = T, left0 = left;
            walk_asm<B>(A, T, R, bad, left, base);
            *(volatile __attribute__((address_space(3))) uint16_t *)(uintptr_t)T = (uint16_t)A;        // where the next block starts: the last unit's end
            flag_set(F_GB0 + 4 * s, nblocks - left0);
            flag_set(F_NUNITS + 4 * s, (T - T0) >> 1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            flag_set(F_TRAILED + 4 * s, k + 1);
            const uint32_t oe = (A & 0xfff0u) >> 4;
            Pn = slab0 + (uint64_t)k * CW + oe;
            k++;
            flag_set(F_WALKED, k - k0);
            if (!left) break;                                               // the blocks ran out
            o = oe - CW;                                                    // (the walk left the window: oe >= CW)
            if (k >= nwin || Pn >= a.in_bits) break;                        // the slab ends here, or the stream does (a damaged one)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        flag_set(F_STOP, k);                                                // k windows were walked
        uint32_t Rn = 0;
        for (int c = 0; c < B; c++) Rn |= ((R[c] >> 1) & 7u) << (4 * c);
        S->P = stuck ? ~0ull : Pn; S->gb = nblocks - left; S->rungs = Rn; S->bad = (bad & 1u) | (stuck ? 1u : 0u);
        // (a stream that ends with blocks left -- cut short -- is not walked on: the call goes to the one-lane parser, whose reader
        // gives zeros behind the end like the reference's)
        if ((bad & 1u) || stuck || (left && Pn >= a.in_bits)) atomicOr(a.status, 1u);
        return;
    }
    if (wave <= 6) {
        // loaders: group g (three waves, a third of the rows each) takes the windows of parity g, into slot g
        const uint32_t g = (wave - 1) / 3, part = (wave - 1) % 3;
        constexpr uint32_t NV = ROWS / 192;                                 // sixty-four rows a load: loads of a wave per window
        for (uint32_t k = k0 + ((k0 ^ g) & 1u); k < nwin; k += 2) {
            const uint4 *src = wt + (uint64_t)k * ROWS;
            // (named values, not an array: the array went to scratch memory)
#define CH_REP19(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18)
            static_assert(NV == 19, "CH_REP19");
#define CH_LD(i) const uint4 v##i = src[lane + 64 * (part + 3 * i)];
            CH_REP19(CH_LD)
            uint32_t spin = 0;
            bool stop = false;
            while (true) {                                                  // the slot is free when the window two back has been walked
                if (flag_get(F_STOP) != 0xffffffffu) { stop = true; break; }
                if (flag_get(F_WALKED) + NSLOT > k - k0) break;
                if (++spin >= SPIN_MAX) { stop = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (stop) break;
            uint4 *slot = (uint4 *)(smem + g * SLOT);
            // (a pause after every store: nineteen 1 KB stores back to back hold the LDS long enough to stall the walk's reads)
#define CH_ST(i) slot[lane + 64 * (part + 3 * i)] = v##i; __builtin_amdgcn_s_sleep(3);
            CH_REP19(CH_ST)
#undef CH_ST
#undef CH_LD
#undef CH_REP19
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) flag_set(F_READY + 16 * g + 4 * part, k + 1);
        }
        return;
    }
    // writer: the trail of a walked window gives the position and rung of every unit: lengths by difference
    for (uint32_t k = k0;; k++) {
        const uint32_t s = k % NSLOT;
        uint32_t spin = 0;
        bool stop = false;
        while (flag_get(F_TRAILED + 4 * s) != k + 1) {
            const uint32_t st = flag_get(F_STOP);
            if ((st != 0xffffffffu && k >= st) || ++spin >= SPIN_MAX) { stop = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (stop) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint32_t gb0 = flag_get(F_GB0 + 4 * s), n = flag_get(F_NUNITS + 4 * s);
        const uint16_t *tr = (const uint16_t *)(smem + TR0 + s * TR_BYTES);
        const uint64_t wpos = slab0 + (uint64_t)k * CW;
        uint8_t *ul = (uint8_t *)a.idx.ulen + (uint64_t)gb0 * B;
        for (uint32_t j = lane; j < n; j += 64) {
            const uint32_t o0 = tr[j] >> 4, o1 = tr[j + 1] >> 4;
            ul[j] = (uint8_t)(o1 - o0);
            if (j % B == 0 && (gb0 + j / B) % NB == 0) {
                const uint64_t seg = (gb0 + j / B) / NB;
                a.idx.bitpos[seg] = wpos + o0;
#pragma unroll
                for (int c = 0; c < B; c++) a.idx.rung[seg * B + c] = (uint8_t)((tr[j + c] >> 1) & 7u);
            }
        }
        // (no release fence: the trail has been READ -- LDS operations of a wave are in order -- and the index stores may still be
        // on their way; a fence would hold the slot for a memory round trip per window)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) flag_set(F_READY + 16 * s + 12, k + NSLOT + 1);      // the slot is free for the window that takes it next
    }
}

// ---- the same for plain 16-bit streams, and for the 8-bit streams the hand-ordered chain above does not take -------
// Sixteen rungs, codes of up to 17 bits, units of up to 278: a row is sixteen 16-bit entries (32 bytes), a window 1536
// positions (48 KB, two of them side by side in LDS: slot 1 is reached through the read's immediate offset, so the
// addresses the walk carries stay window-relative).  A block has up to 16 bands here and can be longer than a window,
// so the walk changes windows between any two UNITS (every look-up position is inside the window: no margin rows,
// 32 table bytes per stream bit); the rungs of the bands live in a small LDS array, and the trail holds the entries
// read (next position | rung out), from which a writer wave derives unit lengths and segment entries.  The walk loop is
// plain C++ here (about 1.5 x the cycles per unit of the hand-ordered 8-bit loop).
// UB = 3: 8-bit streams of 2 or more than 4 bands (and the common-factor streams of any band count but one): eight rungs, rows of
// sixteen bytes, windows of 3072 positions -- the same kernels, half the table bytes a stream bit.
// CF (round 4): COMMON-FACTOR streams of several bands.  A unit with the signal code -- a common-factor or index unit: its values
// decide the rung it leaves (reference QB3decode.h:619-716) -- is marked in its table entry; the walking lane parses it outright from
// the stream (parse_unit over global memory: the few per cent of units that have it cost microseconds each) and puts the entry the
// table could not hold into the trail; the factor in force per band is part of the walk's state, and once a unit has brought one the
// lane leaves the factors in force at every segment start (idx.cf, zeroed beforehand).  The writer wave makes the lane-per-unit
// decoder's dwords (bits | entering rung) or, for 8-bit streams of 1 / 3 / 4 bands, adds the units up into the block table.
template <uint32_t UB> struct chainN {
    static constexpr uint32_t NR = 1u << UB, ROWB = 2 * NR, CW = 49152 / ROWB, WIN_BYTES = CW * ROWB, WIN_U4 = WIN_BYTES / 16;     // 49152 bytes, 3072 sixteen-byte pieces
    static constexpr uint32_t MAXC = NR + 1, MAXU = UB + 2 + 16 * MAXC;         // longest code, longest plain unit (149 / 278 bits)
    static constexpr uint32_t NP = CW + ((6 + 15 * MAXC + 2 + 31) & ~31u);       // positions a table workgroup looks at
    static constexpr uint32_t TR_BYTES = ((CW / 2 + 8) * 2 + 15) & ~15u;         // trail of a window: a unit is at least two bits
    static constexpr uint32_t NWS = (CW + 384 + 32 + 31) / 32, STR_BYTES = (NWS * 4 + 15) & ~15u;         // CF: the stream words of a window and of the longest unit that starts in it
    static constexpr uint32_t TR0 = 2 * WIN_BYTES, RS0 = TR0 + 2 * TR_BYTES, WR0 = RS0 + 64, CFS0 = WR0 + 64, STR0 = CFS0 + 64, META = STR0 + 2 * STR_BYTES, LDS_BYTES = META + 128;
    static constexpr uint32_t F_READY = META /* [2][4] */, F_TRAILED = META + 32, F_NUNITS = META + 40, F_O0 = META + 48,
                              F_WALKED = META + 56, F_STOP = META + 60, F_U0 = META + 64 /* u64[2] */;
    static_assert(NWS <= 128 && NP % 32 == 0 && NP >= CW + 6 + 15 * MAXC + 2 && WIN_BYTES == 0xc000 && (CW + MAXU + 64) * ROWB < 65536 && WIN_U4 % 192 == 0, "window layout");
};

template <uint32_t UB>
__global__ void __launch_bounds__(256) walk_tableN_kernel(const DecArgs a0, uint4 *tab, uint64_t slab0, uint32_t nwin, uint64_t tab_pitch) {
    typedef chainN<UB> W;
    constexpr uint32_t NR = W::NR, NP = W::NP, CW = W::CW, ROWB = W::ROWB, MAXC = W::MAXC;
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    const uint64_t p0 = slab0 + (uint64_t)blockIdx.x * CW;
    if (p0 >= a.in_bits + 2 * CW) return;                                   // (uniform) far beyond the stream: no walk comes here
    __shared__ uint32_t words[NP / 32 + 3];
    __shared__ uint8_t nA[NR - 1][NP], nB[NR - 1][NP];
    const uint32_t tid = threadIdx.x;
    const uint64_t q0 = a.in_bit0 + p0, w0 = q0 >> 5, endw = (a.in_bit0 + a.in_bits + 31) >> 5;
    const uint32_t sh = (uint32_t)q0 & 31;
    for (uint32_t i = tid; i < NP / 32 + 3; i += 256) words[i] = w0 + i < endw ? a.in32[w0 + i] : 0u;
    __syncthreads();
    auto bits = [&](uint32_t i) { const uint32_t b = sh + i, k = b >> 5; return __builtin_amdgcn_alignbit(words[k + 1], words[k], b & 31); };
    for (uint32_t i = tid; i < NP; i += 256) {                              // one code: r, r + 1 or r + 2 bits
        const uint32_t x = bits(i), e = (x & 1) + ((x & 3) == 3);
#pragma unroll
        for (uint32_t r = 1; r < NR; r++) nA[r - 1][i] = (uint8_t)(r + e);
    }
    __syncthreads();
    uint8_t (*src)[NP] = nA, (*dst)[NP] = nB;
    uint32_t valid = NP;
#pragma unroll 1
    for (uint32_t lvl = 0; lvl < 3; lvl++) {                                // 2, 4, 8 codes (eight codes are at most 136 bits: a byte)
        valid -= MAXC << lvl;
        for (uint32_t r = 0; r < NR - 1; r++)
            for (uint32_t i = tid; i < valid; i += 256) { const uint32_t n = src[r][i]; dst[r][i] = (uint8_t)(n + src[r][i + n]); }
        __syncthreads();
        uint8_t (*t)[NP] = src; src = dst; dst = t;
    }
    // src = eight codes, valid for i < NP - 7 * MAXC; sixteen = eight + eight, formed here (up to 272: not a byte)
    uint4 *out = tab + ((uint64_t)blockIdx.y * tab_pitch + (uint64_t)blockIdx.x * W::WIN_U4);
    for (uint32_t o = tid; o < CW; o += 256) {
        const uint32_t x = bits(o);
        uint32_t delta = 0; bool sig = false;
        const uint32_t cs = walk_switch<UB>(x, delta, sig);                 // from rung 0: the step itself
        const uint32_t len0 = cs + (((x >> cs) & 1) ? 17 : 1);              // rung 0: one flag, then 16 raw bits
        uint32_t e[NR];
#pragma unroll
        for (uint32_t rin = 0; rin < NR; rin++) {
            const uint32_t r = (rin + delta) & (NR - 1);
            uint32_t u = len0;
            if (r) { const uint32_t n8 = src[r - 1][o + cs]; u = cs + n8 + src[r - 1][o + cs + n8]; }
            e[rin] = ((o + u) * ROWB) | (r << 1) | (sig ? 1u : 0u);
        }
        if constexpr (NR == 16) {
            out[2 * o] = make_uint4(e[0] | e[1] << 16, e[2] | e[3] << 16, e[4] | e[5] << 16, e[6] | e[7] << 16);
            out[2 * o + 1] = make_uint4(e[8] | e[9] << 16, e[10] | e[11] << 16, e[12] | e[13] << 16, e[14] | e[15] << 16);
        } else out[o] = make_uint4(e[0] | e[1] << 16, e[2] | e[3] << 16, e[4] | e[5] << 16, e[6] | e[7] << 16);
    }
}

// A workgroup per tile, eight waves: wave 0 (one lane) walks; waves 1-6 load windows (two groups of three, as in the
// 8-bit kernel); wave 7 writes unit lengths and segment entries from the trail.
template <uint32_t UB, bool CF>
__global__ void __launch_bounds__(512) walk_chainN_kernel(const DecArgs a0, const uint4 *tab, uint64_t slab0, uint32_t nwin, uint64_t tab_pitch,
                                                          WalkState16 *states, uint32_t first_round) {
    typedef chainN<UB> W;
    typedef typename std::conditional<UB == 3, uint8_t, uint16_t>::type T;
    constexpr uint32_t NR = W::NR, CW = W::CW, ROWB = W::ROWB, WIN_BYTES = W::WIN_BYTES, WIN_U4 = W::WIN_U4, TR_BYTES = W::TR_BYTES, TR0 = W::TR0, RS0 = W::RS0, WR0 = W::WR0,
                       CFS0 = W::CFS0, STR0 = W::STR0, STR_BYTES = W::STR_BYTES, NWS = W::NWS, META = W::META, F_READY = W::F_READY, F_TRAILED = W::F_TRAILED, F_NUNITS = W::F_NUNITS, F_O0 = W::F_O0, F_WALKED = W::F_WALKED,
                       F_STOP = W::F_STOP, F_U0 = W::F_U0;
    using chain::flag_get; using chain::flag_set; using chain::SPIN_MAX;
    const DecArgs a = dec_for_tile(a0, blockIdx.x);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t B = a.g.bands, NB = a.g.seg_blocks;
    const uint64_t nunits = a.g.nblocks * B;
    WalkState16 *S = states + blockIdx.x;
    const uint64_t P0 = first_round ? 0 : S->P, U_in = first_round ? 0 : S->unit, R_in = first_round ? 0 : S->rungs;
    const uint64_t slab_end = slab0 + (uint64_t)nwin * CW;
    if (P0 < slab0 || P0 >= slab_end || P0 >= a.in_bits || U_in >= nunits) return;         // (uniform) nothing of this tile in this slab
    const uint32_t k0 = (uint32_t)((P0 - slab0) / CW);                      // the window the walk starts in
    volatile uint32_t *rs = (volatile uint32_t *)(smem + RS0), *wr = (volatile uint32_t *)(smem + WR0);     // rung * 2 per band: the walk's, the writer's
    volatile uint32_t *cfl = (volatile uint32_t *)(smem + CFS0);            // CF: the factor in force per band
    if (tid < 32) {     // (the first two windows find their trail slots free)
        uint32_t v = tid == (F_STOP - META) / 4 ? 0xffffffffu : 0u;
        if (tid == (k0 & 1) * 4 + 3) v = k0 + 1;
        if (tid == ((k0 + 1) & 1) * 4 + 3) v = k0 + 2;
        ((uint32_t *)(smem + META))[tid] = v;
    }
    if (tid < 16) { const uint32_t r2 = (uint32_t)((R_in >> (4 * tid)) & 15u) << 1; rs[tid] = r2; wr[tid] = r2; cfl[tid] = (CF && !first_round) ? S->cfs[tid] : 0u; }
    __syncthreads();
    const uint4 *wt = tab + (uint64_t)blockIdx.x * tab_pitch;
    auto ready = [&](uint32_t slot, uint32_t want) {
        const chain::u32x4_t f = *(volatile __attribute__((address_space(3))) chain::u32x4_t *)(uintptr_t)(F_READY + 16 * slot);
        return f.x == want && f.y == want && f.z == want && f.w == want;
    };

    if (wave == 0) {
        if (lane) return;
        uint32_t bad = first_round ? 0 : S->bad, k = k0, o = (uint32_t)((P0 - slab0) % CW);
        uint64_t U = U_in, Pn = P0;
        uint32_t c = (uint32_t)(U % B);
        bool stuck = false;
        const uint32_t useg_n = B * NB;                                     // CF: units of a segment; units left in the one the walk is in
        uint32_t useg = (uint32_t)(U % useg_n), cf_any = (CF && !first_round) ? S->cf_any : 0u;
        useg = useg ? useg_n - useg : 0u;
        while (true) {
            const uint32_t s = k & 1;
            uint32_t spin = 0;
            while (!ready(s, k + 1) && ++spin < SPIN_MAX) __builtin_amdgcn_s_sleep(1);
            if (spin >= SPIN_MAX) { stuck = true; break; }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            uint32_t A = o * ROWB + rs[c], n = 0;
            const uint64_t left64 = nunits - U;
            uint32_t left = left64 > 0xffffffffull ? 0xffffffffu : (uint32_t)left64;
            typedef const __attribute__((address_space(3))) uint16_t *LdsHalf;
            typedef __attribute__((address_space(3))) uint16_t *LdsHalfW;
            typedef __attribute__((address_space(3))) uint32_t *LdsWordW;
            const uint32_t wbase = s * WIN_BYTES;
            LdsHalfW trw = (LdsHalfW)(uintptr_t)(TR0 + s * TR_BYTES);
            LdsWordW rsw = (LdsWordW)(uintptr_t)RS0;
            constexpr uint32_t M = 0xffffu & ~(ROWB - 1), RM = (NR - 1) << 1;
            if constexpr (CF) {
                // a unit per turn; a unit with the signal code is parsed from the stream (its table entry knows neither its length nor
                // the rung it leaves); at every segment start, once a unit has brought a factor, the factors in force go to the index
                const uint64_t wpos = slab0 + (uint64_t)k * CW;
                // (as below, ONE dependent LDS read a unit: the rung of the band that comes next is fetched a unit ahead -- with two bands
                // "the band after next" is this one, whose rung the entry just read holds)
                uint32_t cn = c + 1 == B ? 0 : c + 1;
                uint32_t rn = rsw[cn];
                while (A < CW * ROWB && left) {
                    const uint32_t cn2 = cn + 1 == B ? 0 : cn + 1;
                    uint32_t e = *(LdsHalf)(uintptr_t)(wbase + A);
                    uint32_t r2 = rsw[cn2];
                    if (cf_any) {           // (from the first unit that brought a factor on: the factors in force at every segment start)
                        if (useg == 0) {
                            const uint64_t seg = (U + n) / useg_n;
                            for (uint32_t cc = 0; cc < B; cc++) ((T *)a.idx.cf)[seg * B + cc] = (T)cfl[cc];
                            useg = useg_n;
                        }
                        useg--;
                    }
                    if (e & 1u) {           // parsed from the window's stream words, which the loaders put beside its table rows
                        const uint32_t ou = A / ROWB;
                        ReaderT<LdsWords> rd;
                        const uint32_t at = (uint32_t)((a.in_bit0 + wpos) & 31) + ou;
                        rd.init((LdsWords)(uintptr_t)(STR0 + s * STR_BYTES), at, 32ull * NWS);
                        uint32_t rung = (A & RM) >> 1, fl = 0;
                        T cf = (T)cfl[c], g[16];
                        const bool ok = parse_unit<T, CM_BEST, ReaderT<LdsWords>>(rd, rung, cf, g, &fl);
                        const uint64_t len = rd.position() - at;
                        if (!ok || len > W::MAXU + 64) bad |= 1u;
                        if (fl & 2u) {
                            cfl[c] = (uint32_t)cf;
                            if (!cf_any) { cf_any = 1; const uint32_t m = (uint32_t)((U + n + 1) % useg_n); useg = m ? useg_n - m : 0u; }       // (segments from the next unit on)
                        }
                        e = (uint32_t)((ou + (len > W::MAXU + 64 ? 1u : (uint32_t)len)) * ROWB) | (rung << 1);
                    }
                    trw[n++] = (uint16_t)e;
                    rsw[c] = e & RM;
                    if (cn2 == c) r2 = e & RM;
                    if (cn == c) rn = e & RM;           // (one band)
                    A = (e & M) | rn;
                    c = cn; cn = cn2; rn = r2; left--;
                }
            } else
            // a unit per turn, until one starts beyond the window: ONE dependent LDS read a unit -- the rung of the band
            // that comes next is fetched a unit ahead (from registers for one or two bands, else from the LDS array)
            if (B == 1) {
                while (A < CW * ROWB && left) {
                    const uint32_t e = *(LdsHalf)(uintptr_t)(wbase + A);
                    trw[n++] = (uint16_t)e; bad |= e;
                    A = e & (M | RM);
                    left--;
                }
                rsw[0] = A & RM;
            } else if (B == 2) {
                uint32_t rn = rsw[c ^ 1];
                while (A < CW * ROWB && left) {
                    const uint32_t e = *(LdsHalf)(uintptr_t)(wbase + A);
                    trw[n++] = (uint16_t)e; bad |= e;
                    A = (e & M) | rn;
                    rn = e & RM;                // this band comes again after the next unit
                    c ^= 1; left--;
                }
                rsw[c] = A & RM; rsw[c ^ 1] = rn;
            } else {
                uint32_t cn = c + 1 == B ? 0 : c + 1;
                uint32_t rn = rsw[cn];
                while (A < CW * ROWB && left) {
                    const uint32_t cn2 = cn + 1 == B ? 0 : cn + 1;
                    const uint32_t e = *(LdsHalf)(uintptr_t)(wbase + A);
                    const uint32_t r2 = rsw[cn2];                           // (three bands or more: not the band being written below)
                    trw[n++] = (uint16_t)e; bad |= e;
                    rsw[c] = e & RM;
                    A = (e & M) | rn;
                    c = cn; cn = cn2; rn = r2; left--;
                }
            }
            *(volatile uint64_t *)(smem + F_U0 + 8 * s) = U;
            flag_set(F_NUNITS + 4 * s, n);
            flag_set(F_O0 + 4 * s, o);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            flag_set(F_TRAILED + 4 * s, k + 1);
            U += n;
            const uint32_t oe = A / ROWB;
            Pn = slab0 + (uint64_t)k * CW + oe;
            k++;
            flag_set(F_WALKED, k - k0);
            if (U >= nunits) break;                                         // the units ran out
            o = oe - CW;                                                    // (the walk left the window: oe >= CW)
            if (k >= nwin || Pn >= a.in_bits) break;                        // the slab ends here, or the stream does (a damaged one)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        flag_set(F_STOP, k);
        uint64_t Rn = 0;
        for (uint32_t i = 0; i < B; i++) Rn |= (uint64_t)((rs[i] >> 1) & 15u) << (4 * i);
        S->P = stuck ? ~0ull : Pn; S->unit = U; S->rungs = Rn; S->bad = (bad & 1u) | (stuck ? 1u : 0u);
        if constexpr (CF) { for (uint32_t i = 0; i < B; i++) S->cfs[i] = cfl[i]; S->cf_any = cf_any; }
        if ((bad & 1u) || stuck || (U < nunits && Pn >= a.in_bits)) atomicOr(a.status, 1u);      // (cut short: the one-lane parser takes it)
        return;
    }
    if (wave <= 6) {
        const uint32_t g = (wave - 1) / 3, part = (wave - 1) % 3;
        constexpr uint32_t NV = WIN_U4 / 192;
        static_assert(NV == 16, "CH16_REP");
        for (uint32_t k = k0 + ((k0 ^ g) & 1u); k < nwin; k += 2) {
            const uint4 *src = wt + (uint64_t)k * WIN_U4;
#define CH16_REP(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define CH16_LD(i) const uint4 v##i = src[lane + 64 * (part + 3 * i)];
            CH16_REP(CH16_LD)
            uint32_t sw0 = 0, sw1 = 0;                                      // CF: the window's stream words (the first loader wave of the group)
            if (CF && part == 0) {
                const uint64_t w0 = (a.in_bit0 + slab0 + (uint64_t)k * CW) >> 5, endw = (a.in_bit0 + a.in_bits + 31) >> 5;
                sw0 = (lane < NWS && w0 + lane < endw) ? a.in32[w0 + lane] : 0u;
                sw1 = (lane + 64 < NWS && w0 + lane + 64 < endw) ? a.in32[w0 + lane + 64] : 0u;
            }
            uint32_t spin = 0;
            bool stop = false;
            while (true) {                                                  // the slot is free when the window two back has been walked
                if (flag_get(F_STOP) != 0xffffffffu) { stop = true; break; }
                if (flag_get(F_WALKED) + 2 > k - k0) break;
                if (++spin >= SPIN_MAX) { stop = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (stop) break;
            uint4 *slot = (uint4 *)(smem + g * WIN_BYTES);
            if (CF && part == 0) {
                uint32_t *sws = (uint32_t *)(smem + STR0 + g * STR_BYTES);
                if (lane < NWS) sws[lane] = sw0;
                if (lane + 64 < NWS) sws[lane + 64] = sw1;
            }
#define CH16_ST(i) slot[lane + 64 * (part + 3 * i)] = v##i;
            CH16_REP(CH16_ST)
#undef CH16_ST
#undef CH16_LD
#undef CH16_REP
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) flag_set(F_READY + 16 * g + 4 * part, k + 1);
        }
        return;
    }
    // writer: entry j of the trail = (ROWB * position the unit ENDS at | rung of its band after it): lengths by difference
    for (uint32_t k = k0;; k++) {
        const uint32_t s = k & 1;
        uint32_t spin = 0;
        bool stop = false;
        while (flag_get(F_TRAILED + 4 * s) != k + 1) {
            const uint32_t st = flag_get(F_STOP);
            if ((st != 0xffffffffu && k >= st) || ++spin >= SPIN_MAX) { stop = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (stop) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint64_t U0 = *(volatile uint64_t *)(smem + F_U0 + 8 * s);
        const uint32_t n = flag_get(F_NUNITS + 4 * s), o_first = flag_get(F_O0 + 4 * s);
        const uint16_t *tr = (const uint16_t *)(smem + TR0 + s * TR_BYTES);
        const uint64_t wpos = slab0 + (uint64_t)k * CW;
        for (uint32_t j = lane; j < n; j += 64) {
            const uint32_t o0 = j ? tr[j - 1] / ROWB : o_first, o1 = tr[j] / ROWB;
            const uint64_t Uj = U0 + j;
            const uint32_t cj = (uint32_t)(Uj % B);
            if constexpr (CF) {             // the rung the unit is entered with: what its band's unit before it left
                const int32_t jb = (int32_t)j - (int32_t)B;
                const uint32_t rin = ((jb >= 0 ? (uint32_t)tr[jb] : wr[cj]) >> 1) & (NR - 1);
                if (a.g.ulen_sz == ULEN_UNIT) ((uint32_t *)a.idx.ulen)[Uj] = (o1 - o0) | rin << 16;
                else atomicAdd((uint32_t *)a.idx.ulen + Uj / B, (o1 - o0) | rin << (16 + 4 * cj));      // (8-bit, 1 / 3 / 4 bands: the block table, zeroed beforehand)
            } else if constexpr (UB == 3) ((uint8_t *)a.idx.ulen)[Uj] = (uint8_t)(o1 - o0);
            else ((uint16_t *)a.idx.ulen)[Uj] = (uint16_t)(o1 - o0);
            if (cj == 0 && (Uj / B) % NB == 0) {            // a segment starts here: position, and every band's rung as the block finds it
                const uint64_t seg = Uj / B / NB;
                a.idx.bitpos[seg] = wpos + o0;
                for (uint32_t cc = 0; cc < B; cc++) {       // band cc's unit before this one: B - cc units back
                    const int32_t jj = (int32_t)j - (int32_t)(B - cc);
                    a.idx.rung[seg * B + cc] = (uint8_t)(((jj >= 0 ? (uint32_t)tr[jj] : wr[cc]) >> 1) & (NR - 1));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // the rung every band has after this window: the last unit of each band in it
        if (lane < B) {
            const uint32_t cl = (uint32_t)((U0 + n - 1) % B);               // band of the window's last unit
            const uint32_t back = (cl + B - lane) % B;                      // band `lane` last came `back` units before it
            if (n > back) wr[lane] = tr[n - 1 - back] & ((NR - 1) << 1);
        }
        // (no release fence: the trail has been READ -- LDS operations of a wave are in order -- and the index stores may still be
        // on their way; a fence would hold the slot for a memory round trip per window)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) flag_set(F_READY + 16 * s + 12, k + 3);              // the trail slot is free for the window that takes it next
    }
}

// ---- the same for plain 32- and 64-bit streams (FTL / BASE) -----------------------------------------------------------
// Thirty-two or sixty-four rungs would make a row 64-128 bytes and a window a few units long.  But the rungs a stream
// visits keep to a narrow band (a band's rung moves with the local range of the data): the table is built for SIXTEEN
// CONSECUTIVE RUNGS [R0, R0 + 16), in the 16-bit layout (a row of sixteen 16-bit entries per position, rungs relative to
// R0), and an entry whose unit leaves the band carries the stop bit -- the walk then gives up on the table and the
// call falls back to the one-lane parser (a stream that ranges over more than sixteen rungs: rare, and no worse off than
// before).  R0 comes from the stream's first index segment, which walk_probe_kernel parses outright (one lane; the
// stream starts at rung 0, outside any band that fits real data): it leaves the walk's entry state behind that segment.
// Code lengths: a code at rung r takes r, r + 1 or r + 2 bits by its two low bits whatever r is, so the table workgroup
// keeps the EXTRA bits of 2, 4 and 8 codes (at most 16: a byte) per rung and position and adds the multiples of r.
constexpr uint32_t WIDE_NG = 4, WIDE_NWR = 3, WIDE_NT = 8;   // loader groups of three waves, writer waves, trail slots
constexpr uint32_t WIDE_THREADS = 64 * (1 + 3 * WIDE_NG + WIDE_NWR);
// NR_: rungs in the band.  8 or 16: a row of 16-bit entries (16 or 32 bytes a position), entry = (position the unit ends at) *
// ROWB | 2 * rung | stop -- one dependent read a unit.  14: BYTE entries, a row of sixteen bytes -- what the walk costs is the
// table bytes ONE CU can stream (15.8 GB/s measured, whatever the slab size or the number of loader waves: a CU keeps about
// 128 cache lines in flight), so half the bytes is half the time: bytes 0 .. 13 hold, per rung of the band, the EXTRA bits of
// the sixteen codes that start behind the switch (0 .. 32; rung 0: 1 or 17, the flag and the raw bits), bytes 14, 15 the
// switch (bits 0-3 its length, 4-9 the rung step, 10 the signal).  Two dependent reads a unit: the switch, then the extras
// of the rung it leads to; unit length = switch + 16 * rung + extras.
template <uint32_t UB, uint32_t NR_> struct chainW {
    static constexpr bool BYTE = NR_ == 14;
    static constexpr uint32_t NRUNG = 1u << UB, NR = NR_, ROWB = BYTE ? 16 : 2 * NR_, MAXC = NRUNG + 1, MAXU = UB + 2 + 16 * MAXC;
    // a window of the walk: as many positions as the 16-bit entries can address ((CW + MAXU) * ROWB < 65536), a multiple of 96 (the
    // loaders' 192 sixteen-byte pieces a turn) and of TCW, the positions ONE table workgroup tabulates (its LDS holds 32 bytes a position)
    static constexpr uint32_t CW = ROWB == 16 ? 2880 : (UB == 5 ? 1440 : 960), WIN_BYTES = CW * ROWB, WIN_U4 = WIN_BYTES / 16, TCW = 480;
    static constexpr uint32_t NP = (TCW + UB + 2 + 15 * MAXC + 2 + 31) & ~31u;     // positions a table workgroup looks at
    // One lane walks; what it waits for must never be one memory round trip per window.  A window's load takes about three
    // times as long as its walk: four loader groups (three waves each) keep four windows in flight for the two slots.  A
    // writer wave ends its turn waiting for its index stores (the compiler drains the store counter before the next spin
    // loop): about as long again -- so three writers take the windows in turn, and the trails wait for them in a ring of
    // eight slots, each with the walk's state at the window's start (units done, every band's rung).
    static constexpr uint32_t NG = WIDE_NG, NWR = WIDE_NWR, NT = WIDE_NT;
    static constexpr uint32_t TR_BYTES = ((CW / 2 + 8) * 2 + 15) & ~15u;
    static constexpr uint32_t TR0 = 2 * WIN_BYTES, TM0 = TR0 + NT * TR_BYTES /* [NT] x 32 bytes: units done (u64), rungs (u64), units, first position */,
                              RS0 = TM0 + NT * 32, META = RS0 + 64, LDS_BYTES = META + 128;
    static constexpr uint32_t F_READY = META /* [2][4] */, F_TRAILED = META + 32 /* [NT] */, F_TFREE = META + 64 /* [NT] */, F_WALKED = META + 96, F_STOP = META + 100;
    static_assert(((CW + MAXU) * ROWB < 65536 || (BYTE && CW + MAXU < 4096)) && WIN_U4 % 192 == 0 && CW % TCW == 0 && (UB == 5 || UB == 6) && (NR_ == 8 || NR_ == 14 || NR_ == 16), "window layout of the wide types");
};

template <uint32_t UB, uint32_t NRB>
__global__ void __launch_bounds__(256) walk_tableW_kernel(const DecArgs a0, uint4 *tab, uint64_t slab0, uint32_t nwin, uint64_t tab_pitch, const WalkState16 *states) {
    typedef chainW<UB, NRB> W;
    constexpr uint32_t NP = W::NP, CW = W::CW, TCW = W::TCW, NR = W::NR, MAXC = W::MAXC, ROWB = W::ROWB, NRUNG = W::NRUNG;
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    const uint64_t p0 = slab0 + (uint64_t)blockIdx.x * TCW;                // blockIdx.x: a piece of TCW positions; CW / TCW pieces a window
    const uint32_t ow = (uint32_t)(((uint64_t)blockIdx.x * TCW) % CW);     // the piece's place in its window: entries count positions from the window's start
    if (p0 >= a.in_bits + 2 * CW || states[blockIdx.y].bad) return;        // (uniform) far beyond the stream, or no walk will come
    const uint32_t R0 = states[blockIdx.y].pad;
    __shared__ uint32_t words[NP / 32 + 3];
    __shared__ uint8_t t1[NP], eA[NR][NP], eB[NR][NP];
    const uint32_t tid = threadIdx.x;
    const uint64_t q0 = a.in_bit0 + p0, w0 = q0 >> 5, endw = (a.in_bit0 + a.in_bits + 31) >> 5;
    const uint32_t sh = (uint32_t)q0 & 31;
    for (uint32_t i = tid; i < NP / 32 + 3; i += 256) words[i] = w0 + i < endw ? a.in32[w0 + i] : 0u;
    __syncthreads();
    auto bits = [&](uint32_t i) { const uint32_t b = sh + i, k = b >> 5; return __builtin_amdgcn_alignbit(words[k + 1], words[k], b & 31); };
    for (uint32_t i = tid; i < NP; i += 256) { const uint32_t x = bits(i); t1[i] = (uint8_t)((x & 1) + ((x & 3) == 3)); }   // a code's extra bits
    __syncthreads();
    uint32_t valid = NP - MAXC;
    for (uint32_t b = 0; b < NR; b++) {                                     // two codes
        const uint32_t r = R0 + b;
        if (r) for (uint32_t i = tid; i < valid; i += 256) { const uint32_t e = t1[i]; eA[b][i] = (uint8_t)(e + t1[i + r + e]); }
    }
    __syncthreads();
    uint8_t (*src)[NP] = eA, (*dst)[NP] = eB;
#pragma unroll 1
    for (uint32_t lvl = 1; lvl < 3; lvl++) {                                // four, eight codes: extras add, positions move by k * r + extras
        valid -= MAXC << lvl;
        for (uint32_t b = 0; b < NR; b++) {
            const uint32_t r = R0 + b, kr = r << lvl;
            if (r) for (uint32_t i = tid; i < valid; i += 256) { const uint32_t e = src[b][i]; dst[b][i] = (uint8_t)(e + src[b][i + kr + e]); }
        }
        __syncthreads();
        uint8_t (*t)[NP] = src; src = dst; dst = t;
    }
    // src = the extras of eight codes; sixteen = eight + eight, formed here
    uint4 *out = tab + ((uint64_t)blockIdx.y * tab_pitch + (uint64_t)blockIdx.x * (ROWB / 16 * TCW));      // (rows are consecutive: ROWB / 16 sixteen-byte pieces a position)
    for (uint32_t o = tid; o < TCW; o += 256) {
        const uint32_t x = bits(o);
        uint32_t delta = 0; bool sig = false;
        const uint32_t cs = walk_switch<UB>(x, delta, sig);                 // from rung 0: the step itself
        const uint32_t len0 = cs + (((bits(o + cs)) & 1) ? 17 : 1);         // rung 0: one flag, then 16 raw bits
        if constexpr (W::BYTE) {
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t b = 0; b < NR; b++) {                             // the rung the switch LEADS to: the extras of the sixteen codes behind it
                const uint32_t r = R0 + b;
                uint32_t ex = len0 - cs;
                if (r) { const uint32_t e8 = src[b][o + cs]; ex = e8 + src[b][o + cs + 8 * r + e8]; }
                w[b >> 2] |= ex << (8 * (b & 3));
            }
            w[3] |= (cs | (delta << 4) | ((sig ? 1u : 0u) << 10)) << 16;
            out[o] = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            uint32_t e[NR];
#pragma unroll
            for (uint32_t bin = 0; bin < NR; bin++) {
                const uint32_t r = (R0 + bin + delta) & (NRUNG - 1), rb = r - R0;
                const bool out_of_band = rb >= NR;
                const uint32_t bb = out_of_band ? 0u : rb;
                uint32_t u = len0;
                if (r && !out_of_band) { const uint32_t n8 = 8 * r + src[bb][o + cs]; u = cs + n8 + 8 * r + src[bb][o + cs + n8]; }
                e[bin] = ((ow + o + (out_of_band ? 1u : u)) * ROWB) | (bb << 1) | ((sig || out_of_band) ? 1u : 0u);
            }
            if (NR == 16) {
                out[2 * o] = make_uint4(e[0] | e[1] << 16, e[2] | e[3] << 16, e[4] | e[5] << 16, e[6] | e[7] << 16);
                out[2 * o + 1] = make_uint4(e[8 % NR] | e[9 % NR] << 16, e[10 % NR] | e[11 % NR] << 16, e[12 % NR] | e[13 % NR] << 16, e[14 % NR] | e[15 % NR] << 16);
            } else out[o] = make_uint4(e[0] | e[1] << 16, e[2] | e[3] << 16, e[4] | e[5] << 16, e[6] | e[7] << 16);
        }
    }
}

// The walk: walk_chain16_kernel's organisation (a lane chases, six waves load windows, one writes the index) with the
// wide types' window size; rungs are relative to the band's R0 on the way, absolute in the index.
template <uint32_t UB, uint32_t NRB>
__global__ void __launch_bounds__(WIDE_THREADS) walk_chainW_kernel(const DecArgs a0, const uint4 *tab, uint64_t slab0, uint32_t nwin, uint64_t tab_pitch, WalkState16 *states) {
    typedef chainW<UB, NRB> W;
    constexpr uint32_t CW = W::CW, ROWB = W::ROWB, NR = W::NR, WIN_BYTES = W::WIN_BYTES, WIN_U4 = W::WIN_U4, TR0 = W::TR0, TR_BYTES = W::TR_BYTES, TM0 = W::TM0, RS0 = W::RS0,
                       META = W::META, F_READY = W::F_READY, F_TRAILED = W::F_TRAILED, F_TFREE = W::F_TFREE, F_WALKED = W::F_WALKED, F_STOP = W::F_STOP,
                       NG = W::NG, NWR = W::NWR, NT = W::NT;
    using chain::flag_get; using chain::flag_set; using chain::SPIN_MAX;
    const DecArgs a = dec_for_tile(a0, blockIdx.x);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t B = a.g.bands, NB = a.g.seg_blocks;
    const uint64_t nunits = a.g.nblocks * B;
    WalkState16 *S = states + blockIdx.x;
    const uint64_t P0 = S->P, U_in = S->unit, R_in = S->rungs;
    const uint32_t R0 = S->pad;
    const uint64_t slab_end = slab0 + (uint64_t)nwin * CW;
    if (S->bad || P0 < slab0 || P0 >= slab_end || P0 >= a.in_bits || U_in >= nunits) return;  // (uniform) nothing of this tile in this slab
    const uint32_t k0 = (uint32_t)((P0 - slab0) / CW);                      // the window the walk starts in
    constexpr uint32_t RSH = W::BYTE ? 0 : 1;                               // rs[]: the bands' rungs (relative to R0), times two for the entry tables
    volatile uint32_t *rs = (volatile uint32_t *)(smem + RS0);
    if (tid < 32) ((uint32_t *)(smem + META))[tid] = tid == (F_STOP - META) / 4 ? 0xffffffffu : 0u;
    if (tid < 16) rs[tid] = (uint32_t)((R_in >> (4 * tid)) & 15u) << RSH;
    __syncthreads();
    const uint4 *wt = tab + (uint64_t)blockIdx.x * tab_pitch;

    if (wave == 0) {
        if (lane) return;
        uint32_t bad = 0, k = k0, o = (uint32_t)((P0 - slab0) % CW);
        uint64_t U = U_in, Pn = P0;
        uint32_t c = (uint32_t)(U % B);
        bool stuck = false;
        while (true) {
            const uint32_t s = k & 1, ts = k % NT;
            uint32_t spin = 0;
            while (true) {      // the window's three parts are in LDS, and the trail slot has been read out (it held window k - NT)
                const bool here = flag_get(F_READY + 16 * s) == k + 1 && flag_get(F_READY + 16 * s + 4) == k + 1 && flag_get(F_READY + 16 * s + 8) == k + 1;
                const bool slot = k - k0 < NT || flag_get(F_TFREE + 4 * ts) == k - NT + 1;
                if (here && slot) break;
                if (++spin >= SPIN_MAX) { stuck = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (stuck) break;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            uint64_t Rw = 0;                                                // every band's rung as the window finds it
            for (uint32_t i = 0; i < B; i++) Rw |= (uint64_t)((rs[i] >> RSH) & 15u) << (4 * i);
            volatile uint64_t *tm = (volatile uint64_t *)(smem + TM0 + 32 * ts);
            tm[0] = U; tm[1] = Rw;
            uint32_t A = o * ROWB + (W::BYTE ? 0u : rs[c]), n = 0;
            const uint64_t left64 = nunits - U;
            uint32_t left = left64 > 0xffffffffull ? 0xffffffffu : (uint32_t)left64;
            typedef const __attribute__((address_space(3))) uint16_t *LdsHalf;
            typedef __attribute__((address_space(3))) uint16_t *LdsHalfW;
            typedef __attribute__((address_space(3))) uint32_t *LdsWordW;
            const uint32_t wbase = s * WIN_BYTES;
            LdsHalfW trw = (LdsHalfW)(uintptr_t)(TR0 + ts * TR_BYTES);
            LdsWordW rsw = (LdsWordW)(uintptr_t)RS0;
            constexpr uint32_t M = 0xffffu & ~(ROWB - 1), RM = (NR - 1) << 1;
            // a unit per turn, until one starts beyond the window or an entry carries the stop bit (a unit that leaves the band of
            // rungs, or the signal code): ONE dependent LDS read a unit (byte tables: two)
            if constexpr (W::BYTE) {
                typedef const __attribute__((address_space(3))) uint8_t *LdsByte;
                uint32_t rb = rsw[c];                                       // the band's rung, relative to R0
                while (A < CW * ROWB && left && !(bad & 1u)) {              // (A: the row of the unit's first bit)
                    const uint32_t sw = *(LdsHalf)(uintptr_t)(wbase + A + 14);
                    const uint32_t rabs = (R0 + rb + ((sw >> 4) & 63u)) & (W::NRUNG - 1), rnew = rabs - R0;
                    const bool stop = ((sw >> 10) & 1u) || rnew >= NR;
                    const uint32_t ex = *(LdsByte)(uintptr_t)(wbase + A + (stop ? 0u : rnew));
                    const uint32_t oe = (A >> 4) + (sw & 15u) + 16 * rabs + ex;         // where the unit ends: the next one's first bit
                    trw[n++] = (uint16_t)((oe << 4) | (rnew & 15u)); bad |= stop ? 1u : 0u;
                    rsw[c] = rnew;
                    A = oe << 4;
                    c = c + 1 == B ? 0 : c + 1;
                    rb = B == 1 ? rnew : rsw[c];
                    left--;
                }
            } else if (B == 1) {
                while (A < CW * ROWB && left && !(bad & 1u)) {
                    const uint32_t e = *(LdsHalf)(uintptr_t)(wbase + A);
                    trw[n++] = (uint16_t)e; bad |= e;
                    A = e & (M | RM);
                    left--;
                }
                rsw[0] = A & RM;
            } else {
                uint32_t cn = c + 1 == B ? 0 : c + 1;
                uint32_t rn = rsw[cn];
                while (A < CW * ROWB && left && !(bad & 1u)) {
                    const uint32_t cn2 = cn + 1 == B ? 0 : cn + 1;
                    const uint32_t e = *(LdsHalf)(uintptr_t)(wbase + A);
                    const uint32_t r2 = B > 2 ? rsw[cn2] : 0u;
                    trw[n++] = (uint16_t)e; bad |= e;
                    rsw[c] = e & RM;
                    A = (e & M) | rn;
                    if (B == 2) rn = e & RM; else rn = r2;                  // (two bands: this band comes again after the next unit)
                    c = cn; cn = cn2; left--;
                }
            }
            ((volatile uint32_t *)tm)[4] = n; ((volatile uint32_t *)tm)[5] = o;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            flag_set(F_TRAILED + 4 * ts, k + 1);
            U += n;
            const uint32_t oe = A / ROWB;
            Pn = slab0 + (uint64_t)k * CW + oe;
            k++;
            flag_set(F_WALKED, k - k0);
            if (U >= nunits || (bad & 1u)) break;                           // the units ran out, or the table does not carry this stream
            o = oe - CW;                                                    // (the walk left the window: oe >= CW)
            if (k >= nwin || Pn >= a.in_bits) break;                        // the slab ends here, or the stream does (a damaged one)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        flag_set(F_STOP, k);
        uint64_t Rn = 0;
        for (uint32_t i = 0; i < B; i++) Rn |= (uint64_t)((rs[i] >> RSH) & 15u) << (4 * i);
        S->P = stuck ? ~0ull : Pn; S->unit = U; S->rungs = Rn; S->bad = (bad & 1u) | (stuck ? 1u : 0u);
        if ((bad & 1u) || stuck || (U < nunits && Pn >= a.in_bits)) atomicOr(a.status, 1u);      // (cut short: the one-lane parser takes it)
        return;
    }
    if (wave <= 3 * NG) {
        const uint32_t g = (wave - 1) / 3, part = (wave - 1) % 3;
        constexpr uint32_t NV = WIN_U4 / 192;                               // sixteen-byte pieces a lane moves per window
        for (uint32_t k = k0 + ((g + NG - k0 % NG) % NG); k < nwin; k += NG) {
            const uint4 *src = wt + (uint64_t)k * WIN_U4;
            uint4 v[NV];
#pragma unroll
            for (uint32_t i = 0; i < NV; i++) v[i] = src[lane + 64 * (part + 3 * i)];
            uint32_t spin = 0;
            bool stop = false;
            while (true) {                                                  // the slot is free when the window two back has been walked
                if (flag_get(F_STOP) != 0xffffffffu) { stop = true; break; }
                if (flag_get(F_WALKED) + 2 > k - k0) break;
                if (++spin >= SPIN_MAX) { stop = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (stop) break;
            uint4 *slot = (uint4 *)(smem + (k & 1) * WIN_BYTES);
#pragma unroll
            for (uint32_t i = 0; i < NV; i++) slot[lane + 64 * (part + 3 * i)] = v[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) flag_set(F_READY + 16 * (k & 1) + 4 * part, k + 1);
        }
        return;
    }
    // writers: entry j of a window's trail = (position the unit ENDS at | rung of its band after it): lengths by difference
    const uint32_t wtr = wave - 1 - 3 * NG;
    for (uint32_t k = k0 + wtr;; k += NWR) {
        const uint32_t ts = k % NT;
        uint32_t spin = 0;
        bool stop = false;
        while (flag_get(F_TRAILED + 4 * ts) != k + 1) {
            const uint32_t st = flag_get(F_STOP);
            if ((st != 0xffffffffu && k >= st) || ++spin >= SPIN_MAX) { stop = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (stop) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const volatile uint64_t *tm = (const volatile uint64_t *)(smem + TM0 + 32 * ts);
        const uint64_t U0 = tm[0], Rw = tm[1];
        const uint32_t n = ((const volatile uint32_t *)tm)[4], o_first = ((const volatile uint32_t *)tm)[5];
        const uint16_t *tr = (const uint16_t *)(smem + TR0 + ts * TR_BYTES);
        const uint64_t wpos = slab0 + (uint64_t)k * CW;
        uint16_t *ul = (uint16_t *)a.idx.ulen + U0;
        for (uint32_t j = lane; j < n; j += 64) {
            constexpr uint32_t PSH = W::BYTE ? 4 : 0, PDIV = W::BYTE ? 1 : ROWB, RMASK = W::BYTE ? 15u : NR - 1;      // a trail entry: position << 4 | rung (byte tables), position * ROWB | rung << 1
            const uint32_t o0 = j ? (tr[j - 1] >> PSH) / PDIV : o_first, o1 = (tr[j] >> PSH) / PDIV;
            ul[j] = (uint16_t)(o1 - o0);
            const uint64_t Uj = U0 + j;
            if (Uj % B == 0 && (Uj / B) % NB == 0) {        // a segment starts here: position, and every band's rung as the block finds it
                const uint64_t seg = Uj / B / NB;
                a.idx.bitpos[seg] = wpos + o0;
                for (uint32_t cc = 0; cc < B; cc++) {       // band cc's unit before this one: B - cc units back, or the window's entering state
                    const int32_t jj = (int32_t)j - (int32_t)(B - cc);
                    const uint32_t rb = jj >= 0 ? ((uint32_t)tr[jj] >> RSH) & RMASK : (uint32_t)(Rw >> (4 * cc)) & 15u;
                    a.idx.rung[seg * B + cc] = (uint8_t)(R0 + rb);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the trail has been READ: LDS operations of a wave are in order; the index stores may still be on their way)
        if (lane == 0) flag_set(F_TFREE + 4 * ts, k + 1);
    }
}

// Slabs of the streams are tabulated by the whole chip, then walked by a workgroup per tile, slab after slab; the
// table of the next slab is built (on a stream of its own, in the other half of the memory) while this one is walked.
// tab: [walk state per tile][windows of table rows per tile] x 2; max_bits: the longest stream of the call.
template <typename TABLE, typename CHAIN>
static void walk_in_slabs(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits, uint32_t cw, uint32_t win_u4, size_t state_size,
                          TABLE &&launch_table, CHAIN &&launch_chain) {
    const uint32_t nt = a.ntiles;
    const size_t state_bytes = ((size_t)nt * state_size + 255) & ~(size_t)255;
    uint8_t *base = (uint8_t *)tab;
    const uint64_t need = (max_bits + cw - 1) / cw;                         // windows of the longest stream
    const uint64_t cap = (tab_bytes - state_bytes) / ((uint64_t)win_u4 * 16 * nt);  // windows per tile the memory holds
    // one round when the streams are short (nothing to overlap, and a stream costs more to create than it saves)
    if (need <= cap && need * cw <= (8u << 20)) {
        const uint64_t pitch = need * win_u4;
        uint4 *rows = (uint4 *)(base + state_bytes);
        { ProfScope ps("dec_index_table", st); launch_table(st, rows, (uint64_t)0, (uint32_t)need, pitch); }
        ProfScope ps("dec_index_serial", st);
        launch_chain(st, rows, (uint64_t)0, (uint32_t)need, pitch, base, 1u);
        return;
    }
    // rounds of at most half the memory, and at least four of them
    uint64_t nwin = cap / 2;
    if (nwin > (need + 3) / 4) nwin = (need + 3) / 4;
    if (nwin < 16) nwin = 16;                                               // (walk_table_min_bytes holds 2 x 16)
    if (nwin > 0x7fffffffu / win_u4) nwin = 0x7fffffffu / win_u4;
    const uint64_t pitch = nwin * win_u4;                                   // in rows of sixteen bytes
    uint4 *rows[2] = {(uint4 *)(base + state_bytes), (uint4 *)(base + state_bytes) + pitch * nt};
    hipStream_t aux = nullptr;
    hipEvent_t ev_tab[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr}, ev_start = nullptr;
    bool ok = hipStreamCreateWithFlags(&aux, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < 2 && ok; i++)
        ok = hipEventCreateWithFlags(&ev_tab[i], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ev_done[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&ev_start, hipEventDisableTiming) == hipSuccess;
    if (ok) { (void)hipEventRecord(ev_start, st); (void)hipStreamWaitEvent(aux, ev_start, 0); }
    hipStream_t tst = ok ? aux : st;                                        // (no second stream: everything in order on the caller's)
    uint32_t first = 1, j = 0;
    for (uint64_t s0 = 0; s0 < max_bits; s0 += nwin * cw, first = 0, j++) {
        const int h = j & 1;
        if (ok && j >= 2) (void)hipStreamWaitEvent(aux, ev_done[h], 0);     // the walk of two rounds ago has left this half
        { ProfScope ps("dec_index_table", tst); launch_table(tst, rows[h], s0, (uint32_t)nwin, pitch); }
        if (ok) { (void)hipEventRecord(ev_tab[h], aux); (void)hipStreamWaitEvent(st, ev_tab[h], 0); }
        { ProfScope ps("dec_index_serial", st); launch_chain(st, rows[h], s0, (uint32_t)nwin, pitch, base, first); }
        if (ok) (void)hipEventRecord(ev_done[h], st);
    }
    // (destroying a stream or an event with work pending is deferred by the runtime until that work is done)
    for (int i = 0; i < 2; i++) { if (ev_tab[i]) (void)hipEventDestroy(ev_tab[i]); if (ev_done[i]) (void)hipEventDestroy(ev_done[i]); }
    if (ev_start) (void)hipEventDestroy(ev_start);
    if (aux) (void)hipStreamDestroy(aux);
}
template <uint32_t U, uint32_t N> struct WideTag { static constexpr uint32_t UB_ = U, NR_ = N; };
bool walk_chain_lds_ok() {
    static const bool lds_ok = [] {
        bool ok = true;
        ok = ok && hipFuncSetAttribute((const void *)walk_chain_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, chain::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_chain_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, chain::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_chain_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, chain::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_chain_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, chain::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_chainN_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, chainN<4>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_chainN_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, chainN<4>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_chainN_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, chainN<3>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_chainN_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, chainN<3>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_chainW_kernel<5, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, chainW<5, 16>::LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void *)walk_chainW_kernel<6, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, chainW<6, 16>::LDS_BYTES) == hipSuccess;
        return ok;
    }();
    return lds_ok;
}
// positions of a chain window (16-bit data: 32 table bytes a stream bit; 32/64-bit: the table of sixteen rungs, windows of 1440 / 960 positions)
uint32_t walk_cw(uint32_t tsz) { return tsz == 2 ? chainN<4>::CW : tsz == 4 ? chainW<5, 16>::CW : tsz == 8 ? chainW<6, 16>::CW : chain::CW; }
uint32_t walk_win_bytes(uint32_t tsz) { return tsz == 1 ? chain::WIN_BYTES : walk_cw(tsz) * 32; }

void walk_chain_8bit(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits) {
    const uint32_t nt = a.ntiles;
    walk_in_slabs(a, st, tab, tab_bytes, max_bits, chain::CW, chain::ROWS, sizeof(WalkState),
        [&](hipStream_t s, uint4 *rows, uint64_t s0, uint32_t nwin, uint64_t pitch) {
            hipLaunchKernelGGL(walk_table_kernel, dim3(nwin, nt), dim3(256), 0, s, a, rows, s0, nwin, pitch); },
        [&](hipStream_t s, const uint4 *rows, uint64_t s0, uint32_t nwin, uint64_t pitch, uint8_t *states, uint32_t first) {
            using namespace chain;
            WalkState *ws = (WalkState *)states;
            if (a.g.bands == 1) hipLaunchKernelGGL(walk_chain_kernel<1>, dim3(nt), dim3(512), LDS_BYTES, s, a, rows, s0, nwin, pitch, ws, first);
            else if (a.g.bands == 2) hipLaunchKernelGGL(walk_chain_kernel<2>, dim3(nt), dim3(512), LDS_BYTES, s, a, rows, s0, nwin, pitch, ws, first);
            else if (a.g.bands == 3) hipLaunchKernelGGL(walk_chain_kernel<3>, dim3(nt), dim3(512), LDS_BYTES, s, a, rows, s0, nwin, pitch, ws, first);
            else hipLaunchKernelGGL(walk_chain_kernel<4>, dim3(nt), dim3(512), LDS_BYTES, s, a, rows, s0, nwin, pitch, ws, first); });
}
// 16-bit streams, and (UB = 3) the 8-bit streams walk_chain_8bit's hand-ordered loops do not take; common-factor streams of several bands
template <uint32_t UB>
static void walk_chain_n(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits) {
    typedef chainN<UB> W;
    const uint32_t nt = a.ntiles;
    const bool cf = a.g.mode == CM_BEST;
    walk_in_slabs(a, st, tab, tab_bytes, max_bits, W::CW, W::WIN_U4, sizeof(WalkState16),
        [&](hipStream_t s, uint4 *rows, uint64_t s0, uint32_t nwin, uint64_t pitch) {
            hipLaunchKernelGGL(walk_tableN_kernel<UB>, dim3(nwin, nt), dim3(256), 0, s, a, rows, s0, nwin, pitch); },
        [&](hipStream_t s, const uint4 *rows, uint64_t s0, uint32_t nwin, uint64_t pitch, uint8_t *states, uint32_t first) {
            if (cf) hipLaunchKernelGGL((walk_chainN_kernel<UB, true>), dim3(nt), dim3(512), W::LDS_BYTES, s, a, rows, s0, nwin, pitch, (WalkState16 *)states, first);
            else hipLaunchKernelGGL((walk_chainN_kernel<UB, false>), dim3(nt), dim3(512), W::LDS_BYTES, s, a, rows, s0, nwin, pitch, (WalkState16 *)states, first); });
}
void walk_chain_16bit(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits) { walk_chain_n<4>(a, st, tab, tab_bytes, max_bits); }
void walk_chain_8bit_any(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits) { walk_chain_n<3>(a, st, tab, tab_bytes, max_bits); }
// 32/64-bit FTL/BASE: the first segment parsed outright (band of rungs, entry state), then table + chain
void walk_chain_wide(const DecArgs &a, hipStream_t st, void *tab, size_t tab_bytes, uint64_t max_bits) {
    const uint32_t nt = a.ntiles;
    WalkState16 *states = (WalkState16 *)tab;
    {
        ProfScope ps("dec_index_serial", st);
        if (a.g.tsz == 4) hipLaunchKernelGGL((walk_probe_kernel<uint32_t, CM_FTL>), dim3(nt), dim3(64), 0, st, a, states, 16u, 0u);
        else hipLaunchKernelGGL((walk_probe_kernel<uint64_t, CM_FTL>), dim3(nt), dim3(64), 0, st, a, states, 16u, 0u);
    }
    auto run = [&](auto tag) {
        constexpr uint32_t UB = decltype(tag)::UB_, NRB = decltype(tag)::NR_;
        typedef chainW<UB, NRB> W;
        walk_in_slabs(a, st, tab, tab_bytes, max_bits, W::CW, W::WIN_U4, sizeof(WalkState16),
            [&](hipStream_t s, uint4 *rows, uint64_t s0, uint32_t nwin, uint64_t pitch) {
                hipLaunchKernelGGL((walk_tableW_kernel<UB, NRB>), dim3(nwin * (W::CW / W::TCW), nt), dim3(256), 0, s, a, rows, s0, nwin, pitch, (const WalkState16 *)states); },
            [&](hipStream_t s, const uint4 *rows, uint64_t s0, uint32_t nwin, uint64_t pitch, uint8_t *sts, uint32_t) {
                hipLaunchKernelGGL((walk_chainW_kernel<UB, NRB>), dim3(nt), dim3(WIDE_THREADS), W::LDS_BYTES, s, a, rows, s0, nwin, pitch, (WalkState16 *)sts); });
    };
    // (a band of eight rungs and a byte-entry table of fourteen were built and measured: DESIGN.md section 4, "Tried and measured")
    if (a.g.tsz == 4) run(WideTag<5, 16>()); else run(WideTag<6, 16>());
}

}  // namespace qb3dev
