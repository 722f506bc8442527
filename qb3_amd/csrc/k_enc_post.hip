// qb3_amd/csrc/k_enc_post.hip -- after the chunks are coded: offsets (scan), concatenation, seams, header, index chunks
#include "qb3_kernels.h"

namespace qb3dev {

// Exclusive scan of the chunk bit counts in ONE launch: a workgroup per SCAN_GROUP chunks (4 per thread) leaves its chunks'
// offsets inside the group and the group's sum; the workgroup that finishes LAST (a counter behind the group sums, zeroed
// by the coding kernel of the same call) scans the group sums in place -- entry [ngroups] gets the total.  64-bit offsets:
// a 16384^2 x 3 stream exceeds 2^32 bits.  The consumers add group offset and chunk offset (chunk_start).
__global__ void __launch_bounds__(SCAN_GROUP / 4) enc_scan_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    __shared__ uint32_t wsum[16];
    __shared__ uint64_t wsum64[16];
    __shared__ uint64_t carry;
    __shared__ uint32_t is_last;
    const uint32_t tid = threadIdx.x, i0 = blockIdx.x * SCAN_GROUP + 4 * tid;
    const uint32_t ngroups = (a.nchunks + SCAN_GROUP - 1) / SCAN_GROUP;
    uint32_t v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = (i0 + k < a.nchunks) ? a.chunk_bits[i0 + k] : 0; sum += v[k]; }
    uint32_t total;
    uint64_t off = block_exscan(sum, wsum, &total);
#pragma unroll
    for (int k = 0; k < 4; k++) if (i0 + k < a.nchunks) { a.chunk_off[i0 + k] = off; off += v[k]; }
    if (tid == 0) {
        a.group_sum[blockIdx.x] = total;
        __threadfence();                                    // the sum is out before the count says so (agent scope: the XCDs' L2s)
        is_last = atomicAdd((uint32_t *)&a.group_sum[ngroups + 1], 1u) == gridDim.x - 1;
        carry = 0;
    }
    __syncthreads();
    if (!is_last) return;                                   // (uniform)
    __threadfence();
    for (uint32_t base = 0; base < ngroups; base += blockDim.x) {
        const uint32_t i = base + tid;
        const uint64_t g = i < ngroups ? __hip_atomic_load(&a.group_sum[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        const uint64_t ex = block_exscan_v<uint64_t>(g, wsum64);
        const uint64_t c0 = carry;
        if (i < ngroups) a.group_sum[i] = c0 + ex;
        __syncthreads();
        if (tid == blockDim.x - 1) carry = c0 + ex + g;
        __syncthreads();
    }
    if (tid == 0) { a.group_sum[ngroups] = carry; a.res->zero_run = 0; a.res->ff_pairs = 0; a.res->zero_dwords = 0; }      // (enc_concat_kernel counts into them)
}

// start of chunk k in the stream, in bits (k == nchunks: the stream length)
__device__ __forceinline__ uint64_t chunk_start(const EncArgs &a, uint32_t k) {
    if (k >= a.nchunks) return a.group_sum[(a.nchunks + SCAN_GROUP - 1) / SCAN_GROUP];
    return a.group_sum[k / SCAN_GROUP] + a.chunk_off[k];
}

// at how many bytes of cur do four zero bytes in a row start (nxt: the stream dword behind cur)
__device__ __forceinline__ uint32_t zero_runs_in(uint32_t cur, uint32_t nxt) {
    return (uint32_t)(cur == 0) + (uint32_t)(__builtin_amdgcn_alignbit(nxt, cur, 8) == 0) +
           (uint32_t)(__builtin_amdgcn_alignbit(nxt, cur, 16) == 0) + (uint32_t)(__builtin_amdgcn_alignbit(nxt, cur, 24) == 0);
}
// how many bytes of v are 0xff and followed, inside v, by another 0xff
__device__ __forceinline__ uint32_t ff_pairs_in(uint32_t v) {
    const uint32_t t = v & (v >> 8) & 0x00ffffffu;                     // a byte of t is 0xff where the pair stands
    return (uint32_t)__popc(((t & 0x7f7f7fu) + 0x010101u) & t & 0x808080u);
}

// Concatenate: one WAVE per chunk reads the chunk's slot, funnel-shifts it to its bit position and stores the
// dwords that lie wholly inside the chunk; the first and last shifted dword go to the seam table.
// With zrun_probe (the RLE0 modes) the wave also counts, among the dwords it moves, the positions at which four zero bytes
// in a row start and the 0xff bytes followed by another in the same dword -- RLE0 (reference QB3encode.cpp:536-565) can only
// shorten a stream whose zero runs outweigh its pairs of 0xff (rle0_may_win, qb3_dev.h), and counting here spares the byte
// pass over the finished stream whenever the counts decide.  Zero runs are counted EXACTLY: here the positions whose four bytes lie
// in dwords this chunk owns alone (a dword shared with a neighbouring chunk counts as non-zero), in finish_seam -- on the finished
// stream -- the positions that touch a chunk boundary.  (Round 3 tested a shared dword as each chunk sees it, the neighbour's bits
// zero: on the side of running the pass, and config 2's QB3M_BEST stream, which has no zero run at all, came out with 5 401 -- one
// chunk boundary in twelve -- and paid 0.2 ms for the size pass every call.)  Pairs of 0xff are counted too rarely (never across
// dwords, never in bits another chunk owns): on the side of running the pass.
__global__ void __launch_bounds__(256) enc_concat_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    const uint32_t chunk = a.chunk0 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (chunk >= a.chunk_end) return;
    const uint64_t G = (uint64_t)a.out_bit0 + chunk_start(a, chunk);
    const uint32_t total = a.chunk_bits[chunk];
    const uint32_t phase = (uint32_t)(G & 31), nsrc = (total + 31) >> 5;
    const uint32_t nd = (phase + total + 31) >> 5, tailbits = (phase + total) & 31;
    const uint32_t *slot = a.scratch + (uint64_t)chunk * a.slot_dw;    // 16-byte aligned (slot_dw is a multiple of 4)
    const uint4 *slot4 = (const uint4 *)slot;
    uint32_t *gout = a.out32 + (G >> 5);
    // a lane moves four dwords per step (one 16-byte load, the next one already in flight): memory-level parallelism
    // is what this copy needs.  Output dword d = source dwords d-1, d funnel-shifted by the chunk's bit phase.
    const uint32_t ng = (nd + 3) >> 2, sh = (32 - phase) & 31;
    constexpr int NQ = 4;                                               // 16-byte loads in flight per lane
    uint32_t zrun = 0, ffp = 0, zdw = 0;
    for (uint32_t gb = 0; gb < ng; gb += 64 * NQ) {
        uint4 cur[NQ];
        uint32_t before[NQ];                                            // lane 0: the dword before its group
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const uint32_t g = gb + 64 * q + lane;
            cur[q] = make_uint4(0, 0, 0, 0); before[q] = 0;
            if (g < ng && 4 * g < nsrc) cur[q] = slot4[g];
            if (lane == 0 && g && g < ng && 4 * g - 1 < nsrc) before[q] = slot[4 * g - 1];
        }
        uint32_t first[NQ], last[NQ];                                   // (zrun_probe) a group's first and last output dword
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const uint32_t g = gb + 64 * q + lane, d = 4 * g;
            uint32_t s[4] = { cur[q].x, cur[q].y, cur[q].z, cur[q].w };
#pragma unroll
            for (int k = 0; k < 4; k++) if (d + k >= nsrc) s[k] = 0;    // the slot is only defined up to nsrc
            uint32_t prv = __shfl_up(s[3], 1, 64);
            if (lane == 0) prv = before[q];
            first[q] = last[q] = 0xffffffffu;
            if (g >= ng) continue;
            uint32_t v[4];
            if (phase) {
                v[0] = __builtin_amdgcn_alignbit(s[0], prv, sh);
#pragma unroll
                for (int k = 1; k < 4; k++) v[k] = __builtin_amdgcn_alignbit(s[k], s[k - 1], sh);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = s[k];
            }
            if (d > 0 && d + 4 < nd) {                                  // no seam dword in the group
                u32x4_a4 o = { v[0], v[1], v[2], v[3] };
                *(u32x4_a4 *)(gout + d) = o;
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t dd = d + k;
                    if (dd < nd) {
                        const bool shared = (dd == 0 && phase) || (dd == nd - 1 && tailbits);
                        if (!shared) gout[dd] = v[k];
                        if (dd == 0) a.seams[2 * chunk] = v[k];
                        if (dd == nd - 1) a.seams[2 * chunk + 1] = v[k];
                    }
                }
            }
            if (a.zrun_probe) {         // (wave uniform) runs that start in dwords 0 .. 2 of the group; dwords behind the chunk count as non-zero
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (d + k >= nd) v[k] = 0xffffffffu;
                    else {
                        ffp += ff_pairs_in(v[k]);
                        const bool shared = (d + k == 0 && phase) || (d + k == nd - 1 && tailbits);
                        zdw += (uint32_t)(v[k] == 0 && !shared);            // (all-zero dwords that are this chunk's alone: 4 KB of zeros hold hundreds)
                        if (shared) v[k] = 0xffffffffu;                     // (zero runs that touch a dword two chunks share: finish_seam counts them, on the finished dword)
                    }
                }
                zrun += zero_runs_in(v[0], v[1]) + zero_runs_in(v[1], v[2]) + zero_runs_in(v[2], v[3]);
                first[q] = v[0]; last[q] = v[3];
            }
        }
        if (a.zrun_probe) {             // ... and in a group's last dword: the dword behind it is the next lane's first, lane 63's the next batch's
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                uint32_t nxt = (uint32_t)__shfl_down((int)first[q], 1, 64);
                const uint32_t wrap = q + 1 < NQ ? (uint32_t)__shfl((int)first[q + 1 < NQ ? q + 1 : q], 0, 64) : 0xffffffffu;
                if (lane == 63) {
                    nxt = wrap;
                    if (q + 1 == NQ) {                                  // the first dword of the next round of the loop
                        const uint32_t dn = 4 * (gb + 64 * NQ);
                        if (dn < nd) {
                            const uint32_t s0 = dn < nsrc ? slot[dn] : 0u, sp = dn - 1 < nsrc ? slot[dn - 1] : 0u;
                            nxt = phase ? __builtin_amdgcn_alignbit(s0, sp, sh) : s0;
                            if (dn == nd - 1 && tailbits) nxt = 0xffffffffu;    // (shared with the next chunk: finish_seam's)
                        }
                    }
                }
                zrun += zero_runs_in(last[q], nxt);
            }
        }
    }
    if (a.zrun_probe) {
#pragma unroll
        for (int o = 32; o; o >>= 1) { zrun += (uint32_t)__shfl_xor((int)zrun, o, 64); ffp += (uint32_t)__shfl_xor((int)ffp, o, 64); zdw += (uint32_t)__shfl_xor((int)zdw, o, 64); }
        if (lane == 0 && zdw) atomicAdd((unsigned long long *)&a.res->zero_dwords, (unsigned long long)zdw);
        if (lane == 0 && zrun) atomicAdd((unsigned long long *)&a.res->zero_run, (unsigned long long)zrun);
        if (lane == 0 && ffp) atomicAdd((unsigned long long *)&a.res->ff_pairs, (unsigned long long)ffp);
    }
}

// What is left to do once the chunks stand in the stream, in ONE launch:
// * a thread per chunk boundary: a dword that holds the end of one chunk and the start of the next is the OR of their edge
//   dwords; the thread of the FIRST boundary inside a dword assembles it (the stream's first dword, when the header ends
//   inside it, is written byte by byte from the stream's first byte on: the bytes in front belong to whoever writes the header);
// * the stream length;
// * the chunk-relative index positions become stream positions, and the thread that does that for a segment also writes
//   the segment's entry of the restart table -- every ix_spe-th segment entry of the index, packed little endian, in
//   chunks of ix_per_chunk entries, each a lower-case (ignorable) "ix" chunk followed by a 4-byte "zz" pad chunk, then "DT".
//   Chunk head: "ix", length (the whole chunk: the reference skips unknown chunks by that many bytes from the chunk
//   start, QB3decode.cpp:254-255), version 3, flags (bit 0: entries carry the common factors, bit 1: block lengths),
//   the chunk's check (ix_seal_kernel), blocks per entry.  The pad makes the container parse the same if a reader adds the 4
//   head bytes to the length.  (Entries with block lengths: their fill kernel writes the fixed fields too.);
// * the container header in front of the stream (device flavour).
// the dword that holds the boundary in front of chunk k (k == nchunks: the stream's end): the OR of the edge dwords of the
// chunks that meet in it, written by the thread of the FIRST boundary inside the dword.  cs(j): start of chunk j in the stream.
// positions in the last three bytes of `prev` at which four zero bytes start (they reach into `cur`)
__device__ __forceinline__ uint32_t zero_runs_into(uint32_t prev, uint32_t cur) { return zero_runs_in(prev, cur) - (uint32_t)(prev == 0); }
template <class CS>
__device__ __forceinline__ void finish_seam(const EncArgs &a, uint32_t k, CS cs) {
    const uint64_t Ek = (uint64_t)a.out_bit0 + cs(k);
    if (k == a.nchunks) { a.res->total_bits = Ek - a.out_bit0; }
    const uint64_t d = Ek >> 5;
    // (RLE0 modes) the zero runs enc_concat_kernel left out: the positions whose four bytes touch this boundary.  The dwords either
    // side of it are the chunks' own and final; bytes in front of the stream and behind it count as non-zero (they are not RLE0's input).
    const uint64_t d_last = ((uint64_t)a.out_bit0 + cs(a.nchunks) + 31) / 32;      // dwords the stream reaches into
    // (a chunk of less than three dwords next to the boundary -- the last chunk of a raster can be one block -- has its shared dwords side
    // by side, each assembled by another thread: no reading them here; sixteen positions are more than touch such a chunk)
    const bool tiny = a.zrun_probe && ((k > 0 && cs(k) - cs(k - 1) < 96) || (k < a.nchunks && cs(k + 1) - cs(k) < 96));
    if (tiny) atomicAdd((unsigned long long *)&a.res->zero_run, 16ull);
    if ((Ek & 31) == 0) {
        if (a.zrun_probe && !tiny && k > 0 && k < a.nchunks) {      // chunks meet on a dword boundary: runs from the last three bytes of one into the other
            const uint32_t n = zero_runs_into(a.out32[d - 1], a.out32[d]);
            if (n) atomicAdd((unsigned long long *)&a.res->zero_run, (unsigned long long)n);
        }
        return;
    }
    if (k > 0) { const uint64_t Ep = (uint64_t)a.out_bit0 + cs(k - 1); if ((Ep >> 5) == d && (Ep & 31)) return; }
    uint32_t v = k > 0 ? a.seams[2 * (k - 1) + 1] : 0u;
    for (uint32_t j = k; j < a.nchunks; j++) {
        v |= a.seams[2 * j];
        const uint64_t En = (uint64_t)a.out_bit0 + cs(j + 1);
        if ((En >> 5) != d || (En & 31) == 0) break;       // chunk j reaches the end of the dword
    }
    if (k == 0 && a.out_bit0) {                             // the dword the header ends in: the stream's bytes only
        uint8_t *p8 = (uint8_t *)(a.out32 + d);
        for (uint32_t i = a.out_bit0 >> 3; i < 4; i++) p8[i] = (uint8_t)(v >> (8 * i));
    } else a.out32[d] = v;
    if (a.zrun_probe && !tiny) {
        uint32_t vz = v;
        if (k == 0 && a.out_bit0) vz |= 0xffffffffu >> (32 - a.out_bit0);             // (header bytes)
        if (d + 1 == d_last) { const uint32_t used = (uint32_t)(((uint64_t)a.out_bit0 + cs(a.nchunks) + 7) / 8 - 4 * d); if (used < 4) vz |= 0xffffffffu << (8 * used); }     // (bytes behind the stream)
        uint32_t n = zero_runs_in(vz, d + 1 < d_last ? a.out32[d + 1] : 0xffffffffu);
        if (k > 0) n += zero_runs_into(a.out32[d - 1], vz);
        if (n) atomicAdd((unsigned long long *)&a.res->zero_run, (unsigned long long)n);
    }
}

__global__ void __launch_bounds__(256) enc_finish_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
    const uint32_t B = a.g.bands, tsz = a.g.tsz;
    if ((a.finish_what & 2) && blockIdx.x == 0 && a.hdr_len) {     // the stream starts at out32 + out_bit0 / 8; the prepared header bytes end hdr_back before
        uint8_t *start = (uint8_t *)a.out32 + (a.out_bit0 >> 3) - a.hdr_back;
        for (uint32_t i = threadIdx.x; i < a.hdr_len; i += blockDim.x) start[i] = a0.hdr[i];
    }
    if ((a.finish_what & 2) && a.have_idx)
        for (uint64_t sgi = k; sgi < a.g.nseg; sgi += nthreads) {
            const uint64_t v = a.idx.bitpos[sgi];
            const uint64_t bp = chunk_start(a, (uint32_t)(v >> 32)) + (v & 0xffffffffu);
            a.idx.bitpos[sgi] = bp;
            if (!a.ix_dst || sgi % a.ix_spe) continue;
            const uint32_t ke = (uint32_t)(sgi / a.ix_spe);        // the segment's entry of the restart table
            const uint32_t c = ke / a.ix_per_chunk, j = ke - c * a.ix_per_chunk;
            const uint32_t here = (a.ix_K - c * a.ix_per_chunk < a.ix_per_chunk) ? a.ix_K - c * a.ix_per_chunk : a.ix_per_chunk;   // entries of this chunk
            uint8_t *chunk = a.ix_dst + (uint64_t)c * (IX_HEAD + IX_PAD + (uint64_t)a.ix_per_chunk * a.ix_E);
            if (j == 0) {
                const uint32_t len = IX_HEAD + here * a.ix_E;
                chunk[0] = 'i'; chunk[1] = 'x'; chunk[2] = (uint8_t)len; chunk[3] = (uint8_t)(len >> 8);
                chunk[4] = 3; chunk[5] = (a.g.mode == CM_BEST ? 1 : 0) | (a.ix_bl ? 2 : 0); chunk[6] = 0; chunk[7] = 0;      // (bytes 6, 7: ix_seal_kernel)
                for (uint32_t i = 0; i < 4; i++) chunk[8 + i] = (uint8_t)(a.ix_blocks >> (8 * i));
                uint8_t *pad = chunk + len;
                pad[0] = 'z'; pad[1] = 'z'; pad[2] = 4; pad[3] = 0;
                if (c * a.ix_per_chunk + here == a.ix_K) { pad[4] = 'D'; pad[5] = 'T'; }
            }
            if (a.ix_bl) continue;              // (entries with block lengths: their fill kernel writes the fixed fields too)
            uint8_t *e = chunk + IX_HEAD + (uint64_t)j * a.ix_E;
            for (uint32_t i = 0; i < 6; i++) e[i] = (uint8_t)(bp >> (8 * i));
            e += 6;
            for (uint32_t c2 = 0; c2 < B; c2++) e[c2] = a.idx.rung[sgi * B + c2];
            e += B;
            const uint8_t *pv = (const uint8_t *)a.idx.prev + sgi * B * tsz;
            for (uint32_t i = 0; i < B * tsz; i++) e[i] = pv[i];
            if (a.g.mode == CM_BEST) {
                e += B * tsz;
                const uint8_t *cf = (const uint8_t *)a.idx.cf + sgi * B * tsz;
                for (uint32_t i = 0; i < B * tsz; i++) e[i] = cf[i];
            }
        }
    if (!(a.finish_what & 1) || k > a.nchunks) return;
    finish_seam(a, k, [&](uint32_t j) { return chunk_start(a, j); });
}

// ---- the strips of a pipelined host call (qb3_api.cpp, encode_pipelined).  A strip is a scan group of chunks; the strips are
// launched in order on one stream, so what a strip needs from the ones before -- the bits they produced, their last chunk's
// edge dword -- is there: offsets are prefix sums, the dependency only points backwards.
// scan: the strip's chunk offsets inside the group, and the running total behind it: group_sum[strip + 1] = group_sum[strip] + bits
__global__ void __launch_bounds__(SCAN_GROUP / 4) enc_scan_strip_kernel(const EncArgs a0, const uint32_t strip) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    __shared__ uint32_t wsum[16];
    const uint32_t tid = threadIdx.x, i0 = strip * SCAN_GROUP + 4 * tid;
    uint32_t v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = (i0 + k < a.nchunks) ? a.chunk_bits[i0 + k] : 0; sum += v[k]; }
    uint32_t total;
    uint64_t off = block_exscan(sum, wsum, &total);
#pragma unroll
    for (int k = 0; k < 4; k++) if (i0 + k < a.nchunks) { a.chunk_off[i0 + k] = off; off += v[k]; }
    if (tid == 0) {
        const uint64_t base = strip ? a.group_sum[strip] : 0ull;
        if (!strip) a.group_sum[0] = 0;
        a.group_sum[strip + 1] = base + total;              // (the last strip's: the stream length, where chunk_start looks for it)
    }
}
// seams: the boundaries in front of the strip's chunks (the one in front of its first chunk closes the strip before), and
// behind the last strip the stream's end
__global__ void __launch_bounds__(256) enc_finish_strip_kernel(const EncArgs a0, const uint32_t strip) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    const uint32_t k = a.chunk0 + blockIdx.x * blockDim.x + threadIdx.x;
    const bool last = a.chunk_end == a.nchunks;
    if (k > a.chunk_end || (k == a.chunk_end && !last)) return;
    // (the offset of the chunk behind the strip is the running total: its own chunk_off is the next strip's to write)
    finish_seam(a, k, [&](uint32_t j) { return j >= a.chunk_end ? a.group_sum[strip + 1] : a.group_sum[j / SCAN_GROUP] + a.chunk_off[j]; });
}

// Block lengths behind the entries' fixed fields (tables of level 2).  A block's bit length is the sum of its units'
// lengths (the index has them).  Four ten-bit fields are five whole bytes and an entry's 64 blocks are sixteen such
// groups: a thread per group of four blocks reads their 4 * B length bytes (whole dwords) and writes five bytes of
// the entry -- no two threads share a byte.
__global__ void __launch_bounds__(256) ix_bl_fill_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    static_assert(IX_BL_BITS == 10, "groups of four fields are five bytes");
    const uint64_t grp = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;       // blocks 4 * grp .. 4 * grp + 3
    const uint32_t B = a.g.bands;
    const uint64_t k = grp >> 4;                                                // 16 groups an entry
    if (k >= a.ix_K) return;
    const uint64_t nblocks = a.g.nblocks, blk0 = 4 * grp;
    uint32_t len[4] = {0, 0, 0, 0};
    if (blk0 + 4 <= nblocks && ((uintptr_t)a.idx.ulen & 3) == 0) {
        const uint32_t *ul = (const uint32_t *)((const uint8_t *)a.idx.ulen + blk0 * B);    // 4 * B bytes: B dwords
        if (B == 1) { const uint32_t v = ul[0]; len[0] = v & 255; len[1] = (v >> 8) & 255; len[2] = (v >> 16) & 255; len[3] = v >> 24; }
        else if (B == 3) {
            const uint32_t v0 = ul[0], v1 = ul[1], v2 = ul[2];
            len[0] = (v0 & 255) + ((v0 >> 8) & 255) + ((v0 >> 16) & 255);
            len[1] = (v0 >> 24) + (v1 & 255) + ((v1 >> 8) & 255);
            len[2] = ((v1 >> 16) & 255) + (v1 >> 24) + (v2 & 255);
            len[3] = ((v2 >> 8) & 255) + ((v2 >> 16) & 255) + (v2 >> 24);
        } else {
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) { const uint32_t v = ul[q]; len[q] = (v & 255) + ((v >> 8) & 255) + ((v >> 16) & 255) + (v >> 24); }
        }
    } else {
        const uint8_t *ul = (const uint8_t *)a.idx.ulen + blk0 * B;
        for (uint32_t q = 0; q < 4; q++)
            if (blk0 + q < nblocks) for (uint32_t c = 0; c < B; c++) len[q] += ul[q * B + c];
    }
    const uint64_t bits = (uint64_t)len[0] | (uint64_t)len[1] << 10 | (uint64_t)len[2] << 20 | (uint64_t)len[3] << 30;
    const uint32_t c = (uint32_t)(k / a.ix_per_chunk), jj = (uint32_t)(k - (uint64_t)c * a.ix_per_chunk);
    uint8_t *e0 = a.ix_dst + (uint64_t)c * (IX_HEAD + IX_PAD + (uint64_t)a.ix_per_chunk * a.ix_E) + IX_HEAD + (uint64_t)jj * a.ix_E;
    uint8_t *e = e0 + 6 + 2 * B + 5 * (uint32_t)(grp & 15);
#pragma unroll
    for (uint32_t i = 0; i < 5; i++) e[i] = (uint8_t)(bits >> (8 * i));
    // the entry's fixed fields (an entry per segment: ix_spe == 1, 8-bit values, no factors): at most 14 bytes, one per thread of
    // the entry's sixteen
    const uint32_t t = (uint32_t)(grp & 15);
    if (t < 6) e0[t] = (uint8_t)(a.idx.bitpos[k] >> (8 * t));
    else if (t < 6 + B) e0[t] = a.idx.rung[k * B + (t - 6)];
    else if (t < 6 + 2 * B) e0[t] = ((const uint8_t *)a.idx.prev)[k * B + (t - 6 - B)];
}

// four 24-bit fields as three dwords at any byte address (global memory takes unaligned dword stores)
__device__ __forceinline__ void ix_store12(uint8_t *e, const uint32_t (&f)[4]) {
    typedef uint32_t u32_a1 __attribute__((aligned(1)));
    u32_a1 *d = (u32_a1 *)e;
    d[0] = f[0] | f[1] << 24; d[1] = f[1] >> 8 | f[2] << 16; d[2] = f[2] >> 16 | f[3] << 8;
}
// The same for common-factor streams with a block table (8-bit grey / RGB / RGBA; 32/64-bit, one band): a three-byte field per
// block -- its bits (12) and the rungs its units are entered with (8-bit data: 3 bits a band; wide data: the band's whole rung)
// -- from the index's block table; a thread per four blocks writes twelve bytes and its share of the entry's fixed part
// (position, rungs, entering values, factors in force: 6 + bands * (1 + 2 * value size) bytes)
__global__ void __launch_bounds__(256) ix_bl_best_fill_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    const uint64_t grp = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;       // blocks 4 * grp .. 4 * grp + 3
    const uint32_t B = a.g.bands, tsz = a.g.tsz;
    const uint64_t k = grp >> 4;                                                // 16 groups an entry (64 blocks)
    if (k >= a.ix_K) return;
    const uint64_t nblocks = a.g.nblocks, blk0 = 4 * grp;
    const uint32_t c = (uint32_t)(k / a.ix_per_chunk), jj = (uint32_t)(k - (uint64_t)c * a.ix_per_chunk);
    uint8_t *e0 = a.ix_dst + (uint64_t)c * (IX_HEAD + IX_PAD + (uint64_t)a.ix_per_chunk * a.ix_E) + IX_HEAD + (uint64_t)jj * a.ix_E;
    const uint32_t t = (uint32_t)(grp & 15), fixed = 6 + B * (1 + 2 * tsz);
    uint8_t *e = e0 + fixed + 4 * IX_BL_BEST_BYTES * t;
    uint32_t fl[4];
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) {
        const uint32_t bt = blk0 + q < nblocks ? ((const uint32_t *)a.idx.ulen)[blk0 + q] : 0u;
        uint32_t f = bt & 0xfffu;
        if (tsz == 1) {
#pragma unroll
            for (uint32_t cc = 0; cc < 4; cc++) f |= ((bt >> (16 + 4 * cc)) & 7u) << (12 + 3 * cc);
        } else f |= ((bt >> 16) & 63u) << 12;
        fl[q] = f;
    }
    ix_store12(e, fl);          // (four three-byte fields: three dwords at whatever address the entry puts them)
    for (uint32_t i = t; i < fixed; i += 16) {
        uint8_t v;
        if (i < 6) v = (uint8_t)(a.idx.bitpos[k] >> (8 * i));
        else if (i < 6 + B) v = a.idx.rung[k * B + (i - 6)];
        else if (i < 6 + B + B * tsz) v = ((const uint8_t *)a.idx.prev)[k * B * tsz + (i - 6 - B)];
        else v = ((const uint8_t *)a.idx.cf)[k * B * tsz + (i - 6 - B - B * tsz)];
        e0[i] = v;
    }
}

// The same for 16-bit rasters of four or eight bands: a field is the bit length of a band PAIR (two units), two fields per
// lane of the decoder's wave (lane = block of the segment x band group of four), 128 fields an entry.  A thread per four
// fields (two lanes): five whole bytes; the entry's first 6 + 3 * bands threads also write one byte each of its fixed part.
__global__ void __launch_bounds__(256) ix_bl16_fill_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    static_assert(IX_BL_BITS == 10, "groups of four fields are five bytes");
    const uint64_t grp = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;       // fields 4 * grp .. 4 * grp + 3 of entry grp / 32
    // (a single band: one field per lane -- the unit's length -- 64 fields an entry, 16 threads)
    const uint32_t B = a.g.bands, BG = a.px16_bg, FPL = B == 1 ? 1 : 2, NG = B / BG, NB = 64 / NG, tpe = 16 * FPL;
    const uint64_t k = grp / tpe;
    if (k >= a.ix_K) return;
    const uint32_t t = (uint32_t)(grp - k * tpe);
    uint64_t bits = 0;
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) {
        const uint32_t field = 4 * t + q, lane = field / FPL, pair = field - lane * FPL;
        const uint32_t slot = lane / NG, g4 = lane - slot * NG;
        const uint64_t blk = k * NB + slot;
        uint32_t len = 0;
        if (slot < NB && blk < a.g.nblocks) {       // (lanes behind the segment's blocks x groups: nothing)
            // the lane's bands: BG from band BG * g4; four: the pairs (0,1), (2,3); three: (0,1) and band 2; two: a field a band; one: the unit
            const uint16_t *ul = (const uint16_t *)a.idx.ulen + blk * B + BG * g4;
            if (BG == 1) len = ul[0];
            else if (BG == 2) len = ul[pair];
            else if (BG == 3) len = pair ? (uint32_t)ul[2] : (uint32_t)ul[0] + ul[1];
            else len = (uint32_t)ul[2 * pair] + ul[2 * pair + 1];
        }
        bits |= (uint64_t)len << (IX_BL_BITS * q);
    }
    const uint32_t c = (uint32_t)(k / a.ix_per_chunk), jj = (uint32_t)(k - (uint64_t)c * a.ix_per_chunk);
    uint8_t *e0 = a.ix_dst + (uint64_t)c * (IX_HEAD + IX_PAD + (uint64_t)a.ix_per_chunk * a.ix_E) + IX_HEAD + (uint64_t)jj * a.ix_E;
    uint8_t *e = e0 + 6 + 3 * B + 5 * t;
#pragma unroll
    for (uint32_t i = 0; i < 5; i++) e[i] = (uint8_t)(bits >> (8 * i));
    // fixed part: bit position, a rung byte per band, the entering values (two bytes a band): 6 + 3 * B <= 30 bytes
    if (t < 6) e0[t] = (uint8_t)(a.idx.bitpos[k] >> (8 * t));
    else if (t < 6 + B) e0[t] = a.idx.rung[k * B + (t - 6)];
    else if (t < 6 + 3 * B) e0[t] = ((const uint8_t *)a.idx.prev)[k * 2 * B + (t - 6 - B)];
}

// 32/64-bit rasters: a twelve-bit length per UNIT of the segment (the unit-parallel decoder wants every unit's place).
// A thread per two fields: three whole bytes; the entry's threads share its fixed part a byte each.
__global__ void __launch_bounds__(256) ix_blw_fill_kernel(const EncArgs a0, const uint32_t tpe) {     // tpe: threads per entry
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    static_assert(IX_BL_BITS_WIDE == 12, "pairs of fields are three bytes");
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t B = a.g.bands, tsz = a.g.tsz, upe = a.ix_blocks * B;         // units per entry
    const uint64_t k = idx / tpe;
    if (k >= a.ix_K) return;
    const uint32_t t = (uint32_t)(idx - k * tpe);
    const uint64_t u0 = k * upe, nunits = a.g.nblocks * B;
    uint32_t len[2] = {0, 0};
    for (uint32_t q = 0; q < 2; q++) {
        const uint32_t f = 2 * t + q;
        if (f < upe && u0 + f < nunits) len[q] = a.g.ulen_sz == 1 ? (uint32_t)((const uint8_t *)a.idx.ulen)[u0 + f] : (uint32_t)((const uint16_t *)a.idx.ulen)[u0 + f];
    }
    const uint32_t bits = len[0] | len[1] << 12;
    const uint32_t c = (uint32_t)(k / a.ix_per_chunk), jj = (uint32_t)(k - (uint64_t)c * a.ix_per_chunk);
    uint8_t *e0 = a.ix_dst + (uint64_t)c * (IX_HEAD + IX_PAD + (uint64_t)a.ix_per_chunk * a.ix_E) + IX_HEAD + (uint64_t)jj * a.ix_E;
    const uint32_t fixed = 6 + B * (1 + tsz), nbytes = (upe * 12 + 7) / 8;
    uint8_t *e = e0 + fixed + 3 * t;
    for (uint32_t i = 0; i < 3; i++) if (3 * t + i < nbytes) e[i] = (uint8_t)(bits >> (8 * i));
    for (uint32_t i = t; i < fixed; i += tpe) {         // bit position, a rung byte per band, the entering values
        if (i < 6) e0[i] = (uint8_t)(a.idx.bitpos[k] >> (8 * i));
        else if (i < 6 + B) e0[i] = a.idx.rung[k * B + (i - 6)];
        else e0[i] = ((const uint8_t *)a.idx.prev)[k * B * tsz + (i - 6 - B)];
    }
}

// Common-factor streams of the lane-per-unit decoder (k_dec_pxu.hip): a three-byte field per UNIT of the entry's segment -- its bits (12)
// | the rung it is entered with << 12 -- from the index's unit table; a thread per four units, sixteen threads an entry (a segment is at
// most 64 units), which also share the entry's fixed part
__global__ void __launch_bounds__(256) ix_blu_best_fill_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t B = a.g.bands, tsz = a.g.tsz, upe = a.ix_blocks * B;         // units per entry
    const uint64_t k = idx >> 4;
    if (k >= a.ix_K) return;
    const uint32_t t = (uint32_t)(idx & 15);
    const uint64_t u0 = k * upe, nunits = a.g.nblocks * B;
    const uint32_t c = (uint32_t)(k / a.ix_per_chunk), jj = (uint32_t)(k - (uint64_t)c * a.ix_per_chunk);
    uint8_t *e0 = a.ix_dst + (uint64_t)c * (IX_HEAD + IX_PAD + (uint64_t)a.ix_per_chunk * a.ix_E) + IX_HEAD + (uint64_t)jj * a.ix_E;
    const uint32_t fixed = 6 + B * (1 + 2 * tsz);
    if (4 * t + 4 <= upe) {     // four whole fields: three dwords
        uint32_t fl[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            const uint32_t f = 4 * t + q;
            const uint32_t bt = u0 + f < nunits ? ((const uint32_t *)a.idx.ulen)[u0 + f] : 0u;
            fl[q] = (bt & 0xfffu) | ((bt >> 16) & 63u) << 12;
        }
        ix_store12(e0 + fixed + IX_BL_BEST_BYTES * 4 * t, fl);
    } else
        for (uint32_t q = 0; q < 4; q++) {
            const uint32_t f = 4 * t + q;
            if (f >= upe) break;
            const uint32_t bt = u0 + f < nunits ? ((const uint32_t *)a.idx.ulen)[u0 + f] : 0u;
            const uint32_t fld = (bt & 0xfffu) | ((bt >> 16) & 63u) << 12;
            uint8_t *e = e0 + fixed + IX_BL_BEST_BYTES * f;
            e[0] = (uint8_t)fld; e[1] = (uint8_t)(fld >> 8); e[2] = (uint8_t)(fld >> 16);
        }
    for (uint32_t i = t; i < fixed; i += 16) {
        uint8_t v;
        if (i < 6) v = (uint8_t)(a.idx.bitpos[k] >> (8 * i));
        else if (i < 6 + B) v = a.idx.rung[k * B + (i - 6)];
        else if (i < 6 + B + B * tsz) v = ((const uint8_t *)a.idx.prev)[k * B * tsz + (i - 6 - B)];
        else v = ((const uint8_t *)a.idx.cf)[k * B * tsz + (i - 6 - B - B * tsz)];
        e0[i] = v;
    }
}

// The last word on the table: every chunk's 16-bit check of its entries into the head's reserved bytes (ix_sum_part)
__global__ void __launch_bounds__(256) ix_seal_kernel(const EncArgs a0) {
    const EncArgs a = enc_for_tile(a0, blockIdx.y);
    __shared__ uint32_t part[4];
    const uint32_t c = blockIdx.x;
    const uint32_t here = (a.ix_K - c * a.ix_per_chunk < a.ix_per_chunk) ? a.ix_K - c * a.ix_per_chunk : a.ix_per_chunk;
    uint8_t *chunk = a.ix_dst + (uint64_t)c * (IX_HEAD + IX_PAD + (uint64_t)a.ix_per_chunk * a.ix_E);
    uint32_t s = ix_sum_part(chunk + IX_HEAD, here * a.ix_E, threadIdx.x, 256);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += (uint32_t)__shfl_xor((int)s, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t f = ix_sum_fold(part[0] + part[1] + part[2] + part[3]);
        chunk[6] = (uint8_t)f; chunk[7] = (uint8_t)(f >> 8);
    }
}

static void launch_enc_tables(const EncArgs &a, hipStream_t st);
void launch_enc_post(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    const uint32_t nt = a.ntiles;
    {
        ProfScope ps("enc_scan", st);
        hipLaunchKernelGGL(enc_scan_kernel, dim3((plan.nchunks + SCAN_GROUP - 1) / SCAN_GROUP, nt), dim3(SCAN_GROUP / 4), 0, st, a);
    }
    {
        ProfScope ps("enc_concat", st);
        hipLaunchKernelGGL(enc_concat_kernel, dim3((plan.nchunks + 3) / 4, nt), dim3(256), 0, st, a);
    }
    ProfScope ps("enc_seams", st);
    {       // boundaries, stream length, index positions, the restart table's entries and chunk heads, the header: one launch
        const uint64_t want = std::max<uint64_t>(plan.nchunks + 1, a.have_idx ? std::min<uint64_t>(a.g.nseg, (uint64_t)1 << 20) : 0);
        hipLaunchKernelGGL(enc_finish_kernel, dim3((uint32_t)((want + 255) / 256), nt), dim3(256), 0, st, a);
    }
    launch_enc_tables(a, st);
}
// one strip (a.chunk0 .. a.chunk_end = scan group `strip`): offsets, concatenation, seams -- behind it the stream is final up
// to the dword that holds the strip's end
void launch_enc_post_strip(const EncArgs &a, const EncPlan &plan, hipStream_t st, uint32_t strip) {
    (void)plan;
    const uint32_t nt = a.ntiles, n = a.chunk_end - a.chunk0;
    { ProfScope ps("enc_scan", st); hipLaunchKernelGGL(enc_scan_strip_kernel, dim3(1, nt), dim3(SCAN_GROUP / 4), 0, st, a, strip); }
    { ProfScope ps("enc_concat", st); hipLaunchKernelGGL(enc_concat_kernel, dim3((n + 3) / 4, nt), dim3(256), 0, st, a); }
    ProfScope ps("enc_seams", st);
    hipLaunchKernelGGL(enc_finish_strip_kernel, dim3((n + 1 + 255) / 256, nt), dim3(256), 0, st, a, strip);
}
// behind the last strip: index positions, the restart table, the header bytes (a.finish_what == 2)
void launch_enc_post_tail(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    (void)plan;
    ProfScope ps("enc_seams", st);
    if (a.have_idx || a.hdr_len) {
        const uint64_t want = std::max<uint64_t>(1, a.have_idx ? std::min<uint64_t>(a.g.nseg, (uint64_t)1 << 20) : 0);
        hipLaunchKernelGGL(enc_finish_kernel, dim3((uint32_t)((want + 255) / 256), a.ntiles), dim3(256), 0, st, a);
    }
    launch_enc_tables(a, st);
}
static void launch_enc_tables(const EncArgs &a, hipStream_t st) {
    const uint32_t nt = a.ntiles;
    if (a.ix_dst && a.have_idx && a.ix_bl && lane_per_unit_shape(a.g.tsz, a.g.mode, a.g.bands)) {       // a field per unit (k_dec_pxu.hip)
        if (a.g.mode == CM_BEST) hipLaunchKernelGGL(ix_blu_best_fill_kernel, dim3((uint32_t)(((uint64_t)a.ix_K * 16 + 255) / 256), nt), dim3(256), 0, st, a);
        else {
            const uint32_t tpe = (a.ix_blocks * a.g.bands + 1) / 2;
            hipLaunchKernelGGL(ix_blw_fill_kernel, dim3((uint32_t)(((uint64_t)a.ix_K * tpe + 255) / 256), nt), dim3(256), 0, st, a, tpe);
        }
        hipLaunchKernelGGL(ix_seal_kernel, dim3((a.ix_K + a.ix_per_chunk - 1) / a.ix_per_chunk, nt), dim3(256), 0, st, a);
        return;
    }
    if (a.ix_dst && a.have_idx && a.ix_bl && a.g.mode == CM_BEST) hipLaunchKernelGGL(ix_bl_best_fill_kernel, dim3((uint32_t)(((uint64_t)a.ix_K * 16 + 255) / 256), nt), dim3(256), 0, st, a);
    else if (a.ix_dst && a.have_idx && a.ix_bl && a.g.tsz == 1) hipLaunchKernelGGL(ix_bl_fill_kernel, dim3((uint32_t)(((uint64_t)a.ix_K * 16 + 255) / 256), nt), dim3(256), 0, st, a);
    if (a.ix_dst && a.have_idx && a.ix_bl && a.g.tsz >= 4 && a.g.mode != CM_BEST) {
        const uint32_t tpe = (a.ix_blocks * a.g.bands + 1) / 2;
        hipLaunchKernelGGL(ix_blw_fill_kernel, dim3((uint32_t)(((uint64_t)a.ix_K * tpe + 255) / 256), nt), dim3(256), 0, st, a, tpe);
    }
    if (a.ix_dst && a.have_idx && a.ix_bl && a.g.tsz == 2 && a.g.mode != CM_BEST) hipLaunchKernelGGL(ix_bl16_fill_kernel, dim3((uint32_t)(((uint64_t)a.ix_K * (a.g.bands == 1 ? 16 : 32) + 255) / 256), nt), dim3(256), 0, st, a);
    if (a.ix_dst && a.have_idx) hipLaunchKernelGGL(ix_seal_kernel, dim3((a.ix_K + a.ix_per_chunk - 1) / a.ix_per_chunk, nt), dim3(256), 0, st, a);
}

}  // namespace qb3dev
