// qb3_amd/csrc/qb3_dev.h -- internal interface between the host API (qb3_api.cpp) and the HIP kernels
// (qb3_kernels.hip).  Nothing here is exported from the shared library.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace qb3dev {

constexpr int MAXBANDS = 16;
constexpr uint64_t ZCURVE = 0x0145236789cdabefull;   // reference QB3common.h:185
constexpr uint64_t HILBERT = 0x01548cd9aefb7623ull;  // reference QB3common.h:193

enum CodecMode { CM_FTL = 0, CM_BASE = 1, CM_BEST = 2 };   // no step / step / step + common factor + index

// Image geometry and coding parameters shared by encode and decode.
struct Geometry {
    uint32_t w, h, bands, tsz;      // tsz = bytes per value (1,2,4,8)
    uint64_t stride;                // line stride in values
    uint64_t order;                 // curve nibbles, never 0 here
    uint32_t nbx, nby;              // blocks per row / column (ceil)
    uint64_t nblocks;
    uint32_t mode;                  // CodecMode
    uint32_t seg_blocks;            // blocks per index segment
    uint64_t nseg;                  // number of index segments
    uint32_t ulen_sz;               // bytes per entry of the per-unit bit-length table (1, 2; 0 = none); 4: a dword per BLOCK instead
                                    // (8-bit common-factor streams of 1/3/4 bands: the block's bits | entering rungs << 16, four bits a band);
                                    // ULEN_UNIT: a dword per UNIT (every other common-factor stream of several bands: the unit's bits | the rung it is entered with << 16)
    uint8_t cband[MAXBANDS];
};

// Per-band coder state, the reference's band_state (QB3common.h:63-65)
struct BandState {
    uint64_t prev[MAXBANDS];
    uint64_t cf[MAXBANDS];
    uint8_t rung[MAXBANDS];
};

// Decode index, device resident.  Layout in one allocation of index_bytes(g):
//   u64 bitpos[nseg]; T prev[nseg*bands]; T cf[nseg*bands]; u8 rung[nseg*bands]; ulen[nblocks*bands]
// (each 8-byte aligned).  A segment entry is the coder state on entering a run of seg_blocks blocks.  For
// FTL/BASE streams segments are as large as a decoder workgroup and `ulen` holds the bit length of every unit
// (u8 for 8-bit data, u16 otherwise), which is what lets the decoder place every unit with a scan; for the
// common-factor modes segments are short, there is no ulen, and a lane walks each segment serially.
struct IndexView {
    uint64_t *bitpos;
    void *prev;
    void *cf;
    uint8_t *rung;
    void *ulen;
};
size_t index_bytes(const Geometry &g);
IndexView index_view(const Geometry &g, void *base);
uint32_t seg_blocks_for(const Geometry &g);      // needs w, h, bands, tsz, stride, order, mode, cband
uint32_t ulen_size_for(uint32_t tsz, uint32_t mode, uint32_t bands);
constexpr uint32_t ULEN_UNIT = 8;
constexpr uint32_t WIDE_PAD_DW = 40;      // zero words behind the staged words of a 32/64-bit segment (wide_values_lds, qb3_wide.h)
inline size_t ulen_table_bytes(const Geometry &g) { return g.ulen_sz == 4 ? (size_t)g.nblocks * 4 : g.ulen_sz == ULEN_UNIT ? (size_t)g.nblocks * g.bands * 4 : (size_t)g.nblocks * g.bands * g.ulen_sz; }
// common-factor streams whose index holds a dword per BLOCK (its bits | the rungs its units are entered with << 16) for a
// lane-per-block decoder: 8-bit rasters of 1/3/4 bands (four bits a band), 16/32/64-bit rasters of one band (the whole rung)
inline bool best_block_table(uint32_t tsz, uint32_t mode, uint32_t bands) {
    return mode == CM_BEST && ((tsz == 1 && (bands == 1 || bands == 3 || bands == 4)) || (tsz >= 2 && bands == 1));
}

// Rasters of the lane-per-UNIT kernels (k_dec_pxu.hip): every shape no lane-per-block kernel takes -- 8-bit rasters of 2 or more than 4
// bands, 16-bit rasters of an odd band count above 4, 32/64-bit rasters of several bands; of the common-factor streams every one without a
// block table.  A wave owns an index segment of 64 / bands blocks, a lane one unit.  A function of value size, band count and mode
// only: encoder and decoder derive the segment size from it.
inline bool lane_per_unit_shape(uint32_t tsz, uint32_t mode, uint32_t bands) {
    if (bands < 1 || bands > (uint32_t)MAXBANDS) return false;
    if (mode == CM_BEST) return !best_block_table(tsz, mode, bands);
    if (tsz == 1) return !(bands == 1 || bands == 3 || bands == 4);
    if (tsz == 2) return bands > 4 && (bands & 1);
    return bands >= 2;
}

// Results the encoder hands back to the host (device resident, copied once per encode)
struct EncResult {
    uint64_t total_bits;
    uint64_t prev[MAXBANDS];
    uint64_t cf[MAXBANDS];
    uint32_t rung[MAXBANDS];
    // asked for with launch_encode's zrun_probe (the RLE0 modes), counted while the chunks are concatenated:
    uint64_t zero_run;          // byte positions at which four zero bytes in a row start -- at least as many as there are
    uint64_t ff_pairs;          // 0xff bytes followed by another 0xff -- at most as many as there are (rle0_may_win)
    uint64_t zero_dwords;       // aligned all-zero dwords that belong to one chunk of the stream alone -- at most as many as there are
};

// RLE0 (reference QB3encode.cpp:271-332) writes three bytes for every PAIR of 0xff bytes (a run of L of them holds L / 2
// pairs, at least (L - 1) / 2; the stream's last two bytes are copied) and three bytes for 4 + r zero bytes, which saves
// at most one byte per position at which four zero bytes start.  So its output is at least n + ff_pairs / 2 - 1 - zero_run
// bytes, and it can only be shorter than n when that is below n: for most streams the counts decide without the byte pass.
inline bool rle0_may_win(const EncResult &r) { return r.zero_run > 0 && 2 * r.zero_run + 2 > r.ff_pairs; }
// 4 KB of zero bytes hold at least 1022 aligned all-zero dwords, of which at most two per chunk of the stream are shared with a
// neighbour; a chunk is at least min_chunk_bytes long (the plan's blocks per chunk x bands x two bits), so 4 KB touch at most
// 4096 / min_chunk_bytes + 2 of them (the 8-bit lane-per-block plan: 63 bytes, 67 chunks, 888 dwords left); 4 KB of 0xff hold
// three pairs in each of those dwords.  Below both counts the table of one-valued 4 KB chunks the byte pass skips long runs
// by is all "no" and need not be made.  Plans with chunks too short for the count to say anything always make the table.
inline bool rle0_no_uniform_chunk(const EncResult &r, uint64_t min_chunk_bytes) {
    if (min_chunk_bytes < 16) return false;
    const uint64_t shared = 2 * (4096 / min_chunk_bytes + 2);
    return shared < 1000 && r.zero_dwords < 1022 - shared && r.ff_pairs < 1024;
}

// Workspace sizes
struct EncPlan {
    uint32_t threads;       // workgroup size = units per chunk (incl. halo units)
    uint32_t slots;         // block slots per chunk, slot 0 is the halo block
    uint32_t nchunks;
    size_t lds_bytes;
    size_t ws_bytes;        // chunk counts/offsets, seam table, per-chunk scratch slots, EncResult (last)
    uint32_t nbp;           // payload blocks per chunk
    bool px;                // 8-bit 1/3/4-band register-resident kernel applies (lane per block; also for the common-factor modes: k_enc_px_best.hip)
    bool px_rgb;            //   ... with the default R-G,G,B-G map (else identity)
    bool px16;              // 16-bit register-resident kernel applies (lane per block and band group)
    uint32_t px16_bg, px16_ng;      //   ... bands per lane (1..4), lanes per block
    bool pxw;               // 32/64-bit single-band register-resident kernel applies (lane per block; k_enc_pxw.hip)
    bool pxw_best;          // ... its front end under the common-factor analysis (k_enc_best.hip, FRONT = 1 / 2)
};
EncPlan plan_encode(const Geometry &g);
constexpr uint32_t PXB_LDS_FIXED = 11664;      // LDS of the 8-bit common-factor lane-per-block encoder in front of its bit buffer

// Optional restart table carried INSIDE the container as ignorable chunks ("ix", include/qb3x.h): K entries, one per
// `blocks` blocks: [bit position, 6 bytes][rung, 1 byte per band][prev, tsz bytes per band][cf, the same,
// common-factor modes only].  A chunk holds at most 64 KB, so the table is a run of chunks of `per_chunk` entries
// (the last one may hold fewer), each [12-byte head][entries][4-byte "zz" pad chunk]; `base` points at the first
// chunk's first byte.  With it a stream that arrives without the out-of-band index is walked from K points at once.
constexpr uint32_t IX_HEAD = 12, IX_PAD = 4;
constexpr uint32_t IX_BL_BITS_WIDE = 12;  // ... of a unit length in a table of 32/64-bit data (a unit is at most 1048 bits; two fields are three whole bytes)
constexpr uint32_t IX_BL_BITS = 10;       // bits of a block length in a table entry (8-bit data, at most four bands: 4 x 149 < 1024)
struct IxTable {
    uint8_t *base = nullptr;        // device pointer (encode: where the chunks go, "DT" follows; decode: where they are)
    uint32_t K = 0, blocks = 0, entry_bytes = 0, per_chunk = 0;
    bool pads = true;               // false: a version 1 table (one chunk, no pad chunk behind it)
    uint32_t version = 3;           // decode: what the chunks say (3: every chunk carries a 16-bit check of its entries in the head's reserved bytes)
    bool check_heads = false;       // decode: the chunk heads behind the first were not read on the host: the check kernel looks at them
    bool block_lens = false;        // entries carry the bit length of every block of their segment (10 bits each): 8-bit lane-per-block rasters
    bool own_index = false;         // encode: the index is the library's own, only sampled for the table -- no unit lengths needed
};
uint32_t ix_entry_bytes(const Geometry &g, bool block_lens = false);
bool ix_block_lens_ok(const Geometry &g);         // can a table for this geometry carry block lengths
uint32_t ix_bl_fields(const Geometry &g);         // ... how many fields an entry then ends with
uint32_t px16_bands_per_lane(const Geometry &g);  // 16-bit lane-per-block kernels: bands a lane owns (0: the split does not apply)
inline uint32_t ix_bl_bits(uint32_t tsz) { return tsz >= 4 ? IX_BL_BITS_WIDE : IX_BL_BITS; }
constexpr uint32_t IX_BL_BEST_BYTES = 3;  // ... of a block's field in a table of 8-bit common-factor data: the block's bits (12) | the rungs its units are entered with (3 bits a band) << 12
// bytes of the fields behind the fixed part of an entry that covers `blocks` blocks (cf: the stream is a common-factor one)
// (rasters of the lane-per-unit kernels: a field per UNIT -- three bytes in a common-factor table (the unit's bits | the rung it is
// entered with << 12), twelve bits of length otherwise, whatever the value size)
inline uint32_t ix_bl_bytes(uint32_t tsz, uint32_t bands, uint32_t blocks, bool cf) {
    const bool per_unit = lane_per_unit_shape(tsz, cf ? CM_BEST : CM_FTL, bands);
    if (cf) return IX_BL_BEST_BYTES * blocks * (per_unit ? bands : 1u);
    if (per_unit) return (blocks * bands * IX_BL_BITS_WIDE + 7) / 8;
    const uint32_t fields = tsz == 1 ? blocks : tsz == 2 ? (bands == 1 ? 64u : 128u) : blocks * bands;
    return (fields * ix_bl_bits(tsz) + 7) / 8;
}
// the table this library writes for a geometry (needs seg_blocks, nseg, bands, tsz, mode); K == 0: none
IxTable ix_layout(const Geometry &g, int level = 1);       // level 2: with block lengths where the geometry allows
inline size_t ix_chunks(const IxTable &t) { return t.per_chunk ? (t.K + t.per_chunk - 1) / t.per_chunk : 0; }
inline size_t ix_total_bytes(const IxTable &t) { return ix_chunks(t) * (IX_HEAD + (t.pads ? IX_PAD : 0)) + (size_t)t.K * t.entry_bytes; }

// Batched tiles: n images/streams laid out at fixed byte pitches, processed by one set of launches (blockIdx.y).
// n == 0 means a single image.  ws_pitch = plan.ws_bytes of one tile; idx_pitch = index_bytes of one tile.
struct TileBatch { uint32_t n = 0; uint64_t src_pitch = 0, dst_pitch = 0, ws_pitch = 0, idx_pitch = 0; uint64_t max_bits = 0; /* decode: the longest stream */ };
// Plain 8- and 16-bit streams (no index, no restart table) are walked through a table of unit lengths by position (k_dec_walk.hip).
// walk_table_bytes: memory that takes the whole call in one round; less means more rounds, down to walk_table_min_bytes.
size_t walk_table_bytes(uint32_t ntiles, uint64_t max_bits, uint32_t tsz);
size_t walk_table_min_bytes(uint32_t ntiles, uint32_t tsz);
size_t walk_memory_bytes(const Geometry &g, uint32_t ntiles, uint64_t max_bits);     // what the walk that will run for this raster wants (exits need far less than chains)
size_t walk_table_cap();                // what a decoder allocates at most (1 GiB; QB3_WALK_TAB_KB overrides)
struct DecPlan;
bool walk_table_applies(const Geometry &g, const DecPlan &plan);

// Encode the block stream of one image.  All pointers are device pointers.
//   img        image, g.tsz-byte values
//   out32      4-byte aligned address at or below the first stream byte
//   out_bit0   bit offset of the first stream bit relative to out32 (multiple of 8, < 32)
//   st_in      initial band state (host copy, passed by value to the kernels)
//   ws         workspace of plan.ws_bytes; the EncResult is its LAST sizeof(EncResult) bytes
//   index      optional decode index (nullptr = none)
// Launches on `stream`, does not synchronise.  Returns hipError_t as int.
// ix: base == nullptr when no table is wanted
int launch_encode(const Geometry &g, const EncPlan &plan, const void *img, uint32_t *out32, uint32_t out_bit0,
                  const BandState &st_in, void *ws, void *index, void *stream, const TileBatch &tb = TileBatch(),
                  const uint8_t *hdr = nullptr, uint32_t hdr_len = 0,      // hdr: container header stamped before each stream
                  const IxTable &ix = IxTable(),
                  bool zrun_probe = false);     // count runs of zero bytes and pairs of 0xff while concatenating (EncResult::zero_run, ff_pairs)

// Strips of a pipelined host call (k_host.hip): can this coding be cut into strips; how many; blocks of the raster a strip and
// its predecessors cover (what must be in device memory before it is coded); where the running stream length behind a strip is
bool encode_strips_ok(const Geometry &g, const EncPlan &plan);
uint32_t encode_strip_count(const EncPlan &plan);
uint64_t encode_strip_blocks(const EncPlan &plan, uint32_t strip);
const uint64_t *encode_strip_total(const Geometry &g, const EncPlan &plan, void *ws, uint32_t strip);
int launch_encode_strip(const Geometry &g, const EncPlan &plan, const void *img, uint32_t *out32, uint32_t out_bit0,
                        const BandState &st_in, void *ws, void *index, void *stream, const uint8_t *hdr, uint32_t hdr_len, const IxTable &ix, uint32_t strip);
int launch_encode_tail(const Geometry &g, const EncPlan &plan, const void *img, uint32_t *out32, uint32_t out_bit0,
                       const BandState &st_in, void *ws, void *index, void *stream, const uint8_t *hdr, uint32_t hdr_len, const IxTable &ix);

struct DecPlan {
    uint32_t threads;       // lanes per workgroup, one index segment per lane
    uint32_t nwg;
    size_t lds_bytes;
    size_t ws_bytes;        // scratch for the foreign-stream path: rebuilt index + status word
    // unit-parallel kernel (FTL/BASE with ordinary band maps): one workgroup per index segment
    bool fast;
    uint32_t threads2, bpp, passes, in_cap_dw;
    size_t lds2_bytes;
    // 8-bit 1/3/4-band lane-per-block kernel
    bool px, px_rgb;
    size_t lds_px;
    uint32_t px_cap_dw;     // staging capacity (dwords) of the lane-per-block kernels, per wave
    bool px16;              // 16-bit lane-per-(block, band group) kernel applies
    uint32_t px16_bg, px16_ng;
    bool px_best;           // 8-bit common-factor streams: the lane-per-block decoder applies (k_dec_px_best.hip)
    bool pxw;               // 32/64-bit single-band FTL/BASE streams: the lane-per-block decoder applies (k_dec_pxw.hip)
    bool pxw_best;          // ... and its common-factor counterpart (dec_pxw_best_kernel)
    size_t lds_pxw;
    bool pxu, pxu_best;     // the lane-per-unit decoders apply (k_dec_pxu.hip: lane_per_unit_shape): FTL / BASE, common factor
};
DecPlan plan_decode(const Geometry &g);

struct DecStrip { uint64_t seg0, nseg; bool first; };     // first: zero the status words, check the table; later strips skip both
// Decode a block stream.  in32/in_bit0 locate the first stream bit like out32/out_bit0 above; in_bits is
// the stream length in bits (bytes*8).  index == nullptr: the index is first rebuilt in ws by a serial
// boundary scan on the GPU.  status (device u32 inside ws, zeroed here) gets nonzero on decode failure.
// Returns hipError_t as int; does not synchronise.
int launch_decode(const Geometry &g, const DecPlan &plan, const uint32_t *in32, uint32_t in_bit0, uint64_t in_bits,
                  void *img, const void *index, void *ws, uint32_t **status_out, void *stream, const TileBatch &tb = TileBatch(),
                  const uint64_t *tile_bits = nullptr,     // tile_bits: device array, stream length of each tile in bits
                  const IxTable &ix = IxTable(),
                  void *walk_tab = nullptr, size_t walk_tab_bytes = 0,     // table memory for plain 8-bit streams (null: the one-wave walk)
                  bool full_staging = false,    // 16-bit data: worst-case LDS staging (after a call that ended with status bit 4)
                  uint32_t wide_band = 16,      // plain 32/64-bit streams: rungs the walk's table covers (16; 14: byte entries, 8: half rows -- QB3_WIDE_BAND, test hooks; the first
                                                // unit of a block row, entered from the far end of the row before, sits many rungs above its neighbours)
                  const DecStrip *strip = nullptr);     // a strip of a pipelined host call (decode_strips_ok): only these segments, from the container's table

// can a container's table (with block fields, an entry per index segment) be decoded strip by strip: one launch of a
// lane-per-block decoder per range of segments, nothing else
bool decode_strips_ok(const Geometry &g, const DecPlan &plan, const IxTable &ix);

// The RLE0 byte pass of the *_RLE modes on device buffers (k_rle0.hip; reference QB3encode.cpp:271-332, QB3decode.cpp:267-307).
// ws: rle0_ws_bytes(n) bytes of device memory.  rle0_device_size returns the size of the coded (decode = false) or
// expanded (decode = true) form and synchronises the stream; rle0_device_write, called next with the same arguments,
// writes it (does not synchronise).
size_t rle0_ws_bytes(uint64_t n);
int rle0_device_size(const void *d_src, uint64_t n, void *ws, bool decode, uint64_t *total, void *stream, bool no_uniform_chunk = false);     // no_uniform_chunk: the caller knows that no 4 KB of the bytes hold one value only
int rle0_device_write(const void *d_src, uint64_t n, void *ws, bool decode, void *d_dst, void *stream);


// Elementwise helpers on device buffers (quantisation, reference QB3encode.cpp:137-186 / QB3decode.cpp:77-107)
int launch_quantize(void *dst, const void *src, const Geometry &g, int dtype, uint64_t q, bool away, void *stream);
int launch_dequantize(void *img, const Geometry &g, int dtype, uint64_t q, void *stream);

// Optional per-kernel timing with HIP events recorded on the launch stream (off by default).
void prof_enable(int level);        // 0 off, 1 every kernel, 2 the long kernels only (less event traffic)
void prof_reset();
void prof_collect();                                    // call after the stream was synchronised
bool prof_get(const char *name, double *total_ms, uint64_t *count);
int  prof_names(char *buf, size_t bufsize);             // comma separated kernel names seen so far

const char *last_error();
void set_error(const char *what, int hip_err);

}  // namespace qb3dev
