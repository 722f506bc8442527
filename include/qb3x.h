/*
 * include/qb3x.h -- extensions of the MI355X-native QB3 library that the reference does not have:
 * device-resident buffers, an out-of-band decode index, batched tiles.  The reference API (QB3.h)
 * only knows host pointers and one image per call (reference QB3lib/QB3.h:119-139).
 *
 * Why an index: a QB3 block stream has no restart points; a unit's bit position and rung depend on
 * every earlier unit (reference QB3decode.h:445-454), so a foreign stream can only be parsed
 * serially.  The encoder here can emit, next to the (bit-identical) stream, a small side table:
 * for every SEGMENT of `seg_blocks` consecutive 4x4 blocks the bit position of its first unit and the
 * per-band encoder state {previous value, rung, common factor} on entry -- i.e. the band_state the
 * reference keeps in its handle (reference QB3common.h:63-65), sampled along the stream.  With it,
 * decode is parallel over segments.  Without it the library first rebuilds the table on the GPU with
 * a serial boundary scan of the stream (slow, latency bound), then decodes in parallel.
 *
 * `stream` arguments are hipStream_t passed as void* (NULL = the default stream).  All `d_` pointers
 * are device pointers on the current HIP device.
 */
#ifndef QB3_AMD_QB3X_H
#define QB3_AMD_QB3X_H
#include "QB3.h"

#if defined(__cplusplus)
extern "C" {
#endif

/* Number of usable HIP devices; 0 means the library cannot encode or decode. */
int qb3x_device_count(void);

/* Device buffers of destroyed handles are kept (at most 24 buffers, 3 GiB) for the next handle of the process, so that a
 * caller who opens, decodes and closes a container per tile does not pay hipMalloc / hipFree every time; qb3x_trim returns
 * them to the HIP runtime.  (No counterpart in the reference, which owns no device memory.) */
void qb3x_trim(void);

/* Diagnostic: the status bits of the decoder handle's last decode call.  Bits 0, 1, 3, 4 are errors (the call returned 0),
 * bit 2 "the stream ended early" (the reference's reader clamps, QB3decode bitstream.h:36), bit 5 "the container's restart
 * table failed its check and was not used", bit 6 "a plain stream's walk by exits handed super-windows to its one hopping
 * lane" (slower, same pixels). */
unsigned qb3x_last_decode_status(const decsp p);

/* Bytes of device memory an index for this encoder's geometry needs (0 on error). */
size_t qb3x_index_size(const encsp p);
/* Same, from a decoder handle (valid after qb3_read_info). */
size_t qb3x_decoder_index_size(const decsp p);

/* Device-resident encode.  d_src: image laid out as qb3_encode expects; d_dst: at least
 * qb3_max_encoded_size(p) bytes, 4-byte aligned; d_index: qb3x_index_size(p) bytes or NULL.
 * Writes the complete QB3 container (headers + stream, STORED fallback included) to d_dst.
 * Returns its size in bytes, 0 on error (see qb3_get_encoder_state).  Synchronises `stream` once,
 * to learn the stream length.  Mode, band map, stride and handle statefulness as qb3_encode. */
size_t qb3x_encode_device(encsp p, const void *d_src, void *d_dst, void *d_index, void *stream);

/* Device-resident decode.  p: handle from qb3_read_start + qb3_read_info over a HOST copy of at least
 * the headers (source_size must be the true stream size); d_src: device copy of the whole container
 * (same bytes, 4-byte aligned); d_dst: qb3_decoded_size(p) bytes; d_index: index written by
 * qb3x_encode_device for this very stream, or NULL to rebuild it with the serial scan.
 * Returns decoded bytes, 0 on error. */
size_t qb3x_decode_device(decsp p, const void *d_src, void *d_dst, const void *d_index, void *stream);

/* Batched tiles: n images of the encoder's geometry, image i at d_src + i*src_pitch, container i
 * written at d_dst + i*dst_pitch (dst_pitch >= qb3_max_encoded_size, multiple of 4), index i at
 * d_index + i*qb3x_index_size (or NULL).  sizes[i] receives the container size (0 = failed).
 * The band state is reset before every tile (tiles are independent streams).  Returns the number of
 * tiles encoded.  One host synchronisation for the whole batch.  With qb3x_set_encoder_index_chunk on, every tile's
 * container carries its own restart table (at the same offset in all of them). */
size_t qb3x_encode_tiles(encsp p, const void *d_src, size_t n, size_t src_pitch,
                         void *d_dst, size_t dst_pitch, void *d_index, size_t *sizes, void *stream);

/* Batched decode of n containers of the image size and type of handle p (parsed from tile 0);
 * sizes[i] = container size of tile i.  Tiles whose header matches tile 0's go through one set of launches; a tile
 * of another kind -- a raw-stored tile in a batch of coded ones, as qb3x_encode_tiles writes for incompressible
 * data, or the reverse -- is parsed and decoded on its own.  d_index = NULL: the restart tables inside the containers
 * are used when every tile of the batch has one where tile 0 has it, else the streams are walked.  Returns the number
 * of tiles decoded;
 * qb3x_decode_tile_ok(p, i) then tells which (1 = tile i of the last call was decoded). */
size_t qb3x_decode_tiles(decsp p, const void *d_src, size_t n, size_t src_pitch, const size_t *sizes,
                         void *d_dst, size_t dst_pitch, const void *d_index, void *stream);
int qb3x_decode_tile_ok(const decsp p, size_t i);

/* qb3_read_start for the device flavour: `header` is a host copy of the FIRST header_size bytes of a container of
 * stream_size bytes (the parser never reads beyond the copy; qb3_read_info fails when the copy ends before the
 * container's "DT" mark).  qb3x_header_size_bound(first bytes, how many) says how many bytes always suffice,
 * from the container's first 11 bytes (0: not a QB3 container). */
decsp qb3x_read_start(void *header, size_t header_size, size_t stream_size, size_t *image_size);
/* ... and for a container that is in DEVICE memory: qb3_read_start + qb3_read_info in one call; the handle keeps its own
 * copy of the few header bytes it needs (two small device-to-host copies, whatever the size of a restart table: the
 * table's chunk heads and checks are verified on the device before it is used).  Returns a handle ready for
 * qb3x_decode_device / qb3x_decode_tiles, or NULL.  No counterpart in the reference (its containers are in host memory,
 * QB3.h:133-141). */
decsp qb3x_read_start_device(const void *d_container, size_t nbytes, size_t *image_size, void *stream);
size_t qb3x_header_size_bound(const void *container, size_t avail);
/* After qb3_read_info: entries of the restart table found in the container's header chunks that the decoder will use
 * (0: none, or chunks that do not form one table -- the stream is then walked).  No counterpart in the reference. */
size_t qb3x_decoder_table_entries(const decsp p);

/* Self-indexing containers (off by default: the container then differs from the reference's by a few chunks).
 * When on, qb3_encode / qb3x_encode_device put a restart table -- the bit position and band state at the start of every
 * index segment of an FTL/BASE stream (64 blocks of 8-bit grey/RGB/RGBA: 12 bytes, 0.7 % of a typical stream), at about
 * every 64th unit of a common-factor stream (every 32nd for 32/64-bit data; 1.5-6 %) -- into the container in front of "DT", as ignorable (lower-case)
 * chunks: "ix" chunks of at most 64 KB, each followed by a 4-byte pad chunk "zz".
 * An "ix" chunk: 'i' 'x', u16 length of the whole chunk, u8 version (3), u8 flags (bit 0: entries carry common factors),
 * u16 check of the chunk's entries (version 3; reserved and zero in versions 1 and 2, which are still read), u32 blocks per
 * entry, then the entries (flag bit 1: they end with block lengths, see below).  The check: the sum over the n entry bytes
 * b[i] of (b[i] + 1) * (i * 0x9e3779b1 + 1) modulo 2^32, folded to 16 bits (low half XOR high half).  The table sits in a
 * chunk the format does not protect, and the decoder takes positions, rungs, entering values and lengths from it: before
 * using it the decoder verifies every chunk's head and check ON THE DEVICE, and on a mismatch -- or when the decode that
 * relied on the table fails -- decodes the stream WITHOUT the table (the plain walk: what the reference, which skips the
 * chunk, does).  A damaged table costs time, never pixels.  An entry: 6-byte little-endian bit position of its first unit
 * (from the first stream bit), a rung byte per band, the value entering each band (the type's width, little-endian),
 * and with flag bit 0 the common factor entering each band likewise.  Entry k starts at block k * (blocks per entry);
 * every chunk but the last holds the same number of entries.  The reference's decoder steps over them
 * (QB3decode.cpp:251-255) and decodes the same pixels: it skips an unknown chunk by its length field counted from the
 * chunk START, so the field holds the whole chunk size, and the pad makes the container parse the same for a reader
 * that adds the 4 head bytes to it.  This library's decoder uses the table when no out-of-band index is given:
 * qb3_read_data / qb3x_decode_device(d_index = NULL) then walk (FTL/BASE) or decode (common-factor modes) the stream
 * from every entry at once, one lane each, instead of serially.  qb3_max_encoded_size() grows by the table's size while the switch is on.  Not written for
 * narrow images and STORED output; a container whose RLE0 pass wins keeps its table in front of "DT" (the entries describe the
 * block stream, which the decoder has again once it has expanded the bytes).  A decoder handle for a container in device memory: qb3x_read_start_device
 * (two small copies whatever the table's size); or, from a host copy of the container up to its "DT" mark,
 * qb3x_read_start (qb3x_header_size_bound() bytes always suffice).  qb3_max_encoded_size() does not depend on the mode:
 * it is the room of the largest table any mode writes for the raster (callers size their buffer before setting the mode).
 * Callers that only know the reference API (LD_PRELOAD, relinked tools) can set QB3X_INDEX_CHUNK=1 (or 2) in the
 * environment: it is read when an encoder handle is created.
 * on = 2 -- entries with BLOCK LENGTHS: for the rasters the 8-bit lane-per-block decoder takes (uint8, 1/3/4 bands, FTL/BASE,
 * Hilbert or Z order) the "ix" chunks' flag bit 1 is set and every entry (one per 64-block segment) ends with the bit
 * lengths of its segment's blocks, ten bits each, little endian: 80 more bytes, 5.5 % of a typical RGB stream instead of
 * 0.7 %.  The decoder then needs neither a walk nor an index -- one kernel, twice the decode rate from the container alone
 * (16384 x 16384 x 3: 0.37 ms instead of 0.65).  16-bit rasters of 2, 3, 4, 6 or 8 bands (FTL/BASE): an entry (one per 64 lanes of
 * the decoder's wave; a lane owns up to four bands of a block) ends with two fields per lane -- four bands: the bit lengths of the
 * band PAIRS (0,1) and (2,3); three: of (0,1) and of band 2; two: of each band -- ten bits each, 160 bytes -- about 5 % of a typical stream (8192 x 8192 x 8: 0.65 ms instead of 0.94); 16-bit rasters of ONE band:
 * a field per block (its unit's length), 80 bytes an entry.  32/64-bit rasters
 * (FTL/BASE, where the unit-parallel decoder applies): an entry ends with a twelve-bit length per UNIT of its segment
 * (band-minor, little endian) -- about 10 % of a stream of small units (4096 x 4096 int32: 0.06 ms instead of 0.40).
 * Every OTHER FTL/BASE raster (8-bit data of 2 or more than 4 bands, 16-bit data of an odd band count above 4, 32/64-bit data of
 * several bands: the lane-per-unit decoder, a wave per segment of 64 / bands blocks): the same twelve-bit length per unit.
 * Common-factor streams of one band of 16/32/64-bit data: a three-byte field per block (= unit) at either level: its bits (12)
 * | the rung it is entered with << 12.  8-bit common-factor streams of 1, 3 or 4 bands (round 3): at EITHER level an entry
 * per 64-block segment -- position, rungs, entering values, factors in force -- that ends with a three-byte field per
 * block: the block's bits (12) | the rungs its units are entered with (3 bits a band) << 12, little endian; 6 + 3 * bands +
 * 192 bytes an entry, about 12 % of a typical stream.  It is what the lane-per-block decoder of those streams works from
 * (a common-factor unit leaves its band at the rung of the MULTIPLIED values, so rungs cannot be scanned from the switch
 * codes; the factor in force is found by a ballot of the units that bring their own): 16384 x 16384 x 3 in QB3M_BEST
 * decodes from the container alone in 0.47 ms (round 2: 2.4 ms).  Every other common-factor stream (several bands; round 4):
 * at either level an entry per segment of 64 / bands blocks that ends with a three-byte field per UNIT (band-minor): the
 * unit's bits (12) | the rung it is entered with << 12 -- what the lane-per-unit decoder works from (8192 x 8192 x 8 uint16
 * in QB3M_CF_H from the container alone: 1.0 ms; 2.3 before). */
void qb3x_set_encoder_index_chunk(encsp p, int on);

/* Compatibility switches. */
#define QB3X_REF_CBAND0 1u      /* decoder: reproduce reference defect (no CB chunk => every band adds band 0,
                                   reference QB3decode.cpp:138 + QB3decode.h:560-567) instead of identity */
void qb3x_set_decoder_compat(decsp p, unsigned flags);

/* Aliases for the names BASELINE.json uses; the reference has no such symbols. */
decsp  qb3_create_decoder(void *source, size_t source_size, size_t *image_size);  /* read_start + read_info */
size_t qb3_decode(decsp p, void *destination);                                     /* read_data */

/* Per-kernel timing for benchmarks: when enabled, every kernel the library launches is bracketed by HIP
 * events on the launch stream; totals are resolved at the library's own synchronisation points.
 * Kernel names: enc_units, enc_scan, enc_concat, enc_seams, enc_best_units, enc_best_scan, enc_best_recode,
 * dec_index_table, dec_index_serial, dec_index_prev, dec_index_scan, dec_units, dec_segments.
 * level: 0 off, 1 every kernel, 2 all but the microsecond kernels (enc_scan, enc_seams, enc_best_scan), whose two
 * events cost more than they take. */
void qb3x_profile_enable(int level);
void qb3x_profile_reset(void);
int  qb3x_profile_get(const char *kernel, double *total_ms, uint64_t *count);   /* 1 if the kernel was seen */
int  qb3x_profile_names(char *buf, size_t bufsize);                             /* comma separated, returns count */

/* FNV-1a (64 bit) of n host bytes -- the checksum this project's reference anchors are published with (SURVEY.md
 * Appendix C), for callers that verify containers.  seed = 0 starts a hash, a previous result continues it. */
uint64_t qb3x_fnv1a64(const void *data, size_t n, uint64_t seed);

/* The RLE0 byte pass of the *_RLE modes (reference QB3encode.cpp:271-332, QB3decode.cpp:267-307) on DEVICE buffers: the
 * coded (decode = 0) or expanded (decode != 0) form of the n bytes at d_src, bit for bit what the reference's serial
 * loops produce.  d_dst == NULL asks for the size only.  Returns the size; 0 on failure or when it exceeds dst_cap.
 * The library uses it for QB3M_RLE / QB3M_CF_RLE (and the legacy _H modes) so that such streams stay on the device. */
size_t qb3x_rle0_device(const void *d_src, size_t n, void *d_dst, size_t dst_cap, int decode, void *stream);

/* Last HIP error string seen by this thread inside the library ("" if none). */
const char *qb3x_last_error(void);

#if defined(__cplusplus)
}
#endif
#endif
