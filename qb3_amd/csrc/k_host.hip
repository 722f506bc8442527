// qb3_amd/csrc/k_host.hip -- host side of the kernels: plans, workspace layout, launch orchestration, profiling, small elementwise kernels
#include "qb3_kernels.h"
#include "qb3_walk.h"

namespace qb3dev {

// ------------------------------------------------------------------ host side of the kernels
static thread_local char g_err[256] = "";
const char *last_error() { return g_err; }
void set_error(const char *what, int e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, e ? hipGetErrorString((hipError_t)e) : "failed");
}

// debugging switches: read once, the first time any handle plans a launch
const Tuning &tuning() {
    static const Tuning t = [] {
        auto on = [](const char *name) { const char *e = getenv(name); return e && e[0] && e[0] != '0'; };
        Tuning v;
        v.no_px = on("QB3_NO_PX");                   // generic kernels also where a lane-per-block kernel applies
        v.slow_walk = on("QB3_SLOW_WALK");           // plain 8-bit streams: the one-wave walk instead of the table walk
        const char *cap = getenv("QB3_WALK_TAB_KB"); // plain 8-bit streams: bytes of table memory (a small one means many rounds)
        v.walk_tab_kb = cap ? (size_t)strtoull(cap, nullptr, 10) : 0;
        v.slow_index = on("QB3_SLOW_INDEX");         // index-less streams: the one-lane index rebuild instead of the walkers
        { const char *e = getenv("QB3_WIDE_BAND"); v.wide_band = e && e[0] ? atoi(e) : 0; }      // plain 32/64-bit streams: rungs in the table's band (test hook)
        v.no_bl = on("QB3_NO_BLOCK_LENGTHS");        // containers whose table carries block lengths: walk them like the others
        { const char *e = getenv("QB3_BEST_SAMPLE_MIN"); v.best_sample_min = e && e[0] ? (uint32_t)strtoul(e, nullptr, 10) : BEST_SAMPLE_MIN; }   // chunks from which the common-factor encoders sample before they code (0: always -- how the tests reach the two-pass coding on small rasters)
        { const char *e = getenv("QB3_EXITS_FROM"); v.exits_from = e && e[0] ? (int64_t)strtoll(e, nullptr, 10) : -1; }   // stream bits from which rasters of two / three bands walk by exits instead of the chain (-1: the measured crossovers; 0: always -- how the tests reach the exits on small rasters)
        return v;
    }();
    return t;
}

// ---- per-kernel timing: hipEvents recorded on the launch stream, resolved after the caller's sync
}  // namespace qb3dev
#include <map>
#include <mutex>
#include <string>
#include <vector>
namespace qb3dev {
struct ProfPending { const char *name; hipEvent_t a, b; };
static std::mutex g_prof_mu;
static int g_prof_level = 0;
static std::vector<ProfPending> g_prof_pending;
static std::vector<hipEvent_t> g_prof_pool;
static std::map<std::string, std::pair<double, uint64_t>> g_prof_acc;
void prof_enable(int level) { std::lock_guard<std::mutex> l(g_prof_mu); g_prof_level = level; }
// level 2 skips the microsecond kernels: two events per kernel cost more than those kernels take
static bool prof_minor(const char *n) { const std::string s(n); return s == "enc_scan" || s == "enc_seams" || s == "enc_best_scan" || s == "enc_best_recode"; }
void prof_reset() { std::lock_guard<std::mutex> l(g_prof_mu); g_prof_acc.clear(); }
static hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
ProfScope::ProfScope(const char *n, hipStream_t s) : st(s), name(n) {
    std::lock_guard<std::mutex> l(g_prof_mu);
    // (level 3: the coding kernels alone -- "..._units" -- an event pair costs the stream about 15 us, a tenth of the concatenation it would time)
    on = g_prof_level == 1 || (g_prof_level == 2 && !prof_minor(n)) || (g_prof_level >= 3 && std::string(n).find("_units") != std::string::npos);
    if (on) { a = prof_event(); b = prof_event(); (void)hipEventRecord(a, st); }
}
ProfScope::~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(b, st);
    std::lock_guard<std::mutex> l(g_prof_mu);
    g_prof_pending.push_back({name, a, b});
}
// Resolves the event pairs that have completed; a pair still in flight (another thread's stream) stays pending.
void prof_collect() {
    std::lock_guard<std::mutex> l(g_prof_mu);
    std::vector<ProfPending> keep;
    for (auto &p : g_prof_pending) {
        float ms = 0;
        const hipError_t e = hipEventElapsedTime(&ms, p.a, p.b);
        if (e == hipErrorNotReady) { keep.push_back(p); continue; }
        if (e == hipSuccess) { auto &acc = g_prof_acc[p.name]; acc.first += ms; acc.second++; }
        g_prof_pool.push_back(p.a); g_prof_pool.push_back(p.b);
    }
    (void)hipGetLastError();       // hipErrorNotReady is not an error of the caller's
    g_prof_pending.swap(keep);
}
bool prof_get(const char *name, double *total_ms, uint64_t *count) {
    std::lock_guard<std::mutex> l(g_prof_mu);
    auto it = g_prof_acc.find(name);
    if (it == g_prof_acc.end()) return false;
    *total_ms = it->second.first; *count = it->second.second;
    return true;
}
int prof_names(char *buf, size_t n) {
    std::lock_guard<std::mutex> l(g_prof_mu);
    std::string s;
    for (auto &kv : g_prof_acc) { if (!s.empty()) s += ","; s += kv.first; }
    snprintf(buf, n, "%s", s.c_str());
    return (int)g_prof_acc.size();
}


// Decoder workgroup geometry of the unit-parallel kernel: threads, blocks per pass, passes
static void fast_geometry(uint32_t bands, uint32_t tsz, uint32_t *threads, uint32_t *bpp, uint32_t *passes) {
    *threads = tsz == 8 ? 128 : 256;
    *bpp = *threads / bands;
    uint32_t k = (uint32_t)(16384 / ((size_t)*bpp * bands * tsz * 16));    // keep the pixel tile near 16 KB
    *passes = k < 1 ? 1 : (k > 3 ? 3 : k);
}
// Blocks per index segment.  A function of stream-intrinsic properties only (bands, value size, mode): encoder
// and decoder must agree on it whatever their strides, band maps or buffer alignments are.
// 16-bit lane-per-block kernels: bands = ng x bg, bg <= 4 bands per lane (0: no such split)
// 16-bit lane-per-block kernels: bands = ng x bg, bg <= 4 bands per lane (0: no such split).  (Measured, round 3: two bands a
// lane for rasters of four and eight bands -- 65-99 registers instead of 124, 5-7 waves a SIMD instead of 4 -- made
// config 3 SLOWER: decode 0.72 ms against 0.58, encode 0.43 against 0.41: twice the lanes pay the per-lane overheads
// -- index loads, scans, the store address arithmetic -- and the stores turn from 32-byte pieces into 4-byte ones.)
static void px16_split(const Geometry &g, uint32_t *bg, uint32_t *ng) {
    const uint32_t B = g.bands;
    *bg = B <= 4 ? B : (B % 4 == 0) ? 4 : (B % 2 == 0) ? 2 : 0;
    *ng = *bg ? B / *bg : 0;
}
uint32_t px16_bands_per_lane(const Geometry &g) { uint32_t bg, ng; px16_split(g, &bg, &ng); return bg; }

uint32_t seg_blocks_for(const Geometry &g) {
    // rasters of the lane-per-unit kernels (k_dec_pxu.hip): a wave owns a segment, a lane a unit
    if (lane_per_unit_shape(g.tsz, g.mode, g.bands)) return 64 / g.bands;
    if (g.mode != CM_BEST) {      // one segment = the blocks of one decoder workgroup, at most 256
        // 8-bit grey/RGB/RGBA: the lane-per-block decoder gives a segment to a WAVE (a function of type and band
        // count only: encoder and decoder must agree whatever kernel either of them ends up using)
        if (g.tsz == 1 && (g.bands == 1 || g.bands == 3 || g.bands == 4)) return 64;
        if (g.tsz == 2) {       // 16-bit: a wave = 64 lanes of (block, band group)
            uint32_t bg, ng;
            px16_split(g, &bg, &ng);
            if (bg) return 64 / ng;
        }
        if (g.tsz >= 4 && g.bands == 1) return 64;       // 32/64-bit, one band: the lane-per-block decoder gives a segment to a wave
        uint32_t threads, bpp, passes;
        fast_geometry(g.bands, g.tsz, &threads, &bpp, &passes);
        while (passes > 1 && bpp * passes > 256) passes--;
        return bpp * passes;
    }
    // common-factor modes, 8-bit grey/RGB/RGBA: a wave of the lane-per-block decoder owns a segment (the index has a dword per block)
    if (best_block_table(g.tsz, g.mode, g.bands)) return 64;
    // common-factor modes: a lane walks the segment serially, keep it short (12 units)
    uint32_t s = 12 / g.bands;
    return s ? s : 1;
}
// fields of IX_BL_BITS bits behind an entry's fixed part: a block length per block of the segment (8-bit data) or two
// band-pair lengths per lane of the decoder's wave (16-bit data, four bands a lane: 64 lanes)
// ... or a unit length per unit of the segment (32/64-bit data: the unit-parallel decoder)
uint32_t ix_bl_fields(const Geometry &g) { return lane_per_unit_shape(g.tsz, g.mode, g.bands) ? g.seg_blocks * g.bands : g.tsz == 1 ? g.seg_blocks : g.tsz == 2 ? (g.bands == 1 ? 64 : 128) : g.seg_blocks * g.bands; }
uint32_t ix_entry_bytes(const Geometry &g, bool block_lens) {
    return 6 + g.bands * (1 + g.tsz * (g.mode == CM_BEST ? 2 : 1)) + (block_lens ? ix_bl_bytes(g.tsz, g.bands, g.seg_blocks, g.mode == CM_BEST) : 0);
}
// Block lengths: for the rasters the 8-bit lane-per-block decoder takes (a block of at most four units of at most 149 bits
// fits ten bits), an entry per 64-block segment
bool ix_block_lens_ok(const Geometry &g) {
    // rasters of the lane-per-unit kernels: a field per unit of the segment (common factor: bits | entering rung; else twelve bits of length)
    if (lane_per_unit_shape(g.tsz, g.mode, g.bands)) return g.seg_blocks == 64 / g.bands;
    // 8-bit common-factor streams of 1/3/4 bands: a field per block of the 64-block segment (its bits and entering rungs),
    // what the lane-per-block decoder needs besides the entry's fixed part; without it a lane would walk 64 blocks
    if (g.mode == CM_BEST) return best_block_table(g.tsz, g.mode, g.bands) && g.seg_blocks == 64;
    // 32/64-bit data: where the unit-parallel decoder applies (its workgroup's tile fits LDS), a twelve-bit length per unit
    if (g.tsz >= 4) return g.seg_blocks != 0 && plan_decode(g).fast && 6 + g.bands * (1 + g.tsz) + (g.seg_blocks * g.bands * IX_BL_BITS_WIDE + 7) / 8 <= 32768;
    if (!(g.order == HILBERT || g.order == ZCURVE)) return false;
    if (g.tsz == 1) return (g.bands == 1 || g.bands == 3 || g.bands == 4) && g.seg_blocks == 64;
    // 16-bit, four or eight bands: a lane of the decoder owns four bands = two pairs (two units of at most 278 bits fit ten bits)
    // 16-bit data the lane-per-block decoder takes, up to eight bands: a lane owns up to four bands (a single band: one unit, one field)
    if (g.tsz != 2 || g.bands > 8) return false;
    uint32_t bg = 0, ng = 0;
    px16_split(g, &bg, &ng);
    return bg != 0 && g.seg_blocks == 64 / ng;
}
// One entry per index segment for FTL/BASE streams (a lane then walks one segment, lengths only, and the segment's
// entering values come straight from its entry), one per about 64 units for the common-factor modes.  For 8-bit RGB
// that is 12 bytes per 64 blocks: 0.7 % of a typical stream.
IxTable ix_layout(const Geometry &g, int level) {
    IxTable t;
    if (!g.seg_blocks || !g.nseg) return t;
    t.block_lens = (level >= 2 || best_block_table(g.tsz, g.mode, g.bands) || (g.mode == CM_BEST && lane_per_unit_shape(g.tsz, g.mode, g.bands))) && ix_block_lens_ok(g);
    t.entry_bytes = ix_entry_bytes(g, t.block_lens);
    const uint64_t units_per_seg = (uint64_t)g.seg_blocks * g.bands;
    const bool per_seg = g.mode != CM_BEST;
    // (common-factor streams: the lane that starts at an entry parses whole units from global memory, bound by latency --
    // 64 units an entry keeps four times the lanes in flight that 256 did, for 1.5-3 % of the stream)
    // (... and 32 for 32/64-bit data, whose images have fewer units for the same bytes: a 4096 x 4096 band is 1 M units)
    // (level 2, common-factor streams: the entries closer together -- 24 units, 12 for 32/64-bit data: two to three times the lanes,
    // each with a piece as much shorter, for 10-20 % of the stream)
    const uint64_t target = level >= 2 ? (g.tsz >= 4 ? 12 : 24) : (g.tsz >= 4 ? 32 : 64);
    const uint64_t spe = (per_seg || units_per_seg >= target || t.block_lens) ? 1 : target / units_per_seg; // index segments per entry
    t.blocks = (uint32_t)(spe * g.seg_blocks);
    t.K = (uint32_t)((g.nseg + spe - 1) / spe);
    t.per_chunk = (65535 - IX_HEAD) / t.entry_bytes;
    return t;
}
uint32_t ulen_size_for(uint32_t tsz, uint32_t mode, uint32_t bands) { return mode == CM_BEST ? (best_block_table(tsz, mode, bands) ? 4 : ULEN_UNIT) : (tsz == 1 ? 1 : 2); }

static size_t align8(size_t v) { return (v + 7) & ~(size_t)7; }
size_t index_bytes(const Geometry &g) {
    const size_t n = (size_t)g.nseg * g.bands;
    return align8(8 * (size_t)g.nseg) + 2 * align8(n * g.tsz) + align8(n) + align8(ulen_table_bytes(g));
}
IndexView index_view(const Geometry &g, void *base) {
    IndexView v;
    uint8_t *p = (uint8_t *)base;
    const size_t n = (size_t)g.nseg * g.bands;
    v.bitpos = (uint64_t *)p; p += align8(8 * (size_t)g.nseg);
    v.prev = p; p += align8(n * g.tsz);
    v.cf = p; p += align8(n * g.tsz);
    v.rung = p; p += align8(n);
    v.ulen = g.ulen_sz ? p : nullptr;
    return v;
}

uint32_t magic_div(uint32_t d) { return d == 1 ? 0u : (uint32_t)(((1ull << 32) + d - 1) / d); }   // exact for n*d < 2^32/d-ish, n small

uint32_t max_unit_bits(uint32_t tsz, uint32_t mode) {
    const uint32_t ub = tsz == 1 ? 3 : tsz == 2 ? 4 : tsz == 4 ? 5 : 6;
    const uint32_t plain = ub + 2 + 16 * (8 * tsz + 1);
    // common factor: signal + switch + 2 flags + own-rung switch + factor code + group
    return mode == CM_BEST ? plain + 3 * ub + 8 + 8 * tsz + 2 : plain;
}

// encoder workspace layout (all 8-byte aligned), EncResult last
struct EncWs { size_t bits, off, gsum, seams, scratch, cwhas, cwval, centry, cparts, cwused, segfe, rneed, rlist, res, total; uint32_t slot_dw, ngroups; };
static EncWs enc_ws_layout(const Geometry &g, uint32_t nchunks, uint32_t nbp, uint32_t threads) {
    EncWs w;
    // a multiple of 4 dwords: slots are 16-byte aligned (the px kernel copies them out as uint4)
    w.slot_dw = (uint32_t)(((31 + (size_t)nbp * g.bands * max_unit_bits(g.tsz, g.mode)) / 32 + 1 + 3) & ~(size_t)3);
    w.ngroups = (nchunks + SCAN_GROUP - 1) / SCAN_GROUP;
    size_t o = 0;
    w.bits = o; o += align8(4 * (size_t)nchunks);
    w.off = o; o += 8 * (size_t)nchunks;
    w.gsum = o; o += 8 * ((size_t)w.ngroups + 2);          // (+ the total, + enc_scan_kernel's count of finished workgroups)
    w.seams = o; o += 8 * (size_t)nchunks;
    o = (o + 15) & ~(size_t)15;
    w.scratch = o; o += align8(4 * (size_t)nchunks * w.slot_dw);
    const size_t nb = g.mode == CM_BEST ? (size_t)nchunks * g.bands : 0;
    w.cwhas = o; o += align8(nb);
    w.cwval = o; o += 8 * nb;
    w.centry = o; o += 8 * nb;
    w.cparts = o; o += g.mode == CM_BEST ? 4 * (32 * (size_t)MAXBANDS + 2) : 0;       // (+ the recode counter)
    w.cwused = o; o += align8(nb);
    w.segfe = o; o += g.mode == CM_BEST ? align8((size_t)g.nseg * g.bands) : 0;
    w.rneed = o; o += g.mode == CM_BEST ? align8(4 * (size_t)nchunks) : 0;
    w.rlist = o; o += g.mode == CM_BEST ? align8(4 * (size_t)nchunks) : 0;
    w.res = o; o += sizeof(EncResult);
    w.total = o;
    return w;
}

// the 8-bit lane-per-block kernels need: uint8, 1/3/4 bands, rows of whole blocks at dword-aligned addresses,
// Hilbert or Z curve, identity or default RGB(A) band map
static bool px_eligible(const Geometry &g, bool *rgb) {
    if (g.tsz != 1 || !(g.bands == 1 || g.bands == 3 || g.bands == 4)) return false;
    if (g.w < 4 || g.h < 4) return false;                  // any width, stride and pointer: rows are read and written unaligned
    if (g.order != HILBERT && g.order != ZCURVE) return false;
    bool ident = true, def = g.bands >= 3;
    for (uint32_t c = 0; c < g.bands; c++) {
        ident = ident && g.cband[c] == c;
        def = def && g.cband[c] == ((c == 0 || c == 2) ? 1u : c);
    }
    *rgb = def && !ident;
    return (ident || def) && !tuning().no_px;
}

// 16-bit lane-per-(block, band group) kernels: bands = NG x BG with BG <= 4; a group must be whole dwords per
// pixel unless it is the whole pixel (BG = bands = 1 or 3)
static bool px16_eligible(const Geometry &g, bool *rgb, uint32_t *bg, uint32_t *ng) {
    if (g.tsz != 2 || g.mode == CM_BEST || tuning().no_px) return false;
    if (g.w < 4 || g.h < 4) return false;                  // any width and stride: rows are read and written at halfword alignment
    if (g.order != HILBERT && g.order != ZCURVE) return false;
    const uint32_t B = g.bands;
    px16_split(g, bg, ng);
    if (!*bg) return false;
    // identity, or R-G, G, B-G on the first three bands (they must sit in one lane: 3 or 4 bands per group)
    bool ident = true, def = *bg >= 3;
    for (uint32_t c = 0; c < B; c++) {
        ident = ident && g.cband[c] == c;
        def = def && g.cband[c] == ((c == 0 || c == 2) ? 1u : c);
    }
    *rgb = def && !ident;
    return ident || def;
}

// the 32/64-bit lane-per-block kernels: one band (a block is a unit), FTL / BASE, Hilbert or Z curve; any width, stride and
// (value-aligned) pointer
// (the common-factor kernels also take 16-bit rasters of one band -- int16 elevation: their FTL / BASE streams have the 16-bit kernels)
static bool pxw_eligible(const Geometry &g, bool best = false) {
    return (g.tsz >= 4 || (best && g.tsz == 2)) && g.bands == 1 && (g.mode == CM_BEST) == best && g.w >= 4 && g.h >= 4 && (g.order == HILBERT || g.order == ZCURVE) && !tuning().no_px;
}

EncPlan plan_encode(const Geometry &g) {
    EncPlan p;
    const uint32_t dpr = g.bands * g.tsz;
    p.px = px_eligible(g, &p.px_rgb);
    p.px16 = false; p.px16_bg = p.px16_ng = 0;
    p.pxw = false;
    p.pxw_best = !p.px && pxw_eligible(g, true);        // (same chunks as the generic plan below: one band, slots = threads)
    if (!p.px && pxw_eligible(g)) {
        p.pxw = true;
        p.threads = g.tsz == 8 ? 128 : 256; p.slots = p.threads; p.nbp = p.threads - 1;
        p.nchunks = (uint32_t)((g.nblocks + p.nbp - 1) / p.nbp);
        const EncWs L = enc_ws_layout(g, p.nchunks, p.nbp, p.threads);
        p.lds_bytes = 2048 + 64 + 4 * (size_t)L.slot_dw;
        p.ws_bytes = L.total;
        return p;
    }
    if (p.px) {
        p.threads = 256; p.slots = 256; p.nbp = 255;
        p.nchunks = (uint32_t)((g.nblocks + 254) / 255);
        const EncWs L = enc_ws_layout(g, p.nchunks, p.nbp, p.threads);
        p.lds_bytes = (g.mode == CM_BEST ? PXB_LDS_FIXED : 2048 + 256) + 4 * (size_t)L.slot_dw;
        p.ws_bytes = L.total;
        return p;
    }
    bool rgb16 = false;
    if (px16_eligible(g, &rgb16, &p.px16_bg, &p.px16_ng)) {
        p.px16 = true; p.px_rgb = rgb16;
        p.threads = 256; p.slots = 256 / p.px16_ng; p.nbp = p.slots - 1;
        p.nchunks = (uint32_t)((g.nblocks + p.nbp - 1) / p.nbp);
        const EncWs L = enc_ws_layout(g, p.nchunks, p.nbp, p.threads);
        p.lds_bytes = 2048 + 256 + 1024 + 4 * (size_t)L.slot_dw;
        p.ws_bytes = L.total;
        return p;
    }
    p.threads = g.tsz == 8 ? 128 : 256;
    p.slots = p.threads / g.bands;
    const uint32_t nbp = p.slots - 1;
    p.nbp = nbp;
    p.nchunks = (uint32_t)((g.nblocks + nbp - 1) / nbp);
    // (the pixel tile and the bit buffer -- of slot_dw dwords, what the common-factor kernels zero -- share their memory: enc_front, qb3_enc_front.h)
    const size_t outdw = enc_ws_layout(g, p.nchunks, nbp, p.threads).slot_dw, tiledw = 4 * (size_t)p.slots * dpr;
    p.lds_bytes = 8 * (size_t)p.slots + 4 * std::max(tiledw, (outdw + 1) & ~(size_t)1) + 256 + 1024 + (((size_t)p.slots * g.bands + 7) & ~(size_t)7);
    if (g.mode == CM_BEST) p.lds_bytes += 8 * (size_t)p.threads + 8 * 16 + 4 * MAXBANDS + 8 + 4 * (size_t)p.slots + 4 * 512;     // the writer board: a value per lane, a ballot per wave, a word per band; a word per block (the index's block table)
    // (the lane-per-block front end has no tile: scan scratch, the code table, the bit buffer -- of the worst common-factor unit -- and the board)
    if (p.pxw_best) p.lds_bytes = 256 + 1024 + 4 * (size_t)enc_ws_layout(g, p.nchunks, nbp, p.threads).slot_dw + 8 + 8 * (size_t)p.threads + 8 * 16 + 4 * MAXBANDS + 8 + 4 * (size_t)p.slots + 4 * 512;
    p.ws_bytes = enc_ws_layout(g, p.nchunks, nbp, p.threads).total;
    return p;
}

static void launch_enc_units(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    if (a.g.mode == CM_BEST) { if (plan.px && a.g.tsz == 1) launch_enc_px_best(a, plan, st); else launch_enc_best(a, plan, st); }
    else if (plan.px && a.g.tsz == 1) {
        ProfScope ps("enc_units", st);
        launch_enc_px(a, plan, st);
    }
    else if (plan.px16 && a.g.tsz == 2 && ((uintptr_t)a.img & 1) == 0) { ProfScope ps("enc_units", st); launch_enc_px16(a, plan, st); }
    else if (plan.pxw && ((uintptr_t)a.img & (a.g.tsz - 1)) == 0 && !(a.ts_img & (a.g.tsz - 1))) { ProfScope ps("enc_units", st); launch_enc_pxw(a, plan, st); }
    else { ProfScope ps("enc_units", st); launch_enc_generic(a, plan, st); }
}
static int launch_encode_all(const EncArgs &a, const EncPlan &plan, hipStream_t st) {
    launch_enc_units(a, plan, st);
    launch_enc_post(a, plan, st);
    HIPCHK(hipGetLastError());
    return 0;
}

// the argument block of the encoder kernels (workspace carve, index view, table layout)
static bool make_enc_args(EncArgs &a, const Geometry &g, const EncPlan &plan, const void *img, uint32_t *out32, uint32_t out_bit0,
                          const BandState &st_in, void *ws, void *index, const TileBatch &tb,
                          const uint8_t *hdr, uint32_t hdr_len, const IxTable &ix, bool zrun_probe) {
    a.zrun_probe = zrun_probe ? 1u : 0u;
    a.ix_dst = ix.base; a.ix_K = ix.K; a.ix_E = ix.entry_bytes; a.ix_per_chunk = ix.per_chunk; a.ix_blocks = ix.blocks;
    a.ix_spe = g.seg_blocks ? ix.blocks / g.seg_blocks : 0;
    a.ntiles = tb.n ? tb.n : 1; a.ts_img = tb.src_pitch; a.ts_out = tb.dst_pitch; a.ts_ws = tb.ws_pitch; a.ts_idx = tb.idx_pitch;
    a.hdr_len = hdr_len <= sizeof(a.hdr) ? hdr_len : 0;
    a.hdr_back = a.hdr_len + (ix.base ? (uint32_t)ix_total_bytes(ix) + 2 : 0);
    for (uint32_t i = 0; i < a.hdr_len; i++) a.hdr[i] = hdr[i];
    a.g = g; a.img = img; a.out32 = out32; a.out_bit0 = out_bit0;
    a.slots = plan.slots; a.nchunks = plan.nchunks; a.dpr = g.bands * g.tsz;
    a.chunk0 = 0; a.chunk_end = plan.nchunks; a.finish_what = 3;
    a.magic_dpr = magic_div(a.dpr); a.magic_bands = magic_div(g.bands);
    uint8_t *w = (uint8_t *)ws;
    const EncWs L = enc_ws_layout(g, plan.nchunks, plan.nbp, plan.threads);
    a.chunk_bits = (uint32_t *)(w + L.bits);
    a.chunk_off = (uint64_t *)(w + L.off);
    a.group_sum = (uint64_t *)(w + L.gsum);
    a.seams = (uint32_t *)(w + L.seams);
    a.scratch = (uint32_t *)(w + L.scratch);
    a.cw_has = w + L.cwhas; a.cw_val = (uint64_t *)(w + L.cwval); a.centry = (uint64_t *)(w + L.centry); a.centry_parts = (uint32_t *)(w + L.cparts); a.recode_n = a.centry_parts + 32 * MAXBANDS;
    a.cw_used = w + L.cwused; a.seg_from_entry = w + L.segfe; a.recode_need = (uint32_t *)(w + L.rneed); a.recode_list = (uint32_t *)(w + L.rlist);
    a.slot_dw = L.slot_dw;
    a.px_ng = plan.px16 ? plan.px16_ng : 1; a.px_magic_ng = magic_div(a.px_ng);
    a.px16_bg = g.tsz == 2 ? px16_bands_per_lane(g) : 0;
    a.px_aligned = !(g.w & 3) && !((g.stride * g.tsz) & 3) && !((uintptr_t)img & 3) && !(tb.src_pitch & 3);
    a.res = (EncResult *)(w + L.res);
    a.st = st_in;
    a.have_idx = index != nullptr;
    a.idx_no_ulen = index && ix.base && ix.own_index && !ix.block_lens;      // (block lengths are sums of the unit lengths)
    a.ix_bl = ix.base && ix.block_lens;
    a.idx = index ? index_view(g, index) : IndexView{nullptr, nullptr, nullptr, nullptr, nullptr};
    if (g.tsz != 1 && g.tsz != 2 && g.tsz != 4 && g.tsz != 8) { set_error("encode: bad value size", 0); return false; }
    return true;
}

int launch_encode(const Geometry &g, const EncPlan &plan, const void *img, uint32_t *out32, uint32_t out_bit0,
                  const BandState &st_in, void *ws, void *index, void *stream, const TileBatch &tb,
                  const uint8_t *hdr, uint32_t hdr_len, const IxTable &ix, bool zrun_probe) {
    EncArgs a;
    if (!make_enc_args(a, g, plan, img, out32, out_bit0, st_in, ws, index, tb, hdr, hdr_len, ix, zrun_probe)) return -1;
    return launch_encode_all(a, plan, (hipStream_t)stream);
}

// ---- a pipelined host call codes the raster strip by strip: a strip is a scan group of chunks (SCAN_GROUP of them), coded,
// scanned, moved into place and sealed by launches of its own, so that its part of the stream is final -- and can go down the
// link -- while later strips are still on their way up.  FTL / BASE only (the common-factor modes carry a factor across chunks).
bool encode_strips_ok(const Geometry &g, const EncPlan &plan) { return g.mode != CM_BEST && plan.nchunks > 2 * SCAN_GROUP; }
uint32_t encode_strip_count(const EncPlan &plan) { return (plan.nchunks + SCAN_GROUP - 1) / SCAN_GROUP; }
uint64_t encode_strip_blocks(const EncPlan &plan, uint32_t strip) {         // blocks (from the raster's first) the strip and those before it hold
    const uint64_t c = std::min<uint64_t>(plan.nchunks, ((uint64_t)strip + 1) * SCAN_GROUP);
    return c * plan.nbp;
}
const uint64_t *encode_strip_total(const Geometry &g, const EncPlan &plan, void *ws, uint32_t strip) {     // device address of "stream bits behind this strip"
    const EncWs L = enc_ws_layout(g, plan.nchunks, plan.nbp, plan.threads);
    return (const uint64_t *)((uint8_t *)ws + L.gsum) + strip + 1;
}
int launch_encode_strip(const Geometry &g, const EncPlan &plan, const void *img, uint32_t *out32, uint32_t out_bit0,
                        const BandState &st_in, void *ws, void *index, void *stream, const uint8_t *hdr, uint32_t hdr_len, const IxTable &ix, uint32_t strip) {
    EncArgs a;
    if (!make_enc_args(a, g, plan, img, out32, out_bit0, st_in, ws, index, TileBatch(), hdr, hdr_len, ix, false)) return -1;
    a.chunk0 = strip * SCAN_GROUP;
    a.chunk_end = std::min<uint32_t>(plan.nchunks, a.chunk0 + SCAN_GROUP);
    a.finish_what = 1;
    hipStream_t st = (hipStream_t)stream;
    launch_enc_units(a, plan, st);
    launch_enc_post_strip(a, plan, st, strip);
    HIPCHK(hipGetLastError());
    return 0;
}
int launch_encode_tail(const Geometry &g, const EncPlan &plan, const void *img, uint32_t *out32, uint32_t out_bit0,
                       const BandState &st_in, void *ws, void *index, void *stream, const uint8_t *hdr, uint32_t hdr_len, const IxTable &ix) {
    EncArgs a;
    if (!make_enc_args(a, g, plan, img, out32, out_bit0, st_in, ws, index, TileBatch(), hdr, hdr_len, ix, false)) return -1;
    a.finish_what = 2;
    launch_enc_post_tail(a, plan, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return 0;
}

// LDS dwords per decoder lane: 16*bands values + 2*bands state values + bands rung bytes.  The count is made
// odd (conflict-free lane stride) for <= 4 byte values; 8-byte values need an 8-byte aligned lane base, so
// there it is made 2 mod 4.
static uint32_t dec_lane_dwords(const Geometry &g) {
    uint32_t dw = (uint32_t)((16 * g.bands * g.tsz + 2 * g.bands * g.tsz + g.bands + 3) / 4);
    if (g.tsz == 8) { dw = (dw + 1) & ~1u; if ((dw & 3) == 0) dw += 2; }
    else dw |= 1;
    return dw;
}

DecPlan plan_decode(const Geometry &g) {
    DecPlan p;
    // per lane: 16*bands values + 2*bands state values + bands rung bytes, rounded to an odd dword count
    const uint32_t lane_dw = dec_lane_dwords(g);
    uint32_t threads = 64;
    while (threads > 1 && (size_t)threads * lane_dw * 4 > 48 * 1024) threads >>= 1;
    p.threads = threads;
    p.nwg = (uint32_t)((g.nseg + threads - 1) / threads);
    p.lds_bytes = (size_t)threads * lane_dw * 4;
    p.ws_bytes = align8(index_bytes(g)) + 64;          // per tile: rebuilt index + its share of the status words
    // unit-parallel kernel: FTL/BASE with a per-unit length table, and every core band must itself be core
    // (true for every map the encoder's setter can produce, reference QB3encode.cpp:70-72); anything else keeps
    // the lane-per-segment kernel
    bool simple = g.mode != CM_BEST && g.ulen_sz != 0;
    for (uint32_t c = 0; c < g.bands; c++) simple = simple && g.cband[g.cband[c]] == g.cband[c];
    fast_geometry(g.bands, g.tsz, &p.threads2, &p.bpp, &p.passes);
    const uint32_t dpr = g.bands * g.tsz, NB = g.seg_blocks;
    p.passes = (NB + p.bpp - 1) / p.bpp;
    p.in_cap_dw = (NB * dpr * 4 + 8 + 1) & ~1u;         // room for a stream as large as the raw blocks
    p.lds2_bytes = 8 * (size_t)NB + 8 * 16 + 8 * 2 * MAXBANDS + 4 * 2 * MAXBANDS + 4 * (size_t)((p.bpp + 1) & ~1u)
                 + 4 * (size_t)p.in_cap_dw + 16 * (size_t)NB * dpr + align8(2 * (size_t)p.bpp * g.bands) + 2048;
    p.fast = simple && p.lds2_bytes <= 64 * 1024;
    // 8-bit lane-per-block kernel
    bool rgb = false;
    p.px = p.fast && g.mode != CM_BEST && px_eligible(g, &rgb);
    p.px_rgb = rgb;
    // staging of the px kernel: the longest valid segment (every unit at its maximum) + the word the first unit
    // starts in + 8 zero words, after the 4 KB table, the scan scratch and the unit lengths
    p.px_cap_dw = (uint32_t)(((size_t)NB * g.bands * max_unit_bits(g.tsz, g.mode) + 31) / 32 + 2);
    p.px = p.px && NB <= 64;
    p.lds_px = 4096 + 4 * 4 * ((size_t)p.px_cap_dw + 8);       // table + four waves' staging
    // ... and its common-factor counterpart (the index has a dword per block: ulen_sz == 4)
    p.px_best = false;
    if (g.mode == CM_BEST && g.ulen_sz == 4 && NB == 64 && px_eligible(g, &rgb)) { p.px_best = true; p.px_rgb = rgb; }
    p.px16 = false; p.px16_bg = p.px16_ng = 0;
    bool rgb16 = false;
    if (!p.px && p.fast && px16_eligible(g, &rgb16, &p.px16_bg, &p.px16_ng) && NB * p.px16_ng <= 64) {
        p.px16 = true; p.px_rgb = rgb16;
        p.px_cap_dw = (p.px_cap_dw + 4 + 3) & ~3u;             // staged from a 16-byte aligned word, in 16-byte pieces
        p.lds_px = 4096 + 4 * 4 * ((size_t)p.px_cap_dw + 16);
    }
    // 32/64-bit, one band: a wave per 64-block segment; staging for the longest valid segment + WIDE_PAD_DW zero words, behind the 2 KB table
    p.pxw = !p.px && !p.px16 && p.fast && pxw_eligible(g) && NB == 64;
    p.lds_pxw = 0;
    if (p.pxw) {
        p.px_cap_dw = (uint32_t)(((size_t)NB * max_unit_bits(g.tsz, g.mode) + 31) / 32 + 2);
        p.lds_pxw = 2048 + 4 * 4 * ((size_t)p.px_cap_dw + WIDE_PAD_DW);
    }
    // ... and the common-factor streams of such rasters (the index has a dword per block: ulen_sz == 4); no table, no barrier
    p.pxw_best = g.ulen_sz == 4 && NB == 64 && pxw_eligible(g, true);
    if (p.pxw_best) {
        p.px_cap_dw = (uint32_t)(((size_t)NB * max_unit_bits(g.tsz, g.mode) + 31) / 32 + 2);
        p.lds_pxw = 4 * 4 * ((size_t)p.px_cap_dw + WIDE_PAD_DW);
    }
    // every other raster: a wave per segment of 64 / bands blocks, a lane per unit (k_dec_pxu.hip); core bands must themselves be core
    bool core_ok = true;
    for (uint32_t c = 0; c < g.bands; c++) core_ok = core_ok && g.cband[c] < g.bands && g.cband[g.cband[c]] == g.cband[c];
    const bool pxu_shape = lane_per_unit_shape(g.tsz, g.mode, g.bands) && NB == 64 / g.bands && core_ok && g.w >= 4 && g.h >= 4 && !tuning().no_px;
    p.pxu = pxu_shape && g.mode != CM_BEST && g.ulen_sz != 0 && !p.px && !p.px16 && !p.pxw;
    p.pxu_best = pxu_shape && g.mode == CM_BEST && g.ulen_sz == ULEN_UNIT;
    if (p.pxu || p.pxu_best) p.px_cap_dw = (uint32_t)(((size_t)NB * g.bands * max_unit_bits(g.tsz, g.mode) + 31) / 32 + 2);
    return p;
}

// plain common-factor streams of several bands that walk by the chain (k_dec_walk_chain.hip, walk_chainN_kernel<UB, true>): 8- and 16-bit
// rasters of the lane-per-unit decoder, and 8-bit RGBA (grey and RGB go by exits)
static bool best_chain_applies(const Geometry &g, const DecPlan &plan) {
    return g.mode == CM_BEST && g.tsz <= 2 && g.bands >= 2 && (plan.pxu_best || (plan.px_best && g.tsz == 1 && g.bands == 4));
}
size_t walk_table_cap() { return tuning().walk_tab_kb ? tuning().walk_tab_kb << 10 : (size_t)1 << 30; }
bool walk_table_applies(const Geometry &g, const DecPlan &plan) {
    // 8- and 16-bit rasters the lane-per-block decoders take; 32/64-bit rasters the unit-parallel decoder takes (a band of sixteen rungs)
    // ... and single-band common-factor streams of any width (the exits of k_dec_walk.hip)
    // ... common-factor streams of several bands of 8- and 16-bit data: the chain with the signal units parsed by the walking lane
    if (g.mode == CM_BEST) return (g.bands == 1 || (g.bands == 3 && g.tsz == 1) || best_chain_applies(g, plan)) && !tuning().slow_walk && !tuning().slow_index;
    // (8- and 16-bit rasters of the lane-per-unit decoder too: the 16-bit chain's kernels take any band count and segment size)
    return (((plan.px || plan.pxu) && g.tsz == 1) || ((plan.px16 || plan.pxu) && g.tsz == 2) || (g.tsz >= 4 && plan.fast)) && !tuning().slow_walk && !tuning().slow_index;
}

// A restart table is untrusted input that the decoder takes positions, rungs and values from: its chunks are checked (ix_check_chunk,
// qb3_kernels.h) -- by this kernel in front of the decoder, or by workgroups of the decoder's own launch (dec_px_kernel, DecArgs::chk_wgs).
// A mismatch raises status bit 5; the host then decodes the call again without the table.  A workgroup per chunk.
__global__ void __launch_bounds__(256) ix_check_kernel(const DecArgs a0) {
    const DecArgs a = dec_for_tile(a0, blockIdx.y);
    __shared__ uint32_t part[4];
    ix_check_chunk(a, blockIdx.x, part);
}

static int launch_decode_all(const DecArgs &a, const DecPlan &plan, bool rebuild, hipStream_t st, void *walk_tab, size_t walk_tab_bytes, uint64_t max_bits) {
    const bool best = a.g.mode == CM_BEST;
    const bool use_px = plan.px && !best && a.g.tsz == 1;
    const bool need_check = rebuild && a.ix && a.ix_K && (a.ix_ver >= 3 || a.ix_check_heads);
    const bool use_px16 = plan.px16 && !best && a.g.tsz == 2 && ((uintptr_t)a.img & 1) == 0;
    // 32/64-bit FTL/BASE streams that bring a restart table with an entry per index segment: the lengths-only walk too
    const bool wide_walk = rebuild && a.ix && !best && a.g.tsz >= 4 && plan.fast && a.ix_blocks == a.g.seg_blocks && a.g.ulen_sz == 2;
    const bool unit_parallel = !use_px && !use_px16 && plan.fast && !best && a.g.tsz >= 4;
    // ... of them, one band: the lane-per-block decoder (a wave per segment) instead of the unit-parallel workgroup
    const bool use_pxw = unit_parallel && plan.pxw && ((uintptr_t)a.img & (a.g.tsz - 1)) == 0 && !(a.ts_img & (a.g.tsz - 1));
    // every raster no lane-per-block kernel takes: a lane per unit (value-aligned pointers)
    const bool val_aligned = ((uintptr_t)a.img & (a.g.tsz - 1)) == 0 && !(a.ts_img & (a.g.tsz - 1));
    const bool use_pxu = !use_px && !use_px16 && !use_pxw && plan.pxu && plan.fast && !best && val_aligned;
    auto dec_units = [&](const DecArgs &t) {
        if (use_px) launch_dec_px(t, plan, st); else if (use_px16) launch_dec_px16(t, plan, st); else if (use_pxw) launch_dec_pxw(t, plan, st);
        else if (use_pxu) launch_dec_pxu(t, plan, st); else launch_dec_generic(t, plan, st);
    };
    const bool best_pxw = best && plan.pxw_best && ((uintptr_t)a.img & (a.g.tsz - 1)) == 0 && !(a.ts_img & (a.g.tsz - 1));
    const bool best_pxu = best && plan.pxu_best && val_aligned;
    const bool best_px = (best && plan.px_best && a.g.tsz == 1) || best_pxw || best_pxu;       // a lane-per-block (or per-unit) common-factor decoder applies
    auto dec_best_lpb = [&](const DecArgs &t) { if (best_pxw) launch_dec_pxw_best(t, plan, st); else if (best_pxu) launch_dec_pxu_best(t, plan, st); else launch_dec_px_best(t, plan, st); };
    // the table's check: the wave-per-segment decoders that work from the entries alone make it with the first workgroups of their own
    // launch (DecArgs::chk_wgs: one launch, not two); everything else has ix_check_kernel in front
    const bool from_entries = rebuild && a.ix && a.ix_bl && a.ix_blocks == a.g.seg_blocks && !tuning().slow_index && !tuning().no_bl;
    const bool fold_check = need_check && from_entries && (best_px || (!best && (use_px || (use_px16 && ix_block_lens_ok(a.g)) || use_pxw || use_pxu)));
    if (need_check && !fold_check)
        hipLaunchKernelGGL(ix_check_kernel, dim3((a.ix_K + a.ix_per_chunk - 1) / a.ix_per_chunk, a.ntiles), dim3(256), 0, st, a);
    if (rebuild && best_px && a.ix && a.ix_bl && a.ix_blocks == a.g.seg_blocks && !tuning().slow_index && !tuning().no_bl) {
        // the container's table has a field per block (bits, entering rungs): the lane-per-block decoder works from the entries alone
        DecArgs t = a;
        t.bl_mode = 1;
        if (fold_check) t.chk_wgs = (a.ix_K + a.ix_per_chunk - 1) / a.ix_per_chunk;
        ProfScope ps("dec_units", st);
        dec_best_lpb(t);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (rebuild && (use_px || (use_px16 && ix_block_lens_ok(a.g)) || unit_parallel || use_pxu) && a.ix && a.ix_bl && a.ix_blocks == a.g.seg_blocks && !tuning().slow_index && !tuning().no_bl) {
        // the container's table carries block (16-bit data: band pair) lengths: the lane-per-block decoder works from the entries alone
        DecArgs t = a;
        t.bl_mode = 1;
        if (fold_check) t.chk_wgs = (a.ix_K + a.ix_per_chunk - 1) / a.ix_per_chunk;
        ProfScope ps("dec_units", st);
        dec_units(t);
        HIPCHK(hipGetLastError());
        return 0;
    }
    // plain 32/64-bit FTL/BASE streams: unit lengths through the table of a band of rungs, when the caller brought memory for it
    const bool wide_plain = rebuild && !a.ix && unit_parallel && walk_tab && walk_tab_bytes >= walk_table_min_bytes(a.ntiles, a.g.tsz) && !tuning().slow_walk;
    const bool pxu16_plain = rebuild && !a.ix && use_pxu && a.g.tsz <= 2 && walk_tab && walk_tab_bytes >= walk_table_min_bytes(a.ntiles, a.g.tsz) && !tuning().slow_walk;
    if (rebuild && (use_px || use_px16 || wide_walk || wide_plain || pxu16_plain) && !tuning().slow_index) {
        // index-less stream through the lane-per-block kernels: walk the lengths, then let the parallel decoder itself
        // produce the values entering the segments (totals pass + scan)
        // plain 8-bit stream: through the table of unit lengths by position when the caller brought memory for it
        const bool has_ix = a.ix != nullptr;
        bool have_prev = false;             // the index's entering values are there already
        if ((use_px || use_px16 || wide_plain || pxu16_plain) && !has_ix && walk_tab && walk_tab_bytes >= walk_table_min_bytes(a.ntiles, a.g.tsz) && !tuning().slow_walk) launch_dec_walk_table(a, st, walk_tab, walk_tab_bytes, max_bits);
        else if (has_ix) { ProfScope ps("dec_index_serial", st); launch_dec_walk(a, st); }
        else { ProfScope ps("dec_index_serial", st); launch_dec_index_serial(a, st); have_prev = true; }       // no table memory: one lane parses the stream (values included)
        if (!have_prev && !(a.ix && a.ix_blocks == a.g.seg_blocks) && !wide_walk) {         // (an entry per segment: the walk copied the entering values)
          {
            ProfScope ps("dec_index_prev", st);
            DecArgs t = a;
            t.totals_only = 1;
            dec_units(t);
          }
          ProfScope ps("dec_index_scan", st);
          launch_prev_scan(a, st);
        }
    } else if (rebuild && !a.from_ix) {
        // plain single-band common-factor streams: segment entries by the walk through exits, entering values by a scan
        // of the segments' sums; anything else (and that walk when it has no memory): one lane parses the stream
        const bool best_plain = best && !a.ix && (a.g.bands == 1 || ((a.g.bands == 3 || a.g.bands == 2) && a.g.tsz == 1)) && walk_tab && walk_tab_bytes >= walk_table_min_bytes(a.ntiles, a.g.tsz) &&
                                !tuning().slow_walk && !tuning().slow_index && launch_dec_walk_best(a, st, walk_tab, walk_tab_bytes, max_bits);
        // ... of several bands (8- and 16-bit data): the chain, the walking lane parsing the units with the signal code; values as below
        const bool best_chain = best && !best_plain && !a.ix && best_chain_applies(a.g, plan) && walk_tab && walk_tab_bytes >= walk_table_min_bytes(a.ntiles, a.g.tsz) &&
                                walk_chain_lds_ok() && !tuning().slow_walk && !tuning().slow_index;
        if (best_plain) { ProfScope ps("dec_index_scan", st); launch_prev_scan(a, st); }
        else if (best_chain) {
            {   // the factors in force start from zero (the lane writes them from the first unit that brings one on); the block table is added up
                const size_t cfb = (size_t)a.g.nseg * a.g.bands * a.g.tsz, ulb = a.g.ulen_sz == 4 ? (size_t)a.g.nblocks * 4 : 0;
                if (a.ntiles > 1) { (void)hipMemset2DAsync(a.idx.cf, a.ts_idx, 0, cfb, a.ntiles, st); if (ulb) (void)hipMemset2DAsync(a.idx.ulen, a.ts_idx, 0, ulb, a.ntiles, st); }
                else { (void)hipMemsetAsync(a.idx.cf, 0, cfb, st); if (ulb) (void)hipMemsetAsync(a.idx.ulen, 0, ulb, st); }
            }
            if (a.g.tsz == 2) walk_chain_16bit(a, st, walk_tab, walk_tab_bytes, max_bits); else walk_chain_8bit_any(a, st, walk_tab, walk_tab_bytes, max_bits);
            { ProfScope ps("dec_index_prev", st); DecArgs t = a; t.totals_only = 1; if (best_pxu) launch_dec_pxu_best(t, plan, st); else launch_dec_generic(t, plan, st); }
            ProfScope ps("dec_index_scan", st);
            launch_prev_scan(a, st);
        }
        else if (best && !a.ix && dec_index_walk_best_ok(a) && !tuning().slow_index) {
            // common-factor streams the exits do not take (several bands; no table memory): one wave walks lengths (units with
            // the signal code parsed outright), the generic decoder adds up every segment's values, a scan makes entering values
            // of the sums
            { ProfScope ps("dec_index_serial", st); launch_dec_index_walk_best(a, st); }
            { ProfScope ps("dec_index_prev", st); DecArgs t = a; t.totals_only = 1; if (best_pxu) launch_dec_pxu_best(t, plan, st); else launch_dec_generic(t, plan, st); }
            ProfScope ps("dec_index_scan", st);
            launch_prev_scan(a, st);
        }
        else { ProfScope ps("dec_index_serial", st); launch_dec_index_serial(a, st); }
    }
    if (best_px && !a.from_ix) { ProfScope ps("dec_units", st); dec_best_lpb(a); }
    else if (use_px) { ProfScope ps("dec_units", st); launch_dec_px(a, plan, st); }
    else if (use_px16) { ProfScope ps("dec_units", st); launch_dec_px16(a, plan, st); }
    else if (use_pxw) { ProfScope ps("dec_units", st); launch_dec_pxw(a, plan, st); }
    else if (use_pxu) { ProfScope ps("dec_units", st); launch_dec_pxu(a, plan, st); }
    else { ProfScope ps(plan.fast && !best ? "dec_units" : "dec_segments", st); launch_dec_generic(a, plan, st); }
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_decode(const Geometry &g, const DecPlan &plan_in, const uint32_t *in32, uint32_t in_bit0, uint64_t in_bits,
                  void *img, const void *index, void *ws, uint32_t **status_out, void *stream, const TileBatch &tb,
                  const uint64_t *tile_bits, const IxTable &ix, void *walk_tab, size_t walk_tab_bytes, bool full_staging, uint32_t wide_band, const DecStrip *strip) {
    hipStream_t st = (hipStream_t)stream;
    DecArgs a;
    // 16-bit lane-per-block decoder: a wave stages its segment in LDS, and four worst-case segments (278 bits a unit) keep
    // the CU at 4 workgroups.  Size the staging for a third above THIS stream's average segment instead; a segment that
    // does not fit raises status bit 4 and the caller runs the call again with full_staging.
    DecPlan plan = plan_in;
    if (plan.px16 && !full_staging && g.nseg) {
        const uint64_t bits = tb.n ? tb.max_bits : in_bits;
        uint64_t cap = bits / 32 / g.nseg;
        cap = (cap + cap / 3 + 64 + 3) & ~(uint64_t)3;
        if (bits && cap < plan.px_cap_dw) { plan.px_cap_dw = (uint32_t)cap; plan.lds_px = 4096 + 4 * 4 * ((size_t)cap + 16); }
    }
    if ((plan.pxw || plan.pxw_best) && !full_staging && g.nseg) {          // the same for 32/64-bit data (worst case: 4.3 / 8.4 KB a wave)
        const uint64_t bits = tb.n ? tb.max_bits : in_bits;
        uint64_t cap = bits / 32 / g.nseg;
        cap = (cap + cap / 2 + 64 + 3) & ~(uint64_t)3;
        if (bits && cap < plan.px_cap_dw) { plan.px_cap_dw = (uint32_t)cap; plan.lds_pxw = (plan.pxw ? 2048 : 0) + 4 * 4 * ((size_t)cap + WIDE_PAD_DW); }
    }
    if ((plan.pxu || plan.pxu_best) && !full_staging && g.nseg) {        // ... and for the lane-per-unit kernels
        const uint64_t bits = tb.n ? tb.max_bits : in_bits;
        uint64_t cap = bits / 32 / g.nseg;
        cap = (cap + cap / 2 + 64 + 3) & ~(uint64_t)3;
        if (bits && cap < plan.px_cap_dw) plan.px_cap_dw = (uint32_t)cap;
    }
    a.in_cap_full = plan_in.px_cap_dw;
    // the container's coarse restart table is usable when it matches this geometry and this library's segments
    a.ix = nullptr; a.ix_K = a.ix_blocks = a.ix_E = a.ix_per_chunk = a.ix_pad = 0; a.ix_bl = 0; a.ix_ver = 0; a.ix_check_heads = 0; a.chk_wgs = 0;
    if (ix.base && ix.blocks && ix.per_chunk && ix.blocks % g.seg_blocks == 0 && ix.entry_bytes == ix_entry_bytes(g, ix.block_lens) &&
        (!ix.block_lens || (ix_block_lens_ok(g) && ix.blocks == g.seg_blocks)) &&
        ix.K == (g.nblocks + ix.blocks - 1) / ix.blocks) {
        a.ix = ix.base; a.ix_K = ix.K; a.ix_blocks = ix.blocks; a.ix_E = ix.entry_bytes; a.ix_per_chunk = ix.per_chunk;
        a.ix_pad = ix.pads ? IX_PAD : 0;
        a.ix_bl = ix.block_lens;
        a.ix_ver = ix.version; a.ix_check_heads = ix.check_heads ? 1u : 0u;
    }
    // lane-per-segment decoder: LDS for the stream words of a workgroup's segments, half as much again as the average,
    // when that is at most 24 KB (more would cost more in resident workgroups than the staging saves; a longer span is
    // read from global memory).  With a restart table in the container and no index, the lanes decode straight from the
    // table's entries (from_ix: no index is rebuilt at all; their pieces are long, so usually not staged).
    a.seg_cap_dw = 0;
    const bool lane_per_segment = !(plan.fast && g.mode != CM_BEST);
    a.from_ix = (lane_per_segment && index == nullptr && a.ix && !tuning().slow_index) ? 1u : 0u;
    if (g.nseg && lane_per_segment) {
        const uint64_t bits = tb.n ? tb.max_bits : in_bits;
        uint64_t cap = bits / 32 * plan.threads / (a.from_ix ? a.ix_K : g.nseg);
        cap = (cap + cap / 2 + 64 + 3) & ~(uint64_t)3;
        if (bits && cap <= 24 * 1024 / 4) a.seg_cap_dw = (uint32_t)cap;
    }
    a.g = g; a.in32 = in32; a.in_bit0 = in_bit0; a.in_bits = in_bits; a.img = img;
    a.ntiles = tb.n ? tb.n : 1; a.ts_in = tb.src_pitch; a.ts_img = tb.dst_pitch; a.tile_bits = tile_bits;
    uint8_t *w = (uint8_t *)ws;
    const bool rebuild = index == nullptr;
    // workspace: [status words, 64 bytes per 16 tiles][rebuilt indices, one per tile]
    const size_t status_bytes = ((4 * (size_t)a.ntiles + 63) / 64) * 64;
    a.status = (uint32_t *)w;
    a.idx = index_view(g, rebuild ? (void *)(w + status_bytes) : const_cast<void *>(index));
    a.ts_idx = rebuild ? align8(index_bytes(g)) : tb.idx_pitch;
    if (!strip || strip->first) HIPCHK(hipMemsetAsync(a.status, 0, status_bytes, st));
    a.seg0 = strip ? strip->seg0 : 0;
    a.seg_end = strip ? std::min<uint64_t>(g.nseg, strip->seg0 + strip->nseg) : g.nseg;
    a.lane_dw = dec_lane_dwords(g);
    a.dpr = g.bands * g.tsz;
    a.bpp = plan.bpp; a.passes = plan.passes; a.in_cap_dw = (plan.px || plan.px16 || plan.px_best || plan.pxw || plan.pxw_best || plan.pxu || plan.pxu_best) ? plan.px_cap_dw : plan.in_cap_dw;
    a.px_ng = plan.px16 ? plan.px16_ng : 1; a.px_magic_ng = magic_div(a.px_ng);
    a.totals_only = 0;
    a.bl_mode = 0;
    a.wide_band = tuning().wide_band ? (uint32_t)tuning().wide_band : wide_band;
    a.px_aligned = !(g.w & 3) && !((g.stride * g.tsz) & 3) && !((uintptr_t)img & 3) && !(tb.dst_pitch & 3);
    a.magic_bpp = magic_div(plan.bpp); a.magic_dpr = magic_div(a.dpr); a.magic_bands = magic_div(g.bands);
    *status_out = a.status;
    if (g.tsz != 1 && g.tsz != 2 && g.tsz != 4 && g.tsz != 8) { set_error("decode: bad value size", 0); return -1; }
    if (strip) {        // one launch of the lane-per-block decoder over the strip's segments, from the table's entries
        if (rebuild && strip->first && a.ix && a.ix_K && (a.ix_ver >= 3 || a.ix_check_heads))
            hipLaunchKernelGGL(ix_check_kernel, dim3((a.ix_K + a.ix_per_chunk - 1) / a.ix_per_chunk, a.ntiles), dim3(256), 0, st, a);
        if (!rebuild || !a.ix || !decode_strips_ok(g, plan_in, ix)) { set_error("decode: strips need the container's table", 0); return -1; }
        a.bl_mode = 1;
        ProfScope ps("dec_units", st);
        const bool best = g.mode == CM_BEST;
        if (best && plan.pxw_best) launch_dec_pxw_best(a, plan, st);
        else if (best && plan.pxu_best) launch_dec_pxu_best(a, plan, st);
        else if (best) launch_dec_px_best(a, plan, st);
        else if (plan.px) launch_dec_px(a, plan, st);
        else if (plan.px16) launch_dec_px16(a, plan, st);
        else if (plan.pxu) launch_dec_pxu(a, plan, st);
        else launch_dec_pxw(a, plan, st);
        HIPCHK(hipGetLastError());
        return 0;
    }
    return launch_decode_all(a, plan, rebuild, st, walk_tab, walk_tab_bytes, tb.n ? tb.max_bits : in_bits);
}

bool decode_strips_ok(const Geometry &g, const DecPlan &plan, const IxTable &ix) {
    if (!ix.base || !ix.block_lens || !ix.blocks || ix.blocks != g.seg_blocks || !ix.per_chunk || tuning().slow_index || tuning().no_bl) return false;
    if (ix.entry_bytes != ix_entry_bytes(g, true) || !ix_block_lens_ok(g) || ix.K != (g.nblocks + ix.blocks - 1) / ix.blocks) return false;
    if (g.mode == CM_BEST) return (plan.px_best && g.tsz == 1) || plan.pxw_best || plan.pxu_best;
    return (plan.px && g.tsz == 1) || (plan.px16 && g.tsz == 2) || plan.pxw || (plan.pxu && plan.fast);
}

}  // namespace qb3dev

// ------------------------------------------------------------------ quantisation (elementwise, HBM bound)
namespace qb3dev {

// reference QB3encode.cpp:137-186: round to nearest; ties toward zero, or away from zero when `away`
template <typename TS>
__global__ void quantize_kernel(TS *dst, const TS *src, uint32_t rowvals, uint32_t rows, uint64_t stride, uint64_t quanta, int away) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)rowvals * rows) return;
    const uint64_t y = i / rowvals, x = i - y * rowvals;
    const TS v = src[y * stride + x], q = (TS)quanta;
    TS r;
    if (q == 2) r = away ? (TS)(v / 2 + v % 2) : (TS)(v / 2);
    else if (q == 3) r = (TS)(v / 3 + (v % 3) / 2);
    else if (q == 4) r = away ? (TS)(v / 4 + (v % 4) / 2) : (TS)(v / 4 + (v % 4) / 3);
    else {
        const TS m = (TS)(v % q);
        const bool neg = v < (TS)0;
        if (away) { const TS h = (TS)(q / 2 + q % 2); r = (TS)(v / q + (!neg & (m >= h)) - (neg & ((TS)(m + h) <= (TS)0))); }
        else { const TS h = (TS)(q / 2); r = (TS)(v / q + (!neg & (m > h)) - (neg & ((TS)(m + h) < (TS)0))); }
    }
    dst[i] = r;
}

// reference QB3decode.cpp:77-107: multiply back, saturating
template <typename TS>
__global__ void dequantize_kernel(TS *img, uint32_t rowvals, uint32_t rows, uint64_t stride, uint64_t quanta) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)rowvals * rows) return;
    const uint64_t y = i / rowvals, x = i - y * rowvals;
    constexpr bool is_signed = (TS)-1 < (TS)0;
    constexpr TS tmax = is_signed ? (TS)(((uint64_t)1 << (8 * sizeof(TS) - 1)) - 1) : (TS)~(TS)0;
    constexpr TS tmin = is_signed ? (TS)((uint64_t)1 << (8 * sizeof(TS) - 1)) : (TS)0;
    const TS q = (TS)quanta, mai = (TS)(tmax / q), mii = (TS)(tmin / q);
    const TS v = img[y * stride + x];
    TS r = (v <= mai) ? (TS)(v * q) : tmax;
    if (is_signed && q > 2 && v < mii) r = tmin;
    img[y * stride + x] = r;
}

template <typename TS> static int launch_q(void *dst, const void *src, const Geometry &g, uint64_t q, bool away, hipStream_t st) {
    const uint64_t n = (uint64_t)g.w * g.bands * g.h;
    hipLaunchKernelGGL(quantize_kernel<TS>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, (TS *)dst, (const TS *)src,
                       g.w * g.bands, g.h, g.stride, q, away ? 1 : 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("quantize", (int)e); return (int)e; }
    return 0;
}
template <typename TS> static int launch_dq(void *img, const Geometry &g, uint64_t q, hipStream_t st) {
    const uint64_t n = (uint64_t)g.w * g.bands * g.h;
    hipLaunchKernelGGL(dequantize_kernel<TS>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, (TS *)img, g.w * g.bands, g.h, g.stride, q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dequantize", (int)e); return (int)e; }
    return 0;
}

int launch_quantize(void *dst, const void *src, const Geometry &g, int dtype, uint64_t q, bool away, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
    case 0: return launch_q<uint8_t>(dst, src, g, q, away, st);   case 1: return launch_q<int8_t>(dst, src, g, q, away, st);
    case 2: return launch_q<uint16_t>(dst, src, g, q, away, st);  case 3: return launch_q<int16_t>(dst, src, g, q, away, st);
    case 4: return launch_q<uint32_t>(dst, src, g, q, away, st);  case 5: return launch_q<int32_t>(dst, src, g, q, away, st);
    case 6: return launch_q<uint64_t>(dst, src, g, q, away, st);  case 7: return launch_q<int64_t>(dst, src, g, q, away, st);
    }
    return -1;
}
int launch_dequantize(void *img, const Geometry &g, int dtype, uint64_t q, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
    case 0: return launch_dq<uint8_t>(img, g, q, st);   case 1: return launch_dq<int8_t>(img, g, q, st);
    case 2: return launch_dq<uint16_t>(img, g, q, st);  case 3: return launch_dq<int16_t>(img, g, q, st);
    case 4: return launch_dq<uint32_t>(img, g, q, st);  case 5: return launch_dq<int32_t>(img, g, q, st);
    case 6: return launch_dq<uint64_t>(img, g, q, st);  case 7: return launch_dq<int64_t>(img, g, q, st);
    }
    return -1;
}

}  // namespace qb3dev
