#!/usr/bin/env python3
"""profiles/summarise.py -- turn rocprofv3 output into the summaries committed under profiles/.

    python3 profiles/summarise.py TAG TRACE_DIR FETCH_DIR WRITE_DIR [SQ_DIR]          the headline (BASELINE configs[1])
    python3 profiles/summarise.py --stats-only TAG_WL TRACE_DIR                       kernel trace of another workload
    python3 profiles/summarise.py --workload TAG WL FETCH_DIR WRITE_DIR SQ_DIR        its counter passes

TRACE_DIR  output of  rocprofv3 --kernel-trace --stats -d TRACE_DIR -o t --output-format csv -- python3 bench.py ...
FETCH_DIR  output of  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ... -- python3 bench.py ...     (own pass)
WRITE_DIR  output of  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d ... -- python3 bench.py ...     (own pass)
SQ_DIR     output of  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU ... GRBM_GUI_ACTIVE -d ... -- python3 bench.py ...

Writes profiles/TAG[_WL]_kernel_stats.csv, profiles/TAG[_WL]_pmc_hbm.csv, profiles/TAG[_WL]_sq.csv and REPLACES the records
of that workload in profiles/pmc_traffic.json (what bench.py reports as roofline.traffic and roofline.valu).  Every record
is made from ONE run: collecting the same tag again overwrites it, nothing is ever added to what an earlier run left, and
records of kernels the run did not launch are dropped.  A record's hbm_bytes_per_launch is a row of the CSV named in its
`source` (the row of its symbol, or the "(sum)" row when several symbols run under one profile name).

The SQ pass answers what actually limits the kernels: a wave64 VALU instruction occupies its SIMD-32 for 2 clocks
(MI355X_MICROARCH.md; one wave alone issues every 4), so SQ_INSTS_VALU x 2 / 1024 SIMDs is the VALU issue time;
GRBM_GUI_ACTIVE / 8 XCDs is the kernel's duration in clocks.  Counter unit and the gfx950 correction follow
MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are KiB per dispatch; FETCH_SIZE counts 128-byte requests as 64 bytes on
gfx950 (x2), WRITE_SIZE is exact.
"""
import collections
import csv
import re
import glob
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SQ_NAMES = ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE")
# library profile names (qb3x_profile_names) by a substring of the kernel symbol
KEYS = [("walk_exitW_kernel", "dec_index_table"), ("walk_exitB_kernel", "dec_index_table"), ("walk_exitB_chain", "dec_index_serial"), ("walk_exit_", "dec_index_serial"),
        ("walk_probe", "dec_index_serial"), ("walk_tableW", "dec_index_table"), ("walk_chainW", "dec_index_serial"), ("walk_table16", "dec_index_table"), ("walk_chain16", "dec_index_serial"),
        ("enc_px_best_sample_kernel", "enc_best_sample"), ("enc_best_sample_kernel", "enc_best_sample"), ("enc_pxw_best_kernel", "enc_best_units"),
        ("enc_px_best_kernel", "enc_best_units"), ("dec_px_best_kernel", "dec_units"), ("dec_pxw_best_kernel", "dec_units"), ("dec_pxu_best_kernel", "dec_units"), ("dec_pxu_kernel", "dec_units"), ("ix_blu_best_fill", "ix_bl_fill"), ("ix_bl_best_fill", "ix_bl_fill"),
        ("best_idx_fix", "enc_best_idx_fix"), ("ix_bl16_fill", "ix_bl_fill"), ("ix_blw_fill", "ix_bl_fill"), ("rle0_", "rle0"), ("enc_px_kernel", "enc_units"), ("enc_px16_kernel", "enc_units"),
        ("enc_pxw_kernel", "enc_units"), ("enc_kernel", "enc_units"),
        ("enc_best_kernel<unsigned char, false>", "enc_best_recode"), ("enc_best_kernel<unsigned short, false>", "enc_best_recode"),
        ("enc_best_kernel<unsigned int, false>", "enc_best_recode"), ("enc_best_kernel<unsigned long, false>", "enc_best_recode"),
        ("enc_best_kernel", "enc_best_units"), ("best_scan", "enc_best_scan"),
        ("enc_scan2", "enc_scan2"), ("enc_scan", "enc_scan"), ("enc_concat", "enc_concat"), ("enc_seam", "enc_seams"), ("enc_finish", "enc_finish"), ("ix_seal", "ix_seal"), ("ix_check", "ix_check"),
        ("write_header", "write_header"), ("ix_bl_fill", "ix_bl_fill"), ("ix_fill", "ix_fill"), ("dec_px_kernel", "dec_units"), ("dec_px16_kernel", "dec_units"), ("dec_pxw_kernel", "dec_units"), ("dec3_kernel", "dec_units"),
        ("dec_walk_lanes", "dec_index_serial"), ("dec_walk_kernel", "dec_index_serial"), ("prev_scan", "dec_index_scan"),
        ("walk_table_kernel", "dec_index_table"), ("walk_chain_kernel", "dec_index_serial"),
        ("dec_index_serial", "dec_index_serial"), ("dec_index_staged", "dec_index_serial"), ("dec_kernel", "dec_segments")]
# kernels with a variant that decodes from the container's own table (last template argument BL = true): the two variants
# run in different calls (decode from the container alone / with the out-of-band index), never in one -- apart, not summed
BL_VARIANTS = ("dec_px_kernel<", "dec_px16_kernel<", "dec_px_best_kernel<", "dec_pxw_kernel<", "dec_pxw_best_kernel<", "dec3_kernel<", "dec_pxu_kernel<", "dec_pxu_best_kernel<")


def key_of(name):
    if "qb3dev" not in name:
        return None
    if ("enc_px_best_kernel" in name or "enc_pxw_best_kernel" in name) and re.search(r", false>\(", name):
        return "enc_best_recode"
    if re.search(r"enc_best_kernel<[a-z ]+, false", name):      # (FIRST = false: the recode pass, whatever the front end)
        return "enc_best_recode"
    for sub, key in KEYS:
        if sub in name:
            if key == "dec_units" and any(v in name for v in BL_VARIANTS) and re.search(r", true>\(", name):
                return "dec_units_bl"
            return key
    return "other"


def short(name):
    return re.sub(r"\(.*\)$", "", name.replace("qb3dev::", "").replace("void ", ""))


def find(d, suffix):
    m = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not m:
        raise SystemExit("no %s under %s" % (suffix, d))
    return m[0]


def kernel_stats(trace_dir):
    rows = list(csv.DictReader(open(find(trace_dir, "kernel_trace.csv"))))
    acc = collections.defaultdict(list)
    for r in rows:
        acc[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = []
    for name, d in acc.items():
        if "qb3dev" in name:
            out.append((name, len(d), sum(d), sum(d) / len(d), min(d), max(d)))
    out.sort(key=lambda t: -t[2])
    return out


def pmc_by_name(d, counter):
    """average per dispatch of `counter`, and the dispatch count, per kernel SYMBOL of this library"""
    rows = list(csv.DictReader(open(find(d, "counter_collection.csv"))))
    acc = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter and "qb3dev" in r["Kernel_Name"]:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def subtag_of(wl, name):
    """the key of bench.py's `workloads` a kernel of workload wl belongs to (config 4 runs four rasters in one process);
    None: the headline, whose records sit at the top level of pmc_traffic.json"""
    if wl is None:
        return None
    if wl == "c4":
        t = "i64" if "unsigned long" in name else "i32"
        best = any(k in name for k in ("enc_best", "enc_pxw_best", "dec_pxw_best", "best_scan", "best_idx_fix", "dec_kernel<", "dec_index"))
        return "c4_%s_%s" % (t, "best" if best else "ftl")
    if wl == "shapes":      # one raster per value type and mode family (bench.py run_other): the symbol says which
        t = "u64" if "unsigned long" in name else "u32" if "unsigned int" in name else "u16" if "unsigned short" in name else "u8" if "unsigned char" in name else None
        best = "best" in name
        return {("u8", False): "u8x5_ftl", ("u16", False): "u16x7_base", ("u32", False): "i32x2_ftl", ("u64", False): "i64x2_ftl",
                ("u8", True): "u8x5_cf", ("u32", True): "i32x3_best"}.get((t, best), "shapes")
    return {"c2best": "c2_best", "c5": "c5_one_rank", "c3cf": "c3_cf"}.get(wl, wl)


def counters(tag, wl, fetch_dir, write_dir, sq_dir, cmd):
    """TAG[_WL]_pmc_hbm.csv / TAG[_WL]_sq.csv per kernel SYMBOL (+ a "(sum)" row per profile name run under several symbols),
    and the records of this run in pmc_traffic.json -- made from scratch, replacing whatever the file held for them"""
    stem = tag if wl is None else "%s_%s" % (tag, wl)
    fetch, nf = pmc_by_name(fetch_dir, "FETCH_SIZE")
    write, _ = pmc_by_name(write_dir, "WRITE_SIZE")
    fresh = collections.defaultdict(dict)          # subtag -> profile name -> record
    groups = collections.defaultdict(list)         # (subtag, profile name) -> [(symbol, dispatches, fetch KiB, write KiB, bytes)]
    for name in sorted(set(fetch) | set(write)):
        fk, wk = fetch.get(name, 0.0), write.get(name, 0.0)
        groups[(subtag_of(wl, name), key_of(name))].append((short(name), nf.get(name, 0), fk, wk, int(round((2 * fk + wk) * 1024))))
    src_hbm = "profiles/%s_pmc_hbm.csv" % stem
    with open(os.path.join(HERE, stem + "_pmc_hbm.csv"), "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- %s\n" % cmd)
        f.write("# KiB per dispatch, averaged over the dispatches of ONE run; gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE x2, WRITE_SIZE x1\n")
        f.write("# a \"(sum)\" row: the symbols above it run under one profile name in one call (a sample kernel beside the main one)\n")
        f.write("workload,kernel,symbol,dispatches,FETCH_SIZE_KiB,WRITE_SIZE_KiB,hbm_bytes_per_launch_corrected\n")
        for (st, k), syms in sorted(groups.items(), key=lambda kv: (str(kv[0][0]), str(kv[0][1]))):
            for s, n, fk, wk, b in syms:
                f.write('%s,%s,"%s",%d,%.1f,%.1f,%d\n' % (st or "c2", k, s, n, fk, wk, b))
            tot = sum(t[4] for t in syms)
            if len(syms) > 1:
                f.write('%s,%s,"(sum)",%d,%.1f,%.1f,%d\n' % (st or "c2", k, max(t[1] for t in syms), sum(t[2] for t in syms), sum(t[3] for t in syms), tot))
            fresh[st][k] = {"hbm_bytes_per_launch": tot, "fetch_KiB": round(sum(t[2] for t in syms), 1), "write_KiB": round(sum(t[3] for t in syms), 1),
                            "symbols": [t[0] for t in syms], "source": src_hbm}
    if sq_dir:
        vals = {n: pmc_by_name(sq_dir, n)[0] for n in SQ_NAMES}
        src_sq = "profiles/%s_sq.csv" % stem
        with open(os.path.join(HERE, stem + "_sq.csv"), "w") as f:
            f.write("# rocprofv3 --kernel-trace --pmc %s -- %s\n" % (" ".join(SQ_NAMES), cmd))
            f.write("# per dispatch, summed over the 8 XCDs. valu_issue_frac = SQ_INSTS_VALU*2/1024 / (GRBM_GUI_ACTIVE/8); "
                    "lds_busy_frac = SQ_LDS_IDX_ACTIVE/256 / (GRBM_GUI_ACTIVE/8)\n")
            f.write("workload,kernel,symbol," + ",".join(SQ_NAMES) + ",valu_per_wave,valu_issue_frac,lds_busy_frac\n")
            best = {}
            for name in sorted(vals["SQ_WAVES"]):
                v = {n: vals[n].get(name, 0.0) for n in SQ_NAMES}
                dur = v["GRBM_GUI_ACTIVE"] / 8 or 1.0
                vpw = v["SQ_INSTS_VALU"] / (v["SQ_WAVES"] or 1.0)
                vf, lf = v["SQ_INSTS_VALU"] * 2 / 1024 / dur, v["SQ_LDS_IDX_ACTIVE"] / 256 / dur
                st, k = subtag_of(wl, name), key_of(name)
                f.write('%s,%s,"%s",' % (st or "c2", k, short(name)) + ",".join("%.0f" % v[n] for n in SQ_NAMES) + ",%.0f,%.3f,%.3f\n" % (vpw, vf, lf))
                if v["SQ_INSTS_VALU"] >= best.get((st, k), -1.0):       # several symbols under one name: the one that issues most
                    best[(st, k)] = v["SQ_INSTS_VALU"]
                    fresh[st].setdefault(k, {})["valu"] = {"insts_per_wave": round(vpw), "issue_time_frac": round(vf, 3), "lds_busy_frac": round(lf, 3),
                                                           "symbol": short(name), "source": src_sq}
    path = os.path.join(HERE, "pmc_traffic.json")
    try:
        traffic = json.load(open(path))
    except (OSError, ValueError):
        traffic = {}
    is_kernel_record = lambda v: isinstance(v, dict) and ("hbm_bytes_per_launch" in v or "valu" in v)
    for st, recs in fresh.items():
        if st is None:          # the headline: every top-level kernel record goes, the workloads' dictionaries stay
            traffic = {k: v for k, v in traffic.items() if not is_kernel_record(v)}
            traffic.update(recs)
        else:
            traffic[st] = recs
    with open(path, "w") as f:
        json.dump(traffic, f, indent=1, sort_keys=True)


def main():
    if sys.argv[1] == "--workload":         # TAG WL FETCH_DIR WRITE_DIR SQ_DIR
        tag, wl, fetch_dir, write_dir, sq_dir = sys.argv[2:7]
        counters(tag, wl, fetch_dir, write_dir, sq_dir, os.environ.get("PROFILE_CMD", ""))
        return
    stats_only = sys.argv[1] == "--stats-only"
    if stats_only:
        sys.argv.pop(1)
    tag, trace_dir = sys.argv[1:3]
    cmd = os.environ.get("PROFILE_CMD", "python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline")
    stats = kernel_stats(trace_dir)
    with open(os.path.join(HERE, tag + "_kernel_stats.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- %s   (MI355X)\n" % cmd)
        f.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs\n")
        for name, n, tot, avg, lo, hi in stats:
            f.write('"%s",%d,%d,%.1f,%d,%d\n' % (name, n, tot, avg, lo, hi))
    for name, n, tot, avg, lo, hi in stats:
        print("%-28s calls %3d avg %9.1f us" % (key_of(name), n, avg / 1e3))
    if stats_only:
        return
    fetch_dir, write_dir = sys.argv[3:5]
    counters(tag, None, fetch_dir, write_dir, sys.argv[5] if len(sys.argv) > 5 else None, cmd)


if __name__ == "__main__":
    main()
