#!/bin/bash
# tools/sq_probe.sh OUTDIR W H BANDS DTYPE GEN MODE -- SQ counters and kernel times of one raster's kernels (tools/kernel_probe.py under rocprofv3)
# a measuring aid; run on the GPU box from the repo root
set -e
OUT=gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p $OUT
TAG=$(echo "$*" | tr ' ' '_')
rocprofv3 --kernel-trace --stats -d $OUT/t_$TAG -o t --output-format csv -- python3 tools/kernel_probe.py "$@" 3 > $OUT/probe_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/s_$TAG -o s --output-format csv -- python3 tools/kernel_probe.py "$@" 3 > /dev/null 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections, re
out, tag = sys.argv[1], sys.argv[2]
def short(n):
    n = re.sub(r"^void ", "", n); n = n.replace("qb3dev::", "")
    return n[:70]
st = glob.glob(f"{out}/t_{tag}/**/*kernel_stats.csv", recursive=True)
times = {}
for f in st:
    for r in csv.DictReader(open(f)):
        times[short(r["Name"])] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(f"{out}/s_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
with open(f"{out}/sq_{tag}.txt", "w") as fo:
    for k, c in sorted(acc.items(), key=lambda kv: -times.get(kv[0], (0, 0))[0]):
        n = max(cnt[k], 1); w = c["SQ_WAVES"] / n
        if w < 1: continue
        t = times.get(k, (0, 0))
        line = "%-70s %8.1f us x%-3d waves %8d  valu/wave %6.0f  salu/wave %5.0f  lds/wave %5.0f  lds_active %.2f  bank_conf %.2f" % (
            k, t[0], t[1], w, c["SQ_INSTS_VALU"] / n / w, c["SQ_INSTS_SALU"] / n / w, c["SQ_INSTS_LDS"] / n / w,
            c["SQ_LDS_IDX_ACTIVE"] / max(c["GRBM_GUI_ACTIVE"], 1) / 256 * 8 if c["GRBM_GUI_ACTIVE"] else 0, c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1))
        print(line); fo.write(line + "\n")
PY
