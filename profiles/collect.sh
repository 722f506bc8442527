#!/bin/bash
# profiles/collect.sh TAG -- the rocprofv3 runs behind profiles/TAG_* (run on the GPU box from the repo root).
# Counters are collected in their own passes with --kernel-trace only (no sys/hip/hsa tracing next to --pmc).
# TAG_*: BASELINE configs[1] alone (bench.py --no-workloads): kernel trace, FETCH_SIZE, WRITE_SIZE and SQ passes.
# TAG_c2best / TAG_c3 / TAG_c4 / TAG_c5 / TAG_plain: the other configurations (bench.py --workload ...): kernel trace and
# the same FETCH_SIZE, WRITE_SIZE and SQ passes (TAG_<wl>_pmc_hbm.csv, TAG_<wl>_sq.csv;
# pmc_traffic.json keeps them under the workload's name in bench.py's `workloads`).  WORKLOADS="c2best c3" limits the list.
set -e
TAG=${1:-rXX}
OUT=gpurun_out/prof_$TAG
CMD="bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-workloads"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p $OUT
if [ -z "$SKIP_C2" ]; then      # (SKIP_C2=1: only the workloads named in WORKLOADS)
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $CMD > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python3 $CMD > /dev/null 2> $OUT/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python3 $CMD > /dev/null 2> $OUT/write.log
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/sq -o s --output-format csv -- python3 $CMD > /dev/null 2> $OUT/sq.log
PROFILE_CMD="python3 $CMD" python3 profiles/summarise.py $TAG $OUT/trace $OUT/fetch $OUT/write $OUT/sq
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc_hbm.csv profiles/${TAG}_sq.csv profiles/pmc_traffic.json $OUT/
grep '^{' $OUT/bench_under_rocprof.json > $OUT/${TAG}_bench_under_rocprof.json || true
fi
for WL in ${WORKLOADS:-c2best c3 c3cf c4 c5 plain shapes}; do
    W="bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload $WL"
    rocprofv3 --kernel-trace --stats -d $OUT/trace_$WL -o t --output-format csv -- python3 $W > $OUT/${TAG}_${WL}_bench_under_rocprof.json 2> $OUT/trace_$WL.log
    PROFILE_CMD="python3 $W" python3 profiles/summarise.py --stats-only ${TAG}_$WL $OUT/trace_$WL
    cp profiles/${TAG}_${WL}_kernel_stats.csv $OUT/
    if true; then       # (plain included since round 3: its walks take milliseconds now)
        rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch_$WL -o f --output-format csv -- python3 $W > /dev/null 2> $OUT/fetch_$WL.log
        rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write_$WL -o w --output-format csv -- python3 $W > /dev/null 2> $OUT/write_$WL.log
        rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/sq_$WL -o s --output-format csv -- python3 $W > /dev/null 2> $OUT/sq_$WL.log
        PROFILE_CMD="python3 $W" python3 profiles/summarise.py --workload $TAG $WL $OUT/fetch_$WL $OUT/write_$WL $OUT/sq_$WL
        cp profiles/${TAG}_${WL}_pmc_hbm.csv profiles/${TAG}_${WL}_sq.csv profiles/pmc_traffic.json $OUT/
    fi
done
