#!/bin/bash
# SQ counters of the scoring kernels in the headline step of bench.py (two passes of 8 counters,
# each a run of its own), mean per dispatch by kernel.  Usage (through gpurun): bash tools/pmc_score.sh
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_score
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SLAM_BENCH_STREAMS=1 SLAM_BENCH_RANSAC=0 SLAM_BENCH_INGEST=0 SLAM_BENCH_FULLSEQ=0 SLAM_BENCH_STRESS=0 SLAM_BENCH_CONFIG0=0
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES --output-format csv -d $OUT/p1 -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p1.log 2>&1 || { echo p1 failed; tail -5 $OUT/p1.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_WAVES --output-format csv -d $OUT/p2 -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p2.log 2>&1 || { echo p2 failed; tail -5 $OUT/p2.log; exit 1; }
python3 - <<'PY'
import csv,glob,collections,os
out=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/pmc_score"
for p in ("p1","p2"):
    for f in glob.glob(f"{out}/{p}/*/*_counter_collection.csv"):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            n=r["Kernel_Name"]
            if "score_" in n:
                agg[(n.split("::")[-1].split("(")[0],r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k,v in sorted(agg.items()):
            print(k[0],k[1],len(v),"%.4g"%(sum(v)/len(v)))
PY
