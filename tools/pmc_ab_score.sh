#!/bin/bash
# SQ counters of two scoring kernel variants on the benchmark's batch (tools/ab_score.py), one pass of 8
# counters, mean per dispatch by kernel.  Usage (through gpurun): bash tools/pmc_ab_score.sh wide mfma
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_ab_score
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export AB_PAIRS=${AB_PAIRS:-127}
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/p1 -- python3 $REPO/tools/ab_score.py "$@" > $OUT/p1.log 2>&1 || { echo p1 failed; tail -5 $OUT/p1.log; exit 1; }
tail -3 $OUT/p1.log
python3 - <<'PY'
import csv,glob,collections,os
out=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/pmc_ab_score"
for f in glob.glob(f"{out}/p1/*/*_counter_collection.csv"):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        m = __import__("re").search(r"(score_\w+?kernel(<\w+>)?)", n)
        if m:
            agg[(m.group(1),r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()):
        print(k[0],k[1],len(v),"%.5g"%(sum(v)/len(v)))
PY
