set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_knn
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $OUT/p1 -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p1.log 2>&1 || { echo p1 failed; tail -5 $OUT/p1.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p2 -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p2.log 2>&1 || { echo p2 failed; tail -5 $OUT/p2.log; exit 1; }
python3 - <<'PY'
import csv,glob,collections,os
out=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/pmc_knn"
for p in ("p1","p2"):
    for f in glob.glob(f"{out}/{p}/*/*_counter_collection.csv"):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            n=r["Kernel_Name"]
            if "hamming_mfma" in n or "hamming_fp4" in n or "score_mixed" in n:
                agg[(("knn" if "hamming" in n else "score"),r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k,v in sorted(agg.items()):
            print(k[0],k[1],len(v),"%.4g"%(sum(v)/len(v)))
PY
