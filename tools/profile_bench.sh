#!/bin/bash
# Profiles `bench.py` on the GPU box: one kernel-trace/stats pass, two HBM-traffic PMC passes
# (FETCH_SIZE, WRITE_SIZE) and two instruction-issue PMC passes (8 SQ counters each), every pass a
# run of its own, then summarises them into gpurun_out/profiles_<tag>/ (copy what is to be kept
# into profiles/).  Usage (through gpurun):   bash tools/profile_bench.sh r02
#
# SLAM_BENCH_RANSAC=0 SLAM_BENCH_INGEST=0 SLAM_BENCH_FULLSEQ=0 SLAM_BENCH_STRESS=0: only the headline step runs, so every
# dispatch of a kernel is the headline launch; SLAM_BENCH_STREAMS=1: one batch in flight, so that a
# kernel's duration is its own and not that of two kernels sharing the chip; 60 steps so that the
# first, cold launch of each kernel does not weigh on the average.
set -u
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export SLAM_BENCH_STREAMS=1 SLAM_BENCH_RANSAC=0 SLAM_BENCH_INGEST=0 SLAM_BENCH_FULLSEQ=0 SLAM_BENCH_STRESS=0 SLAM_BENCH_CONFIG0=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --steps 60 --warmup 3 --no-cpu-baseline > "$OUT/bench_trace.log" 2>&1 || { echo "trace pass failed"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_fetch.log" 2>&1 || { echo "fetch pass failed"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_write.log" 2>&1 || { echo "write pass failed"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_issue1" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_issue1.log" 2>&1 || { echo "issue pass 1 failed"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_issue2" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_issue2.log" 2>&1 || { echo "issue pass 2 failed"; exit 1; }
python3 "$REPO/tools/summarize_profile.py" "$OUT" "$REPO/gpurun_out/profiles_$TAG" "$TAG"
