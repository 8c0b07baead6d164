#!/bin/bash
# (SLAM_BENCH_RANSAC=0 SLAM_BENCH_INGEST=0: only the headline step is traced, so per-kernel averages are the
# headline launches; SLAM_BENCH_STREAMS=1: one batch in flight, so that a kernel's duration is its own and not
# that of two kernels sharing the chip; 60 steps so that the first, cold launch of each kernel does not weigh on the average.)
# Profiles `bench.py` on the GPU box: one kernel-trace/stats pass and two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE), then summarises them into profiles/.  Usage (through gpurun):
#   bash tools/profile_bench.sh r01
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SLAM_BENCH_STREAMS=1 SLAM_BENCH_RANSAC=0 SLAM_BENCH_INGEST=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --steps 60 --warmup 3 --no-cpu-baseline > "$OUT/bench_trace.log" 2>&1 || { echo "trace pass failed"; exit 1; }
SLAM_BENCH_STREAMS=1 SLAM_BENCH_RANSAC=0 SLAM_BENCH_INGEST=0 timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_fetch.log" 2>&1 || { echo "fetch pass failed"; exit 1; }
SLAM_BENCH_STREAMS=1 SLAM_BENCH_RANSAC=0 SLAM_BENCH_INGEST=0 timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_write.log" 2>&1 || { echo "write pass failed"; exit 1; }
python3 "$REPO/tools/summarize_profile.py" "$OUT" "$REPO/gpurun_out/profiles_$TAG" "$TAG"
