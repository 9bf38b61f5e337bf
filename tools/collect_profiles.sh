#!/bin/bash
# Reproduces every file under profiles/ on a 1-GPU MI355X box.  Run from the repo root:
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/collect_profiles.sh'
# then copy gpurun_out/r01/* into profiles/ (names below).  Counter passes are separate runs (FETCH_SIZE and WRITE_SIZE do
# not fit one pass) and are never combined with trace domains other than the kernel trace.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r01
mkdir -p "$OUT"; rm -rf "$OUT"/*
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats of the default bench command  -> r01_bench_kernel_stats.csv, r01_bench_kernel_durations.json
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python3 "$R/bench.py" --steps 20 --warmup 3 > "$OUT/bench_under_rocprof.log" 2>&1
# 2. HBM traffic of the same command (short run + the copy probe used for calibration)  -> pmc_traffic.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --copy-probe 8 > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --copy-probe 8 > "$OUT/pmc_write.log" 2>&1
cd "$R"
python tools/pmc_summary.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_traffic.json" > /dev/null
python tools/trace_kernel_avg.py "$OUT/trace/bench_kernel_trace.csv" "$OUT/r01_bench_kernel_durations.json" > /dev/null
rm -f "$OUT/trace/bench_kernel_trace.csv"                       # tens of MB; the two summaries above are what is kept
cp "$OUT/trace/bench_kernel_stats.csv" "$OUT/r01_bench_kernel_stats.csv"
# the bench reads profiles/pmc_traffic.json and profiles/r01_bench_kernel_durations.json: refresh them before the plain runs
cp "$OUT/pmc_traffic.json" profiles/pmc_traffic.json
cp "$OUT/r01_bench_kernel_durations.json" profiles/r01_bench_kernel_durations.json
# 3. the bench lines themselves
python bench.py > "$OUT/r01_bench_n1.json" 2>/dev/null
python bench.py --replay inorder > "$OUT/r01_bench_n1_inorder.json" 2>/dev/null
# 4. per-codec, per-configuration and low-rank tables
python tools/codec_table.py 2>&1 | grep "^|" > "$OUT/r01_codec_table.md"
python tools/config_table.py 2>&1 | grep "^|" > "$OUT/r01_config_table.md"
python tools/lowrank_bench.py 2>&1 | grep -v amdgpu.ids > "$OUT/r01_lowrank_bench.txt"
tail -c 400 "$OUT/r01_bench_n1.json"
