#!/bin/bash
# Reproduces the round-2 files under profiles/ on a 1-GPU MI355X box.  Run from the repo root:
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh'
# then copy gpurun_out/r02/* into profiles/.  Counter passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass)
# and are never combined with trace domains other than the kernel trace; the profiled program follows `--` directly.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02
mkdir -p "$OUT"; rm -rf "$OUT"/*
KEY=$(python3 "$R/bench.py" --print-config-key)
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats of the default bench command (the in-order replay, one gated launch per layer)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python3 "$R/bench.py" --steps 20 --warmup 3 > "$OUT/bench_under_rocprof.log" 2>&1
# 2. HBM traffic of the same command (short run + the copy probe used for calibration)
PSTEPS=3
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-secondary --copy-probe 8 > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-secondary --copy-probe 8 > "$OUT/pmc_write.log" 2>&1
cd "$R"
python3 tools/pmc_summary.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/r02_pmc_traffic.json" "$KEY" $PSTEPS > /dev/null
python3 tools/trace_kernel_avg.py "$OUT/trace/bench_kernel_trace.csv" "$OUT/r02_bench_kernel_durations.json" "$KEY" > /dev/null
rm -f "$OUT/trace/bench_kernel_trace.csv"                       # tens of MB; the two summaries above are what is kept
cp "$OUT/trace/bench_kernel_stats.csv" "$OUT/r02_bench_kernel_stats.csv"
rm -rf "$OUT/trace" "$OUT/pmc_fetch" "$OUT/pmc_write"
# the bench reads profiles/r02_pmc_traffic.json and profiles/r02_bench_kernel_durations.json: refresh them before the plain runs
cp "$OUT/r02_pmc_traffic.json" profiles/r02_pmc_traffic.json
cp "$OUT/r02_bench_kernel_durations.json" profiles/r02_bench_kernel_durations.json
# 3. the bench lines themselves: headline (in order), 2-bit preset, cross-layer pipeline (upper bound, not deployable)
python3 bench.py > "$OUT/r02_bench_n1.json" 2>/dev/null
python3 bench.py --codec int2 > "$OUT/r02_bench_n1_int2.json" 2>/dev/null
python3 bench.py --replay pipelined > "$OUT/r02_bench_n1_pipelined.json" 2>/dev/null
python3 bench.py --own-ef ride --no-cpu-baseline > "$OUT/r02_bench_n1_two_launches.json" 2>/dev/null
python3 bench.py --own-ef inline --no-cpu-baseline > "$OUT/r02_bench_n1_inline_ef.json" 2>/dev/null
# 4. the deployable path with real attention (SURVEY 8d protocol 2), the compress launch's phase timeline
python3 tools/overlap_bench.py --steps 20 --json "$OUT/r02_overlap.json" > /dev/null 2>&1
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d "$OUT/ovtrace" -o ov -- python3 "$R/tools/overlap_bench.py" --steps 3 --json /tmp/ov_prof.json > /dev/null 2>&1)
OVT=$(find "$OUT/ovtrace" -name "*kernel_trace.csv" | head -1)
[ -n "$OVT" ] && python3 tools/overlap_trace.py "$OVT" "$OUT/r02_overlap_trace.json" > /dev/null 2>&1
rm -rf "$OUT/ovtrace"
python3 tools/fused_stamps.py 2>&1 | grep -v amdgpu.ids > "$OUT/r02_compress_timeline.txt"
python3 tools/gated_stamps.py 2>&1 | grep -v amdgpu.ids > "$OUT/r02_gated_layer_timeline.txt"
# 5. per-codec, per-configuration and low-rank tables
python3 tools/codec_table.py 2>&1 | grep "^|" > "$OUT/r02_codec_table.md"
python3 tools/config_table.py 2>&1 | grep "^|" > "$OUT/r02_config_table.md"
python3 tools/lowrank_bench.py 2>&1 | grep -v amdgpu.ids > "$OUT/r02_lowrank_bench.txt"
tail -c 600 "$OUT/r02_bench_n1.json"
