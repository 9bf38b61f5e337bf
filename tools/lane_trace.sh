#!/bin/bash
# Kernel traces of the exchange-lane leg and of the attention-only leg of tools/overlap_bench.py (rocprofv3 --kernel-trace), summarised
# by tools/overlap_trace.py; one layer of the lane leg as a timeline.  Run on the GPU box from the repo root; writes into $1 (default gpurun_out/lane).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$R/gpurun_out/lane}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for LEG in lane attention_on_compute_lane; do
  rm -rf "$OUT/tr_$LEG"
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr_$LEG" -o ov -- python3 "$R/tools/overlap_bench.py" --steps 3 --legs $LEG > "$OUT/trace_$LEG.log" 2>&1
  T=$(find "$OUT/tr_$LEG" -name "*kernel_trace.csv" | head -1)
  if [ -n "$T" ]; then
    python3 "$R/tools/overlap_trace.py" "$T" "$OUT/overlap_trace_$LEG.json" > /dev/null 2>&1
    [ "$LEG" = lane ] && python3 "$R/tools/overlap_layer_dump.py" "$T" 44 > "$OUT/lane_layer_timeline.txt" 2>&1
  fi
  rm -rf "$OUT/tr_$LEG"
done
# the low-rank preset's default path (factor chain on the compute lane, the peers' reconstructions on the exchange lane): one layer
rm -rf "$OUT/tr_lr"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr_lr" -o ov -- python3 "$R/tools/overlap_bench.py" --preset lowrank8 --steps 3 --legs default --quiet > "$OUT/trace_lowrank8.log" 2>&1
T=$(find "$OUT/tr_lr" -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python3 "$R/tools/lowrank_lane_dump.py" "$T" > "$OUT/lowrank_lane_layer_timeline.txt" 2>&1
rm -rf "$OUT/tr_lr"
cd "$R"
