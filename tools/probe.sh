cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in none; do
  if [ $v = none ]; then unset CFX_FUSED_DBG; else export CFX_FUSED_DBG=$v; fi
  for rows in 32; do
    export CFX_STATS_ROWS=$rows
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_${v}_$rows -o b -- python3 $R/tools/fused_probe.py 300 > /dev/null 2>&1
    echo "== dbg=$v rows=$rows"; python3 -c "
import csv,sys
for r in csv.DictReader(open('$R/gpurun_out/p_${v}_$rows/b_kernel_stats.csv')):
    if 'absmean' in r['Name'] or 'binary' in r['Name']: print('  %-40s calls %4s avg %8.2f us min %8.2f max %8.2f' % (r['Name'][:40], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
"
    rm -rf $R/gpurun_out/p_${v}_$rows
  done
done
