set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/lrprof; mkdir -p $OUT; rm -rf $OUT/*
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/lowrank_traffic.py > $OUT/plain.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o lr -- python3 $R/tools/lowrank_traffic.py > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o pmc -- python3 $R/tools/lowrank_traffic.py > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o pmc -- python3 $R/tools/lowrank_traffic.py > $OUT/write.log 2>&1
cd $R
python3 tools/pmc_summary.py $OUT/fetch $OUT/write $OUT/r06_lowrank_pmc_traffic.json > /dev/null 2>&1
ST=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$ST" ] && cp $ST $OUT/r06_lowrank_kernel_stats.csv
rm -rf $OUT/trace $OUT/fetch $OUT/write
cat $OUT/plain.txt | grep -v amdgpu
