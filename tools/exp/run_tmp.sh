cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for a in 1.0 0.5; do
OUT=$R/gpurun_out/prof/fade_$a
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o kt --output-format csv -- python $R/bench.py --no-cpu --no-per-depth --no-configs --no-kernel-timing --prime 5 --steps 10 --warmup 3 --alpha $a > $OUT/log.txt 2>&1
cp $(find $OUT -name kt_kernel_stats.csv | head -1) $R/gpurun_out/fade_stats_$a.csv
rm -rf $OUT
done
