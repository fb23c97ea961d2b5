export TMPDIR=/tmp
out=gpurun_out/kt; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -- python3 bench.py --workload quicked --no-cpu-baseline --no-e2e --no-strong --steps 3 --warmup 1 --sync-each-step > $out/log.txt 2>&1
cp $out/p/*/*kernel_stats.csv $out/stats.csv
rm -rf $out/p
