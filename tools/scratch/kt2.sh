export TMPDIR=/tmp
out=gpurun_out/kt2; mkdir -p $out
for lds in 1 0; do
QE_FILL_LDS=$lds QE_EXP_FILL_DUMMY=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/p$lds -- python3 bench.py --workload quicked --no-cpu-baseline --no-e2e --no-strong --steps 3 --warmup 1 --sync-each-step > $out/log$lds.txt 2>&1
cp $out/p$lds/*/*kernel_trace.csv $out/trace$lds.csv
rm -rf $out/p$lds
done
