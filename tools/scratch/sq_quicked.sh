export TMPDIR=/tmp
out=gpurun_out/sqq; mkdir -p $out
for m in 1 0; do
export QE_FILL_MULTI=$m
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $out/p$m -- python3 bench.py --workload quicked --no-cpu-baseline --no-e2e --no-strong --steps 1 --warmup 0 --sync-each-step > $out/log$m.txt 2>&1
cp $out/p$m/*/*counter_collection.csv $out/sq_multi$m.csv
rm -rf $out/p$m
done
