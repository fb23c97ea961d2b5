cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06_a
python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r06_a/gputest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_a/gputest.log
one="--no-workloads --no-strong --indel-pairs 0 --no-cpu-baseline --no-e2e --cfg5-pairs 0 --cfg4-pairs 0 --mixed-share 0"
for w in 2 3 2 3; do
  QE_SCORE_WAVES=$w python bench.py $one --steps 20 --warmup 3 > gpurun_out/r06_a/ab_waves${w}_$RANDOM.json 2>> gpurun_out/r06_a/ab.err
done
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_a/bench_line.json 2> gpurun_out/r06_a/bench_line.err ) 2> gpurun_out/r06_a/bench_line_time.txt
