cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06_e
python -m pytest tests -m gpu -x -q > gpurun_out/r06_e/gputest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_e/gputest.log
one="--no-strong --indel-pairs 0 --no-cpu-baseline --no-e2e --cfg5-pairs 0 --mixed-share 0"
python bench.py $one --steps 20 --warmup 3 > gpurun_out/r06_e/line_cfg4_quicked.json 2> gpurun_out/r06_e/line.err
python tests/soak_long.py 9200 6 > gpurun_out/r06_e/soak_long.txt 2>&1
