mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --maxfail=40 --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log; grep -E "^E " gpurun_out/pytest_gpu.log | head -10
timeout 900 python bench.py --steps 1500 --warmup 100 --cpu-seconds 0 > gpurun_out/bench_gt.log 2>&1; tail -1 gpurun_out/bench_gt.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1f -o bench --output-format csv -- python $R/bench.py --steps 300 --warmup 50 --cpu-seconds 0 > $R/gpurun_out/prof_bench.log 2>&1
