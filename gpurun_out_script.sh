mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
for B in 68 1088; do python tools/fwd_only.py $B fused 10; done 2>&1 | grep forward
timeout 600 python -m pytest tests/test_gpu_net.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -E "^E|passed|failed|FAILED" | head -20
