mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -rs 2>&1 | tail -15
