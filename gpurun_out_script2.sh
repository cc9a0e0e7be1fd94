mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --maxfail=40 --tb=short -p no:cacheprovider 2>&1 | tail -2
for z in 1 2 4; do echo "LRG_SPLIT=$z"; LRG_SPLIT=$z timeout 900 python bench.py --steps 1500 --warmup 100 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['in_loop'])"; done
