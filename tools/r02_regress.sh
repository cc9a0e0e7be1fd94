#!/bin/bash
# the new regression tests on the current build; the busy-chip test on the previous commit's build (must fail there)
mkdir -p gpurun_out
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_configs.py -k "busy_chip or voxel_grid" -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -6
if [ -d _old ]; then
  cd _old
  python -c "import __graft_entry__ as g; g.build()" > $R/gpurun_out/build_old.log 2>&1 || { tail -20 $R/gpurun_out/build_old.log; exit 1; }
  echo "== previous commit"
  timeout 900 python -m pytest tests/test_gpu_configs.py -k "busy_chip" -m gpu -q --tb=line -p no:cacheprovider 2>&1 | tail -4 | cut -c1-300
fi
