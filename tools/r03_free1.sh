#!/bin/bash
# first run of the free-running launches: their own tests (strict time limits: a lost hand-over must show as an error, not a hang),
# then the lock-step suites that the refactoring of the tile / front code touched
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 420 python -m pytest tests/test_gpu_free_run.py -m gpu -x -q --tb=short -p no:cacheprovider > gpurun_out/r03_free1.log 2>&1
echo "free-run pytest exit $?" >> gpurun_out/r03_free1.log
tail -30 gpurun_out/r03_free1.log
timeout 400 python -m pytest tests/test_gpu_net.py tests/test_gpu_grow.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r03_free1_regress.log 2>&1
echo "regress pytest exit $?" >> gpurun_out/r03_free1_regress.log
tail -5 gpurun_out/r03_free1_regress.log
