#!/bin/bash
# build + run a pytest selection on the GPU box; log under gpurun_out/<name>.log      usage: gpu_pytest.sh <name> <pytest args...>
NAME=$1; shift
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout 1500 python -m pytest "$@" -q --tb=short -p no:cacheprovider > gpurun_out/$NAME.log 2>&1
echo "pytest exit $?" >> gpurun_out/$NAME.log
tail -5 gpurun_out/$NAME.log; grep -E "^(E |FAILED|ERROR)" gpurun_out/$NAME.log | head -40
