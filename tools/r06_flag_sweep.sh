#!/bin/bash
# round 6: the steady leg (68 rooms in flight) per set of hipcc flags in $1 (";"-separated; "-" = none): rebuilds the library on the GPU box for each
IFS=';' read -ra FL <<< "$1"
for f in "${FL[@]}"; do
  [ "$f" = "-" ] && f=""
  export LRG_HIPCC_FLAGS="$f"
  python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -5 gpurun_out/build.log; continue; }
  echo "== flags: $f"
  tools/r06_sweep.sh "X=1;X=2"
done
