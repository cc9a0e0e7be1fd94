#!/bin/bash
# layer-streamed LrgNet evaluation on the fused tile: per-layer kernel durations of the instantiation variants (LRG_LAYER_VARIANT 0 | 1 | 2) at B = 1088
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for V in ${VARIANTS:-2 3 4}; do
  echo "== variant $V"; LRG_LAYER_VARIANT=$V python $R/tools/fwd_only.py 1088 streamed-tiles 20
  rm -rf /tmp/lv$V; LRG_LAYER_VARIANT=$V timeout 300 rocprofv3 --kernel-trace -d /tmp/lv$V -o t --output-format csv -- python $R/tools/fwd_only.py 1088 streamed-tiles 3 > /dev/null 2>&1
  python - $(find /tmp/lv$V -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'lrg_' in r['Kernel_Name'] and 'pack' not in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
per = len(rows) // 4
for r in rows[-per:]:
    print('   %-70s %8.1f us' % (r['Kernel_Name'][:70], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
PY
done
