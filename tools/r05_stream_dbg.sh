#!/bin/bash
# what bounds the streaming layer kernels: per-layer durations with parts of the kernel switched off (LRG_STREAM_DBG: 1 no stores, 2 ring not refilled, 4 no B reads)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for DBG in ${DBGS:-0 1 2 4 7}; do
  echo "== LRG_STREAM_DBG=$DBG (variant ${LRG_LAYER_VARIANT:-default})"
  rm -rf /tmp/lv; LRG_STREAM_DBG=$DBG timeout 300 rocprofv3 --kernel-trace -d /tmp/lv -o t --output-format csv -- python $R/tools/fwd_only.py 1088 streamed-tiles 3 > /dev/null 2>&1
  python - $(find /tmp/lv -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'lrg_' in r['Kernel_Name'] and 'pack' not in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
per = len(rows) // 4
print('   ' + '  '.join('%s %.0f' % (r['Kernel_Name'].split('<')[1].split('>')[0].replace(', ', ',') if '<' in r['Kernel_Name'] else r['Kernel_Name'][4:10], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows[-per:]))
PY
done
