#!/bin/bash
# SQ counter pass over a short run of the grow loop (bench.py, one lane): per-kernel wave-cycle breakdown and MFMA busy cycles
R=$GRAFT_REPO_ROOT; OUT=${1:-gpurun_out/pmc_sq_loop.csv}
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmcl
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    -d /tmp/pmcl -o s --output-format csv -- python $R/bench.py --steps 200 --warmup 30 --lanes 1 --cpu-seconds 0 --p0-rooms 0 > /tmp/pmcl.log 2>&1
f=$(find /tmp/pmcl -name "*counter_collection.csv" | head -1)
python - "$f" "$R/$OUT" <<'PY'
import csv, sys
agg = {}
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:60]
    if 'lrg_' not in k: continue
    agg.setdefault(k, {}).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
names = ['SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'GRBM_GUI_ACTIVE']
with open(sys.argv[2], 'w') as out:
    w = csv.writer(out)
    w.writerow(['kernel', 'launches'] + [n + '_per_launch' for n in names] + ['mfma_busy/(gui_active*1024 simds)'])
    for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get('GRBM_GUI_ACTIVE', [0]))):
        n = len(d.get('SQ_WAVE_CYCLES', [0]))
        v = [sum(d.get(c, [0])) / max(1, n) for c in names]
        row = [k, n] + ['%.0f' % x for x in v] + ['%.3f' % (v[4] / max(1.0, v[6] * 1024))]
        w.writerow(row); print(row)
PY
