#!/bin/bash
# SQ counter pass (no trace domains) over the dense LrgNet evaluation: wave-cycle breakdown, MFMA busy cycles, LDS bank conflicts.
# usage (on the GPU box): tools/pmc_sq.sh <B> <out.csv>
B=${1:-68}; OUT=${2:-gpurun_out/pmc_sq_$B.csv}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcsq
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES \
    -d /tmp/pmcsq -o s --output-format csv -- python $R/tools/fwd_only.py $B fused 5 > /tmp/pmcsq.log 2>&1
f=$(find /tmp/pmcsq -name "*counter_collection.csv" | head -1)
python - "$f" "$R/$OUT" <<'PY'
import csv, sys
agg = {}
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:64]
    if 'lrg_' not in k or 'pack' in k:
        continue
    d = agg.setdefault(k, {})
    d.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
names = ['SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_LDS_BANK_CONFLICT',
         'SQ_LDS_IDX_ACTIVE', 'SQ_BUSY_CYCLES']
with open(sys.argv[2], 'w') as out:
    w = csv.writer(out)
    w.writerow(['kernel', 'launches'] + names + ['wait_any/wave', 'wait_inst/wave', 'active/wave', 'lds_conflict/lds_active'])
    for k, d in agg.items():
        n = len(d.get('SQ_WAVE_CYCLES', [0]))
        v = [sum(d.get(c, [0])) / max(1, n) for c in names]
        wave = max(v[0], 1.0)
        row = [k, n] + ['%.0f' % x for x in v] + ['%.3f' % (v[1] / wave), '%.3f' % (v[2] / wave), '%.3f' % (v[3] / wave), '%.4f' % (v[5] / max(v[6], 1.0))]
        w.writerow(row)
        print(row)
PY
