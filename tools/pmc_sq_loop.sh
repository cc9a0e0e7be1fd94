#!/bin/bash
# SQ counter pass over a short run of the grow loop (bench.py, one lane): per-kernel wave-cycle breakdown and MFMA busy cycles
R=$GRAFT_REPO_ROOT; OUT=${1:-gpurun_out/pmc_sq_loop.csv}
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmcl
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    -d /tmp/pmcl -o s --output-format csv -- python $R/bench.py --steps 1 --warmup 6 --iters-per-step 256 --lanes 1 --graph 0 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0 > /tmp/pmcl.log 2>&1
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
    w.writerow(['kernel', 'launches'] + [n + '_per_launch' for n in names] + ['mfma_busy/(gui_active/8 xcds*1024 simds)'])
    for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get('GRBM_GUI_ACTIVE', [0]))):
        n = len(d.get('SQ_WAVE_CYCLES', [0]))
        v = [sum(d.get(c, [0])) / max(1, n) for c in names]
        row = [k, n] + ['%.0f' % x for x in v] + ['%.3f' % (v[4] / max(1.0, v[6] / 8 * 1024))]
        w.writerow(row); print(row)
# the loop's five kernels: chip-wide matrix-pipe utilisation per iteration = sum of MFMA busy cycles / (sum of GPU-active cycles x 1024 SIMDs)
import json
loop = [k for k in agg if any(t in k for t in ('lrg_front_greedy', 'lrg_front_big', 'lrg_head_gemm')) or ('lrg_fused_stack_kernel' in k and 'true>' in k.replace(' ', ''))]
loop = [k for k in agg if len(agg[k].get('GRBM_GUI_ACTIVE', [])) > 500]
per = {}
tot_mfma = tot_gui = 0.0
for k in loop:
    d = agg[k]; n = len(d['GRBM_GUI_ACTIVE'])
    m, g = sum(d.get('SQ_VALU_MFMA_BUSY_CYCLES', [0])) / n, sum(d['GRBM_GUI_ACTIVE']) / n
    per[k] = {'launches': n, 'gui_active_cycles': g, 'mfma_busy_cycles': m, 'mfma_util_chipwide': m / (g / 8 * 1024),
              'wave_cycles': sum(d.get('SQ_WAVE_CYCLES', [0])) / n, 'wait_any': sum(d.get('SQ_WAIT_ANY', [0])) / n,
              'active_inst_any': sum(d.get('SQ_ACTIVE_INST_ANY', [0])) / n}
    tot_mfma += m; tot_gui += g
json.dump({'source': 'tools/pmc_sq_loop.sh: rocprofv3 --pmc (SQ counters, their own pass), bench.py loop, 1 lane, 68 rooms in flight, kernels launched > 500 times',
           'definition': 'SQ_VALU_MFMA_BUSY_CYCLES (SIMD cycles, all XCDs) / (GRBM_GUI_ACTIVE (summed over the 8 XCDs by rocprofv3) / 8 x 1024 SIMDs)',
           'mfma_util_chipwide': tot_mfma / (tot_gui / 8 * 1024), 'gui_active_cycles_per_iteration': tot_gui, 'mfma_busy_cycles_per_iteration': tot_mfma,
           'round_1_same_definition': {'source': 'profiles/r01_pmc_sq_loop.csv (nine-launch iteration, every slot tiled on its own)',
                                       'mfma_util_chipwide': 146011534.0 / (5337588.0 / 8 * 1024), 'gui_active_cycles_per_iteration': 5337588.0,
                                       'mfma_busy_cycles_per_iteration': 146011534.0},
           'kernels': per}, open(sys.argv[2].replace('.csv', '.json'), 'w'), indent=1)
print('chip-wide MFMA utilisation over the loop kernels: %.3f' % (tot_mfma / (tot_gui / 8 * 1024)))
PY
