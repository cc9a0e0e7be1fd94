#!/bin/bash
# HBM traffic and matrix-pipe occupancy of the free-running kernel (lrg_grow_async_kernel = the timed loop of bench.py), per launch:
# rocprofv3 --pmc passes of their own (FETCH_SIZE | WRITE_SIZE | SQ counters), the byte counters corrected on a 256 MiB copy as
# MI355X_MICROARCH.md prescribes (tools/pmc_calib.py), the launch duration from a --kernel-trace pass of the same command.
#   usage (GPU box): tools/pmc_free_run.sh <commit> [out.json]
R=$GRAFT_REPO_ROOT; COMMIT=${1:-unknown}; OUT=${2:-gpurun_out/r06_pmc_free_run.json}
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmcf && mkdir -p /tmp/pmcf
# (counter collection serialises kernels: the two kernels of a register-tile launch -- resident together, waiting for each other -- cannot run under it; the counters are
#  read on the ONE-kernel formulation of the same loop, LRG_FREE_RUN_WAVES=-1, and the file says so)
export LRG_FREE_RUN_WAVES=-1
B="python $R/bench.py --gpus 1 --steps 6 --warmup 4 --cpu-seconds 0 --p0-rooms 0 --named-configs 0 --fixed-rooms 0 --one-room-ks= --steady-slots="
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pmcf/kt -o kt --output-format csv -- $B > /tmp/pmcf/kt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmcf/fetch -o f --output-format csv -- $B > /tmp/pmcf/f.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmcf/write -o w --output-format csv -- $B > /tmp/pmcf/w.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/pmcf/sq -o s --output-format csv -- $B > /tmp/pmcf/s.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmcf/calib_fetch -o cf --output-format csv -- python $R/tools/pmc_calib.py > /tmp/pmcf/cf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmcf/calib_write -o cw --output-format csv -- python $R/tools/pmc_calib.py > /tmp/pmcf/cw.log 2>&1
cp $(find /tmp/pmcf/kt -name "*kernel_stats.csv" | head -1) $R/${OUT%.json}_kernel_stats.csv 2>/dev/null
python - /tmp/pmcf "$R/$OUT" "$COMMIT" "$R" <<'PY'
import csv, glob, json, os, sys
root, outp, commit, repo = sys.argv[1:5]
sys.path.insert(0, repo)
def per_kernel(d, counter):
    out = {}
    fs = glob.glob(os.path.join(root, d, '**', '*counter_collection.csv'), recursive=True)
    for r in csv.DictReader(open(fs[0])):
        if r['Counter_Name'] == counter:
            out.setdefault(r['Kernel_Name'], []).append(float(r['Counter_Value']))
    return out
MiB = 1 << 20
cf, cw = per_kernel('calib_fetch', 'FETCH_SIZE'), per_kernel('calib_write', 'WRITE_SIZE')
copyk = max(cf, key=lambda k: sum(cf[k]))
kf = 256 * MiB / (sum(cf[copyk]) / len(cf[copyk]) * 1024)
kw = 256 * MiB / (sum(cw[copyk]) / len(cw[copyk]) * 1024)
K = 'lrg_grow_async_kernel'
KW = 'lrg_grow_async_worker_kernel'      # (a two-kernel launch: the CUs that run tiles are a kernel of their own, resident beside the front workgroups' -- per launch = the sum of the two)
def mine(d, name=K):
    return [v for k, vs in d.items() if name in k for v in vs]
def per_launch(d):
    a, b = mine(d, K)[1:], mine(d, KW)[1:]      # (the launches of the timed steps: all of them but the first)
    return (sum(a) / len(a) if a else 0.0) + (sum(b) / len(b) if b else 0.0), len(a), len(b)
fs, ws = per_kernel('fetch', 'FETCH_SIZE'), per_kernel('write', 'WRITE_SIZE')
rd_kb, nf, nfw = per_launch(fs)
wr_kb, _, _ = per_launch(ws)
f = mine(fs)[1:]
rd, wr = rd_kb * 1024 * kf, wr_kb * 1024 * kw
sqd = {c: per_kernel('sq', c) for c in ('SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_BUSY_CYCLES')}
sq = {c: [per_launch(sqd[c])[0]] for c in sqd}
mf = sq['SQ_VALU_MFMA_BUSY_CYCLES'][0]
gui = max(sum(mine(sqd['GRBM_GUI_ACTIVE'], K)[1:]) / max(1, len(mine(sqd['GRBM_GUI_ACTIVE'], K)[1:])), 1.0)      # (the front kernel spans the launch)
launch_ms = None
ks = glob.glob(os.path.join(root, 'kt', '**', '*kernel_stats.csv'), recursive=True)
for r in csv.DictReader(open(ks[0])):
    if K in r['Name']:
        launch_ms = float(r['AverageNs']) * 1e-6
from learn_region_grow_amd import _lib
import bench
res = dict(formulation='one kernel, team tiles (LRG_FREE_RUN_WAVES=-1): rocprofv3 serialises kernels under --pmc, the two kernels of the default register-tile launch cannot be resident together there', source='tools/pmc_free_run.sh: rocprofv3 --pmc passes of their own over `bench.py --gpus 1 --steps 6 --warmup 4` (68 rooms in flight, 25 ms launches)',
           commit=commit, abi=_lib.load().lrg_abi_version(), kernel_sources_sha16=bench.kernel_sources_sha16(), kernel=K + (' + ' + KW if nfw else ''), launches_profiled=len(f), worker_kernel_launches_profiled=nfw, fetch_correction=kf, write_correction=kw,
           read_bytes_per_launch=rd, write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr, launch_ms=launch_ms,
           hbm_GBps=(rd + wr) / (launch_ms * 1e-3) / 1e9 if launch_ms else None,
           frac_of_hbm_peak=(rd + wr) / (launch_ms * 1e-3) / 1e9 / 8000.0 if launch_ms else None,
           mfma_busy_cycles_per_launch=mf, gui_active_cycles_per_launch=gui, mfma_util_chipwide=mf / (gui / 8 * 1024),
           mfma_definition='SQ_VALU_MFMA_BUSY_CYCLES (SIMD cycles, all XCDs) / (GRBM_GUI_ACTIVE (summed over the 8 XCDs by rocprofv3) / 8 x 1024 SIMDs)',
           wave_cycles_per_launch=sum(sq['SQ_WAVE_CYCLES']) / max(1, len(sq['SQ_WAVE_CYCLES'])), wait_any_per_launch=sum(sq['SQ_WAIT_ANY']) / max(1, len(sq['SQ_WAIT_ANY'])))
json.dump(res, open(outp, 'w'), indent=1)
print(json.dumps(res))
PY
