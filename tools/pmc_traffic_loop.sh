#!/bin/bash
# HBM traffic of the loop's kernels per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in passes of their own, corrected on
# a 256 MiB copy as tools/pmc_run.sh does).   usage (GPU box): tools/pmc_traffic_loop.sh [out.json]
R=$GRAFT_REPO_ROOT; OUT=${1:-gpurun_out/r02_traffic_loop.json}
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmct && mkdir -p /tmp/pmct
B="python $R/bench.py --steps 1 --warmup 6 --iters-per-step 256 --lanes 1 --graph 0 --cpu-seconds 0 --p0-rooms 0 --fixed-rooms 0"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmct/fetch -o f --output-format csv -- $B > /tmp/pmct/f.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmct/write -o w --output-format csv -- $B > /tmp/pmct/w.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmct/calib_fetch -o cf --output-format csv -- python $R/tools/pmc_calib.py > /tmp/pmct/cf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmct/calib_write -o cw --output-format csv -- python $R/tools/pmc_calib.py > /tmp/pmct/cw.log 2>&1
for d in fetch write calib_fetch calib_write; do f=$(find /tmp/pmct/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f /tmp/pmct/$d/ 2>/dev/null; done
python - /tmp/pmct "$R/$OUT" <<'PY'
import csv, glob, json, os, sys
def per_kernel(d, counter):
    out = {}
    f = glob.glob(os.path.join(d, '*counter_collection.csv'))[0]
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter:
            out.setdefault(r['Kernel_Name'], []).append(float(r['Counter_Value']))
    return out
root, outp = sys.argv[1], sys.argv[2]
MiB = 1 << 20
cf, cw = per_kernel(root + '/calib_fetch', 'FETCH_SIZE'), per_kernel(root + '/calib_write', 'WRITE_SIZE')
copyk = max(cf, key=lambda k: sum(cf[k]))
kf = 256 * MiB / (sum(cf[copyk]) / len(cf[copyk]) * 1024)
kw = 256 * MiB / (sum(cw[copyk]) / len(cw[copyk]) * 1024)
f, w = per_kernel(root + '/fetch', 'FETCH_SIZE'), per_kernel(root + '/write', 'WRITE_SIZE')
kernels, tr, tw = {}, 0.0, 0.0
for k in f:
    if 'lrg_' not in k or len(f[k]) < 500:       # the loop's kernels: launched once per iteration
        continue
    r = sum(f[k]) / len(f[k]) * 1024 * kf
    wr = sum(w.get(k, [0])) / max(1, len(w.get(k, [0]))) * 1024 * kw
    kernels[k[:70]] = dict(launches=len(f[k]), read_bytes_per_launch=r, write_bytes_per_launch=wr)
    tr += r; tw += wr
res = dict(source='tools/pmc_traffic_loop.sh: bench.py loop, 1 lane, 68 rooms in flight', fetch_correction=kf, write_correction=kw,
           read_bytes_per_iteration=tr, write_bytes_per_iteration=tw, hbm_bytes_per_iteration=tr + tw, kernels=kernels)
json.dump(res, open(outp, 'w'), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != 'kernels'}))
for k, v in kernels.items(): print('  %-72s read %8.0f KB  write %8.0f KB' % (k, v['read_bytes_per_launch'] / 1024, v['write_bytes_per_launch'] / 1024))
PY
