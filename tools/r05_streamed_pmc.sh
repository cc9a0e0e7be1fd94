#!/bin/bash
# north_star's literal MLP metric on the current build: HBM bytes (PMC) / kernel time / 8 TB/s for the layer-STREAMED evaluation
# (`--net-mode streamed`: one lrg_pointwise_mfma_kernel launch per layer, activations through HBM), at B = 68 and B = 1088 instances.
# Separate --pmc passes (FETCH_SIZE | WRITE_SIZE), durations from a --kernel-trace pass of the same command, counters corrected on a 256 MiB copy
# (MI355X_MICROARCH.md, HBM section).   usage (GPU box): tools/r05_streamed_pmc.sh <commit> [out.json]
R=$GRAFT_REPO_ROOT; COMMIT=${1:-unknown}; OUT=${2:-gpurun_out/r05_traffic_streamed.json}; MODE=${MODE:-streamed}   # MODE=streamed-tiles: the layer launches on the fused tile
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmcs && mkdir -p /tmp/pmcs
N=10
timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmcs/calib_fetch -o cf --output-format csv -- python $R/tools/pmc_calib.py > /tmp/pmcs/cf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmcs/calib_write -o cw --output-format csv -- python $R/tools/pmc_calib.py > /tmp/pmcs/cw.log 2>&1
for B in 68 1088; do
  C="python $R/tools/fwd_only.py $B $MODE $((N-1))"
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pmcs/kt$B -o kt --output-format csv -- $C > /tmp/pmcs/kt$B.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmcs/fetch$B -o f --output-format csv -- $C > /tmp/pmcs/f$B.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmcs/write$B -o w --output-format csv -- $C > /tmp/pmcs/w$B.log 2>&1
  cp $(find /tmp/pmcs/kt$B -name "*kernel_stats.csv" | head -1) $R/${OUT%.json}_B${B}_kernel_stats.csv 2>/dev/null
done
python - /tmp/pmcs "$R/$OUT" "$COMMIT" $N $MODE <<'PY'
import csv, glob, json, os, sys
root, outp, commit, nfwd = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
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
res = dict(source='tools/r05_streamed_pmc.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE passes of their own over tools/fwd_only.py <B> %s; durations' % sys.argv[5] + ' from a --kernel-trace --stats pass',
           commit=commit, fetch_correction=kf, write_correction=kw, hbm_peak_GBps=8000.0, forwards_profiled=nfwd, batches={})
for B in (68, 1088):
    f, w = per_kernel('fetch%d' % B, 'FETCH_SIZE'), per_kernel('write%d' % B, 'WRITE_SIZE')
    ks = glob.glob(os.path.join(root, 'kt%d' % B, '**', '*kernel_stats.csv'), recursive=True)
    dur = {r['Name']: (float(r['TotalDurationNs']), int(r['Calls'])) for r in csv.DictReader(open(ks[0]))}
    kernels, tot_b, tot_t = {}, 0.0, 0.0
    for k in f:
        if 'lrg_' not in k or 'lrg_pack_weights' in k:
            continue
        rd, wr = sum(f[k]) * 1024 * kf, sum(w.get(k, [0])) * 1024 * kw
        t_ns, calls = dur.get(k, (0.0, 0))
        kernels[k[:64]] = dict(read_bytes_per_forward=rd / nfwd, write_bytes_per_forward=wr / nfwd, calls=calls, seconds_per_forward=t_ns * 1e-9 / nfwd,
                               hbm_GBps=(rd + wr) / (t_ns * 1e-9) / 1e9 if t_ns else None, frac_of_hbm_peak=(rd + wr) / (t_ns * 1e-9) / 8e12 if t_ns else None)
        tot_b += rd + wr; tot_t += t_ns * 1e-9
    res['batches'][str(B)] = dict(instances=B, hbm_bytes_per_forward=tot_b / nfwd, algorithmic_bytes_per_forward=10297344.0 * B, kernel_seconds_per_forward=tot_t / nfwd,
                                  all_kernels_frac_of_hbm_peak=tot_b / tot_t / 8e12 if tot_t else None, kernels=kernels)
json.dump(res, open(outp, 'w'), indent=1)
for B, b in res['batches'].items():
    for k, v in b['kernels'].items():
        print(B, k, 'frac_of_hbm_peak', v['frac_of_hbm_peak'], 'GB/s', v['hbm_GBps'])
PY
