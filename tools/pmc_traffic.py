#!/usr/bin/env python3
"""Turn two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE) of tools/fwd_only.py and of tools/pmc_calib.py into HBM
bytes per lrg_forward call, with the gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE counts half of a wide
coalesced read: the factor is measured on the calibration copy, as is the WRITE_SIZE factor).
usage: pmc_traffic.py <dir with fetch/ write/ calib_fetch/ calib_write/ subdirs> <forwards per run> <out.json>"""
import csv, glob, json, os, sys


def per_kernel(d, counter):
    out = {}
    f = glob.glob(os.path.join(d, '*counter_collection.csv'))[0]
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter:
            out.setdefault(r['Kernel_Name'], []).append(float(r['Counter_Value']))
    return out


def main():
    root, nfwd, outp = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    MiB = 1 << 20
    cf = per_kernel(os.path.join(root, 'calib_fetch'), 'FETCH_SIZE')
    cw = per_kernel(os.path.join(root, 'calib_write'), 'WRITE_SIZE')
    copyk = max(cf, key=lambda k: sum(cf[k]))           # the big copy kernel
    fetch_kb = sum(cf[copyk]) / len(cf[copyk])
    write_kb = sum(cw[copyk]) / len(cw[copyk])
    kf = 256 * MiB / (fetch_kb * 1024)                  # true bytes per reported byte
    kw = 256 * MiB / (write_kb * 1024)
    f = per_kernel(os.path.join(root, 'fetch'), 'FETCH_SIZE')
    w = per_kernel(os.path.join(root, 'write'), 'WRITE_SIZE')
    kernels = {}
    tot_r = tot_w = 0.0
    for k in f:
        if 'lrg_' not in k or 'lrg_pack_weights' in k:      # (the one-time operand image is not part of a forward)
            continue
        r = sum(f[k]) * 1024 * kf / nfwd
        wr = sum(w.get(k, [0])) * 1024 * kw / nfwd
        kernels[k[:60]] = dict(read_bytes=r, write_bytes=wr)
        tot_r += r; tot_w += wr
    res = dict(read_bytes_per_forward=tot_r, write_bytes_per_forward=tot_w, hbm_bytes_per_forward=tot_r + tot_w,
               fetch_correction=kf, write_correction=kw, calibration='256 MiB copy: FETCH_SIZE %.0f KB, WRITE_SIZE %.0f KB reported' % (fetch_kb, write_kb),
               forwards_profiled=nfwd, kernels=kernels)
    json.dump(res, open(outp, 'w'), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != 'kernels'}))


if __name__ == '__main__':
    main()
