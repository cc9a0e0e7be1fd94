"""Round 6: region_grow.py on two small rooms, repeated: return codes and the per-region lines must not change from run to run.  python tools/r06_cli_soak.py [runs]"""
import os, subprocess, sys, tempfile, pathlib, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_gpu_cli import _write_inputs
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
tmp = pathlib.Path(tempfile.mkdtemp())
h5, prefix, weights = _write_inputs(tmp, 2)
cmd = [sys.executable, os.path.join(ROOT, 'region_grow.py'), '--h5', h5, '--ckpt', prefix, '--policy', 'gt', '--seed', '6']
seen = {}
for k in range(runs):
    r = subprocess.run(cmd + (['--timing'] if k % 2 else []), capture_output=True, text=True, cwd=str(tmp), timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith(('room ', 'Area '))]
    key = (r.returncode, hashlib.md5('\n'.join(lines).encode()).hexdigest())
    seen.setdefault(key, []).append(k)
    if r.returncode != 0:
        print('run', k, 'rc', r.returncode, r.stderr[-600:], flush=True)
print({str(k): v for k, v in seen.items()})
