"""GPU: the register tiles of csrc/lrg_wave_tile.inl against a host evaluation in the MFMA formulation's summation order, bit for bit.

The free-running launches' tiles keep a layer's accumulators in registers as the next layer's MFMA operands (the product computed transposed).  Every output must be
the chain of float32 FMAs the team tiles compute -- per output k = 8g + 0, 4, 1, 5, 2, 6, 3, 7 over the k-groups, then the bias, then the ReLU
(learn_region_grow_util.py:106-123, :138-162; lrg_fused_tile.inl) -- or a Bernoulli draw within an ulp of its confidence flips once in millions of steps, which no
label comparison would catch.  tools/wave_tile_probe.hip evaluates 18 tiles of random rows with every tile form (one-wavefront PREFIX / POOL tasks, the team-of-four
branch tile, the team-of-four head tile incl. the 2-wide last layer's eight-lane butterfly) and compares conv[1], layer 3, the pooled maxima and the logits with a
host loop of fmaf in that order."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, 'tools', 'build', 'wave_tile_probe')


def test_register_tiles_equal_a_host_fma_chain(cuda_device):
    if not os.path.exists(PROBE):      # (built by __graft_entry__.build(); here for a tree that was not)
        os.makedirs(os.path.dirname(PROBE), exist_ok=True)
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-I', os.path.join(ROOT, 'learn_region_grow_amd', 'csrc'),
                               os.path.join(ROOT, 'tools', 'wave_tile_probe.hip'), '-o', PROBE], stderr=subprocess.DEVNULL)
    r = subprocess.run([PROBE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    out = r.stdout
    m = re.search(r'conv\[1\]: (\d+) of (\d+) values differ from the host chain; layer 3: (\d+) of (\d+); pooled: (\d+) of (\d+)', out)
    assert m and int(m.group(1)) == 0 and int(m.group(3)) == 0 and int(m.group(5)) == 0 and int(m.group(2)) == 36864 and int(m.group(6)) == 3072, out
    m = re.search(r'register tile \(team of four\): conv\[1\] (\d+) of (\d+) differ; pooled (\d+) of (\d+)', out)
    assert m and int(m.group(1)) == 0 and int(m.group(3)) == 0, out
    m = re.search(r'register head tile: logits (\d+) of (\d+) differ', out)
    assert m and int(m.group(1)) == 0 and int(m.group(2)) == 1152, out
