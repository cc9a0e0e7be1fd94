"""CPU: the C-ABI library builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports every
symbol include/lrg_hip.h declares, with the struct layouts the Python mirror expects.  No compute calls."""
import ctypes
import os
import re

from conftest import REPO


def declared_functions():
    src = open(os.path.join(REPO, 'include', 'lrg_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    names = re.findall(r'^\s*(?:int|size_t|const char \*|const float \*)\s*\*?\s*(lrg_\w+)\s*\(', src, flags=re.M)
    return sorted(set(names))


def test_library_exports_every_declared_symbol(hip_lib):
    from learn_region_grow_amd import _lib
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(hip_lib, n), 'liblrg_hip.so does not export %s' % n
    assert sorted(_lib.EXPORTS) == names, 'Python binding and header disagree'


def test_abi_and_struct_layout(hip_lib):
    from learn_region_grow_amd import _lib
    header = int(re.search(r'#define\s+LRG_ABI_VERSION\s+(\d+)', open(os.path.join(REPO, 'include', 'lrg_hip.h')).read()).group(1))
    assert hip_lib.lrg_abi_version() == header == _lib.LRG_ABI_VERSION
    # the binding INTEGRATION.md shows asserts the same number
    doc = open(os.path.join(REPO, 'INTEGRATION.md')).read()
    quoted = [int(m) for m in re.findall(r'lrg_abi_version\(\)\s*==\s*(\d+)', doc)]
    assert quoted and all(q == header for q in quoted), 'INTEGRATION.md asserts ABI %s, the header says %d' % (quoted, header)
    assert hip_lib.lrg_target_arch() == b'gfx950'
    for which, st in enumerate((_lib.LrgWeights, _lib.LrgRoom, _lib.LrgSlot, _lib.LrgGrowParams, _lib.LrgStepBuffers,
                               _lib.LrgPackedBuffers, _lib.LrgBeamGroup, _lib.LrgAsyncBuffers, _lib.LrgFillJob)):
        assert hip_lib.lrg_struct_size(which) == ctypes.sizeof(st)


def test_workspace_arithmetic_is_host_only(hip_lib):
    from learn_region_grow_amd import _lib
    w = _lib.LrgWeights()
    w.feature_size, w.n_conv, w.n_head = 13, 5, 3
    for i, c in enumerate([64, 64, 64, 128, 512]):
        w.conv_ch[i] = c
    for i, c in enumerate([256, 128, 2]):
        w.head_ch[i] = c
    nbytes = hip_lib.lrg_forward_workspace_bytes(ctypes.byref(w), 4, 512, 512)
    # 2 branches x 512 rows x (64+64+64+128+512) + pooled + 2 hb + 2 heads x 512 x (256+128), per instance
    per_inst = 2 * 512 * 832 + 1024 + 2 * 256 + 2 * 512 * 384
    # + one image of the packed kernels (used when the caller supplies none): Cin rounded up to 8, conv[1] rows of head 0
    packed = hip_lib.lrg_packed_weights_bytes(ctypes.byref(w))
    assert packed == 4 * (2 * (16 * 64 + 64 * 64 + 64 * 64 + 64 * 128 + 128 * 512) + 2 * (64 * 256 + 256 * 128))
    assert nbytes >= 4 * per_inst * 4 + packed and nbytes < 4 * per_inst * 4 + packed + 64 * 1024
    off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip_lib.lrg_forward_workspace_view(ctypes.byref(w), 4, 512, 512, 2, 0, ctypes.byref(off), ctypes.byref(cnt)) == 0
    assert cnt.value == 4 * 1024
    w.n_head = 7
    assert hip_lib.lrg_forward_workspace_bytes(ctypes.byref(w), 4, 512, 512) == 0      # rejected, not crashed


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from learn_region_grow_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    import pytest
    with pytest.raises(_lib.LrgHipError):
        _lib.load()
