"""TensorFlow "bundle" (checkpoint V2) reader / writer -- the ``saver.restore(sess, MODEL_PATH)`` step of the reference
(test_region_grow.py:92-93, test_random_restart.py:106-107) without TensorFlow.

A checkpoint ``<prefix>`` is two files: ``<prefix>.index`` -- a LevelDB-format sorted string table whose keys are variable
names and whose values are serialized ``BundleEntryProto`` messages (dtype, shape, shard, offset, size, masked CRC-32C) --
and ``<prefix>.data-00000-of-00001`` -- the raw little-endian tensor bytes.  The reference's trained model is
``models/lrgnet_model5.ckpt`` (99 entries: the 32 ``lrg_*`` trainables, their Adam slots and three scalars; the data blob
is not distributed with the repository).  ``load_lrgnet_weights`` returns the ``name -> [1,Cin,Cout] / [Cout]`` dict that
``LrgNetHIP.load_weights`` takes; ``write_bundle`` writes a checkpoint the reference's ``tf.train.Saver`` can restore
(e.g. the synthetic weights used here, for cross-checking against the TensorFlow graph on a machine that has it).

Only what those files use is implemented: uncompressed table blocks, single-shard bundles, dense tensors.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8
# tensorflow/core/framework/types.proto
DTYPES = {1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 4: np.dtype('u1'), 5: np.dtype('<i2'),
          6: np.dtype('i1'), 9: np.dtype('<i8'), 10: np.dtype('?'), 17: np.dtype('<u2'), 22: np.dtype('<u4'),
          23: np.dtype('<u8')}
_DTYPE_ENUM = {v: k for k, v in DTYPES.items()}


class BundleError(ValueError):
    pass


# ---- CRC-32C (Castagnoli), slicing-by-8, and LevelDB's masking ----------------------------------------------------
def _make_tables():
    poly = 0x82f63b78
    t0 = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        t0[i] = c
    tabs = [t0]
    for _ in range(7):
        prev = tabs[-1]
        tabs.append((prev >> np.uint32(8)) ^ t0[prev & np.uint32(0xff)])
    return [t.tolist() for t in tabs]


_T = _make_tables()


def crc32c(data, crc=0):
    b = memoryview(bytes(data) if not isinstance(data, (bytes, bytearray, memoryview)) else data).cast('B')
    c = crc ^ 0xffffffff
    n = len(b)
    i = 0
    t0, t1, t2, t3, t4, t5, t6, t7 = _T
    n8 = n - (n % 8)
    if n8:
        words = np.frombuffer(b[:n8], dtype='<u4').tolist()
        for j in range(0, len(words), 2):
            lo = words[j] ^ c
            hi = words[j + 1]
            c = (t7[lo & 0xff] ^ t6[(lo >> 8) & 0xff] ^ t5[(lo >> 16) & 0xff] ^ t4[lo >> 24] ^
                 t3[hi & 0xff] ^ t2[(hi >> 8) & 0xff] ^ t1[(hi >> 16) & 0xff] ^ t0[hi >> 24])
        i = n8
    while i < n:
        c = t0[(c ^ b[i]) & 0xff] ^ (c >> 8)
        i += 1
    return c ^ 0xffffffff


def mask_crc(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xffffffff) + _MASK_DELTA) & 0xffffffff


def unmask_crc(masked):
    rot = (masked - _MASK_DELTA) & 0xffffffff
    return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ---- varints / the two protobuf messages ---------------------------------------------------------------------------
def _get_varint(buf, pos):
    out = shift = 0
    while True:
        if pos >= len(buf):
            raise BundleError('truncated varint')
        b = buf[pos]
        pos += 1
        out |= (b & 0x7f) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 70:
            raise BundleError('varint too long')


def _put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        if v < 0x80:
            out.append(v)
            return bytes(out)
        out.append((v & 0x7f) | 0x80)
        v >>= 7


def _fields(buf):
    """Yield (field number, wire type, value) of one protobuf message (values: int or bytes)."""
    pos = 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        num, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _get_varint(buf, pos)
        elif wt == 1:
            val = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            val = bytes(buf[pos:pos + ln])
            if len(val) != ln:
                raise BundleError('truncated field')
            pos += ln
        elif wt == 5:
            val = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise BundleError('unsupported wire type %d' % wt)
        yield num, wt, val


def _parse_shape(buf):
    dims = []
    for num, _, val in _fields(buf):
        if num == 2:                                  # repeated Dim dim = 2 { int64 size = 1; string name = 2 }
            size = 0
            for n2, _, v2 in _fields(val):
                if n2 == 1:
                    size = v2 - (1 << 64) if v2 >> 63 else v2
            dims.append(size)
        elif num == 3 and val:
            raise BundleError('unknown-rank shape')
    return tuple(dims)


class BundleEntry:
    __slots__ = ('dtype', 'shape', 'shard_id', 'offset', 'size', 'crc32c', 'sliced')

    def __init__(self):
        self.dtype, self.shape, self.shard_id, self.offset, self.size, self.crc32c, self.sliced = 0, (), 0, 0, 0, 0, False

    def __repr__(self):
        return 'BundleEntry(dtype=%d, shape=%s, shard=%d, offset=%d, size=%d, crc=0x%08x)' % (
            self.dtype, self.shape, self.shard_id, self.offset, self.size, self.crc32c)


def _parse_entry(buf):
    e = BundleEntry()
    for num, _, val in _fields(buf):                  # tensorflow/core/protobuf/tensor_bundle.proto
        if num == 1:
            e.dtype = val
        elif num == 2:
            e.shape = _parse_shape(val)
        elif num == 3:
            e.shard_id = val
        elif num == 4:
            e.offset = val
        elif num == 5:
            e.size = val
        elif num == 6:
            e.crc32c = val
        elif num == 7:
            e.sliced = True
    return e


def _encode_entry(dtype_enum, shape, offset, size, crc_masked):
    dims = b''.join(b'\x12' + _put_varint(len(d)) + d for d in (b'\x08' + _put_varint(int(s)) for s in shape))
    out = b'\x08' + _put_varint(dtype_enum) + b'\x12' + _put_varint(len(dims)) + dims
    if offset:
        out += b'\x20' + _put_varint(offset)
    out += b'\x28' + _put_varint(size) + b'\x35' + struct.pack('<I', crc_masked)
    return out


# ---- the sorted string table ---------------------------------------------------------------------------------------
def _read_block(data, offset, size, verify):
    raw = data[offset:offset + size]
    trailer = data[offset + size:offset + size + 5]
    if len(raw) != size or len(trailer) != 5:
        raise BundleError('table block out of range')
    if trailer[0] != 0:
        raise BundleError('compressed table blocks (type %d) are not supported' % trailer[0])
    if verify:
        want = struct.unpack('<I', trailer[1:5])[0]
        if mask_crc(crc32c(trailer[:1], crc32c(raw))) != want:
            raise BundleError('table block checksum mismatch at offset %d' % offset)
    return raw


def _block_entries(block):
    if len(block) < 4:
        raise BundleError('table block too small')
    nrestarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestarts
    if end < 0:
        raise BundleError('bad restart array')
    pos, key = 0, b''
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > end:
            raise BundleError('corrupt table entry')
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_table(path, verify=True):
    """key (bytes) -> value (bytes) of a LevelDB-format table file, in key order."""
    data = open(path, 'rb').read()
    if len(data) < 48:
        raise BundleError('%s: too small for a table' % path)
    footer = data[-48:]
    if struct.unpack('<Q', footer[40:])[0] != TABLE_MAGIC:
        raise BundleError('%s: bad table magic' % path)
    pos = 0
    _, pos = _get_varint(footer, pos)                 # metaindex handle (unused)
    _, pos = _get_varint(footer, pos)
    ioff, pos = _get_varint(footer, pos)
    isize, pos = _get_varint(footer, pos)
    out = {}
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, p = _get_varint(handle, 0)
        bsize, p = _get_varint(handle, p)
        for k, v in _block_entries(_read_block(data, boff, bsize, verify)):
            out[k] = v
    return out


def _build_block(items, restart_interval=16):
    out = bytearray()
    restarts = []
    prev = b''
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts))
    return bytes(out)


def write_table(path, items):
    """items: iterable of (key bytes, value bytes) in strictly increasing key order; one data block."""
    items = list(items)
    for a, b in zip(items, items[1:]):
        if not a[0] < b[0]:
            raise BundleError('table keys must be strictly increasing')
    out = bytearray()

    def emit(block):
        off = len(out)
        out.extend(block)
        out.extend(b'\x00' + struct.pack('<I', mask_crc(crc32c(b'\x00', crc32c(block)))))
        return _put_varint(off) + _put_varint(len(block))

    data_handle = emit(_build_block(items))
    meta_handle = emit(_build_block([]))
    last_key = items[-1][0] if items else b''
    index_handle = emit(_build_block([(last_key, data_handle)], restart_interval=1))
    footer = meta_handle + index_handle
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
    out.extend(footer)
    with open(path, 'wb') as f:
        f.write(bytes(out))


# ---- bundles -------------------------------------------------------------------------------------------------------
def _data_path(prefix, shard, num_shards):
    return '%s.data-%05d-of-%05d' % (prefix, shard, num_shards)


def read_bundle_index(prefix, verify=True):
    """-> (header dict, {variable name: BundleEntry}) of ``<prefix>.index``."""
    table = read_table(prefix + '.index', verify=verify)
    if b'' not in table:
        raise BundleError('%s.index: no bundle header' % prefix)
    header = dict(num_shards=1, endianness=0)
    for num, _, val in _fields(table[b'']):
        if num == 1:
            header['num_shards'] = val
        elif num == 2:
            header['endianness'] = val
    if header['endianness'] != 0:
        raise BundleError('big-endian bundles are not supported')
    entries = {k.decode('utf-8'): _parse_entry(v) for k, v in table.items() if k != b''}
    return header, entries


def load_bundle(prefix, names=None, verify=True):
    """name -> ndarray for the named variables (default: all dense ones) of checkpoint ``prefix``."""
    header, entries = read_bundle_index(prefix, verify=verify)
    if names is None:
        names = [n for n, e in entries.items() if not e.sliced and e.dtype in DTYPES]
    out = {}
    files = {}
    try:
        for name in names:
            if name not in entries:
                raise KeyError('%s: no variable named %r' % (prefix, name))
            e = entries[name]
            if e.sliced:
                raise BundleError('%s: partitioned variables are not supported' % name)
            if e.dtype not in DTYPES:
                raise BundleError('%s: unsupported dtype enum %d' % (name, e.dtype))
            dt = DTYPES[e.dtype]
            count = int(np.prod(e.shape, dtype=np.int64)) if e.shape else 1
            if count * dt.itemsize != e.size:
                raise BundleError('%s: %d bytes on disk for shape %s of %s' % (name, e.size, e.shape, dt))
            if e.shard_id not in files:
                files[e.shard_id] = open(_data_path(prefix, e.shard_id, header['num_shards']), 'rb')
            f = files[e.shard_id]
            f.seek(e.offset)
            raw = f.read(e.size)
            if len(raw) != e.size:
                raise BundleError('%s: data file truncated' % name)
            if verify and mask_crc(crc32c(raw)) != e.crc32c:
                raise BundleError('%s: CRC-32C mismatch (corrupt checkpoint data)' % name)
            out[name] = np.frombuffer(raw, dtype=dt).reshape(e.shape).copy()
    finally:
        for f in files.values():
            f.close()
    return out


def write_bundle(prefix, tensors):
    """Write ``{name: array}`` as a single-shard checkpoint ``<prefix>.index`` + ``<prefix>.data-00000-of-00001``."""
    names = sorted(tensors, key=lambda n: n.encode('utf-8'))
    header = b'\x08\x01' + b'\x1a\x02\x08\x01'        # num_shards = 1, (little endian,) version { producer: 1 }
    items = [(b'', header)]
    offset = 0
    d = os.path.dirname(prefix)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(_data_path(prefix, 0, 1), 'wb') as f:
        for name in names:
            a = np.asarray(tensors[name])
            a = np.ascontiguousarray(a).reshape(a.shape)          # (ascontiguousarray alone promotes scalars to [1])
            dt = a.dtype.newbyteorder('<') if a.dtype.itemsize > 1 else a.dtype
            if dt not in _DTYPE_ENUM:
                raise BundleError('%s: dtype %s cannot be stored' % (name, a.dtype))
            a = a.astype(dt, copy=False)
            raw = a.tobytes()
            enum = _DTYPE_ENUM[dt]
            items.append((name.encode('utf-8'), _encode_entry(enum, a.shape, offset, len(raw), mask_crc(crc32c(raw)))))
            f.write(raw)
            offset += len(raw)
    write_table(prefix + '.index', items)


def lrgnet_variable_shapes(feature_size=13, lite=0):
    """name -> TF shape of the 32 (lite: fewer) trainable variables (learn_region_grow_util.py:77-85,:107-159)."""
    from .lrgnet import CONV_CHANNELS, CONV2_CHANNELS
    lite = 0 if lite is None else int(lite)
    cc, c2 = CONV_CHANNELS[lite], CONV2_CHANNELS[lite]
    shapes = {}
    for pre in ('lrg_', 'lrg_neighbor_'):
        for i, c in enumerate(cc):
            shapes['%skernel%d' % (pre, i)] = (1, feature_size if i == 0 else cc[i - 1], c)
            shapes['%sbias%d' % (pre, i)] = (c,)
    for pre in ('lrg_add_', 'lrg_remove_'):
        for i, c in enumerate(c2):
            shapes['%skernel%d' % (pre, i)] = (1, cc[-1] * 2 + cc[1] if i == 0 else c2[i - 1], c)
            shapes['%sbias%d' % (pre, i)] = (c,)
        shapes['%skernel%d' % (pre, len(c2))] = (1, c2[-1], 2)
        shapes['%sbias%d' % (pre, len(c2))] = (2,)
    return shapes


def lrgnet_checkpoint_tensors(weights, adam_m=None, adam_v=None, step=0, beta1=0.9, beta2=0.999):
    """Everything ``tf.compat.v1.train.Saver().save`` writes for an LrgNet graph -- which is what ``Saver().restore``
    (test_region_grow.py:92-93) then asks for: the trainable variables, their Adam slots ``<name>/Adam`` (first moment) and
    ``<name>/Adam_1`` (second moment), ``beta1_power`` / ``beta2_power`` (learn_region_grow_util.py:188: AdamOptimizer) and
    ``Variable`` (the global step ``batch``, :187, int32) -- 99 entries for lite 0, the key set of models/lrgnet_model5.ckpt.index.
    weights / adam_m / adam_v: name -> array in the TF shapes (missing slots: zeros, a freshly built optimizer)."""
    out = {}
    for k, w in weights.items():
        w = np.asarray(w, dtype=np.float32)
        out[k] = w
        out[k + '/Adam'] = np.asarray(adam_m[k], dtype=np.float32).reshape(w.shape) if adam_m is not None else np.zeros_like(w)
        out[k + '/Adam_1'] = np.asarray(adam_v[k], dtype=np.float32).reshape(w.shape) if adam_v is not None else np.zeros_like(w)
    # TF1's AdamOptimizer creates the two accumulators AT beta (not 1) and multiplies them once per applied step, so a
    # checkpoint at global step t holds beta ** (t + 1); a freshly built optimizer (t = 0) holds beta itself -- with 1.0 there a
    # resumed reference run would compute lr * sqrt(1 - 1) / (1 - 1).
    out['beta1_power'] = np.float32(float(beta1) ** (int(step) + 1))
    out['beta2_power'] = np.float32(float(beta2) ** (int(step) + 1))
    out['Variable'] = np.int32(step)
    return out


def load_lrgnet_weights(prefix, feature_size=13, lite=0, verify=True):
    """The trainable variables of an LrgNet checkpoint (optimizer slots ignored), shapes checked against the
    architecture ``LrgNet(..., feature_size, lite)`` builds -- what ``saver.restore`` would have assigned."""
    shapes = lrgnet_variable_shapes(feature_size, lite)
    _, entries = read_bundle_index(prefix, verify=verify)
    for name, shp in shapes.items():
        if name not in entries:
            raise KeyError('%s: variable %s missing (wrong --lite / architecture?)' % (prefix, name))
        if tuple(entries[name].shape) != tuple(shp):
            raise BundleError('%s: checkpoint has shape %s, the network needs %s' % (name, entries[name].shape, shp))
        if entries[name].dtype != 1:
            raise BundleError('%s: not float32' % name)
    return load_bundle(prefix, names=list(shapes), verify=verify)
