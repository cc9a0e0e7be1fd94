"""A small HDF5 reader / writer for the room files of the reference (no h5py in this interpreter).

The reference stores rooms as two datasets in the root group -- ``points`` float32 ``[N,8]`` (xyz, rgb, object id, class
id) and ``count_room`` int32 ``[R]`` -- written by h5py with ``compression='gzip'`` (tools/generate_synthetic_rooms.py:
111-115, tools/stage_data.py) and read back whole (learn_region_grow_util.py:11-31).  That needs only a corner of the
format, implemented here from the HDF5 file-format specification:

  read : superblock v0/v1 (and v2/v3 with v1-style groups), version-1 and version-2 object headers, old-style groups
         (symbol table + B-tree v1 + local heap), link messages of compact new-style groups, dataspace v1/v2,
         fixed-point / floating-point datatypes, compact / contiguous / chunked (B-tree v1) layouts, the deflate, shuffle and
         fletcher32 filters.
  write: superblock v0, one root group, contiguous little-endian datasets (what h5py reads back without options).
"""
import struct
import zlib

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xffffffffffffffff


class H5Error(ValueError):
    pass


class _File:
    def __init__(self, data):
        self.d = data
        self.O = 8
        self.L = 8
        self.base = 0

    def u(self, pos, size):
        if pos + size > len(self.d):
            raise H5Error('read past end of file')
        return int.from_bytes(self.d[pos:pos + size], 'little')

    def off(self, pos):
        return self.u(pos, self.O)

    def length(self, pos):
        return self.u(pos, self.L)


class Dataset:
    def __init__(self, f, name, shape, dtype, layout, filters):
        self._f, self.name, self.shape, self.dtype, self._layout, self._filters = f, name, shape, dtype, layout, filters

    def __repr__(self):
        return '<h5lite.Dataset %s %s %s>' % (self.name, self.shape, self.dtype)

    def read(self):
        f = self._f
        kind = self._layout[0]
        count = int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1
        nbytes = count * self.dtype.itemsize
        if kind == 'compact':
            raw = self._layout[1]
            return np.frombuffer(raw[:nbytes], dtype=self.dtype).reshape(self.shape).copy()
        if kind == 'contiguous':
            addr = self._layout[1]
            if addr == UNDEF or count == 0:
                return np.zeros(self.shape, dtype=self.dtype)
            if self._filters:
                raise H5Error('filters on a contiguous dataset')
            start = f.base + addr
            if start + nbytes > len(f.d):
                raise H5Error('%s: data beyond end of file' % self.name)
            return np.frombuffer(f.d, dtype=self.dtype, count=count, offset=start).reshape(self.shape).copy()
        if kind == 'chunked':
            return self._read_chunked()
        raise H5Error('unsupported layout %r' % (kind,))

    # ---- chunked storage: B-tree v1, node type 1 --------------------------------------------------------------------
    def _read_chunked(self):
        _, btree, cdims = self._layout
        rank = len(self.shape)
        if len(cdims) != rank:
            raise H5Error('chunk rank mismatch')
        out = np.zeros(self.shape, dtype=self.dtype)
        if btree == UNDEF or out.size == 0:
            return out
        for offsets, addr, size, mask in self._chunks(btree, rank):
            raw = bytes(self._f.d[self._f.base + addr:self._f.base + addr + size])
            if len(raw) != size:
                raise H5Error('%s: chunk beyond end of file' % self.name)
            raw = self._defilter(raw, mask)
            want = int(np.prod(cdims)) * self.dtype.itemsize
            if len(raw) < want:
                raise H5Error('%s: short chunk (%d < %d bytes)' % (self.name, len(raw), want))
            chunk = np.frombuffer(raw, dtype=self.dtype, count=int(np.prod(cdims))).reshape(cdims)
            sel_out, sel_in = [], []
            skip = False
            for o, c, n in zip(offsets, cdims, self.shape):
                if o >= n:
                    skip = True
                    break
                m = min(c, n - o)
                sel_out.append(slice(o, o + m))
                sel_in.append(slice(0, m))
            if not skip:
                out[tuple(sel_out)] = chunk[tuple(sel_in)]
        return out

    def _chunks(self, addr, rank):
        f = self._f
        pos = f.base + addr
        if f.d[pos:pos + 4] != b'TREE':
            raise H5Error('bad chunk B-tree node')
        ntype, level, used = f.d[pos + 4], f.d[pos + 5], f.u(pos + 6, 2)
        if ntype != 1:
            raise H5Error('not a chunk B-tree')
        p = pos + 8 + 2 * f.O
        keysz = 8 + 8 * (rank + 1)
        for _ in range(used):
            size, mask = f.u(p, 4), f.u(p + 4, 4)
            offsets = [f.u(p + 8 + 8 * i, 8) for i in range(rank)]
            child = f.off(p + keysz)
            p += keysz + f.O
            if level == 0:
                yield offsets, child, size, mask
            else:
                for c in self._chunks(child, rank):
                    yield c

    def _defilter(self, raw, mask):
        for i in range(len(self._filters) - 1, -1, -1):
            fid, cd = self._filters[i]
            if mask >> i & 1:
                continue
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                es = cd[0] if cd else self.dtype.itemsize
                n = len(raw) // es
                if es > 1 and n > 0:
                    body = np.frombuffer(raw, dtype=np.uint8, count=n * es).reshape(es, n).T.tobytes()
                    raw = body + raw[n * es:]
            elif fid == 3:
                raw = raw[:-4]
            else:
                raise H5Error('unsupported filter id %d' % fid)
        return raw


def _parse_dataspace(f, m):
    ver = m[0]
    rank = m[1]
    flags = m[2]
    if ver == 1:
        p = 8
    elif ver == 2:
        p = 4
        if m[3] == 2:      # null dataspace
            return (0,)
    else:
        raise H5Error('dataspace version %d' % ver)
    dims = tuple(int.from_bytes(m[p + f.L * i:p + f.L * (i + 1)], 'little') for i in range(rank))
    del flags
    return dims


def _parse_datatype(m):
    cls, ver = m[0] & 0x0f, m[0] >> 4
    bits0 = m[1]
    size = struct.unpack_from('<I', m, 4)[0]
    del ver
    order = '>' if bits0 & 1 else '<'
    if cls == 0:
        kind = 'i' if bits0 & 8 else 'u'
    elif cls == 1:
        kind = 'f'
    else:
        raise H5Error('unsupported datatype class %d' % cls)
    if size == 1:
        order = '|'
    return np.dtype('%s%s%d' % (order, kind, size))


def _parse_layout(f, m):
    ver = m[0]
    if ver == 3:
        cls = m[1]
        if cls == 0:
            size = struct.unpack_from('<H', m, 2)[0]
            return ('compact', bytes(m[4:4 + size]))
        if cls == 1:
            return ('contiguous', int.from_bytes(m[2:2 + f.O], 'little'), int.from_bytes(m[2 + f.O:2 + f.O + f.L], 'little'))
        if cls == 2:
            nd = m[2]
            btree = int.from_bytes(m[3:3 + f.O], 'little')
            dims = struct.unpack_from('<%dI' % nd, m, 3 + f.O)
            return ('chunked', btree, tuple(dims[:-1]))
        raise H5Error('layout class %d' % cls)
    if ver in (1, 2):
        nd, cls = m[1], m[2]
        p = 8
        addr = None
        if cls != 0:
            addr = int.from_bytes(m[p:p + f.O], 'little')
            p += f.O
        dims = struct.unpack_from('<%dI' % nd, m, p)
        p += 4 * nd
        if cls == 1:
            return ('contiguous', addr, 0)
        if cls == 2:
            return ('chunked', addr, tuple(dims[:-1]))
        size = struct.unpack_from('<I', m, p)[0]
        return ('compact', bytes(m[p + 4:p + 4 + size]))
    raise H5Error('data layout version %d is not supported (file written with libver="latest"?)' % ver)


def _parse_filters(m):
    ver, n = m[0], m[1]
    out = []
    if ver == 1:
        p = 8
        for _ in range(n):
            fid, nlen, _flags, ncd = struct.unpack_from('<HHHH', m, p)
            p += 8 + ((nlen + 7) & ~7)
            cd = struct.unpack_from('<%dI' % ncd, m, p)
            p += 4 * ncd + (4 if ncd & 1 else 0)
            out.append((fid, cd))
    elif ver == 2:
        p = 2
        for _ in range(n):
            fid = struct.unpack_from('<H', m, p)[0]
            p += 2
            nlen = 0
            if fid >= 256:
                nlen = struct.unpack_from('<H', m, p)[0]
                p += 2
            _flags, ncd = struct.unpack_from('<HH', m, p)
            p += 4 + nlen
            cd = struct.unpack_from('<%dI' % ncd, m, p)
            p += 4 * ncd
            out.append((fid, cd))
    else:
        raise H5Error('filter pipeline version %d' % ver)
    return out


def _messages(f, addr):
    """Yield (type, bytes) of every header message of the object at `addr` (v1 and v2 object headers)."""
    pos = f.base + addr
    if f.d[pos:pos + 4] == b'OHDR':
        ver, flags = f.d[pos + 4], f.d[pos + 5]
        if ver != 2:
            raise H5Error('object header version %d' % ver)
        p = pos + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        csz = 1 << (flags & 3)
        chunk0 = f.u(p, csz)
        p += csz
        blocks = [(p, p + chunk0)]
        track = bool(flags & 0x04)
        while blocks:
            p, end = blocks.pop(0)
            while p + 4 + (2 if track else 0) <= end:
                mtype, msize, _mflags = f.d[p], f.u(p + 1, 2), f.d[p + 3]
                p += 4 + (2 if track else 0)
                body = f.d[p:p + msize]
                p += msize
                if mtype == 0x10:
                    caddr = int.from_bytes(body[:f.O], 'little')
                    clen = int.from_bytes(body[f.O:f.O + f.L], 'little')
                    cp = f.base + caddr
                    if f.d[cp:cp + 4] != b'OCHK':
                        raise H5Error('bad continuation block')
                    blocks.append((cp + 4, cp + clen - 4))
                elif mtype != 0:
                    yield mtype, bytes(body)
        return
    ver = f.d[pos]
    if ver != 1:
        raise H5Error('object header version %d at %d' % (ver, addr))
    nmsg = f.u(pos + 2, 2)
    hsize = f.u(pos + 8, 4)
    blocks = [(pos + 16, pos + 16 + hsize)]
    seen = 0
    while blocks and seen < nmsg:
        p, end = blocks.pop(0)
        while p + 8 <= end and seen < nmsg:
            mtype, msize = f.u(p, 2), f.u(p + 2, 2)
            body = f.d[p + 8:p + 8 + msize]
            p += 8 + msize
            seen += 1
            if mtype == 0x10:
                caddr = int.from_bytes(body[:f.O], 'little')
                clen = int.from_bytes(body[f.O:f.O + f.L], 'little')
                blocks.append((f.base + caddr, f.base + caddr + clen))
            elif mtype != 0:
                yield mtype, bytes(body)


def _group_links(f, addr):
    """name -> object header address of the members of the group whose header is at `addr`."""
    links = {}
    for mtype, m in _messages(f, addr):
        if mtype == 0x11:                                    # symbol table: B-tree + local heap
            btree = int.from_bytes(m[:f.O], 'little')
            heap = int.from_bytes(m[f.O:2 * f.O], 'little')
            hp = f.base + heap
            if f.d[hp:hp + 4] != b'HEAP':
                raise H5Error('bad local heap')
            seg = f.base + f.off(hp + 8 + 2 * f.L)
            _walk_group_btree(f, btree, seg, links)
        elif mtype == 0x06:                                  # link message (compact new-style group)
            ver, flags = m[0], m[1]
            if ver != 1:
                raise H5Error('link message version %d' % ver)
            p = 2
            ltype = 0
            if flags & 0x08:
                ltype = m[p]
                p += 1
            if flags & 0x04:
                p += 8
            if flags & 0x10:
                p += 1
            lsz = 1 << (flags & 3)
            nlen = int.from_bytes(m[p:p + lsz], 'little')
            p += lsz
            name = m[p:p + nlen].decode('utf-8')
            p += nlen
            if ltype == 0:
                links[name] = int.from_bytes(m[p:p + f.O], 'little')
        elif mtype == 0x02:
            raise H5Error('dense new-style groups (fractal heap) are not supported')
    return links


def _walk_group_btree(f, addr, seg, links):
    if addr == UNDEF:
        return
    pos = f.base + addr
    if f.d[pos:pos + 4] != b'TREE':
        raise H5Error('bad group B-tree node')
    ntype, level, used = f.d[pos + 4], f.d[pos + 5], f.u(pos + 6, 2)
    if ntype != 0:
        raise H5Error('not a group B-tree')
    p = pos + 8 + 2 * f.O + f.L                              # skip key 0
    for _ in range(used):
        child = f.off(p)
        p += f.O + f.L
        if level > 0:
            _walk_group_btree(f, child, seg, links)
            continue
        sp = f.base + child
        if f.d[sp:sp + 4] != b'SNOD':
            raise H5Error('bad symbol table node')
        nsym = f.u(sp + 6, 2)
        e = sp + 8
        for _s in range(nsym):
            noff = f.off(e)
            oaddr = f.off(e + f.O)
            end = f.d.index(b'\x00', seg + noff)
            links[f.d[seg + noff:end].decode('utf-8')] = oaddr
            e += 2 * f.O + 24


class File:
    """Read-only view of an HDF5 file: ``File(path)['points'].read()``; ``keys()`` lists the root group."""

    def __init__(self, path):
        with open(path, 'rb') as fh:
            data = fh.read()
        pos = 0
        while True:
            if data[pos:pos + 8] == SIGNATURE:
                break
            pos = 512 if pos == 0 else pos * 2
            if pos + 8 > len(data):
                raise H5Error('%s: not an HDF5 file' % path)
        f = _File(data)
        ver = data[pos + 8]
        if ver in (0, 1):
            f.O, f.L = data[pos + 13], data[pos + 14]
            p = pos + 24 + (4 if ver == 1 else 0)
            f.base = f.off(p)
            root_entry = p + 4 * f.O
            root = f.off(root_entry + f.O)
        elif ver in (2, 3):
            f.O, f.L = data[pos + 9], data[pos + 10]
            f.base = f.off(pos + 12)
            root = f.off(pos + 12 + 3 * f.O)
        else:
            raise H5Error('superblock version %d' % ver)
        if f.O not in (4, 8) or f.L not in (4, 8):
            raise H5Error('unsupported offset/length sizes')
        f.base += 0
        self._f = f
        self._links = _group_links(f, root)

    def keys(self):
        return sorted(self._links)

    def __contains__(self, name):
        return name in self._links

    def __getitem__(self, name):
        if name not in self._links:
            raise KeyError(name)
        f = self._f
        shape = dtype = layout = None
        filters = []
        for mtype, m in _messages(f, self._links[name]):
            if mtype == 0x01:
                shape = _parse_dataspace(f, m)
            elif mtype == 0x03:
                dtype = _parse_datatype(m)
            elif mtype == 0x08:
                layout = _parse_layout(f, m)
            elif mtype == 0x0b:
                filters = _parse_filters(m)
        if shape is None or dtype is None or layout is None:
            raise H5Error('%s is not a dataset this reader understands' % name)
        return Dataset(f, name, shape, dtype, layout, filters)

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


# ---- writer ----------------------------------------------------------------------------------------------------------
def _pad8(b):
    return b + b'\x00' * (-len(b) % 8)


def _msg(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack('<HHB3x', mtype, len(body), flags) + body


def _dtype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind == 'f':
        if dt.itemsize == 4:
            props = struct.pack('<HHBBBBI', 0, 32, 23, 8, 0, 23, 127)
            bits = (0x20, 31, 0)
        elif dt.itemsize == 8:
            props = struct.pack('<HHBBBBI', 0, 64, 52, 11, 0, 52, 1023)
            bits = (0x20, 63, 0)
        else:
            raise H5Error('float%d' % (8 * dt.itemsize))
        return struct.pack('<BBBBI', 0x11, bits[0], bits[1], bits[2], dt.itemsize) + props
    if dt.kind in 'iu':
        return struct.pack('<BBBBI', 0x10, 0x08 if dt.kind == 'i' else 0, 0, 0, dt.itemsize) + struct.pack('<HH', 0, 8 * dt.itemsize)
    raise H5Error('dtype %s cannot be written' % dt)


def write_file(path, datasets):
    """Write ``{name: array}`` as contiguous little-endian datasets in the root group of a new HDF5 file."""
    names = sorted(datasets)
    arrays = {}
    for n in names:
        a = np.asarray(datasets[n])
        a = np.ascontiguousarray(a).reshape(a.shape)
        if a.dtype.byteorder == '>':
            a = a.astype(a.dtype.newbyteorder('<'))
        arrays[n] = a
    O = 8
    out = bytearray(b'\x00' * 96)                            # superblock (56) + root symbol table entry (40)

    def alloc(b):
        while len(out) % 8:
            out.append(0)
        addr = len(out)
        out.extend(b)
        return addr

    # local heap data segment: names
    seg = bytearray(b'\x00' * 8)                             # offset 0 = the empty string
    name_off = {}
    for n in names:
        name_off[n] = len(seg)
        seg += _pad8(n.encode('utf-8') + b'\x00')
    free_off = len(seg)
    seg += struct.pack('<QQ', 1, 16)                         # one free block: next = 1 (none), size 16
    seg_size = len(seg)
    # datasets: raw data then object headers
    obj_addr = {}
    for n in names:
        a = arrays[n]
        daddr = alloc(a.tobytes()) if a.size else UNDEF
        space = struct.pack('<BBB5x', 1, a.ndim, 0) + b''.join(struct.pack('<Q', s) for s in a.shape)
        layout = struct.pack('<BB', 3, 1) + struct.pack('<QQ', daddr, a.nbytes)
        fill = struct.pack('<BBBB', 2, 2, 2, 0)              # v2: allocate late, write at alloc time, undefined value
        msgs = _msg(0x01, space) + _msg(0x03, _dtype_msg(a.dtype), flags=1) + _msg(0x05, fill) + _msg(0x08, layout)
        hdr = struct.pack('<BBHII4x', 1, 0, 4, 1, len(msgs)) + msgs
        obj_addr[n] = alloc(hdr)
    heap_data = alloc(bytes(seg))
    heap = alloc(b'HEAP' + struct.pack('<B3xQQQ', 0, seg_size, free_off, heap_data))
    # one symbol table node with every entry (2K = 8 entries fit with the default K = 4 ... use K large enough)
    K = max(4, (len(names) + 1) // 2)
    snod = bytearray(b'SNOD' + struct.pack('<BBH', 1, 0, len(names)))
    for n in names:
        snod += struct.pack('<QQII16x', name_off[n], obj_addr[n], 0, 0)
    snod += b'\x00' * (40 * (2 * K - len(names)))
    snod_addr = alloc(bytes(snod))
    btree = bytearray(b'TREE' + struct.pack('<BBHQQ', 0, 0, 1 if names else 0, UNDEF, UNDEF))
    btree += struct.pack('<Q', 0)
    if names:
        btree += struct.pack('<QQ', snod_addr, name_off[names[-1]])
    btree += b'\x00' * ((2 * 16 + 1) * 8 + 2 * 16 * 8 - (len(btree) - 24))   # room for 2K keys/children at the internal K = 16
    btree_addr = alloc(bytes(btree))
    root_msgs = _msg(0x11, struct.pack('<QQ', btree_addr, heap))
    root_hdr = struct.pack('<BBHII4x', 1, 0, 1, 1, len(root_msgs)) + root_msgs
    root_addr = alloc(root_hdr)
    eof = len(out)
    sb = SIGNATURE + struct.pack('<BBBBBBBBHHI', 0, 0, 0, 0, 0, O, 8, 0, K, 16, 0)
    sb += struct.pack('<QQQQ', 0, UNDEF, eof, UNDEF)
    sb += struct.pack('<QQII', 0, root_addr, 1, 0) + struct.pack('<QQ', btree_addr, heap)
    out[:len(sb)] = sb
    with open(path, 'wb') as fh:
        fh.write(bytes(out))
