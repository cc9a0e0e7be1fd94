"""File formats either side of the region-grow path, with the reference's function names and behaviour:

``loadFromH5``  learn_region_grow_util.py:11-31   rooms out of an HDF5 file ('points' [N,F+2], 'count_room' [R])
``saveToH5``    tools/generate_synthetic_rooms.py:111-115 (what the reference's staging tools write)
``savePLY``     learn_region_grow_util.py:57-73   ASCII PLY, "%f %f %f %d %d %d" per vertex
``savePCD``     learn_region_grow_util.py:33-55   ASCII PCD v0.7 with packed rgb
``label_colors`` test_region_grow.py:368-371      the colour table of the result clouds

HDF5 goes through ``h5lite`` (h5py is not installed here); weights through ``checkpoint``.
"""
import numpy as np

from . import h5lite


def loadFromH5(filename, load_labels=True):
    """Rooms of an HDF5 room file: 'points' holds every room's rows back to back, 'count_room' how many belong to each room.
    Returns the per-room arrays, or (features, object ids, class ids) per room with the last two columns split off."""
    f = h5lite.File(filename)
    rows, counts = f['points'].read(), f['count_room'].read()
    rooms = np.split(rows, np.cumsum(counts)[:-1]) if len(counts) else []
    if not load_labels:
        return rooms
    return ([r[:, :-2] for r in rooms], [r[:, -2].astype(int) for r in rooms], [r[:, -1].astype(int) for r in rooms])


def saveToH5(filename, rooms):
    """rooms: list of [n_i, F+2] arrays (features, object id, class id) -> 'points' float32 + 'count_room' int32."""
    count = np.array([len(r) for r in rooms], dtype=np.int32)
    pts = np.vstack(rooms).astype(np.float32) if len(rooms) else np.zeros((0, 8), np.float32)
    h5lite.write_file(filename, {'points': pts, 'count_room': count})


def savePLY(filename, points):
    with open(filename, 'w') as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % len(points))
        for p in points:
            f.write("%f %f %f %d %d %d\n" % (p[0], p[1], p[2], p[3], p[4], p[5]))
    print('Saved to %s: (%d points)' % (filename, len(points)))


def savePCD(filename, points):
    if len(points) == 0:
        return
    n = len(points)
    with open(filename, 'w') as f:
        f.write("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F I\n"
                "COUNT 1 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA ascii\n" % (n, n))
        for p in points:
            rgb = (int(p[3]) << 16) | (int(p[4]) << 8) | int(p[5])
            f.write("%f %f %f %d\n" % (p[0], p[1], p[2], rgb))
    print('Saved %d points to %s' % (n, filename))


def label_colors(n_labels):
    """Colour per cluster id as the reference draws them (test_region_grow.py:368-370): RandomState(0), id 0 grey."""
    obj_color = np.random.RandomState(0).randint(0, 255, (n_labels, 3))
    obj_color[0] = [100, 100, 100]
    return obj_color
