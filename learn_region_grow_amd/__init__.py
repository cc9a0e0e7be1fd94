"""learn_region_grow_amd -- MI355X-native LRGNet region-grow inference path.

Host side (Python) mirrors the reference's operator interface for this path
(``LrgNet`` -> ``LrgNetHIP``, the ``test_region_grow.py`` loop -> ``RegionGrower``,
``tf_ops/grouping`` -> ``grouping``) and drives hand-written HIP kernels for gfx950
through the C-ABI library ``liblrg_hip.so`` (include/lrg_hip.h).
"""
__version__ = "0.1.0"
