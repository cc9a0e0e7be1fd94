"""CPU oracle for the LRGNet region-grow hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it, and only as the checker / CPU baseline.
``learn_region_grow_amd`` never imports this package.

Each module restates one piece of the reference (jingdao/learn_region_grow)
in NumPy / plain C and cites the reference ``file:line`` it follows:

  lrgnet_ref.py     LrgNet forward + logged scalars   learn_region_grow_util.py:75-186
  grow_ref.py       greedy + random-restart grow loop test_region_grow.py:175-316,
                                                     test_random_restart.py:141-303
  preprocess_ref.py voxel equalisation, normals       test_region_grow.py:119-173
  metrics_ref.py    per-room metric block             test_region_grow.py:319-355
  rng_ref.py        legacy (reference-order) RNG and the counter-based RNG spec
  grouping_ref.c    ball query / top-k / gather       tf_ops/grouping/tf_grouping_g.cu:3-123

Pinning (see DESIGN.md "Oracle pinning"):
  * grouping_ref.c is checked against the reference's own CPU functions compiled
    from /root/reference/tf_ops/grouping/test/*.cpp (oracle/_ref, built by
    oracle/Makefile) and against the known-answer vector of selection_sort.cpp.
  * lrgnet_ref.py / grow_ref.py are checked against golden vectors produced by
    executing the reference's own, unmodified Python (LrgNet.__init__ and the
    whole test_region_grow.py / test_random_restart.py scripts) in the build
    container under a NumPy stand-in for the absent TensorFlow
    (tests/golden/make_golden.py).  TensorFlow's kernels themselves are not
    available anywhere in this environment, so fp32 kernel-level numerics of
    TF are not pinned; wiring, control flow, RNG order and labels are.
"""
