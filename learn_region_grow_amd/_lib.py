"""ctypes binding of liblrg_hip.so (include/lrg_hip.h).

There is no CPU fallback: if the library is missing or a call fails, an exception is raised.
``build()`` compiles the HIP sources in-tree with hipcc for gfx950.
"""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_PATH = os.path.join(HERE, 'liblrg_hip.so')
SOURCES = ['lrg_net.hip', 'lrg_fused.hip', 'lrg_grow.hip', 'lrg_grouping.hip', 'lrg_preprocess.hip', 'lrg_train.hip']

LRG_ABI_VERSION = 10      # what this binding was written against (include/lrg_hip.h: LRG_ABI_VERSION; tests/test_capi.py compares them and INTEGRATION.md)
LRG_EINVAL = -1000
LRG_ERESIDENCY = -1100     # lrg_grow_async: its workgroups cannot all be resident at once on this stream / device
LRG_MAX_CONV = 5
LRG_MAX_HEAD = 3
LRG_FWD_FUSE_POOL = 1
LRG_FWD_FUSED = 2
LRG_FWD_KEEP_ACTS = 4
LRG_FWD_POOL_ZEROED = 8
LRG_FWD_STREAM_TILES = 32

(LRG_IDLE, LRG_ACTIVE, LRG_STOP_NONEIGHBOR, LRG_STOP_NOEXPAND, LRG_STOP_STUCK, LRG_STOP_EMPTY, LRG_STOP_MAXSTEPS,
 LRG_DONE, LRG_WAIT, LRG_PENDING) = range(10)
REASON_NAMES = {LRG_STOP_NONEIGHBOR: 'noneighbor', LRG_STOP_NOEXPAND: 'noexpand', LRG_STOP_STUCK: 'stuck',
                LRG_STOP_EMPTY: 'empty', LRG_STOP_MAXSTEPS: 'maxsteps'}

_fp = ctypes.c_void_p


class LrgWeights(ctypes.Structure):
    _fields_ = [('feature_size', ctypes.c_int32), ('n_conv', ctypes.c_int32), ('n_head', ctypes.c_int32),
                ('reserved', ctypes.c_int32),
                ('conv_ch', ctypes.c_int32 * LRG_MAX_CONV), ('head_ch', ctypes.c_int32 * LRG_MAX_HEAD),
                ('inlier_w', _fp * LRG_MAX_CONV), ('inlier_b', _fp * LRG_MAX_CONV),
                ('neighbor_w', _fp * LRG_MAX_CONV), ('neighbor_b', _fp * LRG_MAX_CONV),
                ('add_w', _fp * LRG_MAX_HEAD), ('add_b', _fp * LRG_MAX_HEAD),
                ('rmv_w', _fp * LRG_MAX_HEAD), ('rmv_b', _fp * LRG_MAX_HEAD), ('packed', _fp)]


class LrgRoom(ctypes.Structure):
    _fields_ = [('points', _fp), ('voxels', _fp), ('obj_id', _fp), ('order', _fp), ('visited', _fp), ('label', _fp),
                ('hash_keys', _fp), ('hash_vals', _fp), ('region_log', _fp),
                ('n', ctypes.c_int32), ('hash_mask', ctypes.c_int32), ('next_cluster_id', ctypes.c_int32),
                ('seed_cursor', ctypes.c_int32), ('n_regions', ctypes.c_int32), ('done', ctypes.c_int32),
                ('room_id', ctypes.c_int32), ('pad', ctypes.c_int32), ('pvox', _fp), ('vox_origin', ctypes.c_int32 * 3),
                ('chan_stride', ctypes.c_int32), ('chan_major', _fp), ('vgrid', _fp), ('vgrid_dim', ctypes.c_int32 * 3),
                ('pad2', ctypes.c_int32)]


class LrgSlot(ctypes.Structure):
    _fields_ = [('cur', _fp), ('best', _fp), ('cur_idx', _fp), ('cand_idx', _fp),
                ('room', ctypes.c_int32), ('status', ctypes.c_int32), ('seed', ctypes.c_int32),
                ('restart', ctypes.c_int32), ('step', ctypes.c_int32), ('steps_total', ctypes.c_int32),
                ('stuck', ctypes.c_int32), ('nc', ctypes.c_int32), ('ne', ctypes.c_int32),
                ('updated', ctypes.c_int32), ('count', ctypes.c_int32), ('best_count', ctypes.c_int32),
                ('best_restart', ctypes.c_int32), ('last_reason', ctypes.c_int32),
                ('mn', ctypes.c_int32 * 3), ('mx', ctypes.c_int32 * 3),
                ('seq_mn', ctypes.c_int32 * 3), ('seq_mx', ctypes.c_int32 * 3),
                ('target', ctypes.c_int32), ('pad', ctypes.c_int32), ('chunk_cnt', _fp), ('scan_cnt', ctypes.c_int32),
                ('scan_mn', ctypes.c_int32 * 3), ('scan_mx', ctypes.c_int32 * 3), ('query', ctypes.c_int32),
                ('acc_add', ctypes.c_int32), ('acc_rmv', ctypes.c_int32), ('ml_score', ctypes.c_double), ('ml_best', ctypes.c_double),
                ('spec_pos', ctypes.c_int32), ('spec_flags', ctypes.c_int32)]


LRG_SCAN_CHUNK = 4096


class LrgGrowParams(ctypes.Structure):
    _fields_ = [('resolution', ctypes.c_float), ('feature_size', ctypes.c_int32), ('n_inlier', ctypes.c_int32),
                ('n_neighbor', ctypes.c_int32), ('cluster_threshold', ctypes.c_int32), ('restarts', ctypes.c_int32),
                ('group_size', ctypes.c_int32), ('max_region_steps', ctypes.c_int32), ('rng_seed', ctypes.c_uint32),
                ('policy', ctypes.c_int32), ('scoring', ctypes.c_int32)]


class LrgStepBuffers(ctypes.Structure):
    _fields_ = [('center', _fp), ('sample_in', _fp), ('sample_nb', _fp), ('inlier', _fp), ('neighbor', _fp),
                ('gt_remove', _fp), ('gt_add', _fp), ('add_logits', _fp), ('rmv_logits', _fp), ('workspace', _fp),
                ('workspace_bytes', ctypes.c_size_t), ('stats', _fp), ('rows_in', _fp), ('rows_nb', _fp)]


class LrgPackedBuffers(ctypes.Structure):
    _fields_ = [('center', _fp), ('sample_in', _fp), ('sample_nb', _fp), ('x_in', _fp), ('x_nb', _fp),
                ('row_slot_in', _fp), ('row_slot_nb', _fp), ('upd_in', _fp), ('upd_nb', _fp), ('rmv_logits', _fp),
                ('add_logits', _fp), ('slot_rows', _fp), ('counters', _fp), ('workspace', _fp),
                ('workspace_bytes', ctypes.c_size_t), ('stats', _fp), ('row_cap', ctypes.c_int32), ('rooms_have_pvox', ctypes.c_int32), ('slot_big', _fp), ('phase_ticks', _fp)]


class LrgAsyncBuffers(ctypes.Structure):
    _fields_ = [('queue', _fp), ('queue_bytes', ctypes.c_size_t), ('sync', _fp), ('front_workgroups', ctypes.c_int32), ('teams', ctypes.c_int32),
                ('compute_units', ctypes.c_int32), ('poll_sleep', ctypes.c_int32), ('branch_parts', ctypes.c_int32), ('gemv_units', ctypes.c_int32), ('room_queue', _fp), ('work', _fp),
                ('fill_list', _fp), ('fill_best', _fp), ('fill_sync', _fp), ('fill_label_base', _fp), ('fill_out_base', _fp), ('fill_rooms', ctypes.c_int32), ('fill_wgs', ctypes.c_int32), ('rows16', ctypes.c_int32), ('speculate', ctypes.c_int32), ('branch_waves', ctypes.c_int32), ('start_wait_us', ctypes.c_int32),
                ('pool_rows', _fp), ('pool_rows_bytes', ctypes.c_size_t), ('debug_ticks', _fp),
                ('tail_ctl', _fp), ('tail_rows', ctypes.c_int32), ('tail_close_us', ctypes.c_int32)]


class LrgFillJob(ctypes.Structure):
    _fields_ = [('points', _fp), ('label_in', _fp), ('label_out', _fp), ('n', ctypes.c_int32), ('reserved', ctypes.c_int32)]


class LrgBeamGroup(ctypes.Structure):
    _fields_ = [('parent', _fp), ('cap', ctypes.c_int32), ('room', ctypes.c_int32), ('seed', ctypes.c_int32), ('level', ctypes.c_int32),
                ('stuck', ctypes.c_int32), ('steps', ctypes.c_int32), ('nq', ctypes.c_int32), ('pending', ctypes.c_int32),
                ('done', ctypes.c_int32), ('seq_mn', ctypes.c_int32 * 3), ('seq_mx', ctypes.c_int32 * 3),
                ('q_count', ctypes.c_int32 * 16), ('q_parent', ctypes.c_int32 * 16), ('q_mn', ctypes.c_int32 * 48),
                ('q_mx', ctypes.c_int32 * 48), ('pad', ctypes.c_int32)]


LRG_ROW_TILE = 32
LRG_LOG_WORDS = 8
LRG_PACKED_MAX_POINTS = 32 * 4096      # lrg_grow_step_packed: rooms up to 131072 points
LRG_PACKED_AUTO_POINTS = 32 * 4096     # ... chosen by default up to its limit (round 2 stopped at 65536: the chunk-parallel scans of lrg_grow_step
                                       # won on 100 k-point scenes then; eight KITTI-shaped scenes under the trained weights and the Bernoulli
                                       # policy now: packed 88 k against 69 k instance-steps/s, profiles/r03_kitti_*.json)
LRG_FREE_RUN_AUTO_POINTS = 32 * 4096   # free-running launches by default up to the packed limit too: eight 100 k-point scenes, one front workgroup per
                                       # scene, one team per CU: 108 k instance-steps/s against 88 k lock-step (profiles/r03_kitti2_*.json)
LRG_FREE_RUN_AUTO_SLOTS = 480           # ... and up to this many slots in flight.  Fixed work of 2 176 rooms, rooms/s free-running | lock-step (round 5, shared tail tiles --
                                       # branch and head stacks -- and 44 front workgroups from 224 slots on: profiles/r05_tail_heads*.txt; lock-step:
                                       # profiles/r04_teams_units_sweep.txt, r05_b_bench_default.json): 136: 721, 272: 864-882 | 834, 320: 898 | 865, 400: 884-911 | 894-899,
                                       # 480: 898-903, 544: 906-911 | 926-934 (round 4: 836-857 at 272, the crossover at 300)
LRG_VGRID_MAX_CELLS = 1 << 26          # dense voxel grid of a room (LrgRoom.vgrid): at most 64 M cells (256 MB) per room ...
LRG_VGRID_TOTAL_CELLS = 1 << 31        # ... and 8 GB for the rooms of one grower
LRG_DONE_RING = 1020
LRG_STATS_WORDS = 4 + LRG_DONE_RING


class LrgHipError(RuntimeError):
    pass


def build(verbose=False):
    """hipcc --offload-arch=gfx950 -> learn_region_grow_amd/liblrg_hip.so (cross-compiles without a GPU).
    One object per translation unit (csrc/build/, compiled side by side, only the stale ones), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    import glob
    headers = sorted(glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(CSRC, '*.inl'))) + \
        [os.path.join(os.path.dirname(HERE), 'include', 'lrg_hip.h')]
    hnew = max(os.path.getmtime(h) for h in headers)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    # -ffp-contract=off: no implicit FMA fusion, so the kernels that restate NumPy float32 arithmetic (voxel keys,
    # fill-in distances, ball query) round exactly like the reference; hot loops use explicit fmaf / MFMA.
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC'] + os.environ.get('LRG_HIPCC_FLAGS', '').split()
    bdir = os.path.join(CSRC, 'build')
    os.makedirs(bdir, exist_ok=True)
    stamp = os.path.join(bdir, 'flags.txt')
    if not os.path.exists(stamp) or open(stamp).read() != ' '.join(flags):
        for o in glob.glob(os.path.join(bdir, '*.o')):
            os.remove(o)
        with open(stamp, 'w') as f:
            f.write(' '.join(flags))
    objs, jobs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(bdir, s.replace('.hip', '.o'))
        objs.append(obj)
        if not os.path.exists(obj) or os.path.getmtime(obj) < max(hnew, os.path.getmtime(src)):
            jobs.append([hipcc] + flags + ['-c', '-o', obj, src])
    if not jobs and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(o) for o in objs):
        return LIB_PATH

    def run(cmd):
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd, stderr=None if verbose else subprocess.DEVNULL)
    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    run([hipcc, '--offload-arch=gfx950', '-fPIC', '-shared', '-o', LIB_PATH] + objs)
    return LIB_PATH


_SIGS = {
    'lrg_abi_version': (ctypes.c_int, []),
    'lrg_target_arch': (ctypes.c_char_p, []),
    'lrg_struct_size': (ctypes.c_size_t, [ctypes.c_int]),
    'lrg_packed_weights_bytes': (ctypes.c_size_t, [ctypes.POINTER(LrgWeights)]),
    'lrg_pack_weights': (ctypes.c_int, [ctypes.POINTER(LrgWeights), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    'lrg_forward_workspace_bytes': (ctypes.c_size_t, [ctypes.POINTER(LrgWeights), ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    'lrg_forward': (ctypes.c_int, [ctypes.POINTER(LrgWeights), _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp,
                                   _fp, ctypes.c_size_t, ctypes.c_uint, _fp]),
    'lrg_forward_rows': (ctypes.c_int, [ctypes.POINTER(LrgWeights), _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp,
                                        _fp, _fp, _fp, ctypes.c_size_t, ctypes.c_uint, _fp]),
    'lrg_forward_workspace_view': (ctypes.c_int, [ctypes.POINTER(LrgWeights), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t),
                                                  ctypes.POINTER(ctypes.c_size_t)]),
    'lrg_pointwise_layer': (ctypes.c_int, [_fp, ctypes.c_int, _fp, ctypes.c_int, _fp, _fp, ctypes.c_long, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, ctypes.c_int, _fp]),
    'lrg_segmax': (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp]),
    'lrg_head_pool_gemv': (ctypes.c_int, [_fp, _fp, ctypes.c_int, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp]),
    'lrg_head_final': (ctypes.c_int, [_fp, _fp, _fp, _fp, ctypes.c_long, ctypes.c_int, _fp]),
    'lrg_voxelize': (ctypes.c_int, [_fp, ctypes.c_int, ctypes.c_int, ctypes.c_float, _fp, _fp]),
    'lrg_bind_group': (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp]),
    'lrg_voxel_pack': (ctypes.c_int, [_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp]),
    'lrg_voxel_grid_build': (ctypes.c_int, [_fp] + [ctypes.c_int] * 7 + [_fp, _fp]),
    'lrg_voxel_hash_build': (ctypes.c_int, [_fp, ctypes.c_int, _fp, _fp, ctypes.c_int, _fp, _fp]),
    'lrg_bbox_stop': (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(LrgGrowParams), _fp]),
    'lrg_advance': (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.POINTER(LrgGrowParams), _fp, _fp]),
    'lrg_box_query': (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(LrgGrowParams), _fp]),
    'lrg_median': (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.POINTER(LrgGrowParams), _fp, _fp]),
    'lrg_sample': (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.POINTER(LrgGrowParams), _fp, _fp, _fp]),
    'lrg_gather_center': (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.POINTER(LrgGrowParams), _fp, _fp, _fp, _fp, _fp,
                                         _fp, _fp, _fp, _fp, _fp]),
    'lrg_prepare': (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.POINTER(LrgGrowParams), _fp, _fp, _fp, _fp, _fp, _fp, _fp,
                                   _fp, _fp, _fp, _fp]),
    'lrg_mask_update': (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.POINTER(LrgGrowParams), _fp, _fp, _fp, _fp, _fp,
                                       _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    'lrg_grow_step': (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(LrgGrowParams), ctypes.POINTER(LrgWeights),
                                     ctypes.POINTER(LrgStepBuffers), ctypes.c_int, ctypes.c_uint, _fp]),
    'lrg_forward_packed_workspace_bytes': (ctypes.c_size_t, [ctypes.POINTER(LrgWeights), ctypes.c_int, ctypes.c_int]),
    'lrg_forward_packed_pooled_view': (ctypes.c_int, [ctypes.POINTER(LrgWeights), ctypes.c_int, ctypes.c_int,
                                                      ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    'lrg_forward_packed': (ctypes.c_int, [ctypes.POINTER(LrgWeights), _fp, _fp, _fp, _fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int,
                                          _fp, _fp, _fp, ctypes.c_size_t, ctypes.c_uint, _fp]),
    'lrg_packed_rows_center': (ctypes.c_void_p, [ctypes.POINTER(LrgGrowParams), ctypes.POINTER(LrgPackedBuffers)]),
    'lrg_grow_step_packed': (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(LrgGrowParams),
                                            ctypes.POINTER(LrgWeights), ctypes.POINTER(LrgPackedBuffers), _fp]),
    'lrg_grow_async_queue_bytes': (ctypes.c_size_t, [ctypes.c_int]),
    'lrg_grow_async_tail_bytes': (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    'lrg_grow_async': (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(LrgGrowParams), ctypes.POINTER(LrgWeights),
                                      ctypes.POINTER(LrgPackedBuffers), ctypes.POINTER(LrgAsyncBuffers), ctypes.c_int, ctypes.c_int, _fp]),
    'lrg_front_step': (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(LrgGrowParams),
                                      ctypes.POINTER(LrgWeights), ctypes.POINTER(LrgPackedBuffers), _fp]),
    'lrg_step_graph_create': (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(LrgGrowParams),
                                             ctypes.POINTER(LrgWeights), ctypes.POINTER(LrgPackedBuffers), ctypes.c_int, _fp,
                                             ctypes.POINTER(ctypes.c_void_p)]),
    'lrg_step_graph_launch': (ctypes.c_int, [_fp, _fp]),
    'lrg_step_graph_destroy': (ctypes.c_int, [_fp]),
    'lrg_stream_create_cu_mask': (ctypes.c_int, [_fp, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    'lrg_stream_destroy': (ctypes.c_int, [_fp]),
    'lrg_beam_advance': (ctypes.c_int, [_fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(LrgGrowParams), _fp, _fp]),
    'lrg_beam_level': (ctypes.c_int, [_fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(LrgGrowParams),
                                      ctypes.POINTER(LrgWeights), ctypes.POINTER(LrgStepBuffers), ctypes.c_uint, _fp]),
    'lrg_nn1_fill': (ctypes.c_int, [_fp, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp]),
    'lrg_nn1_fill_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int]),
    'lrg_nn1_fill_ws': (ctypes.c_int, [_fp, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, ctypes.c_size_t, _fp]),
    'lrg_nn1_fill_batch_workspace_bytes': (ctypes.c_size_t, [ctypes.POINTER(LrgFillJob), ctypes.c_int]),
    'lrg_nn1_fill_batch': (ctypes.c_int, [ctypes.POINTER(LrgFillJob), ctypes.c_int, ctypes.c_int, _fp, ctypes.c_size_t, _fp]),
    'lrg_query_ball_point': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, _fp,
                                            _fp, _fp, _fp, _fp]),
    'lrg_query_ball_group': (ctypes.c_int, [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_int, _fp, _fp, _fp, _fp, _fp, _fp, _fp, ctypes.c_int, _fp]),
    'lrg_selection_sort': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, _fp]),
    'lrg_group_point': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp,
                                       _fp, _fp]),
    'lrg_group_point_grad': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp,
                                            _fp, _fp, _fp]),
    'lrg_knn_topk': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, _fp, _fp]),
    'lrg_pairwise_sqdist': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, _fp]),
    'lrg_gemm_f32': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, ctypes.c_int, ctypes.c_int, _fp, ctypes.c_int, ctypes.c_int,
                                    _fp, ctypes.c_int, _fp, _fp, ctypes.c_int, _fp]),
    'lrg_ce_grad': (ctypes.c_int, [_fp, _fp, ctypes.c_long, ctypes.c_float, ctypes.c_float, _fp, _fp, _fp]),
    'lrg_pool_backward': (ctypes.c_int, [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp]),
    'lrg_segment_colsum': (ctypes.c_int, [_fp, ctypes.c_long, ctypes.c_int, ctypes.c_int, _fp, _fp]),
    'lrg_adam_step': (ctypes.c_int, [_fp, _fp, _fp, _fp, ctypes.c_long, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _fp]),
    'lrg_preprocess_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int]),
    'lrg_preprocess': (ctypes.c_int, [_fp, ctypes.c_int, _fp, _fp, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, _fp,
                                      ctypes.c_size_t, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    'lrg_preprocess_status': (ctypes.c_int, [_fp, ctypes.c_int, ctypes.POINTER(ctypes.c_int32), _fp]),
    'lrg_grow_async_pool_rows_bytes': (ctypes.c_size_t, [ctypes.POINTER(LrgWeights), ctypes.c_int]),
    'lrg_preprocess_unsafe_normals': (ctypes.c_int, [_fp, ctypes.c_int, ctypes.c_int, _fp, _fp]),
}

EXPORTS = sorted(_SIGS)
_lib = None


def load():
    """Load liblrg_hip.so and bind every declared entry point; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LrgHipError('%s not found: run `python -c "import __graft_entry__ as g; g.build()"` (hipcc, gfx950). '
                          'There is no CPU fallback.' % LIB_PATH)
    # torch first: its bundled HIP runtime must be the one in the process (device memory and streams come from
    # torch); loading this library before torch would bind it to a second copy of libamdhip64.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.lrg_abi_version() != LRG_ABI_VERSION:
        raise LrgHipError('ABI version mismatch: the library is %d, this binding %d' % (lib.lrg_abi_version(), LRG_ABI_VERSION))
    for which, st in enumerate((LrgWeights, LrgRoom, LrgSlot, LrgGrowParams, LrgStepBuffers, LrgPackedBuffers, LrgBeamGroup, LrgAsyncBuffers, LrgFillJob)):
        if lib.lrg_struct_size(which) != ctypes.sizeof(st):
            raise LrgHipError('struct layout mismatch for %s: C %d vs ctypes %d' %
                              (st.__name__, lib.lrg_struct_size(which), ctypes.sizeof(st)))
    _lib = lib
    return lib


def check(rc, what):
    if rc == LRG_ERESIDENCY:
        raise LrgHipError('%s refused (LRG_ERESIDENCY): the launch is one workgroup per compute unit and all of them must run at once, but the '
                          'kernel does not fit a CU or the stream may use fewer CUs than the launch has workgroups (a CU-masked stream: pass the '
                          'number of CUs it may use as LrgAsyncBuffers.compute_units)' % what)
    if rc != 0:
        raise LrgHipError('%s failed with code %d' % (what, rc))
