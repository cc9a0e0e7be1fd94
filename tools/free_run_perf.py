#!/usr/bin/env python3
"""Steady-state rate of the free-running launches (lrg_grow_async) on the 68-room Area-5-shaped set, by front workgroups, tile teams
and steps per launch -- next to the lock-step iterations in the same process.
    python tools/free_run_perf.py [--rooms 68] [--seconds 1.5] [--configs fronts:teams:steps[:budget_us],...]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rooms', type=int, default=68)
    ap.add_argument('--seconds', type=float, default=1.5)
    ap.add_argument('--workload', default='area5')
    ap.add_argument('--configs', default='0:0:64,68:3:64,34:3:64,17:3:64,34:2:64,34:1:64,34:3:16,34:3:256')
    ap.add_argument('--lockstep', type=int, default=1)
    ap.add_argument('--out', default=None)
    ap.add_argument('--in-flight', type=int, default=0, help='rooms in flight (default: --rooms)')
    ap.add_argument('--jobs', type=int, default=0, help='> 0: that many room jobs (the geometries x random-stream keys) through the slots, reset -> labels')
    args = ap.parse_args()
    import torch
    from learn_region_grow_amd import synthetic, workloads
    from learn_region_grow_amd.lrgnet import LrgNetHIP
    from learn_region_grow_amd.grow import RegionGrower, LanedRegionGrower
    dev = torch.device('cuda:0')
    weights = synthetic.load_trained_weights()
    extra = {}
    if args.workload == 'kitti':
        rooms = workloads.kitti_scenes(min(args.rooms, 8), seed_base=5000, cache_dir='/tmp/lrg_cache')
        extra = dict(resolution=0.3, packed=True)
    elif args.workload == 'scannet':
        rooms = workloads.scannet_rooms(min(args.rooms, 39), seed_base=7000, cache_dir='/tmp/lrg_cache')
    else:
        rooms = workloads.area5_rooms(args.rooms, seed_base=1000, cache_dir='/tmp/lrg_cache')
    net = LrgNetHIP(1, 1, 512, 512, 13, 0, device=dev, mode='fused').load_weights(weights)
    stream = torch.cuda.Stream(device=dev)
    results = []

    def breakdown(gr, row):
        b = gr.free_run_breakdown()      # LRG_FREE_RUN_DEBUG=1: ticks of 10 ns -> microseconds
        if b:
            row.update(b)

    def run(gr, label):
        with torch.cuda.stream(stream):
            gr.load_rooms(rooms)
            for g in range(gr.n_groups):
                gr.bind(g, g)

            def cycle(seconds):
                t_end = time.perf_counter() + seconds
                while time.perf_counter() < t_end:
                    gr.enqueue()
                    for g in gr.poll_done():
                        r = gr.group_room[g]
                        gr.fill(r)
                        gr.reset_room(r)
                        gr.bind(g, r)
            cycle(0.7)
            torch.cuda.synchronize()
            s0 = gr.d_stats[:4].cpu().numpy().astype(np.float64)
            t0 = time.perf_counter()
            cycle(args.seconds)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            s1 = gr.d_stats[:4].cpu().numpy().astype(np.float64)
        row = dict(config=label, steps_per_sec=(s1[2] - s0[2]) / (t1 - t0), rooms_per_sec=(s1[1] - s0[1]) / (t1 - t0), given_up=int(s1[3]))
        breakdown(gr, row)
        print(json.dumps(row), flush=True)
        results.append(row)

    def run_jobs(gr, label):
        jobs = [dict(rooms[j % len(rooms)], room_id=100000 + j) for j in range(args.jobs)]
        with torch.cuda.stream(stream):
            gr.load_rooms(jobs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gr.grow_loaded(fill=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            st = gr.d_stats[:4].cpu().numpy().astype(np.float64)
            ok = bool((gr.d_filled[:int(gr.room_off[gr.n_rooms - 1]) + gr.room_n[-1]] >= 0).all())
            lab = int(sum(int((gr.d_filled[int(gr.room_off[k]):int(gr.room_off[k]) + gr.room_n[k]] > 0).all()) for k in range(gr.n_rooms)))
        row = dict(config=label, jobs=args.jobs, seconds=t1 - t0, rooms_per_sec=args.jobs / (t1 - t0), steps_per_sec=st[2] / (t1 - t0),
                   rooms_fully_labeled=lab, given_up=int(st[3]))
        breakdown(gr, row)
        print(json.dumps(row), flush=True)
        results.append(row)

    kw = dict(rooms_in_flight=args.in_flight or len(rooms), rng='counter', policy='net', seed=0, **extra)
    if args.jobs:
        if args.lockstep:
            from learn_region_grow_amd.grow import LanedRegionGrower
            lg = LanedRegionGrower(net, free_run=False, graph_iterations=4, **kw)
            jobs = [dict(rooms[j % len(rooms)], room_id=100000 + j) for j in range(args.jobs)]
            lg.load_rooms(jobs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lg.grow_loaded(fill=True)
            torch.cuda.synchronize()
            print(json.dumps(dict(config='lock-step, lanes auto', jobs=args.jobs, rooms_per_sec=args.jobs / (time.perf_counter() - t0))), flush=True)
        for c in args.configs.split(','):
            p = [int(x) for x in c.split(':')]
            run_jobs(RegionGrower(net, free_run=True, free_run_fronts=p[0], free_run_teams=p[1], free_run_steps=p[2],
                                  free_run_budget_us=p[3] if len(p) > 3 else 0, **kw), 'free-run jobs fronts=%d teams=%d steps=%d budget_us=%d' % (p[0], p[1], p[2], p[3] if len(p) > 3 else 0))
        if args.out:
            json.dump(results, open(args.out, 'w'), indent=1)
        return
    if args.lockstep:
        run(RegionGrower(net, free_run=False, graph_iterations=4, **kw), 'lock-step, one lane, graph x4')
    for c in args.configs.split(','):
        p = [int(x) for x in c.split(':')]
        fronts, teams, steps = p[0], p[1], p[2]
        budget = p[3] if len(p) > 3 else 0
        run(RegionGrower(net, free_run=True, free_run_fronts=fronts, free_run_teams=teams, free_run_steps=steps, free_run_budget_us=budget, **kw),
            'free-run fronts=%d teams=%d steps=%d budget_us=%d' % (fronts, teams, steps, budget))
    if args.out:
        json.dump(results, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
