#!/bin/bash
# lanes on disjoint CU sets (lrg_stream_create_cu_mask) against lanes sharing the chip
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
rm -f gpurun_out/lanes_cu.txt
for Q in 4 8; do
for L in 1 2 3 4 6; do
  for CP in 0 1; do
    [ $L -eq 1 ] && [ $CP -eq 1 ] && continue
    [ $Q -eq 8 ] && [ $L -lt 4 ] && continue
    GPU_MAX_HW_QUEUES=$Q timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --p0-rooms 0 --lanes $L --cu-partition $CP > /tmp/b.log 2>&1
    echo "hwq $Q lanes $L cu-partition $CP: $(grep '^{' /tmp/b.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f steps/s  fixed %.0f rooms/s  steady %.0f rooms/s' % (d['value'], d['rooms_per_sec'], d['rooms_per_sec_steady_cycling']))" 2>&1 | tail -1)" | tee -a gpurun_out/lanes_cu.txt
  done
done
done
