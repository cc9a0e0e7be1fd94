#!/bin/bash
# How many kernels from different HIP streams run side by side, by HIP's hardware-queue limit (round 2 saw two).
mkdir -p gpurun_out
for q in default 8 16; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  echo "== GPU_MAX_HW_QUEUES=$q ==" >> gpurun_out/r03_stream_overlap.txt
  timeout 120 ./tools/stream_overlap.bin 0 8 2>&1 | grep "grid  64  spin 20" >> gpurun_out/r03_stream_overlap.txt
done
unset GPU_MAX_HW_QUEUES
cat gpurun_out/r03_stream_overlap.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_0_bench_baseline.json 2> gpurun_out/r03_0_bench_baseline.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03_0_bench_baseline.json').read().strip().splitlines()[-1])
print('baseline bench: %.0f %s, %.0f rooms/s fixed, %.1f us/iter' % (d['value'], d['unit'], d['rooms_per_sec'], 1e3*d['ms_per_iteration']))
PY
