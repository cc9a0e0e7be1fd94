#!/bin/bash
# front kernel after the round-trip trims: loop tests, determinism, loop rates, phase stamps
bash tools/r02_determinism.sh 2>&1 | grep -v "^== \(three\|one\|two lanes, no voxel\)" 
echo "== phase stamps, area5"; bash tools/trace_run.sh 1 68 tools/trace_front.py 2>&1 | grep -v amdgpu.ids | tail -16
