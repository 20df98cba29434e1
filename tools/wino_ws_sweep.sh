#!/bin/bash
# usage (GPU box, repo root): tools/wino_ws_sweep.sh -> detector frames/s over the Winograd workspace cap (chunks of tiles)
for mb in 1024 512 256 128 64 32; do
  export OPDET_WINO_WS_MB=$mb
  a=$(python bench.py --mode detect --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  a2=$(python bench.py --mode detect --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  b=$(python bench.py --mode detect --batch 1 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  c=$(python tools/detector_full_time.py 1 2>/dev/null | tail -1 | cut -c1-90)
  echo "ws_mb=$mb : 16/pass $a $a2 frames/s ; 1/pass (in flight) $b ; alone: $c"
done
