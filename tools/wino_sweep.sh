#!/bin/bash
# usage (GPU box, repo root): tools/wino_sweep.sh -> detector frames/s for Winograd thresholds / workspace caps
# (16 frames per pass; one frame per pass with passes in flight; stage times of one call alone)
CFGS=("0 256 3000 8192" "1 256 3000 8192" "1 256 3000 2048" "1 256 3000 1024" "1 128 3000 2048" "1 256 800 2048" "1 256 200 2048")
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  export OPDET_WINOGRAD=$1 OPDET_WINO_MIN_CIN=$2 OPDET_WINO_MIN_TILES=$3 OPDET_WINO_WS_MB=$4
  a=$(python bench.py --mode detect --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  a2=$(python bench.py --mode detect --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  b=$(python bench.py --mode detect --batch 1 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  c=$(python tools/detector_full_time.py 1 2>/dev/null | tail -1 | cut -c1-90)
  echo "winograd=$1 min_cin=$2 min_tiles=$3 ws_mb=$4 : 16/pass $a $a2 frames/s ; 1/pass (in flight) $b ; alone: $c"
done
