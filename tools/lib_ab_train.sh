#!/bin/bash
# A/B of alternative builds of libopnet_hip.so (objectpermanence_amd/lib/alt_*.so) on the 32-clip training step, alternating runs
run() { OPNET_HIP_LIB=$1 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline --repeats ${REPS:-7} 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')]
d=json.loads(l[-1]) if l else None
print('$2', d['value'], d['ms_per_step']) if d else print('$2 FAILED')"; }
for i in 1 2; do
  run "" default
  for f in objectpermanence_amd/lib/alt_*.so; do run $PWD/$f $(basename $f .so); done
done
