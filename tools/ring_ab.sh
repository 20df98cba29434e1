#!/bin/bash
# A/B of the ring layout against the full-history layout of opnet_xcd_forward on the driver's command (alternating runs)
for i in 1 2 3; do
  for r in 1 0; do
    OPNET_XCD_RING=$r python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ring=$r', d['value'], d['value_min'], d['value_max'], d['roofline']['frac'], d['roofline']['launch_ms'])"
  done
done
for r in 1 0; do OPNET_XCD_RING=$r python bench.py --steps 200 --warmup 5 --no-cpu-baseline --repeats 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('200 steps ring=$r', d['value'], d['roofline']['frac'])"; done
