#!/bin/bash
# usage (GPU box, repo root): tools/attn_zs_sweep.sh  -> per-kernel average times of the flash training attention at 32 clips for forced sweep splits
for zs in ${ZS_LIST:-0 3 4 5 6 8}; do
  export OPSEQ_ATTN_ZS=$zs
  bash tools/prof_stats.sh r6_zs$zs python tools/transformer_train_time.py 32 | grep -E "attention_" | sed "s/^/ZS=$zs /"
done
