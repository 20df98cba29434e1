#!/bin/bash
# A/B of the yielding-gap parameters of seqt_forward (builds under objectpermanence_amd/lib/variant_*.so, -DST_NY2 / ST_NY3 / ST_TAIL)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { # lib, tag
  for spec in "baseline_lstm:256,384,512" "non_linear_lstm:128,192,256"; do
    m=${spec%%:*}; c=${spec#*:}
    OPNET_HIP_LIB=$1 python tools/seqt_probe.py --models $m --clips $c --parity "" --reps 5 2>/dev/null | grep throughput | awk -v t="$2" '{print t, $1, $4, $13, $14}'
  done
}
run "" default
for f in objectpermanence_amd/lib/variant_*.so; do run $PWD/$f $(basename $f .so); done
run "" default
