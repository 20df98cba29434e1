#!/bin/bash
# Run on the GPU box from the repo root: HBM-side traffic of the headline kernel per launch shape (bench.py's roofline.traffic) ->
# gpurun_out/r6_pmc_traffic.json (copy to profiles/).  Two separate --pmc passes over the driver's command, no trace domains.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd $R && timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/$c -o p -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --repeats 3 > $O/$c.log 2>&1)
done
cd $R
python tools/pmc_reduce.py shapes $O/FETCH_SIZE $O/WRITE_SIZE 160,400,640,1024 profiles/r5_pmc_traffic.json > gpurun_out/r6_pmc_traffic.json 2> gpurun_out/r6_pmc_traffic.err
head -c 1500 gpurun_out/r6_pmc_traffic.json; cat gpurun_out/r6_pmc_traffic.err | tail -3
