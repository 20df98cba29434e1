#!/bin/bash
# rocprofv3 over NOTHING BUT throughput-form passes of 256 one-clip transformer_lstm requests (tools/transformer_pass_only.py): the
# matrix-pipe busy fraction (own PMC pass) and the kernel stats of every kernel of such a pass -> gpurun_out/trpass/
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/trpass
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc -o p -- python tools/transformer_pass_only.py 256 6 4 > $O/pmc.log 2>&1)
(cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python tools/transformer_pass_only.py 256 6 4 > $O/kt.log 2>&1)
cd $R
python tools/pmc_reduce.py mfma $O/pmc > $O/mfma_util_transformer_pass256.json 2>&1
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/transformer_pass256_kernel_stats.csv \;
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
head -14 $O/transformer_pass256_kernel_stats.csv | cut -c1-130
python - <<PY
import json
d = json.load(open("$O/mfma_util_transformer_pass256.json"))
for k, v in d["kernels"].items():
    if v["mfma_util"] > 0:
        print(k, v["mfma_util"])
PY
