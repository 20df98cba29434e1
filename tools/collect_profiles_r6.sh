#!/bin/bash
# Run on the GPU box from the repo root: the rocprofv3 evidence of round 6 -> gpurun_out/r6p/ (copied into profiles/ as r6_*)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() { # name, cmd...
  local name=$1; shift
  (cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o p -- "$@" > $O/$name.log 2>&1)
  find $O/$name -name "*kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \;
}
pmc() { # name, counters, cmd...
  local name=$1; local ctr=$2; shift; shift
  (cd $R && rocprofv3 --pmc $ctr --output-format csv -d $O/$name -o p -- "$@" > $O/$name.log 2>&1)
}
MF="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
prof bench python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --repeats 5
prof train python bench.py --mode train --steps 20 --warmup 5 --loss l1 --no-cpu-baseline --repeats 3
prof transformer_train_b32 python tools/transformer_train_time.py 32
prof transformer_train_b1 python tools/transformer_train_time.py 1
prof siblings_train python tools/baseline_train_time.py 1 32
pmc pmc_mfma_tt32 "$MF" python tools/transformer_train_time.py 32
pmc pmc_mfma_tt1 "$MF" python tools/transformer_train_time.py 1
pmc pmc_mfma_sibtrain "$MF" python tools/baseline_train_time.py 32
pmc pmc_mfma "$MF" python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --repeats 3
pmc pmc_fetch FETCH_SIZE python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --repeats 3
pmc pmc_write WRITE_SIZE python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --repeats 3
pmc pmc_fetch_tt32 FETCH_SIZE python tools/transformer_train_time.py 32
pmc pmc_write_tt32 WRITE_SIZE python tools/transformer_train_time.py 32
cd $R
python tools/pmc_reduce.py mfma $O/pmc_mfma_tt32 > $O/mfma_util_transformer_train_b32.json 2>&1
python tools/pmc_reduce.py mfma $O/pmc_mfma_tt1 > $O/mfma_util_transformer_train_b1.json 2>&1
python tools/pmc_reduce.py mfma $O/pmc_mfma_sibtrain > $O/mfma_util_siblings_train.json 2>&1
python tools/pmc_reduce.py mfma $O/pmc_mfma > $O/mfma_util_bench.json 2>&1
python tools/pmc_reduce.py shapes $O/pmc_fetch $O/pmc_write 160,400,640,1024 profiles/r5_pmc_traffic.json > $O/pmc_traffic.json 2>&1
python tools/pmc_reduce.py traffic $O/pmc_fetch_tt32 $O/pmc_write_tt32 attention_bwd > $O/pmc_traffic_attention_bwd.json 2>&1
python tools/pmc_reduce.py traffic $O/pmc_fetch_tt32 $O/pmc_write_tt32 attention_train_fwd > $O/pmc_traffic_attention_train_fwd.json 2>&1
# the bench lines themselves (no profiler attached)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python bench.py --mode train --steps 20 --warmup 5 --loss l1 > $O/bench_train_b32.json 2> $O/bench_train_b32.err
python bench.py --mode train --global-batch 256 --steps 10 --warmup 3 --loss l1 --no-cpu-baseline > $O/bench_train_global256_n1.json 2> $O/bench_train_global256_n1.err
python bench.py --mode train --force-dist --steps 20 --warmup 5 --loss l1 --no-cpu-baseline > $O/bench_train_b32_forcedist.json 2> $O/bench_train_b32_forcedist.err
python bench.py --mode transformer > $O/bench_transformer.json 2> $O/bench_transformer.err
python bench.py --mode detect > $O/bench_detect16.json 2> $O/bench_detect16.err
python bench.py --mode detect --batch 1 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_detect1.json 2> $O/bench_detect1.err
python tools/transformer_train_time.py 1 8 16 32 > $O/transformer_train_time.txt 2>&1
python tools/baseline_train_time.py 1 16 32 > $O/siblings_train_time.txt 2>&1
python tools/train_time.py 32 48 64 96 128 256 2>&1 | grep "^B=" > $O/opnet_train_time.txt
python - <<'PY' > $O/bench_xcd_forward_launches.csv 2>&1
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r6p"
for path in glob.glob(O + "/bench/**/*kernel_trace.csv", recursive=True):
    print("dispatch,kernel,duration_ns")
    for r in csv.DictReader(open(path)):
        if "opnet_xcd_forward" in r["Kernel_Name"]:
            print(f'{r["Dispatch_Id"]},{r["Kernel_Name"].split("(")[0].replace("void ", "")},{int(r["End_Timestamp"]) - int(r["Start_Timestamp"])}')
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +2M -delete
ls $O | head -60
