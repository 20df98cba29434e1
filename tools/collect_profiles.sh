#!/bin/bash
# Run on the GPU box from the repo root: the rocprofv3 evidence of the round -> gpurun_out/r5p/ (copied into profiles/ by hand)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() { # name, cmd...
  local name=$1; shift
  (cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o p -- "$@" > $O/$name.log 2>&1)
  cp $O/$name/*kernel_stats.csv $O/${name}_kernel_stats.csv 2>/dev/null || find $O/$name -name "*kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \;
}
pmc() { # name, counters, cmd...
  local name=$1; local ctr=$2; shift; shift
  (cd $R && rocprofv3 --pmc $ctr --output-format csv -d $O/$name -o p -- "$@" > $O/$name.log 2>&1)
}
prof bench python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --repeats 5
prof bench_transformer python bench.py --mode transformer --no-cpu-baseline --repeats 3
prof bench_transformer_exact python bench.py --mode transformer --exact --no-cpu-baseline --repeats 3
prof train python bench.py --mode train --steps 20 --warmup 5 --loss l1 --no-cpu-baseline --repeats 3
prof siblings python tools/seqt_probe.py --reps 3 --clips 64,256 --parity 17
prof detector python bench.py --mode detect --no-cpu-baseline
pmc pmc_fetch FETCH_SIZE python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --repeats 3
pmc pmc_write WRITE_SIZE python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --repeats 3
pmc pmc_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --repeats 3
pmc pmc_mfma_tr "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" python bench.py --mode transformer --no-cpu-baseline --repeats 3
pmc pmc_mfma_tr_exact "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" python bench.py --mode transformer --exact --no-cpu-baseline --repeats 3
pmc pmc_mfma_sib "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" python tools/seqt_probe.py --reps 3 --clips 256 --parity 17
pmc pmc_fetch_tr FETCH_SIZE python bench.py --mode transformer --no-cpu-baseline --repeats 2
pmc pmc_write_tr WRITE_SIZE python bench.py --mode transformer --no-cpu-baseline --repeats 2
pmc pmc_mfma_train "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" python bench.py --mode train --steps 20 --warmup 5 --loss l1 --no-cpu-baseline --repeats 3
pmc pmc_mfma_det "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" python bench.py --mode detect --no-cpu-baseline --steps 4 --warmup 1
cd $R
python tools/pmc_reduce.py shapes $O/pmc_fetch $O/pmc_write 160,400,640 profiles/r4_pmc_traffic.json > $O/pmc_traffic.json 2>&1
python tools/pmc_reduce.py traffic $O/pmc_fetch_tr $O/pmc_write_tr seqt_forward > $O/pmc_traffic_seqt.json 2>&1
python tools/pmc_reduce.py mfma $O/pmc_mfma > $O/mfma_util_bench.json 2>&1
python tools/pmc_reduce.py mfma $O/pmc_mfma_tr > $O/mfma_util_transformer.json 2>&1
python tools/pmc_reduce.py mfma $O/pmc_mfma_tr_exact > $O/mfma_util_transformer_exact.json 2>&1
python tools/pmc_reduce.py mfma $O/pmc_mfma_sib > $O/mfma_util_siblings.json 2>&1
python tools/pmc_reduce.py mfma $O/pmc_mfma_train > $O/mfma_util_train.json 2>&1
python tools/pmc_reduce.py mfma $O/pmc_mfma_det > $O/mfma_util_detector.json 2>&1
# the bench lines themselves (no profiler attached)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python bench.py --mode transformer > $O/bench_transformer.json 2> $O/bench_transformer.err
python bench.py --mode transformer --exact > $O/bench_transformer_exact.json 2> $O/bench_transformer_exact.err
python bench.py --mode train --steps 20 --warmup 5 --loss l1 > $O/bench_train_b32.json 2> $O/bench_train_b32.err
python bench.py --mode train --force-dist --steps 20 --warmup 5 --loss l1 --no-cpu-baseline > $O/bench_train_b32_forcedist.json 2> $O/bench_train_b32_forcedist.err
python bench.py --mode detect > $O/bench_detect16.json 2> $O/bench_detect16.err
python bench.py --mode detect --batch 1 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_detect1.json 2> $O/bench_detect1.err
PYTHONPATH=$R python tools/sort_time.py > $O/detector_sort_time.txt 2>&1
python tools/detector_full_time.py 1 2 4 16 > $O/detector_full_time.txt 2>&1
# per-launch durations of the persistent forward out of the kernel trace (the stats CSV only has the mean over all launch shapes)
python - <<'PY' > $O/bench_xcd_forward_launches.csv 2>&1
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r5p"
for path in glob.glob(O + "/bench/**/*kernel_trace.csv", recursive=True):
    print("dispatch,kernel,duration_ns")
    for r in csv.DictReader(open(path)):
        if "opnet_xcd_forward" in r["Kernel_Name"]:
            print(f'{r["Dispatch_Id"]},{r["Kernel_Name"].split("(")[0].replace("void ", "")},{int(r["End_Timestamp"]) - int(r["Start_Timestamp"])}')
PY
# the detector pass by layer, the one-frame call kernel by kernel, the served transformer pass
bash tools/detector_layer_times.sh > $O/detlayers.log 2>&1; cp $R/gpurun_out/detlayers/conv_dispatches.txt $O/detector_conv_dispatches.txt
bash tools/detector_one_frame_trace.sh > $O/det1.log 2>&1; cp $R/gpurun_out/det1/one_frame_kernels.txt $O/detector_one_frame_kernels.txt
python tools/transformer_serving_time.py 4 > $O/transformer_serving_time.txt 2>&1
python tools/transformer_pass_kernels.py 256 4 > $O/transformer_pass_kernels_256.txt 2>&1
[ -x tools/probes/ffn_probe ] && timeout 300 tools/probes/ffn_probe > $O/ffn_probe.txt 2>&1
python tools/transformer_server_host_time.py 256 > $O/transformer_server_host_time.txt 2>&1
for d in 1 2 3 4; do echo "passes in flight: $d"; OPDET_IN_FLIGHT=$d python bench.py --mode detect --no-cpu-baseline --steps 12 --warmup 4 2>/dev/null | tail -1 | cut -c1-260; OPDET_IN_FLIGHT=$d python bench.py --mode detect --batch 1 --steps 60 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-260; done > $O/detector_passes_in_flight.txt 2>&1
# keep only the small artefacts
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +2M -delete
ls -la $O | head -40
