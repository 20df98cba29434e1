# rocprofv3 evidence of the round-2 training step (32 clips x 300 frames, one MI355X): kernel stats + MFMA utilisation.
set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/prof_r2_train
rm -rf $O; mkdir -p $O
cd /tmp
B="python $R/bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline"
$B > $O/bench_train.json 2> $O/bench_train.err
OPNET_XCD4=0 $B > $O/bench_train_chain.json 2> $O/bench_train_chain.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o train -- $B > $O/kt.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_m -o m -- $B > $O/pmc_m.log 2>&1
cd $R
python tools/pmc_reduce.py mfma $O/pmc_m > $O/mfma.json 2> $O/mfma.err
for f in $(find $O -name "*kernel_stats.csv"); do echo "== $f"; head -12 $f; done
tail -1 $O/bench_train.json | cut -c1-400; tail -1 $O/bench_train_chain.json | cut -c1-200
head -40 $O/mfma.json; cat $O/mfma.err | tail -3
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*counter_collection.csv" -size +2M -delete
du -sh $O
