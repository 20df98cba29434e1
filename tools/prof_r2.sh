set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/prof_r2
rm -rf $O; mkdir -p $O
cd /tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- $B > $O/kt.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- $B > $O/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- $B > $O/pmc_w.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_m -o m -- $B > $O/pmc_m.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_sib -o sib -- python $R/tools/siblings_time.py > $O/kt_sib.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sib -o m -- python $R/tools/siblings_time.py > $O/pmc_sib.log 2>&1
cd $R
python tools/pmc_reduce.py traffic $O/pmc_f $O/pmc_w opnet_xcd_forward > $O/traffic.json 2> $O/traffic.err
python tools/pmc_reduce.py mfma $O/pmc_m $O/pmc_sib > $O/mfma.json 2> $O/mfma.err
find $O -name "*kernel_stats.csv" | head
for f in $(find $O -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f; done
cat $O/traffic.json $O/traffic.err; head -60 $O/mfma.json; tail -2 $O/kt.log
# keep the merge small: drop the per-dispatch traces
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*counter_collection.csv" -size +2M -delete
du -sh $O
