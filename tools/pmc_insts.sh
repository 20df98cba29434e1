export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/pmc_insts; rm -rf $O; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_[A-Z_0-9]*\|SQ_ACTIVE_INST_[A-Z_0-9]*\|SQ_WAIT_[A-Z_0-9]*\|SQ_WAVE_CYCLES\|SQ_BUSY_CYCLES\|SQ_VALU_MFMA_BUSY_CYCLES" | sort -u | tr '\n' ' ' > $O/counters.txt
cat $O/counters.txt; echo
cd /tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d $O/p1 -o p -- python $R/tools/xcd_probe.py --batches 512 --no-chain --reps 2 > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -o p -- python $R/tools/xcd_probe.py --batches 512 --no-chain --reps 2 > $O/p2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc_insts/p1", "gpurun_out/pmc_insts/p2"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "opnet_xcd_forward" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in sorted(acc.items()):
        print(d[-2:], k, "per launch %.4g" % (v / max(n, 1)), "launches", n)
PY
tail -3 $O/p1.log
find $O -name "*.csv" -size +1M -delete
