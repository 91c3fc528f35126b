cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_sweep
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sweep -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/sweep_probe.py --reps 3 --variants 13 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sweep2 -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/sweep_probe.py --reps 3 --variants 13 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc_sweep", "gpurun_out/pmc_sweep2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in agg.items():
            if "sweep_kernel" in k or "stats_kernel" in k or "ccdf_kernel" in k:
                print(k, {n: round(sum(v) / len(v)) for n, v in c.items()}, "launches", len(next(iter(c.values()))))
PY
