cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmcx1 -- python3 $R/tools/microbench.py xcorr > $R/gpurun_out/pmcx1.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmcx2 -- python3 $R/tools/microbench.py xcorr > $R/gpurun_out/pmcx2.log 2>&1
cd $R && python3 - <<'PY'
import csv,glob,collections
for d in ("pmcx1","pmcx2"):
    for f in glob.glob("gpurun_out/%s/*/*counter_collection.csv"%d):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "groupdw_kernel" not in k: continue
            k=k[k.index("groupdw_kernel"):][:30]+" grid"+r.get("Grid_Size","")
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
        for k,v in agg.items():
            print(d,k)
            for c,x in sorted(v.items()): print("   %-28s %.4g  (per launch %.4g)"%(c,x,x/n[(k,c)]))
PY
rm -rf gpurun_out/pmcx1 gpurun_out/pmcx2
