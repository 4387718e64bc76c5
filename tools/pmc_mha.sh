# Counter passes over the 1024x1024 attention kernels (forward and backward), with and without dropout:
# vector-ALU / matrix / LDS instruction counts and busy cycles per kernel.  Output: gpurun_out/pmc_mha.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_mha.txt
: > $OUT
for DROP in 0 0.1; do
export SPOT_MB_DROP=$DROP
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc1 -- python3 $R/tools/microbench.py mha1 > $R/gpurun_out/pmc1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc2 -- python3 $R/tools/microbench.py mha1 > $R/gpurun_out/pmc2.log 2>&1 || exit 1
cd $R && python3 - $DROP >> $OUT <<'PY'
import csv,glob,collections,sys,re
print("== attention dropout", sys.argv[1], "(B=40, 8 heads, 1024 x 1024; counters summed over the launches of the run, per launch in brackets)")
tot = collections.defaultdict(dict)
for d in ("pmc1","pmc2"):
    for f in glob.glob("gpurun_out/%s/*/*counter_collection.csv"%d):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            m=re.search(r"mha_\w+", r["Kernel_Name"])
            if not m: continue
            k=m.group(0)
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        for k,v in agg.items():
            for c,x in v.items(): tot[k][c]=(x,len(n[k]))
for k,v in tot.items():
    print(k)
    for c,(x,nl) in sorted(v.items()): print("   %-30s %.5g   [%.5g per launch, %d launches]"%(c,x,x/nl,nl))
    if "SQ_INSTS_VALU" in v and v.get("SQ_INSTS_MFMA", (0, 0))[0] > 0:
        valu, mfma = v["SQ_INSTS_VALU"][0], v["SQ_INSTS_MFMA"][0]
        print("   vector-ALU instructions that are not matrix instructions per matrix instruction: %.3f" % ((valu - mfma) / mfma))
PY
cd /tmp
cp $(ls $R/gpurun_out/pmc2/*/*counter_collection.csv | head -1) $R/gpurun_out/pmc2_last.csv
rm -rf $R/gpurun_out/pmc1 $R/gpurun_out/pmc2
done
cat $OUT
