#!/bin/bash
# Counter passes over the cross-correlation microbenchmark (tools/microbench.py xcorr): wave / issue / LDS counters in
# two SQ passes, HBM bytes in two TCC passes (FETCH_SIZE and WRITE_SIZE do not fit one).  Prints per-kernel sums and
# per-launch averages for every groupdw* kernel; FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests at 64).
#   usage: tools/pmc_xcorr.sh > profiles/<name>.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmcx1 -- python3 $R/tools/microbench.py xcorr > $R/gpurun_out/pmcx1.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmcx2 -- python3 $R/tools/microbench.py xcorr > $R/gpurun_out/pmcx2.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcx3 -- python3 $R/tools/microbench.py xcorr > $R/gpurun_out/pmcx3.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcx4 -- python3 $R/tools/microbench.py xcorr > $R/gpurun_out/pmcx4.log 2>&1
cd $R && python3 - <<'PY'
import csv,glob,collections
print("# rocprofv3 --pmc over `python3 tools/microbench.py xcorr` (4 passes); SQ_* in quad-cycles summed over all waves / SEs")
for d in ("pmcx1","pmcx2","pmcx3","pmcx4"):
    for f in glob.glob("gpurun_out/%s/*/*counter_collection.csv"%d):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "groupdw" not in k: continue
            k=k[k.index("groupdw"):][:36]+" grid "+r.get("Grid_Size","")
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
        for k,v in sorted(agg.items()):
            for c,x in sorted(v.items()):
                per=x/n[(k,c)]
                note=""
                if c=="FETCH_SIZE": note="  -> %.1f MB read per launch (x2 correction, KiB units)"%(per*2*1024/1e6)
                if c=="WRITE_SIZE": note="  -> %.1f MB written per launch (KiB units)"%(per*1024/1e6)
                print("%-48s %-24s launches %3d  per launch %.5g%s"%(k,c,n[(k,c)],per,note))
PY
tail -8 gpurun_out/pmcx1.log
rm -rf gpurun_out/pmcx1 gpurun_out/pmcx2 gpurun_out/pmcx3 gpurun_out/pmcx4
