cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/kt
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt -- python3 $R/tools/microbench.py gemm > $R/gpurun_out/kt.log 2>&1 || exit 1
cd $R
python3 - <<'PY'
import csv,glob,collections,re
f=glob.glob("gpurun_out/kt/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
# group consecutive launches of gemm kernels by grid size + name, report median duration
groups=collections.OrderedDict()
for r in rows:
    n=r["Kernel_Name"]
    if "gemm" not in n: continue
    k=(re.sub(r"\(.*","",n)[-60:], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), r.get("Grid_Size_Z","") )
    groups.setdefault(k,[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in groups.items():
    v.sort()
    print("%-62s grid %8s z %4s  n %4d  median %8.1f us  min %8.1f" % (k[0],k[1],k[2],len(v),v[len(v)//2],v[0]))
PY
rm -rf gpurun_out/kt
