#!/bin/bash
# rocprofv3 kernel trace of the tracker benchmark (cfg 1); per-kernel-class time of the last 20 frames
# (frames are delimited by the tracker's decode kernel).  Output: gpurun_out/tracker_frame.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/tk
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tk -- python3 $R/tools/measure_configs.py cfg1 > $O/tk.log 2>&1 || exit 1
cd $R
python3 - <<'PY' > gpurun_out/tracker_frame.txt
import csv,glob,collections,re
f=glob.glob("gpurun_out/tk/*/*kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
dec=[i for i,r in enumerate(rows) if "tracker_decode" in r["Kernel_Name"]]
a,b=dec[-21],dec[-1]
win=rows[a+1:b+1]
span=(int(win[-1]["End_Timestamp"])-int(win[0]["Start_Timestamp"]))/1e3/20
agg=collections.defaultdict(lambda:[0,0.0])
def short(n):
    m=re.search(r"(\w+)(<[^(]*>)?\(",n)
    return (m.group(1)+(m.group(2) or ""))[:60] if m else n[:60]
for r in win:
    k=short(r["Kernel_Name"]); agg[k][0]+=1; agg[k][1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
tot=sum(v[1] for v in agg.values())/20
print("per frame: %.1f us wall between decode kernels, %.1f us of kernel time, %.1f launches" % (span, tot, len(win)/20))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1]):
    print("%-62s %6.1f launches  %8.1f us  (%5.1f us each)" % (k, v[0]/20, v[1]/20, v[1]/v[0]))
PY
rm -rf gpurun_out/tk
head -40 gpurun_out/tracker_frame.txt
tail -2 gpurun_out/tk.log
