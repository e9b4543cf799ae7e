cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/bt -o b -- python3 $GRAFT_REPO_ROOT/bench.py --no-children --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob("/tmp/bt/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r["Start_Timestamp"]))
# find headline steps: count_twist_wave_kernel<4,...,true> followed by norms and rowwise
out=[]
for i,r in enumerate(rows):
    n=r["Kernel_Name"].split("(")[0].replace("void kpop::","").replace("kpop::","")
    out.append((n,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,int(r["Start_Timestamp"]),int(r["End_Timestamp"])))
idx=[i for i,o in enumerate(out) if o[0].startswith("count_twist_wave_kernel<4") and o[1]>900]
for i in idx[3:5]:
    for j in range(i,min(i+5,len(out))):
        gap=(out[j][2]-out[j-1][3])/1e3 if j>i else 0
        print("%-60s %8.1f us  (gap before %5.1f)"%(out[j][0][:60],out[j][1],gap))
    print()
PY
