# per-kernel stats + one step's kernel timeline (gaps) of the full workload
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-trace_r04}; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o t -- python3 bench.py --steps 10 --warmup 3 --workload ${WL:-full} --no-cpu-baseline --no-kernel-pass > $O/bench.json 2> $O/err.txt
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("$O/stats/t_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# find last k_adam -> one full step = from after an adam to next adam
idx=[i for i,r in enumerate(rows) if "k_adam" in r["Kernel_Name"]]
a,b=idx[-3],idx[-2]
t0=int(rows[a]["End_Timestamp"])
prev=t0; tot=0
for r in rows[a+1:b+1]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    name=r["Kernel_Name"][:70]
    print(f"{(s-t0)/1e3:9.1f}us gap {(s-prev)/1e3:6.1f} dur {(e-s)/1e3:8.1f}  {name}")
    prev=e
print("step span us", (int(rows[b]["End_Timestamp"])-t0)/1e3)
PY
