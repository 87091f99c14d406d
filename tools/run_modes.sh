cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in -1 0 3 4 9 12 13; do
  rm -rf gpurun_out/it_$m
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/it_$m -- python tools/intra_probe.py --reps 2 --mode $m > gpurun_out/it_$m.log 2>&1
  f=$(ls gpurun_out/it_$m/*/*kernel_trace.csv | head -1)
  python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$f")) if "rocclr" not in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
half=rows[len(rows)//2:]
tot=(int(half[-1]["End_Timestamp"])-int(half[0]["Start_Timestamp"]))/1e3
by={}
for r in half:
    k=r["Kernel_Name"].split("(")[0].split("::")[-1][:24]
    by.setdefault(k,[0,0]); by[k][0]+=1; by[k][1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
print("mode $m total %.1f us"%tot, {k:(v[0],round(v[1],1)) for k,v in by.items()})
PY
done
