cd $GRAFT_REPO_ROOT
DAV1D_HIP_TRACE_INTRA=1 timeout 60 python tests/e2e.py --key-frame 1 --frames 3 --tile-cols 16 --tile-rows 8 --threads 32 2>&1 | grep "intra flow" | tail -1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/fl_1
timeout 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fl_1 -- python tools/intra_probe.py --reps 2 --flow 1 > gpurun_out/fl_1.log 2>&1
f=$(ls gpurun_out/fl_1/*/*kernel_trace.csv | head -1)
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$f")) if "rocclr" not in r["Kernel_Name"]]
for r in rows: print("bench intra pass (46 steps) as one launch: %.1f us" % ((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
