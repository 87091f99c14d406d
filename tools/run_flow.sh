cd $GRAFT_REPO_ROOT
timeout 100 python -m pytest tests/test_postchain.py -m gpu -x -q -k "one-launch" 2>&1 | tail -2
for g in 256 512; do
echo "== groups $g"
DAV1D_HIP_FLOW_GROUPS=$g DAV1D_HIP_TRACE_INTRA=1 timeout 60 python -m dav1d_amd.e2e --key-frame 1 --frames 3 --tile-cols 8 2>&1 | grep "intra flow" | tail -1
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for g in 256 1024; do
  rm -rf gpurun_out/fl_1
  DAV1D_HIP_FLOW_GROUPS=$g timeout 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fl_1 -- python tools/intra_probe.py --reps 2 --flow 1 > gpurun_out/fl_1.log 2>&1
  f=$(ls gpurun_out/fl_1/*/*kernel_trace.csv | head -1)
  python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$f")) if "rocclr" not in r["Kernel_Name"]]
for r in rows: print("bench intra pass, groups $g: flow kernel %.1f us" % ((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
done
