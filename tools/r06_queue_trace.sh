#!/bin/bash
# which hardware queue every stream of the step lands on (rocprofv3 kernel trace: Queue_Id, Stream_Id), default dealing and with unused streams in front of the second context's side streams
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06g
for cfg in "default X=1" "pad_0_2 DAV1D_HIP_STREAM_PAD=0,2" "pad_1 DAV1D_HIP_STREAM_PAD=1" "pad_2 DAV1D_HIP_STREAM_PAD=2"; do
  set -- $cfg
  env $2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06g/qtrace_$1 -- python bench.py --step-only --steps 6 --warmup 2 > gpurun_out/r06g/qtrace_$1.log 2>&1
  f=$(find gpurun_out/r06g/qtrace_$1 -name "*kernel_trace.csv" | head -1)
  cp "$f" gpurun_out/r06g/qtrace_$1.csv; rm -rf gpurun_out/r06g/qtrace_$1
done
ls -la gpurun_out/r06g/
