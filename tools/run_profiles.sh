#!/bin/bash
# round-2 profile bundle (run through gpurun): calibration, kernel stats of the default command, PMC passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02/calib -- tools/calib/fetch_calib > gpurun_out/r02/calib.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/default_cmd -- python bench.py --no-cpu --no-e2e --no-c1 > gpurun_out/r02/default_cmd.log 2>&1
timeout 900 bash tools/pmc_profile.sh r02_pmc > gpurun_out/r02/pmc.log 2>&1
tail -5 gpurun_out/r02/calib.log
ls gpurun_out/r02 gpurun_out/r02_pmc | head -40
