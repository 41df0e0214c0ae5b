#!/bin/bash
# (round 3) copy of the script each gpurun call ran: see profiles/r03_job_C5_50M.json
# one invocation of the 50 M-row C5 job; state travels through gpurun_out/c5/ (copied into gpurun_tmp/ between calls)
mkdir -p gpurun_out/c5
cp gpurun_tmp/c5st.rank0.npz gpurun_out/c5/st.rank0.npz 2>/dev/null
timeout -k 10 1150 python -u -m clip_amd.job --config C5 --rows 50000000 --topk 250000 --progress \
  --state gpurun_out/c5/st --max-seconds 880 --out gpurun_out/c5/ttp_top250000.csv \
  > gpurun_out/c5/job_$(date +%s).json 2> gpurun_out/c5/progress.log
rc=$?
tail -3 gpurun_out/c5/progress.log; cat gpurun_out/c5/job_*.json | tail -2; ls -la gpurun_out/c5/ | head
exit $rc
