cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
(timeout 300 python tools/run_event_check.py small 2>&1 | grep -v amdgpu.ids | tail -4)
python tools/run_match_ab.py 11000000 6 8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3/ab_c4.log
python tools/run_match_ab.py 1650000 6 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3/ab_c3.log
LAMBDAS=0.5 AE_CE_PROF=1 timeout 1500 python tools/run_match_check.py fidelity blobs6 60000 40 gpurun_out/r3/fid_blobs6.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3/fid_blobs6.log
