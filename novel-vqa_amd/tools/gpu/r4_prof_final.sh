cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 bash novel-vqa_amd/tools/profile_r04.sh > gpurun_out/prof_r04.log 2>&1 && echo headline ok &&
timeout -k 10 500 bash novel-vqa_amd/tools/profile_r04.sh ragged "--ragged" > gpurun_out/prof_r04_ragged.log 2>&1 && echo ragged ok
tail -5 gpurun_out/prof_r04.log; tail -3 gpurun_out/prof_r04_ragged.log
