cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 bash novel-vqa_amd/tools/profile_r04.sh > gpurun_out/prof_r04.log 2>&1 && echo headline ok &&
timeout -k 10 400 bash novel-vqa_amd/tools/profile_r04.sh bf16_arch2 "--arch 2 --bf16" > gpurun_out/prof_r04_bf16.log 2>&1 && echo bf16 ok
tail -5 gpurun_out/prof_r04.log
