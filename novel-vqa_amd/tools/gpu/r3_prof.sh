cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 bash novel-vqa_amd/tools/profile_r03.sh > gpurun_out/prof_r03.log 2>&1 && echo headline ok &&
timeout -k 10 400 bash novel-vqa_amd/tools/profile_r03.sh bf16_arch2 "--arch 2 --bf16" > gpurun_out/prof_r03_bf16.log 2>&1 && echo bf16 ok &&
timeout -k 10 400 bash novel-vqa_amd/tools/profile_vgg_r03.sh f32 64 > gpurun_out/prof_vgg_r03_f32.log 2>&1 && echo vgg ok &&
timeout -k 10 400 bash novel-vqa_amd/tools/profile_vgg_r03.sh bf16 64 bf16 > gpurun_out/prof_vgg_r03_bf16.log 2>&1 && echo vggbf16 ok
tail -5 gpurun_out/prof_r03.log
