--[[
eval_arch1_lf.lua -- 002_train_vqa_arch1/004_eval_model_lf.lua: late fusion, second half.

scores = weight_vgg * VGGOutTest + weight_inception * InceptionOutTest (:109-132), then the open-ended argmax over all
answers and the multiple-choice argmax over the non-zero candidates of MC_ans_test (:144-186), written in the reference's
result format.  The reference does this on the host from the file compute_lf_answers.lua wrote; no network runs here, so
no library call is needed -- the file is shipped so that the whole late-fusion route has a twin (options :21-32).
trainer.late_fusion / trainer.multiple_choice_argmax / trainer.results_json are the executed twins
(tests/test_gpu_variants.py).  Not executable in the build image (no LuaJIT).
]]--
require 'torch'
require 'hdf5'
local cjson = require 'cjson'

cmd = torch.CmdLine()
cmd:option('-input_ques_h5','data_prepro.h5','path to the h5file containing the preprocessed dataset')
cmd:option('-input_json','data_prepro.json','path to the json file containing additional info and vocab')
cmd:option('-input_features_h5','outputVectors.h5', 'path to the h5file containing computed output vectors')
cmd:option('-out_path', 'result/', 'path to save output json file')
cmd:option('-batch_size',500,'batch_size for each iterations')
cmd:option('-num_output', 1000, 'number of output answers')
cmd:option('-weight_vgg', 0.5, 'scaling for vgg in ensemble')
cmd:option('-weight_inception', 0.5, 'scaling for inception in ensemble')
opt = cmd:parse(arg)

local f = io.open(opt.input_json, 'r'); local json_file = cjson.decode(f:read()); f:close()
local h5 = hdf5.open(opt.input_ques_h5, 'r')
local ques_id = h5:read('/question_id_test'):all()
local MC_ans  = h5:read('/MC_ans_test'):all()
h5:close()
h5 = hdf5.open(opt.input_features_h5, 'r')
local vgg_out = h5:read('/VGGOutTest'):all():double()
local inception_out = h5:read('/InceptionOutTest'):all():double()
h5:close()

local scores = torch.mul(vgg_out, opt.weight_vgg):add(opt.weight_inception, inception_out)   -- :112-113,131
local nqs = scores:size(1)
local _, pred = torch.max(scores, 2)                                                -- :149

local function saveJson(fname, t) local fo = io.open(fname, 'w'); fo:write(cjson.encode(t)); fo:close() end
local response = {}
for i = 1, nqs do
  table.insert(response, {question_id = ques_id[i], answer = json_file['ix_to_ans'][tostring(pred[{i, 1}])]})
end
paths.mkdir(opt.out_path)
saveJson(opt.out_path .. 'OpenEnded_mscoco_lstm_results.json', response)

local mc_response = {}                                                              -- :170-184
for i = 1, nqs do
  local mc_prob, tmp_idx = {}, {}
  local mc_idx = MC_ans[i]
  for j = 1, mc_idx:size(1) do
    if mc_idx[j] ~= 0 then
      table.insert(mc_prob, scores[{i, mc_idx[j]}])
      table.insert(tmp_idx, mc_idx[j])
    end
  end
  local _, best = torch.max(torch.Tensor(mc_prob), 1)
  table.insert(mc_response, {question_id = ques_id[i], answer = json_file['ix_to_ans'][tostring(tmp_idx[best[1]])]})
end
saveJson(opt.out_path .. 'MultipleChoice_mscoco_lstm_results.json', mc_response)
