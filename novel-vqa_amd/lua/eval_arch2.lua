--[[
eval_arch2.lua -- 003_train_vqa_arch2/004_eval_model_baseline.lua with the forward pass on libnvqa.

Same options (004_eval_model_baseline.lua:24-40), same inputs (test split of data_prepro.h5 / data_img.h5), the
checkpoint table {cnn_w, encoder_w_q, multimodal_w} written by train_arch2.lua / 002_train_baseline.lua (:400-402)
and the same two result files (:268-301).  evaluateModel() (:222-257) -- cnn_projection, nn.Encoder, classifier in
evaluate mode, batch by batch -- is nvqa_evaluate; the questions stay LEFT-aligned with 0 = null (the script's
fv_q is the transpose of the rows handed over here).  A fresh process never ran a backward, so the aliased initial
state of misc/Encoder_lstm.lua:238-239 is still zero here: no -ref_quirks option is needed.  Not executable in the
build image (no LuaJIT); VQATrainer.predict_mc / trainer.results_json are the executed twins.
]]--
require 'torch'
require 'hdf5'
local cjson = require 'cjson'
local ffi = require 'ffi'
local nvqa = require 'nvqa_ffi'

cmd = torch.CmdLine()
cmd:option('-input_img_h5','data_img.h5','path to the h5file containing the image feature')
cmd:option('-input_ques_h5','data_prepro.h5','path to the h5file containing the preprocessed dataset')
cmd:option('-input_json','data_prepro.json','path to the json file containing additional info and vocab')
cmd:option('-model_path', 'model/lstm.t7', 'path to a model checkpoint to initialize model weights from')
cmd:option('-out_path', 'result/', 'path to save output json file')
cmd:option('-batch_size',500,'batch_size for each iterations')
cmd:option('-input_encoding_size', 512, 'the encoding size of each token in the vocabulary')
cmd:option('-rnn_size',512,'size of the rnn in number of hidden nodes in each layer')
cmd:option('-rnn_layer',1,'number of the rnn layer')
cmd:option('-common_embedding_size', 1024, 'size of the common embedding vector')
cmd:option('-num_output', 1000, 'number of output answers')
cmd:option('-img_norm', 1, 'normalize the image feature. 1 = normalize, 0 = not normalize')
cmd:option('-nhimage', 4096, 'Image vector size')
cmd:option('-gpuid', 0, 'which MI355X to use')
opt = cmd:parse(arg)
opt.num_layers = opt.rnn_layer            -- nvqa.create reads the training script's name for arch2
torch.setdefaulttensortype('torch.FloatTensor')

local f = io.open(opt.input_json, 'r'); local json_file = cjson.decode(f:read()); f:close()
local h5 = hdf5.open(opt.input_ques_h5, 'r')
local question = h5:read('/ques_test'):all():int():contiguous()
local img_list = h5:read('/img_pos_test'):all():long()
local ques_id  = h5:read('/question_id_test'):all()
local MC_ans   = h5:read('/MC_ans_test'):all():int():contiguous()
h5:close()
h5 = hdf5.open(opt.input_img_h5, 'r'); local fv_im = h5:read('/images_test'):all():float(); h5:close()
if opt.img_norm == 1 then
  local nm = torch.sqrt(torch.sum(torch.cmul(fv_im, fv_im), 2))
  fv_im = torch.cdiv(fv_im, torch.repeatTensor(nm, 1, opt.nhimage)):float()
end
local vocabulary_size_q = 0
for _ in pairs(json_file['ix_to_word']) do vocabulary_size_q = vocabulary_size_q + 1 end

local ctx = nvqa.create(2, opt, vocabulary_size_q, question:size(2), opt.gpuid)
local model_param = torch.load(opt.model_path)              -- {cnn_w, encoder_w_q, multimodal_w}
local x = torch.cat({model_param['cnn_w']:float(), model_param['encoder_w_q']:float(),
                     model_param['multimodal_w']:float()}, 1):contiguous()
assert(x:nElement() == tonumber(nvqa.lib.nvqa_param_count(ctx)), 'checkpoint does not match the model options')
nvqa.check(nvqa.lib.nvqa_set_params(ctx, nvqa.fptr(x)))

local nqs, noutput, B = question:size(1), opt.num_output, opt.batch_size
local pred = torch.IntTensor(nqs)
local mc_pred = torch.IntTensor(nqs)
for i = 1, nqs, B do                                        -- evaluateModel(), :222-257
  local r = math.min(i + B - 1, nqs)
  local q = question[{{i, r}}]:contiguous()
  local im = fv_im:index(1, img_list[{{i, r}}]):contiguous()
  local mc = MC_ans[{{i, r}}]:contiguous()
  nvqa.check(nvqa.lib.nvqa_evaluate(ctx, r - i + 1, nvqa.iptr(q), nil, nvqa.fptr(im), nil, nvqa.iptr(mc), MC_ans:size(2),
             nil, ffi.cast('int32_t*', pred[{{i, r}}]:data()), ffi.cast('int32_t*', mc_pred[{{i, r}}]:data()), nil))
end

local function saveJson(fname, t) local fo = io.open(fname, 'w'); fo:write(cjson.encode(t)); fo:close() end
local response, mc_response = {}, {}
for i = 1, nqs do                                           -- :268-301
  table.insert(response, {question_id = ques_id[i], answer = json_file['ix_to_ans'][tostring(pred[i])]})
  table.insert(mc_response, {question_id = ques_id[i], answer = json_file['ix_to_ans'][tostring(mc_pred[i])]})
end
paths.mkdir(opt.out_path)
saveJson(opt.out_path .. 'OpenEnded_mscoco_val2014_lstm_novel_new_2_results.json', response)
saveJson(opt.out_path .. 'MultipleChoice_mscoco_val2014_lstm_novel_new_2_results.json', mc_response)
