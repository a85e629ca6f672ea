--[[
compute_lf_answers.lua -- 002_train_vqa_arch1/003_compute_lf_answers.lua with the forward passes on libnvqa.

Late fusion, first half: the answer scores of TWO trained arch1 models -- one on VGG fc7 features (I = 4096), one on
Inception-v3 features (I = 2048; :388) -- over the train, val and test splits, written to one HDF5 file
(/VGGOut, /InceptionOut, ...Val, ...Test; :476-482) that eval_arch1_lf.lua combines.  Same options (:25-48).  The
evaluate-mode forward of a batch (:201-227) is nvqa_evaluate; the models are the reference's checkpoint tables (:361-369).
Not executable in the build image (no LuaJIT); VQATrainer.predict + trainer.late_fusion are the executed twins.
]]--
require 'torch'
require 'hdf5'
local cjson = require 'cjson'
local ffi = require 'ffi'
local nvqa = require 'nvqa_ffi'

cmd = torch.CmdLine()
cmd:option('-vgg_img_h5','data_img.h5','path to the h5file containing the vgg image feature')
cmd:option('-inception_img_h5','data_img.h5','path to the h5file containing the inception image feature')
cmd:option('-input_ques_h5','data_prepro.h5','path to the h5file containing the preprocessed dataset')
cmd:option('-input_json','data_prepro.json','path to the json file containing additional info and vocab')
cmd:option('-batch_size',500,'batch_size for each iterations')
cmd:option('-input_encoding_size_vgg', 512, 'the encoding size of each token in the vocabulary')
cmd:option('-rnn_size_vgg',512,'size of the rnn in number of hidden nodes in each layer')
cmd:option('-input_encoding_size_incep', 512, 'the encoding size of each token in the vocabulary')
cmd:option('-rnn_size_incep',512,'size of the rnn in number of hidden nodes in each layer')
cmd:option('-rnn_layer',1,'number of the rnn layer')
cmd:option('-common_embedding_size', 1024, 'size of the common embedding vector')
cmd:option('-num_output', 1000, 'number of output answers')
cmd:option('-vgg_norm', 1, 'normalize the vgg image feature. 1 = normalize, 0 = not normalize')
cmd:option('-inception_norm', 1, 'normalize the inception image feature. 1 = normalize, 0 = not normalize')
cmd:option('-vgg_model_path', 'model/model_default_params/lstm.t7', 'path to VGG model')
cmd:option('-inception_model_path', 'model/model_inception_default_params/lstm.t7', 'path to Inception model')
cmd:option('-gpuid', 0, 'which MI355X to use')
cmd:option('-out_path', 'outputVectors.h5', 'output file path')
opt = cmd:parse(arg)
torch.setdefaulttensortype('torch.FloatTensor')

local f = io.open(opt.input_json, 'r'); local json_file = cjson.decode(f:read()); f:close()
local vocabulary_size_q = 0
for _ in pairs(json_file['ix_to_word']) do vocabulary_size_q = vocabulary_size_q + 1 end
local function right_align(seq, len)                       -- misc/RNNUtils.lua:54-61
  local v = seq:clone():fill(0); local N = seq:size(2)
  for i = 1, seq:size(1) do v[i][{{N-len[i]+1,N}}] = seq[i][{{1,len[i]}}] end
  return v
end
local h5 = hdf5.open(opt.input_ques_h5, 'r')
local splits = {}
for _, s in ipairs({{'train', '/ques_train', '/ques_length_train', '/img_pos_train'},
                    {'val', '/ques_val', '/ques_length_val', '/img_pos_val'},
                    {'test', '/ques_test', '/ques_length_test', '/img_pos_test'}}) do
  local q, l = h5:read(s[2]):all():int(), h5:read(s[3]):all():int()
  splits[s[1]] = {question = right_align(q, l):contiguous(), lengths = l, img_list = h5:read(s[4]):all():long()}
end
h5:close()

-- scores of one model over the three splits (:201-283 x 3)
local function predict_all(img_h5, norm, nhimage, E, R, model_path)
  local o = {batch_size = opt.batch_size, input_encoding_size = E, rnn_size = R, rnn_layer = opt.rnn_layer,
             nhimage = nhimage, common_embedding_size = opt.common_embedding_size, num_output = opt.num_output}
  local ctx = nvqa.create(1, o, vocabulary_size_q, splits.train.question:size(2), opt.gpuid)
  local model_param = torch.load(model_path)                                        -- :361-369
  local x = torch.cat({model_param['encoder_w_q']:float(), model_param['embedding_w_q']:float(),
                       model_param['multimodal_w']:float()}, 1):contiguous()
  assert(x:nElement() == tonumber(nvqa.lib.nvqa_param_count(ctx)), 'checkpoint does not match the model options')
  nvqa.check(nvqa.lib.nvqa_set_params(ctx, nvqa.fptr(x)))
  local fh = hdf5.open(img_h5, 'r')
  local out = {}
  for _, s in ipairs({{'train', '/images_train'}, {'val', '/images_val'}, {'test', '/images_test'}}) do
    local fv = fh:read(s[2]):all():float()
    if norm == 1 then                                                               -- :309-316
      local nm = torch.sqrt(torch.sum(torch.cmul(fv, fv), 2))
      fv = torch.cdiv(fv, torch.repeatTensor(nm, 1, nhimage)):float()
    end
    local d = splits[s[1]]
    local n, B = d.question:size(1), opt.batch_size
    local scores = torch.FloatTensor(n, opt.num_output)
    for i = 1, n, B do
      local r = math.min(i + B - 1, n)
      local q = d.question[{{i, r}}]:contiguous()
      local l = d.lengths[{{i, r}}]:contiguous()
      local im = fv:index(1, d.img_list[{{i, r}}]):contiguous()
      nvqa.check(nvqa.lib.nvqa_evaluate(ctx, r - i + 1, nvqa.iptr(q), nvqa.iptr(l), nvqa.fptr(im), nil, nil, 0,
                 nvqa.fptr(scores[{{i, r}}]), nil, nil, nil))
    end
    out[s[1]] = scores
  end
  fh:close()
  return out
end

local vgg = predict_all(opt.vgg_img_h5, opt.vgg_norm, 4096, opt.input_encoding_size_vgg, opt.rnn_size_vgg, opt.vgg_model_path)
local inc = predict_all(opt.inception_img_h5, opt.inception_norm, 2048, opt.input_encoding_size_incep, opt.rnn_size_incep,
                        opt.inception_model_path)
local outputFile = hdf5.open(opt.out_path, 'w')                                     -- :476-482
outputFile:write('/VGGOut', vgg.train)
outputFile:write('/InceptionOut', inc.train)
outputFile:write('/VGGOutVal', vgg.val)
outputFile:write('/InceptionOutVal', inc.val)
outputFile:write('/VGGOutTest', vgg.test)
outputFile:write('/InceptionOutTest', inc.test)
outputFile:close()
