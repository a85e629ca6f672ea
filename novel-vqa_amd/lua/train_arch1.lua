--[[
train_arch1.lua -- 002_train_vqa_arch1/002_train_baseline.lua with the hot path on libnvqa.

Same command-line options (002_train_baseline.lua:16-50), same HDF5/JSON inputs (:84-127),
same log files and .t7 checkpoints (:389-420).  What changed: the nets, dupe_rnn clones,
JdJ and optim.rmsprop (:141-190, :269-335, :408) are one library context.  Needs LuaJIT +
torch + hdf5 + cjson (for the data files only); NOT executable in the build image, where
novel-vqa_amd/host/trainer.py is the executed twin.
]]--
require 'torch'
require 'hdf5'
local cjson = require 'cjson'
local nvqa = require 'nvqa_ffi'

cmd = torch.CmdLine()
cmd:option('-input_img_h5','data_img.h5','path to the h5file containing the image feature')
cmd:option('-input_ques_h5','data_prepro.h5','path to the h5file containing the preprocessed dataset')
cmd:option('-input_json','data_prepro.json','path to the json file containing additional info and vocab')
cmd:option('-learning_rate',3e-4,'learning rate for rmsprop')
cmd:option('-batch_size',500,'batch_size for each iterations')
cmd:option('-max_iters', 150000, 'max number of iterations to run for ')
cmd:option('-nhimage', 4096, 'number of image features')
cmd:option('-input_encoding_size', 200, 'the encoding size of each token in the vocabulary')
cmd:option('-rnn_size',512,'size of the rnn in number of hidden nodes in each layer')
cmd:option('-rnn_layer',2,'number of the rnn layer')
cmd:option('-common_embedding_size', 1024, 'size of the common embedding vector')
cmd:option('-num_output', 1000, 'number of output answers')
cmd:option('-img_norm', 1, 'normalize the image feature. 1 = normalize, 0 = not normalize')
cmd:option('-save_checkpoint_every', 150000, 'how often to save a model checkpoint?')
cmd:option('-checkpoint_path', 'model/', 'folder to save checkpoints')
cmd:option('-gpuid', 0, 'which MI355X to use')
cmd:option('-seed', 123, 'random number generator seed to use')
opt = cmd:parse(arg)
torch.manualSeed(opt.seed)
torch.setdefaulttensortype('torch.FloatTensor')
local decay_factor = 0.99997592083

-- dataset (unchanged from the reference, :84-121) -------------------------------------------
local f = io.open(opt.input_json, 'r'); local json_file = cjson.decode(f:read()); f:close()
local h5 = hdf5.open(opt.input_ques_h5, 'r')
local question = h5:read('/ques_train'):all():int()
local lengths  = h5:read('/ques_length_train'):all():int()
local img_list = h5:read('/img_pos_train'):all():int()
local answers  = h5:read('/answers'):all():int()
local question_val = h5:read('/ques_val'):all():int()                               -- :98-101
local lengths_val  = h5:read('/ques_length_val'):all():int()
local img_list_val = h5:read('/img_pos_val'):all():long()
local answers_val  = h5:read('/answers_val'):all():int()
h5:close()
h5 = hdf5.open(opt.input_img_h5, 'r')
local fv_im = h5:read('/images_train'):all():float()
local fv_im_val = h5:read('/images_val'):all():float()                             -- :109
h5:close()
local function right_align(seq, len)                       -- misc/RNNUtils.lua:54-61
  local v = seq:clone():fill(0); local N = seq:size(2)
  for i = 1, seq:size(1) do v[i][{{N-len[i]+1,N}}] = seq[i][{{1,len[i]}}] end
  return v
end
question = right_align(question, lengths):contiguous()
question_val = right_align(question_val, lengths_val):contiguous()                  -- :114
if opt.img_norm == 1 then                                                           -- :119-121 (the training features are normalised on the device)
  local nm = torch.sqrt(torch.sum(torch.cmul(fv_im_val, fv_im_val), 2))
  fv_im_val = torch.cdiv(fv_im_val, torch.repeatTensor(nm, 1, opt.nhimage)):float()
end
local vocabulary_size_q = 0
for _ in pairs(json_file['ix_to_word']) do vocabulary_size_q = vocabulary_size_q + 1 end

-- the model lives in HBM; so does the dataset (1.35 GB of fc7 features fit 200x over) ---------
local ctx = nvqa.create(1, opt, vocabulary_size_q, question:size(2), opt.gpuid)
nvqa.check(nvqa.lib.nvqa_init_params(ctx, opt.seed, -0.08, 0.08))                 -- :174-181
nvqa.check(nvqa.lib.nvqa_dataset_load(ctx, question:size(1), nvqa.iptr(question), nvqa.iptr(lengths),
           nvqa.iptr(img_list), nvqa.iptr(answers), fv_im:size(1), nvqa.fptr(fv_im), opt.img_norm))

local optimize = {learningRate = opt.learning_rate}
local loss = require('ffi').new('float[1]')
local running_avg
paths.mkdir(opt.checkpoint_path .. 'save')
local fileLogger = io.open(opt.checkpoint_path .. 'save/logFile.txt', 'w')
local fileLoggerVal = io.open(opt.checkpoint_path .. 'save/logFileVal.txt', 'w')

-- validate() (:337-381): evaluate-mode forward over the validation split, batch by batch (the last one short);
-- f_avg = mean of the per-batch mean cross-entropies, running_avg_val as the script keeps it
local running_avg_val
local function validate()
  local nval, B = question_val:size(1), opt.batch_size
  local f_avg, iters, f = 0, 0, require('ffi').new('float[1]')
  for i = 1, nval, B do
    local r = math.min(i + B - 1, nval)
    local q = question_val[{{i, r}}]:contiguous()
    local l = lengths_val[{{i, r}}]:contiguous()
    local im = fv_im_val:index(1, img_list_val[{{i, r}}]):contiguous()
    local y = answers_val[{{i, r}}]:contiguous()
    nvqa.check(nvqa.lib.nvqa_evaluate(ctx, r - i + 1, nvqa.iptr(q), nvqa.iptr(l), nvqa.fptr(im), nvqa.iptr(y),
               nil, 0, nil, nil, nil, f))
    running_avg_val = running_avg_val and (running_avg_val*0.95 + f[0]*0.05) or f[0]
    f_avg = f_avg + f[0]; iters = iters + 1
  end
  return f_avg / iters
end

local function save(path)                                                           -- :401-402
  local n = tonumber(nvqa.lib.nvqa_param_count(ctx))
  local x = torch.FloatTensor(n)
  nvqa.check(nvqa.lib.nvqa_get_params(ctx, nvqa.fptr(x)))
  local seg = require('ffi').new('size_t[3]'); nvqa.check(nvqa.lib.nvqa_segments(ctx, seg))
  local e, m = tonumber(seg[0]), tonumber(seg[1])
  torch.save(path, {encoder_w_q = x[{{1,e}}]:clone(), embedding_w_q = x[{{e+1,e+m}}]:clone(),
                    multimodal_w = x[{{e+m+1,n}}]:clone()})
end

for iter = 1, opt.max_iters do
  if iter % opt.save_checkpoint_every == 0 or iter == 1 then                        -- :395-403
    local loss_val = validate()
    fileLoggerVal:write('validation loss: ' .. loss_val .. ' validation loss avg: ' .. running_avg_val, ' on iter: ' .. iter .. '/' .. opt.max_iters .. '\n')
    print('validation loss: ' .. loss_val .. ' validation loss avg: ' .. running_avg_val .. ' on iter: ' .. iter .. '/' .. opt.max_iters .. '\n')
    save(string.format(opt.checkpoint_path..'save/lstm_save_iter%d.t7', iter))
  end
  -- dataset:next_batch(): qinds[i] = torch.random(nqs)  (:202-205); the gather runs on the device
  local qinds = torch.LongTensor(opt.batch_size):random(question:size(1)):add(-1)
  -- JdJ + optim.rmsprop  (:272-335, :408)
  nvqa.check(nvqa.lib.nvqa_step_indices(ctx, nvqa.lptr(qinds), nvqa.dropout(1, 0.5, opt.seed, iter), loss))
  nvqa.check(nvqa.lib.nvqa_rmsprop_update(ctx, optimize.learningRate, 0.99, 1e-8, 0, 10))
  running_avg = running_avg and (running_avg*0.95 + loss[0]*0.05) or loss[0]
  if iter % 100 == 0 then
    fileLogger:write('training loss: ' .. running_avg, 'on iter: ' .. iter .. '/' .. opt.max_iters .. '\n')
    print('training loss: ' .. running_avg, 'on iter: ' .. iter .. '/' .. opt.max_iters)
  end
  optimize.learningRate = optimize.learningRate * decay_factor                      -- :410
end
fileLogger:close()
fileLoggerVal:close()
save(opt.checkpoint_path .. 'lstm.t7')
