--[[
train_arch1_ae.lua -- the AE-initialised arch1 trainers with the hot path on libnvqa:
    002_train_vqa_arch1/003_train_ae_based.lua       -variant base  (VGG fc7, I = 4096)
    002_train_vqa_arch1/003_train_ae_based_inc.lua   -variant inc   (Inception-v3 pool3, I = 2048; :74)
    002_train_vqa_arch1/003_train_ae_based_ef.lua    -variant ef    (early fusion I = 2048 + 4096, each block L2-normalised
                                                                     on its own; :74,116-124)
    002_train_vqa_arch1/003_train_ae_based_wp.lua    -variant wp    (netdef.AskipB fusion, fusion projections copied from
                                                                     the weak-paired AE, -lr_scale; :30,151-160,344)
Same options as those scripts (003_train_ae_based.lua:16-50), same inputs, logs and checkpoints as train_arch1.lua.  What
differs from the baseline trainer is the initialisation (:65,175-186): the embedding comes from the auto-encoder's lookup
table (its last column, the AE's START token, dropped), the embedding bias is zero, encoder_w_q is the AE encoder, and only
multimodal_w is uniform(-0.08, 0.08).  novel-vqa_amd/host/trainer.py: VQATrainer.init_from_autoencoder is the executed twin
(tests/test_ae_init.py).  Not executable in the build image (no LuaJIT).

The order of the LSTM tensors inside savedParams['encoder'] is nngraph's (the AE encoder is a gModule); libnvqa's is
include/nvqa_layout.h (per layer W_i2h, b_i2h, W_h2h, b_h2h).  -encoder_perm names a torch file holding the LongTensor
ours = foreign:index(1, perm) once that order has been read off a Torch7 installation (host/t7.py: encoder_permutation).
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
cmd:option('-model_path', '', 'loading model parameters (table of 001_train_autoencoder/002_convert_text_model_arch1.lua)')
cmd:option('-variant', 'base', 'base | inc | ef | wp (which 003_train_ae_based*.lua this run stands for)')
cmd:option('-lr_scale', 1, 'learning rate scale for the encoder and embedding layer (wp)')
cmd:option('-encoder_perm', '', 'torch file with the LongTensor that maps the AE encoder\'s tensor order onto libnvqa\'s (optional)')
cmd:option('-learning_rate',1e-4,'learning rate for rmsprop')
cmd:option('-batch_size',500,'batch_size for each iterations')
cmd:option('-max_iters', 25000, 'max number of iterations to run for ')
cmd:option('-input_encoding_size', 512, 'the encoding size of each token in the vocabulary')
cmd:option('-rnn_size',512,'size of the rnn in number of hidden nodes in each layer')
cmd:option('-rnn_layer',1,'number of the rnn layer')
cmd:option('-common_embedding_size', 1024, 'size of the common embedding vector')
cmd:option('-num_output', 1000, 'number of output answers')
cmd:option('-img_norm', 1, 'normalize the image feature. 1 = normalize, 0 = not normalize')
cmd:option('-save_checkpoint_every', 5000, 'how often to save a model checkpoint?')
cmd:option('-checkpoint_path', 'model/', 'folder to save checkpoints')
cmd:option('-gpuid', 0, 'which MI355X to use')
cmd:option('-seed', 123, 'random number generator seed to use')
opt = cmd:parse(arg)
torch.manualSeed(opt.seed)
torch.setdefaulttensortype('torch.FloatTensor')
local decay_factor = 0.99997592083
-- the scripts hard-code the feature width (:74): 4096 (base, wp), 2048 (inc), 2048 + 4096 (ef)
opt.nhimage = ({base = 4096, wp = 4096, inc = 2048, ef = 2048 + 4096})[opt.variant]
assert(opt.nhimage, 'unknown -variant ' .. opt.variant)
-- two-block normalisation of the early-fusion features (003_train_ae_based_ef.lua:116-124): columns 1..2048 and 2049..6144
local norm_split = (opt.variant == 'ef') and 2048 or 1

local savedParams = torch.load(opt.model_path)                                      -- :65
print(savedParams)

-- dataset (as train_arch1.lua; the reference block is :84-124) ---------------------------------------------------------------
local f = io.open(opt.input_json, 'r'); local json_file = cjson.decode(f:read()); f:close()
local h5 = hdf5.open(opt.input_ques_h5, 'r')
local question = h5:read('/ques_train'):all():int()
local lengths  = h5:read('/ques_length_train'):all():int()
local img_list = h5:read('/img_pos_train'):all():int()
local answers  = h5:read('/answers'):all():int()
local question_val = h5:read('/ques_val'):all():int()
local lengths_val  = h5:read('/ques_length_val'):all():int()
local img_list_val = h5:read('/img_pos_val'):all():long()
local answers_val  = h5:read('/answers_val'):all():int()
h5:close()
h5 = hdf5.open(opt.input_img_h5, 'r')
local fv_im = h5:read('/images_train'):all():float()
local fv_im_val = h5:read('/images_val'):all():float()
h5:close()
local function right_align(seq, len)                       -- misc/RNNUtils.lua:54-61
  local v = seq:clone():fill(0); local N = seq:size(2)
  for i = 1, seq:size(1) do v[i][{{N-len[i]+1,N}}] = seq[i][{{1,len[i]}}] end
  return v
end
question = right_align(question, lengths):contiguous()
question_val = right_align(question_val, lengths_val):contiguous()
local function l2_block(x, c0, c1)                         -- one block of columns, in place (:119-122; ef :116-124)
  local blk = x[{{}, {c0, c1}}]
  local nm = torch.sqrt(torch.sum(torch.cmul(blk, blk), 2))
  blk:cdiv(torch.repeatTensor(nm, 1, c1 - c0 + 1))
end
if opt.img_norm == 1 then                                  -- validation features on the host; training features on the device
  if norm_split > 1 then l2_block(fv_im_val, 1, norm_split); l2_block(fv_im_val, norm_split + 1, opt.nhimage)
  else l2_block(fv_im_val, 1, opt.nhimage) end
end
local vocabulary_size_q = 0
for _ in pairs(json_file['ix_to_word']) do vocabulary_size_q = vocabulary_size_q + 1 end

-- model ------------------------------------------------------------------------------------------------------------------------
local ctx = nvqa.create(1, opt, vocabulary_size_q, question:size(2), opt.gpuid)
if opt.variant == 'wp' then nvqa.check(nvqa.lib.nvqa_set_fusion(ctx, 1)) end      -- netdef.AskipB (003_train_ae_based_wp.lua:151)
nvqa.check(nvqa.lib.nvqa_init_params(ctx, opt.seed, -0.08, 0.08))                  -- multimodal_w:uniform(-0.08, 0.08) (:186)
do                                                                                  -- :175-186
  local n = tonumber(nvqa.lib.nvqa_param_count(ctx))
  local seg = ffi.new('size_t[3]'); nvqa.check(nvqa.lib.nvqa_segments(ctx, seg))
  local e, m = tonumber(seg[0]), tonumber(seg[1])
  local x = torch.FloatTensor(n)
  nvqa.check(nvqa.lib.nvqa_get_params(ctx, nvqa.fptr(x)))
  local E, V = opt.input_encoding_size, vocabulary_size_q
  local lookup = savedParams['lookup']:float()                                     -- [E x (V+1)]
  assert(lookup:size(1) == E and lookup:size(2) == V + 1, 'lookup does not match -input_encoding_size / the vocabulary')
  local enc = savedParams['encoder']:float()
  assert(enc:nElement() == e, 'encoder does not match -rnn_size / -rnn_layer / -input_encoding_size')
  -- the reference's converter (002_convert_text_model_arch1.lua:27-39) writes nngraph's order and no marker: without a
  -- permutation only a table this package wrote (layout = 'nvqa') can be copied as it is (host/trainer.py refuses likewise)
  assert(opt.encoder_perm ~= '' or savedParams.layout == 'nvqa',
         "the auto-encoder table has no layout = 'nvqa' marker: the order of the LSTM tensors inside 'encoder' is nngraph's; pass -encoder_perm")
  if opt.encoder_perm ~= '' then enc = enc:index(1, torch.load(opt.encoder_perm):long()) end
  x[{{1, e}}]:copy(enc)                                                             -- encoder_w_q:copy(savedParams['encoder'])
  x[{{e + 1, e + E * V}}]:copy(lookup[{{}, {1, lookup:size(2) - 1}}]:contiguous():view(-1))   -- embedding weight (:177)
  x[{{e + E * V + 1, e + m}}]:fill(0)                                               -- embedding bias (:178)
  if opt.variant == 'wp' then                                                       -- multimodal_w:copy(savedParams['multimodal']) (_wp :155-156)
    local mm = savedParams['multimodal']:float()
    x[{{e + m + 1, e + m + mm:nElement()}}]:copy(mm)                                -- W_q, b_q, W_v, b_v; the classifier stays uniform (:157-160)
  end
  nvqa.check(nvqa.lib.nvqa_set_params(ctx, nvqa.fptr(x)))
end
if opt.variant == 'wp' then                                                         -- gradients of encoder / embedding x lr_scale (_wp :344)
  local sc = ffi.new('float[3]', {opt.lr_scale, opt.lr_scale, 1})
  nvqa.check(nvqa.lib.nvqa_set_grad_scales(ctx, sc))
end
-- l2_normalize: 0 no, 1 whole rows, n > 1 two blocks split at column n (include/nvqa.h)
nvqa.check(nvqa.lib.nvqa_dataset_load(ctx, question:size(1), nvqa.iptr(question), nvqa.iptr(lengths),
           nvqa.iptr(img_list), nvqa.iptr(answers), fv_im:size(1), nvqa.fptr(fv_im),
           opt.img_norm == 1 and norm_split or 0))

local optimize = {learningRate = opt.learning_rate}
local loss = ffi.new('float[1]')
local running_avg, running_avg_val
paths.mkdir(opt.checkpoint_path .. 'save')
local fileLogger = io.open(opt.checkpoint_path .. 'save/logFile.txt', 'w')
local fileLoggerVal = io.open(opt.checkpoint_path .. 'save/logFileVal.txt', 'w')

local function validate()                                                           -- :337-381
  local nval, B = question_val:size(1), opt.batch_size
  local f_avg, iters, f = 0, 0, ffi.new('float[1]')
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
  local seg = ffi.new('size_t[3]'); nvqa.check(nvqa.lib.nvqa_segments(ctx, seg))
  local e, m = tonumber(seg[0]), tonumber(seg[1])
  torch.save(path, {encoder_w_q = x[{{1,e}}]:clone(), embedding_w_q = x[{{e+1,e+m}}]:clone(),
                    multimodal_w = x[{{e+m+1,n}}]:clone()})
end

for iter = 1, opt.max_iters do
  if iter % opt.save_checkpoint_every == 0 or iter == 1 then
    local loss_val = validate()
    fileLoggerVal:write('validation loss: ' .. loss_val .. ' validation loss avg: ' .. running_avg_val, ' on iter: ' .. iter .. '/' .. opt.max_iters .. '\n')
    print('validation loss: ' .. loss_val .. ' validation loss avg: ' .. running_avg_val .. ' on iter: ' .. iter .. '/' .. opt.max_iters .. '\n')
    save(string.format(opt.checkpoint_path..'save/lstm_save_iter%d.t7', iter))
  end
  local qinds = torch.LongTensor(opt.batch_size):random(question:size(1)):add(-1)   -- dataset:next_batch() (:202-205)
  nvqa.check(nvqa.lib.nvqa_step_indices(ctx, nvqa.lptr(qinds), nvqa.dropout(1, 0.5, opt.seed, iter), loss))
  nvqa.check(nvqa.lib.nvqa_rmsprop_update(ctx, optimize.learningRate, 0.99, 1e-8, 0, 10))   -- clamp(-10, 10) + optim.rmsprop (:329,408)
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
