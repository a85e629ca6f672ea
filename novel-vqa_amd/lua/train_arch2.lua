--[[
train_arch2.lua -- 003_train_vqa_arch2/002_train_baseline.lua with the hot path on libnvqa.

Same command-line options (002_train_baseline.lua:16-52), same HDF5/JSON inputs (:86-130), same
log file and .t7 checkpoint table {cnn_w, encoder_w_q, multimodal_w} (:189,198,400-420).  The
cnn_projection Linear, nn.Encoder (image-as-first-token LSTM + shared LookupTable), the classifier,
JdJ and optim.rmsprop with weightDecay 1e-4 (:162-198, :277-333, :408) are one library context.
arch2 keeps the stored LEFT-aligned questions (the script never calls right_align; its fv_q is the
transpose of the rows handed over here, :216).  Not executable in the build image (no LuaJIT);
novel-vqa_amd/host/trainer.py is the executed twin.
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
cmd:option('-drop_prob_ae', 0.5, 'dropout value')
cmd:option('-learning_rate',3e-4,'learning rate for rmsprop')
cmd:option('-batch_size',500,'batch_size for each iterations')
cmd:option('-max_iters', 150000, 'max number of iterations to run for ')
cmd:option('-input_encoding_size', 512, 'size of word representation')
cmd:option('-rnn_size', 512, 'size of the rnn hidden layer')
cmd:option('-num_layers', 1, 'number of hidden layers in RNN')
cmd:option('-common_embedding_size', 1024, 'size of the common embedding vector')
cmd:option('-num_output', 1000, 'number of output answers')
cmd:option('-img_norm', 1, 'normalize the image feature. 1 = normalize, 0 = not normalize')
cmd:option('-nhimage', 4096, 'image feature dimensions')
cmd:option('-ref_quirks', 3, 'reference artefacts to reproduce: 1 = aliased top-layer h0 (misc/Encoder_lstm.lua:238-239), 2 = lookup table '
           .. 'without gradient (misc/Encoder_lstm.lua:49-58); 3 = both = what 002_train_baseline.lua really trains; 0 = the model as designed')
cmd:option('-bf16', 0, '1 = dense products on the bf16 matrix cores (BASELINE config "arch2 ... bf16")')
cmd:option('-save_checkpoint_every', 25000, 'how often to save a model checkpoint?')
cmd:option('-checkpoint_path', 'models_vqa/', 'folder to save checkpoints')
cmd:option('-gpuid', 0, 'which MI355X to use')
cmd:option('-seed', 123, 'random number generator seed to use')
opt = cmd:parse(arg)
torch.manualSeed(opt.seed)
torch.setdefaulttensortype('torch.FloatTensor')
local decay_factor = 0.99997592083                                                  -- :80

-- dataset (:86-121); questions stay left-aligned, 0 = null ---------------------------------------
local f = io.open(opt.input_json, 'r'); local json_file = cjson.decode(f:read()); f:close()
local h5 = hdf5.open(opt.input_ques_h5, 'r')
local question = h5:read('/ques_train'):all():int():contiguous()
local lengths  = h5:read('/ques_length_train'):all():int()
local img_list = h5:read('/img_pos_train'):all():int()
local answers  = h5:read('/answers'):all():int()
local question_val = h5:read('/ques_val'):all():int():contiguous()                  -- :100-103
local img_list_val = h5:read('/img_pos_val'):all():long()
local answers_val  = h5:read('/answers_val'):all():int()
h5:close()
h5 = hdf5.open(opt.input_img_h5, 'r')
local fv_im = h5:read('/images_train'):all():float()
local fv_im_val = h5:read('/images_val'):all():float()
h5:close()
if opt.img_norm == 1 then                                                           -- :119-123 (training features: on the device)
  local nm = torch.sqrt(torch.sum(torch.cmul(fv_im_val, fv_im_val), 2))
  fv_im_val = torch.cdiv(fv_im_val, torch.repeatTensor(nm, 1, opt.nhimage)):float()
end
local vocabulary_size_q = 0
for _ in pairs(json_file['ix_to_word']) do vocabulary_size_q = vocabulary_size_q + 1 end

local ctx = nvqa.create(2, opt, vocabulary_size_q, question:size(2), opt.gpuid)
nvqa.check(nvqa.lib.nvqa_init_params(ctx, opt.seed, -0.08, 0.08))                 -- :174-187
nvqa.check(nvqa.lib.nvqa_set_precision(ctx, opt.bf16))
nvqa.check(nvqa.lib.nvqa_set_ref_quirks(ctx, opt.ref_quirks))   -- a drop-in trains what the reference trains; -ref_quirks 0 fixes both
nvqa.check(nvqa.lib.nvqa_dataset_load(ctx, question:size(1), nvqa.iptr(question), nvqa.iptr(lengths),
           nvqa.iptr(img_list), nvqa.iptr(answers), fv_im:size(1), nvqa.fptr(fv_im), opt.img_norm))

local optimize = {learningRate = opt.learning_rate, weightDecay = 1e-4}             -- :193-198
local loss = ffi.new('float[1]')
local running_avg
paths.mkdir(opt.checkpoint_path .. 'save')
local fileLogger = io.open(opt.checkpoint_path .. 'save/logFile.txt', 'w')
local fileLoggerVal = io.open(opt.checkpoint_path .. 'save/logFileVal.txt', 'w')

-- validate() (:335-378): evaluate-mode forward over the validation split; with -ref_quirks 1 it reads the carried h0
-- exactly as the reference's encoder_model:forward does after a backward
local running_avg_val
local function validate()
  local nval, B = question_val:size(1), opt.batch_size
  local f_avg, iters, f = 0, 0, ffi.new('float[1]')
  for i = 1, nval, B do
    local r = math.min(i + B - 1, nval)
    local q = question_val[{{i, r}}]:contiguous()
    local im = fv_im_val:index(1, img_list_val[{{i, r}}]):contiguous()
    local y = answers_val[{{i, r}}]:contiguous()
    nvqa.check(nvqa.lib.nvqa_evaluate(ctx, r - i + 1, nvqa.iptr(q), nil, nvqa.fptr(im), nvqa.iptr(y), nil, 0, nil, nil, nil, f))
    running_avg_val = running_avg_val and (running_avg_val*0.95 + f[0]*0.05) or f[0]
    f_avg = f_avg + f[0]; iters = iters + 1
  end
  return f_avg / iters
end

local function save(path)                                                           -- :400-402
  local n = tonumber(nvqa.lib.nvqa_param_count(ctx))
  local x = torch.FloatTensor(n)
  nvqa.check(nvqa.lib.nvqa_get_params(ctx, nvqa.fptr(x)))
  local seg = ffi.new('size_t[3]'); nvqa.check(nvqa.lib.nvqa_segments(ctx, seg))
  local a, b = tonumber(seg[0]), tonumber(seg[1])
  torch.save(path, {cnn_w = x[{{1,a}}]:clone(), encoder_w_q = x[{{a+1,a+b}}]:clone(),
                    multimodal_w = x[{{a+b+1,n}}]:clone()})
end

local norms = ffi.new('float[3]')
for iter = 1, opt.max_iters do
  if iter % opt.save_checkpoint_every == 0 or iter == 1 then                        -- :393-403
    local loss_val = validate()
    fileLoggerVal:write('validation loss: ' .. loss_val .. ' validation loss avg: ' .. running_avg_val, ' on iter: ' .. iter .. '/' .. opt.max_iters .. '\n')
    print('validation loss: ' .. loss_val .. ' validation loss avg: ' .. running_avg_val .. ' on iter: ' .. iter .. '/' .. opt.max_iters .. '\n')
    save(string.format(opt.checkpoint_path..'save/lstm_save_iter%d.t7', iter))
  end
  local qinds = torch.LongTensor(opt.batch_size):random(question:size(1)):add(-1)   -- :209-212
  nvqa.check(nvqa.lib.nvqa_step_indices(ctx, nvqa.lptr(qinds), nvqa.dropout(1, opt.drop_prob_ae, opt.seed, iter), loss))
  -- gradients:clamp(-10,10), then optim.rmsprop with weightDecay (:327, :408; misc/rmsprop_lrscale.lua:16-34)
  nvqa.check(nvqa.lib.nvqa_rmsprop_update(ctx, optimize.learningRate, 0.99, 1e-8, optimize.weightDecay, 10))
  running_avg = running_avg and (running_avg*0.95 + loss[0]*0.05) or loss[0]
  if iter % 100 == 0 then                                                           -- :400-407
    -- torch.norm(cnn_w), torch.norm(encoder_w_q), torch.norm(multimodal_w): reduced on the device (a 49 MB copy otherwise)
    nvqa.check(nvqa.lib.nvqa_param_norms(ctx, norms))
    local line = string.format('iter: %6d train loss: %.3f cnn_norm: %.3f enc_norm: %.3f mm_norm: %.3f',
                               iter, running_avg, norms[0], norms[1], norms[2])
    fileLogger:write(line .. '\n')
    print(line)
  end
  optimize.learningRate = optimize.learningRate * decay_factor                      -- :410
end
fileLogger:close()
fileLoggerVal:close()
save(opt.checkpoint_path .. 'lstm.t7')
