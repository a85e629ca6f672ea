--[[
convert_ae_arch2.lua -- the arch2 counterpart of 001_train_autoencoder/002_convert_text_model_arch1.lua:27-39: turns an arch2
text auto-encoder checkpoint (torch.load(path).protos.ae: nn modules) into a table of plain tensors,

    { encoder = <flat parameters of protos.ae.encoder, getParameters() order>, lookup_table = <[(V+1) x E] weight> }

which train_arch2_ae.lua and VQATrainer.init_from_autoencoder read.  003_train_vqa_arch2/003_train_ae_based.lua:74-75,150-152
clones the two modules straight out of the checkpoint; libnvqa's hosts read tensors only (host/t7.py executes nothing from a
file), so this step runs ONCE under Torch7, where nn / nngraph can deserialise the modules.  Not executable in the build image.
]]--
require 'torch'
require 'nn'
require 'nngraph'
local net_utils = require 'misc.net_utils'          -- the reference's (unsanitize_gradients)
require 'misc.AutoEncoder'                          -- class definitions the checkpoint refers to

cmd = torch.CmdLine()
cmd:option('-model_path', '', 'path to the auto-encoder checkpoint')
cmd:option('-save_path', '', 'path to save the tensor table')
local opt = cmd:parse(arg)

local ae = torch.load(opt.model_path).protos.ae
net_utils.unsanitize_gradients(ae.encoder)
local encoder_params = ae.encoder:getParameters()
local lookup_params = ae.lookup_table:parameters()
torch.save(opt.save_path, {encoder = encoder_params:float():clone(), lookup_table = lookup_params[1]:float():clone()})
