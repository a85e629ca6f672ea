--[[
nvqa_ffi.lua -- LuaJIT FFI binding of libnvqa.so (include/nvqa.h).

This is the reference-side stub a maintainer drops next to misc/RNNUtils.lua: the Torch7
scripts keep their CmdLine flags and data loading and replace the body of JdJ (and the
optim.rmsprop call) with calls into the library.  The cdef below mirrors include/nvqa.h
one to one; novel-vqa_amd/host/binding.py is the same binding for Python/ctypes and is the
twin that the test-suite actually executes (no LuaJIT in the build image).
]]--
local ffi = require 'ffi'

ffi.cdef[[
typedef struct nvqa_dims { int32_t arch, B, T, V, E, R, L, I, C, A; } nvqa_dims;
typedef struct nvqa_dropout { int32_t mode; float p; uint64_t seed; uint64_t step; } nvqa_dropout;
typedef struct nvqa_ctx nvqa_ctx;
int nvqa_create(const nvqa_dims *dims, int device, nvqa_ctx **out);
int nvqa_destroy(nvqa_ctx *ctx);
const char *nvqa_last_error(void);
int nvqa_sync(nvqa_ctx *ctx);
size_t nvqa_param_count(const nvqa_ctx *ctx);
int nvqa_segments(const nvqa_ctx *ctx, size_t sizes_out[3]);
int nvqa_init_params(nvqa_ctx *ctx, uint64_t seed, float lo, float hi);
int nvqa_set_params(nvqa_ctx *ctx, const float *params);
int nvqa_get_params(nvqa_ctx *ctx, float *params_out);
int nvqa_get_grads(nvqa_ctx *ctx, float *grads_out, float clamp);
int nvqa_step(nvqa_ctx *ctx, const int32_t *tokens, const int32_t *lengths, const float *img,
              const int32_t *labels, const nvqa_dropout *dropout, float *loss_out);
int nvqa_get_loss(nvqa_ctx *ctx, float *loss_out);
int nvqa_forward(nvqa_ctx *ctx, int32_t n, const int32_t *tokens, const int32_t *lengths,
                 const float *img, float *scores_out, int32_t *argmax_out);
int nvqa_evaluate(nvqa_ctx *ctx, int32_t n, const int32_t *tokens, const int32_t *lengths, const float *img,
                  const int32_t *labels, const int32_t *mc_ans, int32_t n_mc, float *scores_out,
                  int32_t *argmax_out, int32_t *mc_argmax_out, float *loss_out);
int nvqa_rmsprop_update(nvqa_ctx *ctx, float lr, float alpha, float eps, float wd, float clamp);
int nvqa_set_fusion(nvqa_ctx *ctx, int mode);
int nvqa_set_precision(nvqa_ctx *ctx, int bf16);
int nvqa_set_ref_quirks(nvqa_ctx *ctx, int flags);
int nvqa_param_norms(nvqa_ctx *ctx, float out[3]);
int nvqa_persistent_state(const nvqa_ctx *ctx, int out[2]);
int nvqa_set_grad_scales(nvqa_ctx *ctx, const float scales[3]);
int nvqa_dataset_load(nvqa_ctx *ctx, int64_t n_q, const int32_t *questions, const int32_t *lengths,
                      const int32_t *img_pos, const int32_t *answers, int64_t n_img,
                      const float *feats, int l2_normalize);
int nvqa_step_indices(nvqa_ctx *ctx, const int64_t *qinds, const nvqa_dropout *dropout, float *loss_out);
int nvqa_comm_unique_id(void *id_out);
int nvqa_comm_init(nvqa_ctx *ctx, int rank, int world, const void *id);
const char *nvqa_comm_library(void);
typedef struct nvqa_vgg nvqa_vgg;
int nvqa_vgg16_create(int device, int width_div, int input_hw, int max_batch, nvqa_vgg **out);
int nvqa_vgg16_destroy(nvqa_vgg *vgg);
size_t nvqa_vgg16_weight_count(const nvqa_vgg *vgg);
int nvqa_vgg16_feature_dim(const nvqa_vgg *vgg);
int nvqa_vgg16_set_weights(nvqa_vgg *vgg, const float *flat);
int nvqa_vgg16_fc7(nvqa_vgg *vgg, const float *images, int n, float *feats_out);
int nvqa_vgg16_set_precision(nvqa_vgg *vgg, int bf16);
int nvqa_vgg16_preprocess(nvqa_vgg *vgg, const float *rgb, int n, int H, int W, float *out);
int nvqa_step_images(nvqa_ctx *ctx, nvqa_vgg *vgg, const float *images, const int32_t *tokens,
                     const int32_t *lengths, const int32_t *labels, const nvqa_dropout *dropout, float *loss_out);
int nvqa_profile_enable(nvqa_ctx *ctx, int enable);
int nvqa_profile_reset(nvqa_ctx *ctx);
int nvqa_profile_count(const nvqa_ctx *ctx);
const char *nvqa_profile_name(const nvqa_ctx *ctx, int idx);
int nvqa_profile_get(nvqa_ctx *ctx, int idx, double *total_ms, int64_t *launches, double *flops, double *bytes);
]]

local M = {}
M.lib = ffi.load(os.getenv('NVQA_LIB') or 'nvqa')   -- libnvqa.so on the loader path

-- Torch raises Lua errors on misuse and the scripts never pcall: keep that fail-fast behaviour.
local function check(rc)
  if rc ~= 0 then error('libnvqa: ' .. ffi.string(M.lib.nvqa_last_error()), 2) end
end
M.check = check

function M.create(arch, opt, vocabulary_size, buffer_size, device)
  local d = ffi.new('nvqa_dims')
  d.arch = arch; d.B = opt.batch_size; d.T = buffer_size; d.V = vocabulary_size
  d.E = opt.input_encoding_size; d.R = opt.rnn_size
  d.L = (arch == 1) and opt.rnn_layer or opt.num_layers
  d.I = opt.nhimage; d.C = opt.common_embedding_size; d.A = opt.num_output
  local out = ffi.new('nvqa_ctx*[1]')
  check(M.lib.nvqa_create(d, device or 0, out))
  return ffi.gc(out[0], M.lib.nvqa_destroy), d
end

-- pointers to the storage of contiguous torch tensors (IntTensor / FloatTensor / LongTensor)
function M.iptr(t) return ffi.cast('const int32_t*', t:data()) end
function M.fptr(t) return ffi.cast('float*', t:data()) end
function M.lptr(t) return ffi.cast('const int64_t*', t:data()) end

function M.dropout(mode, p, seed, step)
  local d = ffi.new('nvqa_dropout'); d.mode = mode; d.p = p; d.seed = seed; d.step = step
  return d
end

return M
