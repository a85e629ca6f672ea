"""Independent float64 autograd statement of the arch1 / arch2 equations.

Purpose: cross-check oracle/nvqa_oracle.c (SURVEY.md 8c "how the oracle is
validated without the reference").  It deliberately uses a DIFFERENT
formulation from the oracle: no length sort, no time-major packing, the
literal one-hot x Linear embedding, per-row activity masks, and PyTorch
autograd instead of hand-written backward passes.

Equations: SURVEY.md Appendix A.1 / A.2 (002_train_vqa_arch1/misc/LSTM.lua:41-59,
misc/netdef.lua:6-14, 002_train_baseline.lua:141-157,300-320;
003_train_vqa_arch2/misc/Encoder_lstm.lua:152-263).
"""
import numpy as np
import torch

M64 = (1 << 64) - 1


# bf16 operand mode (nvqa_set_precision / oracle.set_precision): both operands of every dense product are
# rounded to bf16 (torch's f32 -> bf16 cast rounds to nearest even) in the forward AND in the two backward
# products; biases and bias gradients are untouched.  Stated as a custom autograd function so that it
# is independent of the oracle's loop nests.
BF16 = False


def _r(t):
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


class _BfMatmul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W):
        ctx.save_for_backward(x, W)
        return _r(x) @ _r(W).t()

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        return _r(dy) @ _r(W), _r(dy).t() @ _r(x)


def lin(x, W, b):
    """nn.Linear: y = x W^T + b."""
    return (_BfMatmul.apply(x, W) if BF16 else x @ W.t()) + b


def hash32(seed, step, site, idx):
    """Python restatement of nvqa_hash32 (include/nvqa_rng.h)."""
    x = (seed ^ ((0x9E3779B97F4A7C15 * (step + 1)) & M64) ^ ((site << 56) & M64)) & M64
    x = (x + idx * 0xD1342543DE82EF95) & M64
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & M64
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & M64
    x ^= x >> 31
    return x >> 32


def drop_scales(dr, site, shape, index_fn):
    """Tensor of dropout multipliers; index_fn(*coords) -> element index."""
    out = np.ones(shape, np.float64)
    if dr is None or dr.mode == 0:
        return torch.from_numpy(out)
    inv_keep = float(np.float32(1.0) / (np.float32(1.0) - np.float32(dr.p)))
    for coords in np.ndindex(*shape):
        u = np.float32(hash32(dr.seed, dr.step, site, index_fn(*coords)) >> 8) * np.float32(1.0 / 16777216.0)
        out[coords] = inv_keep if u >= np.float32(dr.p) else 0.0
    return torch.from_numpy(out)


def _split(params, lo):
    return {k: params[v[0]:v[0] + v[1]] for k, v in lo.items() if not k.startswith("_")}


def _cell(a, c_prev, R):
    i = torch.sigmoid(a[:, 0:R])
    f = torch.sigmoid(a[:, R:2 * R])
    o = torch.sigmoid(a[:, 2 * R:3 * R])
    g = torch.tanh(a[:, 3 * R:4 * R])
    c = f * c_prev + i * g
    return c, o * torch.tanh(c)


def arch1(dims, lo, params_np, tokens, lengths, img, labels, dr=None, train=True, askip=False):
    d = dims
    B, T, V, E, R, L, I, C, A = d.B, d.T, d.V, d.E, d.R, d.L, d.I, d.C, d.A
    params = torch.tensor(np.asarray(params_np, np.float64), requires_grad=True)
    p = _split(params, lo)
    drr = dr if train else None
    We = p["w_e"].view(E, V)
    tok = np.asarray(tokens).reshape(B, T)
    c = [torch.zeros(B, R, dtype=torch.float64) for _ in range(L)]
    h = [torch.zeros(B, R, dtype=torch.float64) for _ in range(L)]
    De = drop_scales(drr, 0, (B, T, E), lambda b, t, e: (b * T + t) * E + e)
    Dl = [None] + [drop_scales(drr, 1, (B, T, R), lambda b, t, j, l=l: (((l - 1) * B + b) * T + t) * R + j)
                   for l in range(1, L)]
    for t in range(T):
        active = torch.from_numpy((tok[:, t] != 0).astype(np.float64)).view(B, 1)
        onehot = torch.zeros(B, V, dtype=torch.float64)
        for b in range(B):
            if tok[b, t] != 0:
                onehot[b, tok[b, t] - 1] = 1.0
        x = torch.tanh(De[:, t, :] * (onehot @ We.t() + p["b_e"]))
        for l in range(L):
            inn = E if l == 0 else R
            u = x if l == 0 else Dl[l][:, t, :] * h[l - 1]
            a = lin(u, p[f"w_i2h{l}"].view(4 * R, inn), p[f"b_i2h{l}"]) \
                + lin(h[l], p[f"w_h2h{l}"].view(4 * R, R), p[f"b_h2h{l}"])
            cn, hn = _cell(a, c[l], R)
            # rows that have not started keep their zero state (RNNUtils.lua:136-145)
            c[l] = active * cn
            h[l] = active * hn
    q = torch.cat([torch.cat([c[l], h[l]], 1) for l in range(L)], 1)  # [c1 h1 c2 h2] LSTM.lua:70
    Q = 2 * R * L
    Dq = drop_scales(drr, 2, (B, Q), lambda b, j: b * Q + j)
    Dv = drop_scales(drr, 3, (B, I), lambda b, j: b * I + j)
    join = askip == 2  # netdef.A_B (misc/netdef.lua:27-35): JoinTable(2)({qc, ic}), classifier Linear(2C, A)
    ZW = 2 * C if join else C
    Dz = drop_scales(drr, 4, (B, ZW), lambda b, j: b * ZW + j)
    v = torch.tensor(np.asarray(img, np.float64).reshape(B, I))
    qc = torch.tanh(lin(Dq * q, p["w_q"].view(C, Q), p["b_q"]))
    ic = torch.tanh(lin(Dv * v, p["w_v"].view(C, I), p["b_v"]))
    fused = torch.cat([qc, ic], 1) if join else (qc + qc * ic if askip else qc * ic)  # netdef.A_B / AskipB / AxB
    scores = lin(Dz * fused, p["w_o"].view(A, ZW), p["b_o"])
    y = torch.tensor(np.asarray(labels, np.int64) - 1)
    loss = torch.nn.functional.cross_entropy(scores, y, reduction="mean")
    grads = None
    if train:
        loss.backward()
        grads = params.grad.numpy().copy()
    return {"loss": float(loss.detach()), "scores": scores.detach().numpy(), "grads": grads}


class AliasedH0:
    """The tensor that misc/Encoder_lstm.lua:238-239 aliases (quirk Q1): slot 2L of init_state_enc becomes
    multimodal_net.gradInput after the first backward and is never re-zeroed (:30-47).  `value` is what a
    forward pass reads as the top layer's h0."""

    def __init__(self, B, R):
        self.value = torch.zeros(B, R, dtype=torch.float64)


def arch2(dims, lo, params_np, tokens, img, labels, dr=None, train=True, aliased_h0=None, no_lookup_grad=False):
    """aliased_h0: an AliasedH0 carried across calls reproduces quirk Q1 with autograd plus the one correction its
    mechanics imply (the step-1 clone's accGradParameters reads the tensor AFTER multimodal_net:backward rewrote
    it).  no_lookup_grad: quirk Q11, the lookup slice of the flat gradient stays zero."""
    d = dims
    B, T, V, E, R, L, I, A = d.B, d.T, d.V, d.E, d.R, d.L, d.I, d.A
    TS = T + 2
    params = torch.tensor(np.asarray(params_np, np.float64), requires_grad=True)
    p = _split(params, lo)
    drr = dr if train else None
    tok = np.asarray(tokens).reshape(B, T)
    Wlk = p["w_lk"].view(V + 1, E)
    v = torch.tensor(np.asarray(img, np.float64).reshape(B, I))
    c = [torch.zeros(B, R, dtype=torch.float64) for _ in range(L)]
    h = [torch.zeros(B, R, dtype=torch.float64) for _ in range(L)]
    h0_old = None
    if aliased_h0 is not None:
        h0_old = aliased_h0.value.clone()
        h[L - 1] = h0_old
    a_top_first = None
    Dl = [None] + [drop_scales(drr, 1, (B, TS, R), lambda b, t, j, l=l: (((l - 1) * B + b) * TS + t) * R + j)
                   for l in range(1, L)]
    for t in range(1, TS + 1):
        if t == 1:
            x = lin(v, p["w_p"].view(E, I), p["b_p"])
        elif t == 2:
            x = Wlk[V].expand(B, E)
        else:
            it = tok[:, t - 3].copy()
            if it.sum() == 0:
                break
            it[it == 0] = 1
            x = Wlk[torch.from_numpy(it.astype(np.int64) - 1)]
        for l in range(L):
            inn = E if l == 0 else R
            u = x if l == 0 else Dl[l][:, t - 1, :] * h[l - 1]
            a = lin(u, p[f"w_i2h{l}"].view(4 * R, inn), p[f"b_i2h{l}"]) \
                + lin(h[l], p[f"w_h2h{l}"].view(4 * R, R), p[f"b_h2h{l}"])
            if t == 1 and l == L - 1 and train:
                a.retain_grad()
                a_top_first = a
            c[l], h[l] = _cell(a, c[l], R)
    hfin = h[L - 1]
    if train:
        hfin.retain_grad()
    Dh = drop_scales(drr, 2, (B, R), lambda b, j: b * R + j)
    scores = lin(Dh * hfin, p["w_o"].view(A, R), p["b_o"])
    y = torch.tensor(np.asarray(labels, np.int64) - 1)
    loss = torch.nn.functional.cross_entropy(scores, y, reduction="mean")
    grads = None
    if train:
        loss.backward()
        grads = params.grad.numpy().copy()
        if aliased_h0 is not None:
            # multimodal_net:backward leaves dL/d(encoder output) in the aliased tensor BEFORE the encoder's backward
            # runs; the step-1 nn.Linear(h2h) of the top layer then forms dW += dgates_1^T x (that tensor)
            h0_new = hfin.grad.clone()
            o, n = lo[f"w_h2h{L - 1}"]
            corr = a_top_first.grad.t() @ (h0_new - h0_old)  # [4R x R]
            grads[o:o + n] += corr.reshape(-1).numpy()
            aliased_h0.value = h0_new
        if no_lookup_grad:
            o, n = lo["w_lk"]
            grads[o:o + n] = 0.0
    return {"loss": float(loss.detach()), "scores": scores.detach().numpy(), "grads": grads}
