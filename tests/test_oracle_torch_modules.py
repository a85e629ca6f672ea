"""The oracle against PyTorch's own library modules.

Torch7 cannot run here, so the oracle is "parity unpinned" with respect to the reference itself.  The nearest
third-party implementation of the same module semantics that IS installed is PyTorch (the nn / optim packages
descend from Torch7's: nn.Linear y = xW^T + b, nn.LSTMCell's cell equations, nn.CrossEntropyLoss = LogSoftMax +
ClassNLL with mean reduction, optim.RMSprop's square_avg / sqrt + eps update, weight decay added to the gradient).
This test assembles the arch1 step from those modules -- torch.nn.LSTMCell (gate order i,f,g,o: the oracle's
i,f,o,g weights are permuted into it), torch.nn.Linear, torch.nn.Embedding, torch.nn.CrossEntropyLoss,
torch.nn.utils.clip_grad_value_, torch.optim.RMSprop -- and compares loss, logits, every gradient and the
updated parameters with the oracle.  tests/ref_autograd.py is a second, hand-written statement; this one leans
on code neither this repository nor the reference wrote."""
import numpy as np
import pytest
import torch

import ref_autograd as ra
from util import relmax

TINY = dict(arch=1, B=6, T=7, V=13, E=8, R=8, L=2, I=12, C=12, A=8)


def _perm_ifog_to_ifgo(w, R):
    """rows [i f o g] (misc/LSTM.lua:45-52) -> PyTorch's [i f g o]."""
    return torch.cat([w[0:R], w[R:2 * R], w[3 * R:4 * R], w[2 * R:3 * R]], 0)


@pytest.mark.parametrize("mode", [0, 1])
def test_arch1_step_and_update_against_torch_nn_modules(orc, mode):
    d = orc.make_dims(**TINY)
    B, T, V, E, R, L, I, C, A = d.B, d.T, d.V, d.E, d.R, d.L, d.I, d.C, d.A
    params = orc.synth_params(d).astype(np.float64)
    tok, lens, img, lab = orc.synth_batch(d, full_length=False)
    dr = orc.Dropout(mode, 0.5, 123, 9)
    lo = orc.layout(d)
    p = {k: torch.tensor(params[v[0]:v[0] + v[1]]) for k, v in lo.items() if not k.startswith("_")}

    torch.set_default_dtype(torch.float64)
    try:
        emb = torch.nn.Embedding(V + 1, E, padding_idx=0)        # row 0 = padding; the reference's one-hot x Linear
        emb.weight.data[1:] = p["w_e"].view(E, V).t()
        emb.weight.data[0] = 0
        b_e = torch.nn.Parameter(p["b_e"].clone())
        cells = []
        for l in range(L):
            inn = E if l == 0 else R
            cell = torch.nn.LSTMCell(inn, R)
            cell.weight_ih.data = _perm_ifog_to_ifgo(p[f"w_i2h{l}"].view(4 * R, inn), R)
            cell.weight_hh.data = _perm_ifog_to_ifgo(p[f"w_h2h{l}"].view(4 * R, R), R)
            cell.bias_ih.data = _perm_ifog_to_ifgo(p[f"b_i2h{l}"], R)
            cell.bias_hh.data = _perm_ifog_to_ifgo(p[f"b_h2h{l}"], R)
            cells.append(cell)
        Q = 2 * R * L
        lin_q, lin_v, lin_o = torch.nn.Linear(Q, C), torch.nn.Linear(I, C), torch.nn.Linear(C, A)
        for m, w, b in ((lin_q, "w_q", "b_q"), (lin_v, "w_v", "b_v"), (lin_o, "w_o", "b_o")):
            m.weight.data = p[w].view(m.out_features, m.in_features).clone()
            m.bias.data = p[b].clone()

        De = ra.drop_scales(dr, 0, (B, T, E), lambda b, t, e: (b * T + t) * E + e)
        Dl = [None] + [ra.drop_scales(dr, 1, (B, T, R), lambda b, t, j, l=l: (((l - 1) * B + b) * T + t) * R + j)
                       for l in range(1, L)]
        Dq = ra.drop_scales(dr, 2, (B, Q), lambda b, j: b * Q + j)
        Dv = ra.drop_scales(dr, 3, (B, I), lambda b, j: b * I + j)
        Dz = ra.drop_scales(dr, 4, (B, C), lambda b, j: b * C + j)
        toks = torch.tensor(np.asarray(tok, np.int64))
        h = [torch.zeros(B, R) for _ in range(L)]
        c = [torch.zeros(B, R) for _ in range(L)]
        for t in range(T):
            active = (toks[:, t] != 0).to(torch.float64).view(B, 1)
            # the reference's one-hot row of a padded position is all zero, so its Linear output is the bias alone;
            # those rows are inactive anyway (state forced to zero below)
            x = torch.tanh(De[:, t, :] * (emb(toks[:, t]) + b_e))
            for l in range(L):
                u = x if l == 0 else Dl[l][:, t, :] * h[l - 1]
                hn, cn = cells[l](u, (h[l], c[l]))
                h[l], c[l] = active * hn, active * cn             # misc/RNNUtils.lua:136-145
        q = torch.cat([torch.cat([c[l], h[l]], 1) for l in range(L)], 1)
        z = torch.tanh(lin_q(Dq * q)) * torch.tanh(lin_v(Dv * torch.tensor(np.asarray(img, np.float64))))
        scores = lin_o(Dz * z)
        loss = torch.nn.CrossEntropyLoss()(scores, torch.tensor(np.asarray(lab, np.int64) - 1))
        loss.backward()

        got = orc.Oracle(np.float64).step(d, params, tok, lens, img, lab, dr)
        assert abs(got["loss"] - float(loss.detach())) < 1e-12
        assert relmax(got["scores"], scores.detach().numpy()) < 1e-12

        def back(g, R_=R):   # PyTorch's [i f g o] gradient rows -> the oracle's [i f o g]
            return torch.cat([g[0:R_], g[R_:2 * R_], g[3 * R_:4 * R_], g[2 * R_:3 * R_]], 0)

        want = {"w_e": emb.weight.grad[1:].t().reshape(-1), "b_e": b_e.grad,
                "w_q": lin_q.weight.grad.reshape(-1), "b_q": lin_q.bias.grad, "w_v": lin_v.weight.grad.reshape(-1),
                "b_v": lin_v.bias.grad, "w_o": lin_o.weight.grad.reshape(-1), "b_o": lin_o.bias.grad}
        for l in range(L):
            want[f"w_i2h{l}"] = back(cells[l].weight_ih.grad).reshape(-1)
            want[f"w_h2h{l}"] = back(cells[l].weight_hh.grad).reshape(-1)
            want[f"b_i2h{l}"] = back(cells[l].bias_ih.grad)
            want[f"b_h2h{l}"] = back(cells[l].bias_hh.grad)
        for k, (o, n) in ((k, v) for k, v in lo.items() if not k.startswith("_")):
            assert relmax(got["grads"][o:o + n], want[k].numpy()) < 1e-10, k

        # clamp(-10, 10) + optim.rmsprop (002_train_baseline.lua:329,408) against clip_grad_value_ + torch.optim.RMSprop
        # on a scaled-up gradient so that the clamp bites, two steps so that the running average matters
        x = torch.nn.Parameter(torch.tensor(params))
        opt = torch.optim.RMSprop([x], lr=3e-4, alpha=0.99, eps=1e-8, weight_decay=1e-4)
        xo, mo = params.copy(), np.zeros_like(params)
        o = orc.Oracle(np.float64)
        for it in range(2):
            g = got["grads"] * (4e3 if it == 0 else 1.0)
            x.grad = torch.tensor(g.copy())
            torch.nn.utils.clip_grad_value_([x], 10.0)
            opt.step()
            go = g.copy()
            o.rmsprop(xo, go, mo, 3e-4, 0.99, 1e-8, 1e-4, 10.0)
        assert np.abs(got["grads"] * 4e3).max() > 10.0
        assert relmax(xo, x.detach().numpy()) < 1e-12
    finally:
        torch.set_default_dtype(torch.float32)


def test_arch2_step_against_torch_nn_modules(orc):
    """arch2 (003_train_vqa_arch2: image-as-first-token encoder, shared nn.LookupTable, START token = row V+1,
    nulls rewritten to word 1, Encoder_lstm.lua:152-227) from torch.nn.Embedding / LSTMCell / Linear."""
    d = orc.make_dims(arch=2, B=5, T=6, V=11, E=8, R=8, L=2, I=12, C=4, A=8)
    B, T, V, E, R, L, I, A = d.B, d.T, d.V, d.E, d.R, d.L, d.I, d.A
    TS = T + 2
    params = orc.synth_params(d).astype(np.float64)
    tok, _, img, lab = orc.synth_batch(d, full_length=False)
    dr = orc.Dropout(1, 0.5, 123, 3)
    lo = orc.layout(d)
    p = {k: torch.tensor(params[v[0]:v[0] + v[1]]) for k, v in lo.items() if not k.startswith("_")}
    torch.set_default_dtype(torch.float64)
    try:
        lookup = torch.nn.Embedding(V + 1, E)
        lookup.weight.data = p["w_lk"].view(V + 1, E).clone()
        proj, cls = torch.nn.Linear(I, E), torch.nn.Linear(R, A)
        proj.weight.data, proj.bias.data = p["w_p"].view(E, I).clone(), p["b_p"].clone()
        cls.weight.data, cls.bias.data = p["w_o"].view(A, R).clone(), p["b_o"].clone()
        cells = []
        for l in range(L):
            inn = E if l == 0 else R
            cell = torch.nn.LSTMCell(inn, R)
            cell.weight_ih.data = _perm_ifog_to_ifgo(p[f"w_i2h{l}"].view(4 * R, inn), R)
            cell.weight_hh.data = _perm_ifog_to_ifgo(p[f"w_h2h{l}"].view(4 * R, R), R)
            cell.bias_ih.data = _perm_ifog_to_ifgo(p[f"b_i2h{l}"], R)
            cell.bias_hh.data = _perm_ifog_to_ifgo(p[f"b_h2h{l}"], R)
            cells.append(cell)
        Dl = [None] + [ra.drop_scales(dr, 1, (B, TS, R), lambda b, t, j, l=l: (((l - 1) * B + b) * TS + t) * R + j)
                       for l in range(1, L)]
        Dh = ra.drop_scales(dr, 2, (B, R), lambda b, j: b * R + j)
        toks = np.asarray(tok).reshape(B, T)
        h = [torch.zeros(B, R) for _ in range(L)]
        c = [torch.zeros(B, R) for _ in range(L)]
        for t in range(1, TS + 1):
            if t == 1:
                x = proj(torch.tensor(np.asarray(img, np.float64)))
            elif t == 2:
                x = lookup(torch.full((B,), V, dtype=torch.int64))
            else:
                it = toks[:, t - 3].copy()
                if it.sum() == 0:
                    break
                it[it == 0] = 1
                x = lookup(torch.tensor(it.astype(np.int64) - 1))
            for l in range(L):
                u = x if l == 0 else Dl[l][:, t - 1, :] * h[l - 1]
                h[l], c[l] = cells[l](u, (h[l], c[l]))
        scores = cls(Dh * h[L - 1])
        loss = torch.nn.CrossEntropyLoss()(scores, torch.tensor(np.asarray(lab, np.int64) - 1))
        loss.backward()
        got = orc.Oracle(np.float64).step(d, params, tok, None, img, lab, dr)
        assert abs(got["loss"] - float(loss.detach())) < 1e-12

        def back(g):
            return torch.cat([g[0:R], g[R:2 * R], g[3 * R:4 * R], g[2 * R:3 * R]], 0)

        want = {"w_lk": lookup.weight.grad.reshape(-1), "w_p": proj.weight.grad.reshape(-1), "b_p": proj.bias.grad,
                "w_o": cls.weight.grad.reshape(-1), "b_o": cls.bias.grad}
        for l in range(L):
            want[f"w_i2h{l}"] = back(cells[l].weight_ih.grad).reshape(-1)
            want[f"w_h2h{l}"] = back(cells[l].weight_hh.grad).reshape(-1)
            want[f"b_i2h{l}"] = back(cells[l].bias_ih.grad)
            want[f"b_h2h{l}"] = back(cells[l].bias_hh.grad)
        for k, (o, n) in ((k, v) for k, v in lo.items() if not k.startswith("_")):
            assert relmax(got["grads"][o:o + n], want[k].numpy()) < 1e-10, k
    finally:
        torch.set_default_dtype(torch.float32)
