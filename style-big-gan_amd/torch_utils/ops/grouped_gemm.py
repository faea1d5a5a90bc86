"""A pass's small fp32 products as one launch (csrc/grouped_gemm.hip, ``sbg_grouped_gemm``), and on top of it the synthesis network's style bank:
every layer's ``styles = affine(w)`` (reference train_parts/generators.py:333, 397; FullyConnectedLayer :117-131) of one pass from ONE launch, the
gradients of all of them from two (weights + biases; the `w` slots).  The reference runs one addmm per layer forward and two GEMMs and a
reduction per layer backward: ~140 launches of 5-40 us per training step."""
import ctypes

import torch

from ... import _lib


_library_products = False


def _mat(t):
    assert t.dtype == torch.float32 and t.ndim == 2 and t.is_cuda
    return t.data_ptr(), t.stride(0), t.stride(1)


def launch(problems, device):
    """problems: list of dicts  c = sum_t alpha_t a_t @ b_t (+ bias * bias_scale)  with keys c, terms = [(a, b, alpha), ...] (one or two; 2-D fp32 CUDA
    tensors or views, any strides), optional bias (1-D, contiguous), bias_scale, rowsum (1-D, contiguous: rowsum_scale * a_0.sum(1)), rowsum_scale"""
    if not problems:
        return
    table = (_lib.GgProblem * len(problems))()
    keep = []
    for q, pr in zip(table, problems):
        c = pr["c"]
        q.c, q.c_rs, q.c_cs = _mat(c)
        q.M, q.N = c.shape
        terms = pr["terms"]
        assert 1 <= len(terms) <= 2
        q.nterms = len(terms)
        for i, (a, b, alpha) in enumerate(terms):
            assert a.shape[0] == c.shape[0] and b.shape[1] == c.shape[1] and a.shape[1] == b.shape[0], (a.shape, b.shape, c.shape)
            ap, ars, acs = _mat(a)
            bp, brs, bcs = _mat(b)
            if i == 0:
                q.a0, q.a0_rs, q.a0_cs, q.b0, q.b0_rs, q.b0_cs, q.K0, q.alpha0 = ap, ars, acs, bp, brs, bcs, a.shape[1], float(alpha)
            else:
                q.a1, q.a1_rs, q.a1_cs, q.b1, q.b1_rs, q.b1_cs, q.K1, q.alpha1 = ap, ars, acs, bp, brs, bcs, a.shape[1], float(alpha)
            keep += [a, b]
        bias = pr.get("bias")
        if bias is not None:
            assert bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == c.shape[1]
            q.bias, q.bias_scale = bias.data_ptr(), float(pr.get("bias_scale", 1.0))
        rs = pr.get("rowsum")
        if rs is not None:
            assert rs.dtype == torch.float32 and rs.is_contiguous() and rs.numel() == c.shape[0]
            q.rowsum, q.rowsum_scale = rs.data_ptr(), float(pr.get("rowsum_scale", 1.0))
    _lib.check(_lib.load().sbg_grouped_gemm(ctypes.cast(table, ctypes.c_void_p), len(problems), _lib.stream_ptr(device)), "sbg_grouped_gemm")


class _StyleBank(torch.autograd.Function):
    """styles_l = alpha_l * ws[:, slot_l] @ W_l^T + beta_l * b_l  for every layer l of `plan` = ((slot, alpha, beta), ...);  inputs: ws [N, L, D]
    fp32, then W_0, b_0, W_1, b_1, ...  Returns one [N, C_l] tensor per layer (contiguous blocks of one allocation).  First order only."""

    @staticmethod
    def forward(ctx, plan, ws, *params):
        ws = ws.contiguous()
        n, _, d = ws.shape
        weights, biases = params[0::2], params[1::2]
        sizes = [w.shape[0] for w in weights]
        flat = torch.empty([n * sum(sizes)], dtype=torch.float32, device=ws.device)
        outs, off, problems = [], 0, []
        for (slot, alpha, beta), w, b, c in zip(plan, weights, biases, sizes):
            assert w.dtype == torch.float32 and w.is_contiguous() and w.shape[1] == d
            out = flat[off:off + n * c].view(n, c)
            off += n * c
            problems.append(dict(c=out, terms=[(ws[:, slot], w.t(), alpha)], bias=b, bias_scale=beta))
            outs.append(out)
        if _library_products:   # diagnosis (scratch/bank_diag3.py): the per-layer library products written into the bank's views, the bits FullyConnectedLayer produces
            for (slot, alpha, beta), w, b, out in zip(plan, weights, biases, outs):
                torch.addmm((b * beta if beta != 1 else b).unsqueeze(0), ws[:, slot], w.t(), alpha=alpha, out=out)
        else:
            launch(problems, ws.device)
        ctx.plan = plan
        ctx.save_for_backward(ws, *weights)
        ctx.has_bias = [b is not None for b in biases]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        if torch.is_grad_enabled():
            raise RuntimeError("style bank: first-order only; set torch_utils.ops.modconv.enabled = False before building a graph that is differentiated twice")
        ws, *weights = ctx.saved_tensors
        n, nslot, d = ws.shape
        plan = ctx.plan
        gs = [None if g is None else g.to(torch.float32).contiguous() for g in grads]
        need_ws = ctx.needs_input_grad[1]
        dparams = [None] * (2 * len(weights))
        problems = []
        for i, ((slot, alpha, beta), w, g) in enumerate(zip(plan, weights, gs)):
            need_w, need_b = ctx.needs_input_grad[2 + 2 * i], ctx.has_bias[i] and ctx.needs_input_grad[3 + 2 * i]
            if g is None:
                if need_w: dparams[2 * i] = torch.zeros_like(w)
                if need_b: dparams[2 * i + 1] = torch.zeros([w.shape[0]], dtype=torch.float32, device=w.device)
                continue
            if need_w or need_b:       # dW_l = alpha g^T ws[:, slot], db_l = beta sum_n g  (the row sums of g^T ride in the same tiles)
                dw = torch.empty_like(w)
                db = torch.empty([w.shape[0]], dtype=torch.float32, device=w.device) if need_b else None
                problems.append(dict(c=dw, terms=[(g.t(), ws[:, slot], alpha)], rowsum=db, rowsum_scale=beta))
                dparams[2 * i], dparams[2 * i + 1] = (dw if need_w else None), db
        dws = None
        if need_ws:                    # d ws[:, s] = sum over the (at most two) layers fed by slot s of alpha_l g_l W_l
            by_slot = {}
            for (slot, alpha, beta), w, g in zip(plan, weights, gs):
                if g is not None:
                    by_slot.setdefault(slot, []).append((g, w, alpha))
            full = all(s in by_slot for s in range(nslot))
            dws = (torch.empty if full else torch.zeros)([n, nslot, d], dtype=torch.float32, device=ws.device)
            for slot, terms in by_slot.items():
                if len(terms) > 2:
                    raise NotImplementedError("style bank: a w slot read by more than two layers")
                problems.append(dict(c=dws[:, slot], terms=terms))
        launch(problems, ws.device)
        return (None, dws, *dparams)


def style_bank(ws, layers):
    """layers: list of (slot, weight [C, D] fp32, bias [C] fp32 or None, alpha, beta) -> list of styles [N, C] fp32"""
    plan = tuple((int(slot), float(alpha), float(beta)) for slot, _, _, alpha, beta in layers)
    params = []
    for _, w, b, _, _ in layers:
        params += [w, b]
    return list(_StyleBank.apply(plan, ws, *params))
