"""StepEngine on the GPU beyond the reference-schedule fixtures (tests/test_reference_vectors_gpu.py): snapshot -> resume continuity with
the bf16 kernels in the loop, the mixed-precision schedules of the BASELINE configs at reduced size (ffhq_sg2: path length + R1 on a resnet
D with conv_clamp -- the in-place residual add that bench.py at 1024x1024 tripped over; sg2attent: attention in G and D; big_gan at its real
128x128 architecture), each checked for finite, moving weights and for the expected phase structure."""
import io

import pytest
import torch

import style_big_gan_amd  # noqa: F401
from style_big_gan_amd.train_parts import trainers

pytestmark = pytest.mark.gpu


def _sg2_kwargs(res=32, cb=1024, cm=64, d_arch="orig", map_layers=2, attn_g=(), attn_d=(), nfp=8, clamp=256):
    gk = dict(z_dim=32, c_dim=0, w_dim=32, img_resolution=res, img_channels=3, attentions=list(attn_g), mapping_kwargs=dict(num_layers=map_layers),
              synthesis_kwargs=dict(channel_base=cb, channel_max=cm, num_fp16_res=nfp, block_kwargs=dict(conv_clamp=clamp)))
    dk = dict(c_dim=0, img_resolution=res, img_channels=3, attentions=list(attn_d), architecture=d_arch, channel_base=cb, channel_max=cm, num_fp16_res=nfp,
              conv_clamp=clamp, epilogue_kwargs=dict(mbstd_group_size=4))
    return gk, dk


def _finite_and_moved(eng, before_g, before_d):
    for net, before in ((eng.G, before_g), (eng.D, before_d)):
        assert all(bool(torch.isfinite(p).all()) for p in net.parameters())
        assert sum(float((a - b).abs().sum()) for a, b in zip(before, net.parameters())) > 0


def test_snapshot_resume_continuity_on_device(dev, tmp_path):
    """state_dict -> torch.save -> weights_only load -> fresh engines continue where the first one would have: same weights, optimizer
    moments and counters after loading; two engines resumed from the same snapshot stay BIT-IDENTICAL through a further iteration with the
    same inputs and seeds (the kernels are deterministic: fixed-order reductions, no float atomics), and they track the original engine to
    within the last-bit differences that library GEMM selection introduces between processes' allocation histories -- which Adam's
    normalised update (beta1 = 0) can turn into a full +-lr step on elements whose gradient is near zero, hence the two-part bound"""
    gk, dk = _sg2_kwargs()
    kw = dict(gen_kwargs=gk, disc_kwargs=dk, loss_arch_kwargs=dict(style_mixing_prob=0), dis_regs=[("r1", dict(r1_gamma=0.1))], g_reg_interval=4,
              d_reg_interval=2, batch=8, batch_gpu=4, ema_kimg=0.05)
    a = trainers.StepEngine(dev, seed=1, **kw)
    gen = torch.Generator(device=dev).manual_seed(3)
    for _ in range(3):
        a.train_iteration(torch.rand(8, 3, 32, 32, device=dev, generator=gen) * 2 - 1, None)
    buf = io.BytesIO()
    torch.save(a.state_dict(), buf)
    resumed = []
    for seed in (99, 100):                                              # different initialisations: everything must come from the snapshot
        buf.seek(0)
        e = trainers.StepEngine(dev, seed=seed, **kw)
        e.load_state_dict(torch.load(buf, map_location=dev, weights_only=True))
        assert e.cur_nimg == a.cur_nimg == 24 and e.batch_idx == a.batch_idx == 3
        for ma, mb in ((a.G, e.G), (a.D, e.D), (a.G_ema, e.G_ema)):
            for (k, va), (_, vb) in zip(ma.state_dict().items(), mb.state_dict().items()):
                assert torch.equal(va, vb), k
        for pa, pe in zip(a.phases, e.phases):
            sa, se = pa.opt.state_dict()["state"], pe.opt.state_dict()["state"]
            assert sa.keys() == se.keys() and all(torch.equal(sa[k]["exp_avg_sq"], se[k]["exp_avg_sq"]) and float(sa[k]["step"]) == float(se[k]["step"]) for k in sa)
        resumed.append(e)
    b, c = resumed
    real = torch.rand(8, 3, 32, 32, device=dev, generator=gen) * 2 - 1
    z = torch.randn(len(a.phases) * 8, 32, device=dev, generator=gen)
    for eng in (a, b, c):
        torch.manual_seed(1234)                                         # the synthesis noise is drawn from the device generator
        eng.train_iteration(real, None, all_gen_z=z)
    snap = None
    buf.seek(0)
    snap = torch.load(buf, map_location=dev, weights_only=True)
    for net, ma, mb, mc in (("G", a.G, b.G, c.G), ("D", a.D, b.D, c.D), ("G_ema", a.G_ema, b.G_ema, c.G_ema)):
        num = den = 0.0
        for (k, va), (_, vb), (_, vc) in zip(ma.state_dict().items(), mb.state_dict().items(), mc.state_dict().items()):
            assert torch.equal(vb, vc), f"{k}: two runs resumed from one snapshot diverged"
            d0 = snap[net][k].float()
            num += float(((va.float() - d0) - (vb.float() - d0)).square().sum()); den += float((vb.float() - d0).square().sum())
        # the original and the resumed engines take the same step up to rounding noise that Adam's normalisation amplifies on near-zero gradients
        assert num <= 0.25 * den, f"{net}: update of the resumed run differs from the original's ({(num / max(den, 1e-30)) ** 0.5:.2f} relative)"
    for e in (a, b, c):
        e.close()


def test_ffhq_sg2_schedule_bf16(dev):
    """configs/ffhq_sg2.yaml's recipe at 64x64: 6 mapping layers, resnet D, mbstd, R1 + path length, bf16 blocks with conv_clamp"""
    gk, dk = _sg2_kwargs(res=64, cb=2048, cm=64, d_arch="resnet", map_layers=6)
    eng = trainers.StepEngine(dev, gen_kwargs=gk, disc_kwargs=dk, loss_arch_kwargs=dict(style_mixing_prob=0), gen_regs=[("ppl", dict(pl_batch_shrink=2, pl_decay=0.01, pl_weight=2.))],
                              dis_regs=[("r1", dict(r1_gamma=1.))], g_reg_interval=2, d_reg_interval=2, batch=8, batch_gpu=4, ema_kimg=0.02)
    assert [(p.name, p.interval, p.idle) for p in eng.phases] == [("Gmain", 1, False), ("Greg", 2, False), ("Dmain", 1, False), ("Dreg", 2, False)]
    bg, bd = [p.detach().clone() for p in eng.G.parameters()], [p.detach().clone() for p in eng.D.parameters()]
    for _ in range(2):
        eng.train_iteration(torch.rand(8, 3, 64, 64, device=dev) * 2 - 1, None)
    _finite_and_moved(eng, bg, bd)
    assert float(eng.loss.gen_regs[0].pl_mean) > 0
    eng.close()


def test_sg2attent_schedule_bf16(dev):
    """configs/sg2attent.yaml at its own 32x32: attention at every G resolution and at D's 32 block, bf16 blocks, R1"""
    gk, dk = _sg2_kwargs(res=32, cb=2048, cm=128, attn_g=[32, 16, 8, 4], attn_d=[32], nfp=3)
    eng = trainers.StepEngine(dev, gen_kwargs=gk, disc_kwargs=dk, loss_arch_kwargs=dict(style_mixing_prob=0), dis_regs=[("r1", dict(r1_gamma=0.01))],
                              g_reg_interval=16, d_reg_interval=4, batch=8, batch_gpu=8, ema_kimg=0.02)
    for m in list(eng.G.modules()) + list(eng.D.modules()):
        if type(m).__name__ == "Attention":
            torch.nn.init.constant_(m.gamma, 0.5)            # gamma starts at 0 (identity); open the branch so its gradients are exercised
    bg, bd = [p.detach().clone() for p in eng.G.parameters()], [p.detach().clone() for p in eng.D.parameters()]
    eng.train_iteration(torch.rand(8, 3, 32, 32, device=dev) * 2 - 1, None)
    _finite_and_moved(eng, bg, bd)
    att = [m for m in eng.D.modules() if type(m).__name__ == "Attention"][0]
    assert float(att.theta.weight.grad.abs().sum()) > 0      # the last D phase of the iteration was Dreg: R1 reached the attention weights
    eng.close()


def test_big_gan_128_step(dev):
    """configs/big_gan.yaml at its real architecture (128x128, ch 64, D attention at 32, 10 classes, hinge, n_dis 4), batch 8"""
    opt = ("adam", dict(lr=2e-4, betas=[0.0, 0.999], eps=1e-8))
    eng = trainers.StepEngine(dev, generator="big_gan", discriminator="big_gan", loss_arch="base", loss="hinge", loss_arch_kwargs=dict(),
                              gen_kwargs=dict(c_dim=10, img_resolution=128, G_shared=False, G_attn="0", G_init="N02", n_classes=10),
                              disc_kwargs=dict(c_dim=10, img_resolution=128, D_attn="32", D_init="N02", n_classes=10), optim_gen=opt, optim_disc=opt,
                              gen_regs=[], dis_regs=[], g_reg_interval=0, d_reg_interval=0, n_dis=4, batch=8, batch_gpu=8, ema_kimg=0.02)
    assert [(p.name, p.interval) for p in eng.phases] == [("Gboth", 4), ("Dboth", 1)]
    bg, bd = [p.detach().clone() for p in eng.G.parameters()], [p.detach().clone() for p in eng.D.parameters()]
    c = torch.nn.functional.one_hot(torch.arange(8, device=dev) % 10, 10).float()
    for _ in range(2):
        eng.train_iteration(torch.rand(8, 3, 128, 128, device=dev) * 2 - 1, c)
    _finite_and_moved(eng, bg, bd)
    eng.close()


def test_merged_discriminator_pass_equals_two_passes(dev):
    """Dmain runs D once over [generated; real] (losses_base._pass_d_adv): the interleaved batch must give the logits and parameter gradients
    of the two separate forwards -- in particular the minibatch-std groups must be the ones each half forms alone -- and an engine iteration
    with the merge on must land where the two-pass iteration lands (tolerance: summation order of the weight gradients)."""
    from style_big_gan_amd.train_parts import discriminators, losses_base
    _, dk = _sg2_kwargs(res=32)
    torch.manual_seed(0)
    D = discriminators.Discriminator(**dk).to(dev)
    assert D.batch_mergeable
    for n in (8, 12):
        a = torch.randn(n, 3, 32, 32, device=dev); b = torch.randn(n, 3, 32, 32, device=dev)
        order = D.merged_batch_order(n)
        assert order is not None and sorted(order) == list(range(2 * n))
        fwd = torch.tensor(order, device=dev); inv = torch.empty_like(fwd); inv[fwd] = torch.arange(2 * n, device=dev)
        la, lb = D(a, None), D(b, None)
        (torch.nn.functional.softplus(la).mean() + torch.nn.functional.softplus(-lb).mean()).backward()
        g_sep = [p.grad.clone() for p in D.parameters()]
        D.zero_grad(set_to_none=True)
        lm = D(torch.cat([a, b]).index_select(0, fwd), None).index_select(0, inv)
        (torch.nn.functional.softplus(lm[:n]).mean() + torch.nn.functional.softplus(-lm[n:]).mean()).backward()
        assert float((lm[:n] - la).abs().max()) <= 2e-3 * float(la.abs().max() + 1) and float((lm[n:] - lb).abs().max()) <= 2e-3 * float(lb.abs().max() + 1)
        for gs, p in zip(g_sep, D.parameters()):
            assert float((p.grad - gs).abs().max()) <= 2e-2 * float(gs.abs().max()) + 1e-6
        D.zero_grad(set_to_none=True)
    assert D.merged_batch_order(6) is None                              # groups of four do not tile six samples: the halves cannot be kept apart

    gk, dk = _sg2_kwargs()
    kw = dict(gen_kwargs=gk, disc_kwargs=dk, loss_arch_kwargs=dict(style_mixing_prob=0), dis_regs=[("r1", dict(r1_gamma=0.1))], g_reg_interval=4,
              d_reg_interval=4, batch=8, batch_gpu=8, ema_kimg=0.05)
    real = torch.rand(8, 3, 32, 32, device=dev) * 2 - 1
    weights = []
    was = losses_base.merge_d_passes
    try:
        for merge in (True, False):
            losses_base.merge_d_passes = merge
            eng = trainers.StepEngine(dev, seed=5, **kw)
            eng.batch_idx = 1                                            # an iteration without the regulariser phases: Gmain + Dmain
            z = torch.randn(len(eng.phases) * 8, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(9))
            torch.manual_seed(77)
            eng.train_iteration(real, None, all_gen_z=z)
            weights.append([p.detach().clone() for p in eng.D.parameters()])
    finally:
        losses_base.merge_d_passes = was
    lr = 0.002
    for pa, pb in zip(*weights):      # Adam's first step moves every element by ~lr * sign(g): elements whose gradient is ~0 may differ by a full step
        d = (pa - pb).abs()
        assert float((d > 0.5 * lr).float().mean()) < 0.02, float((d > 0.5 * lr).float().mean())


def test_rounds_in_one_pass_equal_separate_rounds(dev):
    """StepEngine evaluates the accumulation rounds of a split phase in ONE pass over [round 0; round 1; ...] (StepEngine._rounds_in_one_pass):
    the discriminator must form the minibatch-std groups of the separate rounds for any number of segments, and an iteration (Gmain, Dmain,
    Dreg with R1) must land where the round-by-round iteration lands, including the mapping network's running average, which advances once
    per round.  (The reference's own two-round schedule is met by tests/test_reference_vectors_gpu.py::test_step_engine_against_reference_schedule,
    whose r1 fixture has batch 4 = 2 x batch_gpu 2 and runs through this path.)"""
    from style_big_gan_amd.train_parts import discriminators, generators
    gk, dk = _sg2_kwargs(res=32)
    torch.manual_seed(0)
    D = discriminators.Discriminator(**dk).to(dev)
    n = 8
    for S in (3, 4):
        parts = [torch.randn(n, 3, 32, 32, device=dev) for _ in range(S)]
        order = D.merged_batch_order(n, S)
        assert order is not None and sorted(order) == list(range(S * n))
        fwd = torch.tensor(order, device=dev); inv = torch.empty_like(fwd); inv[fwd] = torch.arange(S * n, device=dev)
        with torch.no_grad():
            sep = torch.cat([D(x, None) for x in parts])
            one = D(torch.cat(parts).index_select(0, fwd), None).index_select(0, inv)
        assert float((one - sep).abs().max()) <= 2e-3 * float(sep.abs().max() + 1)
    assert D.merged_batch_order(6, 3) is None
    # a pass too large for the op layer's 2 GiB tensors runs its highest-resolution blocks over slices (Discriminator.pass_plan): same logits
    x = torch.randn(16, 3, 32, 32, device=dev)
    assert D.pass_plan(16) == (0, 16)
    with torch.no_grad():
        whole = D(x, None)
    D.pass_bytes_limit = 8 * D.peak_activation_bytes() + 1
    try:
        assert D.pass_plan(16) == (2, 8) and D.pass_plan(8) == (0, 8)        # b32 and b16 in two slices of 8, b8 and the epilogue over all 16
        xg = x.clone().requires_grad_(True)
        sliced = D(xg, None)
        assert float((sliced - whole).abs().max()) <= 2e-3 * float(whole.abs().max() + 1)
        sliced.sum().backward()
        assert xg.grad is not None and bool(torch.isfinite(xg.grad).all()) and float(xg.grad.abs().sum()) > 0
    finally:
        del D.pass_bytes_limit                                         # back to the class default
    assert D.peak_activation_bytes() == 32 * 33 * 33 * 2 and generators.Generator(**gk).peak_activation_bytes() == 32 * 33 * 33 * 2

    real = torch.rand(16, 3, 32, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(21)) * 2 - 1
    was = trainers.merge_rounds
    # fp32 networks: the two schedules are the same sums in another order.  Two bounds per phase: the relative L2 error over ALL of the phase's
    # gradients (2e-3: the R1 phase differentiates twice through convolutions evaluated as split-bf16 products), and per tensor 10x that of its
    # largest element -- one tensor may carry a KINK FLIP: the small layers' split-K factor depends on the batch, so a pre-activation within an
    # ulp of zero can land on the other side of the leaky ReLU in the other schedule and move the gradients that pass through it by a finite
    # amount.  (Seen when the styles' fp32 summation order changed: with the per-layer library products written into the style bank's views the
    # schedules agree to 3e-7, with the bank's own order one activation of the second round flips and a [64, 64, 3, 3] gradient moves by 2.8e-3
    # of its largest element; scratch/bank_diag3.py.)  A wrong group order, gain or accumulation is an O(1) error in every tensor.
    # bf16 from 4x4 up: the reorderings also move roundings of the 16-bit activations of two networks in a row
    for nfp, tol in ((0, 2e-3), (8, 5e-2)):
        gk, dk = _sg2_kwargs(res=32, nfp=nfp)
        kw = dict(gen_kwargs=gk, disc_kwargs=dk, loss_arch_kwargs=dict(style_mixing_prob=0), dis_regs=[("r1", dict(r1_gamma=0.1))], g_reg_interval=4,
                  d_reg_interval=4, batch=16, batch_gpu=8, ema_kimg=0.05)
        out = []
        try:
            for merge in (True, False):
                trainers.merge_rounds = merge
                eng = trainers.StepEngine(dev, seed=5, **kw)
                for m in eng.G.synthesis.modules():                      # the per-layer noise is drawn per pass: one draw of 16 vs two of 8
                    if hasattr(m, 'use_noise'):
                        m.use_noise = False
                assert [eng._rounds_in_one_pass(p.name, 2) for p in eng.phases] == [merge, False, merge, merge]      # Gmain, (idle) Greg, Dmain, Dreg
                z = torch.randn(len(eng.phases) * 16, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(9))
                grads = {}
                for ph in eng.phases:                                    # the gradients each phase hands to its optimizer step
                    def step(ph=ph, inner=ph.opt.step):
                        grads[ph.name] = [p.grad.detach().clone() for p in ph.module.parameters() if p.grad is not None]
                        return inner()
                    ph.opt.step = step
                eng.train_iteration(real, None, all_gen_z=z)             # iteration 0: every phase runs
                out.append((grads, eng.G.mapping.w_avg.clone()))
                eng.close()
        finally:
            trainers.merge_rounds = was
        (ga, avg_a), (gb, avg_b) = out
        assert float((avg_a - avg_b).abs().max()) <= 1e-5 * float(avg_b.abs().max() + 1e-3)
        assert set(ga) == set(gb) >= {'Gmain', 'Dmain', 'Dreg'}     # (main and lazy-regulariser slots share an optimizer: a step records under both names)
        for name in gb:
            assert len(ga[name]) == len(gb[name]) > 0
            num = sum(float((a.double() - b.double()).square().sum()) for a, b in zip(ga[name], gb[name])) ** 0.5
            den = sum(float(b.double().square().sum()) for b in gb[name]) ** 0.5
            # (the generator's phase runs before any optimizer step; the discriminator's phases see the generator AFTER its first Adam step, which turns
            # a flipped gradient element into a full learning-rate step of that weight: their bound is the looser one)
            assert num <= (tol if name in ('Gmain', 'Greg') else 5 * tol) * den, (nfp, name, num, den)
            for a, b in zip(ga[name], gb[name]):
                # (absolute floor: R1's gradient w.r.t. a bias of a piecewise-linear network is zero up to rounding -- such tensors are ~1e-7 of noise)
                assert float((a - b).abs().max()) <= 10 * tol * float(b.abs().max()) + 1e-6, (nfp, name, tuple(b.shape), float((a - b).abs().max()), float(b.abs().max()))

    gk2, dk2 = _sg2_kwargs(res=32, attn_g=(16,))                         # power iterations in G's attention block: rounds stay apart
    eng = trainers.StepEngine(dev, seed=5, gen_kwargs=gk2, disc_kwargs=dk2, loss_arch_kwargs=dict(style_mixing_prob=0), batch=16, batch_gpu=8)
    assert not any(eng._rounds_in_one_pass(p.name, 2) for p in eng.phases)
    eng.close()


def test_cat0_is_cat_with_views_backward(dev):
    """torch_utils.misc.cat0 (slice copies into one allocation; what the merged discriminator pass concatenates its batches with) == torch.cat,
    for channel-minor and planar inputs, first and second derivatives."""
    from style_big_gan_amd.torch_utils import misc
    torch.manual_seed(0)
    for fmt in (torch.contiguous_format, torch.channels_last):
        a = torch.randn(3, 5, 8, 8, device=dev).to(memory_format=fmt).requires_grad_(True)
        b = torch.randn(2, 5, 8, 8, device=dev).to(memory_format=fmt).requires_grad_(True)
        y, y_ref = misc.cat0([a, b]), torch.cat([a, b])
        assert torch.equal(y, y_ref) and y.is_contiguous(memory_format=fmt)
        wgt = torch.randn_like(y_ref)
        ga, gb = torch.autograd.grad((y.square() * wgt).sum(), [a, b], create_graph=True)
        ra, rb = torch.autograd.grad((y_ref.square() * wgt).sum(), [a, b], create_graph=True)
        assert torch.equal(ga, ra) and torch.equal(gb, rb)
        (g2,) = torch.autograd.grad(ga.sum() + gb.sum(), [a])
        (r2,) = torch.autograd.grad(ra.sum() + rb.sum(), [a])
        assert torch.equal(g2, r2)
    assert misc.cat0([a]) is a
    mixed = misc.cat0([a, b.to(torch.bfloat16)])          # dtypes differ: plain torch.cat semantics (type promotion)
    assert mixed.dtype == torch.float32
