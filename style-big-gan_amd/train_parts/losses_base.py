"""Loss orchestration (host code): which networks run forward and backward in each training phase.

Interface of the reference's ``train_parts/losses_base.py``: registry ``losses_arch`` with ``'base'`` (:113 -- the reference's
constructor calls a mistyped ``super().__int__`` and cannot be instantiated; here it constructs) and ``'sg2'`` (:119-153,
separate mapping / synthesis modules, style mixing); ``run_G`` / ``run_D`` / ``run_Gws`` (what the regularisers call back
into) and ``accumulate_gradients(phase, real_img, real_c, gen_z, gen_c, sync, gain)`` (what the trainer calls once per
accumulation round).  Statistic names and profiler scope names are the reference's.

Organisation here: a phase name selects a short *program* of passes out of four -- generator adversarial, generator
regularisers, discriminator adversarial, discriminator regularisers (``_PROGRAMS``; reference :85-89) -- and every pass is a
method working on one ``_Round`` record, which also carries what later passes of the same round reuse (the discriminator's
output on the reals and the reals they were computed from).  What each pass computes follows reference :50-81 / :96-109.

Two deliberate scheduling differences, neither of which changes a parameter gradient:
* the reals require grad only when a discriminator regulariser of the SAME phase differentiates with respect to them.
  The reference marks them whenever any discriminator regulariser is configured (:71), so its plain ``Dmain`` back-propagates
  to the image (first-layer data gradient plus the augmentation pipe's backward) only to discard the result;
* discriminator regularisers differentiate D twice with respect to its input, every other phase is first order: the streaming
  fromRGB kernels (``torch_utils/ops/fromrgb.py``, first order only) are switched on for exactly those other phases.

``sync`` gates the data-parallel exchange as in the reference: only the last pass of the last accumulation round of a phase
synchronises; the regularisers receive ``sync = (last regulariser of the list)`` (:94, :109).
"""
import torch

from .. import utils
from ..torch_utils import misc, training_stats
from ..torch_utils.ops import fromrgb as _fromrgb
from ..utils import EasyDict
from .losses import losses
from .regularizations import discriminator_regs, generator_regs

losses_arch = utils.ClassRegistry()

_scope = torch.autograd.profiler.record_function

import os as _os
merge_d_passes = _os.environ.get('SBG_MERGE_D', '1') != '0'      # Dmain: one discriminator pass over [generated; real] (see _pass_d_adv)
_order_cache = {}

#             phase      passes, in execution order
_PROGRAMS = {'Gmain': ('g_adv',), 'Greg': ('g_reg',), 'Gboth': ('g_adv', 'g_reg'),
             'Dmain': ('d_adv',), 'Dreg': ('d_reg',), 'Dboth': ('d_adv', 'd_reg')}


class _Round:
    """inputs of one accumulation round + what its passes hand to each other"""
    __slots__ = ('real_img', 'real_c', 'gen_z', 'gen_c', 'sync', 'gain', 'real_logits', 'real_img_tmp')

    def __init__(self, real_img, real_c, gen_z, gen_c, sync, gain):
        self.real_img, self.real_c, self.gen_z, self.gen_c, self.sync, self.gain = real_img, real_c, gen_z, gen_c, sync, gain
        self.real_logits = self.real_img_tmp = None


def _report_scores(which, logits):
    training_stats.report('Loss/scores/' + which, logits)
    training_stats.report('Loss/signs/' + which, logits.sign())


class LossBase:
    def __init__(self, device, gen_regs, dis_regs, G, D, loss, augment_pipe=None):
        self.device = device
        self.G = G
        self.D = D
        self.augment_pipe = augment_pipe
        # None (not an empty list) when nothing is configured: the reference's attribute convention (:34-35)
        self.gen_regs = [generator_regs[name](**kwargs) for name, kwargs in gen_regs] or None
        self.dis_regs = [discriminator_regs[name](**kwargs) for name, kwargs in dis_regs] or None
        self.loss = losses[loss]()

    # -- network passes the regularisers call back into --------------------------------------------------------------------
    def run_G(self, z, c, sync):
        with misc.ddp_sync(self.G, sync):
            return self.G(z, c)

    def run_D(self, img, c, sync):
        if self.augment_pipe is not None:
            img = self.augment_pipe(img)
        with misc.ddp_sync(self.D, sync):
            return self.D(img, c)

    # -- the trainer's entry point -----------------------------------------------------------------------------------------
    def program(self, phase):
        """passes of `phase` that have something to do (a regulariser pass without regularisers is dropped, reference :88-89)"""
        assert phase in _PROGRAMS, phase
        configured = dict(g_adv=True, d_adv=True, g_reg=self.gen_regs is not None, d_reg=self.dis_regs is not None)
        return [p for p in _PROGRAMS[phase] if configured[p]]

    def accumulate_gradients(self, phase, real_img, real_c, gen_z, gen_c, sync, gain):
        passes = self.program(phase)
        rnd = _Round(real_img, real_c, gen_z, gen_c, sync, gain)
        first_order_d = 'd_reg' not in passes
        fromrgb_was, _fromrgb.enabled = _fromrgb.enabled, first_order_d
        try:
            for k, name in enumerate(passes):
                getattr(self, '_pass_' + name)(rnd, closes_round=(k == len(passes) - 1), reg_follows=('d_reg' in passes[k + 1:]))
        finally:
            _fromrgb.enabled = fromrgb_was

    # -- the four passes ---------------------------------------------------------------------------------------------------
    def _pass_g_adv(self, rnd, closes_round, reg_follows):
        """generator's adversarial loss on D(G(z)) (reference :50-61)"""
        with _scope('Gmain_forward'):
            gen_logits = self.run_D(self.run_G(rnd.gen_z, rnd.gen_c, sync=(rnd.sync and closes_round)), rnd.gen_c, sync=False)
            _report_scores('fake', gen_logits)
            loss_Gmain = self.loss.calc_loss(gen_logits, None)
            training_stats.report('Loss/G/loss', loss_Gmain)
        with _scope('Gmain_backward'):
            loss_Gmain.mul(rnd.gain).backward()

    def _pass_g_reg(self, rnd, closes_round, reg_follows):
        for i, reg in enumerate(self.gen_regs):
            reg.calc_reg(self, rnd.real_img, rnd.real_c, rnd.gen_z, rnd.gen_c, sync=(i == len(self.gen_regs) - 1), gain=rnd.gain)

    def _pass_d_adv(self, rnd, closes_round, reg_follows):
        """discriminator's loss on generated and real images; one backward covers both forwards (reference :64-81)"""
        with _scope('Dgen_forward'):
            gen_img = self.run_G(rnd.gen_z, rnd.gen_c, sync=False)          # G's parameters do not require grad in a D phase: no graph
            order = None if reg_follows else self._merged_order(gen_img, rnd)
            if order is not None:
                gen_logits, rnd.real_logits = self._run_D_merged(gen_img, rnd, order, sync=(rnd.sync and closes_round))
            else:
                gen_logits = self.run_D(gen_img, rnd.gen_c, sync=False)      # exchanged together with the real pass below
                rnd.real_img_tmp = rnd.real_img.detach().requires_grad_(reg_follows)
                rnd.real_logits = self.run_D(rnd.real_img_tmp, rnd.real_c, sync=(rnd.sync and closes_round))
            _report_scores('fake', gen_logits)
            _report_scores('real', rnd.real_logits)
            loss_Dgen = self.loss.calc_loss(rnd.real_logits, gen_logits)
        with _scope('Dgen_backward'):      # a regulariser of the same phase differentiates real_logits again: keep the graph for it
            loss_Dgen.mean().mul(rnd.gain).backward(retain_graph=reg_follows)

    # One discriminator pass over [generated; real] instead of two (the reference runs D twice, :66-77).  Same function values and the same
    # parameter gradients up to summation order: the augmentation pipe still sees the two halves in the reference's order (its random draws
    # are consumed identically), and the samples are interleaved so that the minibatch-std layer forms exactly the groups it forms on each
    # half alone (Discriminator.merged_batch_order).  Only for a plain Dmain (no regulariser differentiating the reals in this phase) and for
    # discriminators that declare `batch_mergeable` (no state carried across forward calls).  Worth it because the fixed cost of a pass --
    # ~600 launches, the latency-bound 4x4 ... 32x32 layers -- is paid once: see DESIGN.md, "passes per round".
    def _merged_order(self, gen_img, rnd):
        if not merge_d_passes:
            return None
        d = getattr(self.D, 'module', self.D)
        if not getattr(d, 'batch_mergeable', False) or gen_img.shape != rnd.real_img.shape:
            return None
        if (rnd.gen_c is None) != (rnd.real_c is None) or (rnd.gen_c is not None and rnd.gen_c.shape != rnd.real_c.shape):
            return None
        n = gen_img.shape[0]
        key = (id(d), n, gen_img.device)
        hit = _order_cache.get(key)
        if hit is None:
            order = d.merged_batch_order(n)
            if order is None:
                hit = (None, None)
            else:
                fwd = torch.tensor(order, dtype=torch.int64, device=gen_img.device)
                inv = torch.empty_like(fwd)
                inv[fwd] = torch.arange(2 * n, device=gen_img.device)
                hit = (fwd, inv)
            _order_cache[key] = hit
        return hit if hit[0] is not None else None

    def _run_D_merged(self, gen_img, rnd, order, sync):
        fwd, inv = order
        n = gen_img.shape[0]
        real = rnd.real_img.detach()
        if self.augment_pipe is not None:
            gen_img, real = self.augment_pipe(gen_img), self.augment_pipe(real)
        x = torch.cat([gen_img, real.to(gen_img.dtype)]).index_select(0, fwd)
        c = torch.cat([rnd.gen_c, rnd.real_c]).index_select(0, fwd) if rnd.gen_c is not None else None
        with misc.ddp_sync(self.D, sync):
            logits = self.D(x, c).index_select(0, inv)
        return logits[:n], logits[n:]

    def _pass_d_reg(self, rnd, closes_round, reg_follows):
        if rnd.real_logits is None:          # no adversarial pass in this phase: the regularisers' shared forward on the reals (:100-105)
            with _scope('Dreg_forward'):
                rnd.real_img_tmp = rnd.real_img.detach().requires_grad_(True)
                rnd.real_logits = self.run_D(rnd.real_img_tmp, rnd.real_c, sync=rnd.sync)
                _report_scores('real', rnd.real_logits)
        for i, reg in enumerate(self.dis_regs):
            reg.calc_reg(self, rnd.real_img, rnd.real_c, rnd.gen_z, rnd.gen_c, rnd.real_logits, rnd.real_img_tmp,
                         sync=(i == len(self.dis_regs) - 1), gain=rnd.gain)


@losses_arch.add_to_registry("base")
class BasicLoss(LossBase):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)


@losses_arch.add_to_registry("sg2")
class SG2Loss(LossBase):
    """StyleGAN2 generator = mapping + synthesis as two data-parallel modules, with style mixing between two mapped latents
    (reference :119-153)."""

    def __init__(self, G_mapping=None, G_synthesis=None, style_mixing_prob=0.9, **kwargs):
        assert G_mapping is not None and G_synthesis is not None
        super().__init__(G=EasyDict(G_mapping=G_mapping, G_synthesis=G_synthesis), **kwargs)
        self.style_mixing_prob = style_mixing_prob

    def _mix_styles(self, ws, z, c):
        """with probability `style_mixing_prob`, layers from a random cut onwards take the latents of a second z (:133-137).
        Cut and coin are drawn on the device as in the reference; slicing by the drawn cut reads it back (one host synchronisation
        per call -- the shipped configs set style_mixing_prob = 0)"""
        with _scope('style_mixing'):
            cutoff = torch.empty([], dtype=torch.int64, device=ws.device).random_(1, ws.shape[1])
            cutoff = torch.where(torch.rand([], device=ws.device) < self.style_mixing_prob, cutoff, torch.full_like(cutoff, ws.shape[1]))
            ws[:, cutoff:] = self.G.G_mapping(torch.randn_like(z), c, skip_w_avg_update=True)[:, cutoff:]
        return ws

    def run_Gws(self, z, c, sync):
        with misc.ddp_sync(self.G.G_mapping, sync):
            ws = self.G.G_mapping(z, c)
            if self.style_mixing_prob > 0:
                ws = self._mix_styles(ws, z, c)
        with misc.ddp_sync(self.G.G_synthesis, sync):
            return self.G.G_synthesis(ws), ws

    def run_G(self, z, c, sync):
        return self.run_Gws(z, c, sync)[0]
