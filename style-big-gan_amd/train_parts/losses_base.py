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

Passes may be wider than the reference's (same values, same parameter gradients up to summation order; DESIGN.md "passes per round"):
``Dmain`` runs the discriminator once over [generated; real], and the trainer may hand several accumulation rounds to one call
(``segments`` > 1: inputs are ``[round 0; round 1; ...]``).  Every discriminator pass over more than one such segment interleaves the samples
so that the minibatch-std layer forms exactly the groups it forms on each segment alone (``Discriminator.merged_batch_order``).

``sync`` gates the data-parallel exchange as in the reference: only the last pass of the last accumulation round of a phase
synchronises; the regularisers receive ``sync = (last regulariser of the list)`` (:94, :109).
"""
import torch

from .. import utils
from ..torch_utils import misc, training_stats
from ..torch_utils.ops import conv_bias_act as _cba
from ..torch_utils.ops import fromrgb as _fromrgb
from ..torch_utils.ops import modconv as _modconv
from ..utils import EasyDict
from .losses import losses
from .regularizations import R1reg, discriminator_regs, generator_regs

losses_arch = utils.ClassRegistry()

_scope = torch.autograd.profiler.record_function

import os as _os
merge_d_passes = _os.environ.get('SBG_MERGE_D', '1') != '0'      # Dmain: one discriminator pass over [generated; real] (see _pass_d_adv)
fuse_d_pairs = _os.environ.get('SBG_FUSE_D_PAIRS', '1') != '0'   # first-order D passes: conv0 + the low-pass of conv1 as one Function (ops/conv_bias_act.py)

#             phase      passes, in execution order
_PROGRAMS = {'Gmain': ('g_adv',), 'Greg': ('g_reg',), 'Gboth': ('g_adv', 'g_reg'),
             'Dmain': ('d_adv',), 'Dreg': ('d_reg',), 'Dboth': ('d_adv', 'd_reg')}


class _Round:
    """inputs of one accumulation round + what its passes hand to each other"""
    __slots__ = ('real_img', 'real_c', 'gen_z', 'gen_c', 'sync', 'gain', 'real_logits', 'real_img_tmp', 'segments')

    def __init__(self, real_img, real_c, gen_z, gen_c, sync, gain, segments=1):
        self.real_img, self.real_c, self.gen_z, self.gen_c, self.sync, self.gain = real_img, real_c, gen_z, gen_c, sync, gain
        self.real_logits = self.real_img_tmp = None
        self.segments = segments      # accumulation rounds held by this record (inputs are [round 0; round 1; ...])


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

    def rounds_mergeable(self, phase, n, rounds):
        """may `rounds` accumulation rounds of `n` samples each be handed to ONE accumulate_gradients call (segments = rounds)?  Only the split
        phases (one pass per round), only regularisers without state from round to round (R1; the path-length regulariser advances its running
        mean per round), and only when the discriminator can keep the rounds' minibatch-std groups apart within its size limit."""
        passes = self.program(phase)
        if len(passes) != 1 or passes[0] == 'g_reg':
            return False
        if passes[0] == 'd_reg' and not all(type(r) is R1reg for r in self.dis_regs):
            return False
        return self._d_order(n, rounds, torch.device(self.device)) is not None

    def accumulate_gradients(self, phase, real_img, real_c, gen_z, gen_c, sync, gain, segments=1):
        passes = self.program(phase)
        rnd = _Round(real_img, real_c, gen_z, gen_c, sync, gain, segments)
        first_order_d = 'd_reg' not in passes
        fromrgb_was, _fromrgb.enabled = _fromrgb.enabled, first_order_d
        pairs_was, _cba.first_order = _cba.first_order, first_order_d and fuse_d_pairs
        # generator regularisers (path length) differentiate G twice: their passes take the arbitrarily differentiable composition, every other pass the
        # first-order fused synthesis layers (ops/modconv.py) -- a phase that holds both kinds ('Gboth') is conservative
        modconv_was, _modconv.enabled = _modconv.enabled, _modconv.enabled and 'g_reg' not in passes
        try:
            for k, name in enumerate(passes):
                getattr(self, '_pass_' + name)(rnd, closes_round=(k == len(passes) - 1), reg_follows=('d_reg' in passes[k + 1:]))
        finally:
            _fromrgb.enabled = fromrgb_was
            _cba.first_order = pairs_was
            _modconv.enabled = modconv_was

    # -- the four passes ---------------------------------------------------------------------------------------------------
    def _pass_g_adv(self, rnd, closes_round, reg_follows):
        """generator's adversarial loss on D(G(z)) (reference :50-61)"""
        with _scope('Gmain_forward'):
            gen_img = self.run_G(rnd.gen_z, rnd.gen_c, sync=(rnd.sync and closes_round))
            if rnd.segments > 1:
                order = self._d_order(gen_img.shape[0] // rnd.segments, rnd.segments, gen_img.device)
                (gen_logits,) = self._run_D_ordered([gen_img], [rnd.gen_c], order, sync=False)
            else:
                gen_logits = self.run_D(gen_img, rnd.gen_c, sync=False)
            _report_scores('fake', gen_logits)
            loss_Gmain = self._loss_per_round(rnd.segments, gen_logits, None)
            training_stats.report('Loss/G/loss', loss_Gmain)
        with _scope('Gmain_backward'):
            loss_Gmain.mean().mul(rnd.gain).backward()

    def _pass_g_reg(self, rnd, closes_round, reg_follows):
        for i, reg in enumerate(self.gen_regs):
            reg.calc_reg(self, rnd.real_img, rnd.real_c, rnd.gen_z, rnd.gen_c, sync=(i == len(self.gen_regs) - 1), gain=rnd.gain)

    def _pass_d_adv(self, rnd, closes_round, reg_follows):
        """discriminator's loss on generated and real images; one backward covers both forwards (reference :64-81)"""
        with _scope('Dgen_forward'):
            gen_img = self.run_G(rnd.gen_z, rnd.gen_c, sync=False)          # G's parameters do not require grad in a D phase: no graph
            n = gen_img.shape[0] // rnd.segments
            both = None if reg_follows else self._both_halves_order(gen_img, rnd, n)
            if both is not None:
                gen_logits, rnd.real_logits = self._run_D_ordered([gen_img, rnd.real_img.detach()], [rnd.gen_c, rnd.real_c], both,
                                                                  sync=(rnd.sync and closes_round))
            elif rnd.segments > 1:
                order = self._d_order(n, rnd.segments, gen_img.device)
                (gen_logits,) = self._run_D_ordered([gen_img], [rnd.gen_c], order, sync=False)
                (rnd.real_logits,) = self._run_D_ordered([rnd.real_img.detach()], [rnd.real_c], order, sync=(rnd.sync and closes_round))
            else:
                gen_logits = self.run_D(gen_img, rnd.gen_c, sync=False)      # exchanged together with the real pass below
                rnd.real_img_tmp = rnd.real_img.detach().requires_grad_(reg_follows)
                rnd.real_logits = self.run_D(rnd.real_img_tmp, rnd.real_c, sync=(rnd.sync and closes_round))
            _report_scores('fake', gen_logits)
            _report_scores('real', rnd.real_logits)
            loss_Dgen = self._loss_per_round(rnd.segments, rnd.real_logits, gen_logits)
        with _scope('Dgen_backward'):      # a regulariser of the same phase differentiates real_logits again: keep the graph for it
            loss_Dgen.mean().mul(rnd.gain).backward(retain_graph=reg_follows)

    def _loss_per_round(self, segments, pred_real, pred_fake):
        """the loss of every accumulation round held by the logits (a scalar for one round, else [segments]): its mean times `segments` is the
        sum over rounds the reference accumulates, and the reported statistic keeps one entry per round"""
        if segments == 1:
            return self.loss.calc_loss(pred_real, pred_fake)
        fake = pred_fake.chunk(segments) if pred_fake is not None else [None] * segments
        return torch.stack([self.loss.calc_loss(r, f) for r, f in zip(pred_real.chunk(segments), fake)])

    # One discriminator pass over several batches of n samples ("segments": the generated and the real half of a round -- the reference runs D
    # twice, :66-77 -- and / or several accumulation rounds).  Same function values and the same parameter gradients up to summation order: the
    # augmentation pipe still sees generated images, then reals (its random draws keep that order), and the samples are interleaved so that the
    # minibatch-std layer forms exactly the groups it forms on each segment alone (Discriminator.merged_batch_order).  Only for discriminators
    # that declare `batch_mergeable` (no state carried across forward calls) and while no tensor of the pass reaches the op layer's 2 GiB
    # limit (Discriminator.pass_plan: the highest-resolution blocks may run over slices of the pass to stay below it).  Worth it because the fixed cost of a pass -- ~600 launches, the latency-bound 4x4 ... 32x32 layers -- is paid once: see
    # DESIGN.md, "passes per round".
    def _d_order(self, n, segments, device):
        """(fwd, inv) index tensors for a pass over `segments` batches of n samples; () = natural order; None = not possible"""
        if segments == 1:
            return ()
        d = getattr(self.D, 'module', self.D)
        if not getattr(d, 'batch_mergeable', False):
            return None
        # cached on the module instance (like its pass plans), keyed by everything the order depends on: a module-level dict keyed by id(d)
        # would hand a new discriminator that re-uses a collected one's id a stale order (other group size / pass limit)
        cache = d.__dict__.setdefault('_merged_order_cache', {})
        key = (n, segments, str(device), getattr(d, 'pass_bytes_limit', None),
               getattr(getattr(getattr(d, 'b4', None), 'mbstd', None), 'group_size', None))
        hit = cache.get(key)
        if hit is None:
            fits = d.pass_plan(segments * n) is not None if hasattr(d, 'pass_plan') else True
            order = d.merged_batch_order(n, segments) if fits else None
            if order is None:
                hit = (None, None)
            else:
                fwd = torch.tensor(order, dtype=torch.int64, device=device)
                inv = torch.empty_like(fwd)
                inv[fwd] = torch.arange(segments * n, device=device)
                hit = (fwd, inv)
            cache[key] = hit
        return hit if hit[0] is not None else None

    def _both_halves_order(self, gen_img, rnd, n):
        """order for ONE pass over [generated; real] of a Dmain round (times the rounds held by `rnd`), or None: two passes"""
        if not merge_d_passes or gen_img.shape != rnd.real_img.shape:
            return None
        if (rnd.gen_c is None) != (rnd.real_c is None) or (rnd.gen_c is not None and rnd.gen_c.shape != rnd.real_c.shape):
            return None
        return self._d_order(n, 2 * rnd.segments, gen_img.device)

    def _run_D_ordered(self, imgs, cs, order, sync):
        """logits of every batch in `imgs` from one discriminator pass over their concatenation arranged by `order` (see _d_order)"""
        if self.augment_pipe is not None:
            imgs = [self.augment_pipe(img) for img in imgs]
        x = misc.cat0([img.to(imgs[0].dtype) for img in imgs])
        c = None if cs[0] is None else (cs[0] if len(cs) == 1 else torch.cat(cs))
        if order:
            x = x.index_select(0, order[0])
            c = c.index_select(0, order[0]) if c is not None else None
        with misc.ddp_sync(self.D, sync):
            logits = self.D(x, c)
        if order:
            logits = logits.index_select(0, order[1])
        return logits.split([img.shape[0] for img in imgs])

    def _pass_d_reg(self, rnd, closes_round, reg_follows):
        if rnd.real_logits is None:          # no adversarial pass in this phase: the regularisers' shared forward on the reals (:100-105)
            with _scope('Dreg_forward'):
                rnd.real_img_tmp = rnd.real_img.detach().requires_grad_(True)
                if rnd.segments > 1:
                    order = self._d_order(rnd.real_img.shape[0] // rnd.segments, rnd.segments, rnd.real_img.device)
                    (rnd.real_logits,) = self._run_D_ordered([rnd.real_img_tmp], [rnd.real_c], order, sync=rnd.sync)
                else:
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
