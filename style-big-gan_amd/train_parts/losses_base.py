"""Loss orchestration: which networks run forward / backward in each training phase (host code).

Mirror of the reference's ``train_parts/losses_base.py``: ``LossBase`` (:28-109) with ``run_G`` / ``run_D`` /
``accumulate_gradients(phase, real_img, real_c, gen_z, gen_c, sync, gain)``, ``BasicLoss`` registered as ``'base'``
(:113 -- the reference's constructor calls a mistyped ``super().__int__`` and cannot be instantiated; here it forwards
to ``LossBase.__init__`` as intended) and ``SG2Loss`` registered as ``'sg2'`` (:119-153) with separate mapping /
synthesis modules and style mixing.  ``sync`` gates the data-parallel gradient all-reduce exactly as in the reference
(only the last accumulation round of a phase reduces, and not when a regulariser follows in the same phase).
"""
import torch

from .. import utils
from ..torch_utils import misc, training_stats
from ..torch_utils.ops import fromrgb as _fromrgb
from ..utils import EasyDict
from .losses import losses
from .regularizations import discriminator_regs, generator_regs

losses_arch = utils.ClassRegistry()


class LossBase:
    def __init__(self, device, gen_regs, dis_regs, G, D, loss, augment_pipe=None):
        self.device = device
        self.G = G
        self.D = D
        self.augment_pipe = augment_pipe
        self.gen_regs = [generator_regs[name](**args) for name, args in gen_regs] if len(gen_regs) > 0 else None
        self.dis_regs = [discriminator_regs[name](**args) for name, args in dis_regs] if len(dis_regs) > 0 else None
        self.loss = losses[loss]()

    def run_G(self, z, c, sync):
        with misc.ddp_sync(self.G, sync):
            return self.G(z, c)

    def run_D(self, img, c, sync):
        if self.augment_pipe is not None:
            img = self.augment_pipe(img)
        with misc.ddp_sync(self.D, sync):
            return self.D(img, c)

    def do_Gmain(self, real_img, real_c, gen_z, gen_c, sync, gain):
        with torch.autograd.profiler.record_function('Gmain_forward'):
            gen_img = self.run_G(gen_z, gen_c, sync=sync)
            gen_logits = self.run_D(gen_img, gen_c, sync=False)
            training_stats.report('Loss/scores/fake', gen_logits)
            training_stats.report('Loss/signs/fake', gen_logits.sign())
            loss_Gmain = self.loss.calc_loss(gen_logits, None)
            training_stats.report('Loss/G/loss', loss_Gmain)
        with torch.autograd.profiler.record_function('Gmain_backward'):
            loss_Gmain.mul(gain).backward()

    def do_Dmain(self, real_img, real_c, gen_z, gen_c, sync, gain, need_real_grad=None):
        with torch.autograd.profiler.record_function('Dgen_forward'):
            gen_img = self.run_G(gen_z, gen_c, sync=False)
            gen_logits = self.run_D(gen_img, gen_c, sync=False)     # synced by the real pass below
            training_stats.report('Loss/scores/fake', gen_logits)
            training_stats.report('Loss/signs/fake', gen_logits.sign())
            # The reference marks the reals as requiring grad whenever a discriminator regulariser is configured (:71), so a plain
            # 'Dmain' phase also back-propagates to the image (first-layer data gradient + the augmentation pipe's backward) and
            # discards the result.  Only a regulariser in the SAME phase ('Dboth') reads that graph; otherwise it is dead work and
            # is not scheduled here.  Parameter gradients are unaffected.
            if need_real_grad is None:
                need_real_grad = self.dis_regs is not None
            real_img_tmp = real_img.detach().requires_grad_(bool(need_real_grad))
            real_logits = self.run_D(real_img_tmp, real_c, sync=sync)
            training_stats.report('Loss/scores/real', real_logits)
            training_stats.report('Loss/signs/real', real_logits.sign())
            loss_Dgen = self.loss.calc_loss(real_logits, gen_logits)
        with torch.autograd.profiler.record_function('Dgen_backward'):
            loss_Dgen.mean().mul(gain).backward()
        return real_logits, real_img_tmp

    def accumulate_gradients(self, phase, real_img, real_c, gen_z, gen_c, sync, gain):
        assert phase in ['Gmain', 'Greg', 'Gboth', 'Dmain', 'Dreg', 'Dboth']
        do_Gmain = phase in ['Gmain', 'Gboth']
        do_Dmain = phase in ['Dmain', 'Dboth']
        do_Greg = phase in ['Greg', 'Gboth'] and self.gen_regs is not None
        do_Dreg = phase in ['Dreg', 'Dboth'] and self.dis_regs is not None
        # discriminator regularisers (R1, gradient penalty) differentiate D twice w.r.t. its input; every other phase is first order
        # and may use the streaming fromRGB kernels (torch_utils/ops/fromrgb.py)
        fromrgb_was = _fromrgb.enabled
        _fromrgb.enabled = not do_Dreg
        try:
            self._accumulate(phase, real_img, real_c, gen_z, gen_c, sync, gain, do_Gmain, do_Dmain, do_Greg, do_Dreg)
        finally:
            _fromrgb.enabled = fromrgb_was

    def _accumulate(self, phase, real_img, real_c, gen_z, gen_c, sync, gain, do_Gmain, do_Dmain, do_Greg, do_Dreg):

        if do_Gmain:
            self.do_Gmain(real_img, real_c, gen_z, gen_c, sync=(sync and not do_Greg), gain=gain)
        if do_Greg:
            for i, reg in enumerate(self.gen_regs):
                reg.calc_reg(self, real_img, real_c, gen_z, gen_c, sync=(i == len(self.gen_regs) - 1), gain=gain)

        real_logits = real_img_tmp = None
        if do_Dmain:
            real_logits, real_img_tmp = self.do_Dmain(real_img, real_c, gen_z, gen_c, sync=(sync and not do_Dreg), gain=gain, need_real_grad=do_Dreg)
        if do_Dreg:
            if not do_Dmain:
                with torch.autograd.profiler.record_function('Dreg_forward'):
                    real_img_tmp = real_img.detach().requires_grad_(True)
                    real_logits = self.run_D(real_img_tmp, real_c, sync=sync)
                    training_stats.report('Loss/scores/real', real_logits)
                    training_stats.report('Loss/signs/real', real_logits.sign())
            for i, reg in enumerate(self.dis_regs):
                reg.calc_reg(self, real_img, real_c, gen_z, gen_c, real_logits, real_img_tmp,
                             sync=(i == len(self.dis_regs) - 1), gain=gain)


@losses_arch.add_to_registry("base")
class BasicLoss(LossBase):
    def __init__(self, **args):
        super().__init__(**args)


@losses_arch.add_to_registry("sg2")
class SG2Loss(LossBase):
    def __init__(self, G_mapping=None, G_synthesis=None, style_mixing_prob=0.9, **args):
        assert G_mapping is not None and G_synthesis is not None
        args.update({'G': EasyDict(G_mapping=G_mapping, G_synthesis=G_synthesis)})
        super().__init__(**args)
        self.style_mixing_prob = style_mixing_prob

    def _map(self, z, c):
        ws = self.G.G_mapping(z, c)
        if self.style_mixing_prob > 0:
            with torch.autograd.profiler.record_function('style_mixing'):
                cutoff = torch.empty([], dtype=torch.int64, device=ws.device).random_(1, ws.shape[1])
                cutoff = torch.where(torch.rand([], device=ws.device) < self.style_mixing_prob, cutoff, torch.full_like(cutoff, ws.shape[1]))
                ws[:, cutoff:] = self.G.G_mapping(torch.randn_like(z), c, skip_w_avg_update=True)[:, cutoff:]
        return ws

    def run_Gws(self, z, c, sync):
        with misc.ddp_sync(self.G.G_mapping, sync):
            ws = self._map(z, c)
        with misc.ddp_sync(self.G.G_synthesis, sync):
            img = self.G.G_synthesis(ws)
        return img, ws

    def run_G(self, z, c, sync):
        return self.run_Gws(z, c, sync)[0]
