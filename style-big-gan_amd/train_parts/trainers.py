"""Training step engine (host code) -- the G+D step the headline metric counts.

Counterpart of the hot loop of the reference's ``train_parts/trainers.py``: phase construction with lazy
regularisation (``setup_training_phases`` :601-633: Gmain / Greg / Dmain / Dreg, lr and betas rescaled by
``mb_ratio = interval / (interval + 1)``), the per-iteration body of ``training_loop`` (:711-765: per-phase
``zero_grad`` -> accumulation rounds of ``loss.accumulate_gradients`` -> ``nan_to_num`` of the gradients ->
``opt.step()``), the generator EMA (:752-761) and the data-parallel wiring of ``distrib_acrros_gpu`` (:587-597) /
``SG2Trainer`` (:883-893: mapping, synthesis and D are separate data-parallel modules so they can be synchronised
independently).  Logging, snapshots, metrics and dataset plumbing are out of scope of this engine.

MI355X specifics: gradients live in flat fp32 buckets (``parallel.GradReducer``) that are all-reduced over RCCL as soon
as their last gradient of the phase's final accumulation round has been written, overlapping xGMI traffic with the rest
of backward; ``nan_to_num`` and Adam run over the flat buckets / foreach lists instead of per-parameter launches.
"""
import copy

import numpy as np
import torch

from .. import utils
from ..parallel import GradReducer
from ..torch_utils import misc, training_stats
from ..utils import EasyDict
from .discriminators import discriminators
from .generators import generators
from .losses_base import losses_arch
from .optimizers import optimizers

trainers = utils.ClassRegistry()


def lazy_reg_opt_kwargs(opt_kwargs, interval):
    """lr and betas of a phase that also carries a regulariser applied every `interval` iterations (reference :619-623)"""
    mb_ratio = interval / (interval + 1)
    out = dict(opt_kwargs)
    out['lr'] = opt_kwargs['lr'] * mb_ratio
    out['betas'] = [beta ** mb_ratio for beta in opt_kwargs['betas']]
    return out


@trainers.add_to_registry("step_engine")
class StepEngine:
    """Owns G, D, G_ema, the optimizers and the phase schedule; `train_iteration` is one G+D step on this rank.

    batch       -- images per iteration on THIS rank (the reference's ``batch // num_gpus``)
    batch_gpu   -- micro-batch per accumulation round
    """

    def __init__(self, device, generator='sg2_classic', discriminator='sg2_classic', gen_kwargs=None, disc_kwargs=None,
                 loss_arch='sg2', loss='softplus', loss_arch_kwargs=None, gen_regs=(), dis_regs=(('r1', dict(r1_gamma=10.)),),
                 optim_gen=('adam', dict(lr=0.0025, betas=[0, 0.99], eps=1e-8)), optim_disc=('adam', dict(lr=0.0025, betas=[0, 0.99], eps=1e-8)),
                 g_reg_interval=16, d_reg_interval=4, batch=64, batch_gpu=32, ema_kimg=10., ema_rampup=None, use_ema=True,
                 world_size=1, rank=0, process_group=None, seed=0):
        self.device = torch.device(device)
        self.rank, self.world_size = rank, world_size
        self.batch, self.batch_gpu = batch, batch_gpu
        assert batch % batch_gpu == 0
        self.ema_kimg, self.ema_rampup, self.use_ema = ema_kimg, ema_rampup, use_ema
        self.cur_nimg = 0
        self.batch_idx = 0

        torch.manual_seed(seed * max(world_size, 1) + rank)     # reference :507-508
        self.G = generators[generator](**(gen_kwargs or {})).train().requires_grad_(False).to(self.device)
        self.D = discriminators[discriminator](**(disc_kwargs or {})).train().requires_grad_(False).to(self.device)
        self.G_ema = copy.deepcopy(self.G).eval() if use_ema else None
        self.z_dim = self.G.z_dim
        self.c_dim = self.G.c_dim or 0

        # data-parallel wrappers (broadcast rank 0's weights, own the gradient buckets)
        self.sg2 = (loss_arch == 'sg2')
        dp = dict(world_size=world_size, process_group=process_group)
        if self.sg2:
            self.dp_modules = dict(G_mapping=GradReducer(self.G.mapping, **dp), G_synthesis=GradReducer(self.G.synthesis, **dp),
                                   D=GradReducer(self.D, **dp))
            la = dict(G_mapping=self.dp_modules['G_mapping'], G_synthesis=self.dp_modules['G_synthesis'])
        else:
            self.dp_modules = dict(G=GradReducer(self.G, **dp), D=GradReducer(self.D, **dp))
            la = dict(G=self.dp_modules['G'])
        la.update(loss_arch_kwargs or {})
        self.loss = losses_arch[loss_arch](device=self.device, gen_regs=list(gen_regs), dis_regs=list(dis_regs),
                                           D=self.dp_modules['D'], loss=loss, **la)

        # phases (reference :601-633)
        self.phases = []
        g_reducers = [self.dp_modules[k] for k in self.dp_modules if k.startswith('G')]
        for name, module, reducers, (opt_name, opt_kwargs), regs, interval in [
                ('G', self.G, g_reducers, optim_gen, gen_regs, g_reg_interval),
                ('D', self.D, [self.dp_modules['D']], optim_disc, dis_regs, d_reg_interval)]:
            opt_kwargs = dict(opt_kwargs)
            if len(regs) == 0 or interval == 0 or interval is None:
                opt = optimizers[opt_name](params=module.parameters(), **opt_kwargs)
                self.phases.append(EasyDict(name=name + ('both' if len(regs) else 'main'), module=module, reducers=reducers, opt=opt, interval=1))
            else:
                opt = optimizers[opt_name](module.parameters(), **lazy_reg_opt_kwargs(opt_kwargs, interval))
                self.phases.append(EasyDict(name=name + 'main', module=module, reducers=reducers, opt=opt, interval=1))
                self.phases.append(EasyDict(name=name + 'reg', module=module, reducers=reducers, opt=opt, interval=interval))

    # ------------------------------------------------------------------------------------------------------------
    def train_iteration(self, real_img, real_c, all_gen_z=None, all_gen_c=None):
        """real_img: [batch, C, H, W] float in [-1, 1] on the device; real_c: [batch, c_dim].  One pass over the phases."""
        assert real_img.shape[0] == self.batch
        n_phase = len(self.phases)
        if all_gen_z is None:
            all_gen_z = torch.randn([n_phase * self.batch, self.z_dim], device=self.device)
        if all_gen_c is None:
            all_gen_c = real_c.repeat(n_phase, 1) if real_c is not None and real_c.numel() else torch.zeros([n_phase * self.batch, self.c_dim], device=self.device)
        reals = real_img.split(self.batch_gpu)
        real_cs = (real_c if real_c is not None else torch.zeros([self.batch, self.c_dim], device=self.device)).split(self.batch_gpu)
        zs = [z.split(self.batch_gpu) for z in all_gen_z.split(self.batch)]
        cs = [c.split(self.batch_gpu) for c in all_gen_c.split(self.batch)]
        rounds = self.batch // self.batch_gpu

        for phase, phase_z, phase_c in zip(self.phases, zs, cs):
            if self.batch_idx % phase.interval != 0:
                continue
            with torch.autograd.profiler.record_function(phase.name):
                for r in phase.reducers:
                    r.zero_grad()
                phase.module.requires_grad_(True)
                for round_idx, (img, c, z, gc) in enumerate(zip(reals, real_cs, phase_z, phase_c)):
                    sync = (round_idx == rounds - 1)
                    self.loss.accumulate_gradients(phase=phase.name, real_img=img, real_c=c, gen_z=z, gen_c=gc, sync=sync, gain=phase.interval)
                phase.module.requires_grad_(False)
            with torch.autograd.profiler.record_function(phase.name + '_opt'):
                for r in phase.reducers:
                    r.finish()              # wait for the all-reduce, average, nan_to_num (reference :745-747)
                phase.opt.step()

        if self.G_ema is not None:
            with torch.autograd.profiler.record_function('Gema'):
                self.update_ema()
        self.cur_nimg += self.batch * self.world_size
        self.batch_idx += 1

    @torch.no_grad()
    def update_ema(self):
        """G_ema <- lerp(G, G_ema, beta), buffers copied (reference :752-761)"""
        global_batch = self.batch * self.world_size
        ema_nimg = self.ema_kimg * 1000
        if self.ema_rampup is not None:
            ema_nimg = min(ema_nimg, self.cur_nimg * self.ema_rampup)
        beta = 0.5 ** (global_batch / max(ema_nimg, 1e-8))
        p_ema, p = list(self.G_ema.parameters()), list(self.G.parameters())
        torch._foreach_lerp_(p_ema, p, 1.0 - beta)      # p_ema + (p - p_ema) * (1 - beta) == p.lerp(p_ema, beta)
        for b_ema, b in zip(self.G_ema.buffers(), self.G.buffers()):
            b_ema.copy_(b)
