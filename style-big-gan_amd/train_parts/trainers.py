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
import os

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

merge_rounds = os.environ.get('SBG_MERGE_ROUNDS', '1') != '0'      # accumulation rounds of a phase in one pass (StepEngine._rounds_in_one_pass)


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
                 g_reg_interval=16, d_reg_interval=4, n_dis=1, batch=64, batch_gpu=32, ema_kimg=10., ema_rampup=None, use_ema=True,
                 world_size=1, rank=0, process_group=None, seed=0,
                 augment_kwargs=None, augment_type='sg2_ada', augment_p=0.0, ada_target=None, ada_interval=4, ada_kimg=500):
        self.device = torch.device(device)
        self.rank, self.world_size = rank, world_size
        self.batch, self.batch_gpu = batch, batch_gpu
        assert batch % batch_gpu == 0
        self.ema_kimg, self.ema_rampup, self.use_ema = ema_kimg, ema_rampup, use_ema
        self.cur_nimg = 0
        self.batch_idx = 0
        self._round_plan = {}
        self._round_proven = set()  # merged-round plans that have run once without exhausting device memory
        self._count_nonfinite = False
        self.comm_stats = None      # bench.py: {phase: dict(pairs=[event pairs around waits for exchanges], nonfinite=device counter)}

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
        if self.G_ema is not None:      # G now holds rank 0's weights (the wrappers broadcast them): the average starts from those on every rank
            misc.copy_params_and_buffers(self.G, self.G_ema, require_all=True)

        # discriminator augmentation + the ADA heuristic's statistics (reference trainers.py:575-584)
        self.augment_pipe, self.ada_stats = None, None
        self.ada_target, self.ada_interval, self.ada_kimg = ada_target, ada_interval, ada_kimg
        if augment_kwargs is not None and (augment_p > 0 or ada_target is not None):
            from .augmentations import augmentations
            self.augment_pipe = augmentations[augment_type](**augment_kwargs).train().requires_grad_(False).to(self.device)
            self.augment_pipe.p.copy_(torch.as_tensor(float(augment_p)))
            if ada_target is not None:
                self.ada_stats = training_stats.Collector(regex='Loss/signs/real')        # kept for reporting parity (reference :584)
                # The reference reads E[sign(D(real))] through that collector: a host synchronisation every `ada_interval` iterations
                # that drains the launch queue.  Here the running [count, sum] of the reported signs stays on the device and the
                # adjustment of `p` is computed there; SBG_ADA_SYNC=1 restores the synchronous path.
                self._ada_sync = os.environ.get('SBG_ADA_SYNC', '0') == '1' or self.device.type != 'cuda'
                if not self._ada_sync:
                    self._ada_acc = torch.zeros([2], dtype=torch.float64, device=self.device)
                    def tap(v, acc=self._ada_acc):
                        if v.device == acc.device:
                            acc.add_(torch.stack([torch.full([], float(v.numel()), dtype=torch.float64, device=v.device), v.sum(dtype=torch.float64)]))
                    self._ada_tap = training_stats.add_tap('Loss/signs/real', tap)
                    self._ada_adopt_at = None                   # iteration at whose start the pipe adopts the announced strength
            la['augment_pipe'] = self.augment_pipe
        # the fused training-time synthesis layers are first order only: the loss orchestration switches them off for the passes of a generator
        # regulariser (path length differentiates G twice) and on for every other pass (losses_base.accumulate_gradients)
        from ..torch_utils.ops import modconv
        modconv.enabled = True
        self.loss = losses_arch[loss_arch](device=self.device, gen_regs=list(gen_regs), dis_regs=list(dis_regs),
                                           D=self.dp_modules['D'], loss=loss, **la)

        # phases (reference :601-633).  The branch is on the interval alone: with an interval > 0 a 'reg' phase slot exists (and the
        # optimizer's lr / betas are rescaled for it) whether or not a regulariser is configured -- the headline sg2ada.yaml has
        # g_reg_interval = 16 and no generator regulariser, and trains G with lr * 16/17.  A slot whose program is empty draws its
        # latents like any other phase and does nothing else (the reference's optimizer step sees no gradients there and skips).
        # `n_dis` stretches only the un-split 'Gboth' phase (:609-610, :618).
        self.phases = []
        g_reducers = [self.dp_modules[k] for k in self.dp_modules if k.startswith('G')]
        for name, module, reducers, (opt_name, opt_kwargs), interval, both_interval in [
                ('G', self.G, g_reducers, optim_gen, g_reg_interval, int(n_dis)),
                ('D', self.D, [self.dp_modules['D']], optim_disc, d_reg_interval, 1)]:
            if interval is None or interval <= 0:
                opt = optimizers[opt_name](params=module.parameters(), **dict(opt_kwargs))
                slots = [(name + 'both', both_interval)]
            else:
                opt = optimizers[opt_name](params=module.parameters(), **lazy_reg_opt_kwargs(dict(opt_kwargs), interval))
                slots = [(name + 'main', 1), (name + 'reg', int(interval))]
            for phase_name, phase_interval in slots:
                self.phases.append(EasyDict(name=phase_name, module=module, reducers=reducers, opt=opt, interval=phase_interval,
                                            idle=len(self.loss.program(phase_name)) == 0))

    def close(self):
        """detach from process-wide hooks (the statistics tap of the ADA heuristic) so that a later engine starts clean"""
        tap = getattr(self, '_ada_tap', None)
        if tap is not None:
            training_stats.remove_tap('Loss/signs/real', tap)
            self._ada_tap = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------------------
    def train_iteration(self, real_img, real_c, all_gen_z=None, all_gen_c=None):
        """real_img: [batch, C, H, W] float in [-1, 1] on the device; real_c: [batch, c_dim].  One pass over the phases."""
        assert real_img.shape[0] == self.batch
        if getattr(self, '_ada_adopt_at', None) is not None and self.batch_idx >= self._ada_adopt_at:
            self.augment_pipe.adopt_strength()
            self._ada_adopt_at = None
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
            if phase.idle or self.batch_idx % phase.interval != 0:
                continue
            with torch.autograd.profiler.record_function(phase.name):
                for r in phase.reducers:
                    r.zero_grad()
                phase.module.requires_grad_(True)
                merged = self._rounds_in_one_pass(phase.name, rounds)
                if merged:
                    # every loss term is a mean over the pass: the mean over `rounds` rounds times `rounds` is the sum of the rounds' means
                    mapping = getattr(self.G, 'mapping', None)
                    first_try = (phase.name, rounds) not in self._round_proven
                    w_avg0 = mapping.w_avg.clone() if first_try and mapping is not None and hasattr(mapping, 'w_avg') else None
                    if mapping is not None:
                        mapping.w_avg_rounds = rounds
                    try:
                        self.loss.accumulate_gradients(phase=phase.name, real_img=real_img, real_c=torch.cat(real_cs), gen_z=torch.cat(phase_z),
                                                       gen_c=torch.cat(phase_c), sync=True, gain=phase.interval * rounds, segments=rounds)
                        self._round_proven.add((phase.name, rounds))
                    except torch.OutOfMemoryError:
                        # `batch_gpu` is the config's memory knob: a run whose rounds fit one at a time must not die because they were merged.
                        # Only the first attempt of a plan may fall back (afterwards the plan is known to fit and an OOM is a real one).
                        if not first_try:
                            raise
                        merged = False
                        self._round_plan[(phase.name, rounds)] = False
                        for r in phase.reducers:
                            r.finish(nan_to_num=False, reduce=False); r.zero_grad()
                        if w_avg0 is not None:
                            mapping.w_avg.copy_(w_avg0)
                        torch.cuda.empty_cache()
                    finally:
                        if mapping is not None:
                            mapping.w_avg_rounds = 1
                if not merged:
                    for round_idx, (img, c, z, gc) in enumerate(zip(reals, real_cs, phase_z, phase_c)):
                        sync = (round_idx == rounds - 1)
                        self.loss.accumulate_gradients(phase=phase.name, real_img=img, real_c=c, gen_z=z, gen_c=gc, sync=sync, gain=phase.interval)
                phase.module.requires_grad_(False)
            with torch.autograd.profiler.record_function(phase.name + '_opt'):
                st = self.comm_stats.setdefault(phase.name, dict(pairs=[], nonfinite=torch.zeros([], dtype=torch.int64, device=self.device), runs=0)) \
                    if self.comm_stats is not None else None
                for r in phase.reducers:
                    if st is not None and self._count_nonfinite:
                        r.nonfinite = st['nonfinite']
                    r.finish()              # wait for the all-reduce, average, nan_to_num (reference :745-747)
                    if st is not None:
                        st['pairs'].extend(r.timing or []); r.timing = []
                        r.nonfinite = None
                if st is not None:
                    st['runs'] += 1
                phase.opt.step()

        if self.G_ema is not None:
            with torch.autograd.profiler.record_function('Gema'):
                self.update_ema()
        self.cur_nimg += self.batch * self.world_size
        self.batch_idx += 1

        # ADA heuristic (reference :768-771): nudge the strength so that E[sign(D(real))] tracks `ada_target`
        if self.ada_stats is not None and self.batch_idx % self.ada_interval == 0:
            step = (self.batch * self.world_size * self.ada_interval) / (self.ada_kimg * 1000)
            if self._ada_sync:
                self.ada_stats.update()
                adjust = np.sign(self.ada_stats['Loss/signs/real'] - self.ada_target) * step
                self.augment_pipe.p.copy_((self.augment_pipe.p + adjust).clamp_(min=0))
            else:
                acc = self._ada_acc
                if self.world_size > 1:
                    torch.distributed.all_reduce(acc)
                mean = acc[1] / acc[0].clamp(min=1)
                adjust = torch.where(acc[0] > 0, torch.sign(mean - self.ada_target) * step, torch.zeros_like(mean))
                self.augment_pipe.p.copy_((self.augment_pipe.p + adjust.to(self.augment_pipe.p.dtype)).clamp_(min=0))
                acc.zero_()
                # the sampler keeps the old strength for exactly one more iteration, then switches (deterministic, identical on all ranks)
                self.augment_pipe.announce_strength_update()
                self._ada_adopt_at = self.batch_idx + 1

    def collect_comm_stats(self, on=True, nonfinite=True):
        """bench.py: per phase, the device-event pairs around every wait for a gradient exchange (what backward did not hide) and, with
        `nonfinite`, the number of non-finite gradient elements BEFORE nan_to_num (one more pass over the flat buckets per phase: warm-up
        only).  No host synchronisation; read after torch.cuda.synchronize()."""
        self.comm_stats = {} if on else None
        self._count_nonfinite = bool(nonfinite)
        for r in self.dp_modules.values():
            r.timing = [] if on else None
            r.nonfinite = None

    def _rounds_in_one_pass(self, phase_name, rounds):
        """The accumulation rounds of a phase exist in the reference because a round is what fits its device (`batch_gpu`); here 288 GB hold
        them all, and every network pass carries ~1 ms of fixed cost ("passes per round").  The rounds of a phase are therefore
        evaluated in ONE pass over [round 0; round 1; ...] when that computes the same thing: per-sample-independent networks (StyleGAN2 blocks
        without attention: no batch statistics, no power iterations), the mapping network's running average advanced once per round in order
        (MappingNetwork.w_avg_rounds), minibatch-std groups kept those of the separate rounds (Discriminator.merged_batch_order), stateless
        regularisers, and no tensor of the pass at the op layer's 2 GiB limit.  Same losses, statistics and parameter gradients up to
        summation order; the per-layer noise and the augmentation pipe draw once for the whole pass instead of once per round (same
        distribution).  SBG_MERGE_ROUNDS=0 keeps the rounds apart."""
        if rounds <= 1 or not merge_rounds or not getattr(self.G, 'rounds_mergeable', False):
            return False
        hit = self._round_plan.get((phase_name, rounds))
        if hit is None:
            syn = getattr(self.G, 'synthesis', None)
            fits = syn.pass_plan(rounds * self.batch_gpu) is not None if hasattr(syn, 'pass_plan') else True
            hit = fits and self.loss.rounds_mergeable(phase_name, self.batch_gpu, rounds)
            self._round_plan[(phase_name, rounds)] = hit
        return hit

    # -- snapshot / resume (reference trainers.py:636-656 pickles whole modules; here plain state dicts + counters) ---------
    def state_dict(self):
        """Everything needed to continue the run: networks, the augmentation pipe's strength, optimizer moments (a superset of the
        reference's snapshot, which drops the optimizers) and the progress counters -- tensors and plain containers only, so the file
        loads with ``torch.load(..., weights_only=True)``."""
        opts = {}
        for ph in self.phases:
            opts.setdefault(ph.name[0], ph.opt.state_dict())       # 'G' / 'D': main and lazy-reg phases share one optimizer
        out = dict(G=self.G.state_dict(), D=self.D.state_dict(), optimizers=opts,
                   progress=dict(cur_nimg=int(self.cur_nimg), batch_idx=int(self.batch_idx)))
        if self.G_ema is not None:
            out['G_ema'] = self.G_ema.state_dict()
        if self.augment_pipe is not None:
            out['augment_pipe'] = self.augment_pipe.state_dict()
        return out

    def load_state_dict(self, state, strict=True, networks_only=False):
        """`networks_only`: transfer learning (reference :350-366 -- weights are copied, the schedule restarts)."""
        self.G.load_state_dict(state['G'], strict=strict)
        self.D.load_state_dict(state['D'], strict=strict)
        if self.G_ema is not None:
            self.G_ema.load_state_dict(state.get('G_ema', state['G']), strict=strict)
        if self.augment_pipe is not None and 'augment_pipe' in state and not networks_only:
            self.augment_pipe.load_state_dict(state['augment_pipe'], strict=strict)
            self.augment_pipe._strength()       # un-announced write: read synchronously, the resumed strength applies from the first call
            self._ada_adopt_at = None
        if networks_only:
            return
        for ph in self.phases:
            if ph.name[0] in state.get('optimizers', {}):
                ph.opt.load_state_dict(state['optimizers'][ph.name[0]])
        self.cur_nimg = int(state['progress']['cur_nimg'])
        self.batch_idx = int(state['progress']['batch_idx'])

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
        b_ema, b = list(self.G_ema.buffers()), list(self.G.buffers())
        if b_ema:
            torch._foreach_copy_(b_ema, b)              # one multi-tensor launch instead of one copy per buffer (noise constants, filters, w_avg)


# ----------------------------------------------------------------------------------------------------------------
# Config-driven trainers: the lifecycle `starter.multiprocesses_main` drives (reference starter.py:32-45), over StepEngine.

class SyntheticDataset:
    """Stands in for the reference's ImageFolderDataset (out of scope: the metric uses synthetic reals): uint8 U[0, 255]
    images and uniformly drawn one-hot labels, exposing the properties the trainer reads (train_parts/datasets.py:129-157)."""

    def __init__(self, resolution=32, num_channels=3, num_classes=0, seed=0):
        self.resolution, self.num_channels, self.label_dim = resolution, num_channels, num_classes
        self.image_shape = [num_channels, resolution, resolution]
        self.has_labels = num_classes > 0
        self._gen = torch.Generator().manual_seed(seed)

    def batch(self, n, device):
        img = torch.randint(0, 256, [n] + self.image_shape, generator=self._gen, dtype=torch.uint8).to(device)
        if self.has_labels:
            idx = torch.randint(0, self.label_dim, [n], generator=self._gen)
            c = torch.nn.functional.one_hot(idx, self.label_dim).float().to(device)
        else:
            c = torch.zeros([n, 0], device=device)
        return img, c


@trainers.add_to_registry("base")
class BaseTrainer:
    """One data-parallel G / D pair (reference BaseTrainer :155-876, hot path only)."""

    split_generator = False      # SG2Trainer synchronises mapping and synthesis separately

    def __init__(self):
        self.rank = 0
        self.config = None

    # -- argument assembly / validation (reference :155-395) -----------------------------------------------------------
    def setup_arguments(self, config):
        gen, perf = config.gen, config.perf
        gpus = int(perf.gpus)
        if gpus < 1 or gpus & (gpus - 1):
            raise ValueError("perf.gpus must be a power of two")
        if gen.batch < 1 or gen.batch_gpu < 1:
            raise ValueError("gen.batch and gen.batch_gpu must be set")
        batch_gpu = min(gen.batch_gpu, gen.batch // gpus)
        if gen.batch % gpus != 0 or gen.batch % (gpus * batch_gpu) != 0:
            raise ValueError("gen.batch must be a multiple of perf.gpus * gen.batch_gpu")
        self.aug = self._augment_arguments(config)
        self.real_data = config.data.dataset != "synthetic"
        if self.real_data:
            from .datasets import datasets
            if config.data.dataset not in datasets:
                raise ValueError(f"data.dataset={config.data.dataset}: use 'synthetic' (data.resolution=<R> data.num_classes=<K>) or one of {sorted(datasets.classes)}")
        self.resume_path = None if config.trans.resume == "noresume" else str(config.trans.resume)     # a network-snapshot-*.pt of this build
        if self.resume_path is not None and not os.path.isfile(self.resume_path):
            raise ValueError(f"trans.resume={self.resume_path}: no such snapshot file (named transfer-learning sources are URL fetches and "
                             "reference .pkl snapshots are pickled modules; neither is loaded here -- pass a .pt written by save_snapshot)")
        self.run_dir = os.path.join(str(config.log.output), str(config.exp.name)) if config.exp.get("name", utils.MISSING) != utils.MISSING else None
        self.snapshot_iterations = None     # iterations between snapshots; None = only on request
        from ..metrics import metric_main
        self.metrics = [str(m) for m in config.log.get("metrics", [])]      # reference :215-217
        bad = [m for m in self.metrics if not metric_main.is_valid_metric(m)]
        if bad:
            raise ValueError(f"log.metrics contains {bad}; valid: {metric_main.list_valid_metrics()}")
        self.metric_detector = config.log.get("metric_detector", None)      # local TorchScript file (or directory holding the reference's file names)
        self.stats_metrics, self.metrics_time = dict(), 0.0
        self.config = config
        self.num_gpus, self.batch_size, self.batch_gpu = gpus, gen.batch, batch_gpu
        if self.real_data:      # reference :230-261: probe the data once, then be explicit about resolution / labels / size
            from .datasets import datasets
            kw = {k: v for k, v in dict(config.get("datasets_args", {}).get(config.data.dataset, {})).items() if k not in ("args", "kwargs") and v != utils.MISSING}
            kw["path"] = config.data.dataset_path
            probe = datasets[config.data.dataset](**kw)
            kw.update(resolution=probe.resolution, use_labels=probe.has_labels, max_size=len(probe))
            if config.data.cond and not kw["use_labels"]:
                raise ValueError("data.cond=true requires labels specified in dataset.json")
            if not config.data.cond:
                kw["use_labels"] = False
            if config.data.subset:
                if not 1 <= config.data.subset <= kw["max_size"]:
                    raise ValueError(f"data.subset must be between 1 and {kw['max_size']}")
                if config.data.subset < kw["max_size"]:
                    kw.update(max_size=int(config.data.subset), random_seed=gen.seed)
            if config.data.mirror:
                kw["xflip"] = True
            probe.close()
            self.training_set_kwargs = kw
            self.data_loader_kwargs = {k: v for k, v in dict(config.get("dataloaders_args", {}).get(config.data.dataloader, {})).items()
                                       if k not in ("args", "kwargs") and v != utils.MISSING}
            self.dataset = datasets[config.data.dataset](**kw)
        else:
            self.dataset = SyntheticDataset(int(config.data.get("resolution", 32)), 3,
                                            int(config.data.get("num_classes", 0)) if config.data.cond else 0, seed=gen.seed)
        common = dict(c_dim=self.dataset.label_dim, img_resolution=self.dataset.resolution, img_channels=self.dataset.num_channels)
        self.G_kwargs = self._model_kwargs(config.gens_args[gen.generator], common)
        self.D_kwargs = self._model_kwargs(config.discs_args[gen.discriminator], common)
        strip = lambda d: {k: v for k, v in d.items() if k != "params"}
        self.G_opt = (gen.optim_gen, strip(config.optim_gen_args[gen.optim_gen]))
        self.D_opt = (gen.optim_disc, strip(config.optim_disc_args[gen.optim_disc]))
        self.gen_regs = [(name, dict(config.gen_regs_all[name])) for name in gen.gen_regs]
        self.dis_regs = [(name, dict(config.disc_regs_all[name])) for name in gen.disc_regs]
        la = dict(config.losses_arch_args[gen.loss_arch])
        self.loss_arch_kwargs = {k: v for k, v in la.items() if k not in ("args", "G_mapping", "G_synthesis") and v != utils.MISSING}
        self.ema_kimg = config.ema.kimg
        self.ema_rampup = config.ema.ramp if config.ema.ramp >= 0 else None
        self.total_kimg = gen.kimg
        if self.metrics and not (isinstance(self.metric_detector, str) and os.path.exists(self.metric_detector)):
            # the reference fetches its detectors from a URL at the first snapshot; this build never downloads.  A run on a real data set that
            # is configured with metrics (the reference's DEFAULT is [fid50k_full, is50k]) and has nothing to compute them with would train
            # for hours and report nothing: refuse at setup.  Synthetic data has no data set to score against: the metrics are dropped, loudly.
            if self.real_data:
                raise ValueError(f"log.metrics={self.metrics} needs log.metric_detector=<local TorchScript file or directory holding "
                                 "inception-2015-12-05.pt / vgg16.pt> (detectors are not downloaded in this build); log.metrics=[] turns them off")
            import warnings
            warnings.warn(f"log.metrics={self.metrics} ignored: data.dataset=synthetic has no data set to score against")
            self.metrics = []
        return self

    @staticmethod
    def _augment_arguments(config):
        """aug.{aug, p, target, augpipe} -> StepEngine keywords (reference :295-335).  The reference looks `aug.augpipe` ('bgc', ...) up
        in a registry that only holds the class name 'sg2_ada' (:335) and fails; the names mean the subsets of
        stylegan2ada/train.py:271-283, which is what is resolved here."""
        from .augmentations import augpipe_specs
        aug = config.aug
        out = dict(augment_kwargs=None, augment_type=aug.get("aug_type", "sg2_ada"), augment_p=0.0, ada_target=None, ada_interval=4, ada_kimg=500)
        if aug.aug == "ada":
            out["ada_target"] = 0.6
        elif aug.aug == "fixed":
            if aug.p < 0:
                raise ValueError(f"--aug={aug.aug} requires specifying --p")
        elif aug.aug != "noaug":
            raise ValueError(f"--aug={aug.aug} not supported")
        if aug.p >= 0:
            if aug.aug != "fixed":
                raise ValueError("--p can only be specified with --aug=fixed")
            if not 0 <= aug.p <= 1:
                raise ValueError("--p must be between 0 and 1")
            out["augment_p"] = float(aug.p)
        if aug.target >= 0:
            if aug.aug != "ada":
                raise ValueError("--target can only be specified with --aug=ada")
            if not 0 <= aug.target <= 1:
                raise ValueError("--target must be between 0 and 1")
            out["ada_target"] = float(aug.target)
        if aug.aug != "noaug":
            if aug.augpipe not in augpipe_specs:
                raise ValueError(f"aug.augpipe={aug.augpipe} not in {sorted(augpipe_specs)}")
            base = config.get("augpipe_specs", {}).get(out["augment_type"], {})        # constructor arguments (std-devs, ranges) from the config
            out["augment_kwargs"] = {**{k: v for k, v in dict(base).items() if k not in ("args", "kwargs")}, **augpipe_specs[aug.augpipe]}
        return out

    @staticmethod
    def _model_kwargs(group, common):
        import inspect
        kw = {k: (dict(v) if isinstance(v, dict) else v) for k, v in group.items() if not (isinstance(v, str) and v == utils.MISSING)}
        kw = {k: v for k, v in kw.items() if k not in ("args", "kwargs")}
        kw.update({k: v for k, v in common.items() if k in group or k in ("c_dim", "img_resolution", "img_channels")})
        return kw

    # -- lifecycle ------------------------------------------------------------------------------------------------------
    def setup_logs(self):
        self.stats = training_stats.Collector(regex=".*")

    def distribute_torch(self, temp_dir):
        if self.num_gpus > 1:
            init_file = os.path.abspath(os.path.join(temp_dir, ".torch_distributed_init"))
            backend = "nccl" if torch.cuda.is_available() else "gloo"
            torch.distributed.init_process_group(backend=backend, init_method=f"file://{init_file}", rank=self.rank, world_size=self.num_gpus)
        sync_device = self.device() if self.num_gpus > 1 else None
        training_stats.init_multiprocessing(rank=self.rank, sync_device=sync_device)

    def device(self):
        return torch.device("cuda", self.rank) if torch.cuda.is_available() else torch.device("cpu")

    def init_params(self):
        seed = self.config.gen.seed
        np.random.seed(seed * self.num_gpus + self.rank)
        torch.manual_seed(seed * self.num_gpus + self.rank)

    def setup_dataset(self):
        """real data: endless rank-sharded stream of uint8 batches (reference :517-524); synthetic batches are drawn on demand"""
        self.training_set_iterator = None
        if self.real_data:
            from .dataloaders import dataloaders
            sampler = misc.InfiniteSampler(dataset=self.dataset, rank=self.rank, num_replicas=self.num_gpus, seed=self.config.gen.seed)
            loader = dataloaders[self.config.data.dataloader](dataset=self.dataset, sampler=sampler, batch_size=self.batch_size // self.num_gpus,
                                                              **self.data_loader_kwargs)
            self.training_set_iterator = iter(loader)

    def next_batch(self, n, device):
        """-> (uint8 images [n, C, H, W], float32 labels [n, label_dim]) on `device` (normalisation happens there)"""
        if self.training_set_iterator is None:
            return self.dataset.batch(n, device)
        img, c = next(self.training_set_iterator)
        return img.to(device, non_blocking=True), c.to(device, non_blocking=True)

    def setup_networks(self):
        gen = self.config.gen
        self.engine = StepEngine(self.device(), generator=gen.generator, discriminator=gen.discriminator, gen_kwargs=self.G_kwargs,
                                 disc_kwargs=self.D_kwargs, loss_arch=gen.loss_arch, loss=gen.loss, loss_arch_kwargs=self.loss_arch_kwargs,
                                 gen_regs=self.gen_regs, dis_regs=self.dis_regs, optim_gen=self.G_opt, optim_disc=self.D_opt,
                                 g_reg_interval=gen.g_reg_interval, d_reg_interval=gen.d_reg_interval, n_dis=int(gen.n_dis),
                                 batch=self.batch_size // self.num_gpus, batch_gpu=self.batch_gpu, ema_kimg=self.config.ema.kimg,
                                 ema_rampup=self.ema_rampup, use_ema=self.config.ema.use_ema, world_size=self.num_gpus, rank=self.rank,
                                 seed=gen.seed, **self.aug)

    def setup_augmentations(self):
        self.augment_pipe = self.engine.augment_pipe        # built by StepEngine (it owns the loss object the pipe plugs into)
        if self.resume_path is not None:    # networks, pipe and optimizers exist now: continue from the snapshot
            self.resume(self.resume_path)

    def save_snapshot(self, cur_nimg=None, run_dir=None):
        """network-snapshot-<kimg>.pt + training_options.json with `start_options` (reference :636-656, :821-832).  Rank 0 writes."""
        run_dir = run_dir or self.run_dir
        assert run_dir is not None, "save_snapshot needs exp.name / log.output or an explicit run_dir"
        cur_nimg = self.engine.cur_nimg if cur_nimg is None else cur_nimg
        path = os.path.join(run_dir, f"network-snapshot-{cur_nimg // 1000:06d}.pt")
        if self.rank == 0:
            import json
            os.makedirs(run_dir, exist_ok=True)
            state = self.engine.state_dict()
            torch.save(state, path)
            options = dict(start_options=dict(cur_nimg=int(self.engine.cur_nimg), batch_idx=int(self.engine.batch_idx)),
                           snapshot=os.path.basename(path), num_gpus=self.num_gpus, batch_size=self.batch_size, batch_gpu=self.batch_gpu)
            with open(os.path.join(run_dir, "training_options.json"), "wt") as f:
                json.dump(options, f, indent=2)
        return path

    def evaluate_metrics(self, snapshot_path=None, detector=None, metrics=None):
        """every configured metric on G_ema (G when the average is off) against the training set (reference :659-674).  The feature
        detector must be local (`log.metric_detector=<TorchScript file | directory>` or the `detector` argument: a path or a callable);
        the reference's URL fetch does not exist here, so without one this raises instead of silently skipping."""
        from ..metrics import metric_main
        detector = detector if detector is not None else self.metric_detector
        names = list(self.metrics if metrics is None else metrics)
        if not names:
            return {}
        if not self.real_data:
            raise ValueError("metrics need a real data set (data.dataset=image_folder ...)")
        kw = dict(detector=None, detector_dir=None)
        if callable(detector) or (isinstance(detector, str) and os.path.isfile(detector)):
            kw["detector"] = detector
        elif isinstance(detector, str) and os.path.isdir(detector):
            kw["detector_dir"] = detector
        else:
            raise ValueError("no local feature detector: set log.metric_detector=<TorchScript file or directory> (detectors are not downloaded)")
        G = self.engine.G_ema if self.engine.G_ema is not None else self.engine.G
        self.metrics_time = 0.0
        for metric in names:
            result = metric_main.calc_metric(metric=metric, dataset_name=self.config.data.dataset, G=G, dataset_kwargs=self.training_set_kwargs,
                                             num_gpus=self.num_gpus, rank=self.rank, device=self.engine.device, **kw)
            self.metrics_time += result["total_time"]
            if self.rank == 0:
                metric_main.report_metric(result, run_dir=self.run_dir, snapshot_pkl=snapshot_path)
            self.stats_metrics.update(result.results)
        return dict(self.stats_metrics)

    def resume(self, path, networks_only=False):
        state = torch.load(path, map_location=self.engine.device, weights_only=True)
        self.engine.load_state_dict(state, networks_only=networks_only)
        if self.num_gpus > 1:       # every rank read the same file; keep the data-parallel invariant explicit
            for module in (self.engine.G, self.engine.D):
                for t in list(module.parameters()) + list(module.buffers()):
                    torch.distributed.broadcast(t.detach(), src=0)
        return state["progress"]

    def distrib_acrros_gpu(self):
        pass        # StepEngine wraps its modules in GradReducer at construction (broadcast of rank 0's weights included)

    def setup_training_phases(self):
        return self.engine.phases

    def export_sample_images(self):
        pass

    def training_loop(self, max_iterations=None):
        eng = self.engine
        total = self.total_kimg * 1000
        it = 0
        while (max_iterations is None or it < max_iterations) and (total < 0 or eng.cur_nimg < total or it == 0):
            img, c = self.next_batch(eng.batch, eng.device)
            eng.train_iteration(img.to(torch.float32) / 127.5 - 1, c)
            it += 1
            if self.snapshot_iterations and it % self.snapshot_iterations == 0:
                path = self.save_snapshot()
                if self.metrics:            # every configured metric after each snapshot, on all ranks (reference :834-836)
                    self.evaluate_metrics(snapshot_path=path)
            if max_iterations is None and total >= 0 and eng.cur_nimg >= total:
                break
        self.stats.update()
        return it


@trainers.add_to_registry("sg2")
class SG2Trainer(BaseTrainer):
    """StyleGAN2 trainer: the generator's mapping and synthesis networks are separate data-parallel modules (reference :881-893)."""
    split_generator = True
