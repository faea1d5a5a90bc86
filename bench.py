#!/usr/bin/env python3
"""Benchmark of the G+D training step on N MI355X.  Headline: images/sec of StyleGAN2-ADA at 256x256, bf16.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--workload sg2ada|ffhq_sg2|sg2attent|big_gan]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one iteration of the reference's training loop body (train_parts/trainers.py:711-765) on synthetic inputs: every phase
that is due (Gmain / Dmain each iteration, lazy regularisers every `interval`-th), each phase = zero_grad -> accumulation rounds of
forward/backward -> gradient all-reduce -> nan_to_num -> Adam, then the G_ema update.  The timed region starts on an iteration on
which every phase is due.  Latents ~ N(0, 1), reals uint8 U[0, 255] generated on the device once, uniformly drawn one-hot labels
for the class-conditional workload; random-init weights (no network for datasets or checkpoints).

Workloads (BASELINE.json `configs`; the default is the one the metric is quoted on):
  sg2ada     configs/sg2ada.yaml @ 256x256: z = w = 512, 2 mapping layers, channel_base 32768, D 'orig', mbstd 32, softplus + R1(0.01)/4,
             a (regulariser-less) Greg slot every 16 -- i.e. G's Adam runs at lr * 16/17 --, batch 64 = 2 x batch_gpu 32, ADA off
  ffhq_sg2   configs/ffhq_sg2.yaml @ 1024x1024: 6 mapping layers, channel_base 16384, D 'resnet', mbstd 8, R1(1)/4 + path length(2, shrink 2)/16
  sg2attent  configs/sg2attent.yaml @ 32x32 (the yaml's data; --res 256 for the SURVEY's second size): G attention at 32/16/8/4, D at 32
  big_gan    configs/big_gan.yaml @ 128x128: class-conditional BigGAN (10 classes), hinge, n_dis 4, D attention at 32, batch 48 (the yaml's
             50 is not divisible by 8 ranks), cross-replica batch norm when N > 1
In the StyleGAN2 workloads every block from 8x8 up runs in bf16 with conv_clamp 256 (the reference's mixed-precision recipe with bf16 in
place of fp16, applied to more blocks than its default num_fp16_res = 4); BigGAN runs in fp32 storage (convolutions as split-bf16 MFMA).

--scaling weak (default): the per-rank batch is fixed, the global batch is batch x N.  --scaling strong: the GLOBAL batch is fixed at the
config's value and each rank takes batch // N of it in rounds of min(batch_gpu, batch // N) -- the reference's own partitioning
(trainers.py:524,737).

Prints ONE JSON line on rank 0 (README / DESIGN.md explain the fields).  `roofline` is measured live: the library brackets every kernel
launch of the timed region with hipEvents on the launch stream (sbg_prof_*), and the dominant kernel kind's algorithmic flops / summed
duration is reported against the dense bf16 MFMA peak.  `cpu_baseline` times the CPU oracle (oracle/, the fixture-pinned restatement of
the reference's eager CPU path) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0

RES, Z_DIM, BATCH, BATCH_GPU = 256, 512, 64, 32
ADAM = dict(lr=0.0025, betas=[0, 0.99], eps=1e-8)


def _bf16_blocks(res):
    """num_fp16_res that puts every block from 8x8 up in reduced precision"""
    return max(res.bit_length() - 3, 0)


def sg2ada_kwargs(res=RES, num_fp16_res=None, conv_clamp=256, channel_base=32768, mbstd=32, mapping_layers=2, d_arch='orig', attn_g=(), attn_d=()):
    nfp = _bf16_blocks(res) if num_fp16_res is None else num_fp16_res
    gk = dict(z_dim=Z_DIM, c_dim=0, w_dim=512, img_resolution=res, img_channels=3, attentions=list(attn_g), mapping_kwargs=dict(num_layers=mapping_layers),
              synthesis_kwargs=dict(channel_base=channel_base, num_fp16_res=nfp, block_kwargs=dict(conv_clamp=conv_clamp)))
    dk = dict(c_dim=0, img_resolution=res, img_channels=3, attentions=list(attn_d), architecture=d_arch, channel_base=channel_base, num_fp16_res=nfp,
              conv_clamp=conv_clamp, epilogue_kwargs=dict(mbstd_group_size=mbstd))
    return gk, dk


def workload(name, res=None, nfp=None):
    """-> dict(res, batch, batch_gpu, c_dim, dtype, label, steps (default K), engine (StepEngine keywords))"""
    sg2 = dict(generator='sg2_classic', discriminator='sg2_classic', loss_arch='sg2', loss='softplus', loss_arch_kwargs=dict(style_mixing_prob=0),
               optim_gen=('adam', dict(ADAM)), optim_disc=('adam', dict(ADAM)))
    if name == 'sg2ada':
        res = res or 256
        gk, dk = sg2ada_kwargs(res=res, num_fp16_res=nfp)
        return dict(res=res, batch=64, batch_gpu=32, c_dim=0, dtype='bf16', steps=8,
                    label=f'configs/sg2ada.yaml @ {res}x{res}: sg2_classic G (skip) + D (orig), softplus + R1(0.01)/4, idle Greg slot /16 (G lr x 16/17)',
                    engine=dict(sg2, gen_kwargs=gk, disc_kwargs=dk, gen_regs=[], dis_regs=[('r1', dict(r1_gamma=0.01))], g_reg_interval=16,
                                d_reg_interval=4, ema_kimg=500, ema_rampup=0.05))
    if name == 'ffhq_sg2':
        res = res or 1024
        gk, dk = sg2ada_kwargs(res=res, channel_base=16384, mbstd=8, mapping_layers=6, d_arch='resnet')
        return dict(res=res, batch=64, batch_gpu=32, c_dim=0, dtype='bf16', steps=16,
                    label=f'configs/ffhq_sg2.yaml @ {res}x{res}: channel_base 16384, 6 mapping layers, D resnet, softplus + R1(1)/4 + path length(weight 2, shrink 2)/16',
                    engine=dict(sg2, gen_kwargs=gk, disc_kwargs=dk, gen_regs=[('ppl', dict(pl_batch_shrink=2, pl_decay=0.01, pl_weight=2.))],
                                dis_regs=[('r1', dict(r1_gamma=1.))], g_reg_interval=16, d_reg_interval=4, ema_kimg=20, ema_rampup=None))
    if name == 'sg2attent':
        res = res or 32
        gk, dk = sg2ada_kwargs(res=res, attn_g=[32, 16, 8, 4], attn_d=[32])
        return dict(res=res, batch=64, batch_gpu=64, c_dim=0, dtype='bf16', steps=8,
                    label=f'configs/sg2attent.yaml @ {res}x{res}: sg2_classic + non-local attention (G at 32/16/8/4, D at 32), softplus + R1(0.01)/4',
                    engine=dict(sg2, gen_kwargs=gk, disc_kwargs=dk, gen_regs=[], dis_regs=[('r1', dict(r1_gamma=0.01))], g_reg_interval=16,
                                d_reg_interval=4, ema_kimg=500, ema_rampup=0.05))
    if name == 'big_gan':
        res = res or 128
        opt = dict(lr=0.0002, betas=[0.0, 0.999], eps=1e-8)
        return dict(res=res, batch=48, batch_gpu=48, c_dim=10, dtype='f32', steps=8,
                    label=f'configs/big_gan.yaml @ {res}x{res}: BigGAN G (ch 64, no attention) + D (ch 64, attention at 32), 10 classes, hinge, n_dis 4 '
                          '(G phase every 4th step, gain 4), batch 48 in place of 50',
                    engine=dict(generator='big_gan', discriminator='big_gan', loss_arch='base', loss='hinge', loss_arch_kwargs=dict(),
                                gen_kwargs=dict(c_dim=10, img_resolution=res, G_shared=False, G_attn='0', G_init='N02', n_classes=10),
                                disc_kwargs=dict(c_dim=10, img_resolution=res, D_attn='32', D_init='N02', n_classes=10),
                                optim_gen=('adam', opt), optim_disc=('adam', dict(opt)), gen_regs=[], dis_regs=[], g_reg_interval=0, d_reg_interval=0,
                                n_dis=4, ema_kimg=500, ema_rampup=None))
    raise SystemExit(f'unknown workload {name}')


def build_engine(device, world_size, rank, wl, batch, batch_gpu, ada=None):
    from style_big_gan_amd.train_parts import trainers
    kw = dict(wl['engine'])
    if wl['engine']['generator'] == 'big_gan' and world_size > 1:
        kw['gen_kwargs'] = dict(kw['gen_kwargs'], cross_replica=True)      # synchronised batch norm over the ranks (reference layers.py:297-298)
    return trainers.StepEngine(device, batch=batch, batch_gpu=batch_gpu, world_size=world_size, rank=rank, seed=0, **kw, **ada_kwargs(ada))


def ada_kwargs(ada):
    """--ada P: the 'bgc' augmentation pipe in front of every discriminator call at starting strength P with the ADA heuristic
    (target 0.6) running -- a secondary measurement; the headline metric is quoted with ADA off."""
    if ada is None:
        return {}
    from style_big_gan_amd.train_parts.augmentations import augpipe_specs
    return dict(augment_kwargs=dict(augpipe_specs['bgc']), augment_p=float(ada), ada_target=0.6, ada_interval=4, ada_kimg=500)


def summarize_kernels(records):
    """aggregate the launch log by kernel kind -> {kind: dict(launches, ms, flops, bytes)}"""
    out = {}
    for r in records:
        k = out.setdefault(r['kind'], dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
        k['launches'] += 1; k['ms'] += r['ms']; k['flops'] += r['flops']; k['bytes'] += r['bytes']
    return out


# kernel kind of the launch log -> kernel names in the rocprofv3 traces
PMC_KERNELS = {'conv_igemm': ('conv_halo_ld_kernel', 'conv_gather_ld_kernel', 'conv_k64_kernel', 'conv_igemm_dma_kernel', 'conv_igemm_kernel', 'conv_up2_kernel', 'conv_thin_kernel'),
               'conv_wgrad': ('conv_wgrad_rows_kernel', 'conv_wgrad_kernel'), 'upfirdn2d': ('upfirdn2d_fir', 'upfirdn2d_kernel'),
               'bias_act': ('bias_act',), 'scale_nc': ('scale_nc',), 'dot_hw': ('dot_hw',)}


def pmc_traffic(kind):
    """HBM bytes per launch of `kind` from the newest (highest tag) profiles/*_traffic.json: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 of separate
    rocprofv3 --pmc passes over this same bench command (profiles/collect.sh; PMC counters cannot be read from inside the
    process).  Returns (bytes_per_launch, source) or (None, None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_traffic.json')))        # tags sort by round and letter: the last one is the newest
    if not files:
        return None, None
    try:
        data = json.load(open(files[-1]))['kernels']
    except Exception:
        return None, None
    tot = n = 0
    for key, rec in data.items():            # keys are kernel names, possibly truncated C++ signatures
        if any(name in key for name in PMC_KERNELS.get(kind, ())) and rec['launches'] > 0:
            tot += rec['hbm_bytes_per_launch'] * rec['launches']; n += rec['launches']
    return (round(tot / n), os.path.relpath(files[-1], ROOT)) if n else (None, None)


def host_cores():
    """CPU threads this process may really use: the scheduler affinity capped by the cgroup's cpu quota (a GPU box shows all of the
    host's cores to os.cpu_count() while the job owns a share of them; timing the oracle on 128 threads over a 16-core share is what
    made round 2's baseline slower than an 8-core container)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except Exception:
            continue
    return max(1, n)


def cpu_worker(batch, threads, res=RES):
    """One Gmain + Dmain + Dreg pass of the CPU oracle on `batch` images at the headline workload's shapes (fp32, `threads` host
    threads); prints {'t_main', 't_reg'} in seconds.  Runs in a child process of the cpu_baseline leg; touches no GPU."""
    torch.set_num_threads(threads)
    from oracle import networks as ON
    from style_big_gan_amd.train_parts import discriminators, generators
    torch.manual_seed(0)
    gk, dk = sg2ada_kwargs(res=res, num_fp16_res=0, conv_clamp=None)
    G = generators.generators['sg2_classic'](**gk)
    D = discriminators.discriminators['sg2_classic'](**dk)
    cfg = ON.default_cfg(z_dim=Z_DIM, w_dim=512, c_dim=0, img_resolution=res, channel_base=32768, mapping_layers=2,
                         g_architecture='skip', d_architecture='orig', conv_clamp=None, mbstd_group_size=32)
    gsd = {k: v.detach().float() for k, v in G.state_dict().items()}
    dsd = {k: v.detach().float() for k, v in D.state_dict().items()}
    n = batch
    z_g, z_d = torch.randn(n, Z_DIM), torch.randn(n, Z_DIM)
    real = torch.randint(0, 256, [n, 3, res, res]).float() / 127.5 - 1
    t0 = time.perf_counter()
    ON.gd_step_grads(gsd, dsd, cfg, z_g, z_d, real, r1_gamma=None, noise_mode='const')         # Gmain + Dmain
    t_main = time.perf_counter() - t0
    t0 = time.perf_counter()
    d_leaf = {k: v.clone().requires_grad_('resample' not in k) for k, v in dsd.items()}      # Dreg alone: D forward on reals, R1 double backward
    real_in = real.clone().requires_grad_(True)
    logits = ON.discriminator(d_leaf, real_in, torch.zeros(n, 0), cfg)
    r1 = torch.autograd.grad(logits.sum(), real_in, create_graph=True)[0]
    pen = (r1.square().sum([1, 2, 3]) * 0.005).mean()
    torch.autograd.grad(pen, [v for v in d_leaf.values() if v.requires_grad], allow_unused=True)
    t_reg = time.perf_counter() - t0
    print(json.dumps(dict(t_main=t_main, t_reg=t_reg)))


def cpu_baseline(res=RES, budget_s=60.0):
    """img/s of a G+D step (Dreg amortised over 4 iterations, like the GPU figure) of the CPU oracle on the host cores this job owns.
    The split of those cores that maximises img/s is searched within `budget_s`: k concurrent processes x (cores / k) threads, each
    running one Gmain + Dmain + Dreg on its own batch; a candidate is skipped when the budget left is less than the previous one took."""
    import subprocess
    cores = host_cores()
    cands = [(1, cores, 2)]
    k = 2
    while cores // k >= 2 and k <= 4:           # at most 4 workers: a GPU box admits 6 processes with the device open, and the workers' imports open it
        cands.append((k, cores // k, 2)); k *= 2
    t_start, best, tried, last = time.perf_counter(), None, [], 0.0
    for procs, threads, batch in cands:
        if tried and budget_s - (time.perf_counter() - t_start) < 1.3 * last:
            break
        t0 = time.perf_counter()
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='')
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-worker', str(batch), str(threads), '--res', str(res)],
                               stdout=subprocess.PIPE, env=env, text=True) for _ in range(procs)]
        outs = [q.communicate()[0] for q in ps]
        last = time.perf_counter() - t0
        if any(q.returncode != 0 for q in ps):
            tried.append(dict(processes=procs, threads=threads, batch=batch, failed=True)); continue
        rec = [json.loads(o.strip().splitlines()[-1]) for o in outs]
        t_main, t_reg = max(r['t_main'] for r in rec), max(r['t_reg'] for r in rec)      # concurrent workers: the slowest one bounds the throughput
        val = procs * batch / (t_main + t_reg / 4)
        tried.append(dict(processes=procs, threads=threads, batch=batch, img_s=round(val, 4), t_main_s=round(t_main, 2), t_reg_s=round(t_reg, 2)))
        if best is None or val > best[0]:
            best = (val, procs, threads, batch, t_main, t_reg)
    if best is None:
        return None
    val, procs, threads, batch, t_main, t_reg = best
    return dict(value=round(val, 4), unit='img/s', cores=procs * threads, kind='port',
                sample=f'oracle/ (CPU restatement of the reference eager path, fp32): {procs} process(es) x {threads} threads, each one Gmain+Dmain '
                       f'({t_main:.1f}s) + one Dreg ({t_reg:.1f}s, /4) on batch {batch} at {res}x{res}, sg2ada shapes; best of the splits tried within '
                       f'{budget_s:.0f}s on the {cores} host cores this job owns (os.cpu_count() = {os.cpu_count()}); the reference proper, imported in '
                       'the build container, ran the same passes at 0.18 img/s on 8 cores (BASELINE.md section 3)',
                splits_tried=tried)


# ----------------------------------------------------------------------------------------------------------------------------------------
# N ranks

def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU, the reference's starter.py:26-30) BEFORE
    anything here has touched a GPU, give them the rendezvous through the environment, pass rank 0's JSON line through, and fail if any
    rank fails.  -> exit code"""
    import socket
    import subprocess
    backend = os.environ.get('SBG_DIST_BACKEND', 'nccl')
    n_dev = torch.cuda.device_count()           # counting devices does not initialise the runtime
    if backend == 'nccl' and n_dev < n:
        print(f'bench.py --gpus {n}: {n} ranks need {n} devices, this machine has {n_dev} '
              '(SBG_DIST_BACKEND=gloo rehearses the N-rank path with ranks sharing devices; it is not a measurement)', file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r % max(n_dev, 1)), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    code = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                rc = procs[r].poll()
                if rc is None:
                    continue
                pending.discard(r)
                if rc != 0 and code == 0:
                    code = rc if rc > 0 else 1
                    print(f'bench.py: rank {r} exited with code {rc}; stopping the other ranks', file=sys.stderr)
                    for q in pending:
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    return code


def init_ranks(args):
    """-> (world, rank, local_rank, backend, ranks_seen).  `ranks_seen` is an all-reduce of ones over the backend: the line proves how many
    ranks took part."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}')
    backend = None
    seen = 1
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = os.environ.get('SBG_DIST_BACKEND', 'nccl')        # 'gloo' only to rehearse N > 1 on a one-GPU box / on the CPU
        if backend == 'nccl':
            if torch.cuda.device_count() <= local_rank:
                raise SystemExit(f'bench.py: rank {rank} needs device {local_rank}, this machine has {torch.cuda.device_count()}')
            torch.cuda.set_device(local_rank)
            torch.distributed.init_process_group(backend, device_id=torch.device('cuda', local_rank))
            ones = torch.ones([1], device=torch.device('cuda', local_rank))
        else:
            torch.distributed.init_process_group(backend)
            ones = torch.ones([1])
        torch.distributed.all_reduce(ones)
        seen = int(ones.item())
        if seen != world:
            raise SystemExit(f'bench.py: {seen} ranks answered the all-reduce, {world} were launched')
    return world, rank, local_rank, backend, seen


# ----------------------------------------------------------------------------------------------------------------------------------------
# one measurement

def measure(args, device, world, rank, name, *, res=None, nfp=None, scaling='weak', steps=None, warmup=None, ada=None, batch_arg=None,
            batch_gpu_arg=None, headline=True):
    """Build the workload's engine, run `warmup` untimed and `steps` timed steps (barrier + synchronize on both sides, MAX over ranks) and
    return the result record (rank 0: everything; other ranks: None)."""
    from style_big_gan_amd import _lib
    wl = workload(name, res, nfp)
    res = wl['res']
    steps = steps if steps is not None else wl['steps']
    warmup = args.warmup if warmup is None else warmup
    cfg_batch = batch_arg or wl['batch']
    if scaling == 'strong':
        if cfg_batch % world:
            raise SystemExit(f'strong scaling: global batch {cfg_batch} is not divisible by {world} ranks')
        batch = cfg_batch // world                                   # reference trainers.py:524
        global_batch = cfg_batch
    else:
        batch, global_batch = cfg_batch, cfg_batch * world
    batch_gpu = min(batch_gpu_arg or wl['batch_gpu'], batch)      # reference trainers.py:203-204
    assert batch % batch_gpu == 0

    eng = build_engine(device, world, rank, wl, batch=batch, batch_gpu=batch_gpu, ada=ada)
    gen = torch.Generator(device=device); gen.manual_seed(1234 + rank)
    real_u8 = torch.randint(0, 256, [batch, 3, res, res], device=device, dtype=torch.uint8, generator=gen)
    real_c = None
    if wl['c_dim']:
        real_c = torch.nn.functional.one_hot(torch.randint(0, wl['c_dim'], [batch], device=device, generator=gen), wl['c_dim']).float()

    def step():
        real = real_u8.to(torch.float32) / 127.5 - 1
        eng.train_iteration(real, real_c)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # warm-up, with the gradients of every phase checked BEFORE nan_to_num: a launch that produces NaN at these shapes would otherwise
    # be zeroed by the sanitiser and the step would still post a throughput (parallel.GradReducer.finish)
    eng.collect_comm_stats(True, nonfinite=True)
    eng.batch_idx = 0
    for _ in range(max(warmup, 1)):
        step()
    barrier()
    nonfinite = {ph: int(st['nonfinite'].item()) for ph, st in eng.comm_stats.items()}
    bad = torch.tensor([sum(nonfinite.values())], dtype=torch.float64, device=device)
    if world > 1:
        torch.distributed.all_reduce(bad)
    if bad.item() > 0:
        raise SystemExit(f'bench.py: {int(bad.item())} non-finite gradient elements during warm-up ({nonfinite} on rank {rank}): no value is reported')
    eng.collect_comm_stats(True, nonfinite=False)
    eng.batch_idx = 0       # the timed region starts on an iteration on which every phase is due: K steps contain ceil(K / interval) of each
    barrier()
    _lib.prof_enable(True)
    _lib.prof_fetch()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]      # per-step durations (torch's stream = the launch stream)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        step()
        marks[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.prof_enable(False)
    records = _lib.prof_fetch()
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t.item())

    # gradient exchange: what the compute stream waited for (not hidden under backward) vs the same all-reduces alone on the wire
    comm = None
    if world > 1:
        comm = {}
        for ph in eng.phases:
            st = eng.comm_stats.get(ph.name)
            if ph.idle or st is None or not st['runs']:
                continue
            exposed = sum(a.elapsed_time(b) for a, b in st['pairs']) / st['runs']
            flats = [b.flat for r in ph.reducers for b in r._buckets]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 3
            barrier()
            e0.record()
            for _ in range(reps):
                for f in flats:
                    torch.distributed.all_reduce(f)
            e1.record()
            torch.cuda.synchronize()
            alone = e0.elapsed_time(e1) / reps
            nbytes = sum(f.numel() * f.element_size() for f in flats)
            comm[ph.name] = dict(grad_mb=round(nbytes / 1e6, 1), buckets=len(flats), allreduce_ms_alone=round(alone, 3),
                                 busbw_gbs=round(2 * (world - 1) / world * nbytes / (alone * 1e-3) / 1e9, 1),
                                 exposed_ms_per_phase=round(exposed, 3), hidden_frac=round(min(max(1.0 - exposed / max(alone, 1e-9), 0.0), 1.0), 3))
    eng.collect_comm_stats(False)

    out = None
    if rank == 0:
        if args.launch_log and headline:
            with open(args.launch_log, 'w') as f:
                for r in records:
                    f.write(json.dumps(dict(kind=r['kind'], dims=list(r['dims']), ms=round(r['ms'], 6), flops=r['flops'], bytes=r['bytes'])) + '\n')
        kern = summarize_kernels(records)
        dom = max(kern.items(), key=lambda kv: kv[1]['ms'])[0] if kern else None
        roofline = None
        if dom is not None:
            k = kern[dom]
            if k['flops'] > 0:
                ach = k['flops'] / (k['ms'] * 1e-3) / 1e12
                roofline = dict(bound='mfma', kernel=dom, achieved=round(ach, 2), peak=MFMA_BF16_PEAK_TFLOPS, unit='TFLOP/s',
                                frac=round(ach / MFMA_BF16_PEAK_TFLOPS, 4), traffic=None,
                                launches=k['launches'], avg_launch_ms=round(k['ms'] / k['launches'], 4))
            else:
                ach = k['bytes'] / (k['ms'] * 1e-3) / 1e9
                roofline = dict(bound='hbm', kernel=dom, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                                frac=round(ach / HBM_PEAK_GBS, 4), traffic=None,
                                launches=k['launches'], avg_launch_ms=round(k['ms'] / k['launches'], 4))
        if roofline is not None and name == 'sg2ada' and nfp is None:      # the committed PMC passes are of the headline workload
            roofline['traffic'], src = pmc_traffic(dom)
            if src:
                roofline['traffic_source'] = src + ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, bytes per launch averaged over the same launches)'
                roofline['algorithmic_bytes_per_launch'] = round(kern[dom]['bytes'] / kern[dom]['launches'])
        # the 256x256 modulated 3x3 conv (M = N*65536 pixels, Cout 128, K = 9*128): the kernel the 40 % MFMA target names
        rounds = batch // batch_gpu
        one_pass = rounds > 1 and all(eng._rounds_in_one_pass(ph.name, rounds) for ph in eng.phases if not ph.idle)
        pass_batch = batch if one_pass else batch_gpu          # samples per network pass
        tgt = [r for r in records if r['kind'] == 'conv_igemm' and r['dims'][1] == 128 and r['dims'][2] == 128 and r['dims'][3] == 9
               and r['dims'][0] == pass_batch * res * res] if name == 'sg2ada' else []
        target = None
        if tgt:
            fl, ms = sum(r['flops'] for r in tgt), sum(r['ms'] for r in tgt)
            target = dict(shape=f'[{pass_batch},128,{res},{res}] (*) [128,128,3,3]', launches=len(tgt),
                          avg_launch_ms=round(ms / len(tgt), 4), tflops=round(fl / ms / 1e9, 2), mfma_frac=round(fl / ms / 1e9 / MFMA_BF16_PEAK_TFLOPS, 4))
        total_ms = sum(v['ms'] for v in kern.values())
        total_flops = sum(v['flops'] for v in kern.values())
        breakdown = {k: dict(launches=v['launches'], ms_per_step=round(v['ms'] / steps, 3),
                             tflops=round(v['flops'] / max(v['ms'], 1e-9) / 1e9, 2), gbs=round(v['bytes'] / max(v['ms'], 1e-9) / 1e6, 1))
                     for k, v in sorted(kern.items(), key=lambda kv: -kv[1]['ms'])}
        if args.kernel_breakdown and headline:
            print(json.dumps(breakdown, indent=1), file=sys.stderr)
            shapes = {}
            for r in records:
                k = shapes.setdefault((r['kind'], r['dims']), [0, 0.0, 0.0, 0.0])
                k[0] += 1; k[1] += r['ms']; k[2] += r['flops']; k[3] += r['bytes']
            print('top launches by total time (kind, dims): launches, ms/step, avg us, TFLOP/s, GB/s', file=sys.stderr)
            for (kind, dims), (cnt, ms, fl, by) in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:90]:
                print(f'  {kind:13s} {str(dims):52s} {cnt:5d} {ms / steps:8.3f} {ms / cnt * 1e3:9.1f} {fl / ms / 1e9:8.1f} {by / ms / 1e6:8.1f}', file=sys.stderr)
        imgs = steps * global_batch
        dtype = 'f32' if nfp == 0 else wl['dtype']          # num_fp16_res 0 = fp32 storage everywhere (convolutions as split-bf16 MFMA products, fp32 accumulate)
        is_headline_metric = name == 'sg2ada' and res == 256 and dtype == 'bf16'
        out = {
            'metric': 'images/sec (G+D step) StyleGAN2-ADA 256x256 bf16' if is_headline_metric else f'images/sec (G+D step) {name} {res}x{res} {dtype}',
            'value': round(imgs / elapsed, 2), 'unit': 'img/s',
            'n_gpus': world, 'steps': steps, 'warmup': max(warmup, 1), 'ms_per_step': round(elapsed / steps * 1e3, 2),
            'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None, 'dtype': dtype, 'data': 'synthetic',
            'config': {'workload': f'{wl["label"]}, global batch {global_batch} = {world} rank(s) x {batch // batch_gpu} round(s) x batch_gpu {batch_gpu}'
                                   + (' (the rounds of a phase evaluated in one pass, minibatch-std groups and loss those of the separate rounds), ' if one_pass else ', ')
                                   + ((f'bf16 from 8x8 up (num_fp16_res {_bf16_blocks(res)}; the reference recipe defaults to 4), conv_clamp 256, ' if nfp is None else
                                       f'bf16 in the {nfp} highest resolutions (num_fp16_res {nfp}; secondary measurement), fp32 storage below, conv_clamp 256, ')
                                      if dtype == 'bf16' else 'fp32 storage, ')
                                   + ('ADA off' if ada is None else f'ADA bgc on (p0 = {ada}, target 0.6; secondary measurement)'),
                       'global_batch': global_batch, 'per_rank_batch': batch, 'parallelism': f'dp{world}'},
            'roofline': roofline, 'target_kernel': target,
            'ms_per_step_median': round(step_ms[len(step_ms) // 2], 2),
            'step_tflops': round(total_flops / (elapsed * 1e3) / 1e9, 1),
            'kernel_ms_per_step': {k: v['ms_per_step'] for k, v in breakdown.items()},
            'sbg_kernel_time_frac_of_step': round(total_ms / (elapsed * 1e3), 3),
            'nonfinite_grads': sum(nonfinite.values()),
        }
        if comm is not None:
            out['comm'] = comm
    eng.close()
    del eng, real_u8
    torch.cuda.empty_cache()
    return out


def slim(rec):
    """a secondary measurement inside the headline line: the numbers, not the per-kernel tables"""
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'ms_per_step_median', 'scaling', 'dtype', 'config', 'step_tflops',
            'nonfinite_grads', 'comm')
    out = {k: rec[k] for k in keep if k in rec}
    if rec.get('roofline'):
        out['roofline_frac'] = rec['roofline']['frac']
    if rec.get('target_kernel'):
        out['target_kernel_mfma_frac'] = rec['target_kernel']['mfma_frac']
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--workload', default='sg2ada', choices=['sg2ada', 'ffhq_sg2', 'sg2attent', 'big_gan'])
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong', 'both'],
                    help="weak (default; the mode the >= 6x at 8 GPUs target is stated for): per-rank batch fixed; strong: global batch fixed; "
                         "both: the weak line with the strong-scaling run in its `secondary` list")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='headline workload: skip the reference-recipe precision run (num_fp16_res 4) after the headline loop')
    ap.add_argument('--res', type=int, default=None, help='resolution override (sg2attent: 32 or 256; otherwise debug only)')
    ap.add_argument('--batch', type=int, default=None, help='per-rank batch (weak) / global batch (strong); default: the config\'s')
    ap.add_argument('--batch-gpu', type=int, default=None)
    ap.add_argument('--ada', type=float, default=None, metavar='P', help="secondary measurement: 'bgc' ADA pipe on, starting strength P (headline = off)")
    ap.add_argument('--num-fp16-res', type=int, default=None, metavar='K',
                    help='secondary measurement: reduced precision in the K highest resolutions only (the reference recipe: 4; 0 = fp32 storage everywhere)')
    ap.add_argument('--kernel-breakdown', action='store_true', help='print the per-kernel launch log summary to stderr')
    ap.add_argument('--single-thread-autograd', action='store_true',
                    help='run backward on the calling thread (profiling under rocprofv3 counter collection: see DESIGN.md, "queue interception")')
    ap.add_argument('--launch-log', default=None, metavar='FILE', help='write the launch log of the timed region (one JSON record per launch, in launch order)')
    ap.add_argument('--rendezvous-only', action='store_true', help='start the ranks, count them with an all-reduce, print that and stop (launcher test; no GPU work)')
    ap.add_argument('--cpu-worker', nargs=2, type=int, default=None, metavar=('BATCH', 'THREADS'), help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_worker is not None:
        cpu_worker(args.cpu_worker[0], args.cpu_worker[1], res=args.res or RES)
        return
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    world, rank, local_rank, backend, ranks_seen = init_ranks(args)
    if args.rendezvous_only:
        if rank == 0:
            print(json.dumps(dict(ranks_seen=ranks_seen, n_gpus=world, backend=backend)))
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm device (the hot path has no CPU fallback)')
    n_dev = torch.cuda.device_count()
    local = local_rank % n_dev
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)

    if args.single_thread_autograd:
        torch.autograd.set_multithreading_enabled(False)
    assert args.num_fp16_res is None or args.workload == 'sg2ada', '--num-fp16-res: headline workload only'

    import style_big_gan_amd  # noqa: F401
    from style_big_gan_amd import _lib
    _lib.load()
    first_scaling = 'weak' if args.scaling == 'both' else args.scaling
    common = dict(res=args.res, ada=args.ada, batch_arg=args.batch, batch_gpu_arg=args.batch_gpu)
    out = measure(args, device, world, rank, args.workload, nfp=args.num_fp16_res, scaling=first_scaling, steps=args.steps, **common)

    secondary = []
    plain_headline = args.workload == 'sg2ada' and args.num_fp16_res is None and args.ada is None and args.res is None
    if args.scaling == 'both' and world > 1:
        rec = measure(args, device, world, rank, args.workload, nfp=args.num_fp16_res, scaling='strong', steps=args.steps, headline=False, **common)
        if rank == 0:
            secondary.append(slim(rec))
    if plain_headline and not args.no_secondary and world == 1:
        # the reference recipe's precision split (bf16 in the 4 highest resolutions, fp32 storage below) on the same clock, same process
        rec = measure(args, device, world, rank, 'sg2ada', nfp=4, scaling=first_scaling, steps=args.steps, headline=False, **common)
        if rank == 0:
            secondary.append(slim(rec))

    if rank == 0:
        out['ranks_seen'] = ranks_seen
        out['scaling_note'] = ('weak: every rank runs the config\'s batch (global batch x N); strong: the config\'s global batch split over the ranks '
                               '(reference trainers.py:524).  The >= 6x at 8 GPUs target of BASELINE.json is stated for the WEAK line.')
        cpu = None
        if world == 1 and not args.no_cpu_baseline and args.workload == 'sg2ada':
            cpu = cpu_baseline(res=args.res or RES)
        out['cpu_baseline'] = cpu
        if secondary:
            out['secondary'] = secondary
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
