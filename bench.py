#!/usr/bin/env python3
"""Headline benchmark: images/sec of the StyleGAN2-ADA G+D training step at 256x256, bf16, on N MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one iteration of the reference's training loop body (train_parts/trainers.py:711-765) on synthetic inputs:
phases Gmain and Dmain every iteration and Dreg (R1, lazy, gain 4) every 4th, each phase = zero_grad -> accumulation rounds of
forward/backward -> gradient all-reduce -> nan_to_num -> Adam, then the G_ema update.  Workload = configs/sg2ada.yaml at
256x256: z = w = 512, 2 mapping layers, channel_base 32768, D architecture 'orig', mbstd group 32, softplus loss, R1 gamma 0.01,
style mixing 0, batch 64 per rank as 2 rounds of batch_gpu 32 (weak scaling: the global batch is 64 x N), every block from 8x8
up in bf16 (num_fp16_res = 7, conv_clamp = 256 -- the reference's mixed-precision recipe with bf16 in place of fp16), ADA off.
Latents ~ N(0, 1), reals uint8 U[0, 255] generated on the device once; random-init weights (no network for datasets).

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field meanings).  `roofline` is measured live: the library
brackets every kernel launch of the timed region with hipEvents on the launch stream (sbg_prof_*), and the dominant kernel's
algorithmic flops / summed duration is reported against the dense bf16 MFMA peak.  `cpu_baseline` times the CPU oracle
(oracle/, the fixture-pinned restatement of the reference's eager CPU path) on a bounded sample of the same workload.
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


def sg2ada_kwargs(res=RES, num_fp16_res=7, conv_clamp=256, channel_base=32768, mbstd=32):
    gk = dict(z_dim=Z_DIM, c_dim=0, w_dim=512, img_resolution=res, img_channels=3, mapping_kwargs=dict(num_layers=2),
              synthesis_kwargs=dict(channel_base=channel_base, num_fp16_res=num_fp16_res, block_kwargs=dict(conv_clamp=conv_clamp)))
    dk = dict(c_dim=0, img_resolution=res, img_channels=3, architecture='orig', channel_base=channel_base, num_fp16_res=num_fp16_res,
              conv_clamp=conv_clamp, epilogue_kwargs=dict(mbstd_group_size=mbstd))
    return gk, dk


def build_engine(device, world_size, rank, batch=BATCH, batch_gpu=BATCH_GPU, res=RES, ada=None):
    from style_big_gan_amd.train_parts import trainers
    gk, dk = sg2ada_kwargs(res=res)
    return trainers.StepEngine(device, generator='sg2_classic', discriminator='sg2_classic', gen_kwargs=gk, disc_kwargs=dk,
                               loss_arch='sg2', loss='softplus', loss_arch_kwargs=dict(style_mixing_prob=0),
                               gen_regs=[], dis_regs=[('r1', dict(r1_gamma=0.01))],
                               optim_gen=('adam', dict(lr=0.0025, betas=[0, 0.99], eps=1e-8)),
                               optim_disc=('adam', dict(lr=0.0025, betas=[0, 0.99], eps=1e-8)),
                               g_reg_interval=16, d_reg_interval=4, batch=batch, batch_gpu=batch_gpu,
                               ema_kimg=500, ema_rampup=0.05, world_size=world_size, rank=rank, seed=0, **ada_kwargs(ada))


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
PMC_KERNELS = {'conv_igemm': ('conv_halo_ld_kernel', 'conv_gather_ld_kernel', 'conv_k64_kernel', 'conv_igemm_dma_kernel', 'conv_igemm_kernel'),
               'conv_wgrad': ('conv_wgrad_rows_kernel', 'conv_wgrad_kernel'), 'upfirdn2d': ('upfirdn2d_fir', 'upfirdn2d_kernel'),
               'bias_act': ('bias_act',), 'scale_nc': ('scale_nc',), 'dot_hw': ('dot_hw',)}


def pmc_traffic(kind):
    """HBM bytes per launch of `kind` from the newest profiles/*_traffic.json: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 of separate
    rocprofv3 --pmc passes over this same bench command (profiles/collect.sh; PMC counters cannot be read from inside the
    process).  Returns (bytes_per_launch, source) or (None, None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_traffic.json')), key=os.path.getmtime)
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


def cpu_baseline(sample_batch=2, res=RES):
    """One Gmain + Dmain + Dreg pass of the CPU oracle on `sample_batch` images at the benchmark's shapes (fp32, all host
    threads); returns img/s of a G+D step with Dreg amortised over 4 iterations, like the GPU figure."""
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
    n = sample_batch
    z_g, z_d = torch.randn(n, Z_DIM), torch.randn(n, Z_DIM)
    real = torch.randint(0, 256, [n, 3, res, res]).float() / 127.5 - 1
    t0 = time.perf_counter()
    ON.gd_step_grads(gsd, dsd, cfg, z_g, z_d, real, r1_gamma=None, noise_mode='const')         # Gmain + Dmain
    t_main = time.perf_counter() - t0
    t0 = time.perf_counter()
    import torch.nn.functional as F     # Dreg alone: D forward on reals, R1 double backward
    d_leaf = {k: v.clone().requires_grad_('resample' not in k) for k, v in dsd.items()}
    real_in = real.clone().requires_grad_(True)
    logits = ON.discriminator(d_leaf, real_in, torch.zeros(n, 0), cfg)
    r1 = torch.autograd.grad(logits.sum(), real_in, create_graph=True)[0]
    pen = (r1.square().sum([1, 2, 3]) * 0.005).mean()
    torch.autograd.grad(pen, [v for v in d_leaf.values() if v.requires_grad], allow_unused=True)
    t_reg = time.perf_counter() - t0
    sec_per_img = (t_main + t_reg / 4) / n
    return dict(value=round(1.0 / sec_per_img, 4), unit='img/s', cores=torch.get_num_threads(), kind='port',
                sample=f'oracle/ (CPU restatement of the reference eager path, fp32), one Gmain+Dmain ({t_main:.1f}s) + one Dreg ({t_reg:.1f}s, /4) '
                       f'on batch {n} at {res}x{res}, sg2ada shapes')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--res', type=int, default=RES, help='debug only; the benchmark is 256')
    ap.add_argument('--batch', type=int, default=BATCH)
    ap.add_argument('--batch-gpu', type=int, default=BATCH_GPU)
    ap.add_argument('--ada', type=float, default=None, metavar='P', help="secondary measurement: 'bgc' ADA pipe on, starting strength P (headline = off)")
    ap.add_argument('--kernel-breakdown', action='store_true', help='print the per-kernel launch log summary to stderr')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm device (the hot path has no CPU fallback)')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = os.environ.get('SBG_DIST_BACKEND', 'nccl')        # 'gloo' only to rehearse N > 1 on a one-GPU box
        torch.distributed.init_process_group(backend, **(dict(device_id=device) if backend == 'nccl' else {}))
    assert world == args.gpus or world == 1, f'launched with WORLD_SIZE={world} but --gpus {args.gpus}'

    import style_big_gan_amd
    from style_big_gan_amd import _lib
    _lib.load()
    eng = build_engine(device, world, rank, batch=args.batch, batch_gpu=args.batch_gpu, res=args.res, ada=args.ada)
    gen = torch.Generator(device=device); gen.manual_seed(1234 + rank)
    real_u8 = torch.randint(0, 256, [args.batch, 3, args.res, args.res], device=device, dtype=torch.uint8, generator=gen)

    def step():
        real = real_u8.to(torch.float32) / 127.5 - 1
        eng.train_iteration(real, None)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    eng.batch_idx = 0       # the timed region starts on an iteration that runs Dreg: K steps contain ceil(K / 4) Dreg phases
    barrier()
    _lib.prof_enable(True)
    _lib.prof_fetch()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.prof_enable(False)
    records = _lib.prof_fetch()

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
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
        if roofline is not None:
            roofline['traffic'], src = pmc_traffic(dom)
            if src:
                roofline['traffic_source'] = src + ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, bytes per launch averaged over the same launches)'
                roofline['algorithmic_bytes_per_launch'] = round(kern[dom]['bytes'] / kern[dom]['launches'])
        # the 256x256 modulated 3x3 conv (M = N*65536 pixels, Cout 128, K = 9*128): the kernel the 40 % MFMA target names
        tgt = [r for r in records if r['kind'] == 'conv_igemm' and r['dims'][1] == 128 and r['dims'][2] == 128 and r['dims'][3] == 9
               and r['dims'][0] == args.batch_gpu * args.res * args.res]
        target = None
        if tgt:
            fl, ms = sum(r['flops'] for r in tgt), sum(r['ms'] for r in tgt)
            target = dict(shape=f'[{args.batch_gpu},128,{args.res},{args.res}] (*) [128,128,3,3]', launches=len(tgt),
                          avg_launch_ms=round(ms / len(tgt), 4), tflops=round(fl / ms / 1e9, 2), mfma_frac=round(fl / ms / 1e9 / MFMA_BF16_PEAK_TFLOPS, 4))
        total_ms = sum(v['ms'] for v in kern.values())
        breakdown = {k: dict(launches=v['launches'], ms_per_step=round(v['ms'] / args.steps, 3),
                             tflops=round(v['flops'] / max(v['ms'], 1e-9) / 1e9, 2), gbs=round(v['bytes'] / max(v['ms'], 1e-9) / 1e6, 1))
                     for k, v in sorted(kern.items(), key=lambda kv: -kv[1]['ms'])}
        if args.kernel_breakdown:
            print(json.dumps(breakdown, indent=1), file=sys.stderr)
            shapes = {}
            for r in records:
                k = shapes.setdefault((r['kind'], r['dims']), [0, 0.0, 0.0, 0.0])
                k[0] += 1; k[1] += r['ms']; k[2] += r['flops']; k[3] += r['bytes']
            print('top launches by total time (kind, dims): launches, ms/step, avg us, TFLOP/s, GB/s', file=sys.stderr)
            for (kind, dims), (cnt, ms, fl, by) in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:90]:
                print(f'  {kind:13s} {str(dims):52s} {cnt:5d} {ms / args.steps:8.3f} {ms / cnt * 1e3:9.1f} {fl / ms / 1e9:8.1f} {by / ms / 1e6:8.1f}', file=sys.stderr)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(res=args.res)
        imgs = args.steps * args.batch * world
        out = {
            'metric': 'images/sec (G+D step) StyleGAN2-ADA 256x256 bf16', 'value': round(imgs / elapsed, 2), 'unit': 'img/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 2),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': f'configs/sg2ada.yaml @ {args.res}x{args.res}: sg2_classic G (skip) + D (orig), softplus + R1(0.01)/4, '
                                   f'batch {args.batch}/rank = {args.batch // args.batch_gpu} x batch_gpu {args.batch_gpu}, num_fp16_res 7 (bf16), conv_clamp 256, '
                                   + ('ADA off' if args.ada is None else f'ADA bgc on (p0 = {args.ada}, target 0.6; secondary measurement)'),
                       'global_batch': args.batch * world, 'parallelism': f'dp{world}'},
            'roofline': roofline, 'target_kernel': target, 'cpu_baseline': cpu,
            'kernel_ms_per_step': {k: v['ms_per_step'] for k, v in breakdown.items()},
            'sbg_kernel_time_frac_of_step': round(total_ms / (elapsed * 1e3), 3),
        }
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
