"""Condense the rocprofv3 passes of profiles/collect.sh into the files committed under profiles/:
   <tag>_kernel_stats.csv    per-kernel totals of the --stats pass (name, calls, total / average ns, share)
   <tag>_kernel_shapes.csv   per (kernel, launch geometry, SHAPE) summary of the --kernel-trace pass: launches, avg / min / max ns and the
                             TFLOP/s that follows from the launch's algorithmic flops.  rocprofv3 knows names and grids; a persistent kernel
                             has ONE grid for every shape, so the shape comes from bench.py's own launch log of the same run (one record per
                             launch, in launch order): the k-th logged launch of a kernel family is the k-th dispatch of that family among the
                             last len(log) dispatches of the trace (one stream, in order).  Durations are rocprofv3's, not the library's.
   <tag>_traffic.json        per-kernel HBM bytes per launch from the PMC passes: 2 x FETCH_SIZE + WRITE_SIZE, in KiB -> bytes
                             (gfx950: FETCH_SIZE reports half of a wide coalesced read stream, MI355X_MICROARCH.md section HBM)
   <tag>_sq_counters.json    per (kernel, shape) SQ / GRBM counters per launch and the matrix-pipe utilisation derived from them:
                             SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs)  [GRBM_GUI_ACTIVE is summed over the 8 XCDs], next to
                             the figure the duration implies (flops / 1024 per SIMD-cycle at 2.4 GHz = 2.5 PFLOP/s dense bf16)."""
import collections, csv, glob, json, os, shutil, sys

out, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
FAMILIES = ('conv_thin_kernel', 'conv_up2_kernel', 'conv_halo8_kernel', 'upfirdn2d_fir_slide_kernel', 'conv_halo_ld_kernel', 'conv_gather_ld_kernel', 'conv_k64_kernel', 'conv_ksplit_reduce', 'upfirdn2d_fir_mfma_kernel',
            'upfirdn2d_fir_fixed_kernel', 'modconv_bwd_kernel', 'conv_igemm_dma_kernel', 'conv_igemm_kernel', 'conv_wgrad_rows_kernel', 'conv_wgrad_halo_kernel',
            'conv_wgrad_kernel', 'wgrad_reduce_kernel', 'upfirdn2d_fir_kernel', 'upfirdn2d_kernel', 'attention_bwd', 'attention_fwd', 'mbstd', 'bias_act', 'scale_nc', 'dot_hw')


def find(sub, suffix):
    hits = glob.glob(os.path.join(out, sub, '**', f'*{suffix}'), recursive=True)
    return hits[0] if hits else None


def short(name):
    for key in FAMILIES:
        if key in name:
            return key
    return name[:60]


def log_family(rec):
    """kernel family a launch-log record was served by (dims[6] carries the launcher's code, see the SbgProfScope calls in csrc/)"""
    code = rec['dims'][6]
    if rec['kind'] == 'conv_igemm':
        top, rem = code // 1000000, code % 1000000
        if top == 3:
            return 'conv_halo_ld_kernel'
        if top == 1:
            return 'conv_k64_kernel'
        if top == 6 and rem < 1000:
            return 'conv_thin_kernel'
        if top == 9:
            return 'conv_up2_kernel'
        if top == 5:
            return 'conv_halo8_kernel'
        return 'conv_gather_ld_kernel' if top >= 4 else None
    if rec['kind'] == 'conv_wgrad':
        return 'conv_wgrad_rows_kernel' if code // 1000000 == 1 else None      # (3xxxxxx: the thin kernel; the generic weight-gradient kernel launches once per tap group: not joined)
    return None


def load_log(path):
    return [json.loads(l) for l in open(path)] if path and os.path.exists(path) else []


def shapes_of(trace_rows, log):
    """-> {dispatch_id: (family, dims tuple, flops)} for the dispatches the launch log accounts for"""
    by_family = collections.defaultdict(list)
    for r in log:
        f = log_family(r)
        if f:
            by_family[f].append(r)
    res = {}
    for fam, recs in by_family.items():
        disp = [r for r in trace_rows if short(r['Kernel_Name']) == fam]
        disp.sort(key=lambda r: int(r['Dispatch_Id']))
        if len(disp) < len(recs):
            print(f'warning: {fam}: {len(recs)} logged launches but only {len(disp)} dispatches in the trace', file=sys.stderr)
            continue
        for d, rec in zip(disp[-len(recs):], recs):
            res[int(d['Dispatch_Id'])] = (fam, tuple(rec['dims']), rec['flops'])
    return res


stats = find('stats', 'kernel_stats.csv')
if stats:
    shutil.copy(stats, os.path.join(here, f'{tag}_kernel_stats.csv'))

# ---- per-shape summary of the kernel trace
trace = find('stats', 'kernel_trace.csv')
log = load_log(os.path.join(out, 'launch_log.jsonl'))
if trace:
    rows = list(csv.DictReader(open(trace)))
    shape = shapes_of(rows, log)
    groups = collections.defaultdict(list)
    for r in rows:
        did = int(r['Dispatch_Id'])
        fam, dims, flops = shape.get(did, (short(r['Kernel_Name']), None, 0.0))
        key = (fam, '%sx%sx%s' % (r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z']), r['Workgroup_Size_X'], r.get('LDS_Block_Size', r.get('Group_Segment_Size', '')), dims)
        groups[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp']), flops))
    with open(os.path.join(here, f'{tag}_kernel_shapes.csv'), 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'grid_work_items', 'workgroup', 'lds_bytes', 'shape_from_launch_log', 'launches', 'total_ns', 'avg_ns', 'min_ns', 'max_ns', 'tflops_at_avg'])
        for key, v in sorted(groups.items(), key=lambda kv: -sum(d for d, _ in kv[1]))[:160]:
            ds = [d for d, _ in v]
            fl = v[0][1]
            w.writerow([key[0], key[1], key[2], key[3], '' if key[4] is None else ' '.join(map(str, key[4])), len(ds), sum(ds), round(sum(ds) / len(ds)), min(ds), max(ds),
                        round(fl / (sum(ds) / len(ds)) / 1e3, 1) if fl else ''])

# ---- HBM traffic
per = collections.defaultdict(lambda: dict(launches=0, fetch_kib=0.0, write_kib=0.0))
for sub, field in (('fetch', 'fetch_kib'), ('write', 'write_kib')):
    f = find(sub, 'counter_collection.csv')
    if not f:
        continue
    seen = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        per[k][field] += float(r['Counter_Value'])
        seen[k] += 1
    for k, n in seen.items():
        per[k]['launches'] = max(per[k]['launches'], n)
res = {}
for k, v in per.items():
    n = max(v['launches'], 1)
    res[k] = dict(launches=v['launches'], fetch_bytes_per_launch=round(2 * v['fetch_kib'] * 1024 / n), write_bytes_per_launch=round(v['write_kib'] * 1024 / n),
                  hbm_bytes_per_launch=round((2 * v['fetch_kib'] + v['write_kib']) * 1024 / n))
complete = bool(res) and all(find(sub, 'counter_collection.csv') for sub in ('fetch', 'write'))
if complete:        # a traffic file is only written when BOTH TCC passes completed (bench.py reads the newest one)
    json.dump(dict(command='python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline', formula='(2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch', kernels=res),
              open(os.path.join(here, f'{tag}_traffic.json'), 'w'), indent=1, sort_keys=True)
else:
    print('traffic: FETCH_SIZE / WRITE_SIZE passes incomplete -- no traffic file written', file=sys.stderr)

# ---- SQ counters per (kernel, shape)
f = find('sq', 'counter_collection.csv')
if f:
    rows = list(csv.DictReader(open(f)))
    disp = {}
    for r in rows:
        d = disp.setdefault(int(r['Dispatch_Id']), dict(Kernel_Name=r['Kernel_Name'], Dispatch_Id=r['Dispatch_Id'], counters={}))
        d['counters'][r['Counter_Name']] = d['counters'].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    shape = shapes_of(list(disp.values()), load_log(os.path.join(out, 'launch_log_sq.jsonl')))
    agg = collections.defaultdict(lambda: dict(launches=0, flops=0.0, counters=collections.Counter()))
    for did, d in disp.items():
        fam, dims, flops = shape.get(did, (short(d['Kernel_Name']), None, 0.0))
        a = agg[(fam, dims)]
        a['launches'] += 1; a['flops'] += flops
        for k, v in d['counters'].items():
            a['counters'][k] += v
    outj = []
    for (fam, dims), a in sorted(agg.items(), key=lambda kv: -kv[1]['counters'].get('GRBM_GUI_ACTIVE', 0)):
        n = a['launches']
        c = {k: v / n for k, v in a['counters'].items()}
        e = dict(kernel=fam, shape=None if dims is None else list(dims), launches=n, per_launch={k: round(v) for k, v in c.items()})
        gui = c.get('GRBM_GUI_ACTIVE', 0) / 8.0
        if gui > 0 and 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
            e['mfma_busy_frac_of_simd_cycles'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024), 4)
        if a['flops'] and 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
            e['algorithmic_flops_per_launch'] = a['flops'] / n
            e['mfma_busy_cycles_expected_for_flops'] = round(a['flops'] / n / 1024)      # 16x16x32 bf16: 16 cycles per 16384 flops = 1024 flops per SIMD-cycle
        if c.get('SQ_LDS_IDX_ACTIVE'):
            e['lds_bank_conflict_frac'] = round(c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE'], 4)
        outj.append(e)
    json.dump(dict(command='python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline', note='counters are sums over all SEs / XCDs; kernels run serialised under counter collection',
                   kernels=outj[:80]), open(os.path.join(here, f'{tag}_sq_counters.json'), 'w'), indent=1)
b = os.path.join(out, 'bench_under_rocprof.json')
if os.path.exists(b):
    shutil.copy(b, os.path.join(here, f'{tag}_bench_under_rocprof.json'))
print(json.dumps(res, indent=1)[:2000])
