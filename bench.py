#!/usr/bin/env python3
"""Throughput of the IST-GCN hot path on MI355X: skeleton-clips/sec, forward + backward (+ SGD step).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|f32] [--model st_gcn_msgcn] [--batch 64]

N=1 workload = BASELINE.json configs[1]: net/st_gcn_msgcn.py (Inception-GCN), NTU-RGB+D xsub shape
(C=3, T=300, V=25, M=2, 60 classes), batch 64 clips per GPU, training step of processor/recognition.py:249-296
(train mode, dropout 0.5, CrossEntropy, SGD-nesterov) on synthetic clips and random-init weights.
N>1: one process per GPU, the batch axis sharded (64 clips per GPU, weak scaling), one flat-bucket gradient all-reduce
per step over RCCL.  Either launched by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`, or
run directly (`python bench.py --gpus N`): with no WORLD_SIZE in the environment this process starts that launcher
itself as a child (before anything touches the GPU), relays rank 0's JSON line and exits with the child's code.
ISTGCN_DIST_BACKEND=gloo rehearses the multi-rank path with several ranks sharing the visible GPU(s).

One JSON line on rank 0, with
  roofline      the dominant kernel family of the step: algorithmic FLOPs (or bytes) of its launches / their summed
                durations, HIP events on the launch stream inside the timed region (ops.PROFILE)
  cpu_baseline  the oracle (CPU restatement of the reference, `oracle/stgcn_ref.py`) timed on this box's host cores on
                a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK = {'mfma_f32': 157.3, 'mfma_bf16': 2500.0, 'hbm': 8000.0}       # TFLOP/s, TFLOP/s, GB/s  (MI355X_MICROARCH.md)
MODELS = {
    'st_gcnold': (dict(layout='ntu-rgb+d', strategy='spatial'), 60, 25),
    'st_gcn_msgcn': (dict(layout='ntu-rgb+d', strategy='spatial_3'), 60, 25),
    'st_gcn_mstcn_1x1': (dict(layout='openpose', strategy='spatial'), 400, 18),
    'st_gcn_multi3_fix_3A_mstcn': (dict(layout='ntu-rgb+d', strategy='spatial_3'), 60, 25),
    'st_gcn_mstcn_1x1_deep': (dict(layout='ntu-rgb+d', strategy='spatial'), 60, 25),
}
FAMILY = {'istgcn_tconv': 'tconv (temporal conv fwd + data-grad, MFMA implicit GEMM)',
          'istgcn_tconv_wgrad': 'tconv_wgrad', 'istgcn_gcn_fwd': 'gcn_fwd (graph conv fwd + data-grad)',
          'istgcn_gcn_wgrad': 'gcn_wgrad', 'istgcn_gcn_bwd_data': 'gcn_bwd_data (graph conv data + adjacency gradient)', 'istgcn_block_out_fwd': 'block_out_fwd', 'istgcn_block_out_bwd': 'block_out_bwd',
          'istgcn_affine2': 'bn_bwd_apply',
          'istgcn_bneck': 'bneck_in + bneck_out (bottleneck chain fwd + data-grad, register-chained streams)',
          'istgcn_bneck_wgrad': 'bneck_wgrad (+ taps: bottleneck weight gradients)'}


PMC_KEY = {'istgcn_tconv': 'tconv kernels', 'istgcn_tconv_wgrad': 'tconv_wgrad kernels', 'istgcn_gcn_fwd': 'gcn_rc_fwd_kernel',
           'istgcn_gcn_bwd_data': 'gcn_rc_bwd_kernel', 'istgcn_gcn_wgrad': 'gcn_rc_wgrad_kernel',
           'istgcn_block_out_fwd': 'block_out_fwd_kernel', 'istgcn_block_out_bwd': 'block_out_bwd_kernel',
           'istgcn_affine2': 'affine2_kernel', 'istgcn_bneck': 'bneck_in/out kernels', 'istgcn_bneck_wgrad': 'bneck_wgrad kernels'}
# BASELINE.json configs[i-1] -> (model, storage type, clips per GPU): the reference's own arithmetic per config (fp32 for
# configs 1/3/4 -- SURVEY 8a; config 2 bf16, config 5 fp16); --dtype / --batch on the command line override.
CONFIGS = {1: ('st_gcnold', 'f32', 2), 2: ('st_gcn_msgcn', 'bf16', 64), 3: ('st_gcn_mstcn_1x1', 'f32', 256),
           4: ('st_gcn_multi3_fix_3A_mstcn', 'f32', 64), 5: ('st_gcn_mstcn_1x1_deep', 'f16', 128)}


# Whole-step roofline (measured there with FlopCounterMode on the reference): GFLOP per clip forward + backward
# (= 3 x forward) as the REFERENCE executes them, and the compulsory HBM bytes per clip of the whole st_gcn block chain under
# the 3-kernel train-BN fusion model, forward, per byte of element size (x 3 for forward + backward).
STEP_WORK = {'st_gcnold': (102.6, 0.277 / 4), 'st_gcn_msgcn': (112.9, 0.138 / 2), 'st_gcn_mstcn_1x1': (22.4, 0.199 / 4),
             'st_gcn_multi3_fix_3A_mstcn': (258.9, 0.277 / 4), 'st_gcn_mstcn_1x1_deep': (84.9, 0.357 / 2)}


def vendor_gemm_tflops(dev, seconds=0.25):
    """What the vendor GEMM (hipBLASLt through torch.matmul) sustains on THIS box, bf16, 8192^3, random operands: the
    practical ceiling of an MFMA-bound kernel next to the 2.5 PFLOP/s dense peak the roofline divides by (the clock the
    chip holds under matrix load is data dependent: all-zero operands run 1.3x faster, tools/gemm_peak.py)."""
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 0
    e0.record()
    t0 = time.perf_counter()
    while it < 20 or (time.perf_counter() - t0 < seconds and it < 400):
        torch.matmul(a, b)
        it += 1
    e1.record()
    torch.cuda.synchronize()
    return round(2.0 * 8192 ** 3 * it / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1)


def pmc_file(model, dtype, batch):
    """Counter summary of tools/profile_bench.sh for exactly this workload, or None: the newest round's file wins."""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_%s_%s_b%d_pmc.json' % (model, dtype, batch))))
    return hits[-1] if hits else None


def cpu_baseline(model_tag, T, seconds_budget=25.0):
    """Oracle train step on the host cores: 1 warm-up + up to 3 timed steps of batch 8 (bounded sample)."""
    from oracle import stgcn_ref as R
    gargs, nc, V = MODELS[model_tag]
    # 16 threads: fastest of {8,16,32,64} on the MI355X box's 256-core host for this model (tools/cpu_thread_sweep.py:
    # 1.48 / 1.91 / 1.82 / 0.93 clips/s); torch's default of one thread per core (256) is 20x slower than that.
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    B = 8
    m = R.RefModel(model_tag, 3, nc, gargs, True, dropout=0.5)
    R.weights_init_(m, seed=0)
    opt = R.make_optimizer(m)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 3, T, V, 2, generator=g)
    y = torch.randint(0, nc, (B,), generator=g)
    R.train_step(m, opt, x, y)                         # warm-up
    t0, n = time.time(), 0
    while n < 3 and (n == 0 or time.time() - t0 < seconds_budget):
        R.train_step(m, opt, x, y)
        n += 1
    dt = (time.time() - t0) / n
    return {'value': round(B / dt, 3), 'unit': 'clips/s', 'cores': cores, 'kind': 'port',
            'sample': '%d timed steps (1 warm-up) of batch %d, same model / clip shape (T=%d,V=%d,M=2), fp32, torch CPU, '
                      'dropout 0.5, SGD step included' % (n, B, T, V)}


def self_launch(n):
    """`python bench.py --gpus N` run directly: start N ranks through torch.distributed.run as a CHILD process (this
    process has not touched the GPU), pass the command line through, relay the child's output, return its exit code."""
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    # a rank that dies (or hangs in a collective) must not leave the launcher waiting forever: bounded run, own process
    # group so that the whole tree can be ended, non-zero exit either way
    limit = float(os.environ.get('ISTGCN_BENCH_TIMEOUT', '1500'))
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return proc.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        import signal
        sys.stderr.write('bench.py: the %d-rank run exceeded %.0f s; ending its process group\n' % (n, limit))
        try:
            os.killpg(proc.pid, signal.SIGTERM)
            proc.wait(timeout=20)
        except Exception:
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except Exception:
                pass
        return 124


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', type=int, default=None, choices=sorted(CONFIGS),
                    help='BASELINE.json configs[i-1]: model, storage type and clips per GPU of that configuration (default: 2)')
    ap.add_argument('--dtype', default=os.environ.get('ISTGCN_BENCH_DTYPE'), choices=['bf16', 'f16', 'f32'])
    ap.add_argument('--model', default=None, choices=sorted(MODELS))
    ap.add_argument('--batch', type=int, default=None, help='clips per GPU')
    ap.add_argument('--frames', type=int, default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--h2d', action='store_true',
                    help='feed every timed step from HOST memory through harness.DeviceStager (pinned double buffer, copy on a side\n'
                         'stream under the previous step): the PCIe-inclusive rate.  Never the headline `value` (inputs resident in HBM).')
    ap.add_argument('--no-vendor-gemm', action='store_true', help='skip the 0.3 s hipBLASLt reference measurement (16-bit runs)')
    ap.add_argument('--breakdown', action='store_true', help='print a per-kernel-family table to stderr')
    ap.add_argument('--graph', action='store_true',
                    help='replay forward+backward from one hipGraph (harness.GraphedStep) in the timed region.  Measured: no gain\n'
                         '(19.02 vs 19.09 ms/step, bf16 batch 64) -- the host already runs ahead of the GPU, the ~5 us between\n'
                         'dependent kernels is device-side -- so the default stays eager, where every launch of the dominant\n'
                         'family carries HIP events inside the timed region')
    ap.add_argument('--loss-scale', type=float, default=None,
                    help='static loss scale of the backward pass (default: 65536 for f16 storage, 1 otherwise)')
    args = ap.parse_args()
    cm, cd, cb = CONFIGS[args.config or 2]
    args.model = args.model or cm
    if args.config is None and args.model != cm:
        cd, cb = ('f16', 128) if args.model.endswith('deep') else ('bf16', 64)
    args.dtype = args.dtype or cd
    args.batch = args.batch or cb

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit('bench.py needs an MI355X (no CPU fallback for the product path)')
    ndev = torch.cuda.device_count()
    if local >= ndev and os.environ.get('ISTGCN_DIST_BACKEND', 'nccl') != 'nccl':
        local = local % ndev              # rehearsal: several ranks share the one visible GPU
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # RCCL ('nccl' on ROCm) over xGMI; ISTGCN_DIST_BACKEND=gloo only to rehearse the multi-rank code path on one GPU
        backend = os.environ.get('ISTGCN_DIST_BACKEND', 'nccl')
        # (this image's RCCL prints a version banner on STDOUT at NCCL_DEBUG=VERSION: keep stdout for the one JSON line)
        if os.environ.get('NCCL_DEBUG', '').upper() in ('', 'VERSION'):
            os.environ['NCCL_DEBUG'] = 'WARN'
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import importlib
    import istgcn_amd  # noqa: F401
    from istgcn_amd import ops, harness, dp
    gargs, nc, V = MODELS[args.model]
    T = args.frames or (600 if args.model.endswith('deep') else 300)
    dt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[args.dtype]
    half = dt != torch.float32
    loss_scale = args.loss_scale if args.loss_scale else (65536.0 if dt == torch.float16 else 1.0)
    torch.manual_seed(0)
    model = importlib.import_module('istgcn_amd.net.' + args.model).Model(3, nc, gargs, True, dropout=0.5,
                                                                        compute_dtype=dt)
    model.apply(harness.weights_init)
    model.to(dev).train()
    sync = dp.FlatGradSync(model) if world > 1 else None
    opt = harness.make_optimizer(model, loss_scale=loss_scale)
    if sync is not None:
        opt.attach_sync(sync)          # the all-reduce runs in place on the optimizer's flat gradient buffer
    g = torch.Generator().manual_seed(1234 + rank)
    B = args.batch
    x = torch.randn(B, 3, T, V, 2, generator=g).to(dev)
    y = torch.randint(0, nc, (B,), generator=g).to(dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ISTGCN_TRACE_KERNELS=<file>: the library's dispatch trace around the whole run -> "<config tag>\tlaunches\tkernel symbol"
    # lines appended to <file> (tools/kernel_coverage.py: which kernels of the library the BASELINE configurations reach)
    trace_file = os.environ.get('ISTGCN_TRACE_KERNELS') if rank == 0 else None
    tracer = ops.trace() if trace_file else None
    if tracer:
        tracer.__enter__()
    use_graph = args.graph and not args.breakdown
    # graph mode: EVERY step of this process (eager warm-up, capture, replays, the eager roofline pass) runs on one side
    # stream -- capture is not allowed on the default stream, and autograd's gradient accumulators stay bound to the
    # stream of the first backward pass they saw
    import contextlib
    stack = contextlib.ExitStack()
    side = None
    if use_graph:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        stack.enter_context(torch.cuda.stream(side))

    # warm-up steps are profiled per launch to find the dominant kernel family; in the timed region only THAT family's
    # launches carry HIP events (two event records per launch on all ~350 launches cost ~3 % of a bf16 step)
    ops.PROFILE = []
    for _ in range(args.warmup):
        harness.train_step(model, opt, x, y, sync)
    barrier()
    if not args.breakdown and ops.PROFILE:
        wfam = {}
        for name, _, _, e0, e1 in ops.PROFILE:
            wfam[name] = wfam.get(name, 0.0) + e0.elapsed_time(e1)
        ops.PROFILE_ONLY = max(wfam, key=wfam.get)
    if use_graph:
        # The timed steps replay forward + backward from ONE hipGraph (harness.GraphedStep; gradient all-reduce and the
        # optimizer's one-launch update stay eager): ~250 launches leave the host as one.  A replay has no per-launch
        # hook, so the dominant family's launch durations are taken -- live, same process, same buffers -- from the same
        # number of EAGER steps run right after the timed region, with HIP events on that family's launches only.
        dom_only, ops.PROFILE, ops.PROFILE_ONLY = ops.PROFILE_ONLY, None, None
        gstep = harness.GraphedStep(model, opt, x, y, warmup=1, stream=side)
        for _ in range(2):
            gstep()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = gstep()
        barrier()
        elapsed = time.perf_counter() - t0
        ops.PROFILE, ops.PROFILE_ONLY = [], dom_only
        for _ in range(args.steps):
            harness.train_step(model, opt, x, y, sync)
        barrier()
    elif args.h2d and os.environ.get('ISTGCN_H2D_PLAIN') == '1':
        # what processor/recognition.py:258 does: every batch moved from pageable host memory by a blocking .to(dev) on the
        # compute stream (A/B against the stager; ISTGCN_H2D_PLAIN=1 python bench.py --h2d)
        xh, yh = x.cpu(), y.cpu()
        ops.PROFILE = []
        barrier()
        t0 = time.perf_counter()
        t_copy = 0.0
        for _ in range(args.steps):
            tc = time.perf_counter()
            data, label = xh.to(dev), yh.to(dev)
            t_copy += time.perf_counter() - tc
            loss = harness.train_step(model, opt, data, label, sync)
        barrier()
        elapsed = time.perf_counter() - t0
        if rank == 0:
            sys.stderr.write('h2d plain: host ms per step in .to(dev): %.2f\n' % (t_copy / args.steps * 1e3))
    elif args.h2d:
        # the same K steps, every batch coming from pageable host memory (what a DataLoader hands over)
        xh, yh = x.cpu(), y.cpu()
        ops.PROFILE = []
        barrier()
        t0 = time.perf_counter()
        # (the first `depth` batches allocate the stager's pinned / device slots -- ~50 ms of hipHostMalloc each --: they run
        #  untimed, the clock starts behind them)
        stager = harness.DeviceStager(((xh, yh) for _ in range(args.steps + 3)), dev, depth=3)
        t_issue, nb0 = 0.0, 0
        for k, (data, label) in enumerate(stager):
            if k == 3:
                barrier()
                ops.PROFILE = []
                t0 = time.perf_counter()
                t_issue = 0.0
                tm0 = dict(stager.timers)
            ti = time.perf_counter()
            loss = harness.train_step(model, opt, data, label, sync)
            t_issue += time.perf_counter() - ti
        barrier()
        elapsed = time.perf_counter() - t0
        if rank == 0:
            tm = {k2: stager.timers[k2] - tm0[k2] for k2 in tm0}
            nb = max(1, tm['batches'])
            sys.stderr.write('h2d: host ms per step -- issuing the training step %.2f, checking the slot %.2f, copy into pinned memory '
                             '%.2f, enqueueing the H2D copies %.2f; blocked on a slot %d times\n' % (
                                 t_issue / args.steps * 1e3, tm['wait_slot'] / nb * 1e3, tm['host_copy'] / nb * 1e3,
                                 tm['enqueue'] / nb * 1e3, tm['blocked']))
    else:
        ops.PROFILE = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = harness.train_step(model, opt, x, y, sync)
        barrier()
        elapsed = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    ops.PROFILE_ONLY = None
    stack.close()
    if tracer:
        tracer.__exit__(None, None, None)
        with open(trace_file, 'a') as f:
            for k, nl in sorted(tracer.kernels.items()):
                f.write('bench:%s/%s/b%d\t%d\t%s\n' % (args.model, args.dtype, B, nl, k))
    per_rank_ms = None
    if world > 1:
        # every rank's own clock around the same K steps (the line's time is their MAX), gathered through the job's
        # backend: the spread says whether a rank lags (a slow GPU, a rank that shares its card in a rehearsal)
        mine = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allt, mine)
        per_rank_ms = [round(float(t.item()) / args.steps * 1e3, 3) for t in allt]
        elapsed = max(float(t.item()) for t in allt)
    loss_val = float(loss)

    # ---- per-family totals from the HIP events of the timed region ----
    fam = {}
    for name, flops, nbytes, e0, e1 in prof:
        rec = fam.setdefault(name, [0, 0.0, 0.0, 0.0])
        rec[0] += 1
        rec[1] += e0.elapsed_time(e1) * 1e-3
        rec[2] += flops
        rec[3] += nbytes
    dom = max(fam, key=lambda k: fam[k][1])
    n, secs, flops, nbytes = fam[dom]
    ridge = (PEAK['mfma_bf16'] if half else PEAK['mfma_f32']) * 1e12 / (PEAK['hbm'] * 1e9)
    if flops / max(nbytes, 1.0) >= ridge:
        peak = PEAK['mfma_bf16'] if half else PEAK['mfma_f32']
        roof = {'bound': 'mfma', 'achieved': round(flops / secs / 1e12, 2), 'peak': peak, 'unit': 'TFLOP/s'}
    else:
        roof = {'bound': 'hbm', 'achieved': round(nbytes / secs / 1e9, 1), 'peak': PEAK['hbm'], 'unit': 'GB/s'}
    roof['frac'] = round(roof['achieved'] / roof['peak'], 4)
    # HBM bytes per launch and MFMA-pipe utilisation of that kernel family from the PMC passes of the same command
    # (tools/profile_bench.sh: FETCH_SIZE / WRITE_SIZE / SQ_* in separate rocprofv3 runs, corrected as MI355X_MICROARCH.md
    # prescribes; summary committed under profiles/) -- only for the configuration they were collected on
    roof['traffic'] = None
    counter_bytes_step = None
    roof['algorithmic_bytes_per_launch'] = round(nbytes / n)
    pf = pmc_file(args.model, args.dtype, B)
    if pf and T == (600 if args.model.endswith('deep') else 300):
        summary = json.load(open(pf))
        pmc = summary['kernels'].get(PMC_KEY.get(dom, ''))
        # the counters are quoted only next to the library they were collected on: the summary carries the hash of the
        # kernel sources (tools/profile_summarise.py), the loaded library knows the tree it was built from
        from istgcn_amd import _lib
        if summary.get('csrc_hash') != _lib.build_id():
            roof['pmc_stale'] = 'counters in %s were collected on build %s, this library is %s' % (
                os.path.relpath(pf, ROOT), summary.get('csrc_hash'), _lib.build_id())
            pmc = None
        if pmc:
            roof['traffic'] = pmc.get('hbm_bytes_avg')
            roof['mfma_util'] = pmc.get('mfma_util')
            roof['pmc_source'] = os.path.relpath(pf, ROOT)
            counter_bytes_step = summary.get('hbm_bytes_per_step')
    roof['kernel'] = FAMILY.get(dom, dom)
    roof['launches'] = n
    roof['avg_launch_ms'] = round(secs / n * 1e3, 4)
    if args.breakdown:
        roof['share_of_gpu_time'] = round(secs / sum(v[1] for v in fam.values()), 3)
    else:
        roof['share_of_step_time'] = round(secs / elapsed, 3)
    if use_graph:
        roof['measured_over'] = '%d eager steps right after the timed region (the timed steps are hipGraph replays)' % args.steps
    if args.breakdown and rank == 0:
        tot = sum(v[1] for v in fam.values())
        for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
            sys.stderr.write('%-24s n=%5d  %8.2f ms/step  %5.1f%%  %8.1f TFLOP/s  %8.1f GB/s\n' % (
                k, v[0], v[1] / args.steps * 1e3, 100 * v[1] / tot, v[2] / v[1] / 1e12, v[3] / v[1] / 1e9))
        sys.stderr.write('kernels %.2f ms/step of %.2f ms/step wall\n' % (tot / args.steps * 1e3, elapsed / args.steps * 1e3))

    # ---- the whole step against ITS roofline: max(reference FLOPs / dense MFMA peak of the storage type, compulsory block
    #      bytes / 8 TB/s) / measured step time (per GPU: weak scaling) ----
    gf_clip, gb_clip_per_byte = STEP_WORK[args.model]
    tscale = T / (600.0 if args.model.endswith('deep') else 300.0)
    step_flops = gf_clip * 1e9 * B * tscale
    step_bytes = gb_clip_per_byte * (2 if half else 4) * 3 * 1e9 * B * tscale
    ms_f = step_flops / ((PEAK['mfma_bf16'] if half else PEAK['mfma_f32']) * 1e12) * 1e3
    ms_b = step_bytes / (PEAK['hbm'] * 1e9) * 1e3
    step_roof = {'flops_per_step': step_flops, 'bytes_per_step': step_bytes, 'ms_at_mfma_peak': round(ms_f, 3),
                 'ms_at_hbm_peak': round(ms_b, 3), 'bound': 'hbm' if ms_b >= ms_f else 'mfma',
                 'frac': round(max(ms_f, ms_b) / (elapsed / args.steps * 1e3), 4),
                 'counter_bytes_per_step': counter_bytes_step,
                 'source': 'SURVEY.md 8(d): reference FLOPs per clip (forward x 3), compulsory block bytes per clip x 3'}
    if rank == 0 and half and not args.no_vendor_gemm:
        roof['vendor_gemm_bf16_tflops'] = vendor_gemm_tflops(dev)
        roof['frac_of_vendor_gemm'] = round(roof['achieved'] / roof['vendor_gemm_bf16_tflops'], 4) if roof['bound'] == 'mfma' else None
    if rank == 0:
        line = {
            'metric': 'skeleton-clips/sec fwd+bwd, NTU V=25 T=300',
            'value': round(B * world * args.steps / elapsed, 2), 'unit': 'clips/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'net/%s.py NTU xsub shape (N,3,%d,%d,2), %d clips/GPU, train step (fwd+bwd+SGD-nesterov), '
                                   'dropout 0.5, random-init weights' % (args.model, T, V, B),
                       'launch': 'fwd+bwd replayed from one hipGraph; all-reduce + SGD eager' if use_graph else 'eager',
                       'input': 'host batches through harness.DeviceStager (pinned double buffer, H2D on a side stream) -- PCIe-inclusive' if args.h2d else 'resident in HBM',
                       'global_batch': B * world, 'parallelism': 'dp%d (batch-sharded, flat-bucket RCCL all-reduce)' % world,
                       'exchange': None if world == 1 else {
                           'backend': dist.get_backend(), 'world_size_reported': dist.get_world_size(),
                           'devices_visible_to_rank0': ndev, 'per_rank_ms_per_step': per_rank_ms,
                           'overlap': bool(getattr(opt, 'overlap', False)),       # ISTGCN_OVERLAP=0: one bucket, reduced in step()
                           'bucket_bytes': opt.bucket_bytes, 'early_bucket_bytes': 4 * getattr(opt, '_early_end', 0),
                           'early_all_reduces_launched_from_backward': getattr(opt, 'early_launches', 0)},
                       'storage': ('%s activations, fp32 accumulate/params, fp64 BN sums%s' % (
                           args.dtype, ', static loss scale %g' % loss_scale if loss_scale != 1.0 else '')) if half else 'fp32'},
            'roofline': roof, 'step_roofline': step_roof, 'final_loss': round(loss_val, 4),
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args.model, T)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
