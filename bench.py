#!/usr/bin/env python3
"""Headline benchmark: images/sec of 256x256 class-conditional HQ-Transformer sampling on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N > 1: launched by torch.distributed.run)

One "step" = one pass of the hot path over one batch per GPU: the 64-position hierarchical AR loop
(`sampling_ihqgpt`) followed by the HQ-VAE decode of the sampled code grids and clamp(0.5x+0.5) -- exactly
what one iteration of the reference's `measure_throughput` loop times (measure_throughput/__main__.py:88-113).
Workload = BASELINE.json configs[1]: ImageNet-256 class-conditional HQ-VAE + 12-layer HQ-Transformer,
batch 64 per GPU, synthetic (random-init weights, as the reference harness itself uses; random class per
step; top_k = top_p = None, temperatures [1, 1]).  Weak scaling: every rank samples its own 64 images,
weights replicated, no collective inside the path (every image is an independent chain); `--gather pixels|codes`
adds an optional RCCL gather of each step's result to rank 0.

The JSON line also carries `roofline` (dominant kernel family, measured live with HIP events on the
launch stream by libhqt's per-launch timers in a separate un-graphed pass) and `cpu_baseline` (the numpy
oracle = a port of the reference's CPU path, timed on this box's host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from hqtransformer_amd import synth  # noqa: E402
from hqtransformer_amd.config import load_config  # noqa: E402
from hqtransformer_amd.models import ImageGPT2  # noqa: E402
from hqtransformer_amd.sampling import sampling_hqtransformer, sampling_ihqgpt  # noqa: E402
from hqtransformer_amd.spec import decoder_plan, work_per_image  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA
F32_PEAK_TFLOPS = 157.3


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=96, help='timed steps; the default keeps the timed region above 5 s (a multiple of --merge: whole passes)')
    p.add_argument('--warmup', type=int, default=6)
    p.add_argument('--sampler', choices=['harness', 'quality'], default='harness',
                   help='harness: top_k = top_p = None, T = [1, 1] (measure_throughput/__main__.py:93-101, the headline); quality: top_k = 2048, '
                        'top_p = 1.0, T = 0.95 on every level (checkpoints/README.md:6, the defaults of sampling_hqmodel.py:27-31) -- exercises the '
                        'radix top-k and the sorted fp64-prefix top-p of the sampler kernel')
    p.add_argument('--config', default=os.path.join(ROOT, 'configs', 'imagenet-12l.yaml'))
    p.add_argument('--batch', type=int, default=64, help='images per GPU per step')
    p.add_argument('--precision', choices=['fast', 'exact'], default='fast',
                   help='AR loop arithmetic: fast = bf16 MFMA (the counterpart of the reference harness\'s fp16 autocast), exact = fp32')
    p.add_argument('--decode-precision', choices=['split', 'fast', 'exact'], default=None,
                   help='HQ-VAE decode arithmetic.  The reference harness decodes outside autocast, i.e. in fp32 (measure_throughput/__main__.py:108-113): '
                        'the default is split = fp32-accurate on the matrix cores (fp16 hi/lo operands, fp32 accumulate; pixels within 1e-4 of the '
                        'fp32 oracle); fast = bf16 (NOT like for like: 0.04 max pixel error), exact = fp32 FMA chains on the vector ALUs')
    p.add_argument('--gather', choices=['pixels', 'codes', 'none'], default=None,
                   help='RCCL gather of every step\'s result to rank 0 (BASELINE.json configs[2]: "RCCL gather over xGMI").  Default: pixels when '
                        'N > 1 (the finished images of every rank land on rank 0 inside the timed region), n/a at N = 1; none = no collective at all '
                        '(the path itself has no exchange step)')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-roofline', action='store_true')
    p.add_argument('--no-graph', action='store_true')
    p.add_argument('--no-exact-mode', action='store_true', help='skip the `exact_mode` record (fp32 AR loop: a few seconds)')
    p.add_argument('--inflight', type=int, default=None, help='passes in flight per GPU: consecutive passes are round-robined over this many '
                   'lanes (own HIP stream, KV cache and activations; shared weights).  1 = one pass at a time.  Default: see --merge')
    p.add_argument('--merge', type=int, default=None, help='execute this many queued steps as ONE device pass of merge x batch rows (every step keeps its own class id, '
                   'Philox seed and global row indices: per step the same draws as unmerged; the weights are streamed once for all of them).  '
                   'Default (neither --merge nor --inflight given): the K timed steps are split over 2 lanes in passes of up to 32 steps '
                   '(merge = min(32, ceil(K / 2))): K = 20 -> 2 passes of 10, K = 96 -> 3 passes of 32; text-conditional and three-level configs: 8 x 3 lanes.  '
                   'With only one of the two given the other defaults to --merge 8 / --inflight 3')
    p.add_argument('--ar-priority', type=int, default=None, choices=[0, 1],
                   help='1: the AR loop of every lane runs on a stream of the highest priority (its decode stays on the lane stream behind an event): an AR kernel '
                        'gets a freed compute unit before the other lane\'s convolution workgroups do.  Default: 1 when several lanes are in flight (see DEFAULT_AR_PRIORITY)')
    p.add_argument('--overlap', action='store_true', help='EXPERIMENT: run the decode of batch k on a second stream underneath the AR '
                   'loop of batch k+1 (measured slower on MI355X: the decoder starves the latency-bound AR kernels)')
    p.add_argument('--skip-decode', action='store_true', help='DEBUG ONLY (counter collection of the AR kernels at large row counts): no decode; the line is marked invalid')
    p.add_argument('--positions', type=int, default=0, help='DEBUG ONLY (counter collection): sample this many top positions '
                   'instead of the full grid; the resulting line is marked invalid')
    return p.parse_args()


def cpu_baseline(cfg, s2, s1, batch):
    """The reference's CPU path on THIS box's host cores: the hqt_cpu_* twins (include/hqt_cpu.h; oracle/cpu/hqt_cpu.cpp: C++ / OpenMP,
    fp32, own blocked GEMM with NUMA-local weight slices; pinned to the reference-generated fixtures by tests/test_cpu_twin.py), timed
    with the reference harness's accounting (measure_throughput/__main__.py:76,93-113): 8 of the 64 AR positions at the bench batch
    (incl. the prompt prefill for text conditioning) + the decode of the whole batch, extrapolated to 64 positions.  Thread count: all
    hardware threads or one per core pair, whichever runs one AR position faster (both reported).  `kind: "native-port"`: a baseline,
    never a fallback -- the product refuses CPU devices.  Three code levels (no C++ twin): the numpy oracle, `kind: "port"`."""
    if s2.levels == 3:
        return cpu_baseline_numpy(cfg, s2, s1, batch)
    from oracle import hqt_cpu
    hqt_cpu.build()
    avail = os.cpu_count() or 1
    w2 = synth.stage2_weights(s2, 0, 'bench')
    w1 = synth.stage1_weights(s1, 1, 'bench')
    cond = synth.text_ids(0, batch, s2.ctx_len_txt, s2.vocab_txt) if s2.cond == 2 else synth.class_ids(0, batch, max(s2.n_classes, 1))
    # How many threads?  os.cpu_count() is the machine; the box may hand this process far fewer cores (affinity mask, cgroup CPU quota), and
    # an OpenMP team larger than that spins against itself (measured on a gpurun box: 0.40 s per position on 128 threads, 27.9 s on 256).
    # Ascending sweep, one AR position each, stopped as soon as a larger team is clearly slower.
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else avail
    quota = None
    try:
        with open('/sys/fs/cgroup/cpu.max') as fp:
            q, per = fp.read().split()[:2]
            quota = None if q == 'max' else float(q) / float(per)
    except (OSError, ValueError):
        pass
    cap = min(avail, affinity)
    cands = sorted({n for n in (4, 8, 16, 32, 64, 128, 256, cap, cap // 2, int(quota) if quota else cap) if 1 <= n <= cap})
    sweep, best = {}, None
    for n in cands:
        twin = hqt_cpu.CpuTwin(s2, None, w2, threads=n)
        twin.sample(cond, batch, 1, None, seed=1)                # first touch of the activations outside the timing
        twin.sample(cond, batch, 1, None, seed=2)
        sweep[n] = round(twin.last_seconds, 4)
        if best is None or sweep[n] < sweep[best[0]]:
            if best is not None:
                best[1].close()
            best = (n, twin)
        else:
            twin.close()
            if sweep[n] > 1.5 * sweep[best[0]]:
                break
    cores, twin = best
    n_pos = 8
    twin.sample(cond, batch, n_pos, None, seed=3)
    t_n = twin.last_seconds
    t_prefill = 0.0
    if s2.cond == 2:                                             # the prompt prefill runs once per batch: separate it from the per-position cost
        twin.sample(cond, batch, 1, None, seed=3)
        t1 = twin.last_seconds
        per_pos = max(t_n - t1, 1e-9) / (n_pos - 1)
        t_prefill, t_ar = max(t1 - per_pos, 0.0), per_pos
    else:
        t_ar = t_n / n_pos
    twin.close()
    r = s1.z_res
    rng = np.random.default_rng(0)
    tw1 = hqt_cpu.CpuTwin(None, s1, None, w1, threads=cores)
    n_dec = batch
    tw1.decode_code(rng.integers(0, s1.n_embed, (n_dec, (r // 2) ** 2)), rng.integers(0, s1.n_embed, (n_dec, (r // 2) ** 2, 4)), clamp01=True, seq_layout=True)
    t_dec = tw1.last_seconds
    isa = tw1.isa
    tw1.close()
    n_positions = (r // 2) ** 2
    per_batch = t_prefill + t_ar * n_positions + t_dec
    return {'value': round(batch / per_batch, 4), 'unit': 'images/s', 'cores': cores, 'kind': 'native-port',
            'sample': f'{n_pos} of {n_positions} AR positions at batch {batch} ({t_ar * 1e3:.1f} ms/position' + (f', prompt prefill {t_prefill:.2f} s' if s2.cond == 2 else '') +
                      f'; extrapolated to {n_positions}) + the decode of all {n_dec} images ({t_dec:.2f} s): hqt_cpu_sample / hqt_cpu_decode_seq, C++ / OpenMP fp32 ({isa} GEMM micro-kernel) '
                      f'on {cores} of {avail} hardware threads',
            'phase_s_per_batch': {'ar': round(t_prefill + t_ar * n_positions, 3), 'decode': round(t_dec, 3)},
            'threads_sweep_s_per_position': {str(k): v for k, v in sorted(sweep.items())},
            'host': {'hardware_threads': avail, 'affinity': affinity, 'cgroup_cpu_quota_cores': quota}}


def cpu_baseline_numpy(cfg, s2, s1, batch):
    """Reference CPU path as restated by the oracle (fp32 numpy/OpenBLAS), bounded sample: 2 top positions of the AR loop at the bench
    batch + the decode of 1 image, scaled to images/s with the reference harness's accounting (64 positions per image batch, decode per
    image).  OpenBLAS with one thread per hardware thread of a 256-thread host is SLOWER on these GEMMs (64 rows: the threads mostly wait
    for each other) than with a fraction of them, so the BLAS pool is sized first: one AR position per candidate thread count, the
    fastest is used for the measurement and reported as `cores` (the threads actually used)."""
    from oracle.hqt_oracle import OracleStage1, OracleStage2, OracleStage2L3
    avail = os.cpu_count() or 1
    three = s2.levels == 3
    w2 = synth.stage2_weights(s2, 0, 'bench')
    w1 = synth.stage1_weights(s1, 1, 'bench')
    orc2, orc1 = (OracleStage2L3 if three else OracleStage2)(s2, w2), OracleStage1(s1, w1)
    n_pos = 2
    cond = synth.text_ids(0, batch, s2.ctx_len_txt, s2.vocab_txt) if s2.cond == 2 else synth.class_ids(0, batch, max(s2.n_classes, 1))
    r = s1.z_res
    rng = np.random.default_rng(0)
    if three:
        noise = np.maximum(rng.standard_exponential((n_pos, 21, batch, s2.vocab_top), dtype=np.float32), np.float32(1e-30))
    else:
        noise = synth.exp_noise(0, n_pos, batch, s2.vocab_top)
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:                     # no pool control: whatever OpenBLAS picks
        threadpool_limits = None
    sweep = {}
    cores = avail
    if threadpool_limits is not None and avail > 16:
        for n in sorted({avail, min(avail, 128), min(avail, 64), min(avail, 32), min(avail, 16), min(avail, 8)}, reverse=True):
            with threadpool_limits(limits=n):
                if not sweep:                                # first touch of the weights outside the timing
                    orc2.sample(cond, batch, 1, noise[:1])
                t0 = time.perf_counter()
                orc2.sample(cond, batch, 1, noise[:1])
                sweep[n] = round(time.perf_counter() - t0, 3)
        cores = min(sweep, key=sweep.get)

    def measure():
        t0 = time.perf_counter()
        orc2.sample(cond, batch, n_pos, noise)
        t_ar = (time.perf_counter() - t0) / n_pos
        t_prefill = 0.0
        if s2.cond == 2:                        # the prompt prefill runs once per batch: separate it from the per-position cost
            t0 = time.perf_counter()
            orc2.sample(cond, batch, 1, noise[:1])
            t1 = time.perf_counter() - t0
            per_pos = max(t_ar * n_pos - t1, 1e-9) / (n_pos - 1)
            t_prefill, t_ar = max(t1 - per_pos, 0.0), per_pos
        code_b = rng.integers(0, s1.n_embed, (1, r, r))
        code_m = rng.integers(0, s1.n_embed, (1, r // 2, r // 2))
        t0 = time.perf_counter()
        if three:
            orc1.decode_codes3([rng.integers(0, s1.n_embed, (1, r // 4, r // 4)), code_m, code_b])
        else:
            orc1.decode_code(code_m, code_b)
        return t_ar, t_prefill, time.perf_counter() - t0
    if threadpool_limits is not None:
        with threadpool_limits(limits=cores):
            t_ar, t_prefill, t_dec = measure()
    else:
        t_ar, t_prefill, t_dec = measure()
    n_positions = (r // (4 if three else 2)) ** 2
    per_image = (t_prefill + t_ar * n_positions) / batch + t_dec
    return {'value': round(1.0 / per_image, 4), 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n_pos} of {n_positions} AR positions at batch {batch} ({t_ar:.2f} s/position' + (f', prompt prefill {t_prefill:.2f} s' if s2.cond == 2 else '') + ') + decode of 1 image '
                      f'({t_dec:.2f} s), fp32 numpy/OpenBLAS oracle on {cores} BLAS threads of {avail} hardware threads, extrapolated per image',
            'blas_threads_sweep_s_per_position': {str(k): v for k, v in sorted(sweep.items())}}


def pmc_traffic(family, rows=None):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 --pmc passes (profiles/pmc_latest.json: FETCH_SIZE x 2 as
    MI355X_MICROARCH.md prescribes for gfx950 + WRITE_SIZE, in bytes).  `rows`: the row count of the pass the record times -- the AR GEMM
    family's traffic depends on it, so it is reported only when the counters were collected AT that row count (by_rows[rows]); otherwise
    None.  rows=None: a family whose launches do not depend on the pass (the decoder runs 64-image chunks whatever the pass)."""
    path = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
    try:
        with open(path) as fp:
            doc = json.load(fp)
    except (OSError, ValueError):
        return None
    if rows is None:
        return doc.get(family)
    return doc.get('by_rows', {}).get(str(int(rows)), {}).get(family)


DEFAULT_AR_PRIORITY = False      # measured: profiles/r05_ar_priority.txt


def default_schedule(steps, merge=None, inflight=None, wide=False):
    """(lanes, steps per pass) of the timed region.  Neither given (class-conditional two-level workload): the K timed steps are split over
    2 lanes in passes of up to 32 steps -- larger passes stream the AR weights less often per image (85.5 ms of AR per 64
    images in 64-row passes, 13.3 in 2048-row passes), 32 bounds the latency of a step and the KV cache of a lane, and two passes must exist
    for the lanes to overlap anything.  The text and three-level workloads (`wide`: a 64-token prefill / 16-fold depth rows per step) take
    3 lanes and passes of up to 16 steps (measured: 8 x 3 1308 / 811, 16 x 3 1346 / 837, 24 x 2 1328 / 727 images/s).  A flag that is given
    is kept; the other one then takes round 2's default (8 steps per pass, 3 lanes)."""
    if merge is None and inflight is None:
        return (3, min(16, max(1, (steps + 2) // 3))) if wide else (2, min(32, max(1, (steps + 1) // 2)))
    return max(1, inflight if inflight is not None else 3), max(1, merge if merge is not None else 8)


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks as a CHILD `python -m torch.distributed.run`
    of this very command line and exit with its code.  Runs before anything touches the GPU (device_count() does not initialise it on
    this image), so the parent never holds a device; an N-GPU flag can therefore never silently produce a 1-GPU line."""
    import socket
    import subprocess
    visible = torch.cuda.device_count()
    if visible < n and not os.environ.get('HQT_BENCH_SHARE_GPU'):
        raise SystemExit(f'bench.py --gpus {n}: {visible} GPU(s) visible -- refusing to print an {n}-GPU line from fewer devices')
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    print(f'[bench] --gpus {n} without a launcher: starting {n} ranks: {" ".join(cmd)}', file=sys.stderr)
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        launch_ranks(args.gpus)                    # does not return
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:                         # a launcher that started a different number of ranks than the flag says: no line at all
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: the line would not describe the run')
    if args.gather is None:
        args.gather = 'pixels' if world > 1 else 'none'
    gather_requested = args.gather
    dist = None
    # HQT_BENCH_FORCE_DIST=1 (test hook): a ONE-rank RCCL process group, so that the nccl branch below -- init, pre-flight, per-step gather,
    # barrier, max-over-ranks -- executes on a one-GPU box too (tests/test_gpu_dist.py); the numbers are those of the plain 1-GPU run
    force_dist = world == 1 and bool(os.environ.get('HQT_BENCH_FORCE_DIST'))
    if force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if gather_requested == 'none' and not any(a.startswith('--gather') for a in sys.argv):
            args.gather = gather_requested = 'pixels'
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # test hook for boxes with fewer GPUs than ranks (exercises the N > 1 control flow only; the numbers mean nothing):
        # HQT_BENCH_SHARE_GPU=1 maps the ranks onto the visible devices round-robin and rendezvous runs over gloo
        share = bool(os.environ.get('HQT_BENCH_SHARE_GPU'))
        if share:
            # several ranks on one device: the persistent AR chain needs every CU of the GPU for its one-workgroup-per-CU grid, and two such
            # launches from two processes keep each other out (each gives up after 1 s and the call reports it).  One process per GPU -- the
            # real multi-GPU run -- is unaffected; the shared-device rehearsal takes the launch chain.
            os.environ['HQT_PERSIST'] = '0'
        if share:
            local_rank %= max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if share:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    # small control tensors (elapsed times) travel on the backend's native device: the GPU for nccl / RCCL, the host for the gloo test hook
    cdev = dev if (dist is None or dist.get_backend() == 'nccl') else torch.device('cpu')

    cfg = load_config(args.config)
    model = ImageGPT2(cfg, seed=0).to(dev).eval()
    s2, s1 = model.stage2.spec, model.stage1.spec
    B = args.batch
    n_pos = (s1.z_res // (4 if s2.levels == 3 else 2)) ** 2
    n_full = n_pos
    if args.positions:
        n_pos = min(n_pos, args.positions)
    fast = args.precision == 'fast'
    dec_prec = args.decode_precision or ('split' if fast else 'exact')
    quality = args.sampler == 'quality'
    tk, tp, T = (2048, 1.0, 0.95) if quality else (None, None, 1.0)
    classes = synth.class_ids(1000 + rank, args.steps + args.warmup + 4, max(s2.n_classes, 1))
    txt_cond = s2.cond == 2                 # text-conditional configs (BASELINE configs[4]): synthetic prompt ids, resident in HBM
    prompts = [torch.from_numpy(synth.text_ids(2000 + 131 * rank + i, B, s2.ctx_len_txt, s2.vocab_txt)).to(dev)
               for i in range(args.steps + args.warmup + 4)] if txt_cond else None

    def cond_of(i):
        return prompts[i % len(prompts)] if txt_cond else int(classes[i % len(classes)])
    H = s1.resolution
    gathered = None
    if dist is not None and args.gather == 'pixels' and rank == 0:
        gathered = [torch.empty((B, s1.out_ch, H, H), dtype=torch.float32, device=dev) for _ in range(world)]

    if dist is not None and args.gather != 'none':
        # pre-flight: one collective of the real size before anything is timed.  A broken fabric / RCCL set-up must not take the whole
        # measurement down: the line then says so and the run goes on without the gather (the path itself has no exchange step)
        try:
            if args.gather == 'pixels':
                dist.gather(torch.zeros((B, s1.out_ch, H, H), dtype=torch.float32, device=dev), gathered, dst=0)
            else:
                dist.all_gather_into_tensor(torch.empty((world * B, 1), dtype=torch.int64, device=dev), torch.zeros((B, 1), dtype=torch.int64, device=dev))
            torch.cuda.synchronize(dev)
        except Exception as e:                      # noqa: BLE001
            print(f'[bench] rank {rank}: --gather {args.gather} failed in the pre-flight ({type(e).__name__}: {e}); continuing without it', file=sys.stderr)
            args.gather = f'none (requested gather failed: {type(e).__name__})'

    # The gather never rides on a lane's compute stream: the collective is queued on a stream of its own behind an event of the finished
    # decode, so the lane goes on with its next pass while the pixels travel -- a slow rank-0 receive (7 x 50 MB per step at 8 GPUs) cannot
    # stall anybody's decode.  Every rank issues its collectives in submission order, i.e. in the same order.
    s_gather = torch.cuda.Stream(device=dev) if (dist is not None and not str(args.gather).startswith('none')) else None

    def gather_step(ct, px):
        if s_gather is None:
            return
        cur = torch.cuda.current_stream(dev)
        done = torch.cuda.Event()
        done.record(cur)
        s_gather.wait_event(done)
        with torch.cuda.stream(s_gather):
            if args.gather == 'pixels':
                dist.gather(px, gathered, dst=0)
                px.record_stream(s_gather)
            elif args.gather == 'codes':
                dist.all_gather_into_tensor(torch.empty((world * B, n_pos), dtype=torch.int64, device=dev), ct)
                ct.record_stream(s_gather)

    three = s2.levels == 3
    samp_kw = (dict(top_k=[tk] * 3, top_p=[tp] * 3, softmax_temperature=[T] * 3) if three else
               dict(top_k_top=tk, top_p_top=tp, top_k_bot=tk, top_p_bot=tp, softmax_temperature=[T, T]))

    def sample_codes(i, graph, nb=None, ar_prec=None):
        """(codes0, rest): rest = codes_bot (two levels) or [codes1, codes2] (three levels).  nb: rows of the pass (default: one step's batch);
        ar_prec: 'fast' | 'exact' | 'split' (default: --precision)."""
        nb = nb or B
        ar_prec = ar_prec or args.precision
        fast = ar_prec == 'fast'
        cond = cond_of(i)
        if txt_cond and nb > B:          # a text batch is as long as its prompt tensor (sampling.py:187-190): a pass of nb rows = nb / B prompt batches
            cond = torch.cat([cond_of(i + j) for j in range(nb // B)], 0)
        if three:
            c = sampling_hqtransformer(model.stage2, num_candidates=nb, cond=cond, top_k=[tk] * 3, top_p=[tp] * 3,
                                       softmax_temperature=[T] * 3, use_fp16=fast, is_tqdm=False, max_seq_len=n_pos, seed=1 + i,
                                       sample_offset=rank * B, use_graph=graph, precision=ar_prec)
            return c[0], c[1:]
        return sampling_ihqgpt(model.stage2, num_candidates=nb, cond=cond, top_k_top=tk, top_p_top=tp, top_k_bot=tk,
                               top_p_bot=tp, softmax_temperature=[T, T], use_fp16=fast, is_tqdm=False, max_seq_len=n_pos,
                               model_stage1=None, seed=1 + i, sample_offset=rank * B, use_graph=graph, precision=ar_prec)

    def decode(ct, cb, m=None, prec=None):
        prec = prec or dec_prec
        if args.skip_decode:
            return torch.zeros((ct.shape[0], s1.out_ch, 8, 8), dtype=torch.float32, device=dev)
        if three:
            return (m or model).stage1.decode_sequences([ct] + list(cb), precision=prec, clamp01=True)
        if n_pos < n_full:      # debug runs: pad the code grids so the decoder still sees full-size inputs
            ct = torch.cat([ct, ct.new_zeros(ct.shape[0], n_full - n_pos)], 1)
            cb = torch.cat([cb, cb.new_zeros(cb.shape[0], n_full - n_pos, 4)], 1)
        return (m or model).stage1.decode_sequences(ct, cb, precision=prec, clamp01=True)

    def step(i, graph=True, nb=None):
        ct, cb = sample_codes(i, graph and not args.no_graph, nb)
        px = decode(ct, cb)
        gather_step(ct, px)
        return ct, cb, px

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i)
    barrier()
    # Optional pipeline experiment (--overlap): decode (+gather) of batch k on a second, lower-priority stream underneath
    # the AR loop of batch k+1.  Measured on MI355X: AR 101 -> 129 ms under contention, net 494 -> 473 images/s, so the
    # default is the serial order of the reference harness.
    overlap = args.overlap
    lo_prio, hi_prio = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else (0, -1)
    s_ar = torch.cuda.Stream(device=dev, priority=hi_prio) if overlap else torch.cuda.current_stream(dev)
    s_dec = torch.cuda.Stream(device=dev, priority=lo_prio) if overlap else s_ar
    # ---- the timed region: K steps, round-robined over `inflight` lanes (hqtransformer_amd/pipeline.py).  Every step is
    #      one complete batch-B pass (64-position AR loop + decode + clamp [+ gather]); lanes only change the schedule.
    from hqtransformer_amd.pipeline import InflightSampler
    inflight, merge = default_schedule(args.steps, args.merge, args.inflight, txt_cond or three)
    if args.positions:
        merge = 1                                  # debug runs (counter collection) sample a few positions of one pass
    rem = args.steps % merge                       # K need not be a multiple: the last pass of the timed region then holds `rem` steps
    ar_priority = bool(args.ar_priority) if args.ar_priority is not None else (DEFAULT_AR_PRIORITY and inflight > 1)
    pipe = InflightSampler(model, lanes=inflight, device=dev, merge=merge, ar_high_priority=ar_priority)

    def after(ct, cb, px):
        gather_step(ct, px)

    debug_short = n_pos < n_full            # --positions (counter collection): the padded decode of step(), one lane, no pipeline
    for li in range(0 if debug_short else max(inflight, args.warmup) * merge):  # every lane at least once: workspace, graph capture
        pipe.submit(B, cond_of(li), seed=1000 + li, max_seq_len=n_pos, use_fp16=fast, sample_offset=rank * B,
                    use_graph=not args.no_graph, after=after, precision=dec_prec, **samp_kw)
    pipe.flush()
    for lane in range(0 if debug_short or not rem else inflight):  # ... and the shorter last pass once per lane too (its own workspace + graph)
        for j in range(rem):
            pipe.submit(B, cond_of(j), seed=2000 + lane * merge + j, max_seq_len=n_pos, use_fp16=fast, sample_offset=rank * B,
                        use_graph=not args.no_graph, after=after, precision=dec_prec, **samp_kw)
        pipe.flush()
    pipe.drain()
    barrier()
    t0 = time.perf_counter()
    kept = [step(args.warmup + k) for k in range(args.steps)] if debug_short else \
           [pipe.submit(B, cond_of(args.warmup + k), seed=1 + args.warmup + k, max_seq_len=n_pos, use_fp16=fast,
                        sample_offset=rank * B, use_graph=not args.no_graph, after=after, precision=dec_prec,
                        order_after_current=not os.environ.get('HQT_BENCH_NO_ORDER'), **samp_kw) for k in range(args.steps)]
    host_submit_s = time.perf_counter() - t0      # when the last step was handed to HIP (wall: includes waiting for room in the HIP queues)
    pipe.drain()
    barrier()
    elapsed_lanes = time.perf_counter() - t0
    elapsed_ranks = [elapsed_lanes]                # per-rank wall time of the timed region (skew visibility: one slow box vs the gather)
    if dist is not None:
        t = torch.tensor([elapsed_lanes], dtype=torch.float64, device=cdev)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        elapsed_ranks = [float(x.item()) for x in allt]
        elapsed_lanes = max(elapsed_ranks)         # the contract's MAX over ranks
    # what the host itself spends per step: one more pass submitted into EMPTY queues (untimed), so that nothing throttles the submitting thread
    # (at most 8 steps: a pass of 32 steps is > 4000 launches, more than the HIP queues take without blocking the submitter -- measured 26 ms
    # "per step" that way, all of it waiting; the first, untimed round captures the graph of that row count)
    host_free_s, host_free_steps = 0.0, min(merge, 8)
    if not debug_short:
        for timed_round in (False, True):
            t1 = time.perf_counter()
            for j in range(host_free_steps):
                pipe.submit(B, cond_of(j), seed=3000 + j, max_seq_len=n_pos, use_fp16=fast, sample_offset=rank * B,
                            use_graph=not args.no_graph, after=None, precision=dec_prec, **samp_kw)
            pipe.flush()
            host_free_s = (time.perf_counter() - t1) / host_free_steps
            pipe.drain()
    host_ms_ranks = [round(1000 * host_submit_s / args.steps, 3)]
    if dist is not None:
        t = torch.tensor([host_submit_s], dtype=torch.float64, device=cdev)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        host_ms_ranks = [round(1000 * float(x.item()) / args.steps, 3) for x in allt]
    del kept
    pipe.release(B, n_pos)            # lane 0 back to the latency-oriented kernels for the one-at-a-time reference pass
    sample_codes(0, not args.no_graph)  # untimed: the policy change re-captures lane 0's graph; keep that out of the pass below

    # ---- reference pass: the same steps one at a time on one lane (the reference harness's order), with per-phase events
    def one_at_a_time(n, ar_prec, prec, gather=True):
        """n batch-B steps strictly one after the other on one lane -- what measure_throughput/__main__.py:84-116 does -- timed between
        barriers, AR / decode split by events.  Returns the record (whole-job images/s, ms per step, phase ms)."""
        for w in range(2):                          # untimed: graph capture / workspace of this precision
            decode(*sample_codes(w, not args.no_graph, ar_prec=ar_prec), prec=prec)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * n)]
        keep = []
        barrier()
        t0 = time.perf_counter()
        for k in range(n):
            i = args.warmup + k
            with torch.cuda.stream(s_ar):
                if k > 0:
                    s_ar.wait_event(ev[3 * k - 1])     # the reference harness runs sample -> decode -> sample ...: nothing of step k + 1 starts before step k's pixels are done
                ev[3 * k].record()
                ct, cb = sample_codes(i, not args.no_graph, ar_prec=ar_prec)
                ev[3 * k + 1].record()
            with torch.cuda.stream(s_dec):
                s_dec.wait_event(ev[3 * k + 1])
                px = decode(ct, cb, prec=prec)
                ev[3 * k + 2].record()
                if gather:
                    gather_step(ct, px)
            keep.append((ct, cb, px))
        barrier()
        el = time.perf_counter() - t0
        model.stage1.range_check()             # SPLIT decode: an activation outside the fp16 range would invalidate the pixels (raises)
        model.stage2.range_check()             # SPLIT AR passes above 256 rows saturate out-of-range activations and only flag them; also a persistent AR launch that gave up
        a_ms = sum(ev[3 * k].elapsed_time(ev[3 * k + 1]) for k in range(n)) / n
        d_ms = sum(ev[3 * k + 1].elapsed_time(ev[3 * k + 2]) for k in range(n)) / n
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return {'value': round(world * B * n / el, 2), 'ms_per_step': round(1000 * el / n, 3), 'steps': n,
                'phase_ms': {'ar': round(a_ms, 3), 'decode': round(d_ms, 3)}}

    n_serial = args.steps if (inflight == 1 and merge == 1) else min(args.steps, 3)
    serial = one_at_a_time(n_serial, args.precision, dec_prec)
    serial['note'] = 'the same steps one at a time on one lane (the reference harness order: measure_throughput/__main__.py:84-116)'
    ar_ms, dec_ms = serial['phase_ms']['ar'], serial['phase_ms']['decode']
    # ---- the arithmetic whose code sequences are BIT-EXACT against the reference's CPU path (fp32 AR loop; the decode stays fp32-accurate):
    #      the same workload, one step at a time and as one merged pass of the timed region's row count
    exact_mode = None
    if fast and not args.no_exact_mode and not args.positions:
        xprec = dec_prec if dec_prec != 'fast' else 'split'
        what = {'exact': 'EXACT: fp32 weights / activations / accumulation -- every nn.Linear of up to 256 rows on the fp32 MATRIX instructions (exact_mfma_gemm_kernel: v_mfma_f32_16x16x4_f32, '
                         'bitwise an fmaf chain), larger ones on the vector ALUs (gemm_tile_kernel) -- codes bit-identical to the oracle (tests/test_gpu_timed_schedule.py)',
                'split': 'SPLIT API: the EXACT launch sequence with every nn.Linear on the matrix cores -- up to 256 rows per GEMM the fp32 matrix instructions (exact_mfma_gemm_kernel: the SAME '
                         'kernels as `exact`, so a one-step-at-a-time record of `split` times what `exact` times), above 256 rows fp16 hi / lo operands with 3 MFMAs per term '
                         '(split_gemm_kernel); fp32 accumulation, LayerNorm, attention, softmax and sampler -- codes bit-identical to the oracle wherever the draw is well-conditioned '
                         '(winner / runner-up margin >= 1.00001: include/hqt.h), logits <= 2e-4 (same tests)'}
        exact_mode = {}
        for arp in ('split', 'exact'):
            ex = one_at_a_time(3, arp, xprec, gather=False)
            ex['precision'] = {'ar': what[arp], 'decode': xprec}
            ex['ar_gemm_kernels'] = ('exact_mfma_gemm_kernel (fp32 matrix instructions): every AR GEMM of a batch-%d step has <= 256 rows' % B) if 4 * B <= 256 else (
                'split_gemm_kernel above 256 rows, exact_mfma_gemm_kernel up to 256' if arp == 'split' else 'gemm_tile_kernel above 256 rows, exact_mfma_gemm_kernel up to 256')
            ex['note'] = '3 batch-%d steps one at a time on one lane' % B
            if 4 * B <= 256:
                ex['same_kernels_as'] = 'split.like_for_like and exact.like_for_like launch the SAME kernels at this batch (every GEMM <= 256 rows): two timings of one thing'
            rec = {'like_for_like': ex}
            if merge > 1:
                m_ex = merge if arp == 'split' else min(merge, 16)
                xp = InflightSampler(model, lanes=1, device=dev, merge=m_ex)
                for rnd in range(2):                    # first round untimed (capture at this row count)
                    barrier()
                    t0 = time.perf_counter()
                    for j in range(m_ex):
                        xp.submit(B, cond_of(j), seed=4000 + j, max_seq_len=n_pos, use_fp16=False, sample_offset=rank * B,
                                  use_graph=not args.no_graph, after=None, precision=xprec, ar_precision=arp, **samp_kw)
                    xp.drain()
                    barrier()
                    el = time.perf_counter() - t0
                rec['merged'] = {'value': round(world * B * m_ex / el, 2), 'ms_per_step': round(1000 * el / m_ex, 3), 'steps': m_ex,
                                 'rows_per_pass': m_ex * B, 'lanes': 1,
                                 'note': f'one pass of {m_ex} merged batch-{B} steps ({m_ex * B} rows), {arp} AR loop + {xprec} decode, one lane'}
                del xp
            exact_mode[arp] = rec
        exact_mode['note'] = ('the arithmetic whose code sequences are bit-identical to the reference CPU path (north_star): `split` = fp32-accurate on the matrix cores, '
                              '`exact` = fp32 arithmetic throughout (fp32 matrix instructions up to 256 rows, vector ALUs above); same workload, same decode; the headline `value` runs the tolerance-gated bf16 AR loop')
    elapsed = elapsed_lanes

    out = None
    if rank == 0:
        work = work_per_image(s2, s1, n_pos)
        value = world * B * args.steps / elapsed
        out = {
            'metric': 'images/sec (256x256 text-cond sampling)' if txt_cond else 'images/sec (256x256 class-cond sampling)', 'value': round(value, 2), 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1000 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16' if fast else 'f32', 'data': 'synthetic',
            'config': {'workload': (f'text-to-image ({s2.ctx_len_txt}-token synthetic prompts, prefill + ' if txt_cond else 'imagenet256-classcond (') + f'hq-vae({"8x8+16x16+32x32, three code levels" if three else "8x8+16x16"})+hq-transformer {s2.n_layers}L/{s2.embed_dim}d), '
                                   f'{args.steps} steps of batch {B}/GPU, {n_pos} top positions, ' + (f'top_k={tk}, top_p={tp}, T={T} (quality-mode sampler)' if quality else 'top_k=top_p=None, T=[1,1]') +
                                   (f'; SCHEDULE: {args.steps} x {B}-row steps merged into {merge * B}-row passes ({merge} steps per pass, {inflight} passes in flight) -- the one-step-at-a-time order of the '
                                    f'reference harness is the top-level `like_for_like` record' if (merge > 1 or inflight > 1) else '; one step at a time (the reference harness order)'),
                       'global_batch': world * B, 'per_gpu_batch': B, 'steps_per_pass': merge, 'parallelism': f'dp{world} (sample-sharded, weights replicated)',
                       # the schedule, in numbers: a device pass executes `merge` steps at once, `inflight` passes are resident per GPU
                       'schedule': (('chosen from K: the timed steps are split over 3 lanes in passes of up to 16 steps (merge = min(16, ceil(K / 3)))' if (txt_cond or three) else
                                     'chosen from K: the timed steps are split over 2 lanes in passes of up to 32 steps (merge = min(32, ceil(K / 2)))')
                                    if (args.merge is None and args.inflight is None) else 'as given (--merge / --inflight; defaults 8 / 3)'),
                       'rows_per_pass': merge * B, 'images_in_flight_per_gpu': merge * B * inflight,
                       'step_latency_ms': round(1000 * elapsed / args.steps * merge * inflight, 1),
                       'step_latency_note': 'time from a step entering its pass to its pixels: one pass per lane, lanes share the GPU (ms_per_step x merge x lanes); '
                                            'serial.ms_per_step is the latency of the unmerged one-step-at-a-time order',
                       'precision': {'ar': 'FAST: bf16 weights + MFMA, fp32 accumulate / LayerNorm / softmax / sampler (the reference harness samples under fp16 autocast)' if fast else 'EXACT: fp32',
                                     'decode': {'split': 'SPLIT: fp32-accurate on the matrix cores (fp16 hi/lo operands, 3 MFMAs per term, fp32 accumulate); pixels within 1e-4 of the fp32 oracle '
                                                         '(the reference harness decodes in fp32, outside autocast)',
                                                'fast': 'FAST: bf16 MFMA (0.04 max pixel error: NOT the reference harness\'s fp32 decode)',
                                                'exact': 'EXACT: fp32 FMA chains on the vector ALUs'}[dec_prec]},
                       'gather': (args.gather + (' (RCCL, every step, inside the timed region, on a stream of its own behind an event of the finished decode)' if args.gather in ('pixels', 'codes') else '')) if dist is not None else 'n/a', 'hip_graph': not args.no_graph,
                       'pipeline': (f'{inflight} passes in flight per GPU (round-robin over {inflight} lanes: own HIP stream, KV cache and '
                                    f'activations, shared weights, throughput-oriented GEMM tiles)') if inflight > 1 else 'one pass at a time',
                       'merge': (f'{merge} consecutive batch-{B} steps execute as ONE device pass of {merge * B} rows (own class id, Philox seed and global row indices per step).  In EXACT (fp32) arithmetic '
                                 f'the draws of a step do not depend on what it is merged with (bit-identical to the unmerged call); in FAST (bf16, this line) the GEMM / attention / sampler kernels are chosen by the row '
                                 f'count of the pass, so draws are reproducible per (seed, schedule) and agree with the unmerged call only within the FAST gates (tests/test_gpu_timed_schedule.py).  '
                                 f'The reference harness runs one batch-{B} step at a time -- that order is the `like_for_like` (= `serial`) record') if merge > 1 else 'none'},
            # bit-exactness against the reference's CPU path holds for the fp32 AR loop (--precision exact, the `exact_mode` record) only; the timed arithmetic is tolerance-gated
            'bit_exact_codes': (not fast),
            'gather_ok': (None if dist is None else (args.gather == gather_requested)),
            # per rank: wall time of the timed region between the two barriers and the images/s of that rank alone; `value` is world * B * K / max(elapsed_s_ranks)
            'elapsed_s_ranks': [round(v, 4) for v in elapsed_ranks], 'value_ranks': [round(B * args.steps / v, 2) for v in elapsed_ranks],
            'host_ms_per_step': max(host_ms_ranks), 'host_ms_per_step_ranks': host_ms_ranks,
            'host_ms_per_step_unthrottled': round(1000 * host_free_s, 3),
            'host_ms_note': 'host_ms_per_step = wall time until the last step of the timed region was handed to HIP / steps -- the runtime blocks (spins) the submitting thread '
                            'when its queues are full, so a value close to ms_per_step means the DEVICE paces the run; host_ms_per_step_unthrottled = the same submission '
                            f'into empty queues (one untimed pass of {host_free_steps} steps, rank 0): what the host itself costs per step',
            'env_switches': {k: v for k, v in sorted(os.environ.items()) if k.startswith('HQT_')},
            'serial': serial,
            # the SAME record under the name VERDICT r03 asked for: BASELINE configs[1] ("batch 64") taken literally, one step at a time
            'like_for_like': dict(serial, images_in_flight_per_gpu=B),
        }
        if exact_mode is not None:
            out['exact_mode'] = exact_mode
        if n_pos < n_full or args.skip_decode:
            out['INVALID'] = f'debug run: {n_pos} of {n_full} top positions sampled' + (', decode skipped' if args.skip_decode else '')

    # ---- roofline: per-launch HIP-event timers inside libhqt, un-graphed pass, rank 0 only
    if rank == 0 and not args.no_roofline:
        # One pass of the timed region = `merge` steps = Bm rows: time the kernels of THAT pass (one lane, un-graphed, per-launch events).
        Bm = merge * B
        e2, e1 = model.stage2.engine(Bm, n_pos), model.stage1.engine(Bm)
        ar_ms_serial = ar_ms
        if inflight > 1 or merge > 1:
            # the timed region ran the throughput-oriented GEMM tiles (hqt_set_policy) when several lanes were in flight: time THOSE
            # kernels, first one graphed pass for the graphed/un-graphed scale, then the un-graphed per-launch pass
            e2.set_policy(1 if inflight > 1 else 0)
            step(0, nb=Bm)                                              # capture under this policy
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            ea.record()
            sample_codes(args.warmup, not args.no_graph, Bm)
            eb.record()
            torch.cuda.synchronize(dev)
            ar_ms = ea.elapsed_time(eb)
        for e in (e2, e1):
            e.timing(True)
            e.timing_reset()
        step(0, graph=False, nb=Bm)
        torch.cuda.synchronize(dev)
        if inflight > 1:
            e2.set_policy(0)
        rep = {}
        for e in (e2, e1):
            for k, v in e.timing_report().items():
                rep[k] = (rep.get(k, (0, 0.0))[0] + v[0], rep.get(k, (0, 0.0))[1] + v[1])
            e.timing(False)
        gemm = {k: v for k, v in rep.items() if k.startswith('gemm_') or k.startswith('persist_')}     # persist_*: the persistent AR chain (one launch = the GEMMs, attention and epilogues of a top position up to its top logits)
        conv = {k: v for k, v in rep.items() if k in ('conv3x3', 'conv1x1', 'conv_out', 'attn_gemm')}
        ar_classes = [k for k in rep if k.startswith('gemm_') or k.startswith('persist_') or k in ('layernorm', 'attention', 'sampler', 'embed')]
        ar_eager_ms = sum(rep[k][1] for k in ar_classes)
        # The timed region replays the AR loop from a hipGraph; the per-launch events can only be recorded in an un-graphed
        # pass, where every launch carries an extra event record and the eager launch gap.  Both passes are timed with HIP
        # events, so the per-launch durations are rescaled by (graphed AR loop time / un-graphed AR loop time); the
        # result matches the rocprofv3 kernel-trace averages of the graphed run (profiles/).
        ar_scale = (ar_ms / ar_eager_ms) if (ar_eager_ms > 0 and not args.no_graph) else 1.0
        gemm_ms = sum(v[1] for v in gemm.values()) * ar_scale
        conv_ms = sum(v[1] for v in conv.values())
        wbytes = (2 if fast else 4) / 2 * work['ar_weight_bytes_per_pos'] * n_pos           # per pass (Bm rows), all AR GEMM launches
        cflops = work['dec_flops'] * Bm
        passes = args.steps / merge                                                          # passes of the timed region
        fam = []
        if gemm_ms > 0:
            n_l = sum(v[0] for v in gemm.values())
            hbm = wbytes / (gemm_ms * 1e-3) / 1e9
            gflops = work['ar_flops'] * Bm * (n_pos / n_full)                                # algorithmic FLOPs of the AR GEMMs of one pass (Bm images)
            tf = gflops / (gemm_ms * 1e-3) / 1e12
            # Which roofline bounds an AR GEMM depends on the rows of the pass: its intensity is ~rows FLOP per weight byte, the ridge of the
            # part 2500 TFLOP/s / 8 TB/s = 312 FLOP/B.  Below 256 rows the launches are weight streams (HBM); merged passes are matrix work.
            mfma_bound = Bm >= 256
            persist = any(k.startswith('persist_') for k in gemm)
            rec = {'kernel': (('persist_kernel (body blocks, ln_f + sos_depth, depth sub-step 0 and head_top of a top position as ONE launch: GEMMs + attention + epilogues; '
                               f'csrc/persist.hip) + stream_gemm_kernel (depth sub-step 1 at {4 * Bm} rows, head_bot): AR weight-streaming family, {Bm}-row passes') if persist else
                              f'tile_gemm_kernel / stream_gemm_kernel: AR GEMM family (qkv / proj / fc1 / fc2 / heads + split-K combine; {Bm}-row passes, {4 * Bm} rows in depth sub-step 1)'),
                   'bound': 'mfma' if mfma_bound else 'hbm',
                   'achieved': round(tf, 1) if mfma_bound else round(hbm, 1), 'peak': MFMA_BF16_PEAK_TFLOPS if mfma_bound else HBM_PEAK_GBS,
                   'unit': 'TFLOP/s' if mfma_bound else 'GB/s',
                   'frac': round(tf / MFMA_BF16_PEAK_TFLOPS, 4) if mfma_bound else round(hbm / HBM_PEAK_GBS, 4),
                   'traffic': pmc_traffic('stream_gemm', Bm), 'launches': n_l, 'avg_launch_us': round(1000 * gemm_ms / n_l, 3),
                   'total_ms': round(gemm_ms, 3), 'algorithmic_flops_per_launch': round(gflops / n_l), 'algorithmic_bytes_per_launch': round(wbytes / n_l),
                   'weight_stream_GBps': round(hbm, 1), 'weight_stream_frac_of_hbm_peak': round(hbm / HBM_PEAK_GBS, 4),
                   'eager_to_graph_scale': round(ar_scale, 4), 'rows_per_pass': Bm,
                   'frac_timed_region': round((gflops * passes / elapsed_lanes / 1e12 / MFMA_BF16_PEAK_TFLOPS) if mfma_bound else (wbytes * passes / elapsed_lanes / 1e9 / HBM_PEAK_GBS), 4),
                   'traffic_source': (f'profiles/pmc_latest.json by_rows[{Bm}]: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE over a bounded run (2 positions) of THESE {Bm}-row kernels, '
                                      'committed with the repository; not collected by this invocation') if pmc_traffic('stream_gemm', Bm) else
                                     f'null: no committed counter pass at {Bm} rows per pass (tools/collect_round_evidence.sh collects them at the rows of the driver and the default schedule)',
                   'note': 'per-launch figure of the kernels the timed region runs (one pass = merge x batch rows; throughput policy when several lanes are in flight), measured one lane '
                           'at a time with per-launch HIP events; algorithmic FLOPs = 2 x rows x N x K of every nn.Linear of the reference (stage2/layers.py), algorithmic bytes = the bf16 '
                           'weights, streamed once per pass',
                   'ar_ms_one_pass_this_schedule': round(ar_ms, 3), 'ar_ms_serial_batch': round(ar_ms_serial, 3)}
            fam.append(rec)
        if conv_ms > 0:
            n_l = sum(v[0] for v in conv.values())
            ach = cflops / (conv_ms * 1e-3) / 1e12
            peak = MFMA_BF16_PEAK_TFLOPS if dec_prec != 'exact' else F32_PEAK_TFLOPS
            # matrix work actually issued per algorithmic FLOP: SPLIT evaluates a product with three fp16 MFMAs; the nearest-x2 upsampling convs run as four
            # 2x2 phase convolutions on the low-resolution image (4 taps instead of 9: the same sums, regrouped)
            issued = (3 if dec_prec == 'split' else 1) * work.get('dec_flops_executed', work['dec_flops']) / work['dec_flops']
            kname = {'split': 'conv3x3_split_ring16_kernel (+ conv2x2_split_up16_kernel for the upsampling convs, split_gemm_kernel for the 1x1 convs and the attention GEMMs; conv_out_direct_kernel -- norm_out + swish + conv_out as fp32 FMAs, 0.26 % of the FLOPs -- is counted in the time of the family): '
                              'HQ-VAE decoder conv family, fp16 hi/lo split operands',
                     'fast': 'conv3x3_halo_kernel (+ conv_glds_kernel for 1x1): HQ-VAE decoder conv family, bf16',
                     'exact': 'gemm_tile_kernel: HQ-VAE decoder conv family, fp32 vector ALUs'}[dec_prec]
            fam.append({'kernel': kname, 'bound': 'mfma', 'achieved': round(ach, 2),
                        'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4), 'traffic': pmc_traffic('decoder_conv'), 'launches': n_l,
                        'avg_launch_us': round(1000 * conv_ms / n_l, 3), 'total_ms': round(conv_ms, 3),
                        'algorithmic_flops_per_launch': round(cflops / n_l), 'matrix_flops_per_algorithmic_flop': round(issued, 4),
                        'achieved_issued': round(ach * issued, 2), 'frac_issued': round(ach * issued / peak, 4), 'images_per_pass': Bm,
                        'frac_timed_region': round(cflops * passes / elapsed_lanes / 1e12 / peak, 4),
                        'note': 'achieved / frac = ALGORITHMIC FLOPs (2 x MACs of the reference\'s nn.Conv2d calls, stage1/modules/layers.py) per launch / average launch duration, against the dense '
                                'f16/bf16 MFMA peak; achieved_issued / frac_issued count the MFMA work actually issued (SPLIT: three fp16 MFMAs per product term)'})
        fam.sort(key=lambda f: -f['total_ms'])
        for f in fam:               # a fraction above the peak means the FLOPs and the timed launches are not the same work: refuse to print it
            if not (0.0 < f['frac'] <= 1.0 and f.get('frac_issued', f['frac']) <= 1.0):
                raise RuntimeError(f"roofline record out of range ({f['kernel'][:40]}...: frac {f['frac']}, issued {f.get('frac_issued')}): the measured pass and its FLOP count disagree")
        if fam:
            out['roofline'] = fam[0]
            out['roofline_other'] = fam[1:]
        out['kernel_ms_per_pass'] = {k: [v[0], round(v[1], 3)] for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1])}
        out['kernel_ms_per_pass_note'] = f'[launches, ms] per pass of {Bm} images ({merge} steps), un-graphed, one lane'

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(cfg, s2, s1, B)
        if not txt_cond and not three and s2.n_layers == 12:
            # the reference ITSELF (PyTorch CPU, fp32, 8 threads) as measured once in the build container, BASELINE.md section 2 -- it cannot
            # run on the GPU box (its sources do not travel); quoted so that the port above can be compared with the real thing
            out['cpu_baseline_reference'] = {'value': 0.40, 'unit': 'images/s', 'cores': 8, 'kind': 'reference',
                                             'sample': 'kakaobrain/hqtransformer on torch CPU in the build container: sampling_ihqgpt 2.19 s per 8 of 64 positions at B = 8 '
                                                       '(2.2 s/image) + decode_code 0.334 s/image at batch 1 = 2.52 s/image (BASELINE.md section 2); not re-measured by this run'}

    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
