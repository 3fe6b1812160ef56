#!/usr/bin/env python
"""bench.py - training samples/s of the caption hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (default) = BASELINE.json configs[2], the per-GPU shape of the 1 -> 8 GPU series (configs[3]):
`expt/nytimes/9_transformer_objects` - full faces+objects model, 4 contexts, weigh_bert, batch 32 per GPU, 512-token
articles, 33-token captions.  One "step" = the complete optimisation step on one synthetic batch: ResNet-152 trunk
(batch-stat BN) + RoBERTa-large (24 layers, 512 tokens) forward, 4-context 4-layer DynamicConv decoder forward +
adaptive-softmax loss + backward, gradient all-reduce (N > 1, RCCL) and BertAdam.  bf16 compute, fp32 master weights,
random-init weights, synthetic data of NYTimes800k shape (no network for datasets / checkpoints).
`--model flattened --batch 16` is configs[1]; at N = 1 it is also measured as the `secondary` block of the same line.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic GEMM-class work per sample (from the module shapes; 1 MAC = 2 FLOP):
#   RoBERTa-large fwd 335 GF, ResNet-152 fwd 23 GF, decoder training step (fwd + dgrad + wgrad of the per-token and
#   head parts, fwd + wgrad of the context K/V projections): 4 contexts 47 GF, 2 contexts 37.9 GF
GF = {'faces_objects': {'encoders': 358.0, 'decoder': 47.0}, 'flattened': {'encoders': 358.0, 'decoder': 37.9}}
WORKLOAD = {
    'faces_objects': 'BASELINE configs[2] (= per-GPU shape of configs[3]): expt/nytimes/9_transformer_objects - full '
                     'faces+objects model (4 contexts: ResNet-152 image regions, RoBERTa-large article with the '
                     '25-layer weigh_bert mix, 4 faces, 64 objects), batch %d/GPU, 512-token articles, 32 caption '
                     'steps; full step = frozen encoders fwd + decoder fwd/loss/bwd + BertAdam',
    'flattened': 'BASELINE configs[1]: expt/nytimes/5_transformer_roberta shape - 4-layer DynamicConv decoder (2 '
                 'contexts: ResNet-152 image regions + RoBERTa-large article), batch %d/GPU, 512-token articles, 32 '
                 'caption steps; full step = frozen encoders fwd + decoder fwd/loss/bwd + BertAdam',
}


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            return next((ln.split(':', 1)[1].strip() for ln in f if ln.startswith('model name')), 'unknown')
    except OSError:
        return 'unknown'


def _physical_cores():
    """Physical cores of the host: distinct (physical id, core id) pairs of /proc/cpuinfo (None if it cannot be read)."""
    try:
        seen, phys = set(), None
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('physical id'):
                phys = ln.split(':')[1].strip()
            elif ln.startswith('core id'):
                seen.add((phys, ln.split(':')[1].strip()))
        return len(seen) or None
    except OSError:
        return None


def measure_gemm_traffic(kernel, batch):
    """HBM bytes per launch of the dominant GEMM kernel, MEASURED in this run: two rocprofv3 passes (--pmc FETCH_SIZE,
    --pmc WRITE_SIZE; counters in their own runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes) over
    tools/pmc_gemm_mix.py - that kernel's launches of one step and nothing else - averaged with the step's launch mix;
    FETCH_SIZE x2 on gfx950 (64 B counted per 128 B request).  -> (bytes, note) or (None, None) when rocprofv3 is not on
    PATH / a pass fails (the caller then falls back to the committed profile and says so)."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which('rocprofv3')
    if exe is None:
        return None, None
    script = os.path.join(ROOT, 'tools', 'pmc_gemm_mix.py')
    per_group = {}
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = tempfile.mkdtemp(prefix='tell_pmc_', dir='/tmp')
            env = dict(os.environ, TMPDIR='/tmp')
            r = subprocess.run([exe, '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', d, '-o', 'p', '--',
                                sys.executable, script, str(batch)], cwd='/tmp', env=env, capture_output=True, text=True,
                               timeout=120)
            groups = [ln.split() for ln in r.stdout.splitlines() if ln.startswith('MIX ')]
            files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
            if r.returncode != 0 or not groups or not files:
                return None, None
            rows = [row for row in csv.DictReader(open(files[0]))
                    if row['Counter_Name'] == counter and 'gemm_nt_q4_kernel' in row['Kernel_Name']]
            key = 'Dispatch_Id' if rows and 'Dispatch_Id' in rows[0] else 'Correlation_Id'
            rows.sort(key=lambda row: int(row[key]))
            if len(rows) != 3 * len(groups):
                return None, None
            for gi, grp in enumerate(groups):
                vals = [float(row['Counter_Value']) for row in rows[3 * gi:3 * gi + 3]]
                per_group.setdefault(grp[1], {'per_step': int(grp[5])})[counter] = sum(vals[1:]) / 2.0       # (first launch: cold)
            shutil.rmtree(d, ignore_errors=True)
    except Exception:                                    # noqa: BLE001 - measurement aid: never fail the bench line
        return None, None
    tot = sum(v['per_step'] for v in per_group.values())
    kb = sum(v['per_step'] * (2.0 * v['FETCH_SIZE'] + v['WRITE_SIZE']) for v in per_group.values()) / tot
    note = ('bytes/launch MEASURED IN THIS RUN: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) '
            'over tools/pmc_gemm_mix.py, FETCH_SIZE x2 (gfx950), averaged with the step\'s launch mix ' +
            ', '.join('%s x%d: %.1f MB' % (k, v['per_step'], (2.0 * v['FETCH_SIZE'] + v['WRITE_SIZE']) / 1024.0)
                      for k, v in per_group.items()))
    return int(kb * 1024), note


def cpu_baseline(model_name='faces_objects', sample_b=4, gen_b=8, budget_s=30.0):
    """The oracle (CPU restatement of the reference, fp32; kind "port") timed on bounded samples of the same workload,
    SURVEY.md 8d: (a) the full optimisation step of the benchmarked model (weigh_bert as benchmarked), (b) the
    decoder-only step (encoder outputs given), with all host threads and with one, (c) greedy generation of 8 samples
    with the reference's control flow (decoder only).  Every leg is warmed once and timed over >= 2 repetitions."""
    from oracle.build import build_model
    from oracle.encoders import resnet152, roberta_large
    from oracle.optim import BertAdam
    import tell_amd  # noqa: F401  (only for the synthetic batch generator)
    from tell_amd.data import synthetic_batch
    fo = model_name == 'faces_objects'
    # torch's CPU kernels stop scaling on this workload well before a 128-thread host is used up (measured on the
    # MI355X box's 2 x EPYC 9575F, fwd+bwd of 4 samples: 16 threads 3.9 s, 32: 3.9 s, 64: 6.4 s, 128: 13.4 s -
    # tools/cpu_thread_sweep.py), so the baseline runs on min(32, available) threads: its best configuration
    cores = min(32, torch.get_num_threads())
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    t_all = time.time()
    model = build_model(model_name, resnet152(), roberta_large(), n_bert_layers=25).train()
    model.weigh_bert = fo                                   # as benchmarked (config 9: true, config 5: false)
    for n, p in model.named_parameters():
        if n.startswith('resnet') or n.startswith('roberta'):
            p.requires_grad_(False)
    opt = BertAdam([p for p in model.parameters() if p.requires_grad])
    batch = synthetic_batch(B=sample_b, article_len=512, caption_len=33, faces_objects=fo, seed=1234)

    def clone(b):
        return {k: ({kk: vv.clone() for kk, vv in v.items()} if isinstance(v, dict) else v.clone())
                for k, v in b.items()}

    def full_step():
        opt.zero_grad()
        model(**clone(batch))['loss'].backward()
        opt.step()

    def timed(fn, reps):
        fn()                                                # warm (first touch of 1.6 GB of weights, thread pools)
        t0 = time.time()
        for _ in range(reps):
            fn()
        return (time.time() - t0) / reps

    legs = {}
    dt = timed(full_step, 2)
    legs['full_step'] = {'samples_per_s': round(sample_b / dt, 3), 's_per_step': round(dt, 3), 'batch': sample_b,
                         'threads': cores}
    # ---- decoder-only step: the encoders' outputs are given (they are frozen; SURVEY 6 probe shape)
    with torch.no_grad():
        b = clone(batch)
        _, _, contexts = model._forward(b['context'], b['image'], b['caption'], b.get('face_embeds'),
                                        b.get('obj_embeds'))
        contexts = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in contexts.items()}
    cap = batch['caption']['roberta']

    def dec_step():
        opt.zero_grad()
        out = model.decoder({'roberta': cap[:, :-1]}, contexts)
        loss, n = model.criterion(model.decoder.adaptive_softmax, out, cap[:, 1:])
        (loss / n).backward()
        opt.step()
    dt = timed(dec_step, 2)
    legs['decoder_step'] = {'samples_per_s': round(sample_b / dt, 3), 's_per_step': round(dt, 3), 'batch': sample_b,
                            'threads': cores}
    if time.time() - t_all < budget_s:
        torch.set_num_threads(1)
        dt1 = timed(dec_step, 1)
        torch.set_num_threads(cores)
        legs['decoder_step_1thread'] = {'samples_per_s': round(sample_b / dt1, 3), 's_per_step': round(dt1, 3),
                                        'batch': sample_b, 'threads': 1}
    # ---- greedy generation, the reference's control flow (K/V projections recomputed every step)
    if time.time() - t_all < budget_s:
        model.eval()
        gb = synthetic_batch(B=gen_b, article_len=512, caption_len=33, faces_objects=fo, seed=99)
        with torch.no_grad():
            _, _, gctx = model._forward(gb['context'], gb['image'], gb['caption'], gb.get('face_embeds'),
                                        gb.get('obj_embeds'))
            steps = 12                                         # bounded: 12 of the <= 100 steps, per-step cost is flat
            t0 = time.time()
            model._generate(gb['caption']['roberta'][:, :1], gctx, gen_len=steps, eos=-1)
            dt = time.time() - t0
        legs['greedy_generation'] = {'captions_per_s_at_100_steps': round(gen_b / (dt / steps * 100), 4),
                                     's_per_token_step': round(dt / steps, 3), 'batch': gen_b, 'threads': cores,
                                     'note': 'decoder only, %d steps timed, scaled to the 100-step cap' % steps}
    return {'value': legs['full_step']['samples_per_s'], 'unit': 'samples/s', 'cores': cores, 'kind': 'port',
            'threads_used': cores, 'host_logical_cpus': os.cpu_count(), 'host_physical_cores': _physical_cores(),
            'cores_note': '`cores` = the threads the baseline ran on (its best configuration: torch stops scaling on this '
                          'workload at 16-32 threads, tools/cpu_thread_sweep.py); the host has `host_physical_cores` cores',
            'cpu_model': _cpu_model(),
            'sample': 'fp32 CPU oracle (torch, %d threads): 2 warmed full optimisation steps of the %s model '
                      '(ResNet-152 + RoBERTa-large fwd, decoder fwd+loss+bwd, BertAdam) on %d samples; other legs '
                      'in `legs`' % (cores, model_name, sample_b),
            'legs': legs, 'wall_s': round(time.time() - t_all, 1)}


def dp_selftest(args):
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, TELL_DP_SELFTEST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1',
               LOCAL_RANK='0')
    cmd = [sys.executable, os.path.abspath(__file__), '--steps', '10', '--warmup', '4', '--no-cpu-baseline', '--no-secondary',
           '--no-generation', '--no-loader', '--no-roofline', '--no-dp-selftest']
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')][-1]
        j = json.loads(line)
        return {'value': j['value'], 'unit': j['unit'], 'ms_per_step': j['ms_per_step'], 'steps': j['steps'], 'warmup': j['warmup'],
                'dp': j.get('dp'), 'note': 'configs[2] step with TELL_DP_SELFTEST=1: 1-rank RCCL group, token-count / non-finite '
                'flag / gradient all-reduce (bf16 wire), BertAdam reading the wire buffer; separate process'}
    except Exception as e:                                   # the headline line must not depend on this leg
        return {'error': repr(e)[:300]}


def hog_child(n):
    """bench.py --cu-hog: hold n CUs with back-to-back 50 ms tell_cu_hog launches until the parent closes our stdin."""
    import threading
    import tell_amd
    from tell_amd import hip
    torch.cuda.set_device(0)
    hip.require_gpu()
    done = threading.Event()
    threading.Thread(target=lambda: (sys.stdin.read(), done.set()), daemon=True).start()
    ticks = 50 * hip.lib().tell_wall_clock_khz()
    for _ in range(4):
        hip.call('tell_cu_hog', n, ticks, None)
    print('ready', flush=True)
    t_end = time.time() + 120
    while not done.is_set() and time.time() < t_end:
        hip.call('tell_cu_hog', n, ticks, None)
        hip.call('tell_cu_hog', n, ticks, None)
        torch.cuda.current_stream().synchronize()       # (two launches deep: the CUs are re-taken within microseconds)
        hip.call('tell_cu_hog', n, ticks, None)
    torch.cuda.synchronize()


def relaunch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    if torch.cuda.device_count() < args.gpus and env.get('TELL_BENCH_ONE_GPU') == '1':
        env.setdefault('TELL_DP_BACKEND', 'gloo')          # rehearsal: every rank on cuda:0, gradients over gloo
    sys.exit(subprocess.call(cmd, env=env))


def measure(args, model_name, batch_size, dev, world, rank, dist, roofline=True):
    """Build the model, run warmup + timed steps (+ the single-stream roofline leg, + the decoder-alone leg)."""
    import tell_amd
    from tell_amd import prof
    from tell_amd.build import build_model
    from tell_amd.data import synthetic_batch
    from tell_amd.training import Trainer
    fo = model_name == 'faces_objects'
    torch.manual_seed(0)                                     # same initial weights on all ranks
    tell_amd.manual_seed(1234 + rank)
    model = build_model(model_name, weigh_bert=fo)
    trainer = Trainer(model, device=dev, capture_after=1)       # fixed shapes: capture at first sight
    batches = [synthetic_batch(batch_size, 512, 33, fo, seed=1234 + rank + 97 * i, device=dev) for i in range(2)]

    def fresh(b):
        return {k: (dict(v) if isinstance(v, dict) else v) for k, v in b.items()}

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    wgrad_default = tell_amd.ops._WGRAD['enabled']
    legs = []                                   # which parts of this function this rank ran (every rank must run the same)

    def set_serial(flag):
        """One stream for everything (flag) or the overlapped production schedule (not flag)."""
        import tell_amd.models.transformer as tr_mod
        from tell_amd import ops
        torch.cuda.synchronize()
        tell_amd.runtime.wait_weight_update()
        tr_mod._OVERLAP = not flag
        tell_amd.graphs.ENABLED = not flag         # eager everything: every GEMM launch passes the timing hook
        ops._WGRAD['enabled'] = (not flag) and wgrad_default
        trainer.async_update = (not flag) and trainer.update_stream is not None

    no_pipeline = args.no_pipeline
    if args.serial:
        set_serial(True)
        no_pipeline = True

    # every step trains batch i and launches the frozen encoders of batch i+1 underneath it (what a training loop
    # with a data loader does); each timed step therefore contains exactly one encoder pass and one decoder pass
    nxt = lambda i: None if no_pipeline else batches[(i + 1) % 2]     # noqa: E731
    want_prof = roofline and not args.no_roofline
    if want_prof:
        # enabled BEFORE the warm-up: the encoder / decoder hipGraphs are recorded there, and a launch inside a captured
        # graph can only be timed by what is captured with it (device timestamps around every 4th large GEMM, prof.py)
        prof.calibrate()
        prof.enable(True)
    for i in range(args.warmup):
        trainer.train_one_batch(fresh(batches[i % 2]), next_batch=nxt(i))
    sync()
    # clock settling: an idle MI355X needs a few hundred milliseconds of load before its clocks stop moving (the first
    # 20-step window of a fresh process read 1-3 % off the later ones); untimed, like the warm-up
    for i in range(getattr(args, 'burn_in', 0)):
        trainer.train_one_batch(fresh(batches[(args.warmup + i) % 2]), next_batch=nxt(args.warmup + i))
    sync()
    if want_prof:
        prof.reset_records()
    if trainer.dp:
        trainer.dp_timing = []
    hog = None
    if getattr(args, 'cu_hog', 0) > 0:
        # contention rehearsal: N workgroups that hold one CU each for the whole timed region (what RCCL's channel kernels
        # do during the gradient exchange), launched by a CHILD PROCESS: its context has hardware queues of its own.  (As a
        # fourth stream of this process the resident kernel shared a hardware queue with one of the step's streams -
        # ROCm multiplexes streams onto GPU_MAX_HW_QUEUES = 4 queues, FIFO each - and the step simply waited for it:
        # 220 ms per step for any N.  An RCCL communicator of this process takes a queue the same way: streams.py.)
        import subprocess
        hog = subprocess.Popen([sys.executable, os.path.abspath(__file__), '--hog-child', str(int(args.cu_hog))],
                               stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
        assert hog.stdout.readline().strip() == 'ready', 'cu-hog child did not start'
    # The timed region is a WINDOW of exactly --steps steps bracketed by barrier + synchronize on both sides.  It is run
    # `windows` times back to back and the MEDIAN window is the headline (`value`, `ms_per_step`); every window is in the
    # line (`windows_ms_per_step`, `spread`): one 0.4 s window cannot resolve a 2 % change on this pool.
    dec_ev = []
    loss = None
    wins = []
    k0 = args.warmup + getattr(args, 'burn_in', 0)
    for w in range(max(1, getattr(args, 'windows', 1))):
        dec_ev = []
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            k = k0 + w * args.steps + i
            if want_prof and not args.serial:                    # decoder half of the step, in situ (main stream)
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            loss = trainer.train_one_batch(fresh(batches[k % 2]), next_batch=nxt(k))
            if want_prof and not args.serial:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                dec_ev.append((e0, e1))
        issued_w = time.perf_counter() - t0       # host finished issuing; the rest of `elapsed` is GPU backlog
        sync()
        wins.append((time.perf_counter() - t0, issued_w))
    legs.append('timed_region')
    tw = torch.tensor([e for e, _ in wins], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)            # per window: the slowest rank
    win_s = [float(x) for x in tw.tolist()]
    order = sorted(range(len(win_s)), key=lambda j: win_s[j])
    med = order[(len(order) - 1) // 2]                        # (lower median for an even count)
    elapsed, issued = win_s[med], wins[med][1]
    if hog is not None:
        hog.stdin.close()
        hog.wait(timeout=60)
    graph_replays = trainer.step_graph.replays if trainer.step_graph is not None else 0
    prof_concurrent = prof.summary() if want_prof else {}
    if want_prof and not args.serial:
        prof_concurrent.update(prof.graph_summary())       # launches replayed from graphs: the last replay of each graph
    prof.enable(False)
    ms = 1e3 * elapsed / args.steps
    res = {'value': round(world * batch_size * args.steps / elapsed, 2), 'ms_per_step': round(ms, 3),
           'windows': len(win_s), 'windows_ms_per_step': [round(1e3 * x / args.steps, 3) for x in win_s],
           'first_window_value': round(world * batch_size * args.steps / win_s[0], 2),
           'spread': round((max(win_s) - min(win_s)) / elapsed, 4), 'burn_in_steps': getattr(args, 'burn_in', 0),
           'host_issue_ms_per_step': round(1e3 * issued / args.steps, 3), 'final_loss_bits': round(float(loss), 4),
           'peak_hbm_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
           'step_graph_replays': graph_replays, 'skipped_steps': trainer.skipped_steps(),
           'resnet_hipgraph': sorted({e['state'] for e in getattr(model.__dict__.get('_resnet_graph'),
                                                                  'entries', {}).values()})}
    if hog is not None:
        res['cu_hog'] = {'cus_held': int(args.cu_hog), 'note': 'N workgroups with 64 KB of LDS each held one CU for the whole '
                         'timed region (tell_cu_hog launched back to back by a child process): rehearsal of RCCL channel kernels next to the 256x256 GEMMs'}
    if trainer.dp:
        # the gradient exchange of every timed step on this rank (HIP events on the update stream), max over ranks
        times = trainer.dp_times()
        tt = torch.tensor([sum(a for a, _ in times) / max(len(times), 1), sum(b for _, b in times) / max(len(times), 1)],
                          dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        res['dp'] = {'allreduce_ms': round(float(tt[0]), 3), 'exposed_allreduce_ms': round(float(tt[1]), 3),
                     'wire_dtype': str(trainer.allreduce_dtype).replace('torch.', ''),
                     'gradient_mbytes': round(trainer.flat.total * (2 if trainer.allreduce_dtype == torch.bfloat16 else 4)
                                              / 1e6, 1),
                     'bucketed_during_backward': trainer.bucketed_reduces > 0,
                     'note': 'per step, max over ranks; exposed = the update stream idle-waiting for the last bucket '
                             '(the next batch\'s encoder streams keep running underneath)'}
    gf = GF[model_name]
    tf = (gf['encoders'] + gf['decoder']) * 1e-3 * world * batch_size / (ms * 1e-3)
    res['step_mfma'] = {'gflop_per_sample': gf['encoders'] + gf['decoder'], 'achieved': round(tf, 1),
                        'peak': 2500.0 * world, 'unit': 'TFLOP/s', 'frac': round(tf / (2500.0 * world), 4)}
    if not want_prof:
        _rank_report(res, trainer, legs, world, rank, dist, dev)
        return res, trainer
    # (with several ranks EVERY rank runs the legs below - their steps contain the gradient exchange - rank 0 reports)
    # ---- decoder-only step (the number north_star sets its MFMA target on): the decoder half alone on an idle GPU
    #      (encoder outputs already there), HIP events on its stream; and the same half inside the timed region,
    #      where it shares the CUs with the next batch's encoders
    dec = {'gflop_per_sample': gf['decoder']}
    if dec_ev:
        ins = sorted(a.elapsed_time(b) for a, b in dec_ev)
        dec['in_step_ms'] = round(ins[len(ins) // 2], 3)
    if not args.serial and world == 1:
        legs.append('decoder_alone')
        encs = []
        for b in batches:                       # both buffer slots of the encoder graphs
            encs.append(trainer.model.encode(b['context'], b['image']))
        torch.cuda.synchronize()
        for rep in range(2):                    # rep 0 warms (an eager pass if a signature is new)
            evs = []
            for i in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                trainer._prefetched = (batches[i % 2]['image'], encs[i % 2])
                e0.record()
                trainer.train_one_batch(fresh(batches[i % 2]))
                e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
        alone = sorted(a.elapsed_time(b) for a, b in evs)
        dec['alone_ms'] = round(alone[len(alone) // 2], 3)
        dtf = gf['decoder'] * 1e-3 * batch_size / (dec['alone_ms'] * 1e-3)
        dec.update(achieved=round(dtf, 1), peak=2500.0, unit='TFLOP/s', frac=round(dtf / 2500.0, 4),
                   note='decoder fwd + loss + bwd + BertAdam replayed as one hipGraph on an otherwise idle GPU; '
                        'frac = algorithmic decoder GEMM work / that time / dense bf16 MFMA peak')
    res['decoder_step'] = dec
    # ---- roofline leg: the same steps on ONE stream, eager, so that a kernel's event-bracketed duration is its own
    # (in the overlapped schedule several streams share the CUs and every duration is inflated)
    prof_summary = prof_concurrent
    if not args.serial and args.roofline_steps > 0:
        legs.append('roofline_leg')
        set_serial(True)
        trainer.train_one_batch(fresh(batches[0]))
        sync()
        prof.enable(True)
        for i in range(args.roofline_steps):
            trainer.train_one_batch(fresh(batches[i % 2]))
        sync()
        prof_summary = prof.summary()
        prof.enable(False)
        set_serial(False)
    if prof_summary:
        nsteps = args.steps if args.serial else args.roofline_steps
        # dominant kernel = largest estimated total time (avg of the timed samples x all its launches)
        name, d = max(prof_summary.items(), key=lambda kv: kv[1]['avg_us'] * kv[1]['launches'])
        achieved = d['work'] / (d['total_ms'] * 1e-3) / 1e12
        peak = 2500.0 if 'bf16' in name.split(',')[0] else 157.3
        traffic, tnote = None, None
        if not args.no_pmc and model_name == 'faces_objects' and name.startswith('gemm_nt_q4_kernel') and rank == 0:
            traffic, tnote = measure_gemm_traffic(name, args.batch or 32)
        for fn in (() if traffic else ('r06_pmc_gemm_traffic.json', 'r05_pmc_gemm_traffic.json', 'r04_pmc_gemm_traffic.json')):
            pmc = os.path.join(ROOT, 'profiles', fn)
            if os.path.exists(pmc):      # HBM bytes per launch of this kernel from the committed PMC passes
                j = json.load(open(pmc))
                if j.get('kernel') == name and j.get('model', 'flattened') == model_name:
                    traffic = j['traffic_bytes_per_launch']
                    tnote = 'bytes/launch, rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE on this ' \
                            'command, profiles/' + fn
                    break
        conc = prof_concurrent.get(name) if not args.serial else None
        iso = {'achieved': round(achieved, 2), 'frac': round(achieved / peak, 4), 'avg_launch_us': round(d['avg_us'], 2),
               'timed_launches': d['timed']}
        if d['timed']:       # the same figure WITHOUT the event-overhead subtraction (the conservative reading)
            raw_ms = d['total_ms'] + d['timed'] * prof.overhead_us() * 1e-3
            raw_tf = d['work'] / (raw_ms * 1e-3) / 1e12
            iso['frac_without_overhead_subtraction'] = round(raw_tf / peak, 4)
            iso['avg_launch_us_without_overhead_subtraction'] = round(d['avg_us'] + prof.overhead_us(), 2)
        if conc:        # headline = the kernel INSIDE the timed region (three streams share the CUs there)
            c_tf = conc['work'] / (conc['total_ms'] * 1e-3) / 1e12
            head = {'achieved': round(c_tf, 2), 'frac': round(c_tf / peak, 4), 'avg_launch_us': round(conc['avg_us'], 2),
                    'timed_launches': conc['timed']}
        else:
            head = iso
        res['roofline'] = {
            'bound': 'mfma', 'kernel': name, 'achieved': head['achieved'], 'peak': peak, 'unit': 'TFLOP/s',
            'frac': head['frac'], 'traffic': traffic, 'traffic_note': tnote,
            'mode': ('single stream (--serial run): HIP events' if args.serial else
                     'in the timed region: this kernel as replayed from the encoder / decoder hipGraphs while three '
                     'streams share the CUs - every 4th launch records its own execution span (first workgroup in -> '
                     'last workgroup out, device wall clock at 100 MHz, csrc/gemm.hip gemm_ts_*; HIP events cannot be '
                     'recorded inside a captured graph on ROCm), last replay of each graph; `isolated` = the same '
                     'kernel in %d extra single-stream eager steps after the timed region (HIP events)'
                     % args.roofline_steps) if conc else
                    ('%d extra single-stream eager steps after the timed region (HIP events)' % args.roofline_steps),
            'launches_per_step': d['launches'] // nsteps, 'avg_launch_us': head['avg_launch_us'],
            'timed_launches': head['timed_launches'],
            'isolated': iso if conc else None,
            # box-speed normaliser: samples/s per TFLOP/s that the dominant GEMM reaches ALONE on this box (boxes of the
            # pool differ by +-3 % in both; their ratio is what a code change moves)
            'value_per_isolated_tflops': round(res['value'] / world / max(iso['achieved'], 1e-9), 4),
            'timing': 'HIP events around every 3rd launch of each GEMM kernel (>= 2 GFLOP), on the launch stream; '
                      'event_overhead_us (bracket around a 1-element kernel minus its 1.5 us) is subtracted',
            'event_overhead_us': round(prof.overhead_us(), 2),
            'all_gemm_kernels': {k: {'avg_us': round(v['avg_us'], 2), 'launches_per_step': v['launches'] // nsteps,
                                     'tflops': round(v['work'] / (v['total_ms'] * 1e-3) / 1e12, 2)}
                                 for k, v in prof_summary.items()}}
    _rank_report(res, trainer, legs, world, rank, dist, dev)
    return res, trainer


def _rank_report(res, trainer, legs, world, rank, dist, dev):
    """Data parallel only: which legs every rank ran (a rank that skips a leg whose steps contain the gradient exchange
    deadlocks the others - round 3 found exactly that) and whether the replicas still hold bit-identical weights after all
    of them (fp32 masters: an exact integer checksum of their bit patterns + their sum)."""
    if world <= 1 or not getattr(trainer, 'dp', False):
        return
    tell = sys.modules['tell_amd']
    tell.runtime.wait_weight_update()
    torch.cuda.synchronize()
    flat = trainer.flat.flat
    bits = flat.view(torch.int32).to(torch.int64)
    mine = {'rank': rank, 'legs': list(legs), 'weights_checksum': int(bits.sum().item()),
            'weights_xor_fold': int((bits * torch.arange(1, bits.numel() + 1, device=dev, dtype=torch.int64) % 1000003).sum().item()),
            'weights_sum': float(flat.double().sum().item())}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    same = all(e['weights_checksum'] == everyone[0]['weights_checksum'] and e['weights_xor_fold'] == everyone[0]['weights_xor_fold']
               for e in everyone)
    res.setdefault('dp', {}).update(
        ranks=everyone, weights_identical_across_ranks=bool(same),
        every_rank_ran_the_same_legs=all(e['legs'] == everyone[0]['legs'] for e in everyone))



def loader_bench(args, dev, n_batches=12, warm=4, variable=False):
    """the training step fed by the data plane instead of resident tensors - pre-tokenised `.npz` shards on
    tmpfs -> reader (`nytimes_faces_ner_matched`) -> BucketIterator -> collate (ids padded, faces / objects NaN-padded,
    uint8 pixels to the device + tell_image_normalize) -> train_one_batch with the NEXT batch's encoders launched
    underneath.  A background thread reads shards, builds instances and collates them into pinned host tensors one
    batch ahead (`collate_host`); the training thread only issues the asynchronous host->device copies + the normalize
    kernel (`to_device`).  -> samples/s over `n_batches` timed batches (bench shape: 512-token articles,
    33-token captions, 4 faces, 64 objects, batch 32).
    variable=True: article / caption lengths, face and object counts drawn as SURVEY 8d describes real data
    (L ~ U{128..512}, T+1 ~ U{9..41}, faces U{0..4}, objects U{0..64}); the BucketIterator sorts by length with padding
    noise, so batches differ in shape: the trainer pads to its shape buckets (128 article tokens, 16 caption tokens: Trainer(shape_buckets=(128, 16))) and
    captures a step graph per bucket signature at its second sighting - this leg times that policy."""
    import queue
    import shutil
    import tempfile
    import threading
    import numpy as np
    import tell_amd
    from tell_amd.build import build_model
    from tell_amd.data import BucketIterator, DatasetReader, write_shard
    from tell_amd.data.iterators import collate_host, to_device
    from tell_amd.training import Trainer
    B = args.batch
    root = tempfile.mkdtemp(prefix='tell_shards_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
    try:
        g = np.random.RandomState(7)
        total = B * (n_batches + warm)
        t0 = time.perf_counter()
        for sh in range(0, total, 128):
            samples = []
            for i in range(sh, min(sh + 128, total)):
                L, T = (g.randint(128, 513), g.randint(9, 42)) if variable else (512, 33)
                if variable == 'wide':                  # captions of 5..98 tokens, articles of 32..512: 4 x 7 bucket pairs
                    L, T = g.randint(32, 513), g.randint(5, 99)
                F, O = (g.randint(0, 5), g.randint(0, 65)) if variable else (4, 64)
                cap = np.r_[0, g.randint(4, 50265, T - 2), 2]
                samples.append({'context_ids': np.r_[0, g.randint(4, 50265, L - 2), 2], 'caption_ids': cap,
                                'image': g.randint(0, 256, (224, 224, 3)).astype(np.uint8),
                                'face_embeds': g.randn(F, 512).astype(np.float32),
                                'obj_embeds': np.abs(g.randn(O, 2048)).astype(np.float32),
                                'metadata': {'caption': '', 'context': '', 'web_url': 'u%d' % i, 'image_path': '', 'image_pos': 0}})
            write_shard(os.path.join(root, 'train-%05d.npz' % (sh // 128)), samples)
        shard_mb = sum(os.path.getsize(os.path.join(root, f)) for f in os.listdir(root)) / 1e6
        write_s = time.perf_counter() - t0
        torch.manual_seed(0)
        model = build_model('faces_objects', weigh_bert=True)
        trainer = Trainer(model, device=dev, capture_after=2 if variable else 1)
        reader = DatasetReader.by_name('nytimes_faces_ner_matched')(use_objects=True, shard_dir=root)
        it = BucketIterator(sorting_keys=[['context', 'num_tokens'], ['caption', 'num_tokens']], batch_size=B)
        q = queue.Queue(maxsize=2)
        PIN = os.environ.get('TELL_LOADER_PIN', '0') == '1'      # page-locking 21 MB per batch costs more than it saves

        epochs = 3 if variable else 1

        def produce():                                  # shard decode + instance building, one batch ahead
            if variable:                                # the iterator's own batching: sorted by length, padding noise
                instances = list(reader._read('train'))
                for ep in range(epochs):
                    for group in it._batches(instances, shuffle=True):
                        if len(group) == B:
                            q.put(collate_host(group, pin=PIN))
                    q.put('epoch')
                q.put(None)
                return
            group = []
            for inst in reader._read('train'):
                group.append(inst)
                if len(group) == B:
                    q.put(collate_host(group, pin=PIN))
                    group = []
            q.put(None)
        th = threading.Thread(target=produce, daemon=True)
        th.start()
        host_ms, issue_ms, n_done, t_start = [], [], 0, None
        marks = []                                     # (steps done, wall clock) at every epoch end (variable leg)
        tok = [0, 0]                                   # padded article / caption tokens of the steps done so far
        tok_marks, tok_start = [], (0, 0)

        def fetch():
            while True:
                g_ = q.get()
                if isinstance(g_, str):                # epoch boundary
                    torch.cuda.synchronize()
                    marks.append((n_done + 1, time.perf_counter()))
                    tok_marks.append((tok[0] + int(cur['context']['roberta'].numel()), tok[1] + int(cur['caption']['roberta'].numel())))
                    continue
                return g_
        cur = to_device(fetch(), dev)
        with tell_amd.hip.bound_stream():
            while cur is not None:
                h0 = time.perf_counter()
                group = fetch()
                nxt = to_device(group, dev) if group is not None else None
                host_ms.append(1e3 * (time.perf_counter() - h0))
                h1 = time.perf_counter()
                n_art, n_cap = int(cur['context']['roberta'].numel()), int(cur['caption']['roberta'].numel())
                trainer.train_one_batch(cur, next_batch=nxt)
                issue_ms.append(1e3 * (time.perf_counter() - h1))
                n_done += 1
                tok[0] += n_art
                tok[1] += n_cap
                if n_done == warm:
                    torch.cuda.synchronize()
                    t_start = time.perf_counter()
                    tok_start = (tok[0], tok[1])
                cur = nxt
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t_start
        timed = n_done - warm
        host = sorted(host_ms[warm:])
        extra = {}
        if variable:
            sg = trainer.step_graph
            extra = {'step_graph_replays': sg.replays if sg is not None else 0, 'steps_total': n_done,
                     'shape_buckets': list(trainer.shape_buckets or ()), 'epochs': epochs}
            if sg is not None:                         # capture policy at work: signatures, captures kept, evictions, pool size
                ready = sum(1 for v in sg.entries.values() if v.get('state') == 'ready')
                extra.update(step_signatures_seen=len(sg.entries), step_graphs_kept=ready, step_graph_evictions=sg.cache.evictions, step_graph_freezes=sg.cache.freezes,
                             step_graph_failed=sum(1 for v in sg.entries.values() if v.get('state') == 'failed'),
                             hbm_reserved_gb=round(torch.cuda.memory_reserved() / 2 ** 30, 2))
            if len(marks) >= 3:                        # steady state: the last epoch (every bucket signature captured)
                (n1, t1), (n2, t2) = marks[-2], marks[-1]
                extra['first_epochs_value'] = round(B * timed / elapsed, 2)
                timed, elapsed = n2 - n1, t2 - t1
                tok_start, tok = tok_marks[-2], list(tok_marks[-1])
        # the WORK behind `value`: padded tokens per second through RoBERTa (article) and through the decoder (caption) - a
        # variable-length leg is comparable with the fixed-shape one through these, not through samples/s
        extra['article_tokens_per_s'] = round((tok[0] - tok_start[0]) / elapsed)
        extra['caption_tokens_per_s'] = round((tok[1] - tok_start[1]) / elapsed)
        extra['mean_article_len'] = round((tok[0] - tok_start[0]) / max(B * timed, 1), 1)
        extra['mean_caption_len'] = round((tok[1] - tok_start[1]) / max(B * timed, 1), 1)
        return {**extra, 'value': round(B * timed / elapsed, 2), 'unit': 'samples/s', 'ms_per_step': round(1e3 * elapsed / timed, 3),
                'batches': timed, 'host_ms_per_batch_on_the_training_thread': round(host[len(host) // 2], 2),
                'step_issue_ms': round(sorted(issue_ms[-timed:])[timed // 2], 2),
                'shard_mbytes': round(shard_mb, 1), 'shard_write_s': round(write_s, 1),
                'window_note': 'the timed window starts at a synchronisation point, where the encoders of its first batch have '
                               'already run (they are launched one step ahead), and the last batch has no successor to '
                               'encode: %d decoder passes, %d encoder passes - this leg reads about 1/%d high against the '
                               'headline, whose window holds as many encoder passes as steps' % (timed, timed - 1, timed),
                'pipeline': 'npz shards on tmpfs -> nytimes_faces_ner_matched reader -> collate_host into pinned tensors '
                            '(background thread, one batch ahead) -> async H2D + tell_image_normalize -> '
                            'Trainer.train_one_batch(next_batch=...)'}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def generate_bench(args, dev, world, rank, dist):
    """BASELINE configs[4]: caption generation throughput of the full faces+objects model (beam 4; the reference itself
    only decodes greedily, `--beam 1`), one replica per GPU, every rank decoding its own shard of synthetic images (no
    collective on the data path: replicas only).  A "step" is one batch of `--batch` captions: frozen encoders
    + K/V projection of the contexts + up to 100 decode steps (random-init weights never emit </s>: every caption runs
    the full 100 steps)."""
    import tell_amd
    from tell_amd.build import build_model
    from tell_amd.data import synthetic_batch
    torch.manual_seed(0)
    model = build_model('faces_objects').to(dev).eval()
    B, beam = args.batch, args.beam
    batches = [synthetic_batch(B, 512, 33, True, seed=4321 + rank + 97 * i, device=dev) for i in range(2)]

    def clone(b):
        return {k: (dict(v) if isinstance(v, dict) else v.clone()) for k, v in b.items()}
    out = None
    pipelined = not getattr(args, 'gen_serial', False)

    def run(n):
        """n batches: pipelined = the test-set loop of commands/evaluate.py (model.generate_stream: encoders of batch
        N+1 on their own streams underneath the decode loop of batch N); serial = one generate() after the other"""
        last = None
        if getattr(args, 'lanes', 1) > 1:
            for _, last in model.generate_lanes((clone(batches[i % 2]) for i in range(n)), beam_size=beam, lanes=args.lanes):
                pass
        elif pipelined:
            for _, last in model.generate_stream((clone(batches[i % 2]) for i in range(n)), beam_size=beam):
                pass
        else:
            for i in range(n):
                last = model.generate(**clone(batches[i % 2]), beam_size=beam)
        return last
    with tell_amd.hip.bound_stream():
        out = run(args.warmup)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = run(args.steps)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        serial_elapsed = None
        if pipelined:                              # the round-4 flow beside it: encoders, then the decode loop, per batch
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps):
                out = model.generate(**clone(batches[i % 2]), beam_size=beam)
            torch.cuda.synchronize()
            serial_elapsed = time.perf_counter() - t1
        # decode loop alone (contexts already encoded): HIP events on its stream -> time per decode step
        with torch.no_grad():
            cap_ids, _, contexts = model._forward(**{k: v for k, v in clone(batches[0]).items() if k != 'metadata'})
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _, ids, _ = model._generate(cap_ids, contexts, beam_size=beam)
            e1.record()
            torch.cuda.synchronize()
            dec_ms, n_steps = e0.elapsed_time(e1), ids.shape[1] - 1
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank != 0:
        return None
    # algorithmic HBM bytes of ONE decode step (bf16): the decoder's per-token weights + the tied softmax tables are
    # read once, the projected K/V of the four contexts once per SAMPLE (the beam's hypotheses are query positions)
    E, FF, H, V = 1024, 4096, 16, 50265
    per_layer = lambda K: (2 * E * E + H * K * E + E * E + 4 * 2 * E * E + 4 * E * E + 2 * FF * E)     # noqa: E731
    w_bytes = 2 * (sum(per_layer(K) for K in (3, 7, 15, 31)) + V * E + 2 * E * E + 2 * E)
    kv_bytes = 2 * 4 * B * (512 + 49 + 64 + 4 + 8) * 2 * E
    step_us = 1e3 * dec_ms / max(n_steps, 1)
    tbs = (w_bytes + kv_bytes) / (step_us * 1e-6) / 1e12
    traffic, tnote = None, None
    pmc = None
    for rnd in ('r06', 'r05', 'r04'):          # the newest committed PMC passes of the decode step
        cand = os.path.join(ROOT, 'profiles', '%s_pmc_generate_%s_traffic.json' % (rnd, 'beam%d' % beam if beam > 1 else 'greedy'))
        if os.path.exists(cand):
            pmc = cand
            break
    pmc = pmc or cand
    if os.path.exists(pmc) and B == 32:          # HBM bytes of one captured decode step from the committed PMC passes
        traffic = json.load(open(pmc))['traffic_bytes_per_step']
        tnote = 'bytes per decode step, rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE summed over the ' \
                'kernels of one step (tools/pmc_generate_traffic.sh), profiles/' + os.path.basename(pmc)
    return {
        'metric': 'caption generation throughput (img+article->caption), beam %d' % beam if beam > 1 else
                  'caption generation throughput (img+article->caption), greedy',
        'value': round(world * B * args.steps / elapsed, 2), 'unit': 'captions/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 2),
        'serial_value': round(world * B * args.steps / serial_elapsed, 2) if serial_elapsed else None,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
        'data': 'synthetic (random pixels, random BPE ids; random-init weights)',
        'config': {'workload': 'BASELINE configs[4]: caption generation, full faces+objects model '
                               '(expt/nytimes/9_transformer_objects), %s, %d captions per batch per GPU, up to 100 '
                               'steps, encoders included%s; replicas only (each rank decodes its own images)'
                               % ('beam %d' % beam if beam > 1 else 'greedy (sampling_topk 1, as the reference)', B,
                                  (' (%d decode loops in flight together on %d streams: generate_lanes)' % (args.lanes, args.lanes))
                                  if getattr(args, 'lanes', 1) > 1 else
                                  ' (batch N+1 encoded underneath the decode loop of batch N: generate_stream)'
                                  if pipelined else ''),
                   'global_batch': world * B, 'article_len': 512, 'decode_steps': int(n_steps),
                   'parallelism': 'replicas x%d' % world},
        'roofline': {'bound': 'hbm', 'kernel': 'captured decode step (one hipGraph replay per generated token)',
                     'achieved': round(tbs, 3), 'peak': 8.0, 'unit': 'TB/s', 'frac': round(tbs / 8.0, 4),
                     'traffic': traffic, 'traffic_note': tnote, 'avg_step_us': round(step_us, 1),
                     'algorithmic_bytes_per_step': int(w_bytes + kv_bytes),
                     'note': 'decoder per-token weights + tied softmax tables (%.0f MB) + projected K/V of the 4 '
                             'contexts read once per sample (%.0f MB), per decode step; duration = HIP events around '
                             'the decode loop / steps' % (w_bytes / 1e6, kv_bytes / 1e6)},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--windows', type=int, default=5, help='timed windows of --steps steps each (barrier + synchronize around '
                    'every one); the median window is the headline value')
    ap.add_argument('--burn-in', dest='burn_in', type=int, default=40, help='untimed clock-settling steps after the warm-up')
    ap.add_argument('--batch', type=int, default=None, help='samples per GPU (configs[2]: 32, configs[1]: 16)')
    ap.add_argument('--model', default='faces_objects', choices=['flattened', 'faces_objects'])
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the configs[1] block')
    ap.add_argument('--cpu-sample', type=int, default=4)
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-pmc', action='store_true', help='do not run the two rocprofv3 --pmc passes that measure roofline.traffic')
    ap.add_argument('--no-dp-selftest', action='store_true', help='skip the 1-rank RCCL leg of the default line')
    ap.add_argument('--cu-hog', type=int, default=0, help='hold N CUs with idle resident workgroups during the timed region (DP contention rehearsal)')
    ap.add_argument('--no-pipeline', action='store_true',
                    help='do not launch the next batch\'s frozen encoders underneath the current decoder step')
    ap.add_argument('--serial', action='store_true',
                    help='run everything on ONE stream, eagerly (no encoder prefetch / overlap, no graphs): per-kernel '
                         'durations are then well defined - the mode of the roofline leg and of the committed '
                         'rocprofv3 kernel summaries')
    ap.add_argument('--generate', action='store_true', help='BASELINE configs[4]: caption generation throughput')
    ap.add_argument('--no-generation', action='store_true', help='skip the configs[4] generation block of the default line')
    ap.add_argument('--no-many-signatures', action='store_true', help='skip the many-signature variable-length leg')
    ap.add_argument('--no-loader', action='store_true',
                    help='skip the data-plane leg (shards on tmpfs -> reader -> iterator -> collate -> training step)')
    ap.add_argument('--beam', type=int, default=4, help='beam size of --generate (1 = greedy, what the reference does)')
    ap.add_argument('--lanes', type=int, default=1, help='--generate: decode loops in flight together (CaptionModel.generate_lanes)')
    ap.add_argument('--gen-serial', action='store_true', help='--generate: encoders, then the decode loop, batch after '
                    'batch (the round-4 flow) instead of the pipelined test-set loop')
    ap.add_argument('--roofline-steps', type=int, default=3,
                    help='extra single-stream steps after the timed region that time the GEMM kernels in isolation')
    ap.add_argument('--hog-child', type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.hog_child:
        return hog_child(args.hog_child)
    if args.batch is None:
        args.batch = 32 if args.model == 'faces_objects' else 16
    args.warmup = max(args.warmup, 0)

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        relaunch(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    # (GPU_MAX_HW_QUEUES stays at its default of 4: the schedule uses three streams - backward/decoder, RoBERTa, ResNet -
    #  and an RCCL communicator brings the fourth; see tell_amd/streams.py for the measurements)
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('TELL_BENCH_ONE_GPU') == '1':      # rehearsal of the multi-rank control flow on a 1-GPU box:
        local_rank = 0                                   # every rank computes on cuda:0 (use TELL_DP_BACKEND=gloo)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    import tell_amd
    tell_amd.streams.warm(dev)         # before RCCL creates its streams: keeps ours on distinct hardware queues
    if world > 1 or os.environ.get('TELL_DP_SELFTEST') in ('1', '2'):   # '2': group only, no DP collectives
        backend = os.environ.get('TELL_DP_BACKEND', 'nccl')             # 'nccl' is RCCL on ROCm
        dist.init_process_group(backend, **({'device_id': dev} if backend == 'nccl' and
                                            os.environ.get('TELL_DP_LAZY') != '1' else {}))
    tell_amd.hip.require_gpu()
    tell_amd.set_compute_dtype(torch.bfloat16 if args.dtype == 'bf16' else torch.float32)

    if args.generate:
        if args.steps == 20 and args.warmup == 5:          # defaults are sized for training steps; a batch of captions
            args.steps, args.warmup = 4, 2                 # is ~0.1 s; the encoder graphs are captured at the 2nd sighting
        result = generate_bench(args, dev, world, rank, dist)
        if rank == 0:
            print(json.dumps(result))
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    res, trainer = measure(args, args.model, args.batch, dev, world, rank, dist)
    if rank == 0:
        result = {
            'metric': 'training samples/sec (img+article->caption)', 'value': res.pop('value'),
            'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': res.pop('ms_per_step'), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'host_issue_ms_per_step': res.pop('host_issue_ms_per_step'),
            'dtype': args.dtype, 'data': 'synthetic (random pixels, random BPE ids; random-init weights)',
            'config': {'workload': WORKLOAD[args.model] % args.batch, 'global_batch': world * args.batch,
                       'article_len': 512, 'caption_len': 33, 'parallelism': 'dp%d' % world,
                       'final_loss_bits': res.pop('final_loss_bits'), 'peak_hbm_gb': res.pop('peak_hbm_gb'),
                       'step_graph_replays': res.pop('step_graph_replays'), 'skipped_steps': res.pop('skipped_steps'),
                       'resnet_hipgraph': res.pop('resnet_hipgraph')},
        }
        result.update(res)                       # roofline, step_mfma, decoder_step
        if world == 1 and args.model == 'faces_objects' and not args.no_secondary and not args.serial \
                and args.dtype == 'bf16':
            # configs[1] in the same line (round-1 headline; kept as a secondary measurement)
            del trainer
            import gc
            gc.collect()
            tell_amd.ops.clear_weight_cache()
            torch.cuda.empty_cache()
            sec, tr2 = measure(args, 'flattened', 16, dev, world, rank, dist, roofline=False)
            result['secondary'] = {'workload': WORKLOAD['flattened'] % 16, 'value': sec['value'], 'unit': 'samples/s',
                                   'ms_per_step': sec['ms_per_step'],
                                   'host_issue_ms_per_step': sec['host_issue_ms_per_step'],
                                   'step_mfma': sec['step_mfma'], 'steps': args.steps, 'warmup': args.warmup}
            del tr2
        if world == 1 and args.model == 'faces_objects' and not args.no_generation and not args.serial \
                and args.dtype == 'bf16':
            # configs[4] in the same line: beam-4 (the config's metric) and greedy (what the reference decodes) caption
            # generation of the full model, encoders included, 3 timed batches of 32 captions x 100 steps each
            trainer = None
            import gc
            gc.collect()
            tell_amd.ops.clear_weight_cache()
            torch.cuda.empty_cache()
            result['generation'] = {}
            # (the reference's validation_iterator has batch_size 16; the batch is the iterator's knob, not the model's:
            #  32 = the figure every round has reported, 128 = the size this GPU's 288 GB are for)
            # (`_lanes2`: CaptionModel.generate_lanes - two caption batches decoded together on two streams, each with its own
            #  captured step; the encoders of a pair run first)
            for beam, gb, ln in ((4, 32, 1), (1, 32, 1), (4, 128, 1), (1, 128, 1), (4, 32, 2), (1, 32, 2), (4, 128, 2), (1, 128, 2)):
                ga = argparse.Namespace(**vars(args))
                ga.batch, ga.beam, ga.steps, ga.warmup, ga.lanes = gb, beam, (4 if gb == 32 else 3) * ln, 2 * ln, ln
                g = generate_bench(ga, dev, world, rank, dist)
                key = ('beam%d' % beam if beam > 1 else 'greedy') + ('' if gb == 32 else '_b%d' % gb) + ('' if ln == 1 else '_lanes%d' % ln)
                result['generation'][key] = {
                    'workload': g['config']['workload'], 'value': g['value'], 'unit': g['unit'],
                    'serial_value': g['serial_value'], 'lanes': ln,
                    'ms_per_batch': g['ms_per_step'], 'steps': g['steps'], 'warmup': g['warmup'],
                    'decode_steps': g['config']['decode_steps'], 'roofline': g['roofline']}
                gc.collect()
                tell_amd.ops.clear_weight_cache()
                torch.cuda.empty_cache()
        if world == 1 and not args.no_loader and not args.serial and args.model == 'faces_objects' and args.dtype == 'bf16':
            trainer = None
            import gc
            gc.collect()
            tell_amd.ops.clear_weight_cache()
            torch.cuda.empty_cache()
            result['loader'] = loader_bench(args, dev)
            result['loader']['resident_input_value'] = result['value']
            gc.collect()
            tell_amd.ops.clear_weight_cache()
            torch.cuda.empty_cache()
            result['loader_variable_lengths'] = loader_bench(args, dev, n_batches=22, warm=2, variable=True)
            if not args.no_many_signatures:
                # the capture policy under MANY shape signatures (captions of 5..98 tokens -> 7 caption buckets x 4 article
                # buckets x 2 encoder-slot parities), shuffled every epoch: signatures seen, graphs kept, evictions, freezes
                # of the thrash guard, reserved HBM; value = the last of 3 epochs, first_epochs_value = all of them
                gc.collect()
                tell_amd.ops.clear_weight_cache()
                torch.cuda.empty_cache()
                result['loader_many_signatures'] = loader_bench(args, dev, n_batches=38, warm=2, variable='wide')
        if world == 1 and not args.no_dp_selftest and not args.serial and args.model == 'faces_objects' and \
                os.environ.get('TELL_DP_SELFTEST') is None and not args.cu_hog:
            # the data-parallel code path's own cost at N = 1: the same step with a 1-rank RCCL group and every
            # collective, cast and the wire-reading BertAdam of the DP schedule, in its own process (an RCCL communicator
            # takes a hardware queue; the headline above runs without one)
            result['dp_selftest'] = dp_selftest(args)
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(args.model, args.cpu_sample)
        print(json.dumps(result))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
