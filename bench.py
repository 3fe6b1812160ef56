#!/usr/bin/env python
"""bench.py - training samples/s of the caption hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = the complete optimisation step of BASELINE.json configs[1] on one synthetic batch:
ResNet-152 trunk (batch-stat BN) + RoBERTa-large (24 layers, 512 tokens) forward, 2-context
4-layer DynamicConv decoder forward + adaptive-softmax loss + backward, gradient all-reduce
(N > 1, RCCL) and BertAdam.  bf16 compute, fp32 master weights, random-init weights,
synthetic data of NYTimes800k shape (no network for datasets / checkpoints).
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


def cpu_baseline(sample_b=2, threads=None):
    """The oracle (CPU restatement of the reference, fp32) timed on a bounded sample of the same
    workload: the full configs[1] model, one optimisation step on `sample_b` samples."""
    from oracle.build import build_model
    from oracle.encoders import resnet152, roberta_large
    from oracle.optim import BertAdam
    import tell_amd  # noqa: F401  (only for the synthetic batch generator)
    from tell_amd.data import synthetic_batch
    if threads:
        torch.set_num_threads(threads)
    torch.manual_seed(0)
    model = build_model('flattened', resnet152(), roberta_large(), n_bert_layers=25).train()
    model.weigh_bert = False
    for n, p in model.named_parameters():
        if n.startswith('resnet') or n.startswith('roberta'):
            p.requires_grad_(False)
    opt = BertAdam([p for p in model.parameters()])
    batch = synthetic_batch(B=sample_b, article_len=512, caption_len=33, seed=1234)

    def step():
        opt.zero_grad()
        out = model(context={'roberta': batch['context']['roberta'].clone()}, image=batch['image'],
                    caption={'roberta': batch['caption']['roberta'].clone()})
        out['loss'].backward()
        opt.step()
    t0 = time.time()
    step()
    dt = time.time() - t0
    cpu = 'unknown'
    try:
        with open('/proc/cpuinfo') as f:
            cpu = next((ln.split(':', 1)[1].strip() for ln in f if ln.startswith('model name')), cpu)
    except OSError:
        pass
    return {'value': round(sample_b / dt, 4), 'unit': 'samples/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'cpu_model': cpu,
            'sample': '1 full optimisation step (ResNet-152 + RoBERTa-large fwd, 2-ctx decoder fwd+loss+bwd, '
                      'BertAdam) of the fp32 CPU oracle on %d samples, %.1f s' % (sample_b, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=16, help='samples per GPU (configs[1]: 16)')
    ap.add_argument('--model', default='flattened', choices=['flattened', 'faces_objects'])
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample', type=int, default=2)
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-pipeline', action='store_true',
                    help='do not launch the next batch\'s frozen encoders underneath the current decoder step')
    ap.add_argument('--serial', action='store_true',
                    help='run everything on ONE stream (no encoder prefetch / overlap, no weight-gradient or update '
                         'stream): per-kernel durations are then well defined - the mode of the roofline leg and of '
                         'the committed rocprofv3 kernel summaries')
    ap.add_argument('--roofline-steps', type=int, default=4,
                    help='extra single-stream steps after the timed region that time the GEMM kernels in isolation')
    args = ap.parse_args()

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
        dist.init_process_group(os.environ.get('TELL_DP_BACKEND', 'nccl'),      # RCCL on ROCm
                                **({'device_id': dev} if os.environ.get('TELL_DP_BACKEND', 'nccl') == 'nccl' and
                                   os.environ.get('TELL_DP_LAZY') != '1' else {}))
    import tell_amd
    from tell_amd import prof
    from tell_amd.build import build_model
    from tell_amd.data import synthetic_batch
    from tell_amd.training import Trainer
    tell_amd.hip.require_gpu()
    tell_amd.set_compute_dtype(torch.bfloat16 if args.dtype == 'bf16' else torch.float32)
    tell_amd.manual_seed(1234 + rank)
    torch.manual_seed(0)                                     # same initial weights on all ranks

    model = build_model(args.model, weigh_bert=(args.model == 'faces_objects'))
    trainer = Trainer(model, device=dev)
    batches = [synthetic_batch(args.batch, 512, 33, args.model == 'faces_objects', seed=1234 + rank + 97 * i,
                               device=dev) for i in range(2)]

    def fresh(b):
        return {k: (dict(v) if isinstance(v, dict) else v) for k, v in b.items()}

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    wgrad_default = tell_amd.ops._WGRAD['enabled']

    def set_serial(flag):
        """One stream for everything (flag) or the overlapped production schedule (not flag)."""
        import tell_amd.models.transformer as tr_mod
        from tell_amd import ops
        torch.cuda.synchronize()
        tell_amd.runtime.wait_weight_update()
        tr_mod._OVERLAP = not flag
        tell_amd.graphs.ENABLED = not flag         # eager encoders: every GEMM launch passes the timing hook
        ops._WGRAD['enabled'] = (not flag) and wgrad_default
        trainer.async_update = (not flag) and trainer.update_stream is not None

    if args.serial:
        set_serial(True)
        args.no_pipeline = True

    # every step trains batch i and launches the frozen encoders of batch i+1 underneath it (what a training loop
    # with a data loader does); each timed step therefore contains exactly one encoder pass and one decoder pass
    nxt = lambda i: None if args.no_pipeline else batches[(i + 1) % 2]     # noqa: E731
    for i in range(args.warmup):
        trainer.train_one_batch(fresh(batches[i % 2]), next_batch=nxt(i))
    sync()
    if not args.no_roofline:
        prof.calibrate()
        prof.enable(True)
    t0 = time.perf_counter()
    loss = None
    for i in range(args.steps):
        loss = trainer.train_one_batch(fresh(batches[(args.warmup + i) % 2]), next_batch=nxt(args.warmup + i))
    issued = time.perf_counter() - t0           # host finished issuing; the rest of `elapsed` is GPU backlog
    sync()
    elapsed = time.perf_counter() - t0
    prof_concurrent = prof.summary() if not args.no_roofline else {}
    prof.enable(False)
    # ---- roofline leg: the same steps on ONE stream, so that a kernel's event-bracketed duration is its own
    # (in the overlapped schedule above several streams share the CUs and every duration is inflated)
    prof_summary = prof_concurrent
    if not args.no_roofline and not args.serial and args.roofline_steps > 0:
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
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        value = world * args.batch * args.steps / elapsed
        result = {
            'metric': 'training samples/sec (img+article->caption)', 'value': round(value, 2),
            'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'host_issue_ms_per_step': round(1e3 * issued / args.steps, 3),
            'dtype': args.dtype, 'data': 'synthetic (random pixels, random BPE ids; random-init weights)',
            'config': {'workload': 'BASELINE configs[1]: 4-layer DynamicConv decoder (2 contexts: ResNet-152 '
                                   'image regions + RoBERTa-large article), batch %d/GPU, 512-token articles, '
                                   '32 caption steps; full step = frozen encoders fwd + decoder fwd/loss/bwd '
                                   '+ BertAdam' % args.batch if args.model == 'flattened' else
                                   'BASELINE configs[2] shape: faces+objects model, batch %d/GPU' % args.batch,
                       'global_batch': world * args.batch, 'article_len': 512, 'caption_len': 33,
                       'parallelism': 'dp%d' % world, 'final_loss_bits': round(float(loss), 4),
                       'peak_hbm_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
                       'resnet_hipgraph': sorted({e['state'] for e in getattr(model.__dict__.get('_resnet_graph'),
                                                                              'entries', {}).values()})},
        }
        if prof_summary:
            # dominant kernel = largest estimated total time (avg of the timed samples x all its launches)
            name, d = max(prof_summary.items(), key=lambda kv: kv[1]['avg_us'] * kv[1]['launches'])
            achieved = d['work'] / (d['total_ms'] * 1e-3) / 1e12
            peak = 2500.0 if 'bf16' in name.split(',')[0] else 157.3
            traffic = None
            pmc = os.path.join(ROOT, 'profiles', 'r01_pmc_gemm_traffic.json')
            if os.path.exists(pmc):      # HBM bytes per launch of this kernel from the committed PMC passes
                j = json.load(open(pmc))
                if j.get('kernel') == name:
                    traffic = j['traffic_bytes_per_launch']
            result['roofline'] = {
                'bound': 'mfma', 'kernel': name, 'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s',
                'frac': round(achieved / peak, 4), 'traffic': traffic,
                'traffic_note': 'bytes/launch, rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE on this '
                                'command, profiles/r01_pmc_gemm_traffic.json',
                'mode': ('single stream (--serial run)' if args.serial else
                         '%d extra single-stream steps after the timed region' % args.roofline_steps),
                'launches_per_step': d['launches'] // (args.steps if args.serial else args.roofline_steps),
                'avg_launch_us': round(d['avg_us'], 2),
                'timed_launches': d['timed'],
                'timing': 'HIP events around every 3rd launch of each GEMM kernel (>= 2 GFLOP), on the launch stream; '
                          'event_overhead_us (bracket around a 1-element kernel minus its 1.5 us) is subtracted',
                'event_overhead_us': round(prof.overhead_us(), 2),
                'concurrent': ({k: {'avg_us': round(v['avg_us'], 2),
                                    'tflops': round(v['work'] / (v['total_ms'] * 1e-3) / 1e12, 2)}
                                for k, v in prof_concurrent.items()} if not args.serial else None),
                'concurrent_note': 'the same kernels timed inside the timed region, where three streams share the CUs',
                'all_gemm_kernels': {k: {'avg_us': round(v['avg_us'], 2),
                                         'launches_per_step': v['launches'] // (args.steps if args.serial else args.roofline_steps),
                                         'tflops': round(v['work'] / (v['total_ms'] * 1e-3) / 1e12, 2)}
                                     for k, v in prof_summary.items()}}
        if args.model == 'flattened' and args.dtype == 'bf16':
            # whole-step MFMA utilisation (BASELINE.json metric): algorithmic GEMM-class work per sample from the module
            # shapes: RoBERTa-large fwd 335 GF, ResNet-152 fwd 23 GF, 2-context decoder training step
            # (4.92 per-token + 0.90 head) x 3 (fwd + dgrad + wgrad) + 10.23 K/V projections x 2 (no dX) = 37.9 GF
            gf = 335.0 + 23.0 + 37.9
            tf = gf * 1e-3 * world * args.batch / (ms * 1e-3)
            result['step_mfma'] = {'gflop_per_sample': gf, 'achieved': round(tf, 1), 'peak': 2500.0 * world,
                                   'unit': 'TFLOP/s', 'frac': round(tf / (2500.0 * world), 4)}
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(args.cpu_sample)
        print(json.dumps(result))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
