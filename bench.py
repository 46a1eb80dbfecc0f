#!/usr/bin/env python3
"""Headline bench: whole-slide tile batches through the HIP compress -> decompress hot path.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A step = one pass of the hot path over one batch of synthetic 1024x1024x3 histology tiles that is
already resident in HBM: analysis conv stack + GDN -> quantise -> rANS encode (chunk bitstreams in
host memory) -> rANS decode -> dequantise -> synthesis deconv stack + IGDN -> uint8 tiles in HBM,
plus the per-tile rate/distortion record.  Tiles shard over ranks with no data-path collective
(weak scaling: the per-GPU batch is fixed); one all_gather of the per-tile statistics closes the
timed region.  Rank 0 prints ONE JSON line.

Arithmetic.  Default `f16x3`: every fp32 operand of the conv / GDN contraction is split into two f16
halves and a product is three v_mfma_f32_32x32x16_f16 (fp32 accumulate, 22 significant bits; range
guard with fp32 repeat, include/cae_hip.h).  `--precision fp32` = exact v_mfma_f32_32x32x2_f32.

Objects in the line besides the driver's contract:
  roofline      dominant fused kernel (conv/deconv + GDN).  `achieved` = MFMA FLOP/s the kernel ISSUES
                (f16x3: 3 per algorithmic FLOP) from the algorithmic FLOP of one launch / its HIP-event time
                (events recorded by the library around the launch, on the launch stream, inside the timed
                region); `peak` = dense peak of the instruction issued (f16: 2500, fp32: 157.3 TFLOP/s);
                `frac` = achieved / peak (<= 1 by construction).  `algorithmic_tflops` and
                `algorithmic_vs_fp32_mfma_peak` are the same time against the useful fp32 work.
                `traffic` = HBM bytes per launch from committed rocprofv3 PMC passes (offline; `traffic_source`).
  fp32_path     (N=1) a short run of the same workload on the exact-fp32 kernels
  tile256       (N=1) a short run on 256x256x3 tiles, 512 per step
  cpu_baseline  the CPU oracle (torch-CPU conv + restated GDN + C rANS) on a bounded sample of the
                same tiles on this host's cores -- a reported baseline, not the target
  parity_vs_cpu the GPU path against that oracle on the very same tiles: bpp, PSNR, bitstreams, pixels
  host_8cpu     (N=1) the same round trip in a child process restricted to 8 CPUs (--cpus 8): the host budget per GPU, measured
  low_rate_state (N=1) the same workload with a low-rate synthetic state (~0.9 bpp instead of 4): brackets the host coder's cost
  train         (N=1) BASELINE config 5: train.train_step on 256x256 patches, batch 16 and 128, against the mixed
                bf16-conv / fp32-GDN roofline
  dropin        (N=1) the reference's OWN call pattern: 1 / 8 / 16 Python threads calling codec.encode(chunk) /
                codec.decode(buf) one 1024^2 chunk at a time on one shared codec (dask's threaded scheduler,
                compress.py:121-128), host memory in, host memory out: tiles/s of each direction, next to the
                pipelined SlideCoder number above (which codes AND decodes every tile)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak (= vector peak)
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 / f16 MFMA peak


def layer_flops(cfg, h, w):
    """Algorithmic FLOP per tile of every fused kernel (2 FLOP/MAC; GDN = its CxC contraction)."""
    L, k = cfg['compression_level'], cfg['kernel_size']
    c_org, c_net, c_bn = cfg['channels_org'], cfg['channels_net'], cfg['channels_bn']
    gdn = cfg['act_layer_type'] == 'GDN'
    enc, dec = [], []
    ch, cw, cin = h, w, c_org
    for i in range(L):
        cout = c_net if i < L - 1 else c_bn
        ch, cw = (ch + 1) // 2, (cw + 1) // 2
        macs = ch * cw * cout * cin * k * k
        if gdn and i < L - 1:
            macs += ch * cw * cout * cout
        enc.append(2 * macs)
        cin = cout
    cin = c_bn
    for i in range(L):
        cout = c_net if i < L - 1 else c_org
        macs = ch * cw * cin * cout * k * k  # every input pixel meets every tap once
        ch, cw = ch * 2, cw * 2
        if gdn and i < L - 1:
            macs += ch * cw * cout * cout
        dec.append(2 * macs)
        cin = cout
    return enc, dec


def oracle_layers(state, part, track):
    sd, out, i = state[part], [], 0
    while f'{track}.{i}.model.0.weight' in sd:
        out.append(dict(weight=sd[f'{track}.{i}.model.0.weight'], bias=sd.get(f'{track}.{i}.model.0.bias'),
                        beta=sd.get(f'{track}.{i}.model.1.beta'), gamma=sd.get(f'{track}.{i}.model.1.gamma')))
        i += 1
    return out


def cpu_baseline(state, cfg, tiles, budget_s=20.0):
    """Time the oracle's codec round trip, one tile per call (the reference's call pattern,
    _autoencoders.py:544), on this host's cores.  -> (report, per-tile [bytes, sse], payloads, reconstructions, latents)"""
    import struct
    from oracle import c_oracle as C
    from oracle import cae_oracle as O

    # the 1-GPU box shares a 256-thread host: use this job's CPU share (16), not every hardware thread
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(ncpu, 16)))
    eb = O.EntropyBottleneckOracle(cfg['channels_bn'])
    eb.load(state['fact_ent'])
    eb.update()
    enc_l = oracle_layers(state, 'encoder', 'analysis_track')
    dec_l = oracle_layers(state, 'decoder', 'synthesis_track')
    done, t0 = 0, time.perf_counter()
    part = dict(analysis=0.0, entropy_encode=0.0, entropy_decode=0.0, synthesis=0.0)
    L = len(dec_l)
    per_tile, payloads, recs, latents = [], [], [], []
    with torch.no_grad():
        for t in tiles:  # O.codec_encode / O.codec_decode, spelled out to time their parts
            h, w, _ = t.shape
            a = time.perf_counter()
            y, _ = O.analysis_forward(O.tile_to_input(t), enc_l)
            b = time.perf_counter()
            buf = struct.pack('>QQ', h, w) + eb.compress(y, C.rans_encode_with_indexes)[0]
            c = time.perf_counter()
            yq = eb.decompress([buf[16:]], (h // 2 ** L, w // 2 ** L), C.rans_decode_with_indexes)
            d = time.perf_counter()
            x_r, _ = O.synthesis_forward(yq, dec_l)
            rec = O.output_to_tile(x_r[0])
            e = time.perf_counter()
            for k, v in zip(part, (b - a, c - b, d - c, e - d)):
                part[k] += v
            done += 1
            per_tile.append((len(buf), float(((rec.astype(np.float64) - t) ** 2).sum())))
            payloads.append(buf[16:])
            recs.append(rec)
            latents.append(y)
            if time.perf_counter() - t0 > budget_s:
                break
    dt = time.perf_counter() - t0
    report = dict(value=done / dt, unit='tiles/s', cores=torch.get_num_threads(), kind='port',
                  sample=f'{done} tiles of {tiles[0].shape[0]}x{tiles[0].shape[1]}x{tiles[0].shape[2]}, '
                         f'encode+decode one tile per call, {dt:.1f} s',
                  ms_per_tile={k: 1e3 * v / done for k, v in part.items()})
    return report, per_tile, payloads, recs, latents


def flip_report(enc, eb, tiles_dev, y_cpu):
    """The float -> integer cliff, characterised: symbols round(y - median) of the GPU latents against those of the
    CPU oracle's own latents on the same tiles.  A flip is float noise if BOTH latents sit next to the rounding
    boundary between the two symbols: `max_boundary_distance` = the largest such distance over all flips."""
    y = enc.forward_u8(tiles_dev).cpu()
    y_cpu = torch.cat(y_cpu)
    m = eb._get_medians().detach().cpu().reshape(1, -1, 1, 1)
    s_gpu, s_cpu = torch.round(y - m), torch.round(y_cpu - m)
    flips = s_gpu != s_cpu
    bound = torch.minimum(s_gpu, s_cpu) + 0.5
    dist = torch.maximum((y - m - bound).abs(), (y_cpu - m - bound).abs())[flips]
    return dict(symbols_flipped=int(flips.sum()), symbols_total=int(flips.numel()),
                flipped_per_tile=[int(v) for v in flips.flatten(1).sum(1)],
                max_symbol_delta=int((s_gpu - s_cpu).abs().max()),
                max_boundary_distance=float(dist.max()) if dist.numel() else 0.0,
                max_latent_abs_diff=float((y - y_cpu).abs().max()))


def batch_variants(tiles_dev, n):
    """n distinct batches from one batch of distinct tiles (flips / transposes on the device): every step codes
    different data, so the host coder sees fresh symbols instead of one batch replayed."""
    ops = [lambda t: t, lambda t: t.flip(1), lambda t: t.flip(2), lambda t: t.transpose(1, 2),
           lambda t: t.flip(1).flip(2), lambda t: t.transpose(1, 2).flip(1), lambda t: t.transpose(1, 2).flip(2),
           lambda t: t.transpose(1, 2).flip(1).flip(2)]
    return [ops[i % len(ops)](tiles_dev).contiguous() for i in range(max(1, min(n, len(ops))))]


def kernel_table(cfg, H, B, enc_ms, enc_calls, dec_ms, dec_calls, precision):
    """Per fused kernel: algorithmic FLOP per launch, HIP-event ms per launch, issued MFMA rate and its fraction
    of the peak of the instruction issued."""
    enc_fl, dec_fl = layer_flops(cfg, H, H)
    issue, peak = (3.0, F16_MFMA_PEAK_TFLOPS) if precision == 'f16x3' else (1.0, FP32_MFMA_PEAK_TFLOPS)
    rows = []
    for i, f in enumerate(enc_fl):
        rows.append([f'analysis.{i} conv{"+GDN" if i < len(enc_fl) - 1 else ""}', f * B, enc_ms[1 + i] / max(enc_calls, 1)])
    for i, f in enumerate(dec_fl):
        rows.append([f'synthesis.{i} deconv{"+IGDN" if i < len(dec_fl) - 1 else ""}', f * B, dec_ms[1 + i] / max(dec_calls, 1)])
    out = []
    for name, flop, ms in rows:
        alg = flop / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        out.append(dict(name=name, ms=ms, flop_per_launch=flop, algorithmic_tflops=alg, issued_tflops=issue * alg,
                        frac_of_issued_peak=issue * alg / peak))
    a_ms = sum(enc_ms[1:]) / max(enc_calls, 1)
    a_alg = sum(enc_fl) * B / (a_ms * 1e-3) / 1e12 if a_ms > 0 else 0.0
    stack = dict(ms=a_ms, algorithmic_tflops=a_alg, issued_tflops=issue * a_alg, frac_of_issued_peak=issue * a_alg / peak,
                 algorithmic_vs_fp32_mfma_peak=a_alg / FP32_MFMA_PEAK_TFLOPS)
    return out, stack, peak


HOST_USE = {}  # host CPU use of the last timed run: average busy CPUs of this process, cgroup throttling


def _throttled():
    try:
        kv = dict(l.split() for l in open('/sys/fs/cgroup/cpu.stat'))
        return int(kv['nr_throttled']), int(kv['throttled_usec'])
    except Exception:
        return None


def timed_run(coder, batches, steps, world, dist, cdev):
    """K steps, software-pipelined, bracketed by barrier + synchronise; max over ranks.  -> (seconds, gathered stats)"""
    from cnn_autoencoder_amd import slide

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    seq = [batches[k % len(batches)] for k in range(steps)]
    fence()
    cpu0, thr0 = time.process_time(), _throttled()
    t0 = time.perf_counter()
    local_stats, _ = coder.run(seq)
    t_run = time.perf_counter() - t0
    local_stats = local_stats.to(cdev)
    g0 = time.perf_counter()
    all_stats = slide.gather_stats(local_stats)  # the one collective of the path (RCCL all_gather)
    torch.cuda.synchronize()
    t_gather = time.perf_counter() - g0
    fence()
    dt = time.perf_counter() - t0
    thr1 = _throttled()
    HOST_USE.update(cpus_busy=(time.process_time() - cpu0) / dt,
                    cgroup_throttled_periods=None if thr0 is None else thr1[0] - thr0[0],
                    cgroup_throttled_ms=None if thr0 is None else (thr1[1] - thr0[1]) / 1e3,
                    all_gather_ms=1e3 * t_gather)
    if world > 1:
        # every rank's own clock, host use and collective time, so that a scaling curve explains itself
        mine = torch.tensor([dt, t_run, t_gather, HOST_USE['cpus_busy'], HOST_USE['cgroup_throttled_ms'] or 0.0,
                             float(len(os.sched_getaffinity(0)))], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        HOST_USE['per_rank'] = [dict(rank=r, seconds=float(v[0]), run_seconds=float(v[1]), all_gather_ms=1e3 * float(v[2]),
                                     cpus_busy=float(v[3]), cgroup_throttled_ms=float(v[4]), cpus_allowed=int(v[5]))
                                for r, v in enumerate(every)]
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, all_stats


def profiled_run(coder, batches, steps, warmup, world, dist, cdev):
    for k in range(warmup):
        coder.run([batches[k % len(batches)]])
    coder.enc.set_profiling(True)
    coder.dec.set_profiling(True)
    coder.enc.get_profile(reset=True)
    coder.dec.get_profile(reset=True)
    dt, stats = timed_run(coder, batches, steps, world, dist, cdev)
    enc_ms, enc_calls = coder.enc.get_profile()
    dec_ms, dec_calls = coder.dec.get_profile()
    coder.enc.set_profiling(False)
    coder.dec.set_profiling(False)
    return dt, stats, (enc_ms, enc_calls, dec_ms, dec_calls)


def make_coder(cae, slide, state, precision):
    codec = cae.ConvolutionalAutoencoder(checkpoint=state)
    for k in ('encoder', 'decoder'):
        codec._model[k].module.precision = precision
    return slide.SlideCoder(codec)


def sub_run(cae, slide, cfg, state, precision, H, tiles_dev, steps, warmup, dist, fit=False):
    """A short N=1 run of another configuration, reported next to the headline: tiles/s, ms/step, dominant kernel."""
    coder = make_coder(cae, slide, state, precision)
    if fit:
        coder.eb.fit_quantiles()
        coder.eb.update(force=True)
    B = tiles_dev.shape[0]
    batches = batch_variants(tiles_dev, 4)
    dt, stats, prof = profiled_run(coder, batches, steps, warmup, 1, dist, tiles_dev.device)
    kernels, stack, peak = kernel_table(cfg, H, B, *prof, precision)
    dom = max(kernels, key=lambda k: k['ms'])
    summ = slide.slide_summary(stats, H * H)
    fallbacks = coder.enc.fp32_fallbacks + coder.dec.fp32_fallbacks
    host_ms = {k: 1e3 * v / steps for k, v in coder.timers.items()}
    del coder
    torch.cuda.empty_cache()
    return dict(precision=precision, tile=H, tiles_per_step=B, steps=steps, tiles_per_s=steps * B / dt,
                host_ms_per_step=host_ms, host_cpus_busy=HOST_USE.get('cpus_busy'),
                ms_per_step=1e3 * dt / steps, bpp=summ['bpp'], psnr_db=summ['psnr'],
                dominant_kernel=dict(name=dom['name'], ms=dom['ms'], issued_tflops=dom['issued_tflops'],
                                     peak=peak, frac=dom['frac_of_issued_peak'],
                                     algorithmic_tflops=dom['algorithmic_tflops']),
                analysis_conv_stack=stack, kernel_ms={k['name']: k['ms'] for k in kernels},
                fp32_fallbacks=fallbacks)


def train_flops(cfg, patch):
    """Algorithmic FLOP of one training sample (config 5, train_cae_ms.py:209-230), split by the arithmetic it runs in:
    convolutions (bf16 MFMA): forward + weight gradient of every layer + data gradient of every layer but the first
    analysis layer (the image needs no gradient); GDN / IGDN contractions (fp32 MFMA): forward + the two backward
    contractions (Gamma^T g_n and g_Gamma = g_n (x) z^2; recomputing the norm in the backward is an implementation
    choice and is NOT counted)."""
    plain = dict(cfg, act_layer_type=None)
    enc_c, dec_c = layer_flops(plain, patch, patch)
    enc_a, dec_a = layer_flops(cfg, patch, patch)
    conv = 3.0 * (sum(enc_c) + sum(dec_c)) - enc_c[0]
    gdn = 3.0 * ((sum(enc_a) - sum(enc_c)) + (sum(dec_a) - sum(dec_c)))
    return conv, gdn


def train_run(cae, cfg, state, batch, patch, steps, warmup):
    """`steps` iterations of train.train_step (forward_func -> GeneralLoss -> backward -> clip -> per-module Adam) on one
    synthetic batch resident in HBM.  -> report with the fraction of the MIXED roofline: bf16 convolution FLOP / the dense
    bf16 MFMA peak + fp32 GDN FLOP / the fp32 MFMA peak, over the measured step time."""
    from cnn_autoencoder_amd import criteria, train
    model = cae.autoencoder_from_state_dict(state, train=True)
    opts = train.setup_optim(model)
    criterion = criteria.GeneralLoss(distortion_lambda=0.01)
    x = torch.rand(batch, cfg['channels_org'], patch, patch, device='cuda', generator=torch.Generator('cuda').manual_seed(1))
    for _ in range(warmup):
        train.train_step(x, model, criterion, opts)
    torch.cuda.synchronize()
    # two timed repetitions of `steps` iterations, the faster one reported (both listed): one repetition of the batch-256 run was
    # once 3.3 x slower on the device clock itself (82 vs 25 ms per step, not reproducible in five later runs)
    reps = []
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            ld = train.train_step(x, model, criterion, opts)
        e1.record()
        torch.cuda.synchronize()
        reps.append(((time.perf_counter() - t0) / steps, e0.elapsed_time(e1) / steps))
    dt, dev_ms = min(reps)
    conv, gdn = train_flops(cfg, patch)
    ideal = batch * (conv / (F16_MFMA_PEAK_TFLOPS * 1e12) + gdn / (FP32_MFMA_PEAK_TFLOPS * 1e12))
    del model, opts
    torch.cuda.empty_cache()
    return dict(batch=batch, patch=patch, steps=steps, ms_per_step=1e3 * dt, samples_per_s=batch / dt,
                device_ms_per_step=dev_ms, ms_per_step_of_each_repetition=[1e3 * r[0] for r in reps],
                conv_gflop_per_step=batch * conv / 1e9, gdn_gflop_per_step=batch * gdn / 1e9,
                mixed_roofline_ms=1e3 * ideal, frac=ideal / dt, loss=float(ld['loss']))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=96)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--tile', type=int, default=1024)
    ap.add_argument('--batch', type=int, default=32, help='tiles per step per GPU')
    ap.add_argument('--distinct', type=int, default=0,
                    help='distinct synthetic tiles per rank (0 = one per batch slot); 4 flip/transpose variants of the '
                         'batch are cycled over the steps')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-sub-runs', action='store_true', help='skip the fp32_path / tile256 side runs')
    ap.add_argument('--precision', choices=['f16x3', 'fp32'], default='f16x3',
                    help='conv/GDN arithmetic: f16x3 = operands split into two f16 halves, 3 f16 MFMAs per product, '
                         'fp32 accumulate (fp32-class accuracy); fp32 = exact v_mfma_f32_32x32x2_f32')
    ap.add_argument('--cpus', type=int, default=0,
                    help='restrict this process to its first N allowed CPUs before anything starts (the host budget of a rank '
                         'on a node with N cores per GPU)')
    ap.add_argument('--rehearse-on-one-gpu', action='store_true',
                    help='N>1 rehearsal on a one-GPU box: every rank uses cuda:0 and the collective runs on gloo '
                         '(RCCL refuses two ranks on one device); numbers are meaningless, the code path is the real one')
    args = ap.parse_args()
    if args.cpus > 0:
        os.sched_setaffinity(0, sorted(os.sched_getaffinity(0))[:args.cpus])

    import torch.distributed as dist
    import cnn_autoencoder_amd as cae
    from cnn_autoencoder_amd import slide, synth

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.rehearse_on_one_gpu:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    cdev = torch.device('cpu') if args.rehearse_on_one_gpu else dev  # gloo rehearsal: host tensors

    cfg = dict(synth.CANONICAL)
    state = synth.synthetic_state(cfg, seed=0)
    coder = make_coder(cae, slide, state, args.precision)
    eb = coder.eb
    eb.fit_quantiles()  # quantiles at the aux-loss fixed point, as after training
    eb.update(force=True)
    state['fact_ent'] = {k: v.detach().cpu() for k, v in eb.state_dict().items()}

    H = args.tile
    B = args.batch
    n_distinct = B if args.distinct <= 0 else min(args.distinct, B)
    first = rank * B  # this rank's own tiles
    base = synth.histo_tiles(n_distinct, H, first_index=first)
    reps = (B + n_distinct - 1) // n_distinct
    tiles_host = np.concatenate([base] * reps)[:B]
    tiles_dev = torch.from_numpy(tiles_host).to(dev)
    batches = batch_variants(tiles_dev, 4)

    dt, all_stats, prof = profiled_run(coder, batches, args.steps, args.warmup, world, dist, cdev)

    if rank == 0:
        total_tiles = world * args.steps * B
        summ = slide.slide_summary(all_stats, H * H)
        kernels, stack, peak = kernel_table(cfg, H, B, *prof, args.precision)
        dom = max(kernels, key=lambda k: k['ms'])
        # HBM bytes per launch of the dominant kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same
        # command, committed under profiles/ (FETCH_SIZE doubled per the gfx950 correction); an OFFLINE value tied
        # to the commit named in that file, only valid for the profiled shape (batch 32, 1024x1024 tiles)
        traffic, traffic_source = None, None
        keys = {'fp32': {'analysis.1': 'conv_s2_kernel<3, 4, 4, true, 2, false>@4194304',
                         'synthesis.2': 'deconv_s2_kernel<3, 4, 4, true>@4194304'},
                'f16x3': {'analysis.1': 'conv_s2_f16_kernel<3, 4, true>@2097152',
                          'synthesis.2': 'deconv_s2_f16_kernel<3, 4, 8, 1, true>@4194304'}}[args.precision]
        key = keys.get(dom['name'].split(' ')[0])
        for name in ('r03_hbm_traffic.json', 'r02_hbm_traffic.json', 'r01_hbm_traffic.json'):
            try:
                tj = json.load(open(os.path.join(ROOT, 'profiles', name)))
                if key and B == 32 and H == 1024 and key in tj[args.precision]:
                    traffic = tj[args.precision][key]['hbm_mb'] * 1e6
                    traffic_source = (f'profiles/{name}: offline rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this '
                                      f'command (build: {tj.get("commit", "see the note in that file")}); {key}')
                    break
            except Exception:
                continue
        gpu_ms = (sum(prof[0]) / max(prof[1], 1), sum(prof[2]) / max(prof[3], 1))
        f16 = args.precision == 'f16x3'
        from cnn_autoencoder_amd import _lib
        lock = int(_lib.lib().cae_coder_lockstep())
        threads = dict(encode=int(_lib.lib().cae_coder_threads(coder.encode_threads, (B + lock - 1) // lock)),
                       decode=int(_lib.lib().cae_coder_threads(coder.decode_threads, (B + lock - 1) // lock)),
                       cpu_budget=int(_lib.lib().cae_cpu_budget()), lockstep=lock,
                       cpus_allowed=len(os.sched_getaffinity(0)))
        line = {
            'metric': 'tiles/sec, compress+decompress round trip of 1024x1024x3 histology tiles',
            'value': total_tiles / dt,
            'unit': 'tiles/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32 (f16x3 split MFMA: 3 f16 MFMAs per product, fp32 accumulate)' if f16 else 'f32',
            'data': f'synthetic: seeded procedural H&E-like tiles ({n_distinct} distinct per rank; 4 flip/transpose '
                    'variants of the batch cycled over the steps), random-init canonical weights (seed 0), quantiles at '
                    'the aux-loss fixed point',
            'config': {'workload': f'{H}x{H}x3 histology tiles, canonical 128/192/L4/k3 GDN model, '
                                   f'{B} tiles per step per GPU, encode+decode', 'tiles_per_step_per_gpu': B,
                       'tile': H, 'sharding': f'contiguous tile blocks over {world} rank(s), 1 all_gather of stats'},
            'slide_stats': {'bpp': summ['bpp'], 'psnr_db': summ['psnr'], 'tiles': summ['tiles']},
            'roofline': {'bound': 'mfma', 'kernel': dom['name'], 'achieved': dom['issued_tflops'], 'peak': peak,
                         'unit': 'TFLOP/s', 'frac': dom['frac_of_issued_peak'], 'traffic': traffic,
                         'traffic_source': traffic_source, 'ms_per_launch': dom['ms'],
                         'flop_per_launch': dom['flop_per_launch'],
                         'algorithmic_tflops': dom['algorithmic_tflops'],
                         'algorithmic_vs_fp32_mfma_peak': dom['algorithmic_tflops'] / FP32_MFMA_PEAK_TFLOPS,
                         'note': ('achieved = MFMA FLOP/s issued = 3 x algorithmic (three v_mfma_f32_32x32x16_f16 per '
                                  'product) against the dense f16 MFMA peak' if f16 else
                                  'exact fp32 MFMA (v_mfma_f32_32x32x2_f32) against its dense peak')},
            'analysis_conv_stack': stack,
            'kernels': kernels,
            'gpu_ms_per_step': {'analysis': gpu_ms[0], 'synthesis': gpu_ms[1]},
            'host_ms_per_step': {k: 1e3 * v / args.steps for k, v in coder.timers.items()},
            'host_coder_threads': threads,
            'host_use': dict(HOST_USE),
            'fp32_fallbacks': coder.enc.fp32_fallbacks + coder.dec.fp32_fallbacks,
        }
        if threads:
            # wall time of a pool x its threads: an upper bound of the CPU time (threads wait for one another's tail)
            line['host_cpu_ms_per_step'] = {k: 1e3 * coder.timers[k] / args.steps * threads[k.split('_')[1]]
                                            for k in ('host_encode', 'host_decode')}
        line['cpu_baseline'] = None
        if world == 1 and not args.no_cpu_baseline:
            report, per_tile, cpu_payloads, cpu_recs, cpu_latents = cpu_baseline(state, cfg, list(tiles_host[:16]))
            line['cpu_baseline'] = report
            # the GPU path on exactly the tiles the oracle coded (BASELINE's "PSNR/bpp parity" at full tile size)
            n = len(per_tile)
            payloads, rec, stats = coder.roundtrip(tiles_dev[:n].contiguous())
            rec = rec.cpu().numpy()
            px = H * H
            mse = lambda sse: sse / (px * 3)
            psnr = lambda sse: 10 * np.log10(255.0 ** 2 / mse(sse)) if sse > 0 else float('inf')
            diff = np.abs(rec.astype(np.int16) - np.stack(cpu_recs).astype(np.int16))
            line['parity_vs_cpu'] = {
                'tiles': n,
                'bpp_gpu': 8.0 * float(stats[:, 0].sum()) / (n * px),
                'bpp_cpu': 8.0 * sum(p[0] for p in per_tile) / (n * px),
                'psnr_gpu': psnr(float(stats[:, 1].sum()) / n), 'psnr_cpu': psnr(sum(p[1] for p in per_tile) / n),
                'bitstreams_identical': int(sum(a == b for a, b in zip(payloads, cpu_payloads))),
                'max_abs_delta': int(diff.max()), 'pixels_differing_frac': float((diff > 0).mean()),
                'note': 'symbols_flipped: symbols of the GPU latents that differ from those of the CPU oracle\'s own '
                        'latents (the integer step is bit-exact given identical latents); max_boundary_distance: how far '
                        'the two latents of a flipped symbol sit from the rounding boundary between them, at most; '
                        'max_abs_delta in uint8 levels between the two reconstructions (where a flipped symbol lands)',
            }
            ref32 = make_coder(cae, slide, state, 'fp32' if f16 else 'f16x3')
            for name, cd in ((args.precision, coder), ('fp32' if f16 else 'f16x3', ref32)):
                line['parity_vs_cpu'][name] = flip_report(cd.enc, cd.eb, tiles_dev[:n].contiguous(), cpu_latents)
            del ref32
        if world == 1 and not args.no_sub_runs:
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            import dropin_bench
            d = dropin_bench.measure(coder.codec, tiles_host, (1, 8, 16), budget_s=1.5)
            coder.codec.close()
            line['dropin'] = {
                'pattern': 'T threads x codec.encode(chunk) / codec.decode(buf), one 1024x1024x3 chunk per call, one shared '
                           'codec; host arrays in and out (PCIe inside the number)',
                'encode_tiles_per_s': d['encode'], 'decode_tiles_per_s': d['decode'],
                'pipelined_round_trip_tiles_per_s': total_tiles / dt,
                'cpus_busy': d['cpus_busy'], 'mean_gpu_batch': d.get('mean_batch'), 'ms_per_call': d.get('ms_per_call'),
                'payloads_identical_to_encode_batch': d['identical'],
            }
            other = 'fp32' if f16 else 'f16x3'
            del coder
            torch.cuda.empty_cache()
            line[f'{other}_path'] = sub_run(cae, slide, cfg, state, other, H, tiles_dev, 16, 2, dist)
            t256 = torch.from_numpy(synth.histo_tiles(64, 256, first_index=10_000)).to(dev)
            t256 = torch.cat(batch_variants(t256, 8))  # 512 tiles per step
            line['tile256'] = sub_run(cae, slide, cfg, state, args.precision, 256, t256, 16, 2, dist)
            del t256
            torch.cuda.empty_cache()
            # the other end of what the host range coder sees: a low-rate state (most symbols zero under a narrow prior,
            # standing in for a trained model) on the same tiles
            low = synth.synthetic_state(cfg, seed=0, **synth.LOW_RATE)
            line['low_rate_state'] = sub_run(cae, slide, cfg, low, args.precision, H, tiles_dev, 16, 2, dist, fit=True)
            # the same round trip with 8 CPUs for this rank (what a node with 8 cores per GPU gives): a child process with its
            # affinity cut to 8 CPUs; the coder pools and the lockstep width follow cae_cpu_budget()
            try:
                import subprocess
                r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpus', '8', '--steps', '32', '--warmup', '2',
                                    '--no-sub-runs', '--no-cpu-baseline', '--precision', args.precision],
                                   capture_output=True, text=True, timeout=300)
                sub = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
                line['host_8cpu'] = dict(tiles_per_s=sub['value'], ms_per_step=sub['ms_per_step'], host_use=sub['host_use'],
                                         host_coder_threads=sub['host_coder_threads'], host_ms_per_step=sub['host_ms_per_step'],
                                         vs_unrestricted=sub['value'] / line['value'])
            except Exception as e:  # noqa: BLE001 - a side measurement must not cost the line
                line['host_8cpu'] = dict(error=repr(e)[:200])
            # BASELINE config 5 (train_cae_ms.py rate-distortion loop): canonical model, 256x256 patches
            line['train'] = {
                'dtype': 'bf16 convolutions (fp32 accumulate), fp32 GDN / IGDN (v_mfma_f32_32x32x2_f32), fp32 optimiser',
                'roofline': 'mixed: conv FLOP / 2500 TFLOP/s (dense bf16 MFMA) + GDN FLOP / 157.3 TFLOP/s (fp32 MFMA); '
                            'frac = that time / measured step time',
                'batch16': train_run(cae, cfg, state, 16, 256, 20, 3),
                'batch128': train_run(cae, cfg, state, 128, 256, 8, 2),
                'batch256': train_run(cae, cfg, state, 256, 256, 5, 2),
            }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
