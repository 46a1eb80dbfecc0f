#!/usr/bin/env python3
"""Headline bench: whole-slide tile batches through the HIP compress -> decompress hot path.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A step = one pass of the hot path over one batch of synthetic 1024x1024x3 histology tiles that is
already resident in HBM: analysis conv stack + GDN -> quantise -> rANS encode (chunk bitstreams in
host memory) -> rANS decode -> dequantise -> synthesis deconv stack + IGDN -> uint8 tiles in HBM,
plus the per-tile rate/distortion record.  Tiles shard over ranks with no data-path collective
(weak scaling: the per-GPU batch is fixed); one all_gather of the per-tile statistics closes the
timed region.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline      dominant fused kernel (conv/deconv + GDN), algorithmic FLOP / HIP-event time, vs the
                dense fp32 MFMA peak (the path computes in exact fp32: v_mfma_f32_32x32x2_f32)
  cpu_baseline  the CPU oracle (torch-CPU conv + restated GDN + C rANS) on a bounded sample of the
                same workload on this host's cores -- a reported baseline, not the target
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


def cpu_baseline(state, cfg, tiles, budget_s=20.0):
    """Time the oracle's codec round trip, one tile per call (the reference's call pattern,
    _autoencoders.py:544), on this host's cores."""
    from oracle import c_oracle as C
    from oracle import cae_oracle as O

    def layers(part, track):
        sd, out, i = state[part], [], 0
        while f'{track}.{i}.model.0.weight' in sd:
            out.append(dict(weight=sd[f'{track}.{i}.model.0.weight'], bias=sd.get(f'{track}.{i}.model.0.bias'),
                            beta=sd.get(f'{track}.{i}.model.1.beta'), gamma=sd.get(f'{track}.{i}.model.1.gamma')))
            i += 1
        return out

    # the 1-GPU box shares a 256-thread host: use this job's CPU share (16), not every hardware thread
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(ncpu, 16)))
    eb = O.EntropyBottleneckOracle(cfg['channels_bn'])
    eb.load(state['fact_ent'])
    eb.update()
    enc_l, dec_l = layers('encoder', 'analysis_track'), layers('decoder', 'synthesis_track')
    import struct
    done, t0 = 0, time.perf_counter()
    part = dict(analysis=0.0, entropy_encode=0.0, entropy_decode=0.0, synthesis=0.0)
    L = len(dec_l)
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
            O.output_to_tile(x_r[0])
            e = time.perf_counter()
            for k, v in zip(part, (b - a, c - b, d - c, e - d)):
                part[k] += v
            done += 1
            if time.perf_counter() - t0 > budget_s:
                break
    dt = time.perf_counter() - t0
    return dict(value=done / dt, unit='tiles/s', cores=torch.get_num_threads(), kind='port',
                sample=f'{done} tiles of {tiles[0].shape[0]}x{tiles[0].shape[1]}x{tiles[0].shape[2]}, '
                       f'encode+decode one tile per call, {dt:.1f} s',
                ms_per_tile={k: 1e3 * v / done for k, v in part.items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=96)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--tile', type=int, default=1024)
    ap.add_argument('--batch', type=int, default=32, help='tiles per step per GPU')
    ap.add_argument('--distinct', type=int, default=4, help='distinct synthetic tiles per rank (tiled to the batch)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--precision', choices=['f16x3', 'fp32'], default='f16x3',
                    help='conv/GDN arithmetic: f16x3 = operands split into two f16 halves, 3 f16 MFMAs per product, '
                         'fp32 accumulate (fp32-class accuracy); fp32 = exact v_mfma_f32_32x32x2_f32')
    ap.add_argument('--rehearse-on-one-gpu', action='store_true',
                    help='N>1 rehearsal on a one-GPU box: every rank uses cuda:0 and the collective runs on gloo '
                         '(RCCL refuses two ranks on one device); numbers are meaningless, the code path is the real one')
    args = ap.parse_args()

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

    os.environ['CAE_PRECISION'] = args.precision
    cfg = dict(synth.CANONICAL)
    state = synth.synthetic_state(cfg, seed=0)
    codec = cae.ConvolutionalAutoencoder(checkpoint=state)
    eb = codec._model['fact_ent'].module
    eb.fit_quantiles()  # quantiles at the aux-loss fixed point, as after training
    eb.update(force=True)
    state['fact_ent'] = {k: v.detach().cpu() for k, v in eb.state_dict().items()}
    coder = slide.SlideCoder(codec)

    H = args.tile
    B = args.batch
    n_distinct = min(args.distinct, B)
    first = rank * args.steps * B  # this rank's block of the slide, in chunk raster order
    base = synth.histo_tiles(n_distinct, H, first_index=first)
    reps = (B + n_distinct - 1) // n_distinct
    tiles_host = np.concatenate([base] * reps)[:B]
    tiles_dev = torch.from_numpy(tiles_host).to(dev)

    for _ in range(args.warmup):
        coder.run([tiles_dev])
    coder.enc.set_profiling(True)
    coder.dec.set_profiling(True)
    coder.enc.get_profile(reset=True)
    coder.dec.get_profile(reset=True)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    # K steps, software-pipelined: the host range-codes batch k while the GPU runs batch k+1 / k-1
    local_stats, _ = coder.run([tiles_dev] * args.steps)
    cdev = torch.device('cpu') if args.rehearse_on_one_gpu else dev  # gloo rehearsal: host tensors
    local_stats = local_stats.to(cdev)
    all_stats = slide.gather_stats(local_stats)  # the one collective of the path (RCCL all_gather)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    enc_ms, enc_calls = coder.enc.get_profile()
    dec_ms, dec_calls = coder.dec.get_profile()
    coder.enc.set_profiling(False)
    coder.dec.set_profiling(False)

    if rank == 0:
        total_tiles = world * args.steps * B
        summ = slide.slide_summary(all_stats, H * H)
        enc_fl, dec_fl = layer_flops(cfg, H, H)
        kernels = []
        for i, f in enumerate(enc_fl):
            kernels.append((f'analysis.{i} conv{"+GDN" if i < len(enc_fl) - 1 else ""}', f * B, enc_ms[1 + i] / max(enc_calls, 1)))
        for i, f in enumerate(dec_fl):
            kernels.append((f'synthesis.{i} deconv{"+IGDN" if i < len(dec_fl) - 1 else ""}', f * B, dec_ms[1 + i] / max(dec_calls, 1)))
        dom = max(kernels, key=lambda k: k[2])
        achieved = dom[1] / (dom[2] * 1e-3) / 1e12
        # HBM bytes per launch of the dominant kernel: measured offline by rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE passes of this same command (profiles/r01_hbm_traffic.json, FETCH_SIZE doubled per
        # the gfx950 correction); only valid for the profiled shape (batch 32, 1024x1024 tiles)
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, 'profiles', 'r01_hbm_traffic.json')))[args.precision]
            keys = {'fp32': {'analysis.1': 'conv_s2_kernel<3, 4, 4, true, 2, false>@4194304',
                             'synthesis.2': 'deconv_s2_kernel<3, 4, 4, true>@4194304'},
                    'f16x3': {'analysis.1': 'conv_s2_f16_kernel<3, 4, true>@2097152',
                              'synthesis.2': 'deconv_s2_f16_kernel<3, 4, 8, 1, true>@4194304'}}[args.precision]
            key = keys.get(dom[0].split(' ')[0])
            if key and B == 32 and H == 1024:
                traffic = tj[key]['hbm_mb'] * 1e6
        except Exception:
            traffic = None
        gpu_ms = (sum(enc_ms) / max(enc_calls, 1), sum(dec_ms) / max(dec_calls, 1))
        f16 = args.precision == 'f16x3'
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
            'dtype': 'f32 (f16x3 split MFMA: 3 f16 MFMAs per product, fp32 accumulate)' if args.precision == 'f16x3' else 'f32',
            'data': f'synthetic: seeded procedural H&E-like tiles ({n_distinct} distinct per rank tiled to the batch), '
                    'random-init canonical weights (seed 0), quantiles at the aux-loss fixed point',
            'config': {'workload': f'{H}x{H}x3 histology tiles, canonical 128/192/L4/k3 GDN model, '
                                   f'{B} tiles per step per GPU, encode+decode', 'tiles_per_step_per_gpu': B,
                       'tile': H, 'sharding': f'contiguous tile blocks over {world} rank(s), 1 all_gather of stats'},
            'parity': {'bpp': summ['bpp'], 'psnr_db': summ['psnr'], 'tiles': summ['tiles']},
            'roofline': {'bound': 'mfma', 'kernel': dom[0], 'achieved': achieved, 'peak': FP32_MFMA_PEAK_TFLOPS,
                         'unit': 'TFLOP/s', 'frac': achieved / FP32_MFMA_PEAK_TFLOPS, 'traffic': traffic,
                         'ms_per_launch': dom[2], 'flop_per_launch': dom[1],
                         'note': ('algorithmic fp32 FLOP/s against the dense fp32 MFMA peak (SURVEY 8d denominator for '
                                  'split-operand paths); the f16x3 kernels issue 3 f16 MFMA FLOP per algorithmic FLOP: '
                                  'f16 MFMA rate = %.0f TFLOP/s = %.3f of the 2500 TFLOP/s dense f16 peak'
                                  % (3 * achieved, 3 * achieved / 2500.0)) if f16 else
                                 'exact fp32 MFMA (v_mfma_f32_32x32x2_f32) against its dense peak'},
            'kernels': [{'name': k[0], 'ms': k[2], 'tflops': k[1] / (k[2] * 1e-3) / 1e12 if k[2] > 0 else None}
                        for k in kernels],
            'gpu_ms_per_step': {'analysis': gpu_ms[0], 'synthesis': gpu_ms[1]},
            'host_ms_per_step': {k: 1e3 * v / args.steps for k, v in coder.timers.items()},
            'analysis_conv_stack_frac_of_fp32_mfma_peak':
                (sum(enc_fl) * B / (sum(enc_ms[1:]) / max(enc_calls, 1) * 1e-3) / 1e12) / FP32_MFMA_PEAK_TFLOPS,
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(state, cfg, list(tiles_host[:16]))
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
