"""Training path (BASELINE config 5) on the GPU: the bf16 convolution kernels, their data and weight
gradients, the fp32 GDN gradient, whole-track gradients and a 20-step loss curve against the CPU restatement
(oracle/train_oracle.py: torch-CPU autograd of the reference-pinned conv stacks + compressai's gradient rules).

Tolerances (stated per north_star: bf16 convolutions, fp32 GDN):
  * against the restatement with the SAME bf16 rounding points: 1e-3 of the tensor's largest magnitude (only the
    summation order differs; observed ~1e-5);
  * fp32 GDN forward / backward alone: 1e-4;
  * against the plain fp32 restatement: 3e-2 (what bf16 operands cost), reported so that the cost stays visible.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def cae(built_lib):
    import cnn_autoencoder_amd as cae
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    return cae


def rel(got: torch.Tensor, want: torch.Tensor) -> float:
    return float((got.double().cpu() - want.double().cpu()).abs().max() / max(float(want.abs().max()), 1e-30))


def bf(x):
    return x.bfloat16().float()


def to_t(x, cp, dtype):
    """NCHW cpu -> T layout on the GPU"""
    n, c, h, w = x.shape
    t = torch.zeros((n, h, w, cp), dtype=torch.float32)
    t[..., :c] = x.permute(0, 2, 3, 1)
    return t.to(dtype).cuda()


def from_t(t, c):
    return t[..., :c].float().cpu().permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize('cin,cout,ks,shape', [(3, 128, 3, (2, 40, 56)), (128, 128, 3, (2, 37, 45)), (48, 192, 5, (1, 33, 50)),
                                               (160, 64, 5, (1, 18, 21)), (32, 32, 3, (3, 16, 16))])
def test_conv_kernels(cae, cin, cout, ks, shape):
    """conv forward, its data gradient (extended domain + fold) and weight gradient against F.conv2d autograd on the
    same bf16-rounded operands."""
    from cnn_autoencoder_amd import _lib, train
    L = _lib.lib()
    torch.manual_seed(cin + cout)
    n, h, w = shape
    x = bf(torch.randn(n, cin, h, w)).requires_grad_(True)
    wt = bf(torch.randn(cout, cin, ks, ks) / (cin * ks * ks) ** 0.5).requires_grad_(True)
    b = torch.randn(cout)
    P = ks // 2
    z = F.conv2d(F.pad(x, (P, P, P, P), mode='reflect'), wt, b, stride=2)
    gz = bf(torch.randn_like(z))
    z.backward(gz)
    cip, cop = train._pad32(cin), train._pad32(cout)
    oh, ow = z.shape[2:]
    x16 = to_t(x.detach(), cip, torch.bfloat16)
    wp = train._pack(wt.detach().cuda(), 1, ks)
    bias_p = torch.zeros(cop).cuda()
    bias_p[:cout] = b.cuda()
    z32 = torch.empty((n, oh, ow, cop), device='cuda')
    z16 = torch.empty((n, oh, ow, cop), device='cuda', dtype=torch.bfloat16)
    _lib.check(L.cae_t_conv_forward(x16.data_ptr(), n, h, w, cip, wp.data_ptr(), ks, z32.data_ptr(), z16.data_ptr(), cop,
                                    bias_p.data_ptr(), None))
    assert rel(from_t(z32, cout), z.detach()) < 1e-3
    assert float(z32[..., cout:].abs().max()) == 0.0 if cop > cout else True
    assert rel(from_t(z16, cout), z.detach()) < 1e-2  # (bf16 copy of the output)
    # data gradient
    g16 = to_t(gz, cop, torch.bfloat16)
    wpd = train._pack(wt.detach().cuda(), 0, ks)
    gext = torch.full((n, h + 2 * P, w + 2 * P, cip), float('nan'), device='cuda')
    _lib.check(L.cae_t_conv_dgrad_ext(g16.data_ptr(), n, oh, ow, cop, wpd.data_ptr(), ks, h, w, gext.data_ptr(), cip, None))
    gx16 = torch.empty((n, h, w, cip), device='cuda', dtype=torch.bfloat16)
    _lib.check(L.cae_t_fold_to_bf16(gext.data_ptr(), n, h, w, P, cip, gx16.data_ptr(), None))
    assert rel(from_t(gx16, cin), x.grad) < 1e-2  # (stored as bf16)
    # weight gradient
    gw = torch.empty((ks * ks, cip, cop), device='cuda')
    _lib.check(L.cae_t_wgrad(x16.data_ptr(), n, h, w, cip, g16.data_ptr(), oh, ow, cop, ks, 1, gw.data_ptr(), None))
    assert rel(train._weight_grad(gw, (cout, cin), ks), wt.grad) < 1e-3
    gb = torch.empty(cop, device='cuda')
    _lib.check(L.cae_t_colsum(g16.data_ptr(), n * oh * ow, cop, gb.data_ptr(), None))
    assert rel(gb[:cout], gz.sum(dim=(0, 2, 3))) < 1e-3


@pytest.mark.parametrize('npb', ['', '3'])
def test_conv_kernels_walk_several_samples_per_block(cae, npb, monkeypatch):
    """gg8_kernel / wgrad8_kernel on the canonical 128 -> 128 layer with blocks that walk several samples (the shapes of the
    other tests give every block one sample): 8 samples of 72 x 120, forward, both data-gradient forms and the weight gradient;
    CAE_GG8_NPB=3 forces ragged sample groups (3, 3, 2) in the gather-GEMM, the weight gradient walks 2 samples per block."""
    from cnn_autoencoder_amd import _lib, train
    L = _lib.lib()
    if npb:
        monkeypatch.setenv('CAE_GG8_NPB', npb)
    torch.manual_seed(7)
    n, c, h, w, ks = 8, 128, 72, 120, 3
    x = bf(torch.randn(n, c, h, w)).requires_grad_(True)
    wt = bf(torch.randn(c, c, ks, ks) / (c * ks * ks) ** 0.5).requires_grad_(True)
    z = F.conv2d(F.pad(x, (1, 1, 1, 1), mode='reflect'), wt, None, stride=2)
    gz = bf(torch.randn_like(z))
    z.backward(gz)
    oh, ow = z.shape[2:]
    x16 = to_t(x.detach(), c, torch.bfloat16)
    wp = train._pack(wt.detach().cuda(), 1, ks)
    z32 = torch.empty((n, oh, ow, c), device='cuda')
    _lib.check(L.cae_t_conv_forward(x16.data_ptr(), n, h, w, c, wp.data_ptr(), ks, z32.data_ptr(), None, c, None, None))
    assert rel(from_t(z32, c), z.detach()) < 1e-3
    g16 = to_t(gz, c, torch.bfloat16)
    wpd = train._pack(wt.detach().cuda(), 0, ks)
    gext = torch.full((n, h + 2, w + 2, c), float('nan'), device='cuda')
    _lib.check(L.cae_t_conv_dgrad_ext(g16.data_ptr(), n, oh, ow, c, wpd.data_ptr(), ks, h, w, gext.data_ptr(), c, None))
    gx16 = torch.empty((n, h, w, c), device='cuda', dtype=torch.bfloat16)
    _lib.check(L.cae_t_fold_to_bf16(gext.data_ptr(), n, h, w, 1, c, gx16.data_ptr(), None))
    assert rel(from_t(gx16, c), x.grad) < 1e-2
    gw = torch.empty((ks * ks, c, c), device='cuda')
    _lib.check(L.cae_t_wgrad(x16.data_ptr(), n, h, w, c, g16.data_ptr(), oh, ow, c, ks, 1, gw.data_ptr(), None))
    assert rel(train._weight_grad(gw, (c, c), ks), wt.grad) < 1e-3
    # the transposed forward of the same weights (four parity launches), as ConvTranspose2d
    xt = bf(torch.randn(n, c, oh, ow))
    wtt = bf(torch.randn(c, c, ks, ks) / (c * ks * ks) ** 0.5)
    u = F.conv_transpose2d(xt, wtt, None, stride=2, padding=1, output_padding=1)
    u32 = torch.empty((n, 2 * oh, 2 * ow, c), device='cuda')
    wpt = train._pack(wtt.cuda(), 0, ks)
    _lib.check(L.cae_t_deconv_forward(to_t(xt, c, torch.bfloat16).data_ptr(), n, oh, ow, c, wpt.data_ptr(), ks, u32.data_ptr(),
                                      None, c, None, None))
    assert rel(from_t(u32, c), u) < 1e-3


@pytest.mark.parametrize('cin,cout,ks,shape', [(192, 128, 3, (2, 9, 13)), (128, 128, 3, (1, 20, 17)), (128, 3, 3, (2, 24, 24)),
                                               (48, 96, 5, (1, 11, 19)), (32, 32, 5, (2, 16, 16))])
def test_deconv_kernels(cae, cin, cout, ks, shape):
    from cnn_autoencoder_amd import _lib, train
    L = _lib.lib()
    torch.manual_seed(cin * 3 + cout)
    n, h, w = shape
    x = bf(torch.randn(n, cin, h, w)).requires_grad_(True)
    wt = bf(torch.randn(cin, cout, ks, ks) / (cin * ks * ks / 4) ** 0.5).requires_grad_(True)
    b = torch.randn(cout)
    z = F.conv_transpose2d(x, wt, b, stride=2, padding=ks // 2, output_padding=1)
    gz = bf(torch.randn_like(z))
    z.backward(gz)
    cip, cop = train._pad32(cin), train._pad32(cout)
    x16 = to_t(x.detach(), cip, torch.bfloat16)
    wp = train._pack(wt.detach().cuda(), 0, ks)
    bias_p = torch.zeros(cop).cuda()
    bias_p[:cout] = b.cuda()
    z32 = torch.full((n, 2 * h, 2 * w, cop), float('nan'), device='cuda')
    _lib.check(L.cae_t_deconv_forward(x16.data_ptr(), n, h, w, cip, wp.data_ptr(), ks, z32.data_ptr(), None, cop,
                                      bias_p.data_ptr(), None))
    assert rel(from_t(z32, cout), z.detach()) < 1e-3
    g16 = to_t(gz, cop, torch.bfloat16)
    wpd = train._pack(wt.detach().cuda(), 1, ks)
    gx32 = torch.empty((n, h, w, cip), device='cuda')
    _lib.check(L.cae_t_deconv_dgrad(g16.data_ptr(), n, h, w, cop, wpd.data_ptr(), ks, gx32.data_ptr(), None, cip, None))
    assert rel(from_t(gx32, cin), x.grad) < 1e-3
    gw = torch.empty((ks * ks, cop, cip), device='cuda')
    _lib.check(L.cae_t_wgrad(g16.data_ptr(), n, 2 * h, 2 * w, cop, x16.data_ptr(), h, w, cip, ks, 0, gw.data_ptr(), None))
    assert rel(train._weight_grad(gw, (cin, cout), ks), wt.grad) < 1e-3


@pytest.mark.parametrize('fused', [False, True])
@pytest.mark.parametrize('inverse', [False, True])
@pytest.mark.parametrize('c,shape,pad', [(128, (2, 19, 23), 1), (48, (1, 40, 31), 2), (192, (1, 9, 14), 0), (32, (3, 8, 8), 1),
                                         (96, (2, 33, 17), 2), (128, (4, 64, 64), 0),
                                         # tiny images: the two fold bands of fold_inplace_kernel overlap and row / column 0 folds too
                                         (32, (2, 3, 5), 2), (64, (1, 2, 3), 1), (32, (1, 4, 3), 2)])
def test_gdn_kernels(cae, inverse, c, shape, pad, fused, monkeypatch):
    """fp32 GDN / IGDN forward and backward (with the reflect fold of an extended-domain gradient) against autograd:
    the three-kernel form and the fused pair (forward saving its factor, one-kernel backward; up to 128 channels)."""
    from cnn_autoencoder_amd import _lib, train
    if fused and train._pad32(c) > 128:
        pytest.skip('the fused kernels are built for at most 128 channels')
    monkeypatch.setenv('CAE_GDN_FUSED', '1' if fused else '0')
    L = _lib.lib()
    torch.manual_seed(c + pad)
    n, h, w = shape
    cp = train._pad32(c)
    z = torch.randn(n, c, h, w).requires_grad_(True)
    beta = (torch.rand(c) + 0.5).requires_grad_(True)
    gamma = (0.1 * torch.eye(c) + 0.02 * torch.rand(c, c)).requires_grad_(True)
    norm = F.conv2d(z ** 2, gamma.reshape(c, c, 1, 1), beta)
    y = z * (torch.sqrt(norm) if inverse else torch.rsqrt(norm))
    gext = torch.randn(n, c, h + 2 * pad, w + 2 * pad)
    # fold: gradient of reflect padding (pad > 0) = what F.pad's backward does
    yp = F.pad(y, (pad,) * 4, mode='reflect') if pad else y
    yp.backward(gext)
    beta_p = torch.ones(cp)
    beta_p[:c] = beta.detach()
    gamma_p = torch.zeros(cp, cp)
    gamma_p[:c, :c] = gamma.detach()
    z32 = to_t(z.detach(), cp, torch.float32)
    y32 = torch.empty_like(z32)
    y16 = torch.empty_like(z32, dtype=torch.bfloat16)
    beta_d, gamma_d = beta_p.cuda(), gamma_p.cuda()  # (kept alive: the call takes raw pointers)
    _lib.check(L.cae_t_gdn_forward(z32.data_ptr(), n * h * w, cp, beta_d.data_ptr(), gamma_d.data_ptr(),
                                   int(inverse), y32.data_ptr(), y16.data_ptr(), None))
    torch.cuda.synchronize()
    assert rel(from_t(y32, c), y.detach()) < 1e-4
    assert rel(from_t(y16, c), y.detach()) < 1e-2
    ge = to_t(gext, cp, torch.float32)
    f = None
    if fused:
        y16, f = train._gdn_forward(z32, beta_d, gamma_d, inverse)
        assert f is not None and rel(from_t(y16, c), y.detach()) < 1e-2
    gz16, gb, gg = train._gdn_backward(z32, ge, pad, beta_d, gamma_d, inverse, f)
    assert rel(from_t(gz16, c), z.grad) < 1e-2  # (stored as bf16)
    assert rel(gb[:c], beta.grad) < 1e-4
    assert rel(gg[:c, :c], gamma.grad) < 1e-4


def _models(cae, cfg, seed):
    """product model (train mode) + oracle parameter lists sharing the same values"""
    from cnn_autoencoder_amd import synth
    from conftest import oracle_layers
    state = synth.synthetic_state(cfg, seed=seed)
    model = cae.autoencoder_from_state_dict(state, train=True)
    leaf = lambda t: None if t is None else t.detach().clone().requires_grad_(True)  # noqa: E731
    layers = {part: [{k: leaf(v) for k, v in l.items() if k in ('weight', 'bias', 'beta', 'gamma', 'pre_weight', 'pre_bias')}
                     for l in oracle_layers(state, part)] for part in ('encoder', 'decoder')}
    return state, model, layers


def _param_names(l):
    """(oracle key, state-dict suffix) pairs of a unit: with a pre-convolution the strided layer is model.2"""
    if l.get('pre_weight') is not None:
        return (('pre_weight', 'model.0.weight'), ('pre_bias', 'model.0.bias'), ('weight', 'model.2.weight'), ('bias', 'model.2.bias'))
    return (('weight', 'model.0.weight'), ('bias', 'model.0.bias'), ('beta', 'model.1.beta'), ('gamma', 'model.1.gamma'))


def _named_grads(module, track):
    out = {}
    for name, p in module.named_parameters():
        if p.grad is not None:
            out[name] = p.grad.detach().cpu()
    return out


@pytest.mark.parametrize('cfgkw,shape', [(dict(channels_net=32, channels_bn=48, compression_level=3), (2, 40, 56)),
                                         (dict(channels_net=128, channels_bn=192, compression_level=4), (1, 64, 64)),
                                         (dict(channels_net=32, channels_bn=48, compression_level=2, act_layer_type=None, bias=True,
                                               kernel_size=5), (2, 37, 45)),
                                         # the reference's default activation family: stride-1 pre-convolutions + LeakyReLU / ReLU
                                         (dict(channels_net=32, channels_bn=48, compression_level=3, act_layer_type='LeakyReLU',
                                               bias=True), (2, 40, 56)),
                                         (dict(channels_net=64, channels_bn=48, compression_level=2, act_layer_type='ReLU',
                                               kernel_size=5), (1, 37, 45))])
def test_track_gradients_match_the_restatement(cae, cfgkw, shape):
    from cnn_autoencoder_amd import synth
    from oracle import train_oracle as T
    cfg = dict(synth.CANONICAL, **cfgkw)
    state, model, layers = _models(cae, cfg, seed=21)
    enc, dec = model['encoder'].module, model['decoder'].module
    n, h, w = shape
    torch.manual_seed(1)
    x = torch.rand(n, 3, h, w)
    # analysis
    y = enc(x.cuda())
    gy = torch.randn_like(y.detach()).cpu()
    y.backward(gy.cuda())
    act = cfg['act_layer_type'] if cfg['act_layer_type'] in ('LeakyReLU', 'ReLU') else None
    y_ref = T.analysis(x, layers['encoder'], bf16=True, act=act)
    y_ref.backward(gy)
    assert y.shape == y_ref.shape and rel(y.detach(), y_ref.detach()) < 1e-3
    y32 = T.analysis(x, [{k: (v.detach() if v is not None else None) for k, v in l.items()} for l in layers['encoder']], bf16=False,
                     act=act)
    assert rel(y.detach(), y32) < 3e-2
    got = _named_grads(enc, 'analysis_track')
    assert got, 'no gradients reached the encoder parameters'
    for i, l in enumerate(layers['encoder']):
        for key, sub in _param_names(l):
            if l.get(key) is not None:
                name = f'analysis_track.{i}.{sub}'
                # The restatement rounds to bf16 at the kernels' rounding points, so the two sides differ by summation
                # order only -- but an activation within float noise of a bf16 rounding boundary then rounds the other way
                # (2^-9 of its value), and a layer whose gradient sums over a few positions only (here down to 4 x 4)
                # shows single flips: measured 2.6e-4 .. 1.2e-3 on the last two layers for two summation orders of the
                # same kernels (chunk-wise / slice-wise staging), 1e-6 .. 7e-5 on the first two.
                positions = n * -(-h // 2 ** (i + 1)) * -(-w // 2 ** (i + 1))
                assert rel(got[name], l[key].grad) < (1e-3 if positions >= 256 else 2.5e-3), name
    # synthesis (from latents of realistic magnitude; h, w even multiples are not required)
    lh, lw = y.shape[2:]
    yq = torch.round(y_ref.detach()).requires_grad_(True)
    yq_dev = yq.detach().cuda().requires_grad_(True)
    x_r, _ = dec(yq_dev)
    gx = torch.randn_like(x_r[0].detach()).cpu()
    x_r[0].backward(gx.cuda())
    xr_ref = T.synthesis(yq, layers['decoder'], bf16=True, act=act)
    xr_ref.backward(gx)
    scale = max(1.0, float(xr_ref.abs().max()))
    assert float((x_r[0].detach().cpu() - xr_ref.detach()).abs().max()) / scale < 1e-3
    assert rel(yq_dev.grad, yq.grad) < 1e-3
    got = _named_grads(dec, 'synthesis_track')
    for i, l in enumerate(layers['decoder']):
        for key, sub in _param_names(l):
            if l.get(key) is not None:
                name = f'synthesis_track.{i}.{sub}'
                assert rel(got[name], l[key].grad) < 1e-3, name


@pytest.mark.parametrize('act,bias,ks', [('GDN', False, 3), ('LeakyReLU', True, 3), (None, True, 5)])
def test_residual_unit_gradients_match_the_restatement(cae, act, bias, ks):
    """Residual units (_autoencoders.py:104-174, :230-304) under autograd: outputs, parameter gradients and the latent
    gradient of both tracks against torch-CPU autograd of the restatement with the kernels' rounding points."""
    from oracle import train_oracle as T
    from conftest import residual_oracle_units
    torch.manual_seed(11)
    kw = dict(channels_org=3, channels_net=32, channels_bn=48, compression_level=2, kernel_size=ks, bias=bias,
              use_residual=True, act_layer_type=act)
    enc, dec = cae.Analyzer(**kw).cuda().train(), cae.Synthesizer(**kw).cuda().train()
    with torch.no_grad():  # GDN parameters away from their initial point (off-diagonal gamma exercised)
        for mod in list(enc.modules()) + list(dec.modules()):
            if isinstance(mod, cae.GDN):
                mod.beta.add_(0.1 * torch.rand_like(mod.beta))
                mod.gamma.add_(0.05 * torch.rand_like(mod.gamma))
    act_name = act if act in ('LeakyReLU', 'ReLU') else None
    x = torch.rand(2, 3, 40, 56)
    for name, mod, track, inp, synthesis in (('analysis', enc, enc.analysis_track, x, False),
                                             ('synthesis', dec, dec.synthesis_track, 2.0 * torch.randn(2, 48, 5, 7), True)):
        units, pairs = residual_oracle_units(track, act_name)
        xin = inp.clone().requires_grad_(True)
        ref = T.residual_track(xin, units, synthesis, bf16=True)
        xdev = inp.cuda().requires_grad_(True)
        out = mod(xdev)
        out = out[0][0] if synthesis else out
        assert out.shape == ref.shape
        g = torch.randn_like(ref.detach())
        ref.backward(g)
        out.backward(g.cuda())
        scale = max(1.0, float(ref.detach().abs().max()))
        assert float((out.detach().cpu() - ref.detach()).abs().max()) / scale < 2e-3, name
        if synthesis:
            assert rel(xdev.grad, xin.grad) < 2.5e-3, name
        got = {n: p.grad.detach().cpu() for n, p in mod.named_parameters() if p.grad is not None}
        prefix = 'synthesis_track.' if synthesis else 'analysis_track.'
        assert len(got) == len(pairs), (sorted(got), [n for n, _ in pairs])
        for pname, leaf in pairs:
            assert rel(got[prefix + pname], leaf.grad) < 2.5e-3, (name, pname)


def test_fused_clip_adam_equals_the_torch_loop(cae, monkeypatch):
    """cae_t_clip_adam (all optimisers of a step in two launches) against the reference's loop -- clip_grad_norm_(1.0),
    torch.optim.Adam.step(), zero_grad() per optimiser (train_cae_ms.py:221-230) -- on the same gradients, three steps:
    parameters and the optimisers' own state (exp_avg, exp_avg_sq, step) agree."""
    from cnn_autoencoder_amd import train
    torch.manual_seed(3)
    shapes = [(128, 128, 3, 3), (128,), (128, 128), (5000,), (48, 1, 3), (7,)]

    def build():
        ps = [torch.nn.Parameter(torch.randn(*s, device='cuda') * 0.1) for s in shapes]
        opts = {'a': torch.optim.Adam([dict(params=ps[:3], lr=1e-3, weight_decay=0.0)]),
                'b': torch.optim.Adam([dict(params=ps[3:5], lr=3e-3, weight_decay=0.01)], betas=(0.8, 0.95)),
                'c_aux': torch.optim.Adam([dict(params=ps[5:], lr=1e-2)])}
        return ps, opts
    ps_a, opts_a = build()
    ps_b, opts_b = build()
    for a, b in zip(ps_a, ps_b):
        b.data.copy_(a.data)
    for step in range(3):
        grads = [torch.randn_like(p) * (10.0 if step == 1 else 0.01) for p in ps_a]  # step 1: the clip is active
        for p, g in zip(ps_a, grads):
            p.grad = g.clone()
        for p, g in zip(ps_b, grads):
            p.grad = g.clone()
        monkeypatch.setenv('CAE_FUSED_OPTIM', '0')
        assert not train.fused_clip_adam(opts_a)
        for opt in opts_a.values():
            torch.nn.utils.clip_grad_norm_(opt.param_groups[0]['params'], max_norm=1.0)
            opt.step()
            opt.zero_grad()
        monkeypatch.setenv('CAE_FUSED_OPTIM', '1')
        assert train.fused_clip_adam(opts_b)
        assert all(p.grad is None for p in ps_b)
        for a, b in zip(ps_a, ps_b):
            assert rel(b.detach(), a.detach()) < 2e-6
    for (ka, oa), (kb, ob) in zip(opts_a.items(), opts_b.items()):
        for pa, pb in zip(oa.param_groups[0]['params'], ob.param_groups[0]['params']):
            sa, sb = oa.state[pa], ob.state[pb]
            assert float(sa['step']) == float(sb['step']) == 3.0
            # (the clip coefficient comes from another fp32 summation order of 1.6e5 squares: ~5e-6, squared in exp_avg_sq)
            assert rel(sb['exp_avg'], sa['exp_avg']) < 2e-5 and rel(sb['exp_avg_sq'], sa['exp_avg_sq']) < 5e-5
    # anything but plain Adam keeps the torch loop
    assert not train.fused_clip_adam({'x': torch.optim.SGD(ps_b[:1], lr=0.1)})


def test_twenty_training_steps_follow_the_restatement(cae):
    """train.train_step (forward_func -> GeneralLoss -> backward -> aux backward -> clip 1.0 -> per-module Adam, the
    quantiles in the `_aux` optimiser) against the same loop on the CPU restatement, same noise: the loss curve matches
    step for step."""
    from cnn_autoencoder_amd import criteria, synth, train
    from oracle import train_oracle as T
    cfg = dict(synth.CANONICAL, channels_net=32, channels_bn=48, compression_level=3)
    state, model, layers = _models(cae, cfg, seed=22)
    eb = model['fact_ent'].module
    n_filters = len(eb.filters)
    eb_ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in eb.named_parameters()}
    target = eb.target.detach().cpu()
    lam = 0.01
    criterion = criteria.GeneralLoss(distortion_lambda=lam)
    opts = train.setup_optim(model, learning_rate=1e-3, aux_learning_rate=1e-2)
    assert set(opts) == {'encoder', 'decoder', 'fact_ent', 'fact_ent_aux'}
    flat = lambda ls: [t for l in ls for t in l.values() if t is not None]  # noqa: E731
    ref_groups = dict(encoder=flat(layers['encoder']), decoder=flat(layers['decoder']),
                      fact_ent=[v for k, v in eb_ref.items() if 'quantiles' not in k],
                      fact_ent_aux=[eb_ref['quantiles']])
    ref_opts = {k: torch.optim.Adam([dict(params=v, lr=1e-2 if k.endswith('_aux') else 1e-3)]) for k, v in ref_groups.items()}
    gen = torch.Generator().manual_seed(3)
    losses, ref_losses = [], []
    for step in range(20):
        x = torch.rand(4, 3, 48, 64, generator=gen)
        noise = torch.rand(4, 48, 6, 8, generator=gen) - 0.5
        eb.fixed_noise = noise
        ld = train.train_step(x.cuda(), model, criterion, opts)
        losses.append(float(ld['loss']))
        # restatement
        y = T.analysis(x, layers['encoder'])
        y_q, p_y = T.entropy_forward(eb_ref, y, noise, n_filters, form=eb.likelihood_form)
        x_r = T.synthesis(y_q, layers['decoder'])
        loss, rate, dist = T.rd_loss(x, x_r, p_y, lam)
        loss.backward()
        T.aux_loss(eb_ref, n_filters, target).backward()
        for opt in ref_opts.values():
            torch.nn.utils.clip_grad_norm_(opt.param_groups[0]['params'], max_norm=1.0)
            opt.step()
            opt.zero_grad()
        ref_losses.append(float(loss))
        assert float(ld['rate_loss']) == pytest.approx(float(rate), rel=2e-3), step
        assert float(ld['dist'][0]) == pytest.approx(float(dist), rel=2e-3), step
    np.testing.assert_allclose(losses, ref_losses, rtol=2e-3)
    assert losses[-1] < losses[0]  # and it trains
    # parameters after 20 steps
    w_ref = layers['encoder'][0]['weight'].detach()
    w_got = model['encoder'].module.analysis_track[0].model[0].weight.detach().cpu()
    assert rel(w_got, w_ref) < 1e-2
    assert rel(eb.quantiles.detach().cpu(), eb_ref['quantiles'].detach()) < 1e-3


@pytest.mark.parametrize('residual,act', [(False, 'LeakyReLU'), (False, 'GDN'), (True, 'ReLU')])
def test_batch_norm_units_train_on_batch_statistics(cae, residual, act):
    """nn.BatchNorm2d in training mode inside the units (_autoencoders.py:72-73, :87-88): outputs and every gradient (batch-norm
    affine parameters included) against torch-CPU autograd with F.batch_norm(training=True); running statistics updated as
    nn.BatchNorm2d does."""
    from conftest import residual_oracle_units
    from oracle import train_oracle as T
    torch.manual_seed(13)
    kw = dict(channels_org=3, channels_net=32, channels_bn=48, compression_level=2, bias=True, batch_norm=True,
              use_residual=residual, act_layer_type=act)
    enc, dec = cae.Analyzer(**kw).cuda().train(), cae.Synthesizer(**kw).cuda().train()
    with torch.no_grad():
        for mod in list(enc.modules()) + list(dec.modules()):
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.uniform_(-0.2, 0.2)
    act_name = act if act in ('LeakyReLU', 'ReLU') else None
    for name, mod, track, inp, synthesis in (('analysis', enc, enc.analysis_track, torch.rand(4, 3, 40, 56), False),
                                             ('synthesis', dec, dec.synthesis_track, 2.0 * torch.randn(4, 48, 5, 7), True)):
        units, pairs = residual_oracle_units(track, act_name)
        xin = inp.clone().requires_grad_(True)
        ref = T.residual_track(xin, units, synthesis, bf16=True)
        bns = [m for m in mod.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        assert bns and all(float(m.num_batches_tracked) == 0 for m in bns)
        xdev = inp.cuda().requires_grad_(True)
        out = mod(xdev)
        out = out[0][0] if synthesis else out
        g = torch.randn_like(ref.detach())
        ref.backward(g)
        out.backward(g.cuda())
        scale = max(1.0, float(ref.detach().abs().max()))
        assert float((out.detach().cpu() - ref.detach()).abs().max()) / scale < 2e-3, name
        if synthesis:
            assert rel(xdev.grad, xin.grad) < 2.5e-3, name
        got = {n: p.grad.detach().cpu() for n, p in mod.named_parameters() if p.grad is not None}
        prefix = 'synthesis_track.' if synthesis else 'analysis_track.'
        assert len(got) == len(pairs), (sorted(got), [n for n, _ in pairs])
        # Parameters whose true gradient is ZERO hold rounding noise on both sides: a bias in front of a batch norm (the batch
        # mean is subtracted again), and any per-channel shift in front of a reflect-padded convolution + batch norm (a constant
        # stays constant under reflect padding).  Such parameters are recognised by their size against the track's largest gradient.
        # (relative part: measured up to 3.1e-3 -- bf16 rounding flips as in the other gradient tests, one more normalisation)
        gmax = max(float(leaf.grad.abs().max()) for _, leaf in pairs)
        pre_bn = {id(d['bias']): float(d['weight'].grad.abs().max()) for U in units for d in U['stages'] + [U]
                  if d.get('bn') is not None and d.get('bias') is not None}
        for pname, leaf in pairs:
            if id(leaf) in pre_bn:  # bias in front of a batch norm: sums of rounding errors on both sides
                assert float(got[prefix + pname].abs().max()) < 1e-2 * pre_bn[id(leaf)], (name, pname)
                assert float(leaf.grad.abs().max()) < 1e-2 * pre_bn[id(leaf)], (name, pname)
            elif float(leaf.grad.abs().max()) < 3e-3 * gmax:  # noise around zero on the restatement's side: noise here too
                assert float(got[prefix + pname].abs().max()) < 1e-2 * gmax, (name, pname)
            else:
                assert rel(got[prefix + pname], leaf.grad) < 5e-3, (name, pname)
        # running statistics moved off their initial values (0 / 1) by one momentum step
        assert all(float(m.num_batches_tracked) == 1 and float(m.running_mean.abs().max()) > 0 for m in bns)


@pytest.mark.parametrize('act,residual', [('ReLU', False), ('GDN', True)])
def test_grouped_layers_train(cae, act, residual):
    """groups=True (depthwise layers, groups = channels_in; _autoencoders.py:64-66, :79-81): the dense kernels run on the
    block-diagonal embedding of the grouped weight, its gradient is the block diagonal of theirs -- against torch-CPU autograd of
    the restatement on the same embedding (that the embedding equals the grouped layer is pinned by the var_groups_* fixtures)."""
    from conftest import residual_oracle_units
    from oracle import train_oracle as T
    torch.manual_seed(17)
    act_name = act if act in ('LeakyReLU', 'ReLU') else None
    common = dict(compression_level=2, bias=True, groups=True, use_residual=residual, act_layer_type=act)
    enc = cae.Analyzer(channels_org=4, channels_net=8, channels_bn=16, **common).cuda().train()
    dec = cae.Synthesizer(channels_org=8, channels_net=8, channels_bn=8, **common).cuda().train()
    for name, mod, track, inp, synthesis in (('analysis', enc, enc.analysis_track, torch.rand(2, 4, 40, 56), False),
                                             ('synthesis', dec, dec.synthesis_track, 2.0 * torch.randn(2, 8, 5, 7), True)):
        units, pairs = residual_oracle_units(track, act_name)
        assert any(l.shape[1] == 1 for _, l in pairs if l.dim() == 4)  # depthwise parameters
        xin = inp.clone().requires_grad_(True)
        ref = T.residual_track(xin, units, synthesis, bf16=True)
        xdev = inp.cuda().requires_grad_(True)
        out = mod(xdev)
        out = out[0][0] if synthesis else out
        g = torch.randn_like(ref.detach())
        ref.backward(g)
        out.backward(g.cuda())
        scale = max(1.0, float(ref.detach().abs().max()))
        assert float((out.detach().cpu() - ref.detach()).abs().max()) / scale < 2e-3, name
        got = {n: p.grad.detach().cpu() for n, p in mod.named_parameters() if p.grad is not None}
        prefix = 'synthesis_track.' if synthesis else 'analysis_track.'
        assert len(got) == len(pairs)
        for pname, leaf in pairs:
            assert got[prefix + pname].shape == leaf.shape and rel(got[prefix + pname], leaf.grad) < 2.5e-3, (name, pname)


def test_training_steps_of_a_residual_model(cae):
    """train.train_step on a residual GDN model (use_residual=True): every parameter of both tracks receives a gradient and
    the rate-distortion loss falls over a few steps on a fixed batch."""
    from cnn_autoencoder_amd import criteria, train
    torch.manual_seed(5)
    model = cae.setup_modules(channels_org=3, channels_net=32, channels_bn=48, compression_level=2, use_residual=True,
                              act_layer_type='GDN')
    model = {k: m.cuda().train() for k, m in model.items()}
    criterion = criteria.GeneralLoss(distortion_lambda=0.01)
    opts = train.setup_optim(model, learning_rate=1e-3, aux_learning_rate=1e-2)
    x = torch.rand(4, 3, 48, 64).cuda()
    before = {n: p.detach().clone() for k in ('encoder', 'decoder') for n, p in model[k].named_parameters()}
    losses = [float(train.train_step(x, model, criterion, opts)['loss']) for _ in range(8)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
    moved = [n for k in ('encoder', 'decoder') for n, p in model[k].named_parameters() if not torch.equal(p.detach(), before[n])]
    assert len(moved) == len(before), sorted(set(before) - set(moved))


@pytest.mark.parametrize('form', ['plain', 'sign_trick'])
def test_fused_density_kernels_match_the_elementwise_graph(cae, form, monkeypatch):
    """cae_t_density_forward / backward (train-mode EntropyBottleneck on the GPU) against the element-wise torch graph of
    the same module (CAE_EB_FUSED=0) and the CPU restatement: outputs, likelihoods (incl. elements clamped at the 1e-9
    bound, where the LowerBound rule decides the gradient), gradient with respect to the input and to every parameter."""
    from oracle import train_oracle as T
    from conftest import residual_oracle_units
    torch.manual_seed(11)
    c = 40
    eb = cae.EntropyBottleneck(c, likelihood_form=form).cuda().train()
    with torch.no_grad():
        for name, p in eb.named_parameters():
            if 'factor' in name:
                p.copy_(0.3 * torch.randn_like(p))
            elif 'matrix' in name:
                p.add_(0.2 * torch.randn_like(p))
    y = (torch.randn(3, c, 9, 7) * 4).cuda()
    y[0, :, 0, 0] = 300.0  # far tail: p < 1e-9, clamped
    y[1, :, 1, 1] = -300.0
    noise = (torch.rand(3, c, 9, 7) - 0.5).cuda()
    eb.fixed_noise = noise
    w_out, w_lik = torch.randn(3, c, 9, 7).cuda(), torch.randn(3, c, 9, 7).cuda()

    def run(fused):
        monkeypatch.setenv('CAE_EB_FUSED', '1' if fused else '0')
        yy = y.clone().requires_grad_(True)
        eb.zero_grad()
        out, lik = eb(yy)
        ((out * w_out).sum() + (torch.log2(lik) * w_lik).sum()).backward()
        return out.detach(), lik.detach(), yy.grad.clone(), {k: v.grad.clone() for k, v in eb.named_parameters() if v.grad is not None}

    o1, l1, gy1, gp1 = run(True)
    o0, l0, gy0, gp0 = run(False)
    assert torch.equal(o1, o0)
    np.testing.assert_allclose(l1.cpu().numpy(), l0.cpu().numpy(), rtol=2e-5, atol=1e-12)
    assert float((l1 == 1e-9).float().mean()) > 0.005  # the clamped elements are there
    assert rel(gy1.cpu(), gy0.cpu()) < 1e-4
    assert set(gp1) == set(gp0) and 'quantiles' not in gp1
    for k in gp0:
        assert rel(gp1[k].cpu(), gp0[k].cpu()) < 1e-4, k
    # and the CPU restatement (forward)
    params = {k: v.detach().cpu() for k, v in eb.named_parameters()}
    _, lik_ref = T.entropy_forward(params, y.cpu(), noise.cpu(), len(eb.filters), form=form)
    np.testing.assert_allclose(l1.cpu().numpy(), lik_ref.numpy(), rtol=1e-4, atol=1e-12)


def test_fused_reparametrisation_matches_the_torch_ops(cae):
    """NonNegativeParametrizer under autograd on the GPU (cae_t_reparam_*: one kernel each way) against its torch-op form
    on the CPU: values and the LowerBound gradient rule (elements below the bound pass only negative gradients)."""
    from cnn_autoencoder_amd.modules import NonNegativeParametrizer
    torch.manual_seed(2)
    for minimum in (0.0, 1e-6):
        rp = NonNegativeParametrizer(minimum=minimum)
        bound = float(rp.lower_bound.bound)
        x = torch.cat([torch.rand(500) * 0.2, torch.full((20,), bound * 0.5), torch.tensor([bound, -1.0])])
        w = torch.randn_like(x)
        x_cpu = x.clone().requires_grad_(True)
        (rp(x_cpu) * w).sum().backward()
        x_gpu = x.cuda().requires_grad_(True)
        out = rp.cuda()(x_gpu)
        (out * w.cuda()).sum().backward()
        np.testing.assert_allclose(out.detach().cpu().numpy(), rp.cpu()(x).numpy(), rtol=1e-6, atol=1e-12)
        np.testing.assert_allclose(x_gpu.grad.cpu().numpy(), x_cpu.grad.numpy(), rtol=1e-6, atol=1e-12)
        assert float((x_cpu.grad == 0).float().mean()) > 0.005  # blocked gradients are part of the case
