"""The BASELINE.json configurations themselves under test (not just the kernels on toy shapes).

config 2  train_ae, english hps with enc_size = emb_size = 1024, **B = 256, bf16** on one MI355X: one step with the
          DEFAULT kernel dispatch (this is where gemm_conv_p8m16_kernel and gemm_wgrad_p8_kernel are chosen), spot-checked
          against fp64 dot products taken on the host over the bf16 operands the kernels really read; the sticky GRU status
          word; loss falls over repeated steps; hipGraph replay == eager at this size.
config 1  16 synthetic segments, 2 speakers, 10 iterations through Trainer.train(mode='pretrain_AE') (the reference's own loop,
          trainer.py:320-347) with the host DataLoader -> DevicePrefetcher in front, log / checkpoint cadence, and
          main.main(['--train_ae', '--synthetic', ...]).
config 3  (DDP) rehearsed on ONE GPU: two rank processes on cuda:0 over gloo drive the product AEStep -- multi-graph step ==
          eager step bit for bit, replicas stay bit-identical, and the 1/world-folded update equals a single-rank step on
          the global batch.

Tolerances (bf16 storage, fp32 accumulate): an output element y = round_bf16(sum_k a_k w_k) is compared with the fp64 sum of
the same bf16 operands: |y - ref| <= 2^-8 |ref| + 2e-4 sum|a_k w_k| (output rounding + fp32 accumulation order); fp32 weight
gradients: |g - ref| <= 2e-4 sum|terms|."""
import io
import json
import os
import re
import sys
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import zs_amd  # noqa: F401
    return torch.device('cuda:0')


def _rows(act):
    """Act -> [rows, ld] view on the device."""
    return act.t[act.off:act.off + act.rows * act.ld].view(act.rows, act.ld)


def _bf(t):
    return t.float().to(torch.bfloat16).double()


def _check(name, got, ref, absref, out_rounding):
    got, ref, absref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64), np.asarray(absref, dtype=np.float64)
    tol = (2.0 ** -8) * np.abs(ref) * (1.0 if out_rounding else 0.0) + 2e-4 * absref + 1e-12
    bad = np.abs(got - ref) > tol
    assert not bad.any(), '%s: %d/%d spot checks off; worst |d|/tol = %.3g' % (name, bad.sum(), bad.size, (np.abs(got - ref) / tol).max())
    assert np.abs(ref).max() > 0, name + ': degenerate spot check (all references are zero)'


def test_config2_b256_bf16_default_dispatch_spot_checks(dev):
    from zs_amd import _lib as L, layers
    from zs_amd.model import Decoder, Encoder
    from zs_amd.trainer import AEStep
    torch.manual_seed(0)
    B, T, Fb, E, ch, nspk = 256, 128, 513, 1024, 1024, 102
    enc = Encoder(ns=0.01, dp=0.5, enc_size=E, seg_len=128, enc_mode='multilabel_binary', dtype='bf16').to(dev)
    dec = Decoder(ns=0.01, c_in=E, c_h=ch, c_a=nspk, seg_len=128, dtype='bf16').to(dev)
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(B, T, Fb, generator=g) * (1 - 1e-8) + 1e-8).to(dev)
    c = torch.randint(0, nspk, (B,), generator=g).to(dev)
    ae = AEStep(enc, dec, lr=1e-4, max_grad_norm=5.0, use_graph=False)
    l0 = ae.step(x, c, seed=11, update=False).item()
    layers.check_status(dev)
    assert np.isfinite(l0)
    de = dec._engine()
    tp = de.tape
    tag = '_%d_%d_%d' % (de.uid, B, T // 8)
    rng = np.random.RandomState(3)
    n_chk = 256

    # ---- (1) p8m16 forward: decoder conv5 (k3, 1024 -> 2048, T = 64: M = 16384, 512 tiles), pixel-shuffle channel packing
    xin, xe, ya, s_, yb, stt, Ti = tp['blocks'][2]
    assert Ti == 64 and ya.C == 2 * ch
    W = _bf(dec.conv5.weight.detach()).cpu()                # [2048, 1024, 3]
    bias = dec.conv5.bias.detach().double().cpu()
    xe_r, ya_r = _rows(xe), _rows(ya)
    bs, ts, ns_ = rng.randint(B, size=n_chk), rng.randint(Ti, size=n_chk), rng.randint(2 * ch, size=n_chk)
    got, ref, absref = [], [], []
    for b, t, n in zip(bs, ts, ns_):
        co = 2 * n if n < ch else 2 * (n - ch) + 1               # ZS_STORE_SPLIT2 packing: n' = r*C + c  <->  co = 2c + r
        acc, aabs = float(bias[co]), abs(float(bias[co]))
        for j in range(3):
            u = t + j - 1
            u = -u if u < 0 else (2 * (Ti - 1) - u if u >= Ti else u)
            prod = xe_r[b * Ti + u, :ch].double().cpu() * W[co, :, j]
            acc += float(prod.sum()); aabs += float(prod.abs().sum())
        r = acc if acc > 0 else 0.01 * acc
        got.append(float(ya_r[b * Ti + t, n])); ref.append(r); absref.append(aabs)
    _check('conv5 forward (p8m16)', got, ref, absref, True)

    # ---- (2) p8m16 forward: decoder dense1 (Linear 1024 -> 1024 on M = 32768 rows)
    xin, xe1, y1, y1e, y2, stt = tp['dense'][0]
    W = _bf(dec.dense1.weight.detach()).cpu()
    bias = dec.dense1.bias.detach().double().cpu()
    xe_r, y_r = _rows(xe1), _rows(y1)
    ms, ns_ = rng.randint(B * T, size=n_chk), rng.randint(ch, size=n_chk)
    got, ref, absref = [], [], []
    for m, n in zip(ms, ns_):
        prod = xe_r[m, :ch].double().cpu() * W[n]
        acc = float(prod.sum()) + float(bias[n])
        got.append(float(y_r[m, n])); ref.append(acc if acc > 0 else 0.01 * acc); absref.append(float(prod.abs().sum()) + abs(float(bias[n])))
    _check('dense1 forward (p8m16)', got, ref, absref, True)

    # ---- (3) p8m16 data gradient: decoder dense2's dgrad on the step's own dz2 (the shared output buffer of the step is
    #          overwritten by later layers, so the same call is issued once more into a fresh buffer)
    dz2 = de.ctx.act('d_dz20' + tag, B, T, ch)
    out = de.ctx.act('t_spot_dgrad', B, T, ch)
    de.dense[0][1].dgrad(dz2, T, out)
    torch.cuda.synchronize()
    Wd = _bf(dec.dense2.weight.detach()).cpu()               # [n, ci]
    dz_r, o_r = _rows(dz2), _rows(out)
    ms, cis = rng.randint(B * T, size=n_chk), rng.randint(ch, size=n_chk)
    got, ref, absref = [], [], []
    for m, ci in zip(ms, cis):
        prod = dz_r[m, :ch].double().cpu() * Wd[:, ci]
        got.append(float(o_r[m, ci])); ref.append(float(prod.sum())); absref.append(float(prod.abs().sum()))
    _check('dense2 data gradient (p8m16)', got, ref, absref, True)

    # ---- (4) gemm_wgrad_p8: dense5.weight [1024, 3072] = dz5^T cat[out, rnn, emb5 x T] over M = 32768 rows
    dz5 = de.ctx.act('d_dz5' + tag, B, T, ch)
    cat3 = tp['cat3']
    gw = dec.grad_view('dense5.weight')
    ncol = cat3.C
    assert ncol == 3 * ch
    cos, cis = rng.randint(ch, size=n_chk), rng.randint(ncol, size=n_chk)
    ycols = _rows(dz5)[:, torch.from_numpy(cos).to(dev)].double().cpu()          # [M, n_chk]
    xcols = _rows(cat3)[:, torch.from_numpy(cis).to(dev)].double().cpu()
    prod = ycols * xcols
    _check('dense5.weight gradient (wgrad_p8)', gw[torch.from_numpy(cos), torch.from_numpy(cis)].double().cpu().numpy(),
           prod.sum(0).numpy(), prod.abs().sum(0).numpy(), False)
    gb = dec.grad_view('dense5.bias')
    yb_ = _rows(dz5)[:, :ch].double().cpu()
    _check('dense5.bias gradient', gb.double().cpu().numpy(), yb_.sum(0).numpy(), yb_.abs().sum(0).numpy(), False)

    # ---- (5) gemm_wgrad_p8 with taps + reflect + split2 packing: conv5.weight [2048, 1024, 3]
    dza = de.ctx.act('d_dza2' + tag, B, Ti, 2 * ch)
    xin, xe, ya, s_, yb, stt, Ti = tp['blocks'][2]
    gw = dec.grad_view('conv5.weight')
    dza_r, xe_r = _rows(dza).view(B, Ti, -1), _rows(xe).view(B, Ti, -1)
    cos, cis, js = rng.randint(2 * ch, size=64), rng.randint(ch, size=64), rng.randint(3, size=64)
    got, ref, absref = [], [], []
    for co, ci, j in zip(cos, cis, js):
        npk = (co // 2) if co % 2 == 0 else ch + co // 2                          # packed column of output channel co
        ycol = dza_r[:, :, npk].double().cpu()                                     # [B, Ti]
        u = np.arange(Ti) + j - 1
        u = np.where(u < 0, -u, np.where(u >= Ti, 2 * (Ti - 1) - u, u))
        xcol = xe_r[:, torch.from_numpy(u).to(dev), ci].double().cpu()
        prod = ycol * xcol
        got.append(float(gw[co, ci, j])); ref.append(float(prod.sum())); absref.append(float(prod.abs().sum()))
    _check('conv5.weight gradient (wgrad_p8, taps)', got, ref, absref, False)

    # ---- (6) emb5.weight through the thin path (engine.py: ZS_THIN_EMB5, default at this width).  emb5 enters twice
    #          (model/model.py:353, 357): added to the GRU input -> per-sample column sums of the RAW dgi W_ih GEMM output; and as
    #          the third K block of dense5 -> (sum_t dz5[b, t]) W5[:, 2ch:] with the column sums rounded to bf16 for that B-row GEMM.
    H = ch // 2
    dgi = _rows(de.ctx.act('d_dgi' + tag, B, T, 6 * H)).view(B, T, -1)
    dz5r = _rows(dz5).view(B, T, -1)
    Wih = torch.cat([_bf(dec.RNN.weight_ih_l0.detach()), _bf(dec.RNN.weight_ih_l0_reverse.detach())], 0).cpu()    # [6H, ch]
    W5e = _bf(dec.dense5.weight.detach()[:, 2 * ch:]).cpu()                                                          # [n, ch]
    ge5 = dec.grad_view('emb5.weight')
    cc = c.cpu().numpy()
    spk_present = np.unique(cc)
    got, ref, tol = [], [], []
    for _ in range(24):
        sp, col = int(spk_present[rng.randint(len(spk_present))]), int(rng.randint(ch))
        bsel = torch.from_numpy(np.flatnonzero(cc == sp)).to(dev)
        t1 = dgi[bsel][:, :, :6 * H].double().cpu() * Wih[:, col]                       # [nb, T, 6H]
        S = _bf(dz5r[bsel][:, :, :ch].float().sum(1)).cpu()                             # [nb, ch]: fp32 column sums, stored as bf16
        t2 = S * W5e[:, col]
        got.append(float(ge5[sp, col])); ref.append(float(t1.sum() + t2.sum()))
        # (the epilogue sums the bf16-staged output values: one output rounding per (b, t) term on top of the fp32 accumulation)
        tol.append(2e-4 * float(t1.abs().sum()) + (2.0 ** -8) * float(t1.sum(-1).abs().sum()) + (2.0 ** -7) * float(t2.abs().sum()) + 1e-12)
    got, ref, tol = np.array(got), np.array(ref), np.array(tol)
    assert np.abs(ref).max() > 0
    assert (np.abs(got - ref) <= tol).all(), 'emb5.weight gradient (thin path): worst |d|/tol = %.3g' % (np.abs(got - ref) / tol).max()

    # ---- (7) the encoder's last data gradient: only the conv bank's 896 of conv2's 1409 input columns (the rest is the input
    #          gradient the reference computes and drops, utils.py:43-45), masked by lrelu'(cat)
    ee = enc._engine()
    etag = '_%d_%d_%d' % (ee.uid, B, T)
    c1, c2 = 128, 512
    dz = _rows(ee.ctx.act('e_dzy2' + etag, B, T, c2))
    dcat = _rows(ee.ctx.act('e_dcat' + etag, B, T, ee.ncat))
    cat = _rows(ee.tape['cat'])
    W2 = _bf(enc.conv2.weight.detach()[:, :, 0]).cpu()                                  # [512, 1409]
    ms, ns_ = rng.randint(B * T, size=n_chk), rng.randint(7 * c1, size=n_chk)
    got, ref, absref = [], [], []
    for m, n in zip(ms, ns_):
        prod = dz[m, :c2].double().cpu() * W2[:, n]
        r = float(prod.sum()) * (1.0 if float(cat[m, n]) > 0 else 0.01)
        got.append(float(dcat[m, n])); ref.append(r); absref.append(float(prod.abs().sum()))
    _check('conv2 data gradient, conv-bank columns (p8m16, n_cols = 896)', got, ref, absref, True)

    # ---- the loss of repeated steps on one batch falls; determinism of the step at this size
    l0b = ae.step(x, c, seed=11, update=False).item()
    assert l0b == l0
    losses = [ae.step(x, c, seed=100 + i).item() for i in range(10)]
    layers.check_status(dev)
    print('B=256 bf16 losses', ['%.4f' % v for v in [l0] + losses])
    assert losses[-1] < l0


def test_config2_b256_graph_equals_eager(dev):
    """hipGraph replay of the B=256 bf16 step (what bench.py times) == the same launches issued eagerly, bit for bit."""
    from zs_amd import layers
    from zs_amd.model import Decoder, Encoder
    from zs_amd.trainer import AEStep
    B, T, Fb, E, ch, nspk = 256, 128, 513, 1024, 1024, 102
    res = []
    for graph_warmup in (2, 10 ** 9):
        torch.manual_seed(0)
        enc = Encoder(ns=0.01, dp=0.5, enc_size=E, seg_len=128, enc_mode='multilabel_binary', dtype='bf16').to(dev)
        dec = Decoder(ns=0.01, c_in=E, c_h=ch, c_a=nspk, seg_len=128, dtype='bf16').to(dev)
        ae = AEStep(enc, dec, lr=1e-4, max_grad_norm=5.0, use_graph=True)
        ae.graph_warmup = graph_warmup
        g = torch.Generator().manual_seed(1)
        x = (torch.rand(B, T, Fb, generator=g) * (1 - 1e-8) + 1e-8).to(dev)
        c = torch.randint(0, nspk, (B,), generator=g).to(dev)
        losses = [ae.step(x, c).item() for _ in range(5)]
        layers.check_status(dev)
        res.append((losses, enc.flat_params()[0].clone(), dec.flat_params()[0].clone(), len(ae._graphs)))
        del ae, enc, dec
        torch.cuda.empty_cache()
    (la, ea, da, na), (lb, eb, db, nb) = res
    assert na == 1 and nb == 0
    assert la == lb, (la, lb)
    assert torch.equal(ea, eb) and torch.equal(da, db)


def test_static_input_buffers_equal_passed_tensors(dev):
    """AEStep.static_inputs: a loader that writes the batch straight into the buffers the captured step reads (no device-to-device
    copy in front of a replay, what bench.py does for its resident batch) gives the step passed its own tensors, bit for bit."""
    from zs_amd import layers
    from zs_amd.model import Decoder, Encoder
    from zs_amd.trainer import AEStep
    B, T, Fb, E, ch, nspk = 32, 128, 513, 1024, 1024, 102
    res = []
    for static in (False, True):
        torch.manual_seed(0)
        enc = Encoder(ns=0.01, dp=0.5, enc_size=E, seg_len=128, enc_mode='multilabel_binary', dtype='bf16').to(dev)
        dec = Decoder(ns=0.01, c_in=E, c_h=ch, c_a=nspk, seg_len=128, dtype='bf16').to(dev)
        ae = AEStep(enc, dec, lr=1e-4, max_grad_norm=5.0, use_graph=True)
        g = torch.Generator().manual_seed(1)
        batches = [((torch.rand(B, T, Fb, generator=g) * (1 - 1e-8) + 1e-8).to(dev), torch.randint(0, nspk, (B,), generator=g).to(dev))
                   for _ in range(5)]
        losses = []
        if static:
            xs, cs = ae.static_inputs(B, T, Fb)
        for x, c in batches:
            if static:
                xs.copy_(x); cs.copy_(c)
                losses.append(ae.step(xs, cs).item())
            else:
                losses.append(ae.step(x, c).item())
        layers.check_status(dev)
        res.append((losses, enc.flat_params()[0].clone(), dec.flat_params()[0].clone(), len(ae._graphs)))
        del ae, enc, dec
        torch.cuda.empty_cache()
    (la, ea, da, na), (lb, eb, db, nb) = res
    assert na == 1 and nb == 1
    assert la == lb, (la, lb)
    assert torch.equal(ea, eb) and torch.equal(da, db)


def _config1_hps(tmp_path, **over):
    d = json.load(open(os.path.join(ROOT, 'hps', 'zerospeech_english_1024.json')))
    d.update(n_speakers=2, n_target_speakers=2, batch_size=16, enc_pretrain_iters=10, max_to_keep=3)
    d.update(over)
    p = str(tmp_path / 'hps_config1.json')
    json.dump(d, open(p, 'w'))
    return p


def test_config1_trainer_loop_pretrain_ae(dev, tmp_path, monkeypatch):
    """BASELINE config 1 through the product loop: 16 segments re-served by the reference's wrap rule, 10 iterations, loss
    falls, console line / scalar tags / checkpoint cadence + rotation as in trainer.py:335-346, checkpoint round trip."""
    from zs_amd.dataloader import DataLoader, SyntheticDataset
    from zs_amd.hps import Hps
    from zs_amd.trainer import Trainer
    monkeypatch.setenv('ZS_CKPT_EVERY', '2')
    hps = Hps(_config1_hps(tmp_path)).get_tuple()
    torch.manual_seed(0)
    ds = SyntheticDataset(16, seg_len=hps.seg_len, n_speakers=2, seed=0)
    dl = DataLoader(ds, hps.batch_size)
    log_dir = str(tmp_path / 'log')
    tr = Trainer(hps, dl, hps.g_mode, hps.enc_mode, log_dir=log_dir, dtype='bf16', device=dev)
    model_path = str(tmp_path / 'model.pth')
    buf = io.StringIO()
    with redirect_stdout(buf):
        tr.train(model_path, 'train', mode='pretrain_AE')
    lines = [l for l in buf.getvalue().replace('\r', '\n').split('\n') if l.startswith('pre_AE:')]
    assert len(lines) == 10
    m = [re.match(r'pre_AE:\[(\d{6})/(\d{6})\], loss_rec=(\d+\.\d{3})$', l) for l in lines]
    assert all(m), lines
    assert [int(k.group(1)) for k in m] == list(range(1, 11)) and all(int(k.group(2)) == 10 for k in m)
    losses = [float(k.group(3)) for k in m]
    assert losses[-1] < losses[0], losses
    assert dl.index == 0                                              # one batch re-served (dataloader.py:48-49)
    # checkpoints every 2 iterations, rolling window: len >= max_keep -> the oldest is removed (trainer.py:118-125)
    kept = sorted(f for f in os.listdir(str(tmp_path)) if f.startswith('model.pth-ae-'))
    assert kept == ['model.pth-ae-10', 'model.pth-ae-8'], kept
    ck = torch.load(model_path + '-ae-10', map_location='cpu', weights_only=True)
    assert {'encoder', 'decoder', 'generator', 'classifier'} <= set(ck.keys())
    assert set(ck['encoder'].keys()) == set(tr.Encoder.state_dict().keys())
    # scalars: tag '<flag>/pre_loss_rec' at iteration 0 (every 100th), step = iteration + 1
    sc = os.path.join(log_dir, 'scalars.jsonl')
    if os.path.exists(sc):
        recs = [json.loads(l) for l in open(sc)]
        assert recs and recs[0]['tag'] == 'train/pre_loss_rec' and recs[0]['step'] == 1
    # reload into a fresh trainer: identical parameters
    tr2 = Trainer(hps, None, hps.g_mode, hps.enc_mode, log_dir=log_dir, dtype='bf16', device=dev)
    with redirect_stdout(io.StringIO()) as out:
        tr2.load_model(model_path + '-ae-10', hps.load_model_list)
    assert '[encoder], [decoder], [generator], Loaded!' in out.getvalue()
    assert torch.equal(tr2.Encoder.flat_params()[0], tr.Encoder.flat_params()[0])
    assert torch.equal(tr2.Decoder.flat_params()[0], tr.Decoder.flat_params()[0])
    # the enc_only=False (generator) branch of test_step runs on the loaded weights
    xs = torch.from_numpy(ds.lin[:1])
    xd, e = tr2.test_step(xs, torch.tensor([1]), enc_only=False, verbose=False)
    assert xd.shape == (1, 513, 128) and e.shape == (1, 1024, 16) and np.isfinite(xd).all()


def test_config1_main_train_ae_synthetic(dev, tmp_path, monkeypatch, capsys):
    sys.path.insert(0, ROOT)
    import main
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv('ZS_CKPT_EVERY', '4')
    hp_path = _config1_hps(tmp_path, enc_pretrain_iters=4)
    main.main(['--train_ae', '--synthetic', '--hps_path', hp_path, '--ckpt_dir', str(tmp_path / 'ck'), '--dtype', 'bf16'])
    out = capsys.readouterr().out
    assert 'pre_AE:[000004/000004]' in out
    assert os.path.exists(str(tmp_path / 'ck' / 'model.pth-ae-4'))


def test_load_model_takes_the_target_classifier_from_clf_path(dev, tmp_path):
    """trainer.py:157-161: with clf_path the TargetClassifier comes from ANOTHER checkpoint (tag `[target_classifier_another]`),
    everything else from model_path."""
    from zs_amd.hps import make_hps
    from zs_amd.trainer import Trainer
    hps = make_hps(enc_size=8, emb_size=32, n_speakers=4, n_target_speakers=2, max_to_keep=5)
    trs = []
    for seed in (1, 2):
        torch.manual_seed(seed)
        tr = Trainer(hps, None, hps.g_mode, hps.enc_mode, log_dir=str(tmp_path / 'log'), dtype='fp32', device=dev)
        tr.save_model(str(tmp_path / ('m%d.pth' % seed)), 's2', 1)
        trs.append(tr)
    tr3 = Trainer(hps, None, hps.g_mode, hps.enc_mode, log_dir=str(tmp_path / 'log'), dtype='fp32', device=dev)
    buf = io.StringIO()
    with redirect_stdout(buf):
        tr3.load_model(str(tmp_path / 'm1.pth-s2-1'), 'encoder, decoder, generator, target_classifier', clf_path=str(tmp_path / 'm2.pth-s2-1'))
    assert '[target_classifier_another]' in buf.getvalue() and '[encoder], [decoder], [generator]' in buf.getvalue()
    assert torch.equal(tr3.TargetClassifier.flat_params()[0], trs[1].TargetClassifier.flat_params()[0])
    assert not torch.equal(tr3.TargetClassifier.flat_params()[0], trs[0].TargetClassifier.flat_params()[0])
    assert torch.equal(tr3.Encoder.flat_params()[0], trs[0].Encoder.flat_params()[0])


def test_pretrain_c_and_train_loops_run(dev, tmp_path, monkeypatch):
    """The speaker-classifier modes of Trainer.train (trainer.py:349-465) run end to end on a small model."""
    from zs_amd.dataloader import DataLoader, SyntheticDataset
    from zs_amd.hps import make_hps
    from zs_amd.trainer import Trainer
    hps = make_hps(enc_size=32, emb_size=64, n_speakers=4, n_target_speakers=2, batch_size=4, dis_pretrain_iters=3, iters=2,
                   n_latent_steps=2, lat_sched_iters=2, max_to_keep=3)
    ds = SyntheticDataset(32, seg_len=128, n_speakers=4, seed=1)
    tr = Trainer(hps, DataLoader(ds, 4), hps.g_mode, hps.enc_mode, log_dir=str(tmp_path / 'log'), dtype='fp32', device=dev)
    buf = io.StringIO()
    with redirect_stdout(buf):
        tr.train(str(tmp_path / 'm.pth'), 'train', mode='pretrain_C')
        tr.train(str(tmp_path / 'm.pth'), 'train', mode='train')
    txt = buf.getvalue()
    assert 'pre_C:[000003/000003], loss_clf=' in txt and 'G:[000002/000002], loss_rec=' in txt and 'D-1:[000002/000002]' in txt


def test_device_prefetcher_never_serves_a_torn_batch(dev):
    """Distinct pageable batches, no host sync between steps, a slow consumer on the GPU: every batch the prefetcher hands out
    must equal the loader's sequence (the pinned staging buffer of a slot may only be rewritten after its H2D copy ran)."""
    from zs_amd.dataloader import DevicePrefetcher

    class Loader(object):
        def __init__(self):
            self.k = 0

        def __next__(self):
            self.k += 1
            return torch.full((8,), self.k, dtype=torch.int64), torch.full((8, 64, 513), float(self.k))

    pf = DevicePrefetcher(Loader(), dev)
    big = torch.randn(4096, 4096, device=dev)
    sums = []
    for k in range(1, 13):
        c, x = next(pf)
        for _ in range(6):
            big = torch.tanh(big @ big * 1e-3)             # keep the GPU busy so the host runs ahead
        sums.append((c.sum(), x.mean(), x.min(), x.max()))  # device scalars: no sync here
    torch.cuda.synchronize()
    for k, (cs, xm, xlo, xhi) in enumerate(sums, start=1):
        assert int(cs.item()) == 8 * k and xm.item() == float(k) and xlo.item() == float(k) and xhi.item() == float(k), k


def test_gru_timeout_is_sticky_and_raises_in_the_loop(dev, tmp_path):
    """A persistent GRU pass whose bounded spin expires (forced: spin limit 0) poisons the step silently on the device;
    the sticky status word makes Trainer.train raise at its next host sync instead of training on."""
    from zs_amd import _lib as L, layers
    from zs_amd.dataloader import DataLoader, SyntheticDataset
    from zs_amd.hps import make_hps
    from zs_amd.trainer import Trainer
    hps = make_hps(enc_size=32, emb_size=256, n_speakers=4, n_target_speakers=2, batch_size=64, enc_pretrain_iters=3)
    ds = SyntheticDataset(64, seg_len=128, n_speakers=4, seed=2)
    tr = Trainer(hps, DataLoader(ds, 64), hps.g_mode, hps.enc_mode, log_dir=str(tmp_path / 'log'), dtype='bf16', device=dev)
    layers.check_status(dev)
    old = L.set_option('gru_spin_limit', 0)
    try:
        with pytest.raises(L.ZsError, match='persistent GRU'):
            with redirect_stdout(io.StringIO()):
                tr.train(str(tmp_path / 'm.pth'), 'train', mode='pretrain_AE')
    finally:
        L.set_option('gru_spin_limit', old)
        torch.cuda.synchronize()
    layers.check_status(dev)                                  # cleared by the raise
    with redirect_stdout(io.StringIO()):
        tr.train(str(tmp_path / 'm.pth'), 'train', mode='pretrain_AE')      # and the loop runs again with the default limit


# ---- config 3 rehearsal: two ranks on one GPU over gloo -------------------------------------------------------------------

_DP_SCRIPT = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import torch
import zs_amd
from zs_amd import parallel, layers
from zs_amd.model import Decoder, Encoder
from zs_amd.trainer import AEStep
import torch.distributed as dist

torch.cuda.set_device(0)
rank, world, _ = parallel.init_from_env('gloo')
dev = torch.device('cuda:0')
out = {}

def build(seed):
    torch.manual_seed(seed)            # deliberately DIFFERENT per rank: sync must come from broadcast_params
    enc = Encoder(c_in=80, c_h1=16, c_h2=64, c_h3=32, ns=0.01, dp=0.0, enc_size=32, seg_len=128, enc_mode='multilabel_binary', dtype='fp32').to(dev)
    dec = Decoder(c_in=32, c_out=80, c_h=64, c_a=4, ns=0.01, seg_len=128, dtype='fp32').to(dev)
    return enc, dec

g = torch.Generator().manual_seed(7)
X = torch.rand(8, 128, 80, generator=g)
C = torch.randint(0, 4, (8,), generator=g)
NOISE = torch.rand(8, 16, 32, 2, generator=g)
lo, hi = parallel.shard_range(8, rank, world)

# (a) broadcast makes the replicas identical
enc, dec = build(100 + rank)
parallel.broadcast_params([enc, dec])
ref = [enc.flat_params()[0].clone(), dec.flat_params()[0].clone()]
for t in ref:
    t2 = t.clone(); dist.broadcast(t2, src=0)
    assert torch.equal(t, t2), 'replicas differ after broadcast_params'
init = [t.clone() for t in ref]

# (b) one eager multi-rank step with injected noise on this rank's half == single-rank step on the global batch
ae = AEStep(enc, dec, lr=1e-3, max_grad_norm=5.0, use_graph=False)
ae.step(X[lo:hi].to(dev), C[lo:hi].to(dev), noise=NOISE[lo:hi].contiguous().to(dev), noise_kind=1, update=False)
sq = [t.clone() for t in ae.grad_norms()]
gsum = [enc.flat_params()[1].clone(), dec.flat_params()[1].clone()]
ae.optimizer_step()
torch.cuda.synchronize()
p_multi = [enc.flat_params()[0].clone(), dec.flat_params()[0].clone()]
if rank == 0:
    # single-process reference on the global batch: same initial weights, no process group in play for this AEStep
    enc1, dec1 = build(1)
    enc1.flat_params()[0].copy_(init[0]); dec1.flat_params()[0].copy_(init[1]); enc1.mark_dirty(); dec1.mark_dirty()
    ae1 = AEStep(enc1, dec1, lr=1e-3, max_grad_norm=5.0, use_graph=False)
    class _One(object):
        scale = 1.0
        def start(self, f): pass
        def finish(self): pass
    ae1.reducer = _One()
    _ws = parallel.world_size
    parallel.world_size = lambda: 1
    ae1.step(X.to(dev), C.to(dev), noise=NOISE.contiguous().to(dev), noise_kind=1, update=False)
    g1 = [enc1.flat_params()[1].clone(), dec1.flat_params()[1].clone()]
    sq1 = [t.clone() for t in ae1.grad_norms()]
    ae1.optimizer_step()
    torch.cuda.synchronize()
    parallel.world_size = _ws
    errs = []
    for a, b in zip(gsum, g1):
        errs.append(((a * 0.5 - b).abs().max() / b.abs().max()).item())       # mean loss per rank: average of the two halves
    out['grad_rel_err'] = errs
    out['norm_rel_err'] = [abs((s.item() ** 0.5) * 0.5 - s1.item() ** 0.5) / (s1.item() ** 0.5) for s, s1 in zip(sq, sq1)]
    out['param_err_over_lr'] = [((a - b).abs().max() / 1e-3).item() for a, b in zip(p_multi, [enc1.flat_params()[0], dec1.flat_params()[0]])]

# (c) multi-graph (bucketed) multi-rank step == eager multi-rank step, bit for bit; replicas stay identical
res = []
for mode in ('graph', 'eager'):
    enc, dec = build(5)
    enc.dp = 0.5
    enc.flat_params()[0].copy_(init[0]); dec.flat_params()[0].copy_(init[1]); enc.mark_dirty(); dec.mark_dirty()
    ae = AEStep(enc, dec, lr=1e-3, max_grad_norm=5.0, use_graph=True)
    if mode == 'eager':
        ae.graph_warmup = 10 ** 9
    gg = torch.Generator().manual_seed(11 + rank)       # distinct per-rank data
    losses = []
    for i in range(6):
        x = torch.rand(4, 128, 80, generator=gg).to(dev)
        c = torch.randint(0, 4, (4,), generator=gg).to(dev)
        losses.append(ae.step(x, c).item())
    layers.check_status(dev)
    res.append((losses, enc.flat_params()[0].clone(), dec.flat_params()[0].clone(), sum(len(v['graphs']) for v in ae._graphs.values())))
(la, ea, da, na), (lb, eb, db, nb) = res
assert na == 6 and nb == 0, (na, nb)         # fwd + dec head | dec convs | enc main | enc bank | dec update | enc update
assert la == lb, (la, lb)
assert torch.equal(ea, eb) and torch.equal(da, db), 'graph and eager multi-rank steps differ'
for t in (ea, da):
    t2 = t.clone(); dist.broadcast(t2, src=0)
    assert torch.equal(t, t2), 'replicas diverged'
out['losses'] = la
if rank == 0:
    print('RESULT ' + json.dumps(out), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def test_config3_two_ranks_on_one_gpu_gloo(dev, tmp_path):
    import socket
    import subprocess
    script = str(tmp_path / 'dp_rehearsal.py')
    open(script, 'w').write(_DP_SCRIPT % {'root': ROOT})
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   ZS_FORCE_DEVICE='0', OMP_NUM_THREADS='2')
        procs.append(subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for pr in procs:
        try:
            o, _ = pr.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, (pr, o) in enumerate(zip(procs, outs)):
        assert pr.returncode == 0, 'rank %d failed:\n%s' % (r, o[-4000:])
    line = [l for l in outs[0].splitlines() if l.startswith('RESULT ')]
    assert line, outs[0][-2000:]
    res = json.loads(line[0][7:])
    print(res)
    assert max(res['grad_rel_err']) < 2e-5, res
    assert max(res['norm_rel_err']) < 1e-5, res
    assert max(res['param_err_over_lr']) < 0.05, res          # Adam moves every weight by ~lr: agreement to a few % of one move
    assert len(set(res['losses'])) == len(res['losses'])


# ---- config 3 over RCCL: the multi-rank step (six hipGraphs, five bucketed all-reduces between them) on ONE rank ------------------------

_RCCL_SCRIPT = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import torch
import zs_amd
from zs_amd import parallel, layers
from zs_amd.model import Decoder, Encoder
from zs_amd.trainer import AEStep
import torch.distributed as dist

rank, world, local = parallel.init_from_env('nccl')         # WORLD_SIZE=1 + ZS_FORCE_MULTI=1: a real RCCL communicator of one rank
assert dist.is_initialized() and dist.get_backend() == 'nccl' and parallel.multi_rank()
dev = torch.device('cuda', local)
res = []
for mode in ('rccl4', 'single'):
    os.environ['ZS_FORCE_MULTI'] = '1' if mode == 'rccl4' else '0'
    assert parallel.multi_rank() == (mode == 'rccl4')
    torch.manual_seed(3)
    enc = Encoder(c_in=80, c_h1=16, c_h2=64, c_h3=32, ns=0.01, dp=0.5, enc_size=32, seg_len=128, enc_mode='multilabel_binary', dtype='bf16').to(dev)
    dec = Decoder(c_in=32, c_out=80, c_h=64, c_a=4, ns=0.01, seg_len=128, dtype='bf16').to(dev)
    if mode == 'rccl4':
        parallel.broadcast_params([enc, dec])                 # (world 1: returns at once; bench.py broadcasts under multi_rank())
        for net in (enc, dec):
            dist.broadcast(net.flat_params()[0], src=0)
            net.mark_dirty()
    ae = AEStep(enc, dec, lr=1e-3, max_grad_norm=5.0, use_graph=True)
    gg = torch.Generator().manual_seed(11)
    losses = []
    for i in range(6 + ae.graph_warmup):
        x = torch.rand(8, 128, 80, generator=gg).to(dev)
        c = torch.randint(0, 4, (8,), generator=gg).to(dev)
        losses.append(ae.step(x, c).item())
    layers.check_status(dev)
    res.append((losses, enc.flat_params()[0].clone(), dec.flat_params()[0].clone(), sum(len(v['graphs']) for v in ae._graphs.values())))
(la, ea, da, na), (lb, eb, db, nb) = res
assert na == 6 and nb == 1, (na, nb)
assert la == lb, (la, lb)
assert torch.equal(ea, eb) and torch.equal(da, db), 'the multi-graph step over RCCL differs from the single-rank graph step'
print('RESULT ' + json.dumps({'losses': la, 'graphs': [na, nb]}), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def test_config3_four_graph_step_over_rccl_one_rank(dev, tmp_path):
    """The multi-rank product path on the real communication backend: ONE child process (fresh: RCCL is initialised before any
    other GPU work there), WORLD_SIZE=1 + ZS_FORCE_MULTI=1, backend 'nccl' (= RCCL).  The six-graph step with the five bucketed
    all-reduces issued between the graphs equals the single-rank one-graph step bit for bit over 6 replayed steps (a one-rank
    all-reduce is the identity and 1/world = 1), dropout and Gumbel noise on."""
    import subprocess
    script = str(tmp_path / 'rccl_one_rank.py')
    open(script, 'w').write(_RCCL_SCRIPT % {'root': ROOT})
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
               ZS_FORCE_MULTI='1', OMP_NUM_THREADS='2')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    pr = subprocess.run([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=420)
    assert pr.returncode == 0, pr.stdout[-4000:]
    line = [l for l in pr.stdout.splitlines() if l.startswith('RESULT ')]
    assert line, pr.stdout[-2000:]
    res = json.loads(line[0][7:])
    assert res['graphs'] == [6, 1] and len(set(res['losses'])) == len(res['losses'])


def test_bench_json_line_is_clean_over_rccl(dev):
    """bench.py under the multi-rank code path on RCCL (one rank, ZS_FORCE_MULTI=1): stdout carries exactly ONE line and it is
    the JSON record (RCCL prints a banner on its first collective; bench.claim_stdout keeps it off stdout), four graph segments."""
    import subprocess
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
               ZS_FORCE_MULTI='1', OMP_NUM_THREADS='2')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    pr = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--batch', '32',
                         '--no-cpu-baseline', '--no-secondary'], env=env, capture_output=True, text=True, timeout=420)
    assert pr.returncode == 0, pr.stderr[-4000:]
    lines = [l for l in pr.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, pr.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 1 and rec['graph_segments'] == 6 and rec['hipgraph'] is True and rec['value'] > 0


def test_host_fed_step_equals_resident_step(dev):
    """trainer.HostFedStep (H2D copy of the next batch as a branch of the captured step): distinct host batches,
    no host synchronisation between steps -- the loss trajectory and the parameters equal those of the same batches fed from
    device memory through AEStep.step, bit for bit, and every batch is consumed exactly once in order."""
    from zs_amd.model import Decoder, Encoder
    from zs_amd.trainer import AEStep

    class Loader(object):
        def __init__(self):
            self.g = torch.Generator().manual_seed(3)
            self.n = 0

        def __next__(self):
            self.n += 1
            return torch.randint(0, 4, (8,), generator=self.g), torch.rand(8, 128, 80, generator=self.g)

    res = []
    for mode in ('host', 'resident'):
        torch.manual_seed(0)
        enc = Encoder(c_in=80, c_h1=16, c_h2=64, c_h3=32, ns=0.01, dp=0.5, enc_size=32, seg_len=128, enc_mode='multilabel_binary', dtype='bf16').to(dev)
        dec = Decoder(c_in=32, c_out=80, c_h=64, c_a=4, ns=0.01, seg_len=128, dtype='bf16').to(dev)
        ae = AEStep(enc, dec, lr=1e-3, max_grad_norm=5.0, use_graph=True)
        ld = Loader()
        losses = []
        if mode == 'host':
            feeder = ae.host_feeder(ld)
            for _ in range(9):
                losses.append(next(feeder))
            losses = [l.clone() for l in losses[-1:]] and losses
            torch.cuda.synchronize()
            final = ae._loss.item()
        else:
            for _ in range(9):
                c, x = next(ld)
                ae.step(x.to(dev), c.to(dev))
            torch.cuda.synchronize()
            final = ae._loss.item()
        res.append((final, enc.flat_params()[0].clone(), dec.flat_params()[0].clone()))
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_thin_emb5_gradient_equals_the_literal_block(dev, monkeypatch, dtype):
    """ZS_THIN_EMB5 (default on where c_h is a multiple of 128): the input gradient of dense5's append_emb block as
    (sum_t dz5[b,t]) W5e -- column sums of the STORED dz5 from the epilogue of the GEMM that produces it (ZsGemmConv.colsum_post),
    times the block as a B-row GEMM -- against computing and column-summing the whole third block: every decoder gradient."""
    from zs_amd import _lib as L
    from zs_amd.layers import join_side
    from zs_amd.model import Decoder
    g = torch.Generator().manual_seed(9)
    B, ch = 5, 128
    bits = (torch.rand(B, 8, 16, generator=g) > 0.5).float()
    cidx = torch.randint(0, 4, (B,), generator=g)
    dl = torch.randn(B, 128, 80, generator=g) * 1e-3
    res = []
    for thin in ('0', '1'):
        monkeypatch.setenv('ZS_THIN_EMB5', thin)
        torch.manual_seed(1)
        dec = Decoder(c_in=8, c_out=80, c_h=ch, c_a=4, ns=0.01, seg_len=128, dtype=dtype).to(dev)
        dec.train()
        eng = dec._engine()
        xd = dec(bits.to(dev), cidx.to(dev))
        dlog = eng.ctx.act('t_dl', B, 128, 80)
        L.call('zs_cast_rows', 'ZsCastRows', eng.ctx.stream, dtype=eng.ctx.dt, src=L.ptr(dl.to(dev).contiguous()), ld_src=80, src_f32=1,
               dst=dlog.ptr(), ld_dst=dlog.ld, dst_f32=0, col_off=0, rows=B * 128, cols=80, fill_cols=dlog.ld, act=L.ZS_ACT_NONE)
        eng.backward(dlog)
        join_side(dev)
        torch.cuda.synchronize()
        res.append((xd.clone(), {k: dec.grad_view(k).clone() for k, _ in dec.named_parameters()}))
    tol = 2e-5 if dtype == 'fp32' else 1.5e-2
    assert torch.equal(res[0][0], res[1][0])
    for k, ref in res[0][1].items():
        scale = ref.abs().max().item()
        if scale > 1e-8:
            assert (res[1][1][k] - ref).abs().max().item() / scale < tol, k
