"""Ragged inference batches (per-sample lengths) through the drop-in modules and convert.encode_batch.

The reference forwards the fragments of an utterance one at a time, batch 1 (convert.py:70-76, 151-168): full seg_len pieces and
a 128..254-frame tail `spec[idx:-1]`.  The build runs all tails as ONE batch padded to the longest with a `lengths[B]` operand;
each sample must come out exactly as if it had been forwarded alone:
  * against the reference's own golden vectors (tests/golden/infer_*.npz, T in {9, 10, 16, 24, 127, 129, 201}) with ALL lengths
    in one ragged batch: logits / x_dec 1e-3, bits identical up to near-ties (the same bounds as the per-length golden test);
  * against the product's own fragment-by-fragment forward given the same Gumbel noise: fp32 bit for bit (same kernels, same
    summation order), bf16 to rounding (the batched GRU may dispatch to another kernel than the batch-1 one);
  * encode_batch on more seg_len fragments than one chunk holds (> 2 x max_batch), no GRU timeout."""
import numpy as np
import pytest
import torch

from conftest import load_golden, sub_sd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import zs_amd  # noqa: F401
    return torch.device('cuda:0')


def _build(m, dtype, dev, d):
    from zs_amd.model import Decoder, Encoder
    enc = Encoder(c_in=m['c_in'], c_h1=m['c_h1'], c_h2=m['c_h2'], c_h3=m['c_h3'], ns=m['ns'], dp=m.get('dp', 0.0),
                  enc_size=m['enc_size'], seg_len=m['seg_len'], enc_mode='multilabel_binary', dtype=dtype).to(dev)
    dec = Decoder(c_in=m['enc_size'], c_out=m['c_in'], c_h=m['c_h'], c_a=m['n_spk'], ns=m['ns'], seg_len=m['seg_len'], dtype=dtype).to(dev)
    enc.load_state_dict(sub_sd(d, 'enc.'))
    dec.load_state_dict(sub_sd(d, 'dec.'))
    return enc.eval(), dec.eval()


def _rel(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return (a - b).abs().max().item() / max(1e-6, b.abs().max().item())


def _t8(n):
    return ((((n + 1) // 2 + 1) // 2) + 1) // 2


@pytest.mark.parametrize('name', ['infer_f80.npz', 'infer_f513.npz'])
def test_ragged_batch_vs_reference_golden(dev, name):
    import zs_oracle as O
    d, m = load_golden(name)
    enc, dec = _build(m, 'fp32', dev, d)
    Ts = list(m['lengths'])
    Tm, F, E = max(Ts), m['c_in'], m['enc_size']
    xs, cs, Gs, lens, owner = [], [], [], [], []
    for T in Ts:
        x = torch.from_numpy(d['x.%d' % T])                                  # [B, F, T]
        G = O.gumbel_from_uniform(torch.from_numpy(d['U.%d' % T]))           # [B, T', E, 2]
        for b in range(x.shape[0]):
            xp = torch.zeros(F, Tm); xp[:, :T] = x[b]
            Gp = torch.zeros(_t8(Tm), E, 2); Gp[:_t8(T)] = G[b]
            xs.append(xp); Gs.append(Gp); cs.append(int(d['c.%d' % T][b])); lens.append(T); owner.append((T, b))
    X, Gall, C = torch.stack(xs).to(dev), torch.stack(Gs).to(dev), torch.tensor(cs, device=dev)
    act, logits = enc(X, G=Gall, lengths=lens)
    # decoder on the REFERENCE's bits (as the per-length golden test does), all lengths in one ragged batch
    ref_bits = torch.zeros(len(lens), E, _t8(Tm))
    for i, (T, b) in enumerate(owner):
        ref_bits[i, :, :_t8(T)] = torch.from_numpy(d['enc_act.%d' % T][b])
    xdec = dec(ref_bits.to(dev), C, lengths=[_t8(T) for T in lens])
    total = flips_total = 0
    for i, (T, b) in enumerate(owner):
        tp = _t8(T)
        ref_logits, ref_act = torch.from_numpy(d['enc.%d' % T][b]), torch.from_numpy(d['enc_act.%d' % T][b])
        e = _rel(logits[i, :, :tp], ref_logits)
        assert e < 1e-3, 'T=%d sample %d: logits rel err %.3g' % (T, b, e)
        flips = (act[i, :, :tp].cpu() != ref_act)
        total += flips.numel(); flips_total += int(flips.sum())
        if flips.any():
            s = ref_logits.t().reshape(tp, E, 2) + Gs[i][:tp]
            margin = (s[..., 0] - s[..., 1]).abs().t()[flips]
            bound = 4 * (logits[i, :, :tp].cpu() - ref_logits).abs().max().item() + 1e-6
            assert margin.max().item() <= bound, 'bit flip at margin %.3g > logit error bound %.3g' % (margin.max().item(), bound)
        e = _rel(xdec[i, :, :8 * tp], d['x_dec.%d' % T][b])
        assert e < 1e-3, 'T=%d sample %d: x_dec rel err %.3g' % (T, b, e)
    print('%s as ONE ragged batch of %d samples (lengths %s): %d/%d MBV bits differ from the reference (near-ties only)' %
          (name, len(lens), Ts, flips_total, total))
    assert flips_total <= max(1, total // 200)
    from zs_amd import layers
    layers.check_status(dev)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_ragged_batch_equals_fragment_by_fragment(dev, dtype):
    """Tail-fragment lengths 128..254 (+ a short utterance) through Trainer-sized layer types at a reduced width."""
    from zs_amd import layers
    from zs_amd.model import Decoder, Encoder
    torch.manual_seed(0)
    E, ch, nspk = 32, 64, 4
    enc = Encoder(c_in=513, c_h1=16, c_h2=64, c_h3=128, ns=0.01, dp=0.5, enc_size=E, seg_len=128, enc_mode='multilabel_binary', dtype=dtype).to(dev).eval()
    dec = Decoder(c_in=E, c_out=513, c_h=ch, c_a=nspk, ns=0.01, seg_len=128, dtype=dtype).to(dev).eval()
    lens = [254, 129, 201, 128, 9, 177, 253, 130, 64, 255 - 1, 202]
    Tm = max(lens)
    g = torch.Generator().manual_seed(4)
    X = torch.zeros(len(lens), 513, Tm)
    for i, n in enumerate(lens):
        X[i, :, :n] = torch.rand(513, n, generator=g)
    c = torch.randint(0, nspk, (len(lens),), generator=g)
    U = torch.rand(len(lens), _t8(Tm), E, 2, generator=g)
    G = -torch.log(-torch.log(U + 1e-20) + 1e-20)
    act, logits = enc(X.to(dev), G=G.to(dev), lengths=lens)
    xdec = dec(act, c.to(dev), lengths=[_t8(n) for n in lens])
    worst = [0.0, 0.0]
    exact = True
    for i, n in enumerate(lens):
        tp = _t8(n)
        a1, l1 = enc(X[i:i + 1, :, :n].to(dev), G=G[i:i + 1, :tp].contiguous().to(dev))
        x1 = dec(a1, c[i:i + 1].to(dev))
        assert a1.shape[2] == tp and x1.shape[2] == 8 * tp
        same = torch.equal(logits[i, :, :tp], l1[0]) and torch.equal(act[i, :, :tp], a1[0]) and torch.equal(xdec[i, :, :8 * tp], x1[0])
        exact &= same
        worst[0] = max(worst[0], _rel(logits[i, :, :tp], l1[0])); worst[1] = max(worst[1], _rel(xdec[i, :, :8 * tp], x1[0]))
        if dtype == 'fp32':
            assert same, 'sample %d (length %d): the ragged batch differs from the fragment forwarded alone' % (i, n)
    layers.check_status(dev)
    print('%s: ragged batch vs fragment-by-fragment: bitwise equal %s, worst rel diff logits %.3g x_dec %.3g' % (dtype, exact, worst[0], worst[1]))
    # bf16: the batch-11 and batch-1 GRU recurrences may run different kernels (accumulation order); a flipped bit changes x_dec
    assert worst[0] < 2e-2


def test_encode_batch_many_fragments_and_lengths(dev, tmp_path):
    """convert.encode_batch on more full fragments than two chunks hold (> 512 seg_len fragments) plus every kind of
    ragged member (short utterances, padded-to-MIN_LEN, tails): shapes follow the fragment rule; for sampled utterances the
    decoded spectrogram equals the Decoder applied fragment by fragment to the bits that call returned."""
    import zs_oracle as O
    from zs_amd import convert as cv, layers
    from zs_amd.hps import make_hps
    from zs_amd.trainer import Trainer
    torch.manual_seed(2)
    hps = make_hps(enc_size=16, emb_size=32, n_speakers=4, n_target_speakers=2)
    tr = Trainer(hps, None, 'targeted_residual', 'multilabel_binary', log_dir=str(tmp_path / 'log'), dtype='fp32', device=dev)
    rng = np.random.RandomState(3)
    lens = [640] * 140 + [5, 9, 100, 128, 129, 255, 256, 257, 300, 383, 384]
    specs = [np.clip(rng.rand(n, 513).astype(np.float32), 1e-8, 1) for n in lens]
    spk = [int(rng.randint(0, 4)) for _ in specs]
    encs, decs = cv.encode_batch(specs, tr, 128, decode_speakers=spk, max_batch=256)
    n_full = sum(sum(1 for a, b in O.fragment_plan(n, 128)[1] if b - a == 128) for n in lens)
    assert n_full > 512
    for n, e, d in zip(lens, encs, decs):
        _, frags, trunc = O.fragment_plan(n, 128)
        t_out = sum(O.out_len(b - a) for a, b in frags)
        assert d.shape == (t_out, 513), (n, d.shape, t_out)
        assert e.shape == (trunc if trunc is not None else t_out // 8, 16) and set(np.unique(e)) <= {0.0, 1.0}
    for u in (0, 139, 141, 144, 145, 148, 150):
        n = lens[u]
        frags = O.fragment_plan(n, 128)[1]
        o8 = o = 0
        for a, b in frags:
            tp = _t8(b - a)
            bits = torch.from_numpy(encs[u][o8:o8 + tp].T[None].copy()).to(dev)
            xd = tr.Decoder(bits, torch.tensor([spk[u]], device=dev))[0].T.cpu().numpy()
            assert np.abs(xd - decs[u][o:o + 8 * tp]).max() < 1e-6, (u, a, b)
            o8 += tp; o += 8 * tp
    layers.check_status(dev)


def test_full_size_inference_vs_oracle(dev):
    """The benchmarked inference model at FULL width (english hps, enc_size = emb_size = 1024, 102 speakers) on a ragged batch of
    tail-like fragments: fp32 HIP path against the oracle (CPU restatement of the reference, pinned by the goldens) -- encoder
    logits and x_dec 1e-3 of scale, bits identical up to near-ties; bf16 (the dtype bench.py --mode resynth runs) against the fp32
    HIP path on the same weights / noise with measured, stated bounds (bit mismatch < 3 %, x_dec given identical bits < 8 % of scale
    at the worst element, < 1 % rms)."""
    import zs_oracle as O
    from zs_amd import layers
    from zs_amd.model import Decoder, Encoder
    torch.manual_seed(0)
    E, ch, nspk = 1024, 1024, 102
    enc = Encoder(ns=0.01, dp=0.5, enc_size=E, seg_len=128, enc_mode='multilabel_binary', dtype='fp32').to(dev).eval()
    dec = Decoder(ns=0.01, c_in=E, c_h=ch, c_a=nspk, seg_len=128, dtype='fp32').to(dev).eval()
    lens = [201, 128, 254, 9]
    Tm = max(lens)
    g = torch.Generator().manual_seed(6)
    X = torch.zeros(len(lens), 513, Tm)
    for i, n in enumerate(lens):
        X[i, :, :n] = torch.rand(513, n, generator=g) * (1 - 1e-8) + 1e-8
    c = torch.randint(0, nspk, (len(lens),), generator=g)
    U = torch.rand(len(lens), _t8(Tm), E, 2, generator=g)
    G = O.gumbel_from_uniform(U)
    act, logits = enc(X.to(dev), G=G.to(dev), lengths=lens)
    xdec = dec(act, c.to(dev), lengths=[_t8(n) for n in lens])
    esd = {k: v.detach().cpu() for k, v in enc.state_dict().items()}
    dsd = {k: v.detach().cpu() for k, v in dec.state_dict().items()}
    total = flips_total = 0
    for i, n in enumerate(lens):
        tp = _t8(n)
        with torch.no_grad():
            o_act, o_logits = O.encoder_forward(esd, X[i:i + 1, :, :n], 0.01, 0.5, E, 128, G=G[i:i + 1, :tp])
            o_dec = O.decoder_forward(dsd, act[i:i + 1, :, :tp].cpu(), c[i:i + 1], 0.01, 128)
        e1, e2 = _rel(logits[i, :, :tp], o_logits[0]), _rel(xdec[i, :, :8 * tp], o_dec[0])
        assert e1 < 1e-3 and e2 < 1e-3, (n, e1, e2)
        flips = (act[i, :, :tp].cpu() != o_act[0])
        total += flips.numel(); flips_total += int(flips.sum())
        if flips.any():
            s = o_logits[0].t().reshape(tp, E, 2) + G[i, :tp]
            margin = (s[..., 0] - s[..., 1]).abs().t()[flips]
            assert margin.max().item() <= 4 * (logits[i, :, :tp].cpu() - o_logits[0]).abs().max().item() + 1e-6
    assert flips_total <= max(1, total // 200)
    # the same weights in bf16 (what the resynthesis bench runs)
    enc16 = Encoder(ns=0.01, dp=0.5, enc_size=E, seg_len=128, enc_mode='multilabel_binary', dtype='bf16').to(dev).eval()
    dec16 = Decoder(ns=0.01, c_in=E, c_h=ch, c_a=nspk, seg_len=128, dtype='bf16').to(dev).eval()
    enc16.load_state_dict(enc.state_dict()); dec16.load_state_dict(dec.state_dict())
    act16, logits16 = enc16(X.to(dev), G=G.to(dev), lengths=lens)
    xdec16 = dec16(act, c.to(dev), lengths=[_t8(n) for n in lens])            # same bits as the fp32 path
    mism = worst = 0.0
    num = den = 0.0
    for i, n in enumerate(lens):
        tp = _t8(n)
        mism += float((act16[i, :, :tp] != act[i, :, :tp]).float().sum())
        d = (xdec16[i, :, :8 * tp] - xdec[i, :, :8 * tp]).float()
        worst = max(worst, float(d.abs().max() / xdec[i, :, :8 * tp].abs().max()))
        num += float((d ** 2).sum()); den += float((xdec[i, :, :8 * tp].float() ** 2).sum())
    rate = mism / sum(_t8(n) * E for n in lens)
    layers.check_status(dev)
    print('full-size ragged inference: fp32 vs oracle: %d/%d bits differ; bf16 vs fp32 path: bit mismatch %.3g, x_dec (same bits) worst %.3g, rms %.3g' %
          (flips_total, total, rate, worst, (num / den) ** 0.5))
    assert rate < 0.03 and worst < 0.08 and (num / den) ** 0.5 < 0.01
