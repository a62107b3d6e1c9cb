"""CPU oracle for the ASR-TTS autoencoder hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain fp32 PyTorch-CPU / NumPy restatement of the reference's
algorithm for the path named by BASELINE.json (`--train_ae`, encode/decode
inference, Griffin-Lim resynthesis).  It is the *checker*: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.
The product path (`zs_amd`) never imports anything from `oracle/`.

Pinning: every network function here is asserted equal to the reference
itself (imported from /root/reference in the build container) by
`oracle/make_golden.py`, which also freezes the input/output vectors under
`tests/golden/`.  `tests/test_oracle_golden.py` re-checks the oracle against
those vectors on every run.  The Griffin-Lim / STFT part restates librosa
(<= 0.7 semantics; librosa is NOT installed and its version is unpinned by the
reference) and is therefore "parity unpinned" against librosa itself; it is
cross-checked against torch.stft/istft and the reference's wav sample-count
law (docs/exp/**.wav) only.

All tensors are channels-first [B, C, T] exactly as in the reference.
Citations are into /root/reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# helpers  (model/model.py:20-110)
# --------------------------------------------------------------------------


def pad_amounts(kernel_size):
    """model/model.py:26-29 -- even kernels pad (k//2, k//2-1), odd (k//2, k//2)."""
    if kernel_size % 2 == 0:
        return kernel_size // 2, kernel_size // 2 - 1
    return kernel_size // 2, kernel_size // 2


def pad_conv1d(x, w, b, seg_len, stride=1):
    """pad_layer() for a Conv1d, model/model.py:20-40.  Padding mode depends on the
    *constructor* seg_len (reflect when >= 64), not on the runtime length."""
    k = w.shape[2]
    mode = 'constant' if seg_len < 64 else 'reflect'
    xp = F.pad(x, pad=pad_amounts(k), mode=mode)
    return F.conv1d(xp, w, b, stride=stride)


def pixel_shuffle_1d(x, r=2):
    """model/model.py:43-51: out[b, c, r*w + i] = in[b, r*c + i, w]."""
    b, c, w = x.shape
    c //= r
    return x.contiguous().view(b, c, r, w).permute(0, 1, 3, 2).contiguous().view(b, c, w * r)


def upsample_nearest2(x):
    """model/model.py:54-56."""
    return x.repeat_interleave(2, dim=2)


def linear_t(x, w, b):
    """time-distributed Linear, model/model.py:69-78.  x [B,C,T] -> [B,C',T]."""
    return (x.permute(0, 2, 1) @ w.t() + b).permute(0, 2, 1)


def instance_norm(x, eps=1e-5):
    """nn.InstanceNorm1d defaults: no affine, biased variance, eps 1e-5.  The result is made
    contiguous like F.instance_norm's, so a following F.dropout consumes the RNG stream in the same
    element order as the reference does."""
    mean = x.mean(dim=2, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=2, keepdim=True)
    return ((x - mean) / torch.sqrt(var + eps)).contiguous()


def dropout(x, p, training, mask=None):
    """nn.Dropout.  `mask` (0/1 keep mask) makes it deterministic for tests."""
    if not training or p == 0.0:
        return x
    if mask is not None:
        return x * mask / (1.0 - p)
    return F.dropout(x, p=p, training=True)


def gru_direction(x_tbc, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of nn.GRU (PyTorch gate order r,z,n; zero initial state;
    model/model.py:59-66).  x [T,B,C] -> [T,B,H]."""
    T, B, _ = x_tbc.shape
    H = w_hh.shape[1]
    h = x_tbc.new_zeros(B, H)
    gi_all = x_tbc @ w_ih.t() + b_ih
    outs = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        gi = gi_all[t]
        gh = h @ w_hh.t() + b_hh
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1.0 - z) * n + z * h
        outs[t] = h
    return torch.stack(outs, dim=0)


def bigru(x_bct, sd, prefix):
    """RNN() wrapper, model/model.py:59-66: [B,C,T] -> [B,2H,T] (fwd ++ bwd)."""
    x = x_bct.permute(2, 0, 1)
    f = gru_direction(x, sd[prefix + 'weight_ih_l0'], sd[prefix + 'weight_hh_l0'],
                      sd[prefix + 'bias_ih_l0'], sd[prefix + 'bias_hh_l0'], False)
    r = gru_direction(x, sd[prefix + 'weight_ih_l0_reverse'], sd[prefix + 'weight_hh_l0_reverse'],
                      sd[prefix + 'bias_ih_l0_reverse'], sd[prefix + 'bias_hh_l0_reverse'], True)
    return torch.cat([f, r], dim=2).permute(1, 2, 0)


def gumbel_from_uniform(U, eps=1e-20):
    """_sample_gumbel, model/model.py:95-98."""
    return -torch.log(-torch.log(U + eps) + eps)


def gumbel_softmax_hard(logits, G, temperature=0.1):
    """gumbel_softmax, model/model.py:93-110, with the Gumbel noise G explicit.
    Returns (straight-through output, soft y).  argmax ties -> first index."""
    y = F.softmax((logits + G) / temperature, dim=-1)
    ind = y.max(dim=-1)[1]
    y_hard = torch.zeros_like(y).scatter_(-1, ind.unsqueeze(-1), 1.0)
    return (y_hard - y).detach() + y, y


def mbv(enc_logits_bct, enc_size, U=None, G=None):
    """multilabel_binary branch, model/model.py:474-480.
    enc_logits [B,2E,T'] -> enc_act [B,E,T'] in {0,1}."""
    proj = enc_logits_bct.permute(0, 2, 1)
    proj = proj.reshape(proj.size(0), proj.size(1), enc_size, 2)
    if G is None:
        if U is None:
            U = torch.rand(proj.shape)
        G = gumbel_from_uniform(U)
    out, _ = gumbel_softmax_hard(proj, G)
    act = out[:, :, :, 0]
    return act.permute(0, 2, 1).contiguous()


# --------------------------------------------------------------------------
# Encoder  (model/model.py:368-489)
# --------------------------------------------------------------------------


def encoder_forward(sd, x, ns, dp, enc_size, seg_len, U=None, G=None, training=False,
                    drop_masks=None):
    """Encoder.forward, enc_mode='multilabel_binary'.  x [B,c_in,T].
    Returns (enc_act [B,E,T/8], enc [B,2E,T/8])."""
    dm = drop_masks if drop_masks is not None else [None] * 6

    def lrelu(v):
        return F.leaky_relu(v, negative_slope=ns)

    outs = [pad_conv1d(x, sd['conv1s.%d.weight' % i], sd['conv1s.%d.bias' % i], seg_len) for i in range(7)]
    out = lrelu(torch.cat(outs + [x], dim=1))                                   # :441-446
    # conv_block([conv2], res=False)                                            # :447
    out = lrelu(pad_conv1d(out, sd['conv2.weight'], sd['conv2.bias'], seg_len))
    out = dropout(instance_norm(out), dp, training, dm[0])
    # three strided blocks with avg-pool residual                                # :416-427, :448-450
    for bi, (ca, cb) in enumerate([(3, 4), (5, 6), (7, 8)]):
        xin = out
        out = lrelu(pad_conv1d(xin, sd['conv%d.weight' % ca], sd['conv%d.bias' % ca], seg_len))
        out = lrelu(pad_conv1d(out, sd['conv%d.weight' % cb], sd['conv%d.bias' % cb], seg_len, stride=2))
        out = dropout(instance_norm(out), dp, training, dm[1 + bi])
        x_pad = F.pad(xin, pad=(0, xin.size(2) % 2), mode='constant' if seg_len < 64 else 'reflect')
        out = F.avg_pool1d(x_pad, kernel_size=2) + out
    # two dense blocks                                                           # :429-438, :452-453
    for bi, (da, db) in enumerate([(1, 2), (3, 4)]):
        xin = out
        out = lrelu(linear_t(xin, sd['dense%d.weight' % da], sd['dense%d.bias' % da]))
        out = lrelu(linear_t(out, sd['dense%d.weight' % db], sd['dense%d.bias' % db]))
        out = dropout(instance_norm(out), dp, training, dm[4 + bi])
        out = out + xin
    out = torch.cat([out, bigru(out, sd, 'RNN.')], dim=1)                       # :454-455
    enc = linear_t(out, sd['linear.weight'], sd['linear.bias'])                 # :475
    return mbv(enc, enc_size, U=U, G=G), enc


# --------------------------------------------------------------------------
# Decoder  (model/model.py:283-365)
# --------------------------------------------------------------------------


def decoder_forward(sd, enc_act, c, ns, seg_len, output_mask=False):
    """Decoder.forward.  enc_act [B,E,T'], c int64 [B] -> [B,c_out,8T']."""

    def lrelu(v):
        return F.leaky_relu(v, negative_slope=ns)

    def emb(i):
        return sd['emb%d.weight' % i][c].unsqueeze(2)

    out = linear_t(enc_act, sd['input_emb.weight'], sd['input_emb.bias'])       # :346
    for bi, (ca, cb) in enumerate([(1, 2), (3, 4), (5, 6)]):                      # :317-331
        e = emb(bi + 1)
        xin = out
        out = lrelu(pad_conv1d(xin + e, sd['conv%d.weight' % ca], sd['conv%d.bias' % ca], seg_len))
        out = pixel_shuffle_1d(out) + e
        out = lrelu(pad_conv1d(out, sd['conv%d.weight' % cb], sd['conv%d.bias' % cb], seg_len))
        out = instance_norm(out) + upsample_nearest2(xin)
    e4 = emb(4)                                                                  # :350-351 (emb4 twice)
    for (da, db) in [(1, 2), (3, 4)]:
        xin = out
        out = lrelu(linear_t(xin + e4, sd['dense%d.weight' % da], sd['dense%d.bias' % da]))
        out = lrelu(linear_t(out + e4, sd['dense%d.weight' % db], sd['dense%d.bias' % db]))
        out = instance_norm(out) + xin
    e5 = emb(5)
    rnn = bigru(out + e5, sd, 'RNN.')                                            # :352-356
    out = torch.cat([out, rnn, e5.expand(-1, -1, out.size(2))], dim=1)           # :81-85, :357
    out = lrelu(linear_t(out, sd['dense5.weight'], sd['dense5.bias']))
    out = linear_t(out, sd['linear.weight'], sd['linear.bias'])
    return torch.tanh(out) if output_mask else torch.sigmoid(out)               # :361-364


# --------------------------------------------------------------------------
# SpeakerClassifier  (model/model.py:231-280) + CE / accuracy (trainer.py:297-313)
# --------------------------------------------------------------------------


def speaker_classifier_forward(sd, x, ns, dp, seg_len, training=False, drop_masks=None):
    dm = drop_masks if drop_masks is not None else [None] * 4

    def lrelu(v):
        return F.leaky_relu(v, negative_slope=ns)

    out = x
    for bi, (ca, cb, res) in enumerate([(1, 2, False), (3, 4, True), (5, 6, True), (7, 8, False)]):
        xin = out
        out = lrelu(pad_conv1d(xin, sd['conv%d.weight' % ca], sd['conv%d.bias' % ca], seg_len))
        out = lrelu(pad_conv1d(out, sd['conv%d.weight' % cb], sd['conv%d.bias' % cb], seg_len))
        out = dropout(instance_norm(out), dp, training, dm[bi])
        if res:
            out = out + xin
    out = F.conv1d(out, sd['conv9.weight'], sd['conv9.bias'])                   # :278 (no padding)
    return out.view(out.size(0), -1)


def cross_entropy(logits, y):
    """nn.CrossEntropyLoss (mean), trainer.py:297-304."""
    return F.cross_entropy(logits, y)


# --------------------------------------------------------------------------
# Stage 2: PatchDiscriminator / TargetClassifier (model/model.py:113-228), the WGAN-GP
# gradient penalty (utils.py:58-77) and the patchGAN losses (trainer.py:257-263, 467-560)
# --------------------------------------------------------------------------


def pad_conv2d(x, w, b, seg_len, stride=1):
    """pad_layer(..., is_2d=True), model/model.py:29-39: (k//2, k//2 | k//2 - 1) on both spatial axes, reflect when the
    constructor seg_len >= 64."""
    k = w.shape[2]
    pl, pr = pad_amounts(k)
    xp = F.pad(x, pad=(pl, pr, pl, pr), mode='constant' if seg_len < 64 else 'reflect')
    return F.conv2d(xp, w, b, stride=stride)


def instance_norm2d(x, eps=1e-5):
    """nn.InstanceNorm2d defaults (no affine, biased variance) over (H, W) per (b, c)."""
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(2, 3), keepdim=True)
    return (x - mean) / torch.sqrt(var + eps)


def dropout2d(x, p, training, mask=None):
    """nn.Dropout2d: whole (b, c) feature maps are zeroed.  `mask` [B, C] (0/1 keep) makes it deterministic."""
    if not training or p == 0.0:
        return x
    if mask is not None:
        return x * mask[:, :, None, None] / (1.0 - p)
    return F.dropout2d(x, p=p, training=True)


def patch_discriminator_forward(sd, x, ns, seg_len, classify=False, dp=0.1, training=True, drop_masks=None, prefix=''):
    """PatchDiscriminator.forward (model/model.py:155-173) / TargetClassifier.forward (:218-228, classify only).
    x [B, 513, T] -> mean_val [B] (, logits [B, n_class]).  `prefix` 'module.' for DataParallel-wrapped state dicts."""
    dm = drop_masks if drop_masks is not None else [None] * 6
    out = x.unsqueeze(1)
    for i in range(1, 7):
        w, b = sd['%sconv%d.weight' % (prefix, i)], sd['%sconv%d.bias' % (prefix, i)]
        out = F.leaky_relu(pad_conv2d(out, w, b, seg_len, stride=2 if i <= 5 else 1), negative_slope=ns)     # conv_block :148-153
        out = dropout2d(instance_norm2d(out), dp, training, dm[i - 1])
    mean_val = None
    if (prefix + 'conv7.weight') in sd:
        val = F.conv2d(out, sd[prefix + 'conv7.weight'], sd[prefix + 'conv7.bias'])
        mean_val = val.view(val.size(0), -1).mean(dim=1)
    if classify:
        logits = F.conv2d(out, sd[prefix + 'conv_classify.weight'], sd[prefix + 'conv_classify.bias'])
        return mean_val, logits.view(logits.size(0), -1)
    return mean_val


def synthetic_patch_sd(n_class, seed, seg_len=128, prefix=''):
    """Deterministic PatchDiscriminator parameters from numpy's frozen legacy generator (model/model.py:114-131 shapes):
    the 10.9 M weights of a stage-2 golden vector are regenerated from (n_class, seed) instead of being stored."""
    rng = np.random.RandomState(seed)
    kt = {128: 4, 64: 2, 32: 1}[seg_len]
    shapes = [('conv1', (64, 1, 5, 5)), ('conv2', (128, 64, 5, 5)), ('conv3', (256, 128, 5, 5)), ('conv4', (512, 256, 5, 5)),
              ('conv5', (512, 512, 5, 5)), ('conv6', (32, 512, 1, 1)), ('conv7', (1, 32, 17, kt)), ('conv_classify', (n_class, 32, 17, kt))]
    sd = {}
    for name, shp in shapes:
        fan_in = shp[1] * shp[2] * shp[3]
        sd[prefix + name + '.weight'] = torch.from_numpy((rng.standard_normal(shp) / math.sqrt(fan_in)).astype(np.float32))
        sd[prefix + name + '.bias'] = torch.from_numpy((rng.standard_normal(shp[0]) * 0.1).astype(np.float32))
    return sd


def gradients_penalty(sd, real, fake, alpha, ns, seg_len, dp=0.1, training=True, drop_masks=None, prefix=''):
    """utils.calculate_gradients_penalty (utils.py:58-77) with the interpolation weights `alpha` [B] given (the reference draws
    torch.rand(B)).  Differentiable w.r.t. the parameters in `sd` (create_graph)."""
    a = alpha.view(-1, 1, 1)
    inter = (a * real + (1 - a) * fake).detach().requires_grad_(True)
    d = patch_discriminator_forward(sd, inter, ns, seg_len, dp=dp, training=training, drop_masks=drop_masks, prefix=prefix)
    g = torch.autograd.grad(outputs=d, inputs=inter, grad_outputs=torch.ones_like(d), create_graph=True, retain_graph=True,
                            only_inputs=True)[0]
    gp = (1.0 - torch.sqrt(1e-12 + torch.sum(g.view(g.size(0), -1) ** 2, dim=1))) ** 2
    return gp.mean()


def patch_d_loss(sd, x_t, x_dec, c_shifted, alpha, hp, masks=None, prefix=''):
    """The discriminator loss of trainer.py:488-494: -beta_dis * w_dis + beta_clf * CE(real_logits, c - shift) + lambda * gp.
    masks: None or three lists of six [B, C] keep masks (real, fake, interpolate passes).  Returns (loss, w_dis, loss_clf, gp, real_logits)."""
    m = masks if masks is not None else [None, None, None]
    kw = dict(ns=hp['ns'], seg_len=hp['seg_len'], dp=hp.get('dp', 0.1), training=hp.get('training', True), prefix=prefix)
    d_real, real_logits = patch_discriminator_forward(sd, x_t, classify=True, drop_masks=m[0], **kw)
    d_fake, _ = patch_discriminator_forward(sd, x_dec, classify=True, drop_masks=m[1], **kw)
    w_dis = torch.mean(d_real - d_fake)                                                  # trainer.py:261
    gp = gradients_penalty(sd, x_t, x_dec, alpha, drop_masks=m[2], **kw)
    loss_clf = cross_entropy(real_logits, c_shifted)
    loss = -hp['beta_dis'] * w_dis + hp['beta_clf'] * loss_clf + hp['lambda_'] * gp
    return loss, w_dis, loss_clf, gp, real_logits


def gen_step(dec_sd, gen_sd, enc_act, c, shift_c, ns, seg_len, g_mode='targeted_residual'):
    """Trainer.gen_step (trainer.py:266-278)."""
    x_dec = decoder_forward(dec_sd, enc_act, c, ns, seg_len)
    if g_mode == 'naive':
        return x_dec + decoder_forward(gen_sd, enc_act, c, ns, seg_len)
    if g_mode == 'targeted':
        return x_dec + decoder_forward(gen_sd, enc_act, c - shift_c, ns, seg_len)
    if g_mode == 'targeted_residual':
        return x_dec + x_dec * decoder_forward(gen_sd, enc_act, c - shift_c, ns, seg_len, output_mask=True)
    raise NotImplementedError(g_mode)


def patch_g_loss(sd, x_gen, c_shifted, hp, masks=None, prefix=''):
    """The generator loss of trainer.py:528-533: beta_clf * CE(fake_logits, c - shift) + beta_gen * (-mean D(x_gen))."""
    kw = dict(ns=hp['ns'], seg_len=hp['seg_len'], dp=hp.get('dp', 0.1), training=hp.get('training', True), prefix=prefix)
    d_fake, fake_logits = patch_discriminator_forward(sd, x_gen, classify=True, drop_masks=masks, **kw)
    loss_adv = -torch.mean(d_fake)
    loss_clf = cross_entropy(fake_logits, c_shifted)
    return hp['beta_clf'] * loss_clf + hp['beta_gen'] * loss_adv, loss_adv, loss_clf, fake_logits


# --------------------------------------------------------------------------
# train_ae step  (trainer.py:320-332, utils.py:48-55, torch.optim.Adam)
# --------------------------------------------------------------------------


def l1_loss(x_dec, x):
    """trainer.py:328."""
    return torch.mean(torch.abs(x_dec - x))


def clip_grad_norm(grads, max_norm):
    """nn.utils.clip_grad_norm_ (L2) over ONE net's grads (utils.py:53-55 calls it per net).
    Returns (total_norm, clipped grads)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, [g * coef for g in grads]


class AdamState(object):
    """torch.optim.Adam(lr, betas=(0.5, 0.9), eps=1e-8), single-tensor math."""

    def __init__(self, params, lr=1e-4, betas=(0.5, 0.9), eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, betas[0], betas[1], eps
        self.step = 0
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]

    def update(self, params, grads):
        self.step += 1
        bc1 = 1.0 - self.b1 ** self.step
        bc2 = 1.0 - self.b2 ** self.step
        out = []
        for i, (p, g) in enumerate(zip(params, grads)):
            self.m[i] = self.m[i] + (g - self.m[i]) * (1.0 - self.b1)          # lerp_
            self.v[i] = self.v[i] * self.b2 + g * g * (1.0 - self.b2)
            denom = self.v[i].sqrt() / math.sqrt(bc2) + self.eps
            out.append(p - (self.lr / bc1) * self.m[i] / denom)
        return out


def train_ae_grads(enc_sd, dec_sd, x, c, hp, U=None, G=None, drop_masks=None, training=True):
    """One forward/backward of trainer.py:326-330.  hp: dict(ns, enc_dp, enc_size, seg_len).
    Returns loss, (enc grads dict, dec grads dict), x_dec, enc_act."""
    enc_p = {k: v.detach().clone().requires_grad_(True) for k, v in enc_sd.items()}
    dec_p = {k: v.detach().clone().requires_grad_(True) for k, v in dec_sd.items()}
    enc_act, _ = encoder_forward(enc_p, x, hp['ns'], hp['enc_dp'], hp['enc_size'], hp['seg_len'],
                                 U=U, G=G, training=training, drop_masks=drop_masks)
    x_dec = decoder_forward(dec_p, enc_act, c, hp['ns'], hp['seg_len'])
    loss = l1_loss(x_dec, x)
    loss.backward()
    ge = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in enc_p.items()}
    gd = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in dec_p.items()}
    return loss.detach(), (ge, gd), x_dec.detach(), enc_act.detach()


class TrainAE(object):
    """Stateful train_ae loop: Adam over Encoder+Decoder params, per-net clip
    (trainer.py:65-66, 320-332)."""

    def __init__(self, enc_sd, dec_sd, hp, lr=1e-4, max_grad_norm=5.0):
        self.enc_sd = {k: v.clone() for k, v in enc_sd.items()}
        self.dec_sd = {k: v.clone() for k, v in dec_sd.items()}
        self.hp = hp
        self.max_grad_norm = max_grad_norm
        self.ek, self.dk = list(self.enc_sd.keys()), list(self.dec_sd.keys())
        self.adam = AdamState([self.enc_sd[k] for k in self.ek] + [self.dec_sd[k] for k in self.dk], lr=lr)

    def step(self, x, c, U=None, G=None, drop_masks=None, training=True):
        loss, (ge, gd), x_dec, enc_act = train_ae_grads(self.enc_sd, self.dec_sd, x, c, self.hp,
                                                        U=U, G=G, drop_masks=drop_masks, training=training)
        ne, gel = clip_grad_norm([ge[k] for k in self.ek], self.max_grad_norm)
        nd, gdl = clip_grad_norm([gd[k] for k in self.dk], self.max_grad_norm)
        params = [self.enc_sd[k] for k in self.ek] + [self.dec_sd[k] for k in self.dk]
        new = self.adam.update(params, gel + gdl)
        for i, k in enumerate(self.ek):
            self.enc_sd[k] = new[i]
        for i, k in enumerate(self.dk):
            self.dec_sd[k] = new[len(self.ek) + i]
        return float(loss), float(ne), float(nd), x_dec, enc_act


# --------------------------------------------------------------------------
# Inference segmenter + encodings text  (convert.py:36, 116-221)
# --------------------------------------------------------------------------

MIN_LEN = 9


def fragment_plan(n_frames, seg_len):
    """Fragment rule of convert()/encode() (convert.py:139-168, 196-214), as pure logic.
    Returns (padded_len, [(start, stop)], truncate_enc_to) where each (start, stop) is the
    python slice spec[start:stop] that is sent through the network.  A tail fragment is
    spec[idx:-1] (last frame dropped); fragments shorter than seg_len are skipped."""
    padded = max(n_frames, MIN_LEN)
    trunc = MIN_LEN // 8 if n_frames < MIN_LEN else None
    if padded <= seg_len:
        return padded, [(0, padded)], trunc
    frags = []
    for idx in range(0, padded, seg_len):
        if idx + 2 * seg_len > padded:
            start, stop = idx, padded - 1
        else:
            start, stop = idx, idx + seg_len
        if stop - start >= seg_len:
            frags.append((start, stop))
        elif idx == 0:
            raise RuntimeError('Please check if input is too short!')
    return padded, frags, None


def encodings_text(encodings):
    """write_encodings(), convert.py:120-125: one line per encoded frame, ints separated by ' '."""
    lines = []
    for enc in encodings:
        lines.append(' '.join(str(int(e)) for e in enc) + '\n')
    return ''.join(lines)


def out_len(t):
    """Decoder output frames for a t-frame fragment: 8*ceil(ceil(ceil(t/2)/2)/2)."""
    for _ in range(3):
        t = (t + 1) // 2
    return 8 * t


# --------------------------------------------------------------------------
# Vocoder: Griffin-Lim + spectrogram2wav  (convert.py:39-62; librosa <=0.7 restated)
# --------------------------------------------------------------------------

SR, N_FFT, HOP, WIN = 16000, 1024, 200, 800          # hps/hps.py:22-28
N_ITER, PREEMPH, MAX_DB, REF_DB = 300, 0.97, 100, 20  # hps/hps.py:31-34


def hann_padded():
    """scipy.signal.get_window('hann', 800, fftbins=True) centre-padded to n_fft
    (librosa.util.pad_center), as librosa.stft/istft do for win_length < n_fft."""
    n = np.arange(WIN, dtype=np.float64)
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / WIN)
    lpad = (N_FFT - WIN) // 2
    out = np.zeros(N_FFT, dtype=np.float64)
    out[lpad:lpad + WIN] = w
    return out.astype(np.float32)


def stft(y):
    """librosa.stft(y, 1024, 200, win_length=800): center=True, reflect pad n_fft//2,
    complex64 [513, 1 + len(y)//hop]."""
    w = hann_padded()
    yp = np.pad(np.asarray(y, dtype=np.float32), N_FFT // 2, mode='reflect')
    n_frames = 1 + (len(yp) - N_FFT) // HOP
    idx = np.arange(N_FFT)[None, :] + HOP * np.arange(n_frames)[:, None]
    frames = yp[idx] * w[None, :]
    return np.fft.rfft(frames, axis=1).astype(np.complex64).T


def istft(S):
    """librosa.istft(S, 200, win_length=800, window='hann') (<=0.7): windowed irfft,
    overlap-add, divide by window sum-square where > tiny(float32), trim n_fft//2 both ends."""
    w = hann_padded()
    n_frames = S.shape[1]
    exp_len = N_FFT + HOP * (n_frames - 1)
    y = np.zeros(exp_len, dtype=np.float32)
    ytmp = (np.fft.irfft(S.T, n=N_FFT, axis=1).astype(np.float32) * w[None, :])
    wss = np.zeros(exp_len, dtype=np.float32)
    wsq = (w.astype(np.float32) ** 2)
    for i in range(n_frames):
        y[i * HOP:i * HOP + N_FFT] += ytmp[i]
        wss[i * HOP:i * HOP + N_FFT] += wsq
    nz = wss > np.finfo(np.float32).tiny
    y[nz] /= wss[nz]
    return y[N_FFT // 2:-(N_FFT // 2)]


def griffin_lim(spectrogram, n_iter=N_ITER):
    """convert.py:39-52.  spectrogram: real [513, T] magnitudes; zero-phase start."""
    S = np.asarray(spectrogram, dtype=np.float32)
    X_best = S.astype(np.complex64)
    for _ in range(n_iter):
        X_t = istft(X_best)
        est = stft(X_t)
        phase = est / np.maximum(1e-8, np.abs(est))
        X_best = S * phase
    return np.real(istft(X_best))


def de_preemphasis(wav, coef=PREEMPH):
    """scipy.signal.lfilter([1], [1, -0.97], wav)  (convert.py:60): y[n] = x[n] + 0.97 y[n-1]."""
    y = np.empty(len(wav), dtype=np.float64)
    acc = 0.0
    for i, v in enumerate(np.asarray(wav, dtype=np.float64)):
        acc = v + coef * acc
        y[i] = acc
    return y


def trim(wav, top_db=60, frame_length=2048, hop_length=512):
    """librosa.effects.trim defaults: RMS (centered frames, reflect pad) in dB relative to the max;
    keep [first, last] non-silent frames."""
    y = np.asarray(wav, dtype=np.float64)
    yp = np.pad(y, frame_length // 2, mode='reflect')
    n_frames = 1 + (len(yp) - frame_length) // hop_length
    idx = np.arange(frame_length)[None, :] + hop_length * np.arange(n_frames)[:, None]
    mse = np.mean(yp[idx] ** 2, axis=1)
    ref = np.max(mse)
    db = 10.0 * np.log10(np.maximum(1e-10, mse)) - 10.0 * np.log10(np.maximum(1e-10, ref))
    nonsilent = np.flatnonzero(db > -top_db)
    if nonsilent.size == 0:
        return y[0:0], (0, 0)
    start = int(nonsilent[0]) * hop_length
    end = min(len(y), (int(nonsilent[-1]) + 1) * hop_length)
    return y[start:end], (start, end)


def denormalize(mag_tf):
    """convert.py:56-58 on mag [T,513] -> amplitude [513,T]."""
    mag = np.asarray(mag_tf).T
    mag = (np.clip(mag, 0, 1) * MAX_DB) - MAX_DB + REF_DB
    return np.power(10.0, mag * 0.05)


def spectrogram2wav(mag_tf, n_iter=N_ITER, do_trim=True):
    """convert.py:55-62."""
    wav = griffin_lim(denormalize(mag_tf), n_iter=n_iter)
    wav = de_preemphasis(wav)
    if do_trim:
        wav, _ = trim(wav)
    return wav.astype(np.float32)


# ---- feature extraction (preprocess.py:227-258), SURVEY 8(f) item 3 -------------------------------------------------
# PARITY UNPINNED: librosa is not installed and the reference pins no version (its call style implies <= 0.7); the mel
# filterbank and the trim are restated from librosa's documented semantics (Slaney mel scale, area normalisation).
N_MELS = 80


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-12) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_basis(sr=SR, n_fft=N_FFT, n_mels=N_MELS):
    """librosa.filters.mel(sr, n_fft, n_mels) (htk=False, norm='slaney', fmin 0, fmax sr/2): float32 [n_mels, 1 + n_fft//2]."""
    fft_f = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def get_spectrograms(y, do_trim=True):
    """preprocess.py:237-258 on an already loaded 16 kHz waveform: (mel [T, 80], mag [T, 513]) float32."""
    y = np.asarray(y, dtype=np.float32)
    if do_trim:
        y = trim(y)[0].astype(np.float32)
    y = np.append(y[0], y[1:] - PREEMPH * y[:-1]).astype(np.float32)
    mag = np.abs(stft(y))                                                      # [513, T]
    mel = mel_basis() @ mag
    mel = 20 * np.log10(np.maximum(1e-5, mel))
    mag = 20 * np.log10(np.maximum(1e-5, mag))
    mel = np.clip((mel - REF_DB + MAX_DB) / MAX_DB, 1e-8, 1)
    mag = np.clip((mag - REF_DB + MAX_DB) / MAX_DB, 1e-8, 1)
    return mel.T.astype(np.float32), mag.T.astype(np.float32)
