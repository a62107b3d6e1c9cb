"""Hyper-parameter surface of the reference (hps/hps.py:17-95): the DSP constants singleton `hp`
and the 32-key `Hps` namedtuple loaded from hps/*.json (a missing or extra key is a TypeError, as in
the reference)."""
import json
from collections import namedtuple


class ProcessingHyperparams(object):
    """hps/hps.py:17-35."""

    def __init__(self):
        self.max_duration = 10.0
        self.sr = 16000
        self.n_fft = 1024
        self.frame_shift = 0.0125
        self.frame_length = 0.05
        self.hop_length = int(self.sr * self.frame_shift)      # 200
        self.win_length = int(self.sr * self.frame_length)     # 800
        self.n_mels = 80
        self.power = 1.2
        self.n_iter = 300
        self.preemphasis = .97
        self.max_db = 100
        self.ref_db = 20
        self.prior_freq = 3000
        self.prior_weight = 0.5


hp = ProcessingHyperparams()

HPS_KEYS = ['g_mode', 'enc_mode', 'load_model_list', 'lr', 'alpha_dis', 'alpha_enc', 'beta_dis', 'beta_gen', 'beta_clf',
            'lambda_', 'ns', 'enc_dp', 'dis_dp', 'max_grad_norm', 'max_step', 'seg_len', 'n_samples', 'enc_size', 'emb_size',
            'n_speakers', 'n_target_speakers', 'n_latent_steps', 'n_patch_steps', 'batch_size', 'lat_sched_iters',
            'enc_pretrain_iters', 'dis_pretrain_iters', 'patch_iters', 'iters', 'tacotron_iters', 'tclf_iters', 'max_to_keep']

# hps/zerospeech_english.json values (enc_size 6 is the challenge's low-bitrate setting; README: 1024 variant)
ENGLISH = dict(g_mode='targeted_residual', enc_mode='multilabel_binary', load_model_list='encoder, decoder, generator',
               lr=0.0001, alpha_dis=1, alpha_enc=0.01, beta_dis=1, beta_gen=1, beta_clf=1, lambda_=10, ns=0.01, enc_dp=0.5,
               dis_dp=0.3, max_grad_norm=5, max_step=5, seg_len=128, n_samples=400000, enc_size=6, emb_size=1024,
               n_speakers=102, n_target_speakers=2, n_latent_steps=5, n_patch_steps=5, batch_size=16, lat_sched_iters=50000,
               enc_pretrain_iters=400000, dis_pretrain_iters=20000, patch_iters=50000, iters=100000, tacotron_iters=500000,
               tclf_iters=10000, max_to_keep=10)


class Hps(object):
    def __init__(self, path=None):
        self.hps = namedtuple('hps', HPS_KEYS)
        if path is not None:
            self.load(path)
            print('[HPS Loader] - Loading from: ', path)
        else:
            print('[HPS Loader] - Using default parameters since no .json file is provided.')
            self._hps = self.hps(**ENGLISH)

    def get_tuple(self):
        return self._hps

    def load(self, path):
        with open(path, 'r') as f_json:
            hps_dict = json.load(f_json)
        self._hps = self.hps(**hps_dict)

    def dump(self, path):
        with open(path, 'w') as f_json:
            json.dump(self._hps._asdict(), f_json, indent=4, separators=(',', ': '))


def make_hps(**overrides):
    d = dict(ENGLISH)
    d.update(overrides)
    return namedtuple('hps', HPS_KEYS)(**d)
