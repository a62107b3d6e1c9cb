"""CPU tests: the C ABI library loads and exports every declared symbol with the declared struct layouts,
the product path fails loudly without a GPU, and the host logic (hps, dataloader, segmenter, CLI, data-parallel
reducer over gloo) behaves like the reference's."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import zs_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    import zs_amd  # noqa: F401
    from zs_amd import _lib, build
    build.build(verbose=False)
    return _lib


def test_library_exports_every_declared_symbol(lib):
    L = lib.lib()
    assert len(lib.FUNCTIONS) >= 24
    missing = [f for f in lib.FUNCTIONS if not hasattr(L, f)]
    assert not missing, missing
    assert L.zs_abi_version() == lib.ENUMS['ZS_ABI_VERSION']


def test_struct_layouts_match_the_c_compiler(lib, tmp_path):
    names = list(lib.STRUCTS)
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(){\n' % lib.HEADER
    for n in names:
        src += 'printf("%s %%zu\\n", sizeof(%s));\n' % (n, n)
        for f, _ in lib._STRUCT_FIELDS[n]:
            src += 'printf("%s.%s %%zu\\n", offsetof(%s,%s));\n' % (n, f, n, f)
    src += 'return 0;}\n'
    c = tmp_path / 'sz.c'
    c.write_text(src)
    subprocess.check_call(['gcc', str(c), '-o', str(tmp_path / 'sz')])
    for line in subprocess.check_output([str(tmp_path / 'sz')]).decode().split('\n'):
        if not line:
            continue
        k, v = line.split()
        if '.' in k:
            n, f = k.split('.')
            assert getattr(lib.STRUCTS[n], f).offset == int(v), k
        else:
            assert ctypes.sizeof(lib.STRUCTS[k]) == int(v), k


def test_invalid_calls_return_errors_not_crashes(lib):
    with pytest.raises(lib.ZsError, match='null operand'):
        lib.call('zs_gemm_conv', 'ZsGemmConv', 0, dtype=0)
    with pytest.raises(lib.ZsError):
        lib.call('zs_instnorm_fwd', 'ZsInstNormFwd', 0, dtype=7)
    assert 'zs_' in lib.last_error()


def test_no_cpu_fallback():
    import zs_amd  # noqa: F401
    from zs_amd import _lib
    from zs_amd.layers import Ctx
    from zs_amd.model import Encoder
    if torch.cuda.is_available():
        pytest.skip('has GPU')
    with pytest.raises(_lib.ZsError, match='no CPU'):
        Ctx('cpu')
    enc = Encoder(c_in=80, c_h1=16, c_h2=32, c_h3=16, enc_size=8, seg_len=128, enc_mode='multilabel_binary')
    with pytest.raises(_lib.ZsError, match='no CPU fallback'):
        enc(torch.rand(1, 80, 16))
    with pytest.raises(NotImplementedError):
        Encoder(enc_mode='nonsense')
    src = open(os.path.join(ROOT, 'zerospeech-tts-without-t_amd', 'model.py')).read() + \
        open(os.path.join(ROOT, 'zerospeech-tts-without-t_amd', 'engine.py')).read() + \
        open(os.path.join(ROOT, 'zerospeech-tts-without-t_amd', 'trainer.py')).read() + \
        open(os.path.join(ROOT, 'zerospeech-tts-without-t_amd', 'convert.py')).read()
    assert 'zs_oracle' not in src and 'oracle' not in src.replace('oracle/', '')      # product never imports the oracle


def test_hps_surface():
    from zs_amd.hps import HPS_KEYS, Hps, hp
    assert len(HPS_KEYS) == 32
    h = Hps(os.path.join(ROOT, 'hps', 'zerospeech_english.json')).get_tuple()
    assert (h.seg_len, h.enc_size, h.emb_size, h.n_speakers, h.batch_size, h.lr, h.max_grad_norm) == (128, 6, 1024, 102, 16, 1e-4, 5)
    assert (hp.sr, hp.n_fft, hp.hop_length, hp.win_length, hp.n_iter, hp.preemphasis, hp.max_db, hp.ref_db) == \
        (16000, 1024, 200, 800, 300, .97, 100, 20)
    d = dict(h._asdict())
    d.pop('lr')
    p = os.path.join(ROOT, 'tests', '_tmp_hps.json')
    try:
        json.dump(d, open(p, 'w'))
        with pytest.raises(TypeError):
            Hps(p)
        d['lr'] = 1e-4; d['extra'] = 1
        json.dump(d, open(p, 'w'))
        with pytest.raises(TypeError):
            Hps(p)
    finally:
        os.remove(p)


def test_dataloader_contract():
    from zs_amd.dataloader import DataLoader, SyntheticDataset
    ds = SyntheticDataset(40, seg_len=16, n_bins=513, n_speakers=7, seed=1)
    dl = DataLoader(ds, batch_size=16)
    c, x = next(dl)
    assert c.dtype == torch.int64 and c.shape == (16,) and x.dtype == torch.float32 and x.shape == (16, 16, 513)
    assert x.min() >= 1e-8 and x.max() <= 1.0
    assert dl.index == 16
    next(dl)                       # index 16 + 2*16 >= 40 -> wraps to 0 (dataloader.py:48-49)
    assert dl.index == 0
    c2, x2 = next(dl)
    assert torch.equal(c, c2) and torch.equal(x, x2)
    dl16 = DataLoader(SyntheticDataset(16, seg_len=8), batch_size=16)   # BASELINE config 1: one batch re-served forever
    a = next(dl16); b = next(dl16)
    assert torch.equal(a[1], b[1]) and dl16.index == 0


def test_segmenter_matches_oracle_and_text_format(tmp_path):
    from zs_amd import convert
    for n in (9, 100, 128, 129, 255, 256, 257, 300, 383, 384, 385, 700, 1000):
        if n <= 128:
            continue
        assert convert.fragments(n, 128) == O.fragment_plan(n, 128)[1], n
    with pytest.raises(RuntimeError, match='too short'):
        convert.fragments(129, 200)                      # one piece of 128 frames < seg_len 200 at idx 0
    enc = np.array([[1., 0., 1., 1.], [0., 0., 0., 1.]])
    convert.write_encodings(str(tmp_path / 'e.txt'), enc)
    assert open(str(tmp_path / 'e.txt')).read() == '1 0 1 1\n0 0 0 1\n' == O.encodings_text(enc)
    assert convert.parse_encodings(enc) == ['1 0 1 1', '0 0 0 1']
    rng = np.random.RandomState(0)
    y = np.concatenate([np.zeros(3000), rng.randn(8000) * 0.1, np.zeros(5000)])
    a, ia = convert.trim(y)
    b, ib = O.trim(y)
    assert ia == ib and np.array_equal(a, b) and 0 < len(a) < len(y)
    convert.write_wav(str(tmp_path / 'w.wav'), np.array([0.0, 0.5, -0.5, 1.0], dtype=np.float32), 16000)
    import wave
    with wave.open(str(tmp_path / 'w.wav')) as f:
        assert (f.getframerate(), f.getsampwidth(), f.getnchannels(), f.getnframes()) == (16000, 2, 1, 4)


def test_cli_surface(capsys):
    sys.path.insert(0, ROOT)
    import main
    args, hps = main.argument_runner(['--train_ae', '--hps_path', os.path.join(ROOT, 'hps', 'zerospeech_english_1024.json')])
    assert args.train_ae and args.g_mode == 'targeted_residual' and args.enc_mode == 'multilabel_binary' and hps.enc_size == 1024
    args, _ = main.argument_runner(['--test', '--enc_only', '--dataset', 'surprise', '--hps_path',
                                    os.path.join(ROOT, 'hps', 'zerospeech_surprise.json')])
    assert args.ckpt_dir == './ckpt_surprise' and args.dataset_path == './data/dataset_surprise.hdf5' and \
        args.synthesis_list == './data/surprise/synthesis.txt' and args.sub_result_dir == './surprise/'
    with pytest.raises(NotImplementedError):
        main.main(['--train_al', '--hps_path', os.path.join(ROOT, 'hps', 'zerospeech_english.json')])


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import zs_amd  # noqa: F401
    from zs_amd import parallel
    from conftest import load_golden, sub_sd
    torch.set_num_threads(1)
    parallel.init_from_env('gloo')
    d, m = load_golden('train_f80.npz')
    g = torch.Generator().manual_seed(0)
    B = 4
    x = torch.rand(B, m['c_in'], 32, generator=g)
    c = torch.randint(0, m['n_spk'], (B,), generator=g)
    G = O.gumbel_from_uniform(torch.rand(B, 4, m['enc_size'], 2, generator=g))
    hp = dict(ns=m['ns'], enc_dp=0.0, enc_size=m['enc_size'], seg_len=m['seg_len'])
    lo, hi = parallel.shard_range(B, rank, world)
    _, (ge, gd), _, _ = O.train_ae_grads(sub_sd(d, 'enc0.'), sub_sd(d, 'dec0.'), x[lo:hi], c[lo:hi], hp, G=G[lo:hi])
    flat = torch.cat([v.reshape(-1) for v in list(ge.values()) + list(gd.values())])
    red = parallel.GradReducer(bucket_bytes=64 << 10)          # several buckets
    # tagged ranges in backward order (trainer.AEStep._multi_actions): the decoder's ranges first, the encoder's last; a wait
    # for one tag leaves the other tag's collectives pending
    ne = sum(v.numel() for v in ge.values())
    cut = ne + (flat.numel() - ne) // 3
    red.start(flat[cut:], tag='dec')
    red.start(flat[ne:cut], tag='dec')
    red.start(flat[:ne], tag='enc')
    red.finish('dec')
    assert red.pending and all(t == 'enc' for t, _, _, _ in red.pending)
    red.finish('enc')
    assert not red.pending
    assert red.scale == 0.5
    flat = flat * red.scale            # the buffers hold the SUM over ranks; the consumer (zs_adam_clip) folds 1/world in
    if rank == 0:
        _, (fe, fd), _, _ = O.train_ae_grads(sub_sd(d, 'enc0.'), sub_sd(d, 'dec0.'), x, c, hp, G=G)
        full = torch.cat([v.reshape(-1) for v in list(fe.values()) + list(fd.values())])
        q.put(((flat - full).abs().max().item(), full.abs().max().item(), parallel.world_size()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_data_parallel_gradient_average_gloo():
    """world_size 2 over gloo: the averaged per-rank gradients equal the single-process global-batch gradient
    (segments are independent: InstanceNorm is per sample)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err, scale, world = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert world == 2 and err <= 2e-5 * scale, (err, scale)


def test_gradient_buckets_partition_the_flat_buffers_in_backward_order():
    """The bucket ranges of the data-parallel step (trainer.AEStep._multi_actions) tile each net's flat gradient buffer exactly,
    and every bucket is a contiguous range whose parameters' gradients are final when its all-reduce starts."""
    from zs_amd.model import Decoder, Encoder
    enc = Encoder(c_in=80, c_h1=16, c_h2=32, c_h3=16, ns=0.01, dp=0.5, enc_size=8, seg_len=128, enc_mode='multilabel_binary')
    dec = Decoder(c_in=8, c_out=80, c_h=32, c_a=4, ns=0.01, seg_len=128)
    nd, ne = dec.flat_params()[1].numel(), enc.flat_params()[1].numel()
    d_tail, d_conv, d_emb = dec.flat_range('dense1.weight', 'linear.bias'), dec.flat_range('conv1.weight', 'conv6.bias'), \
        dec.flat_range('input_emb.weight', 'emb5.weight')
    assert d_conv[0] == 0 and d_conv[1] == d_tail[0] and d_tail[1] == d_emb[0] and d_emb[1] == nd
    e_bank, e_rest = enc.flat_range('conv1s.0.weight', 'conv1s.6.bias'), enc.flat_range('conv2.weight', 'linear.bias')
    assert e_bank[0] == 0 and e_bank[1] == e_rest[0] and e_rest[1] == ne
    names = [n for n, _ in dec.named_parameters()]
    lo = names.index('dense1.weight')
    assert names[lo:names.index('input_emb.weight')] == [n for n in names if n.split('.')[0] in
                                                         ('dense1', 'dense2', 'dense3', 'dense4', 'RNN', 'dense5', 'linear')]
    with pytest.raises(ValueError):
        dec.flat_range('linear.bias', 'conv1.weight')


def test_data_parallel_loader_shards_are_disjoint():
    """world 2: rank r serves items [index + r*B, index + (r+1)*B) of each global batch and both advance by world*B -- no item
    is served twice before the wrap; world 1 is the reference's sequence (dataloader.py:43-52)."""
    from zs_amd.dataloader import DataLoader

    class Ids(object):
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            assert 0 <= i < self.n
            return (np.int64(i), np.full((2, 3), i, dtype=np.float32))

    B, n = 4, 100
    loaders = [DataLoader(Ids(n), B, rank=r, world=2) for r in range(2)]
    seen = []
    for step in range(11):                      # 11 global batches of 8 = 88 items: 88 + 16 >= 100 wraps after the 11th
        assert loaders[0].index == loaders[1].index == step * 8
        for dl in loaders:
            seen += next(dl)[0].tolist()
    assert len(seen) == len(set(seen)) == 88 and sorted(seen) == list(range(88))
    next(loaders[0]); next(loaders[1])
    assert loaders[0].index == loaders[1].index == 0          # wrapped together
    single = DataLoader(Ids(n), B)
    ref = []
    for _ in range(5):
        ref += next(single)[0].tolist()
    assert ref == list(range(20))


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """`bench.py --gpus 2` under a 1-rank environment must be an error, not a 1-GPU number labelled n_gpus 2."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and 'refusing' in r.stderr and r.stdout.strip() == ''


def test_sanitizer_build_command():
    """ZS_SANITIZE=address: host-side instrumentation only (each -fsanitize= directly after -Xarch_host), separate output."""
    from zs_amd import build as zb
    cmd = zb.command(san='address')
    i = cmd.index('-fsanitize=address')
    assert cmd[i - 1] == '-Xarch_host' and cmd[cmd.index('-o') + 1].endswith('libzs_amd.address.so')
    assert '--offload-arch=gfx950' in cmd
    assert zb.command(san='')[zb.command(san='').index('-o') + 1].endswith('libzs_amd.so')


def test_shard_range():
    from zs_amd.parallel import shard_range
    parts = [shard_range(10, r, 4) for r in range(4)]
    assert parts == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert sum(b - a for a, b in parts) == 10
