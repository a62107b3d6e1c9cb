"""Command line of the autoencoder hot path, with the reference's flag names (main.py:27-79):

  python main.py --preprocess [--remake]                  (wav directories -> dataset container + index JSONs)
  python main.py --train_ae [--load_model] [--hps_path hps/zerospeech_english_1024.json] [--synthetic]
  python main.py --train_p | --train_tgat [--load_model] [--synthetic]     (stage 2: patchGAN from an ae checkpoint)
  python main.py --test --enc_only | --test_encode        (needs the preprocessed HDF5 + a checkpoint)
  python main.py --test_single --s_speaker S015 --t_speaker V002 [--enc_only]      (one wav -> result.wav + result.txt)

Multi-GPU training: `python -m torch.distributed.run --nproc-per-node N main.py --train_ae ...` (one
process per GPU; gradients are averaged with RCCL).  Modes outside the stage-1 autoencoder path
(--train_al, --train_c, --train_t, --cross_test,
--test_classify, --encode, --test_asr) are not part of this build and exit with a clear error.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

OUT_OF_SCOPE = ['train_al', 'train_c', 'train_t', 'test_asr', 'cross_test', 'test_classify', 'encode']


def build_parser():
    p = argparse.ArgumentParser(description='zerospeech_project (MI355X autoencoder hot path)')
    for flag in ['preprocess', 'train', 'train_ae', 'train_p', 'train_tgat', 'test', 'test_encode', 'test_single', 'load_model', 'enc_only',
                 'remake', 'synthetic'] + OUT_OF_SCOPE:
        p.add_argument('--' + flag, default=False, action='store_true')
    p.add_argument('--flag', type=str, default='train')
    p.add_argument('--g_mode', default='set_from_hps',
                   choices=['naive', 'targeted', 'targeted_residual', 'enhanced', 'spectrogram', 'tacotron', 'set_from_hps'])
    p.add_argument('--enc_mode', default='set_from_hps',
                   choices=['continues', 'one_hot', 'binary', 'multilabel_binary', 'gumbel_t', 'set_from_hps'])
    p.add_argument('--dataset', choices=['english', 'surprise'], default='english')
    p.add_argument('--dtype', choices=['fp32', 'bf16'], default=os.environ.get('ZS_DTYPE', 'bf16'))
    p.add_argument('--source_path', type=str, default='./data/english/train/unit/')
    p.add_argument('--target_path', type=str, default='./data/english/train/voice/')
    p.add_argument('--test_path', type=str, default='./data/english/test/')
    p.add_argument('--synthesis_list', type=str, default='./data/english/synthesis.txt')
    p.add_argument('--dataset_path', type=str, default='./data/dataset_english.hdf5')
    p.add_argument('--index_path', type=str, default='./data/index_english.json')
    p.add_argument('--index_source_path', type=str, default='./data/index_english_source.json')
    p.add_argument('--index_target_path', type=str, default='./data/index_english_target.json')
    p.add_argument('--speaker2id_path', type=str, default='./data/speaker2id_english.json')
    p.add_argument('--hps_path', type=str, default='./hps/zerospeech_english.json')
    p.add_argument('--ckpt_dir', type=str, default='./ckpt_english')
    p.add_argument('--result_dir', type=str, default='./result')
    p.add_argument('--sub_result_dir', type=str, default='./english/')
    p.add_argument('--model_name', type=str, default='model.pth')
    p.add_argument('--load_train_model_name', type=str, default='model.pth-ae-424000')
    p.add_argument('--load_test_model_name', type=str, default='model.pth-s2-150000')
    p.add_argument('--ckpt_pth', type=str, default=None)
    p.add_argument('--s_speaker', type=str, default='S015')
    p.add_argument('--t_speaker', type=str, default='V002')
    return p


def argument_runner(argv=None):
    from zs_amd.hps import Hps
    parser = build_parser()
    args = parser.parse_args(argv)
    if args.dataset == 'surprise':            # main.py:83-88: english -> surprise in every default path
        for action in parser._actions:
            if isinstance(action.default, str) and 'english' in action.default and \
                    ('path' in action.dest or action.dest in ('synthesis_list', 'sub_result_dir', 'ckpt_dir')):
                action.default = action.default.replace('english', 'surprise')
        args = parser.parse_args(argv)
    print('[Runner] - Dataset: ', args.dataset)
    hps = Hps(args.hps_path).get_tuple()
    if args.g_mode == 'set_from_hps':
        args.g_mode = hps.g_mode
    if args.enc_mode == 'set_from_hps':
        args.enc_mode = hps.enc_mode
    print('[Runner] - Generation mode: ', 'autoencoder only' if args.enc_only else 'with generator')
    print('[Runner] - Generator mode: ', args.g_mode)
    print('[Runner] - Encoder mode: ', args.enc_mode)
    print('[Runner] - Encoding dim: ', hps.enc_size)
    return args, hps


def main(argv=None):
    import zs_amd  # noqa: F401
    args, hps = argument_runner(argv)
    bad = [f for f in OUT_OF_SCOPE if getattr(args, f)]
    if bad:
        raise NotImplementedError('--%s is outside the stage-1 autoencoder hot path this build covers' % bad[0])
    from zs_amd import parallel
    from zs_amd.convert import get_trainer, test_encode, test_from_list, test_single
    from zs_amd.dataloader import DataLoader, Dataset, SyntheticDataset
    from zs_amd.trainer import Trainer

    if args.preprocess:                                   # main.py:113-126
        from zs_amd.preprocess import preprocess
        preprocess(args.source_path, args.target_path, args.test_path, args.dataset_path, args.index_path, args.index_source_path,
                   args.index_target_path, args.speaker2id_path, seg_len=hps.seg_len, n_samples=hps.n_samples, dset=args.flag,
                   remake=args.remake)

    if args.train or args.train_ae or args.train_p or args.train_tgat:                   # main.py:129-162
        rank, world, _ = parallel.init_from_env()
        n_syn = max(4 * hps.batch_size * world, 64)
        if args.synthetic:
            dataset = SyntheticDataset(n_syn, seg_len=hps.seg_len, n_speakers=hps.n_speakers)
            sourceset = SyntheticDataset(n_syn, seg_len=hps.seg_len, n_speakers=max(1, hps.n_speakers - hps.n_target_speakers), seed=1)
            targetset = SyntheticDataset(n_syn, seg_len=hps.seg_len, n_speakers=hps.n_target_speakers, seed=2,
                                         speaker_offset=hps.n_speakers - hps.n_target_speakers)
        else:
            dataset = Dataset(args.dataset_path, args.index_path, seg_len=hps.seg_len)
            sourceset = Dataset(args.dataset_path, args.index_source_path, seg_len=hps.seg_len)
            targetset = Dataset(args.dataset_path, args.index_target_path, seg_len=hps.seg_len)
        mk = lambda ds: DataLoader(ds, hps.batch_size, rank=rank, world=world)           # disjoint per-rank shards of each global batch
        data_loader, source_loader, target_loader = mk(dataset), mk(sourceset), mk(targetset)
        os.makedirs(args.ckpt_dir, exist_ok=True)
        model_path = os.path.join(args.ckpt_dir, args.model_name)
        trainer = Trainer(hps, data_loader, args.g_mode, args.enc_mode, dtype=args.dtype)
        if args.load_model:
            trainer.load_model(os.path.join(args.ckpt_dir, args.load_train_model_name), load_model_list=hps.load_model_list)
        if args.train or args.train_ae:
            trainer.train(model_path, args.flag, mode='pretrain_AE')                     # stage 1: encoder-decoder reconstruction
            trainer.reset_keep()
        if args.train or args.train_p or args.train_tgat:
            trainer.add_duo_loader(source_loader, target_loader)
            trainer.train(model_path, args.flag, mode='patchGAN', target_guided=args.train_tgat)     # stage 2
            trainer.reset_keep()
        if args.train and parallel.rank() == 0:
            # reference main.py:163-173 goes on to the autolocker, t_classify and Tacotron stages: outside the path this build covers
            print("[Runner] - --train ran the pretrain_AE and patchGAN stages; the reference's autolocker / t_classify / Tacotron "
                  'stages are not part of this build and were skipped')

    if args.test or args.test_encode or args.test_single:
        os.makedirs(args.result_dir, exist_ok=True)
        model_path = args.ckpt_pth if args.ckpt_pth is not None else os.path.join(args.ckpt_dir, args.load_test_model_name)
        trainer = get_trainer(args.hps_path, model_path, args.g_mode, args.enc_mode, None)
        result_dir = os.path.join(args.result_dir, args.sub_result_dir)
        os.makedirs(result_dir, exist_ok=True)
        if args.test:
            test_from_list(trainer, hps.seg_len, args.synthesis_list, args.dataset_path, args.speaker2id_path, result_dir, args.enc_only)
        if args.test_single:                              # main.py:190-191
            test_single(trainer, hps.seg_len, args.speaker2id_path, args.result_dir, args.enc_only, args.s_speaker, args.t_speaker)
        if args.test_encode:
            test_encode(trainer, hps.seg_len, args.test_path, args.dataset_path, result_dir, flag='test')


if __name__ == '__main__':
    main()
