"""In-tree build of libzs_amd.so: hipcc cross-compiles the gfx950 code objects without a GPU."""
import glob
import os
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, 'csrc')
OUT = os.path.join(_PKG, 'libzs_amd.so')


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(os.path.dirname(_PKG), 'include', 'zs_amd.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get('HIPCC', 'hipcc')
    cmd = [hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', '-Wno-unused-result',
           '-o', OUT] + sources()
    if verbose:
        print('[zs_amd.build]', ' '.join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
