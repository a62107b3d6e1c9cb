"""In-tree build of libzs_amd.so: hipcc cross-compiles the gfx950 code objects without a GPU.

ZS_SANITIZE=address|undefined builds a second library, libzs_amd.<san>.so, whose HOST code (argument checks, plans,
launch logic, the C-ABI entry points) is instrumented; the device code objects are unchanged (GPU sanitizers are not
available on this pool).  zs_amd._lib loads that variant when ZS_SANITIZE is set in its environment; run the
interpreter with the matching runtime preloaded, e.g. LD_PRELOAD=$(hipcc -print-file-name=libclang_rt.asan-x86_64.so)."""
import glob
import os
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, 'csrc')


def sanitize():
    s = os.environ.get('ZS_SANITIZE', '').strip()
    if s and s not in ('address', 'undefined'):
        raise ValueError('ZS_SANITIZE must be address or undefined (got %r)' % s)
    return s


def out_path(san=None):
    san = sanitize() if san is None else san
    return os.path.join(_PKG, 'libzs_amd.%s.so' % san if san else 'libzs_amd.so')


OUT = out_path('')


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def needs_build(out=None):
    out = out or out_path()
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(os.path.dirname(_PKG), 'include', 'zs_amd.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def _flags(san):
    flags = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-Wno-unused-result']
    if san:
        # host side only: each -fsanitize= directly after -Xarch_host
        flags += ['-Xarch_host', '-fsanitize=%s' % san, '-Xarch_host', '-fno-omit-frame-pointer', '-g']
    return flags


def obj_dir(san=None):
    san = sanitize() if san is None else san
    return os.path.join(_PKG, 'build', san or 'plain')


def command(out=None, san=None):
    """The one-line equivalent of build(): every source in one hipcc call (documentation / tests)."""
    san = sanitize() if san is None else san
    out = out or out_path(san)
    hipcc = os.environ.get('HIPCC', 'hipcc')
    return [hipcc] + _flags(san) + ['-shared'] + (['-shared-libsan'] if san else []) + ['-o', out] + sources()


def build(force=False, verbose=True):
    """One object per source (compiled side by side, only those whose source or a header changed), then the link."""
    out = out_path()
    if not force and not needs_build(out):
        return out
    san = sanitize()
    hipcc = os.environ.get('HIPCC', 'hipcc')
    od = obj_dir(san)
    os.makedirs(od, exist_ok=True)
    hdrs = glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(os.path.dirname(_PKG), 'include', 'zs_amd.h')]
    t_h = max(os.path.getmtime(h) for h in hdrs)
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(od, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), t_h):
            cmd = [hipcc] + _flags(san) + ['-c', '-o', obj, src]
            if verbose:
                print('[zs_amd.build]', ' '.join(cmd))
            jobs.append((src, subprocess.Popen(cmd)))
    bad = [src for src, pr in jobs if pr.wait() != 0]
    if bad:
        raise subprocess.CalledProcessError(1, 'hipcc -c ' + ' '.join(bad))
    link = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + (['-fsanitize=%s' % san, '-shared-libsan'] if san else []) + ['-o', out] + objs
    if verbose:
        print('[zs_amd.build]', ' '.join(link))
    subprocess.check_call(link)
    return out


if __name__ == '__main__':
    build(force='--force' in sys.argv)
