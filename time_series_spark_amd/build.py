"""Builds libtsf_amd.so (the HIP kernels + C-ABI of include/tsf.h) for gfx950, in-tree.

``python -m time_series_spark_amd.build`` or ``__graft_entry__.build()``.  hipcc cross-compiles
without a GPU.  ``-ffp-contract=off`` is part of the numerical contract (canonical arithmetic,
see DESIGN.md): only the explicit ``__builtin_fma`` calls may fuse.
"""
import concurrent.futures
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, '_obj')
LIB = os.path.join(HERE, 'libtsf_amd.so')
ARCH = 'gfx950'
FLAGS = ['--offload-arch=' + ARCH, '-O3', '-ffp-contract=off', '-fPIC', '-std=c++17',
         '-Wno-unused-value', '-pthread'] + os.environ.get('TSF_HIPCC_FLAGS', '').split()


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def _newest_header():
    hs = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(CSRC, '*.inc'))
    hs.append(os.path.join(HERE, '..', 'include', 'tsf.h'))
    hs.append(os.path.join(HERE, '..', 'include', 'tsf_dev.h'))
    return max(os.path.getmtime(h) for h in hs)


# per-file flags (the reason is at the top of the file named)
EXTRA = {'tsf_inst_quad3.hip': ['-mllvm', '-disable-machine-licm'],
         'tsf_inst_quad4.hip': ['-mllvm', '-disable-machine-licm'],
         'tsf_inst_quad.hip': ['-mllvm', '-disable-machine-licm']}


def _compile(src, obj, hdr_mtime, force):
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src)
            and os.path.getmtime(obj) >= hdr_mtime):
        return obj, False
    cmd = [_hipcc()] + FLAGS + EXTRA.get(os.path.basename(src), []) + ['-c', src, '-o', obj]
    subprocess.check_call(cmd)
    return obj, True


def build_lib(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.cpp')))
    hdr = _newest_header()
    jobs = [(s, os.path.join(OBJ, os.path.splitext(os.path.basename(s))[0] + '.o')) for s in srcs]
    changed = False
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        for obj, did in ex.map(lambda j: _compile(j[0], j[1], hdr, force), jobs):
            changed = changed or did
            if verbose and did:
                print('compiled', os.path.basename(obj))
    if changed or not os.path.exists(LIB):
        cmd = [_hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC', '-pthread', '-o', LIB] + [o for _, o in jobs] + ['-lz']
        subprocess.check_call(cmd)
        if verbose:
            print('linked', LIB)
    return LIB


if __name__ == '__main__':
    print(build_lib(force='--force' in sys.argv, verbose=True))
