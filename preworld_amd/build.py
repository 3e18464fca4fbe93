"""Build libpreworld_hip.so in-tree with hipcc for gfx950 (the only supported target).

    python -m preworld_amd.build [--force] [--verbose]

Each .hip file is compiled to an object (parallel), then linked into
preworld_amd/csrc/libpreworld_hip.so, which travels to the GPU box with the snapshot.
"""
import argparse
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libpreworld_hip.so')
ARCH = 'gfx950'

COMMON = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden',
          '-Wall', '-Wno-unused-function', '-fno-gpu-rdc']
# per-file extra flags
EXTRA = {
    # geometry / pooling must round exactly like the oracle: no FMA contraction
    'pw_lss.hip': ['-ffp-contract=off'],
    'pw_lss_fused.hip': ['-ffp-contract=off'],
    'pw_render.hip': ['-ffp-contract=off'],
    'pw_stereo.hip': ['-ffp-contract=off'],
}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def source_hash():
    """sha256[:16] over every kernel source, kernel header and the C ABI header (names + contents, sorted): what
    pw_build_id() of a library built from this tree returns"""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.h')))
    for f in files + [os.path.join('..', '..', 'include', 'preworld_hip.h')]:
        h.update(os.path.basename(f).encode() + b'\0')
        h.update(open(os.path.join(CSRC, f), 'rb').read())
    return h.hexdigest()[:16]


def _write_build_id():
    path = os.path.join(CSRC, 'pw_build_id.inc')
    text = '#define PW_BUILD_ID "%s"\n' % source_hash()
    if not os.path.exists(path) or open(path).read() != text:
        with open(path, 'w') as f:
            f.write(text)


def _needs_build(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, force, verbose):
    obj = os.path.join(CSRC, src[:-4] + '.o')
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in os.listdir(CSRC)
                                        if h.endswith('.h')]
    if src == 'pw_core.hip':
        deps.append(os.path.join(CSRC, 'pw_build_id.inc'))
    deps.append(os.path.join(HERE, '..', 'include', 'preworld_hip.h'))
    if not force and not _needs_build(obj, deps):
        return obj, False
    cmd = ['hipcc'] + COMMON + EXTRA.get(src, []) + ['-c', os.path.join(CSRC, src), '-o', obj]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return obj, True


def build(force=False, verbose=False, jobs=4):
    _write_build_id()
    srcs = sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
        res = list(ex.map(lambda s: _compile(s, force, verbose), srcs))
    objs = [r[0] for r in res]
    if force or any(r[1] for r in res) or not os.path.exists(LIB):
        cmd = ['hipcc', '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--force', action='store_true')
    ap.add_argument('--verbose', action='store_true')
    a = ap.parse_args()
    print(build(a.force, a.verbose))
