"""Build an experiment variant of libpreworld_hip.so next to the real one:
    python tools/build_variant.py NAME [--only=pw_x.hip] -DFLAG [-DFLAG ...]   ->  preworld_amd/csrc/variants/libpreworld_hip_NAME.so
and run anything against it with PW_LIB_PATH=<that file>.  Development aid for A/B timing; never used by the product."""
import concurrent.futures
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from preworld_amd import build as B  # noqa: E402


def main(name, flags):
    out = os.path.join(B.CSRC, 'variants', name)
    os.makedirs(out, exist_ok=True)
    B._write_build_id()
    only = [f[len('--only='):] for f in flags if f.startswith('--only=')]      # other sources: the objects of the real build
    flags = [f for f in flags if not f.startswith('--only=')]
    if only:
        B.build()

    def cc(src):
        if only and src not in only:
            return os.path.join(B.CSRC, src[:-4] + '.o')
        obj = os.path.join(out, src[:-4] + '.o')
        subprocess.check_call(['hipcc'] + B.COMMON + B.EXTRA.get(src, []) + flags + ['-c', os.path.join(B.CSRC, src), '-o', obj])
        return obj
    with concurrent.futures.ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(cc, B.sources()))
    lib = os.path.join(B.CSRC, 'variants', 'libpreworld_hip_%s.so' % name)
    subprocess.check_call(['hipcc', '--offload-arch=' + B.ARCH, '-shared', '-fPIC', '-o', lib] + objs)
    print(lib)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2:])
