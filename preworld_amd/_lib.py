"""ctypes binding of libpreworld_hip.so (the C ABI declared in include/preworld_hip.h).

The prototypes are parsed from the header itself, so the Python side cannot drift from
the C declarations.  There is NO fallback: if the library is missing or a call fails,
this raises -- the product path never routes through a CPU or PyTorch implementation.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PW_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libpreworld_hip.so')      # PW_LIB_PATH: A/B builds (tools/build_variant.py)
HEADER_PATH = os.path.join(_HERE, '..', 'include', 'preworld_hip.h')

_CTYPES = {
    'int': ctypes.c_int, 'int32_t': ctypes.c_int32, 'int64_t': ctypes.c_int64,
    'size_t': ctypes.c_size_t, 'float': ctypes.c_float, 'double': ctypes.c_double,
}


class PreworldHipError(RuntimeError):
    pass


def parse_header(path=HEADER_PATH):
    """Return {name: (restype, [argtypes], [argnames])} for every pw_* declaration."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    src = re.sub(r'//[^\n]*', ' ', src)
    src = re.sub(r'^[ \t]*#[^\n]*', ' ', src, flags=re.M)          # preprocessor lines
    protos = {}
    for m in re.finditer(r'([A-Za-z_][\w\s\*]*?)\b(pw_\w+)\s*\(([^;{]*?)\)\s*;', src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.endswith('*'):
            restype = ctypes.c_char_p if 'char' in ret else ctypes.c_void_p
        else:
            restype = _CTYPES[ret.replace('const', '').strip()]
        argtypes, argnames = [], []
        if args and args != 'void':
            for a in args.split(','):
                a = ' '.join(a.split())
                if '*' in a:
                    argtypes.append(ctypes.c_void_p)
                    argnames.append(a.split('*')[-1].strip())
                else:
                    toks = a.replace('const ', '').split()
                    argtypes.append(_CTYPES[toks[0]])
                    argnames.append(toks[-1])
        protos[name] = (restype, argtypes, argnames)
    return protos


_lib = None
_protos = None


def lib():
    """Load the HIP library (once).  Raises PreworldHipError if it has not been built."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PreworldHipError(
            'libpreworld_hip.so is missing (%s). Build it with `python -m preworld_amd.build` '
            '(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback.' % LIB_PATH)
    # torch first: its wheel carries its own libamdhip64; if this library were loaded before it, the
    # process would hold two HIP runtimes and calls through this one would see no device
    # ("no ROCm-capable device is detected" -- hit by build() followed by smoke() in one process)
    import torch  # noqa: F401
    l = ctypes.CDLL(LIB_PATH)
    _protos = parse_header()
    for name, (restype, argtypes, _) in _protos.items():
        try:
            fn = getattr(l, name)
        except AttributeError:
            raise PreworldHipError('libpreworld_hip.so does not export %s (stale build? '
                                   'run python -m preworld_amd.build --force)' % name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = l
    return _lib


def protos():
    lib()
    return _protos


def call(name, *args):
    """Call an int-returning entry point; raise with pw_last_error() on failure."""
    l = lib()
    rc = getattr(l, name)(*args)
    if rc != 0:
        msg = l.pw_last_error()
        raise PreworldHipError('%s failed (%d): %s' % (name, rc, msg.decode() if msg else ''))


def call_size(name, *args):
    return int(getattr(lib(), name)(*args))
