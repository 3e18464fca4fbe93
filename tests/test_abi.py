"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/preworld_hip.h declares.  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess

from preworld_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_every_declared_symbol():
    path = build.build()
    assert os.path.exists(path)
    protos = _lib.parse_header()
    declared = set(re.findall(r'\b(pw_\w+)\s*\(', open(os.path.join(ROOT, 'include', 'preworld_hip.h')).read()))
    assert declared == set(protos), declared ^ set(protos)
    l = ctypes.CDLL(path)
    for name in declared:
        assert hasattr(l, name), name
    # the library is a gfx950 code object, nothing else
    out = subprocess.run(['/opt/rocm/lib/llvm/bin/clang-offload-bundler', '--list', '--type=o',
                          '--input=' + path], capture_output=True, text=True)
    if out.returncode == 0 and out.stdout.strip():
        assert 'gfx950' in out.stdout


def test_no_gpu_calls_fail_cleanly_and_version():
    l = _lib.lib()
    assert l.pw_version() >= 100
    # argument validation happens before any HIP call
    rc = l.pw_lss_camera_matrices(0, None, None, None, None, None, None, None)
    assert rc == -1
    assert b'pw_lss_camera_matrices' in l.pw_last_error()


def test_product_never_imports_oracle():
    """The product package must not import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, 'preworld_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), f
                assert 'pw_oracle' not in src, f
