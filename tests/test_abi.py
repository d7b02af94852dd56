"""The C ABI: both libraries load, export every symbol the headers declare, and the GPU library
refuses to work (rather than falling back to a CPU path) when no GPU is present."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

import pathtracing_amd as pt
from pathtracing_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(ptk_[a-z_0-9]+)\s*\(", text))


def test_host_library_exports_every_declared_symbol():
    lib = C.CDLL(os.path.join(ROOT, "pathtracing_amd", "libptkhost.so"))
    names = declared_functions("ptk_host.h")
    assert names and names == set(_ffi.HOST_PROTOTYPES)
    for name in names:
        getattr(lib, name)


def test_device_library_exports_every_declared_symbol():
    lib = C.CDLL(os.path.join(ROOT, "pathtracing_amd", "libptk.so"))
    names = declared_functions("ptk.h")
    assert names and names == set(_ffi.DEVICE_PROTOTYPES)
    for name in names:
        getattr(lib, name)


def test_struct_layouts_match_the_header_sizes():
    # sizes computed from the C declarations (8-byte alignment everywhere)
    assert C.sizeof(_ffi.PtkMap) == 24 and C.sizeof(_ffi.PtkMedium) == 88 and C.sizeof(_ffi.PtkMaterial) == 56
    assert C.sizeof(_ffi.PtkObject) == 120 and C.sizeof(_ffi.PtkCamera) == 88 and C.sizeof(_ffi.PtkSettings) == 48
    assert C.sizeof(_ffi.PtkRenderJob) == 32 and C.sizeof(_ffi.PtkDenoiseTask) == 48
    src = r'''
#include <stdio.h>
#include "ptk.h"
int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(PtkMap), sizeof(PtkMedium), sizeof(PtkMaterial),
  sizeof(PtkObject), sizeof(PtkCamera), sizeof(PtkSettings), sizeof(PtkSceneDesc), sizeof(PtkRenderJob), sizeof(PtkStats),
  sizeof(PtkDenoiseTask)); return 0; }'''
    import tempfile
    d = tempfile.mkdtemp()
    open(os.path.join(d, "s.c"), "w").write(src)
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")], check=True)
    sizes = [int(x) for x in subprocess.run([os.path.join(d, "s")], capture_output=True, text=True, check=True).stdout.split()]
    ours = [C.sizeof(t) for t in (_ffi.PtkMap, _ffi.PtkMedium, _ffi.PtkMaterial, _ffi.PtkObject, _ffi.PtkCamera, _ffi.PtkSettings,
                                  _ffi.PtkSceneDesc, _ffi.PtkRenderJob, _ffi.PtkStats, _ffi.PtkDenoiseTask)]
    assert sizes == ours


def test_no_gpu_means_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import pathtracing_amd as pt\n"
            "try:\n    pt.init(0)\nexcept pt.PtkError as e:\n    print('ERR', e)\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert "ERR" in out.stdout and "no CPU path" in out.stdout


def test_product_code_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "pathtracing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".sh")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, f)


def test_buffer_arguments_are_checked_before_they_reach_the_library():
    """The C ABI takes plain pointers; the Python layer refuses arrays whose layout is not what they point at."""
    import numpy as np
    with pytest.raises(TypeError):
        pt._dp(np.zeros(4, dtype=np.float32))
    with pytest.raises(TypeError):
        pt._dp(np.zeros((4, 4))[:, ::2])
    with pytest.raises(TypeError):
        pt._u64p(np.zeros(4, dtype=np.int32))
    assert pt._dp(None) is None and pt._dp(np.zeros(3)) is not None
