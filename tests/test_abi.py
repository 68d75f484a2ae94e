"""CPU tier for the drop-in boundary: libcoclr_hip.so loads without a GPU, exports
every symbol include/coclr_hip.h declares (and nothing the Python binding expects is
missing), the struct mirrors have the C layout, and the host-side planning entry points
(no kernel launch) answer / reject arguments as documented."""
import ctypes as C
import os
import re
import subprocess

import pytest

from coclr_amd import _lib, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "coclr_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(coclr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "header declares %s but the library does not export it" % n
    assert sorted(_lib.EXPORTED_SYMBOLS) == names, "python binding and header disagree"
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (coclr_[a-z0-9_]+)", out))
    assert exported == set(names), exported ^ set(names)
    assert lib.coclr_abi_version() == _lib.ABI_VERSION


def test_header_is_plain_c():
    """The boundary is a C ABI: the header must compile as C (no torch / C++ types)."""
    src = "#include \"%s\"\nint main(void){coclr_conv_desc d; d.N=1; return (int)sizeof(d)*0;}\n" % HEADER
    path = "/tmp/coclr_hdr_check.c"
    open(path, "w").write(src)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-c", path, "-o", "/tmp/coclr_hdr_check.o"])


def test_struct_layouts_match_header():
    """ctypes mirrors vs the C compiler's own layout of the header structs."""
    src = ("#include <stdio.h>\n#include <stddef.h>\n#include \"%s\"\n"
           "int main(void){printf(\"%%zu %%zu %%zu %%zu %%zu %%zu\\n\", sizeof(coclr_conv_desc),"
           "offsetof(coclr_conv_desc, x_nstride), offsetof(coclr_conv_desc, ys_t),"
           "offsetof(coclr_conv_desc, Nx), sizeof(coclr_pool_desc),"
           "offsetof(coclr_pool_desc, x_nstride)); return 0;}\n" % HEADER)
    open("/tmp/coclr_layout.c", "w").write(src)
    subprocess.check_call(["gcc", "-std=c99", "/tmp/coclr_layout.c", "-o", "/tmp/coclr_layout"])
    want = [int(v) for v in subprocess.check_output(["/tmp/coclr_layout"]).split()]
    got = [C.sizeof(_lib.ConvDesc), _lib.ConvDesc.x_nstride.offset, _lib.ConvDesc.ys_t.offset,
           _lib.ConvDesc.Nx.offset, C.sizeof(_lib.PoolDesc), _lib.PoolDesc.x_nstride.offset]
    assert got == want, (got, want)
    assert _lib.ConvDesc.x_nstride.offset == 88 and _lib.PoolDesc.x_nstride.offset == 72


def test_planning_entry_points():
    # Conv_2c.conv1 at the benchmark size: 32 x 64->192, (16,32,32), 1x3x3
    g = ops.ConvGeom(32, 64, 192, (16, 32, 32), (1, 3, 3), (1, 1, 1), (0, 1, 1))
    assert g.odim == (16, 32, 32)
    assert g.ntiles() == 32 * 16 * 32 * 32 // 128
    assert g.wgrad_workspace() % (192 * 64 * 9) == 0 and g.wgrad_workspace() > 0
    d = g.dgrad()
    assert (d.Cin, d.Cout, d.idim, d.odim, d.p, d.d) == (192, 64, (16, 32, 32), (16, 32, 32),
                                                         (0, 1, 1), (1, 1, 1))
    # strided temporal stem conv and its data gradient (input dilation = stride)
    g = ops.ConvGeom(4, 64, 64, (32, 64, 64), (7, 1, 1), (2, 1, 1), (3, 0, 0))
    assert g.odim == (16, 64, 64) and g.ntiles() > 0
    d = g.dgrad()
    assert d.d == (2, 1, 1) and d.s == (1, 1, 1) and d.p == (3, 0, 0) and d.odim == (32, 64, 64)
    assert d.ntiles() > 0
    assert ops.conv_packed_size(3, 64, 49, False) == 49 * 32 * 128
    assert ops.conv_packed_size(3, 64, 49, True) == 49 * 64 * 128
    assert ops.gemm_workspace(32, 128, 16384, 128) == 128 * 32 * 128


def test_rejected_arguments():
    lib = _lib.load()
    # unsupported stencil (3,3,3) -> invalid value, reported through the return code
    bad = ops.ConvGeom(1, 8, 8, (4, 8, 8), (3, 3, 3), (1, 1, 1), (1, 1, 1))
    with pytest.raises(_lib.HipLibraryError):
        bad.ntiles()
    n = C.c_int32(0)
    assert lib.coclr_conv3d_ntiles(None, C.byref(n)) == 1
    # enqueue geometry checks (K % batch, ref model/pretrain.py:90) are host-side
    assert lib.coclr_queue_enqueue(None, None, 128, 100, 32, None, None) == 1
    assert lib.coclr_positive_mask(None, None, None, None, 4, 100, 5, None) == 1   # topk w/o sim
    assert lib.coclr_gemm(None, 1, 1, None, 1, 1, None, 1, None, 0, 4, 4, 1.0, 0, 0, 1, None,
                          None) == 1
    with pytest.raises(ValueError):
        ops.ConvGeom(1, 3, 8, (2, 4, 4), (7, 1, 1), (1, 1, 1), (0, 0, 0))      # empty output
    with pytest.raises(ValueError):
        ops.PoolGeom(1, 3, (1, 1, 1), (2, 2, 2), (2, 2, 2), (0, 0, 0))
