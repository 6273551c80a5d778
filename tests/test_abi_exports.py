"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol that
include/rc_abi.h declares (no compute calls here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rc_abi.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rc_[a-z_0-9]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from nrc_amd import rc_ext
    return rc_ext.load_library()


def test_header_symbols_are_exported(lib):
    names = _declared()
    assert "rc_render_rays" in names and "rc_create" in names and len(names) >= 12
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_covers_header(lib):
    from nrc_amd import rc_ext
    assert set(_declared()) == set(rc_ext.EXPORTS)
    assert lib.rc_abi_version() == rc_ext.RC_ABI_VERSION


def test_config_struct_layout_matches_c():
    """sizeof(rc_config) etc. as seen by a C compiler == the ctypes mirror."""
    from nrc_amd import rc_ext
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "rc_abi.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %d\n", sizeof(rc_config), sizeof(rc_grid_config), sizeof(rc_tensor_desc),
         sizeof(rc_rays), sizeof(rc_randoms), sizeof(rc_outputs), (int)RC_OUT_COUNT);
  printf("%zu %zu\n", offsetof(rc_config, anneal), offsetof(rc_config, num_resample));
  return 0;
}
'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")], check=True)
        out = subprocess.run([os.path.join(d, "t")], check=True, capture_output=True, text=True).stdout.split()
    sizes = [int(x) for x in out]
    assert sizes[0] == ctypes.sizeof(rc_ext.rc_config)
    assert sizes[1] == ctypes.sizeof(rc_ext.rc_grid_config)
    assert sizes[2] == ctypes.sizeof(rc_ext.rc_tensor_desc)
    assert sizes[3] == ctypes.sizeof(rc_ext.rc_rays)
    assert sizes[4] == ctypes.sizeof(rc_ext.rc_randoms)
    assert sizes[5] == ctypes.sizeof(rc_ext.rc_outputs)
    assert sizes[6] == rc_ext.RC_OUT_COUNT
    assert sizes[7] == rc_ext.rc_config.anneal.offset and sizes[8] == rc_ext.rc_config.num_resample.offset


def test_output_table_matches_header_enum():
    from nrc_amd import rc_ext
    src = open(HEADER).read()
    body = src[src.index("typedef enum {\n  RC_OUT_RGB"):src.index("} rc_output_id;")]
    enum = re.findall(r"RC_OUT_([A-Z_0-9]+)", re.sub(r"/\*.*?\*/", "", body, flags=re.S))
    enum = [e for e in enum if e != "COUNT"]
    assert [e.lower() for e in enum] == [n for n, _ in rc_ext.OUTPUTS]


def test_product_fails_loudly_without_library(monkeypatch, tmp_path):
    from nrc_amd import rc_ext
    monkeypatch.setattr(rc_ext, "_LIB", None)
    monkeypatch.setattr(rc_ext, "library_path", lambda: str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rc_ext.load_library()


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "neural-radiance-caching_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "from oracle" not in txt and "import oracle" not in txt, f
