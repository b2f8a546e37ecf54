"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every declared symbol."""
import ctypes
import os

import minicom_amd


def test_library_is_built_and_exports_every_declared_symbol():
    assert os.path.exists(minicom_amd.lib_path()), "build with __graft_entry__.build()"
    lib = minicom_amd.load_library()
    assert len(minicom_amd.ABI_SYMBOLS) >= 9
    for name in minicom_amd.ABI_SYMBOLS:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.mcom_version()


def test_null_context_is_rejected_without_touching_the_gpu():
    lib = minicom_amd.load_library()
    assert lib.mcom_sync(None) == -1
    assert lib.mcom_last_error(None) == b"null context"
    assert lib.mcom_sketch_reads(None, None, None, 0, 100, 31, 0, None) == -1


def test_product_does_not_import_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for d, _, files in os.walk(os.path.join(root, "minicom_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                if "import oracle" in txt or "mcom_oracle" in txt or "mcomo_" in txt:
                    bad.append(f)
    assert not bad, bad
