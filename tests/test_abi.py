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


def test_host_driver_library_exports_the_reference_stage_names():
    import re
    from minicom_amd import pipeline
    lib = pipeline.load_host_library()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "mcom_host.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(mcomh_[a-z0-9_]+)\s*\(", txt)))
    assert declared == sorted(pipeline.HOST_ABI_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_host_driver_fails_loudly_without_a_gpu():
    import numpy as np
    import torch
    import pytest
    from minicom_amd.pipeline import Pipeline
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(minicom_amd.McomError):
        Pipeline(np.full((4, 100), ord("A"), dtype=np.uint8))
    with pytest.raises(minicom_amd.McomError):
        minicom_amd.Context(0)


def test_dict_layout_matches_oracle():
    import oracle
    from minicom_amd.hip import dict_layout
    for L in (37, 64, 80, 81, 100, 101, 150, 151, 256):
        for nd in (0, 1, 2, 3, 5):
            st, en = dict_layout(L, nd)
            os_, oe = oracle.dict_layout(L, nd)
            assert st == os_.tolist() and en == oe.tolist()


def test_null_context_is_rejected_without_touching_the_gpu():
    lib = minicom_amd.load_library()
    assert lib.mcom_sync(None) == -1
    assert lib.mcom_last_error(None) == b"null context"
    assert lib.mcom_sketch_reads(None, None, None, 0, 100, 31, 0, None) == -1


def test_product_does_not_import_the_oracle():
    """nothing under minicom_amd/ may import, name a path under, or execute anything of oracle/ (oracle/_ref included)"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for d, _, files in os.walk(os.path.join(root, "minicom_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"import oracle|from oracle|mcom_oracle|mcomo_|oracle/|[\"']oracle[\"']|[\"'/]_ref[\"'/]|minicom_bin|refdump", txt):
                    bad.append(f)
    assert not bad, bad
