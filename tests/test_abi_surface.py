"""CPU checks of the drop-in boundary: the shared library loads and exports every symbol declared in
include/mopt_hip.h; the ctypes structs match the C layout; the product package never imports the oracle."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "mopt_hip.h")).read()
    return sorted(set(re.findall(r"\b(mopt_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    so = os.path.join(ROOT, "multioptpy_amd", "libmopt_hip.so")
    if not os.path.exists(so):
        import __graft_entry__ as ge
        ge.build()
    lib = C.CDLL(so)
    names = _declared()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.mopt_abi_version.restype = C.c_int
    assert lib.mopt_abi_version() == 1


def test_struct_layout_matches_header(tmp_path):
    from multioptpy_amd import _abi
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mopt_hip.h"\nint main(){printf("%zu %zu %zu %zu\\n", '
                   'sizeof(mopt_rsirfo_params), sizeof(mopt_rsirfo_state), offsetof(mopt_rsirfo_params, max_micro_cycles), '
                   'offsetof(mopt_rsirfo_state, alpha));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    a, b, c, d = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    src2 = tmp_path / "sz2.c"
    src2.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mopt_hip.h"\nint main(){printf("%zu %zu %zu %zu\\n", '
                    'sizeof(mopt_rsprfo_params), sizeof(mopt_rsprfo_state), offsetof(mopt_rsprfo_params, max_micro_cycles), '
                    'offsetof(mopt_rsprfo_state, step_norm));return 0;}\n')
    exe2 = tmp_path / "sz2"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src2), "-o", str(exe2)])
    a2, b2, c2, d2 = [int(v) for v in subprocess.check_output([str(exe2)]).split()]
    assert a2 == C.sizeof(_abi.RsprfoParams) and b2 == C.sizeof(_abi.RsprfoState)
    assert c2 == _abi.RsprfoParams.max_micro_cycles.offset and d2 == _abi.RsprfoState.step_norm.offset
    assert a == C.sizeof(_abi.RsirfoParams)
    assert b == C.sizeof(_abi.RsirfoState)
    assert c == _abi.RsirfoParams.max_micro_cycles.offset
    assert d == _abi.RsirfoState.alpha.offset


def test_rule_enum_matches_dispatch_order():
    from multioptpy_amd import _abi
    txt = open(os.path.join(ROOT, "include", "mopt_hip.h")).read()
    body = txt[txt.index("MOPT_RULE_FLOWCHART = 0"):txt.index("MOPT_RULE_COUNT")]
    enum = [m.lower()[len("mopt_rule_"):] for m in re.findall(r"MOPT_RULE_[A-Z0-9_]+", body)]
    assert enum == _abi.RULE_KEYS
    assert _abi.rule_from_method("rsirfo_block_cfd_fsb_weighted") == _abi.RULE_KEYS.index("block_cfd_fsb_weighted")
    assert _abi.rule_from_method("RFO_FSB") == _abi.RULE_KEYS.index("fsb")
    assert _abi.rule_from_method("something_else") == 0


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "multioptpy_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt or f == "__init__.py" and "oracle" not in txt.replace("the oracle", ""), (dp, f)


def test_missing_device_fails_loudly():
    """No GPU in the build container: creating a context must raise, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from multioptpy_amd import _lib
    with pytest.raises(_lib.MoptError):
        _lib.ctx(0)
