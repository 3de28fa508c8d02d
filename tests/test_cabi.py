"""The C-ABI library loads here (no GPU) and exports every symbol include/promptir_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "promptir_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pir_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from promptir_amd import _lib

    names = _declared()
    assert len(names) >= 30
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in the header but not exported"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert _lib.lib.pir_arch() == b"gfx950"
    assert _lib.lib.pir_abi_version() == _lib.ABI_VERSION


def test_host_side_argument_checks_need_no_gpu():
    from promptir_amd import _lib

    # NULL pointers / bad sizes are rejected on the host before any launch
    assert _lib.lib.pir_add(None, None, None, 10, None) == -22
    assert _lib.lib.pir_gemm_nt_ws_floats(0, 4, 4, 1, 1) == 0
    assert _lib.lib.pir_gemm_nt_ws_floats(48, 48, 16384, 8, 1) > 0
    g = _lib.GemmNN()
    assert _lib.lib.pir_gemm_nn(ctypes.byref(g), None) == -22


def test_module_has_no_cpu_fallback():
    import pytest
    import torch

    from net.model import PromptIR

    net = PromptIR(decoder=True, num_blocks=[1, 1, 1, 1], num_refinement_blocks=1)
    with pytest.raises(RuntimeError, match="no CPU"):
        net(torch.zeros(1, 3, 64, 64))


def test_state_dict_matches_reference_layout():
    import json

    from net.model import PromptIR

    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_shapes.json")))
    net = PromptIR(decoder=True)
    sd = net.state_dict()
    assert list(sd.keys()) == list(ref["shapes"].keys())
    assert all(list(sd[k].shape) == ref["shapes"][k] for k in sd)
    assert sum(p.numel() for p in net.parameters()) == ref["num_params"] == 35_592_263
