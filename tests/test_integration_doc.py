"""INTEGRATION.md prints the ctypes stub a maintainer of the reference would copy.  A stale stub is a silent ABI break (round 4: `dae_buffers` stopped one
field short of the header, so the library read `grad_lo` from whatever followed the struct).  This test executes the stub's three struct definitions as
written in the document and checks names, order and sizeof against _lib.py (which test_abi.py ties to the C header's layout)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_structs():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\nimport ctypes as C, torch\n(.*?)```", doc, flags=re.S).group(1)
    ns = {"C": C}
    for name in ("dae_config", "dae_buffers", "dae_step"):
        m = re.search(r"^class " + name + r"\(C\.Structure\):.*?(?=^class |^\n|^# )", block, flags=re.S | re.M)
        assert m, name
        exec(m.group(0), ns)           # the document's own text
    return ns


def test_integration_stub_structs_match_the_binding():
    from dae_rnn_news_recommendation_amd import _lib
    ns = _stub_structs()
    for name in ("dae_config", "dae_buffers", "dae_step"):
        doc_t, lib_t = ns[name], getattr(_lib, name)
        assert [f[0] for f in doc_t._fields_] == [f[0] for f in lib_t._fields_], name
        assert [C.sizeof(f[1]) for f in doc_t._fields_] == [C.sizeof(f[1]) for f in lib_t._fields_], name
        assert C.sizeof(doc_t) == C.sizeof(lib_t), name
        for f in lib_t._fields_:
            assert getattr(doc_t, f[0]).offset == getattr(lib_t, f[0]).offset, (name, f[0])


def test_integration_doc_names_existing_symbols_and_values():
    from dae_rnn_news_recommendation_amd import _lib
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for sym in set(re.findall(r"`(dae_[a-z0-9_]+)[`(\[]", doc)):
        if sym in ("dae_config", "dae_buffers", "dae_step", "dae_pad", "dae_last_error", "dae_hip", "dae_rnn_news_recommendation_amd"):
            continue
        base = [s for s in _lib.SIGNATURES if s == sym or s.startswith(sym)]        # `dae_triplet_batch_all[_rows]` style names
        assert base, f"INTEGRATION.md names {sym}, which the library does not export"
    m = re.search(r"`DAE_WAIT_DW_CREATED` \((\d+)", doc)
    assert m and int(m.group(1)) == _lib.WAIT_DW_CREATED
    assert "libdae_hip_f16.so" in doc and _lib.AUTO_PRECISION in doc
