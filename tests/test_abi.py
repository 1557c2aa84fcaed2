"""CPU checks of the drop-in boundary: libdae_hip.so loads without a GPU, exports every symbol that
include/dae_hip.h declares, and the ctypes table in _lib.py covers exactly that set."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dae_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(dae_[a-z0-9_]+)\s*\(", src))


@pytest.mark.parametrize("fmt", ["bf16", "f16"])
def test_library_exports_every_declared_symbol(fmt):
    """Both builds of the library (bf16 storage: libdae_hip.so; fp16 storage: libdae_hip_f16.so -- the same sources, csrc/dae_common.h DAE_F16)."""
    from dae_rnn_news_recommendation_amd import _lib
    lib = _lib.load(fmt)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in dae_hip.h but not exported by the {fmt} build"
    assert names == set(_lib.SIGNATURES), names ^ set(_lib.SIGNATURES)
    m = re.search(r"#define DAE_ABI_VERSION (\d+)", open(HEADER).read())
    assert lib.dae_abi_version() == int(m.group(1)) == _lib.ABI_VERSION
    assert lib.dae_storage_format() == (1 if fmt == "f16" else 0)
    assert lib.dae_pad(800) == 896 and lib.dae_pad(10000) == 10112 and lib.dae_pad(128) == 128


def test_precision_table_names_existing_builds():
    from dae_rnn_news_recommendation_amd import _lib
    assert _lib.AUTO_PRECISION in _lib.PRECISIONS
    for name, (fmt, dtype, terms) in _lib.PRECISIONS.items():
        assert fmt in _lib.LIB_PATHS and dtype in (_lib.BF16, _lib.F32, _lib.BF16X3), name
        assert terms is None or 0 <= terms <= _lib.X3T_ALL
    m = re.search(r"#define DAE_WAIT_DW_CREATED (\d+)", open(HEADER).read())
    assert int(m.group(1)) == _lib.WAIT_DW_CREATED and _lib.WAIT_DW_CREATED not in (0, 1, 2)      # outside the error codes


def test_ctypes_arity_matches_header():
    from dae_rnn_news_recommendation_amd import _lib
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", src, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(args), (name, n, len(args))


def test_struct_sizes_match_c():
    """sizeof() of the three ABI structs as the C compiler lays them out == ctypes' layout."""
    from dae_rnn_news_recommendation_amd import _lib
    prog = r'''
#include <stdio.h>
#include "dae_hip.h"
int main(void){ printf("%zu %zu %zu\n", sizeof(dae_config), sizeof(dae_buffers), sizeof(dae_step)); return 0; }
'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c"); exe = os.path.join(d, "s")
        open(c, "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(_lib.dae_config), ctypes.sizeof(_lib.dae_buffers), ctypes.sizeof(_lib.dae_step)]


def test_argument_errors_are_reported_without_a_gpu():
    from dae_rnn_news_recommendation_amd import _lib
    lib = _lib.load()
    cfg = _lib.dae_config(); cfg.n_features = 0
    plan = ctypes.c_void_p()
    assert lib.dae_plan_create(ctypes.byref(cfg), ctypes.byref(plan)) != 0
    assert b"plan_create" in lib.dae_last_error()
    cfg = _lib.dae_config(10000, 500, 800, 0, 1, 1, 0, 0, 1, 0, 0, 0, 0, 0.1, 0.5, 1.0)
    assert lib.dae_plan_create(ctypes.byref(cfg), ctypes.byref(plan)) == 0
    assert lib.dae_plan_workspace_bytes(plan) > 100 << 20
    info = (ctypes.c_int32 * 8)()
    assert lib.dae_plan_info(plan, info) == 0 and list(info)[:3] == [10112, 512, 896]
    lib.dae_plan_destroy(plan)
    # the split mode's lo images are allocated per product term: the fp16 build's default (two W terms) needs ~70 MB less workspace than all terms
    l16 = _lib.load("f16")
    sizes = {}
    for terms in (None, _lib.X3T_ALL):
        cfg = _lib.dae_config(10000, 500, 800, 2, 1, 1, 0, 0, 1, 0, 0, 0, 0, 0.1, 0.5, 1.0)
        assert l16.dae_plan_create(ctypes.byref(cfg), ctypes.byref(plan)) == 0
        if terms is not None:
            assert l16.dae_plan_set_option(plan, b"x3_terms", terms) == 0
        assert l16.dae_plan_info(plan, info) == 0
        sizes[terms] = (l16.dae_plan_workspace_bytes(plan), info[7])
        l16.dae_plan_destroy(plan)
    assert sizes[_lib.X3T_ALL][0] - sizes[None][0] > 60 << 20, sizes
    assert (sizes[None][1] >> 1) & _lib.X3T_ALL == (1 | 4) and (sizes[None][1] >> 16) == 13      # default terms; op_scale 2^13 at B = 800
    assert l16.dae_plan_set_option(None, b"x3_terms", 1) != 0                     # null plan: an argument error, not a crash


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the product package (or the CLI) may import it."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "dae_rnn_news_recommendation_amd")):
        for f in files:
            if f.endswith(".py") and re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(base, f)).read(), flags=re.M):
                bad.append(f)
    for f in ("main_autoencoder.py", "main_autoencoder_triplet.py"):
        pth = os.path.join(ROOT, f)
        if os.path.exists(pth) and re.search(r"^\s*(from|import)\s+oracle\b", open(pth).read(), flags=re.M):
            bad.append(f)
    assert not bad, bad


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dae_rnn_news_recommendation_amd.engine import Engine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Engine(100, 10, 16)


def test_every_kernel_launch_goes_through_the_timed_launcher():
    """bench.py's per-kernel table relies on dae_plan_profile mode 3 seeing EVERY launch of a step: the library launches kernels through
    DAE_LAUNCH (csrc/dae_common.h) alone -- no bare hipLaunchKernelGGL / <<< >>> in the kernel sources -- and the profile call accepts the
    three documented modes without a GPU (argument handling only: a null plan is an argument error in each)."""
    csrc = os.path.join(ROOT, "dae_rnn_news_recommendation_amd", "csrc")
    bad, launches = [], 0
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".h", ".cpp")):
            continue
        text = open(os.path.join(csrc, f)).read()
        launches += len(re.findall(r"\bDAE_LAUNCH\(", text))
        if f == "dae_common.h":
            assert text.count("hipLaunchKernelGGL(") == 1 and "hipExtLaunchKernelGGL(" in text      # the macro's two arms
            continue
        if re.search(r"\bhipLaunchKernelGGL\(|<<<|\bhipExtLaunchKernelGGL\(|\bhipModuleLaunchKernel\(", text):
            bad.append(f)
    assert not bad, bad
    assert launches >= 60, launches
    from dae_rnn_news_recommendation_amd import _lib
    lib = _lib.load()
    for mode in (0, 1, 2, 3):
        assert lib.dae_plan_profile(None, mode) != 0
        assert b"null plan" in lib.dae_last_error()
