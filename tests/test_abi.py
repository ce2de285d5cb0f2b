"""CPU tests of the C-ABI boundary: the shared library loads, exports every symbol include/mi355x_nk.h declares,
its host-only logic works without a GPU, and it fails loudly (no CPU fallback) when no HIP device exists."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mi355x_nk.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nk_[a-z0-9_]+)\s*\(", src)) - {"nk_residual_fn", "nk_jvp_fn", "nk_vjp_fn",
                                                                       "nk_jacvals_fn", "nk_matvec_fn"})


@pytest.fixture(scope="module")
def lib():
    from nonlinearsolve_jl_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def test_header_symbols_all_exported(lib):
    names = _declared_functions()
    assert len(names) >= 55
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/mi355x_nk.h but not exported: {missing}"


def test_ctypes_table_covers_header(lib):
    from nonlinearsolve_jl_amd import _lib
    assert set(_declared_functions()) == set(_lib.SIGNATURES), set(_declared_functions()) ^ set(_lib.SIGNATURES)


def test_struct_layouts_match_header():
    """sizeof of the ctypes mirrors equals what a C compiler computes for the header's structs."""
    import subprocess
    import tempfile
    from nonlinearsolve_jl_amd import _lib
    prog = r'''
#include <stdio.h>
#include "mi355x_nk.h"
int main(void){printf("%zu %zu %zu %zu %zu %zu\n", sizeof(nk_options), sizeof(nk_stats), sizeof(nk_gmres_info),
  sizeof(nk_trace_entry), sizeof(nk_user_callbacks), sizeof(nk_comm_callbacks)); return 0;}'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, "t")]).split()]
    assert sizes == [C.sizeof(_lib.Options), C.sizeof(_lib.Stats), C.sizeof(_lib.GmresInfo), C.sizeof(_lib.TraceEntry),
                     C.sizeof(_lib.UserCallbacks), C.sizeof(_lib.CommCallbacks)]


def test_version_and_options_default(lib):
    from nonlinearsolve_jl_amd import _lib
    assert lib.nk_version().decode().startswith("mi355x_nk")
    o = _lib.Options()
    assert lib.nk_options_default(C.byref(o)) == 0
    assert o.maxiters == 1000 and o.gmres_restart == 30 and o.gmres_maxiters == 300
    assert (o.ew_eta0, o.ew_eta_max, o.ew_gamma, o.ew_alpha, o.ew_safeguard_threshold) == (0.5, 0.9, 0.9, 2.0, 0.1)
    assert (o.step_threshold, o.shrink_threshold, o.expand_threshold, o.shrink_factor, o.expand_factor) == \
        (1e-4, 0.25, 0.75, 0.25, 2.0)
    assert o.max_shrink_times == 32 and o.patience_steps == 100 and o.max_stalled_steps == 32


@pytest.mark.parametrize("n,g,P", [(1024, 1, 8), (1000, 1, 3), (4096 * 4096, 4096, 8), (10, 1, 10), (7, 1, 2)])
def test_partition_range_host_logic(lib, n, g, P):
    import nonlinearsolve_jl_amd as nls
    ranges = [nls.partition_range(n, g, P, r) for r in range(P)]
    assert ranges[0][0] == 0 and ranges[-1][1] == n
    for (b0, e0), (b1, e1) in zip(ranges[:-1], ranges[1:]):
        assert e0 == b1 and b0 <= e0
    assert all(b % g == 0 and e % g == 0 for b, e in ranges)
    sizes = [e - b for b, e in ranges]
    assert max(sizes) - min(sizes) <= g


def test_partition_range_errors(lib):
    import nonlinearsolve_jl_amd as nls
    with pytest.raises(nls.NKError):
        nls.partition_range(10, 3, 2, 0)      # n not a multiple of the granule
    with pytest.raises(nls.NKError):
        nls.partition_range(10, 1, 2, 2)      # rank out of range


def test_no_cpu_fallback_without_gpu(lib):
    """On a box without a HIP device every compute entry point is unreachable: context creation fails loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import nonlinearsolve_jl_amd as nls
    with pytest.raises(nls.NKError, match="no HIP device|no CPU fallback|hip"):
        nls.Context(device=0)
    with pytest.raises(nls.NKError):
        nls.Bratu2D(16)


def test_product_never_imports_oracle():
    """The product package and bench/entry 'value' path must not route through oracle/ (only the checker legs may)."""
    pkg = os.path.join(ROOT, "nonlinearsolve.jl_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "nk_oracle" not in txt, f


def test_ensemble_residuals_compile_for_gfx950_without_a_gpu():
    """hiprtc cross-compiles the user residual + dual-number Jacobian + LU kernel for gfx950 (no device needed);
    a broken source is reported through nk_last_error, not a crash."""
    import ctypes as C
    import ensemble_sources as E
    from nonlinearsolve_jl_amd import _lib as L
    nb = C.c_int64()
    for src, n, npar, flags in [(E.QUADRATIC, 3, 3, 0), (E.P2, 4, 4, 0), (E.TRIG_WITH_JAC, 3, 3, 1), (E.QUADRATIC, 24, 24, 0)]:
        assert L.lib().nk_batch_compile_check(src.encode(), n, npar, flags, C.byref(nb)) == 0, L.lib().nk_last_error()
        assert nb.value > 1000
    bad = b"template <typename T> __device__ void nk_f(const T *u, const double *p, T *f) { f[0] = nope; }"
    assert L.lib().nk_batch_compile_check(bad, 1, 0, 0, C.byref(nb)) != 0
    assert b"nope" in L.lib().nk_last_error()
    assert L.lib().nk_batch_compile_check(E.QUADRATIC.encode(), 65, 1, 0, C.byref(nb)) != 0   # n outside 1..64


def test_plain_c_example_compiles_against_the_header(tmp_path):
    """examples/bratu_c2.c — a C99 caller of the ABI, the shape of the `ccall` binding — builds with -Werror and links."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "bratu_c2"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "bratu_c2.c"),
                           "-L", os.path.join(root, "nonlinearsolve.jl_amd", "lib"), "-lmi355x_nk", "-lm", "-o", str(exe)])
    assert exe.exists()


def test_linsolve_seam_example_compiles_against_the_header(tmp_path):
    """examples/linsolve_seam.c — seam 1 replayed in C99 (set A every step → update tolerances → solve!, device-pointer
    callback operator) — builds with -Werror and links."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "linsolve_seam"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "linsolve_seam.c"),
                           "-L", os.path.join(root, "nonlinearsolve.jl_amd", "lib"), "-lmi355x_nk", "-lm", "-o", str(exe)])
    assert exe.exists()


def test_julia_binding_matches_the_abi():
    """julia/src/MI355XNewtonKrylov.jl (the reference-side binding; Julia is not installed here): every symbol it ccalls is
    declared in the header, and its NKOptions mirror lists the fields of nk_options in the same order with the same types."""
    from nonlinearsolve_jl_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    jl = open(os.path.join(root, "julia", "src", "MI355XNewtonKrylov.jl")).read()
    header = open(os.path.join(root, "include", "mi355x_nk.h")).read()
    called = set(re.findall(r"libnk\.(nk_[a-z0-9_]+)\(", jl))
    assert len(called) >= 15
    missing = [s for s in sorted(called) if not re.search(r"\b%s\s*\(" % s, header)]
    assert not missing, missing
    body = jl[jl.index("mutable struct NKOptions"):]
    body = body[:body.index("\nend")]
    fields = re.findall(r"([A-Za-z0-9_]+)::(Int32|Float64)", body)
    jl_types = {"Int32": C.c_int32, "Float64": C.c_double}
    expect = [(n, ty) for n, ty in _lib.Options._fields_]
    assert [(n, jl_types[ty]) for n, ty in fields] == expect
    # the three defects the round-1 review found by reading, and the zero-copy paths it asked for
    assert "cacheval_keepalive" not in jl                      # LinearCache has no such field: keep-alives live in our workspace
    assert "mutable struct OperatorBox" in jl and "pointer_from_objref(w.box)" in jl   # never objref of an immutable tuple
    host_tr = jl[jl.index("function host_matvec_trampoline"):jl.index("function host_prec_trampoline")]
    assert "unsafe_wrap(Array" in host_tr and "nk_gmres_set_operator_fn_host" in jl    # host pointers only for host operators
    assert "nk_gmres_set_operator_jvp" in jl and "DeviceJVP" in jl                     # device JVP bound directly
    assert "memspace(u)::Cint" in jl and "nk_device_alloc" in jl                       # resident vectors: memspace = 1
    assert all(k in jl for k in ("termination_condition", "linesearch", "precs"))      # forwarded into NKOptions
    info = jl[jl.index("struct GMRESInfo"):]
    info = info[:info.index("\nend")]
    assert re.findall(r"([a-z0-9_]+)::(Int32|Float64)", info) == [(n, "Int32" if t is C.c_int32 else "Float64")
                                                                   for n, t in _lib.GmresInfo._fields_]


def _header_prototypes(header):
    """name → (return type, [parameter types]) of every function declared in include/mi355x_nk.h (comments stripped)."""
    txt = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*(?:\s*\*+)?)\s*\b(nk_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        if "typedef" in txt[max(0, m.start() - 12):m.start()]:
            continue
        params = [] if args in ("", "void") else [a.strip() for a in _split_top(args)]
        protos[name] = (ret.strip(), params)
    return protos


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        depth += ch in "([" 
        depth -= ch in ")]"
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def _c_class(decl):
    """coarse class of a C parameter declaration: pointer / int32 / int64 / double / fnptr"""
    d = decl.strip()
    if "(*" in d or re.search(r"\bnk_[a-z_]+_fn\b", d):
        return "ptr"
    if "*" in d or "[" in d:
        return "ptr"
    base = re.sub(r"\b(const|unsigned)\b", "", d).split()
    ty = base[0] if base else ""
    return {"int": "i32", "int32_t": "i32", "int64_t": "i64", "double": "f64", "uint64_t": "i64"}.get(ty, ty)


_JL_CLASS = {"Cint": "i32", "Int32": "i32", "Int64": "i64", "Float64": "f64", "Cstring": "ptr", "Cdouble": "f64"}


def _jl_class(ty):
    ty = ty.strip()
    return "ptr" if ty.startswith("Ptr{") or ty.startswith("Ref{") else _JL_CLASS.get(ty, ty)


def test_every_julia_ccall_matches_the_header_signature():
    """Every `@ccall libnk.f(arg::T, …)::R` of the Julia binding AND of its AMDGPU extension against the prototype in
    include/mi355x_nk.h: the function exists, takes that many arguments, and every argument / the return value has the matching
    C type class (pointer, 32-bit int, 64-bit int, double). Julia is not installed here: this is what stands between the
    binding and an ABI drift."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "mi355x_nk.h")).read()
    protos = _header_prototypes(header)
    assert len(protos) >= 110 and "nk_gmres_solve" in protos and "nk_precond_create_ilu0" in protos
    ncalls = 0
    for rel in ("julia/src/MI355XNewtonKrylov.jl", "julia/ext/MI355XNewtonKrylovAMDGPUExt.jl"):
        jl = open(os.path.join(root, rel)).read()
        for m in re.finditer(r"@ccall\s*\(?\s*libnk\.(nk_[a-z0-9_]+)\(", jl):
            name = m.group(1)
            i, depth = m.end(), 1
            while depth:                      # the balanced argument list
                depth += jl[i] == "("
                depth -= jl[i] == ")"
                i += 1
            args = _split_top(jl[m.end():i - 1])
            ret = re.match(r"\s*::\s*([A-Za-z0-9_{}]+)", jl[i:])
            assert name in protos, f"{rel}: {name} is not declared in the header"
            cret, cparams = protos[name]
            assert len(args) == len(cparams), f"{rel}: {name} called with {len(args)} arguments, the header declares {len(cparams)}: {cparams}"
            for a, cp in zip(args, cparams):
                jt = a.rsplit("::", 1)[1] if "::" in a else None
                assert jt is not None, f"{rel}: {name}: argument `{a.strip()}` carries no type"
                assert _jl_class(jt) == _c_class(cp), f"{rel}: {name}: `{a.strip()}` vs `{cp}`"
            assert ret is not None and _jl_class(ret.group(1)) == _c_class(cret + " x"), f"{rel}: {name}: return type"
            ncalls += 1
    assert ncalls >= 50


def test_julia_package_manifest_declares_what_the_sources_load():
    """julia/ is a package (Next #5): Project.toml with name / uuid / [deps] / [weakdeps] / [extensions] / [compat]
    after the reference's own manifest (/root/reference/Project.toml:32-68), the module under src/, the AMDGPU extension under
    ext/. Every package a `using` / `import` of the two files names is declared ([deps] for src/, [deps] ∪ [weakdeps] ∪ the
    package itself for ext/), every dependency has a [compat] entry, the extension's trigger is a weak dependency, and the
    UUIDs that can be checked against the reference's manifests agree with them."""
    import tomli
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proj = tomli.load(open(os.path.join(root, "julia", "Project.toml"), "rb"))
    assert proj["name"] == "MI355XNewtonKrylov" and re.fullmatch(r"[0-9a-f]{8}(-[0-9a-f]{4}){3}-[0-9a-f]{12}", proj["uuid"])
    deps, weak, ext = proj["deps"], proj["weakdeps"], proj["extensions"]
    assert os.path.exists(os.path.join(root, "julia", "src", proj["name"] + ".jl"))
    for name, trig in ext.items():
        assert os.path.exists(os.path.join(root, "julia", "ext", name + ".jl"))
        for t in ([trig] if isinstance(trig, str) else trig):
            assert t in weak, f"extension {name} is triggered by {t}, which is not a weak dependency"

    def loaded(rel):
        src = open(os.path.join(root, rel)).read()
        src = re.sub(r"#[^\n]*", "", src)
        out = set()
        for m in re.finditer(r"^\s*(?:using|import)\s+([^\n]+)", src, flags=re.M):
            for part in m.group(1).split(":")[0].split(","):
                out.add(part.strip().split(".")[0])
        return out - {""}
    assert loaded("julia/src/MI355XNewtonKrylov.jl") <= set(deps), loaded("julia/src/MI355XNewtonKrylov.jl") - set(deps)
    ext_loaded = loaded("julia/ext/MI355XNewtonKrylovAMDGPUExt.jl")
    assert ext_loaded <= set(deps) | set(weak) | {proj["name"]}, ext_loaded
    assert "AMDGPU" in ext_loaded
    for d in list(deps) + list(weak):
        assert d in proj["compat"], f"no [compat] entry for {d}"
    assert "julia" in proj["compat"]
    ref = "/root/reference"
    if os.path.isdir(ref):   # the UUIDs the reference's own manifests carry
        known = {}
        for rel in ("Project.toml", "lib/NonlinearSolveBase/Project.toml", "lib/SciMLJacobianOperators/Project.toml"):
            t = tomli.load(open(os.path.join(ref, rel), "rb"))
            known[t["name"]] = t["uuid"]
            for sec in ("deps", "weakdeps", "extras"):
                known.update(t.get(sec, {}))
        for d, u in deps.items():
            assert d in known and known[d] == u, f"[deps] {d}: {u} vs the reference's {known.get(d)}"
        assert proj["compat"]["LinearSolve"] == tomli.load(open(os.path.join(ref, "Project.toml"), "rb"))["compat"]["LinearSolve"]
    # the ADVICE r03 finding: the device trampolines order their work on the stream the library hands over
    extsrc = open(os.path.join(root, "julia", "ext", "MI355XNewtonKrylovAMDGPUExt.jl")).read()
    for tr in ("device_matvec_trampoline", "device_prec_trampoline"):
        body = extsrc[extsrc.index("function " + tr):]
        body = body[:body.index("\nend")]
        assert "on_library_stream(stream)" in body, f"{tr} ignores the stream it is handed"
    assert "hipStreamWaitEvent" in extsrc and "hipEventRecord" in extsrc
