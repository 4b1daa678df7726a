"""The product path stands on the HIP library alone: nothing under bevgen_amd/ reaches into oracle/ (test infrastructure), bench.py touches it only inside its
cpu_baseline leg, __graft_entry__ only inside smoke() / build(), and a missing library is an error, not a fallback."""
import ast
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py_files(top):
    for d, _, files in os.walk(top):
        if "__pycache__" in d:
            continue
        for f in files:
            if f.endswith(".py"):
                yield os.path.join(d, f)


def _oracle_imports(tree):
    """(node, enclosing function name or None) of every import that names the oracle package."""
    hits = []

    def visit(node, fn):
        for child in ast.iter_child_nodes(node):
            inner = child.name if isinstance(child, (ast.FunctionDef, ast.AsyncFunctionDef)) else fn
            if isinstance(child, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in child.names):
                hits.append((child, fn))
            if isinstance(child, ast.ImportFrom) and (child.module or "").split(".")[0] == "oracle":
                hits.append((child, fn))
            visit(child, inner)

    visit(tree, None)
    return hits


def test_the_package_never_touches_the_oracle():
    for path in _py_files(os.path.join(ROOT, "bevgen_amd")):
        src = open(path).read()
        assert not _oracle_imports(ast.parse(src)), path
        for needle in ("oracle/", "oracle.", "import_module(\"oracle", "import_module('oracle"):
            assert needle not in src.replace("the oracle", "").replace("oracle's", ""), (path, needle)
    for path in (os.path.join(ROOT, "bevgen_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "bevgen_amd", "csrc")) if f.endswith((".hip", ".cpp", ".h"))):
        assert "oracle/" not in open(path).read(), path


def test_bench_and_entry_use_the_oracle_only_as_the_checker():
    hits = _oracle_imports(ast.parse(open(os.path.join(ROOT, "bench.py")).read()))
    assert hits and all(fn == "cpu_baseline" for _, fn in hits), [(ast.dump(n), fn) for n, fn in hits]
    hits = _oracle_imports(ast.parse(open(os.path.join(ROOT, "__graft_entry__.py")).read()))
    # (build() may prepare the checker - "building the checker is not using it" - smoke() runs it against the product)
    assert all(fn in ("smoke", "build") for _, fn in hits), [(ast.dump(n), fn) for n, fn in hits]


def test_a_missing_library_is_an_error_not_a_fallback():
    env = dict(os.environ, BEVGEN_LIB_PATH="/nonexistent/libbevgen_hip.so", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", "from bevgen_amd.runtime import Context; Context(None); print('created')"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "created" not in r.stdout
    assert "not found" in r.stderr and "no CPU fallback" in r.stderr, r.stderr[-1500:]
