"""The oracle is test infrastructure: nothing in the product (cuda-pathtrace_amd/, include/) may
include, link, import or execute anything under oracle/, and the product has no CPU fallback."""
import os
import re
import subprocess

from conftest import ROOT

PRODUCT_DIRS = ["cuda-pathtrace_amd", "include"]


def _product_sources():
    for d in PRODUCT_DIRS:
        for base, dirs, files in os.walk(os.path.join(ROOT, d)):
            dirs[:] = [x for x in dirs if x not in ("__pycache__", "alt")]
            for f in files:
                if f.endswith((".h", ".hip", ".cpp", ".c", ".py", "Makefile")):
                    yield os.path.join(base, f)


def test_product_never_references_the_oracle():
    offenders = []
    pat = re.compile(r"(#include[^\n]*oracle|import\s+oracle|load_oracle|libpt_oracle|pto_[a-z_]+\s*\(|/oracle/)")
    for path in _product_sources():
        text = open(path, errors="ignore").read()
        for m in pat.finditer(text):
            line = text[: m.start()].count("\n") + 1
            offenders.append(f"{os.path.relpath(path, ROOT)}:{line}: {m.group(0)}")
    assert not offenders, offenders


def test_product_library_does_not_link_the_oracle(pt):
    out = subprocess.run(["ldd", pt.LIB_PATH], capture_output=True, text=True).stdout
    assert "pt_oracle" not in out
    syms = subprocess.run(["nm", "-D", "--defined-only", pt.LIB_PATH], capture_output=True, text=True).stdout
    assert "pto_" not in syms


def test_bench_uses_the_oracle_only_for_the_cpu_baseline():
    src = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"load_oracle\(\)", src)]
    assert len(uses) == 1
    # the single use sits in the cpu_baseline branch, after the timed region has been reported
    assert src.index('out["cpu_baseline"] = cpu_baseline(') > uses[0] > src.index("renderer, elapsed, kernel_s = measure(rng_mode")
