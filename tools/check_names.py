#!/usr/bin/env python3
"""Undefined-global check without third-party linters: for every function scope of the given modules, every name that resolves
as a module global must exist in the imported module's namespace or in builtins.  `python tools/check_names.py bench benchlib.wan ...`
(used by tests/test_bench_cpu.py: most of bench.py only runs on a GPU box, a NameError there costs a GPU session)."""
import builtins
import importlib
import symtable
import sys


def undefined_globals(modname):
    mod = importlib.import_module(modname)
    src = open(mod.__file__).read()
    top = symtable.symtable(src, mod.__file__, "exec")
    bad = []

    def walk(tab):
        for sym in tab.get_symbols():
            if tab.get_type() != "module" and sym.is_global() and sym.is_referenced():
                n = sym.get_name()
                if not hasattr(mod, n) and not hasattr(builtins, n):
                    bad.append((tab.get_name(), tab.get_lineno(), n))
        for ch in tab.get_children():
            walk(ch)
    walk(top)
    return bad


if __name__ == "__main__":
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    rc = 0
    for m in sys.argv[1:]:
        b = undefined_globals(m)
        print(m, "undefined globals:", b or "none")
        rc |= bool(b)
    sys.exit(rc)
