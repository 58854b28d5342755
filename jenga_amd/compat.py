"""Opt-in import shim: the reference's own module paths resolve to jenga_amd, so `jenga_hyvideo.py`, `jenga_wan.py`,
`jenga_hyvideo_multigpu.py` and the reference's model files import the hot path unchanged.

    python -m jenga_amd.compat jenga_hyvideo.py --video-size 720 1280 ...      (install the hook, then run the script)
    import jenga_amd.compat; jenga_amd.compat.install()                        (same, from code)

`compat/` (repo root) holds one small re-export file per reference module on the path:
hyvideo/modules/{attention_block_triton_diffres, attenion, posemb_layers, norm_layers, xdit_ring_atten}.py, their
hyvideo_i2v / wan counterparts, gilbert.py and xfuser/core/{distributed, long_ctx_attention}.py.  With the reference checkout on sys.path its
packages (`hyvideo`, `wan`, ...) are regular packages and win over any directory added later, so a plain PYTHONPATH entry
cannot replace single submodules; install() therefore puts a finder in front of sys.meta_path that serves exactly the
names listed in ALIASES from compat/ and leaves every other import alone.  Parent packages that cannot be imported at
all (no reference checkout: tests on the GPU box) are served as namespace packages over compat/ by a second finder at the
END of sys.meta_path.  Nothing is patched at run time and no reference text is stored."""
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys

COMPAT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "compat")

# module name the reference imports -> file under compat/
ALIASES = {
    "gilbert": "gilbert.py",
    "hyvideo.modules.attention_block_triton_diffres": "hyvideo/modules/attention_block_triton_diffres.py",
    "hyvideo.modules.attenion": "hyvideo/modules/attenion.py",
    "hyvideo.modules.posemb_layers": "hyvideo/modules/posemb_layers.py",
    "hyvideo.modules.norm_layers": "hyvideo/modules/norm_layers.py",
    "hyvideo.modules.xdit_ring_atten": "hyvideo/modules/xdit_ring_atten.py",
    "hyvideo_i2v.modules.attention_block_triton_diffres": "hyvideo_i2v/modules/attention_block_triton_diffres.py",
    "hyvideo_i2v.modules.attenion": "hyvideo_i2v/modules/attenion.py",
    "hyvideo_i2v.modules.posemb_layers": "hyvideo_i2v/modules/posemb_layers.py",
    "hyvideo_i2v.modules.norm_layers": "hyvideo_i2v/modules/norm_layers.py",
    "wan.modules.attention_block_triton_diffres": "wan/modules/attention_block_triton_diffres.py",
}
# served only when the real package is absent (xfuser is a third-party dependency of the reference)
OPTIONAL_ALIASES = {"xfuser.core.distributed": "xfuser/core/distributed.py",
                    "xfuser.core.long_ctx_attention": "xfuser/core/long_ctx_attention.py"}


class _AliasFinder(importlib.abc.MetaPathFinder):
    def __init__(self, root):
        self.root = root

    def find_spec(self, name, path=None, target=None):
        rel = ALIASES.get(name)
        if rel is None:
            return None
        return importlib.util.spec_from_file_location(name, os.path.join(self.root, rel))


class _FallbackFinder(importlib.abc.MetaPathFinder):
    """Last in sys.meta_path: parent packages (and the optional aliases) nobody else could find."""

    def __init__(self, root):
        self.root = root
        names = list(ALIASES) + list(OPTIONAL_ALIASES)
        self.parents = {".".join(n.split(".")[:i]) for n in names for i in range(1, n.count(".") + 1)}

    def find_spec(self, name, path=None, target=None):
        if name in OPTIONAL_ALIASES:
            return importlib.util.spec_from_file_location(name, os.path.join(self.root, OPTIONAL_ALIASES[name]))
        if name in self.parents:
            spec = importlib.machinery.ModuleSpec(name, None, is_package=True)
            spec.submodule_search_locations = [os.path.join(self.root, *name.split("."))]
            return spec
        return None


_installed = []


def install(compat_dir=None):
    """Idempotent.  Returns the list of module names that now resolve to jenga_amd."""
    if not _installed:
        root = compat_dir or COMPAT_DIR
        if not os.path.isdir(root):
            raise FileNotFoundError(f"{root}: the compat/ directory of the jenga_amd repository is missing")
        first, last = _AliasFinder(root), _FallbackFinder(root)
        sys.meta_path.insert(0, first)
        sys.meta_path.append(last)
        _installed.extend([first, last])
        for name in ALIASES:            # a module imported before install() would keep shadowing the alias
            sys.modules.pop(name, None)
    return sorted(ALIASES)


def uninstall():
    for f in _installed:
        if f in sys.meta_path:
            sys.meta_path.remove(f)
    for name in list(ALIASES) + list(OPTIONAL_ALIASES):
        sys.modules.pop(name, None)
    _installed.clear()


def main(argv=None):
    import runpy
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit("usage: python -m jenga_amd.compat <reference script.py> [its arguments]")
    install()
    script = argv[0]
    sys.argv = argv
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
