"""The measurement scripts under tools/ and examples/ are not run by the suite (they need a GPU), but they must not rot: every one
parses, and every name it imports from ``specforge_amd`` / ``bench`` exists (an ingest refactor once left tools/ingest_procs.py
calling a removed signature until its next GPU run)."""
import ast
import glob
import importlib
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPTS = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "examples", "*.py")) + [os.path.join(ROOT, "bench.py")])


@pytest.mark.parametrize("path", SCRIPTS, ids=[os.path.relpath(p, ROOT) for p in SCRIPTS])
def test_script_parses_and_its_package_imports_exist(path):
    tree = ast.parse(open(path).read(), filename=path)
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module and (node.module == "bench" or node.module.split(".")[0] == "specforge_amd"):
            if node.module.endswith("reference_plugin"):
                continue            # importable only next to a SpecForge checkout
            mod = importlib.import_module(node.module)
            for alias in node.names:
                assert hasattr(mod, alias.name) or importlib.util.find_spec(f"{node.module}.{alias.name}") is not None, \
                    f"{os.path.relpath(path, ROOT)}: {node.module} has no {alias.name}"
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id == "ops":
            from specforge_amd import ops

            assert hasattr(ops, node.attr), f"{os.path.relpath(path, ROOT)}: ops.{node.attr} does not exist"
