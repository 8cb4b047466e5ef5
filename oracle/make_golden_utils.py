"""Fixtures for ``duwu.utils``' list helpers, produced by the REFERENCE's own functions (test infrastructure; run in the build
container only -- /root/reference does not exist on the GPU box).  The reference module imports omegaconf / lightning / hydra at
import time, none of which its pure-Python list helpers use, so those function definitions are compiled from the file's AST on
their own.  Output: tests/golden/utils_helpers.json (inputs + expected outputs; no reference source text).

    python -m oracle.make_golden_utils
"""
import ast
import inspect
import json
import os
import typing

REF = "/root/reference/src/duwu/utils/__init__.py"
KEEP = {"uniq", "default", "remove_none", "balance_sharding_index", "balance_sharding", "balance_sharding_max_size",
        "truncate_or_pad_to_length", "repeat_last", "cycling", "uniform_expansion", "exists"}


def load_reference():
    tree = ast.parse(open(REF).read())
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in KEEP], type_ignores=[])
    ns = {"isfunction": inspect.isfunction, "Literal": typing.Literal}
    exec(compile(mod, REF, "exec"), ns)
    return ns


def main():
    ref = load_reference()
    out = {"balance_sharding_index": [], "truncate_or_pad_to_length": [], "balance_sharding_max_size": [], "uniq": [], "remove_none": []}
    for total in (0, 1, 5, 7, 16, 50, 77):
        for shards in (1, 2, 3, 4, 8):
            out["balance_sharding_index"].append([total, shards, [list(t) for t in ref["balance_sharding_index"](total, shards)]])
    for n in (1, 2, 3, 5):
        xs = [f"p{i}" for i in range(n)]
        for tgt in (0, 1, 2, 4, 7, 12):
            for mode in ("repeat_last", "cycling", "uniform_expansion", "unknown"):
                out["truncate_or_pad_to_length"].append([xs, tgt, mode, ref["truncate_or_pad_to_length"](list(xs), tgt, mode)])
        for ms in (1, 2, 3):
            out["balance_sharding_max_size"].append([xs, ms, [list(x) for x in ref["balance_sharding_max_size"](xs, ms)]])
    out["uniq"].append([[3, 1, 3, 2, 1], list(ref["uniq"]([3, 1, 3, 2, 1]))])
    out["remove_none"].append([[1, None, 2, None], ref["remove_none"]([1, None, 2, None])])
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "utils_helpers.json")
    json.dump(out, open(path, "w"), indent=0)
    print(path, sum(len(v) for v in out.values()), "cases")


if __name__ == "__main__":
    main()
