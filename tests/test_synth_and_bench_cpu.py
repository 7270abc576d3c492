"""CPU checks of the synthetic-weight helpers and of bench.py's presets (no GPU): the state dicts have the reference
modules' layout (the oracle accepts them and runs), and every workload preset is well formed."""
import ast
import os

import torch

from card_amd.synth import cls_state_dict, guided_state_dict
from oracle import card_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_guided_state_dict_runs_through_the_oracle():
    for H, x_dim, y_dim, cat_x, cat_y in ((128, 13, 1, True, True), (256, 5, 1, True, False), (128, 3, 2, False, True)):
        T = 7
        sd = guided_state_dict(x_dim, T, 5, y_dim=y_dim, hidden=H, cat_x=cat_x, cat_y_pred=cat_y)
        d_in = y_dim + (y_dim if cat_y else 0) + (x_dim if cat_x else 0)
        assert sd["lin1.lin.weight"].shape == (H, d_in) and sd["lin4.weight"].shape == (y_dim, H)
        assert sd["lin2.embed.weight"].shape == (T + 1, H)
        w = orc.GuidedWeights.from_state_dict(sd, cat_x=cat_x, cat_y_pred=cat_y)
        x, y = torch.randn(9, x_dim), torch.randn(9, y_dim)
        eps = orc.eps_theta(w, x, y, y, 3)
        assert eps.shape == (9, y_dim) and torch.isfinite(eps).all()


def test_cls_state_dict_runs_through_the_oracle():
    F, C, T = 128, 10, 5
    sd = cls_state_dict(F, C, T, 1)
    assert sd["lin1.lin.weight"].shape == (F, 2 * C) and sd["unetnorm2.running_var"].shape == (F,)
    w = orc.ClsWeights.from_state_dict(sd)
    xf, y = torch.randn(4, F), torch.softmax(torch.randn(4, C), 1)
    alphas, _, omabs = orc.make_schedule("linear", T, 1e-4, 0.3)
    seq = orc.p_sample_loop(w, xf, y, y, T, alphas, omabs, torch.randn(T, 4, C))
    assert len(seq) == T + 1 and all(torch.isfinite(s).all() for s in seq)


def test_bench_presets_are_well_formed():
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    presets = None
    for node in tree.body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") == "WORKLOADS":
            presets = ast.literal_eval(node.value)
    assert presets and "toy" in presets                     # the default (BASELINE.json configs[1])
    for name, (kind, rows, nz, x_dim, hidden, y_dim, desc) in presets.items():
        assert kind in ("guided", "cls") and rows > 0 and nz > 0 and hidden % 128 == 0 and y_dim >= 1 and desc
        if kind == "cls":
            assert x_dim == hidden                          # x is x_feat
    assert presets["toy"][1] * presets["toy"][2] == 10_000 * 1000


def test_product_schedule_helper_matches_the_oracle():
    from card_amd.synth import make_schedule
    for kind, T, lo, hi in (("linear", 1000, 1e-4, 0.02), ("linear", 50, 1e-4, 0.4), ("cosine", 100, 1e-4, 0.02)):
        for a, b in zip(make_schedule(kind, T, lo, hi), orc.make_schedule(kind, T, lo, hi)):
            assert torch.equal(a, b)


def test_oracle_is_imported_only_where_allowed():
    """oracle/ is test infrastructure: the package and the tools never import it; bench.py only inside its cpu_baseline leg."""
    import re
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for d in ("card_amd", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, d)):
            for f in files:
                if f.endswith(".py"):
                    assert not pat.search(open(os.path.join(dirpath, f)).read()), os.path.join(dirpath, f)
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    for node in ast.walk(main):
        if isinstance(node, (ast.Import, ast.ImportFrom)) and "oracle" in ast.unparse(node):
            seg = ast.get_source_segment(src, node)
            # the only import of the oracle inside main() sits under the `not a.no_cpu_baseline` branch
            line = src[:src.index(seg)].count("\n")
            assert "cpu_baseline" in "\n".join(src.splitlines()[line - 12:line + 12]), seg
