"""CPU: the mixed-precision search loop (counterpart of test_quant.py:253-408) with a synthetic score function.
The reference's loop is inline in main() and cannot be imported, so there is no oracle: properties only."""
import random


def _setup():
    import diff_vit_amd as dva
    m = dva.deit_tiny_patch16_224(cfg=dva.Config())
    flops = m.flops()
    rng = random.Random(1)
    gd = [[rng.random() * 0.1 + 0.2, rng.random() * 0.05 + 0.05, rng.random() * 0.04 + 0.04, rng.random() * 0.001] for _ in range(len(flops) - 1)]
    return dva, flops, gd


def test_pareto_and_omega():
    dva, flops, gd = _setup()
    S = dva.search
    bit_list, constraint = S.pareto_candidates(flops, len(gd), random.Random(0), slack=1.6, max_configs=40)
    assert len(bit_list) == 41 and all(len(c) == 50 and c[0] == 8 for c in bit_list)
    assert all(S.model_size(flops, c) <= constraint for c in bit_list)
    assert all(c[1 + 2 * j] == c[2 + 2 * j] for c in bit_list for j in range(24))       # layer pairs tied
    ranked = S.omega_rank(bit_list, gd, [1.0] * len(gd))
    assert [r[1] for r in ranked] == sorted(r[1] for r in ranked)
    # index quirk: bit 4 reads distance column 0, bit 8 column 1
    c = ranked[0][0]
    assert abs(ranked[0][1] - sum(gd[i - 1][0 if c[i] == 4 else 1] for i in range(1, 50))) < 1e-9


def test_evolution_is_deterministic_and_monotone():
    dva, flops, gd = _setup()
    calls = []

    def score(cfg):                       # more 8-bit layers late in the network -> higher "accuracy"
        calls.append(tuple(cfg))
        return sum((i + 1) * (b == 8) for i, b in enumerate(cfg)) / 10.0

    r1, p1 = dva.search.mixed_precision_search(score, flops, gd, seed=3, log=lambda *a: None, evo_iter=3, slack=1.6, max_configs=40)
    n1 = len(calls)
    r2, p2 = dva.search.mixed_precision_search(score, flops, gd, seed=3, log=lambda *a: None, evo_iter=3, slack=1.6, max_configs=40)
    assert p1 == p2 and r1 == r2 and len(calls) == 2 * n1
    assert len(p1) == 25 and [x[1] for x in p1] == sorted((x[1] for x in p1), reverse=True)
    assert p1[0][1] >= max(score(c[0]) for c in r1[:25]) - 1e-9
    assert n1 >= 5 + 25 + 3                # top-5 validation, initial population, children
