"""Mixed-precision bit-width search on top of the fast quantized forward.

Counterpart of the inline search in the reference driver (test_quant.py:253-408): Pareto-style sampling of 4/8-bit
configurations under a model-size constraint, ranking by Omega = sum_i hessian_i * distance_i, validation of the best
five, then an evolutionary search (population 25, 8 iterations, 10 mutations + 10 crossovers per iteration) scored by
top-1 from ``validate``.  Every candidate is one ``model(data, bit_config)`` sweep -- thousands of forwards with a
different ``bit_config`` per call, which the engine serves from the per-bit weight copies of the frozen plan.

Kept from the reference, on purpose (same walk for the same ``random`` seed and the same scores):
  * candidates tie the two layers of each pair: ``[8] + [b for b in cfg for _ in range(2)] + [choice]`` (:269-273);
  * Omega indexes ``global_distance[i-1][k]`` with k = index of the bit in ``bit_choice`` = {0, 1}, i.e. it reads the
    uint3/uint4 weight MSEs of the calibration bookkeeping (models/ptq/layers.py:151-170), not int4/int8 (:293-296);
  * a rejected mutation/crossover child is still appended with the PREVIOUS candidate's score (:355-361, :377-383).
The per-layer Hessian sensitivities (``mean_hessian``; pyhessian in the reference, out of scope here, and undefined as
shipped: test_quant.py:186) are an input; uniform sensitivities reduce Omega to the summed weight distance.
"""
import random


def model_size(FLOPs, bit_config):
    return sum(FLOPs[i] * bit_config[i] for i in range(len(FLOPs)))


def pareto_candidates(FLOPs, n_layers_gd, rng, bit_choice=(4, 8), max_configs=50, slack=1.1):
    """test_quant.py:262-284"""
    bit_choice = list(bit_choice)
    constraint = slack * sum(FLOPs[i] * 4 for i in range(len(FLOPs)))
    bit_list = []
    for _ in range(2 ** min(n_layers_gd, 24)):
        cfg = [rng.choice(bit_choice) for _ in range(len(FLOPs) // 2 - 1)]
        new = [max(bit_choice)] + [b for b in cfg for _ in range(2)] + [rng.choice(bit_choice)]
        if not model_size(FLOPs, new) > constraint and new not in bit_list:
            bit_list.append(new)
        if len(bit_list) > max_configs:
            break
    return bit_list, constraint


def omega_rank(bit_list, global_distance, mean_hessian, bit_choice=(4, 8)):
    """test_quant.py:286-318: [[bit_config, omega], ...] sorted by omega."""
    bit_choice = list(bit_choice)
    out = []
    for cfg in bit_list:
        sel = []
        for i, bit in enumerate(cfg):
            if i == 0:
                continue
            for k, choice in enumerate(bit_choice):
                if choice == bit:
                    sel.append(global_distance[i - 1][k])
                    break
        omega = [mean_hessian[i] * sel[i] for i in range(len(cfg) - 1)]
        out.append([cfg, float(sum(omega))])
    out.sort(key=lambda x: x[-1])
    return out


def evolutionary_search(score_fn, omega_list, FLOPs, constraint, rng, bit_choice=(4, 8), pop_size=25, evo_iter=8,
                        mutate_size=10, mutate_prob=0.5, crossover_size=10, crossover_prob=0.5, log=print):
    """test_quant.py:340-408.  ``score_fn(bit_config) -> top-1``."""
    bit_choice = list(bit_choice)
    parent = [[omega_list[i][0], score_fn(omega_list[i][0])] for i in range(min(pop_size, len(omega_list)))]
    parent.sort(key=lambda x: x[-1], reverse=True)
    val = parent[0][1] if parent else 0.0
    for evo in range(evo_iter):
        children, seen = [], []
        while True:
            old = rng.choice(parent)[0]
            new = [b if rng.random() < mutate_prob else rng.choice(bit_choice) for b in old]
            if not model_size(FLOPs, new) > constraint and new not in seen:
                val = score_fn(new)
            seen.append(new)
            children.append([new, val])
            if len(seen) > mutate_size:
                break
        seen = []
        while True:
            a, b = rng.choice(parent)[0], rng.choice(parent)[0]
            if a == b:
                if len(parent) < 2:
                    break
                continue
            new = [x if rng.random() < crossover_prob else y for x, y in zip(a, b)]
            if not model_size(FLOPs, new) > constraint and new not in seen:
                val = score_fn(new)
            seen.append(new)
            children.append([new, val])
            if len(seen) > crossover_size:
                break
        for child in children:
            if child[1] > parent[-1][1]:
                parent.append(child)
        parent.sort(key=lambda x: x[-1], reverse=True)
        parent = parent[:pop_size]
        log('Evolotionary iteration: ', evo)
    return parent


def mixed_precision_search(score_fn, FLOPs, global_distance, mean_hessian=None, seed=0, log=print, **kw):
    """the whole block test_quant.py:253-408; returns (pareto-ranked list, final population)."""
    assert len(FLOPs) - 1 == len(global_distance)
    if mean_hessian is None:
        mean_hessian = [1.0] * len(global_distance)
    assert len(mean_hessian) == len(global_distance)
    rng = random.Random(seed)
    log('Pareto Frontier.......')
    pk = {k: v for k, v in kw.items() if k in ('max_configs', 'slack')}
    bit_list, constraint = pareto_candidates(FLOPs, len(global_distance), rng, **pk)
    ranked = omega_rank(bit_list, global_distance, mean_hessian)
    log('Hessien-Based Validating...')
    for i in range(min(5, len(ranked))):
        log(ranked[i][0])
        score_fn(ranked[i][0])
    log('Start Evolutionary.......')
    evo = {k: v for k, v in kw.items() if k in ('pop_size', 'evo_iter', 'mutate_size', 'mutate_prob', 'crossover_size', 'crossover_prob')}
    return ranked, evolutionary_search(score_fn, ranked, FLOPs, constraint, rng, log=log, **evo)
