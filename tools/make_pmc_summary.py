"""profiles/pmc_summary[_<model>].json (HBM-side bytes per launch, the `traffic` field of bench.py's roofline) from a per-kernel PMC summary
written by tools/pmc_summary.py.   usage: python tools/make_pmc_summary.py <rXX_pmc_summary.json> <images_per_launch> [model=deit_small]"""
import json, os, sys
src, n_img = sys.argv[1], int(sys.argv[2])
model = sys.argv[3] if len(sys.argv) > 3 else 'deit_small'
d = json.load(open(src))
K = 1024


def find(prefixes):
    for prefix in prefixes:
        ks = [k for k in d if k.startswith(prefix) and d[k].get('dispatches', 0) >= 3 and 'FETCH_SIZE' in d[k] and 'WRITE_SIZE' in d[k]]
        if ks:
            return max(ks, key=lambda k: d[k]['dispatches'])
    return None


def tr(prefixes, fetch_mul):
    k = find(prefixes if isinstance(prefixes, (list, tuple)) else [prefixes])
    if k is None:
        return None
    v = d[k]
    return int(round(fetch_mul * v['FETCH_SIZE'] * K + v['WRITE_SIZE'] * K))


def resid(which):     # proj = even dispatches, fc2 = odd (tools/pmc_summary.py), whichever tile height the launcher picked
    ks = [k for k in d if (k.startswith('k_gemm_dma<6,') or k.startswith('k_gemm_dma<2,')) and k.endswith(which) and 'FETCH_SIZE' in d[k] and 'WRITE_SIZE' in d[k]]
    if not ks:
        return None
    v = d[max(ks, key=lambda k: d[k]['dispatches'])]
    return int(round(2 * v['FETCH_SIZE'] * K + v['WRITE_SIZE'] * K))


out = {"_note": "HBM-side bytes per launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate runs of "
                "`bench.py --model %s --streams 1 --batch %d`: the launch size of one slice of the default bench, one kernel at a time; raw per-kernel "
                "counters in %s). gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128 B request on wide "
                "coalesced reads -> read bytes = 2*FETCH_SIZE KiB for the GEMM and LayerNorm kernels (full rows, 16 B per lane); the "
                "attention kernel reads 64-byte row slices (q/k/v of one head), which FETCH_SIZE counts exactly, so no doubling there; "
                "WRITE_SIZE is exact. proj / fc2 share one kernel: its even / odd dispatches." % (model, n_img, os.path.basename(src)),
       "_images_per_launch": n_img, "_model": model,
       "ln_gemm_qkv": tr(['k_ln_gemm2<0,', 'k_ln_gemm<0,'], 2), "ln_gemm_fc1": tr(['k_ln_gemm2<5,', 'k_ln_gemm<5,'], 2),
       "gemm_qkv": tr('k_gemm_dma<0,', 2), "gemm_fc1": tr(['k_gemm_dma<5,', 'k_gemm_dma<1,'], 2),
       "gemm_proj": resid('#even'), "gemm_fc2": resid('#odd'),
       "layernorm": tr('k_int_layernorm<', 2), "attention": tr('k_lis_attention<', 1), "gemm_embed": tr('k_gemm_i8<3,', 2)}
name = 'pmc_summary.json' if model == 'deit_small' else 'pmc_summary_%s.json' % model
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', name), 'w'), indent=1)
print({k: v for k, v in out.items() if not k.startswith('_note')})
