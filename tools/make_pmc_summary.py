"""profiles/pmc_summary.json (HBM-side bytes per launch, the `traffic` field of bench.py's roofline) from a per-kernel PMC summary
written by tools/pmc_summary.py.   usage: python tools/make_pmc_summary.py <rXX_pmc_summary.json> <images_per_launch>"""
import json, os, sys
src, n_img = sys.argv[1], int(sys.argv[2])
d = json.load(open(src))
K = 1024


def find(prefix):
    ks = [k for k in d if k.startswith(prefix) and d[k].get('dispatches', 0) >= 3 and 'FETCH_SIZE' in d[k]]
    return max(ks, key=lambda k: d[k]['dispatches']) if ks else None


def tr(prefix, fetch_mul):
    k = find(prefix)
    if k is None:
        return None
    v = d[k]
    return int(round(fetch_mul * v['FETCH_SIZE'] * K + v['WRITE_SIZE'] * K))


out = {"_note": "HBM-side bytes per launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate runs of "
                "`bench.py --streams 1 --batch %d`: the launch size of one slice of the default three-slice bench, one kernel at a time; raw per-kernel "
                "counters in %s). gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128 B request on wide "
                "coalesced reads -> read bytes = 2*FETCH_SIZE KiB for the GEMM and LayerNorm kernels (full rows, 16 B per lane); the "
                "attention kernel reads 64-byte row slices (q/k/v of one head), which FETCH_SIZE counts exactly, so no doubling there; "
                "WRITE_SIZE is exact. k_gemm_dma<2,...> covers proj and fc2 launches together (mean of both)." % (n_img, os.path.basename(src)),
       "_images_per_launch": n_img,
       "ln_gemm_qkv": tr('k_ln_gemm<0, 6>', 2), "ln_gemm_fc1": tr('k_ln_gemm<5, 6>', 2), "gemm_resid_mean": tr('k_gemm_dma<2, 3, false>', 2),
       "layernorm": tr('k_int_layernorm<3, 32>', 2), "attention": tr('k_lis_attention<64, 7, false, true>', 1), "gemm_embed": tr('k_gemm_i8<3, false>', 2)}
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'pmc_summary.json'), 'w'), indent=1)
print({k: v for k, v in out.items() if not k.startswith('_note')})
