"""Per-kernel-family MFMA-busy and HBM byte totals from rocprofv3 --pmc counter_collection.csv files (run on the GPU box;
the raw CSVs are too large to travel).  usage: pmc_summary.py OUT.json NAME=counter_collection.csv ..."""
import collections, csv, json, sys

def fam(n):
    for key in ("conv3x3_halo", "igemm_nt", "wgrad_kx3", "wgrad_tn", "attn_fwd", "attn_bwd_dq", "attn_bwd_dkv", "attn_delta", "gn_silu_fwd",
                "gn_silu_bwd_apply", "gn_silu_bwd_reduce", "gn_stats", "gn_finalize", "rownorm_fwd", "rownorm_bwd", "rope_qk", "pack_weight"):
        if key in n:
            return key
    return "other"

out = collections.defaultdict(lambda: collections.defaultdict(float))
for arg in sys.argv[2:]:
    tag, path = arg.split("=", 1)
    for r in csv.DictReader(open(path)):
        f = fam(r["Kernel_Name"])
        out[f][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES"):
            out[f]["dispatches:" + r["Counter_Name"]] += 1
json.dump({k: dict(v) for k, v in out.items()}, open(sys.argv[1], "w"), indent=1)
print("wrote", sys.argv[1])
