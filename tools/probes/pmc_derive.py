"""Per-launch figures from a pmc_summary.py raw JSON (GPU box or build container):

    python tools/probes/pmc_derive.py RAW.json OUT.json images=64 [note="..."]

For every kernel family: HBM bytes per launch (FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads 1/2 of a wide
coalesced read stream -- 16 B per lane, LDS-DMA alike -- so it is doubled; WRITE_SIZE is exact for 16-byte streaming stores
and float atomics: MI355X_MICROARCH.md, HBM section), matrix-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel
cycles) with kernel cycles = SQ_BUSY_CYCLES / 32 shader engines, and the parked share of wave-cycles SQ_WAIT_ANY / SQ_WAVE_CYCLES."""
import json, sys
raw = json.load(open(sys.argv[1]))
kv = dict(a.split("=", 1) for a in sys.argv[3:])
out = {}
for fam, c in raw.items():
    if fam in ("other", "pack_weight"):
        continue
    d = {}
    nf, nw, ns = c.get("dispatches:FETCH_SIZE", 0), c.get("dispatches:WRITE_SIZE", 0), c.get("dispatches:SQ_VALU_MFMA_BUSY_CYCLES", 0)
    if nf:
        d["fetch_bytes_per_launch_corrected"] = c["FETCH_SIZE"] * 1024 * 2 / nf
    if nw:
        d["write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024 / nw
    if nf and nw:
        d["hbm_bytes_per_launch"] = d["fetch_bytes_per_launch_corrected"] + d["write_bytes_per_launch"]
    if ns and c.get("SQ_BUSY_CYCLES"):
        cyc = c["SQ_BUSY_CYCLES"] / 32 / ns
        d["kernel_cycles_per_launch"] = cyc
        d["mfma_utilisation"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / ns / (1024 * cyc), 4)
        d["waves_parked_fraction"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4)
    d["dispatches"] = int(max(nf, nw, ns))
    if "images" in kv:
        d["images"] = int(kv["images"])
    out[fam] = d
out["_method"] = ("tools/collect_profiles.sh: every --pmc pass is its own rocprofv3 run with --kernel-trace only; " + __doc__.split("For every kernel family: ")[1].replace("\n", " "))
if "note" in kv:
    out["_note"] = kv["note"]
json.dump(out, open(sys.argv[2], "w"), indent=1)
print("wrote", sys.argv[2])
