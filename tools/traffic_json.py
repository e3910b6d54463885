"""profiles/round<R>_pmc_<tag>.txt -> profiles/round<R>_traffic.json: HBM bytes per launch and, where collected, the
executed VALU / MFMA instruction counts per launch.  usage: traffic_json.py [round]   (default 2; a tag without a
round-R file falls back to the newest earlier round whose kernel is unchanged)"""
import json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = int(sys.argv[1]) if len(sys.argv) > 1 else 2
out = {}
TAGS = ("edgeconv", "edgeconv_split", "edgeconv_f16", "edgeconv_f16b", "conv5", "conv5_split", "conv5_f16", "conv5_f16_2p", "knn", "knn_mfma",
        "chamfer", "chamfer_c4", "emd_sweep", "emd_match", "group_c5", "sa_mlp3", "bq_cells", "attention", "featknn")
# from round 5 on a round's file only carries what that round measured (entries sourced from round-1 passes of kernels
# that had changed since); earlier rounds keep the fall-back to the newest earlier file
for tag in TAGS:
    path = None
    for r in range(rnd, rnd - 1 if rnd >= 5 else 0, -1):
        cand = os.path.join(root, "profiles", f"round{r}_pmc_{tag}.txt")
        if os.path.exists(cand):
            path = cand
            break
    if path is None:
        continue
    vals = dict(re.findall(r"^(\w+)\s+n=\s*\d+\s+mean=\s*([\d.]+)", open(path).read(), flags=re.M))
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        f, w = float(vals["FETCH_SIZE"]) * 1024, float(vals["WRITE_SIZE"]) * 1024
        out[tag] = {"source": os.path.basename(path), "fetch_size_bytes_raw": f, "write_size_bytes": w,
                    "hbm_bytes_per_launch": 2 * f + w,
                    "note": "FETCH_SIZE x2 (gfx950 rocprofv3 under-reports wide coalesced reads by 2x, MI355X_MICROARCH.md HBM section) + WRITE_SIZE"}
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_LDS", "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT",
                  "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_INSTS_VMEM_RD"):
            if k in vals:
                out[tag][k] = float(vals[k])
json.dump(out, open(os.path.join(root, "profiles", f"round{rnd}_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
