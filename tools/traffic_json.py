"""profiles/round1_pmc_<tag>.txt -> profiles/round1_traffic.json (HBM bytes per launch)."""
import json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for tag in ("edgeconv", "edgeconv_split", "conv5", "conv5_split", "knn", "chamfer"):
    path = os.path.join(root, "profiles", f"round1_pmc_{tag}.txt")
    if not os.path.exists(path):
        continue
    vals = dict(re.findall(r"^(\w+)\s+n=\s*\d+\s+mean=\s*([\d.]+)", open(path).read(), flags=re.M))
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        f, w = float(vals["FETCH_SIZE"]) * 1024, float(vals["WRITE_SIZE"]) * 1024
        out[tag] = {"fetch_size_bytes_raw": f, "write_size_bytes": w,
                    "hbm_bytes_per_launch": 2 * f + w,
                    "note": "FETCH_SIZE x2 (gfx950 rocprofv3 under-reports wide coalesced reads by 2x, MI355X_MICROARCH.md HBM section) + WRITE_SIZE"}
json.dump(out, open(os.path.join(root, "profiles", "round1_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
