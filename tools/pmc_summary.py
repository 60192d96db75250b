"""Summarise the rocprofv3 --pmc passes written by tools/pmc_passes.sh into profiles/:
    python tools/pmc_summary.py <pmc_out_dir> <summary.csv> [traffic.json n_bodies]
Per kernel and counter: launches, mean, min, max (counter values are summed over the dispatch's
dimensions as rocprofv3 reports them).  The traffic file is what bench.py reads for roofline.traffic:
HBM bytes per force launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB; gfx950 counts a 128-B read request as 64 B,
MI355X_MICROARCH.md, HBM section)."""
import collections, csv, glob, json, os, sys

src, out = sys.argv[1], sys.argv[2]
vals = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter -> per-dispatch values
for path in sorted(glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    per_dispatch = collections.defaultdict(float)
    names = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            key = (row["Dispatch_Id"], row["Counter_Name"])
            per_dispatch[key] += float(row["Counter_Value"])
            names[row["Dispatch_Id"]] = row["Kernel_Name"]
    for (disp, counter), v in per_dispatch.items():
        vals[names[disp]][counter].append(v)
with open(out, "w") as f:
    f.write("# rocprofv3 --pmc passes (tools/pmc_passes.sh: one counter group per run, never combined with tracing)\n")
    f.write("# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM section)\n")
    f.write("kernel,counter,launches,mean,min,max\n")
    for kernel in sorted(vals, key=lambda k: -sum(len(v) for v in vals[k].values())):
        if not kernel.startswith(("void murb", "murb")):
            continue
        for counter, v in vals[kernel].items():
            f.write(f"\"{kernel}\",{counter},{len(v)},{sum(v)/len(v):.6g},{min(v):.6g},{max(v):.6g}\n")
if len(sys.argv) > 4:
    force = [k for k in vals if "murb_force_sym_kernel" in k or "murb_force_kernel" in k]
    force.sort(key=lambda k: -len(vals[k].get("FETCH_SIZE", [])))
    k = force[0]
    fetch = sum(vals[k]["FETCH_SIZE"]) / len(vals[k]["FETCH_SIZE"])
    write = sum(vals[k]["WRITE_SIZE"]) / len(vals[k]["WRITE_SIZE"])
    json.dump({"n_bodies": int(sys.argv[4]), "kernel": k, "fetch_size_kib_raw": fetch, "write_size_kib_raw": write,
               "hbm_bytes_per_launch": (2 * fetch + write) * 1024,
               "correction": "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B), WRITE_SIZE as is; MI355X_MICROARCH.md HBM section",
               "source": os.path.basename(out)}, open(sys.argv[3], "w"), indent=1)
