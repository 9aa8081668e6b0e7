#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/profile_round.sh into the small files kept under profiles/:

    <tag>_<prec>_kernel_stats.csv    rocprofv3 --stats per-kernel summary (name, calls, total/avg/min/max ns, %)
    <tag>_<prec>_hbm_traffic.json    HBM bytes per launch of each kernel's LARGEST launch (the news encoder's):
                                     FETCH_SIZE x 2 (gfx950 tallies 128-B read requests at 64 B,
                                     MI355X_MICROARCH.md "HBM") + WRITE_SIZE, both reported in KiB by rocprofv3.

Written next to the raw data (gpurun_out/prof_<tag>/summary/) so it travels back from the GPU box; copy
it into profiles/ to have it judged."""
import csv, glob, json, os, sys
from collections import defaultdict

# kernel symbol -> bench.py timer name (news-encoder instantiations at the bench shape)
TIMER_OF = [("fused_fwd16_kernel", "fused_fwd16"), ("fused_bwd16_pool_kernel", "fused_bwd16_pool"),
            ("fused_bwd16_attn_kernel", "fused_bwd16_attn"), ("gemm16_tn_kernel<true", "dwadd_bwd"), ("gemm16_tn_kernel<false", "dwqkv_bwd"),
            ("gemm16_dx_kernel", "dx_bwd"), ("gather16_kernel", "gather_dropout"), ("scatter_grouped_kernel", "scatter_dropout"),
            ("attn_bwd_kernel", "attn_bwd"), ("attn_fwd_kernel", "attn_fwd"),
            ("gemm_nt_bf16_kernel<19, 0, 0", "qkv_proj_fwd"), ("gemm_nt_bf16_kernel<19, 2, 1", "dctx_bwd"),
            ("gemm_tn_bf16_kernel<0,", "dwqkv_bwd"), ("gemm_tn_bf16_kernel<2,", "dwadd_bwd"),
            ("addattn_fwd_bf16_kernel", "addattn_fwd"), ("addattn_bwd_rows_kernel", "addattn_bwd_rows"),
            ("gather_dropout_kernel", "gather_dropout"), ("scatter_dropout_kernel", "scatter_dropout"),
            ("gemm_nt_kernel<19, 0", "qkv_proj_fwd"), ("gemm_tn_kernel<0", "dwqkv_bwd")]


def main(out_dir, tag, prec):
    summ = os.path.join(out_dir, "summary")
    os.makedirs(summ, exist_ok=True)
    stats = glob.glob(os.path.join(out_dir, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        rows = list(csv.reader(open(stats[0])))
        with open(os.path.join(summ, "%s_%s_kernel_stats.csv" % (tag, prec)), "w", newline="") as f:
            csv.writer(f).writerows(rows)
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(os.path.join(out_dir, "pmc_" + counter, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        by_dispatch = defaultdict(float)
        name_of = {}
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] != counter:
                continue
            by_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"])      # summed over XCD instances
            name_of[r["Dispatch_Id"]] = r["Kernel_Name"]
        largest = defaultdict(float)
        for d, v in by_dispatch.items():
            n = name_of[d].split("(")[0]
            largest[n] = max(largest[n], v * 1024.0)
        for n, v in largest.items():
            per.setdefault(n, {})[counter] = v
    kernels, by_timer = {}, {}
    for n, c in sorted(per.items(), key=lambda kv: -(2 * kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0))):
        if "nrms::" not in n:
            continue
        fetch2 = 2.0 * c.get("FETCH_SIZE", 0.0)
        wr = c.get("WRITE_SIZE", 0.0)
        kernels[n] = {"fetch_bytes_x2_corrected": fetch2, "write_bytes": wr, "total": fetch2 + wr}
        for key, timer in TIMER_OF:
            if key in n:
                for t in timer.split("|"):
                    by_timer.setdefault(t, {"kernel": n, "hbm_bytes_per_launch": fetch2 + wr})
                break
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/profile_round.sh), largest "
                       "(news-encoder) launch of each kernel at B=512, %s mode; FETCH_SIZE doubled per "
                       "MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request on wide coalesced reads; "
                       "uncalibrated for 8-byte-per-lane loads)" % prec,
               "kernels": kernels, "by_timer": by_timer},
              open(os.path.join(summ, "%s_%s_hbm_traffic.json" % (tag, prec)), "w"), indent=1)
    print("summary written to", summ)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "bf16x3")
