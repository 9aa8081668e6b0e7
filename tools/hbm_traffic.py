#!/usr/bin/env python3
"""HBM bytes per train step and per kernel, measured with rocprofv3 PMC counters the way MI355X_MICROARCH.md (HBM section)
prescribes: FETCH_SIZE and WRITE_SIZE in two SEPARATE passes (they do not fit one), kernel trace only beside them, FETCH_SIZE
doubled (gfx950 tallies a 128-byte read request as 64 B on wide coalesced reads; uncalibrated for narrower accesses), both
reported by rocprofv3 in KiB.  Each pass runs tools/train_steps.py (N identical bench-shaped train steps and nothing else) as a
child process; per-kernel totals are divided by N.

    python3 tools/hbm_traffic.py <out_dir> [--steps 4] [--precision fp16]     -> <out_dir>/hbm_traffic.json  (and stdout)

bench.py imports collect() to fill roofline.traffic from the SAME code it just timed when the committed profile is stale."""
import csv, glob, hashlib, json, os, shutil, subprocess, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel symbol -> bench.py timer name
TIMER_OF = [("user64_fwd_kernel", "user64_fwd"), ("user64_bwd_kernel", "user64_bwd"), ("user64_prep_kernel", "user64_prep"),
            ("fused_fwd16v1_kernel", "fused_fwd16"), ("fused_bwd16v1_attn_kernel", "fused_bwd16_attn"), ("gemm16_tn_kernel<true, 4", "dwo_bwd"),
            ("fused_fwd16p_kernel", "fused_fwd16"), ("fused_fwd16_kernel<true, 1>", "fused_fwd16"), ("fused_fwd16_kernel<true, 2>", "fused64_fwd16"),
            ("fused_bwd16_pool_kernel<1>", "fused_bwd16_pool"), ("fused_bwd16_pool_kernel<2>", "fused64_bwd16_pool"),
            ("fused_bwd16_attn_kernel<1>", "fused_bwd16_attn"), ("fused_bwd16_attn_kernel<2>", "fused64_bwd16_attn"),
            ("gemm16_tn_kernel<true", "dwadd_bwd"), ("gemm16_tn_kernel<false", "dwqkv_bwd"),
            ("gemm16_dx_kernel", "dx_bwd"), ("gather16_kernel", "gather_dropout"), ("scatter_grouped_kernel", "scatter_dropout"),
            ("adam_kernel", "adam"), ("attn_bwd", "attn_bwd"), ("attn_fwd", "attn_fwd"),
            ("gemm_nt_bf16_kernel", "gemm_nt_bf16 (user encoder)"), ("gemm_tn_bf16_kernel", "gemm_tn_bf16 (user encoder)"),
            ("addattn_fwd", "addattn_fwd"), ("addattn_bwd_rows", "addattn_bwd_rows")]


def csrc_sha16():
    """Hash of every kernel / header source: a traffic file is valid for exactly the code it was measured on."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "pytorch_news_recommender_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "pytorch_news_recommender_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _short(name):
    n = name.split("(")[0].replace("void ", "").replace("nrms::", "")
    return n


def collect(out_dir, steps=4, precision="fp16", extra=(), timeout=240):
    """Two PMC passes -> {"step_bytes", "by_kernel": {name: {fetch_x2, write, total, launches_per_step}}, "by_timer", ...}.
    Raises RuntimeError when rocprofv3 is missing or a pass fails."""
    roc = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(roc):
        raise RuntimeError("rocprofv3 not found")
    os.makedirs(out_dir, exist_ok=True)
    per = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(int)
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(os.path.abspath(out_dir), "pmc_" + counter)
        shutil.rmtree(d, ignore_errors=True)
        cmd = [roc, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.join(ROOT, "tools", "train_steps.py"), "--steps", str(steps), "--precision", precision] + list(extra)
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            raise RuntimeError("rocprofv3 --pmc %s failed (rc %d): %s" % (counter, r.returncode, r.stderr[-400:]))
        seen = set()
        for row in csv.DictReader(open(files[0])):
            if row["Counter_Name"] != counter:
                continue
            n = _short(row["Kernel_Name"])
            per[n][counter] += float(row["Counter_Value"]) * 1024.0          # KiB, summed over the XCD instances
            if counter == "FETCH_SIZE" and row["Dispatch_Id"] not in seen:
                seen.add(row["Dispatch_Id"])
                launches[n] += 1
    by_kernel, by_timer, total = {}, {}, 0.0
    for n, c in per.items():
        f2, w = 2.0 * c.get("FETCH_SIZE", 0.0) / steps, c.get("WRITE_SIZE", 0.0) / steps
        by_kernel[n] = {"fetch_x2": f2, "write": w, "total": f2 + w, "launches_per_step": launches[n] / float(steps)}
        total += f2 + w
        for key, timer in TIMER_OF:
            if key in n:
                t = by_timer.setdefault(timer, {"hbm_bytes_per_step": 0.0, "kernels": []})
                t["hbm_bytes_per_step"] += f2 + w
                t["kernels"].append(n)
                break
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over %d identical train steps (tools/train_steps.py, "
                   "B=512, %s mode); bytes PER STEP = sum over a kernel's launches / steps; FETCH_SIZE doubled per "
                   "MI355X_MICROARCH.md (HBM section)" % (steps, precision),
           "csrc_sha16": csrc_sha16(), "steps": steps, "precision": precision, "extra_args": list(extra),
           "step_bytes": total, "by_timer": by_timer,
           "by_kernel": dict(sorted(by_kernel.items(), key=lambda kv: -kv[1]["total"]))}
    json.dump(out, open(os.path.join(out_dir, "hbm_traffic.json"), "w"), indent=1)
    return out


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("out_dir")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--precision", default="fp16")
    ap.add_argument("--fp16-user-encoder", action="store_true")
    ap.add_argument("--model", default="v0", choices=["v0", "v1"], help="v1: the nrms_v1 step of bench.py's variants")
    a = ap.parse_args()
    res = collect(a.out_dir, a.steps, a.precision, (["--fp16-user-encoder"] if a.fp16_user_encoder else []) + (["--model", "v1"] if a.model == "v1" else []))
    print("HBM bytes per step: %.3f GB" % (res["step_bytes"] / 1e9))
    for n, v in list(res["by_kernel"].items())[:24]:
        print("   %-70s %.3f GB  (read %.3f, write %.3f; %.1f launches/step)" % (n[:70], v["total"] / 1e9, v["fetch_x2"] / 1e9, v["write"] / 1e9, v["launches_per_step"]))
