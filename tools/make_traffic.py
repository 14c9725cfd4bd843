#!/usr/bin/env python3
"""profiles/rNN_traffic.json from a pmc summary (tools/pmc_summary.py output): beyond-L2 traffic of the dominant kernel
and of the whole step, per launch, with the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (x2: the counter tallies
64 B per 128-B request for wide coalesced reads; WRITE_SIZE is exact).  FETCH_SIZE / WRITE_SIZE are in KB.

    tools/make_traffic.py gpurun_out/prof_r02/pmc_summary.txt r02 > profiles/r02_traffic.json
"""
import json
import re
import sys

BATCH = 32
PX3 = 64 * 128


def parse(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^(\S.*?)\s+\(dispatches (\d+)\)", line)
        if m:
            cur = m.group(1)
            out[cur] = {"dispatches": int(m.group(2))}
        elif cur and "=" in line:
            for kv in line.split():
                k, v = kv.split("=")
                out[cur][k] = float(v)
    return out


def main():
    path, tag = sys.argv[1], sys.argv[2]
    k = parse(path)
    # the level-3 ESP branch kernels: conv_mfma_kernel<32, 8, 26, 9, 1, 5, 28, 25, ...> with a residual (flag bit 2)
    dom = {}
    for name, v in k.items():
        m = re.match(r"conv_mfma_kernel<32, 8, 26, 9, 1, 5, 28, 25, (\d+), (\d+), (\d+)>", name)
        if m and int(m.group(3)) & 2 and "FETCH_SIZE" in v:
            dom[name] = v
    if not dom:
        print(json.dumps({"error": "dominant kernel not found in " + path}))
        return
    name = max(dom, key=lambda n: dom[n]["dispatches"])
    v = dom[name]
    flags = int(re.search(r"(\d+)>$", name).group(1))
    fused = bool(flags & 262144)
    # algorithmic bytes per launch: reduced map in (25 ch) + residual in (128) + output (128) [+ next reduced map out (25)]
    alg = (25 + 128 + 128 + (25 if fused else 0)) * PX3 * 4 * BATCH
    per_step = 0.0
    for n2, v2 in k.items():
        if "FETCH_SIZE" in v2 and "WRITE_SIZE" in v2 and not n2.startswith("void at::"):
            # dispatches are over the whole run; per step = per launch x launches per step (8 ESP blocks, 1 otherwise, ...)
            per_step += (2 * v2["FETCH_SIZE"] + v2["WRITE_SIZE"]) * 1024.0 * v2["dispatches"]
    steps = max(v2["dispatches"] for n2, v2 in k.items() if n2.startswith("stem_kernel<true>") or n2 == "stem_kernel")
    print(json.dumps({
        "kernel": "conv_l3_esp_branches", "kernel_symbol": name, "round": tag, "fused_next_1x1": fused,
        "source": "profiles/%s_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, batch 32)" % tag,
        "fetch_size_kb": v["FETCH_SIZE"], "write_size_kb": v["WRITE_SIZE"],
        "correction": "FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request; MI355X_MICROARCH.md HBM section); WRITE_SIZE exact",
        "traffic_bytes_per_launch": round((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0),
        "algorithmic_bytes_per_launch": alg,
        "whole_step_bytes_beyond_l2": round(per_step / steps),
    }, indent=1))


if __name__ == "__main__":
    main()
