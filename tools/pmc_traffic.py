#!/usr/bin/env python3
"""Per-frame HBM read traffic of the decode frame from a rocprofv3 `--pmc FETCH_SIZE --kernel-trace
--output-format csv` pass (collected separately from the timing runs, as MI355X_MICROARCH.md prescribes).
FETCH_SIZE is in KiB and, on gfx950, reports exactly half of the bytes of wide coalesced streaming reads,
so it is doubled.  usage: python tools/pmc_traffic.py <counter_collection.csv> <n_frames_of_that_run> [out.json]
With out.json the per-frame figure is also written as the small JSON bench.py reads (profiles/pmc_traffic.json)."""
import collections
import csv
import json
import sys


def main(path, n_frames, out_json=None):
    agg = collections.defaultdict(lambda: [0, 0.0])
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == "FETCH_SIZE"]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # load-time work is not part of a frame: runs of >= 256 back-to-back launches of one kernel (the 512 GEMV
    # launches that tabulate fast layer 0's q|k|v per code, the weight packing) are dropped
    keep, i = [], 0
    while i < len(rows):
        j = i
        while j < len(rows) and rows[j]["Kernel_Name"] == rows[i]["Kernel_Name"] and rows[j]["Grid_Size"] == rows[i]["Grid_Size"]:
            j += 1
        if j - i < 256:
            keep.extend(rows[i:j])
        else:
            print(f"# dropped a run of {j - i} consecutive launches of {rows[i]['Kernel_Name'].split('(')[0]} (load-time)")
        i = j
    for r in keep:
        if "fmi::" not in r["Kernel_Name"]:
            continue
        k = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"]) * 1024 * 2
    # frames of the pass, counted from the launches themselves: one embed launch per frame, over the prompt rows for a
    # prefill frame (hundreds of work-groups), over the batch for a decode frame
    n_prefill = sum(n for (name, wgs), (n, _) in agg.items() if "embed_kernel" in name and wgs > 64)
    n_decode = sum(n for (name, wgs), (n, _) in agg.items() if "embed_kernel" in name and wgs <= 64)
    if n_prefill + n_decode == 0:
        n_prefill, n_decode = 1, n_frames - 1
    print(f"# {path}: {n_prefill} prefill frame(s) + {n_decode} graph-replayed decode frames (bench --frames {n_frames})")
    print(f"{'calls':>7} {'avg MB/launch':>14}  kernel [work-groups]")
    frame_bytes = 0.0
    for (name, wgs), (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:7d} {v / n / 1e6:14.3f}  {name} [{wgs}]")
        decode_kernel = any(t in name for t in ("linear_skinny", "attn_decode_fused", "fast_attn", "sample", "embed", "rmsnorm_rows"))
        if decode_kernel:
            frame_bytes += v
    # the prefill frame runs the same tail (fast chain + heads) once; slow layers there use the tiled path
    print(f"\nHBM bytes fetched by the decode-frame kernels: {frame_bytes / 1e9:.3f} GB over the run "
          f"= {frame_bytes / (n_decode + 0.52 * n_prefill) / 1e9:.3f} GB per decode frame "
          f"(a prefill frame's tail counted as 0.52 frame: the fast chain and the heads of a frame)")
    if out_json:
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from fish_speech_amd.build import decode_sources_sha

        with open(out_json, "w") as f:
            json.dump({"decode_sources_sha": decode_sources_sha(), "bytes_per_decode_frame": round(frame_bytes / (n_decode + 0.52 * n_prefill)), "batch": 8,
                       "decode_frames_in_pass": n_decode, "prefill_frames_in_pass": n_prefill,
                       "source": "rocprofv3 --pmc FETCH_SIZE --kernel-trace over `bench.py --frames %d --steps 1 --warmup 0 "
                                 "--no-codec --no-extras --no-cpu-baseline` (tools/make_profiles.sh), summed over the decode-frame "
                                 "kernels by tools/pmc_traffic.py: KiB x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md)" % n_frames},
                      f, indent=1)
            f.write("\n")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else None)
