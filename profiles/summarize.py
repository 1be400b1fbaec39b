#!/usr/bin/env python3
"""Condenses raw rocprofv3 output (gpurun_out/prof) into small text/JSON summaries.

    python profiles/summarize.py <raw_dir> <tag>

Writes <raw_dir>/<tag>_kernel_stats.txt, <raw_dir>/<tag>_pmc.txt and <raw_dir>/traffic.json; copy them
into profiles/ to have them judged.  PMC handling follows MI355X_MICROARCH.md (HBM section):
FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B... no: rocprofv3 reports them in KB, and on
gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by exactly 2x, so read bytes are
doubled before being compared with a byte count; WRITE_SIZE is taken as reported (uncalibrated).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(raw, pattern):
    return sorted(glob.glob(os.path.join(raw, "**", pattern), recursive=True))


def short(name):
    name = name.replace("circl::mlkem::", "").replace("circl::prim::", "").replace("circl::mldsa::", "")
    return name.split("(")[0][:70]


def kernel_stats(raw, tag):
    lines = []
    for f in find(os.path.join(raw, "kt"), "*kernel_stats.csv"):
        with open(f) as fh:
            rows = list(csv.DictReader(fh))
        lines.append(f"# {os.path.relpath(f, raw)}")
        lines.append(f"{'kernel':72s} {'calls':>6s} {'total_ms':>10s} {'avg_ms':>10s} {'min_ms':>10s} {'max_ms':>10s} {'pct':>6s}")
        for r in rows:
            g = lambda k: float(r.get(k, 0) or 0)
            lines.append(f"{short(r['Name']):72s} {int(g('Calls')):6d} {g('TotalDurationNs')/1e6:10.3f} {g('AverageNs')/1e6:10.4f} "
                         f"{g('MinNs')/1e6:10.4f} {g('MaxNs')/1e6:10.4f} {g('Percentage'):6.2f}")
    # also derive from the trace in case the stats file layout differs
    for f in find(os.path.join(raw, "kt"), "*kernel_trace.csv"):
        agg = defaultdict(list)
        meta = {}
        with open(f) as fh:
            for r in csv.DictReader(fh):
                agg[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                meta[r["Kernel_Name"]] = (r.get("VGPR_Count", "?"), r.get("Accum_VGPR_Count", "?"), r.get("SGPR_Count", "?"),
                                          r.get("LDS_Block_Size", "?"), r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"))
        lines.append(f"# {os.path.relpath(f, raw)} (derived from trace; VGPR/AGPR/SGPR/LDS/grid/wg of last launch)")
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            lines.append(f"{short(k):72s} {len(v):6d} {sum(v)/1e6:10.3f} {sum(v)/len(v)/1e6:10.4f} {min(v)/1e6:10.4f} {max(v)/1e6:10.4f}  {meta[k]}")
            # One persistent grid serves every batch size, so the line above blends 2^15-item chunks of the host path, 2^18 and
            # 2^20 launches.  Launches of one batch size have nearly equal durations: split the sorted durations wherever two
            # neighbours differ by more than 1.35x and report each class (>= 3 launches) on its own -- the 2^20 class of the
            # dominant kernel is the figure bench.py's roofline.avg_launch_ms must agree with.
            d = sorted(v)
            classes, cur = [], [d[0]]
            for x in d[1:]:
                if x > 1.35 * cur[-1]:
                    classes.append(cur)
                    cur = []
                cur.append(x)
            classes.append(cur)
            if len(classes) > 1:
                for c in classes:
                    if len(c) >= 3:
                        lines.append(f"{'    duration class':72s} {len(c):6d} {sum(c)/1e6:10.3f} {sum(c)/len(c)/1e6:10.4f} {c[0]/1e6:10.4f} {c[-1]/1e6:10.4f}  median {c[len(c)//2]/1e6:.4f} ms")
    open(os.path.join(raw, f"{tag}_kernel_stats.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


def pmc(raw, tag):
    out_lines = []
    per_kernel = defaultdict(lambda: defaultdict(list))
    for sub, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        for f in find(os.path.join(raw, sub), "*counter_collection.csv"):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if r.get("Counter_Name") == counter:
                        per_kernel[r["Kernel_Name"]][counter].append(float(r["Counter_Value"]))
    traffic = {}
    out_lines.append(f"{'kernel':72s} {'launches':>8s} {'FETCH_SIZE(KB)/launch':>22s} {'WRITE_SIZE(KB)/launch':>22s} {'HBM bytes/launch (2*F+W)*1024':>30s}")
    for k, d in per_kernel.items():
        fz = d.get("FETCH_SIZE", [])
        wz = d.get("WRITE_SIZE", [])
        # skip the warm-up / input-generation launches: use the median-sized ones (the 2^20 batches)
        f_avg = sorted(fz)[len(fz) // 2] if fz else float("nan")
        w_avg = sorted(wz)[len(wz) // 2] if wz else float("nan")
        hbm = (2 * f_avg + w_avg) * 1024
        out_lines.append(f"{short(k):72s} {max(len(fz), len(wz)):8d} {f_avg:22.1f} {w_avg:22.1f} {hbm:30.3e}")
        # <K = 3, MODE = ENCAPS, ABLATE = 0, SCRATCH = true, SHARED = false>: the distinct-key kernel of the headline metric
        if "mlkem_encrypt_kernel<3, 0, 0, true, 0>" in k or "mlkem_encrypt_kernel<3, 0, 0, true, false>" in k:  # (the bool spelling: round-1 builds)
            traffic["mlkem768_encrypt_bytes_per_launch_2p20"] = hbm
            traffic["mlkem768_encrypt_fetch_kb_reported"] = f_avg
            traffic["mlkem768_encrypt_write_kb_reported"] = w_avg
        if "mlkem_encrypt_kernel<3, 0, 0, true, 1>" in k or "mlkem_encrypt_kernel<3, 0, 0, true, true>" in k:  # bench.py's secondary shared-key steps
            traffic["mlkem768_encrypt_shared_key_bytes_per_launch_2p20"] = hbm
        if "mlkem_hash_kernel<3" in k or "mlkem_hash_kernelILi3" in k:
            traffic["mlkem768_hash_bytes_per_launch_2p20"] = hbm
    # which build the figures belong to: bench.py reports them only for the very same libcirclhip.so
    try:
        import hashlib
        h = hashlib.sha256()
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "circl_amd", "libcirclhip.so"), "rb") as fh:
            for blk in iter(lambda: fh.read(1 << 20), b""):
                h.update(blk)
        traffic["lib_sha256"] = h.hexdigest()
    except OSError:
        pass
    # SQ counters of the same kernel (sq*/ passes)
    valu = {}
    sq = defaultdict(list)
    for f in find(os.path.join(raw, "sq1"), "*counter_collection.csv") + find(os.path.join(raw, "sq2"), "*counter_collection.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"]
                if "mlkem_encrypt_kernel<3, 0, 0, true, 0>" in k or "mlkem_encrypt_kernel<3, 0, 0, true, false>" in k:
                    sq[r["Counter_Name"]].append(float(r["Counter_Value"]))
    med = lambda c: sorted(sq[c])[len(sq[c]) // 2] if sq.get(c) else None
    if sq:
        valu = {"mlkem768_encrypt_valu_insts_per_launch_2p20": med("SQ_INSTS_VALU"), "salu": med("SQ_INSTS_SALU"), "lds": med("SQ_INSTS_LDS"),
                "vmem_rd": med("SQ_INSTS_VMEM_RD"), "vmem_wr": med("SQ_INSTS_VMEM_WR"), "waves": med("SQ_WAVES"),
                "grbm_gui_active_per_xcd": (med("GRBM_GUI_ACTIVE") or 0) / 8, "wave_quad_cycles": med("SQ_WAVE_CYCLES"),
                "busy_cycles": med("SQ_BUSY_CYCLES"), "active_inst_valu": med("SQ_ACTIVE_INST_VALU"), "lib_sha256": traffic.get("lib_sha256"),
                "method": "rocprofv3 --pmc SQ_* (own passes, kernel-trace only), median over the launches of bench.py --pmc-child; "
                          "GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_WAVE_CYCLES counts in units of 4 cycles"}
        json.dump(valu, open(os.path.join(raw, "valu.json"), "w"), indent=1)
        out_lines.append("SQ counters of mlkem_encrypt_kernel<3, ENCAPS> per launch of 2^20: " + json.dumps({k: v for k, v in valu.items() if k not in ("method", "lib_sha256")}))
    traffic["method"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (kernel-trace only); "
                         "median over the timed launches; read bytes = 2 x FETCH_SIZE x 1024 (gfx950 reports half of a wide "
                         "coalesced stream, MI355X_MICROARCH.md HBM section); write bytes = WRITE_SIZE x 1024 (uncalibrated)")
    open(os.path.join(raw, f"{tag}_pmc.txt"), "w").write("\n".join(out_lines) + "\n")
    json.dump(traffic, open(os.path.join(raw, "traffic.json"), "w"), indent=1)
    print("\n".join(out_lines))
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    raw, tag = sys.argv[1], sys.argv[2]
    kernel_stats(raw, tag)
    pmc(raw, tag)
