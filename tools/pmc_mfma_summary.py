# MFMA-pipe utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES run (tools/gpu_mfma_util.sh):
#   usage: pmc_mfma_summary.py <rocprof out dir> <round>
# SQ_VALU_MFMA_BUSY_CYCLES sums the cycles in which a SIMD's matrix pipe is busy over all SIMDs of the chip (1024 on MI355X: 256 CUs x 4);
# a v_mfma_f64_16x16x4_f64 keeps its pipe busy for 64 cycles (2048 flop).  Dispatches are grouped by (kernel, grid size): the streaming
# updates launch the same kernels on small systems.  utilisation = busy / (duration x CLOCK x 1024) with the dispatch's own start / end
# timestamps from the same rows (counter collection serialises dispatches, it does not stretch them) at the nominal 2.4 GHz.
import collections, csv, glob, json, os, sys
out, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD, CLOCK_GHZ = 256 * 4, 2.4
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])      # (kernel, grid) -> dispatches, busy cycles, ns
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES":
            continue
        a = acc[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]))]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        try:
            a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        except (KeyError, ValueError):
            pass
lines, js = [], collections.defaultdict(list)
for (k, grid), (n, busy, ns) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if busy <= 0:
        continue
    util = busy / (ns * CLOCK_GHZ * N_SIMD) if ns > 0 else None
    lines.append(f"{k[:40]:40s} grid {grid:9d}  dispatches {n:6d}  MFMA busy cycles/dispatch {busy / n:12.0f} (= {busy / n / 64 * 2048 / 1e6:9.1f} MFLOP)  "
                 f"avg {ns / n / 1e3:8.2f} us  matrix-pipe utilisation {'n/a' if util is None else f'{100 * util:6.2f} %'}")
    js[k.split("::")[-1]].append({"grid": grid, "dispatches": n, "mfma_busy_cycles_per_dispatch": busy / n, "avg_us": ns / n / 1e3, "mfma_util": util})
head = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES over the round's profile command (tools/gpu_mfma_util.sh), by kernel and grid size;\n"
        "utilisation = busy cycles / (dispatch duration x 2.4 GHz x 1024 SIMDs); 64 busy cycles = one v_mfma_f64_16x16x4_f64 = 2048 flop\n")
open(os.path.join(root, "profiles", f"{rnd}_pmc_mfma_util.txt"), "w").write(head + "\n".join(lines) + "\n")
json.dump(js, open(os.path.join(root, "profiles", f"{rnd}_pmc_mfma_util.json"), "w"), indent=1)
print("\n".join(lines[:14]))
