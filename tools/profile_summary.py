# Condenses the rocprofv3 output of tools/profile_round.sh into profiles/<round>_*: kernel stats csv copies, per-kernel PMC sums and
# the JSON files bench.py reads for roofline.traffic.   usage: profile_summary.py <out dir> <round>
import collections, csv, glob, json, os, shutil, sys
out, rnd = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")


def stats_csv(sub, name):
    f = sorted(glob.glob(os.path.join(out, sub, "**", "*kernel_stats.csv"), recursive=True))
    if f:
        shutil.copy(f[0], os.path.join(prof, name))
        rows = list(csv.DictReader(open(f[0])))
        return {r["Name"].split("(")[0]: (int(r["Calls"]), float(r["AverageNs"])) for r in rows}
    return {}


def pmc(sub, counter):
    tot = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                n = r["Kernel_Name"].split("(")[0]
                tot[n][0] += 1
                tot[n][1] += float(r["Counter_Value"])
    return tot


st = stats_csv("trace", f"{rnd}_bench_default_kernel_stats.csv")
ast = stats_csv("atrace", f"{rnd}_assoc_sweep_kernel_stats.csv")
lines = []
for tag, fs, ws, stats in (("bench", "pmc_f", "pmc_w", st), ("assoc", "apmc_f", "apmc_w", ast)):
    F, W = pmc(fs, "FETCH_SIZE"), pmc(ws, "WRITE_SIZE")
    for k in sorted(set(F) | set(W), key=lambda k: -(F[k][1] + W[k][1])):
        nf, sf = F[k]
        nw, sw = W[k]
        lines.append(f"{tag:6s} {k[:48]:48s} dispatches {max(nf, nw):6d}  FETCH_SIZE {sf / max(nf, 1):12.1f} KiB/dispatch (x2 on gfx950)  "
                     f"WRITE_SIZE {sw / max(nw, 1):12.1f} KiB/dispatch")
        short = k.split("::")[-1]
        if short in ("k_chol_step_batched", "k_assoc_sweep", "k_assoc_sweep_512", "k_assoc_sweep_r", "k_pcg_symv", "k_pcg_tl_symv", "k_schur_b", "k_border_syrk", "k_chol_step", "k_sep_gather"):
            # full-size dispatches only: the streaming build also launches these kernels on growing systems
            fetch_kib, write_kib = sf / max(nf, 1), sw / max(nw, 1)
            js = {"kernel": short, "source": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_round.sh {rnd}); "
                                             "average over all dispatches of the run",
                  "dispatches": max(nf, nw), "fetch_size_kib_per_launch": fetch_kib, "write_size_kib_per_launch": write_kib,
                  "fetch_correction": 2.0,
                  "correction_note": "gfx950: FETCH_SIZE counts 128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
                  "hbm_bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0}
            if short in stats:
                js["rocprof_calls"], js["rocprof_avg_ns"] = stats[short]
            js["profile"] = ("dense" if "--dense-profile" in os.environ.get("BENCH_ARGS", "") else
                             "structure" if "--joint" in os.environ.get("BENCH_ARGS", "") else "exact_joint")      # which tiles the solver touched / which pass
            if short == "k_chol_step_batched":
                js["robots"] = 8      # systems per launch: two launch sequences of four on a wide (dense) profile, one sequence per system on a narrow one
            json.dump(js, open(os.path.join(prof, f"{rnd}_pmc_traffic_{short}.json"), "w"), indent=1)
open(os.path.join(prof, f"{rnd}_pmc_hbm_traffic.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:25]))
