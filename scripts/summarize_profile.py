#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (scripts/profile.sh) into committed summaries under profiles/:
   profiles/<round>_<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary
   profiles/<round>_<tag>_pmc.json           per-launch means of the PMC passes + derived numbers
   profiles/hbm_traffic.json                 HBM bytes per launch keyed by workload (bench.py reads it)
usage: scripts/summarize_profile.py <tag> <round> [workload_key]"""
import collections, csv, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import source_sha16          # identifies the sources the profiled binary was built from (bench.py prints the same)
tag, rnd = sys.argv[1], sys.argv[2]
key = sys.argv[3] if len(sys.argv) > 3 else None
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, f"{rnd}_{tag}_kernel_stats.csv"))
vals, launches = {}, {}
for p in ("pmc1", "pmc2", "pmc3", "pmc4"):
    f = os.path.join(src, p, f"{p}_counter_collection.csv")
    if not os.path.exists(f):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "render_kernel<" not in name:
            continue
        targs = [t.strip() for t in name.split("render_kernel<")[1].split(">")[0].split(",")]    # COUNT, PILOT, CTR, SMALL
        if targs[1] == "false" and "finalize" not in name:                                        # skip the PILOT instantiation
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v = sorted(v)
        vals[k] = v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])     # median: the counters are
        launches[k] = len(v)           # device-wide, and a launch now and then picks up a neighbour's traffic (one WRITE_SIZE sample of 6 x the rest)
out = {"source": f"gpurun_out/prof_{tag} (scripts/profile.sh: separate rocprofv3 --pmc passes); medians over the launches",
       "source_sha16": source_sha16(), "per_launch_median": vals,
       "launches_averaged": launches}
if "GRBM_GUI_ACTIVE" in vals:
    cyc = vals["GRBM_GUI_ACTIVE"] / 8.0            # summed over the 8 XCDs
    out["derived"] = {"kernel_cycles": cyc}
    if "SQ_INSTS_VALU" in vals:
        out["derived"]["valu_issue_utilisation_at_2cyc_per_inst"] = vals["SQ_INSTS_VALU"] * 2.0 / (1024 * cyc)
        out["derived"]["valu_note"] = ("per-kernel cycles: launches of consecutive frames overlap, so this under-states the chip's "
                                       "utilisation; over wall time per frame it is SQ_INSTS_VALU*2/(1024*ms_per_step*clock)")
    for a_, b_ in (("SQ_WAIT_ANY", "wave_cycles_parked_frac"), ("SQ_WAIT_INST_ANY", "wave_cycles_issue_stalled_frac")):
        if a_ in vals and "SQ_WAVE_CYCLES" in vals:
            out["derived"][b_] = vals[a_] / vals["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_SALU" in vals and "SQ_INSTS_VALU" in vals:
        out["derived"]["salu_per_valu"] = vals["SQ_INSTS_SALU"] / vals["SQ_INSTS_VALU"]
    if "SQ_THREAD_CYCLES_VALU" in vals and "SQ_ACTIVE_INST_VALU" in vals:
        out["derived"]["valu_thread_utilisation"] = vals["SQ_THREAD_CYCLES_VALU"] / (vals["SQ_ACTIVE_INST_VALU"] * 64.0)
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    # MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of
    # a wide coalesced (16 B/lane) read stream -> doubled; WRITE_SIZE is exact for 16 B/lane stores.
    hbm = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    out["derived"]["hbm_bytes_per_launch"] = hbm
    out["derived"]["hbm_note"] = "(2*FETCH_SIZE + WRITE_SIZE) * 1024, gfx950 correction per MI355X_MICROARCH.md"
    if key:
        tj_path = os.path.join(dst, "hbm_traffic.json")
        tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
        tj[key] = hbm
        tj["_note"] = "HBM bytes per render_kernel launch from rocprofv3 PMC passes; see profiles/*_pmc.json"
        json.dump(tj, open(tj_path, "w"), indent=1, sort_keys=True)
bench_line = None
for log in ("bench_trace.log",):
    lines = [l for l in open(os.path.join(src, log)) if l.startswith("{")]
    if lines:
        open(os.path.join(dst, f"{rnd}_{tag}_bench_under_rocprof.json"), "w").write(lines[-1])
        bench_line = json.loads(lines[-1])
# VALU instructions per wave-iteration of the bounce loop (one world_hit for up to 64 rays): SQ_INSTS_VALU per launch over
# the launch's wave iterations = world_hit calls / (64 x lane utilisation), both from the bench line of the traced run
if bench_line and "SQ_INSTS_VALU" in vals and key:
    v = bench_line["valu"]
    px_spp = [int(x) for x in key.split("_")[-2].split("x")]
    hits_per_launch = v["mean_bounces_per_sample"] * px_spp[0] * px_spp[1] * px_spp[2]
    wave_iters = hits_per_launch / (64.0 * v["lane_utilisation"])
    out["derived"]["valu_insts_per_wave_bounce"] = vals["SQ_INSTS_VALU"] / wave_iters
    vj_path = os.path.join(dst, "valu_pmc.json")
    vj = json.load(open(vj_path)) if os.path.exists(vj_path) else {}
    vj[key] = {"issue_frac": out["derived"].get("valu_issue_utilisation_at_2cyc_per_inst"),
               "thread_utilisation": out["derived"].get("valu_thread_utilisation"),
               "valu_insts_per_wave_bounce": out["derived"]["valu_insts_per_wave_bounce"],
               "valu_insts_per_launch": vals["SQ_INSTS_VALU"],          # deterministic for the workload: bench.py divides it by its own wall time
               "source": f"profiles/{rnd}_{tag}_pmc.json", "source_sha16": source_sha16()}
    vj["_note"] = "SQ counters per render_kernel launch from rocprofv3 --pmc passes of bench.py (scripts/profile.sh); bench.py copies them into valu.pmc_*"
    json.dump(vj, open(vj_path, "w"), indent=1, sort_keys=True)
# the launch timeline of the traced run: consecutive frames' render kernels overlap (two frames in flight)
tl = os.path.join(src, "trace", "trace_kernel_trace.csv")
if os.path.exists(tl):
    rows = [r for r in csv.DictReader(open(tl)) if "render_kernel<" in r["Kernel_Name"] or "finalize_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if rows:
        t0 = int(rows[0]["Start_Timestamp"])
        with open(os.path.join(dst, f"{rnd}_{tag}_launch_timeline.txt"), "w") as f:
            f.write(f"# rocprofv3 --kernel-trace of `bench.py` ({tag}): start / end of the render and finalize launches, ms since the first one.\n"
                    "# Two frames are in flight: launch n+1 starts while launch n's last pixels drain, so a launch's own duration is about\n"
                    "# twice the distance between consecutive ends (= the wall time per step).  ISA resources of the headline instantiation\n"
                    "# (scripts/kernel_resources.py; rocprofv3 reports granules / dynamic LDS as 0): see profiles/README.md.\n")
            for r in rows[:40]:
                nm = "render " + r["Kernel_Name"].split("render_kernel")[1][:34] if "render_kernel" in r["Kernel_Name"] else "finalize"
                s_, e_ = (int(r["Start_Timestamp"]) - t0) * 1e-6, (int(r["End_Timestamp"]) - t0) * 1e-6
                f.write(f"{nm:46s} start {s_:9.3f}  end {e_:9.3f}  duration {e_ - s_:8.3f}\n")
json.dump(out, open(os.path.join(dst, f"{rnd}_{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out.get("derived", {}), indent=1))
print(open(os.path.join(dst, f"{rnd}_{tag}_kernel_stats.csv")).read())
