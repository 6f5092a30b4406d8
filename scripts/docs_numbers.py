#!/usr/bin/env python3
"""One value per quantity: every measured figure quoted in DESIGN.md / README.md is written as `value<!--TOKEN-->` (the marker
is invisible in rendered markdown) and comes from ONE place under profiles/ (TOKENS below).

    python scripts/docs_numbers.py            rewrite the figures in the documents from profiles/ (and fill `@TOKEN@` placeholders)
    python scripts/docs_numbers.py --check    exit 1 if a quoted figure differs from its source (tests/test_docs_numbers.py)
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ("DESIGN.md", "README.md")
R = "r05"


def _json(name):
    return json.load(open(os.path.join(ROOT, "profiles", name)))


def _shard(pattern, nth=0):
    """(Msamples/s per GPU, lane utilisation) of the nth line of profiles/r04_shard_throughput.txt matching `pattern`"""
    hits = [line for line in open(os.path.join(ROOT, "profiles", f"{R}_shard_throughput.txt")) if re.search(pattern, line)]
    m = re.search(r"per-GPU ([\d.]+) Msamples/s, lane utilisation ([\d.]+)", hits[nth])
    return float(m.group(1)), float(m.group(2))


def _phase(name, pattern):
    for line in open(os.path.join(ROOT, "profiles", f"{R}_{name}_phase_profile.txt")):
        m = re.search(pattern, line)
        if m:
            return float(m.group(1))
    raise KeyError(pattern)


# token -> (how to get the value, format)
TOKENS = {
    "C3_VALUE": (lambda: _json(f"{R}_bench_n1.json")["value"], ",.0f"),
    "C3_MS": (lambda: _json(f"{R}_bench_n1.json")["ms_per_step"], ".1f"),
    "CPU_CORES": (lambda: _json(f"{R}_bench_n1.json")["cpu_baseline"]["cores"], "d"),
    "CPU_VALUE": (lambda: _json(f"{R}_bench_n1.json")["cpu_baseline"]["value"], ".2f"),
    "C3_SPEEDUP": (lambda: _json(f"{R}_bench_n1.json")["speedup_vs_cpu_baseline"], ",.0f"),
    "C3_FRAC": (lambda: _json(f"{R}_bench_n1.json")["roofline"]["frac"], ".1e"),
    "C3_TRAFFIC_MB": (lambda: _json("hbm_traffic.json")["cover-glass_1920x1080x512_n1"] * 1e-6, ".1f"),
    "C3_ISSUE": (lambda: _json(f"{R}_c3_pmc.json")["derived"]["valu_issue_utilisation_at_2cyc_per_inst"], ".3f"),
    "C3_VALU": (lambda: _json(f"{R}_c3_pmc.json")["derived"]["valu_insts_per_wave_bounce"], ",.0f"),
    "C3_THREAD": (lambda: _json(f"{R}_c3_pmc.json")["derived"]["valu_thread_utilisation"], ".2f"),
    "C5_ISSUE": (lambda: _json(f"{R}_c5_pmc.json")["derived"]["valu_issue_utilisation_at_2cyc_per_inst"], ".2f"),
    "C5_VALU": (lambda: _json(f"{R}_c5_pmc.json")["derived"]["valu_insts_per_wave_bounce"], ",.0f"),
    "C5_SALU": (lambda: _json(f"{R}_c5_pmc.json")["derived"]["salu_per_valu"], ".2f"),
    "C5_PARKED": (lambda: _json(f"{R}_c5_pmc.json")["derived"]["wave_cycles_parked_frac"], ".2f"),
    "C5_WALL_ISSUE": (lambda: _json(f"{R}_bench_c5_n1.json")["valu_issue"]["frac_over_wall_time"], ".2f"),
    "C3_WALL_ISSUE": (lambda: _json(f"{R}_bench_n1.json")["valu_issue"]["frac_over_wall_time"], ".2f"),
    "C1_VALUE": (lambda: _json(f"{R}_bench_c1_n1.json")["value"], ",.0f"),
    "C2_VALUE": (lambda: _json(f"{R}_bench_c2_n1.json")["value"], ",.0f"),
    "C4_VALUE": (lambda: _json(f"{R}_bench_c4_n1.json")["value"], ",.0f"),
    "C5_VALUE": (lambda: _json(f"{R}_bench_c5_n1.json")["value"], ",.0f"),
    "C5CTR_VALUE": (lambda: _json(f"{R}_bench_c5_counter_n1.json")["value"], ",.0f"),
    "INTERACTIVE_VALUE": (lambda: _json(f"{R}_bench_interactive_n1.json")["value"], ",.0f"),
    "INTERACTIVE_X32_VALUE": (lambda: _json(f"{R}_bench_interactive_x32_n1.json")["value"], ",.0f"),
    "SHARD8_VALUE": (lambda: _shard(r"^stress 1920x1080x4096 shard 0/8 rng_mode 0")[0], ",.0f"),
    "SHARD8_UTIL": (lambda: _shard(r"^stress 1920x1080x4096 shard 0/8 rng_mode 0")[1], ".2f"),
    # (the second / third 0/8 line of the file: the same share with round 4's schedule, MRT_HINT=8,1, and with round 3's,
    # MRT_SLOTS=2 -- see scripts/refresh_measurements.sh)
    "SHARD8_R4_VALUE": (lambda: _shard(r"^stress 1920x1080x4096 shard 0/8 rng_mode 0", 1)[0], ",.0f"),
    "SHARD8_R4_UTIL": (lambda: _shard(r"^stress 1920x1080x4096 shard 0/8 rng_mode 0", 1)[1], ".2f"),
    "SHARD8_OLD_VALUE": (lambda: _shard(r"^stress 1920x1080x4096 shard 0/8 rng_mode 0", 2)[0], ",.0f"),
    "SHARD8_OLD_UTIL": (lambda: _shard(r"^stress 1920x1080x4096 shard 0/8 rng_mode 0", 2)[1], ".2f"),
    "C5_WHOLE_REDRAW_VALUE": (lambda: _shard(r"^stress 1920x1080x4096 shard 0/1 rng_mode 0")[0], ",.0f"),
    "C4SHARD_AUTO_VALUE": (lambda: _shard(r"^cover-glass 3840x2160x1024 shard 0/8 rng_mode 0", 1)[0], ",.0f"),
    "C2_AGAIN_VALUE": (lambda: _json(f"{R}_bench_c2_again_n1.json")["value"], ",.0f"),
    "C2_MEASURE_VALUE": (lambda: _json(f"{R}_bench_c2_measure_n1.json")["value"], ",.0f"),
    "C3_MEASURE_VALUE": (lambda: _json(f"{R}_bench_measure_n1.json")["value"], ",.0f"),
    "SHARD4_VALUE": (lambda: _shard(r"^stress 1920x1080x4096 shard 0/4 rng_mode 0")[0], ",.0f"),
    "SHARD2_VALUE": (lambda: _shard(r"^stress 1920x1080x4096 shard 0/2 rng_mode 0")[0], ",.0f"),
    "C4SHARD_VALUE": (lambda: _shard(r"^cover-glass 3840x2160x1024 shard 0/8 rng_mode 0")[0], ",.0f"),
    "C5_NODE_ROUNDS": (lambda: _phase("c5", r"node rounds/sweep ([\d.]+)"), ".1f"),
    "C5_ROUND_ITEMS": (lambda: _phase("c5", r"node rounds/sweep [\d.]+ \(items/round ([\d.]+)\)"), ".0f"),
}


def values():
    out = {}
    for tok, (get, fmt) in TOKENS.items():
        v = get()
        out[tok] = format(int(v) if fmt == "d" else v, fmt)
    return out


def rewrite(text, vals):
    for tok, v in vals.items():
        text = text.replace(f"@{tok}@", f"{v}<!--{tok}-->")
        text = re.sub(r"[^\s(|]+<!--" + tok + r"-->", lambda m: f"{v}<!--{tok}-->", text)
    return text


def check():
    vals = values()
    bad, seen = [], set()
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for m in re.finditer(r"([^\s(|]+)<!--([A-Z0-9_]+)-->", text):
            got, tok = m.group(1), m.group(2)
            seen.add(tok)
            if tok not in vals:
                bad.append(f"{doc}: unknown token {tok}")
            elif got != vals[tok]:
                bad.append(f"{doc}: {tok} is quoted as {got}, profiles/ say {vals[tok]}")
        for m in re.finditer(r"@([A-Z0-9_]+)@", text):
            bad.append(f"{doc}: unfilled placeholder {m.group(0)}")
    return bad, seen


if __name__ == "__main__":
    if "--check" in sys.argv:
        bad, seen = check()
        print("\n".join(bad) or f"{len(seen)} quoted figures agree with profiles/")
        sys.exit(1 if bad else 0)
    vals = values()
    for doc in DOCS:
        p = os.path.join(ROOT, doc)
        text = rewrite(open(p).read(), vals)            # (read first: opening for writing truncates)
        assert text.strip(), p
        open(p, "w").write(text)
    print(f"rewrote {len(vals)} figures in {', '.join(DOCS)}")
