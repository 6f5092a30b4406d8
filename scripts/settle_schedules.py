#!/usr/bin/env python3
"""Dev helper (GPU box): the launch schedule every bench workload settles at -> profiles/schedules.json.
   python scripts/settle_schedules.py [out.json] [starts]
For each of BASELINE.json's configs (and the shares one GPU of a 2 / 4 / 8-GPU run renders) a fresh context runs mrt_redraw
until the library's controller calls its schedule final (mrt_get_schedule), `starts` times over; the setting most starts agree
on is written, with every start's outcome and rate beside it, keyed "<config>_n<gpus>" (what bench.py pins with
mrt_set_schedule_hint unless --schedule measure)."""
import collections, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import myraytracer_amd as M

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "schedules.json")
starts = int(sys.argv[2]) if len(sys.argv) > 2 else 3
WORK = [   # key, scene, w, h, spp, depth, shard
    ("c1_n1", "default", 400, 225, 16, 8, None),
    ("c2_n1", "cover", 1200, 675, 64, 50, None),
    ("c3_n1", "cover-glass", 1920, 1080, 512, 50, None),
    ("c3_n2", "cover-glass", 2716, 1528, 512, 50, (0, 2)),
    ("c3_n4", "cover-glass", 3840, 2160, 512, 50, (0, 4)),
    ("c3_n8", "cover-glass", 3840, 2160, 1024, 50, (0, 8)),
    ("c4_n1", "cover-glass", 3840, 2160, 1024, 50, None),
    ("c4_n8", "cover-glass", 3840, 2160, 1024, 50, (0, 8)),
    ("c5_n1", "stress", 1920, 1080, 4096, 50, None),
    ("c5_n2", "stress", 1920, 1080, 4096, 50, (0, 2)),
    ("c5_n4", "stress", 1920, 1080, 4096, 50, (0, 4)),
    ("c5_n8", "stress", 1920, 1080, 4096, 50, (0, 8)),
    ("interactive_n1", "cover-glass", 1920, 1080, 1, 50, None),
]
only = os.environ.get("MRT_ONLY")
result = {}
for key, scene, w, h, spp, depth, shard in WORK:
    if only and key not in only.split(","):
        continue
    sp, cam = (M.scene_cover(1, scene == "cover-glass") if scene.startswith("cover") else M.scene_stress(1, 100) if scene == "stress" else (M.scene_default(), None))
    runs = []
    for _ in range(starts):
        with M.State(M.Args(w, h, spp, depth, 1.0), seed=1, shard=shard) as st:
            st.set_world(sp)
            if cam is not None: st.set_camera(cam)
            st.set_draw_counting(False)
            n, t0 = 0, time.perf_counter()
            while not st.get_schedule()["settled"] and n < 2000 and time.perf_counter() - t0 < 150.0:
                st.redraw(); n += 1
            st.sync()
            sch = st.get_schedule()
            k = max(4, min(400, int(1.0 / max(1e-4, (time.perf_counter() - t0) / max(1, n)))))      # about a second of frames
            t1 = time.perf_counter()
            for _ in range(k): st.redraw()
            st.sync()
            dt = time.perf_counter() - t1
            px = w * h if shard is None else sum(min(8, h - 8 * b) * w for b in range((h + 7) // 8) if b % shard[1] == shard[0])
            runs.append({"div": sch["div"], "mult": sch["mult"], "final": sch["settled"], "frames_to_settle": n,
                         "msamples_per_s": round(px * spp * k / dt * 1e-6, 1)})
    votes = collections.Counter((r["div"], r["mult"]) for r in runs if r["final"])
    if votes:
        (div, mult), _ = votes.most_common(1)[0]
        result[key] = {"div": div, "mult": mult, "starts": runs}
    else:
        result[key] = {"div": runs[0]["div"], "mult": runs[0]["mult"], "starts": runs, "note": "no start reached a final schedule within the cap"}
    print(key, result[key], flush=True)
if only and os.path.exists(out_path):
    merged = json.load(open(out_path)); merged.update(result); result = merged
json.dump(result, open(out_path, "w"), indent=1, sort_keys=True)
