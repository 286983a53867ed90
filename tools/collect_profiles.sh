#!/bin/bash
# Run on the GPU box (via gpurun): the judged measurements of one round.
#   1. the default bench line (everything on)                          -> <tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats of the default bench command   -> <tag>_kernel_stats.csv
#   3. in SEPARATE passes the FETCH_SIZE / WRITE_SIZE PMC counters (MI355X_MICROARCH.md: TCC has 4
#      slots, FETCH_SIZE costs 3 and WRITE_SIZE 2 -> one pass each; never mixed with trace domains)
#                                                                       -> <tag>_traffic.json
#   4. the compaction sweep (metric shape x keep) and the other BASELINE configurations
#                                                                       -> <tag>_sweep.jsonl, <tag>_configs.jsonl
# Summaries land in gpurun_out/profiles_<tag>/ ; copy the ones to be judged to profiles/.
set -u
TAG="${1:-r2}"
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/profiles_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
(rocm-smi --showuniqueid --showclocks 2>&1 | grep -E "Unique|mclk|fclk") > "$OUT/${TAG}_box.txt"
cd "$REPO"
timeout 900 python bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.err"
cd /tmp
BENCH="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-adjacent --no-s0 --no-probe --no-engine-cache --no-other-configs --no-native-layout --no-live-traffic --no-parity-gate"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" --output-format csv -- $BENCH > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/stats.log"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" --output-format csv -- $BENCH > /dev/null 2> "$OUT/fetch.log"
timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" --output-format csv -- $BENCH > /dev/null 2> "$OUT/write.log"
python - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
stats = glob.glob(f"{out}/stats/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(stats)))
with open(f"{out}/{tag}_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        if "kvc::" in r["Name"] or "rocclr" in r["Name"]:
            w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                        r["Percentage"], r["MinNs"], r["MaxNs"]])
def pmc(kind, counter):
    f = glob.glob(f"{out}/{kind}/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
res = {}
for k in sorted(set(fetch) | set(write)):
    if "kvc::" not in k:
        continue
    fkb, wkb = fetch.get(k, 0.0), write.get(k, 0.0)
    res[k[:100]] = {"FETCH_SIZE_KB_per_launch": fkb, "WRITE_SIZE_KB_per_launch": wkb,
                    # gfx950: FETCH_SIZE counts 128 B requests as 64 B for wide coalesced
                    # streams (MI355X_MICROARCH.md, HBM) -> doubled; WRITE_SIZE as reported
                    "hbm_bytes_per_launch": (2.0 * fkb + wkb) * 1024.0}
dom = [k for k in res if "compact_runs_kernel" in k]
summary = {"tag": tag, "command": "bench.py --steps 10 --warmup 2 (default workload)", "kernels": res}
if dom:
    summary["dominant_kernel"] = dom[0]
    summary["hbm_bytes_per_launch"] = res[dom[0]]["hbm_bytes_per_launch"]
json.dump(summary, open(f"{out}/{tag}_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "kernels"}))
PY
# 3b. the same two PMC passes for the step in an engine-sized cache (61 x the sequence's blocks, 222 GiB):
#     bench.py's engine_sized_cache.roofline.traffic                      -> <tag>_traffic_engine.json
BENCH_E="$BENCH --spare-blocks 60"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch_e" --output-format csv -- $BENCH_E > /dev/null 2> "$OUT/fetch_e.log"
timeout 600 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write_e" --output-format csv -- $BENCH_E > /dev/null 2> "$OUT/write_e.log"
python - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
def pmc(kind, counter):
    f = glob.glob(f"{out}/{kind}/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
fetch, write = pmc("fetch_e", "FETCH_SIZE"), pmc("write_e", "WRITE_SIZE")
res = {}
for k in sorted(set(fetch) | set(write)):
    if "kvc::" in k:
        fkb, wkb = fetch.get(k, 0.0), write.get(k, 0.0)
        res[k[:100]] = {"FETCH_SIZE_KB_per_launch": fkb, "WRITE_SIZE_KB_per_launch": wkb,
                        "hbm_bytes_per_launch": (2.0 * fkb + wkb) * 1024.0}
dom = [k for k in res if "compact_runs_kernel" in k]
summary = {"tag": tag, "command": "bench.py --steps 10 --warmup 2 --spare-blocks 60 (the default workload's step in a cache of 61 x its blocks)",
           "kernels": res}
if dom:
    summary["dominant_kernel"] = dom[0]
    summary["hbm_bytes_per_launch"] = res[dom[0]]["hbm_bytes_per_launch"]
json.dump(summary, open(f"{out}/{tag}_traffic_engine.json", "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "kernels"}))
PY
# 3b'. the opt-in slot-major in-block layout (KVC_BLOCK_LAYOUT=slot_major): the same step, kernel stats and the two PMC
#      passes of compact_slots_kernel                         -> <tag>_native_kernel_stats.csv, <tag>_native_traffic.json
cd /tmp
BENCH_N="$BENCH --block-layout slot_major"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_n" --output-format csv -- $BENCH_N > "$OUT/${TAG}_native_bench_under_rocprof.json" 2> "$OUT/stats_n.log"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch_n" --output-format csv -- $BENCH_N > /dev/null 2> "$OUT/fetch_n.log"
timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write_n" --output-format csv -- $BENCH_N > /dev/null 2> "$OUT/write_n.log"
python - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
stats = glob.glob(f"{out}/stats_n/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(stats)))
with open(f"{out}/{tag}_native_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        if "kvc::" in r["Name"] or "rocclr" in r["Name"]:
            w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
def pmc(kind, counter):
    f = glob.glob(f"{out}/{kind}/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
fetch, write = pmc("fetch_n", "FETCH_SIZE"), pmc("write_n", "WRITE_SIZE")
res = {}
for k in sorted(set(fetch) | set(write)):
    if "kvc::" in k:
        fkb, wkb = fetch.get(k, 0.0), write.get(k, 0.0)
        res[k[:100]] = {"FETCH_SIZE_KB_per_launch": fkb, "WRITE_SIZE_KB_per_launch": wkb,
                        "hbm_bytes_per_launch": (2.0 * fkb + wkb) * 1024.0}
dom = [k for k in res if "compact_slots_kernel" in k]
summary = {"tag": tag, "command": "bench.py --steps 10 --warmup 2 --block-layout slot_major (default workload, slot-major blocks)", "kernels": res}
if dom:
    summary["dominant_kernel"] = dom[0]
    summary["hbm_bytes_per_launch"] = res[dom[0]]["hbm_bytes_per_launch"]
json.dump(summary, open(f"{out}/{tag}_native_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "kernels"}))
PY
# 3c. config 3 at full size: kernel stats and PMC traffic of the small-eviction schedule
"$REPO/tools/prof_bench.sh" "${TAG}_c3" --config c3 > "$OUT/${TAG}_c3_stats.txt" 2>&1
cp "$REPO/gpurun_out/${TAG}_c3_kernel_stats.csv" "$OUT/" 2>/dev/null
"$REPO/tools/collect_c3_pmc.sh" "$TAG" > "$OUT/${TAG}_c3_pmc.txt" 2>&1
cp "$REPO/gpurun_out/${TAG}_c3_pmc.json" "$OUT/" 2>/dev/null
cd "$REPO"
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-adjacent --no-s0 --no-engine-cache --no-other-configs --no-live-traffic --no-parity-gate"
for shape in perm decay oldest; do
  for keep in 0.5 0.125 0.015625; do
    [ "$shape" = oldest ] && [ "$keep" = 0.015625 ] && continue
    timeout 300 $B --metric-shape $shape --keep $keep >> "$OUT/${TAG}_sweep.jsonl" 2>> "$OUT/sweep.err"
  done
done
for cfg in "--spare-blocks 30" "--spare-blocks 60" "--batch 4" "--batch 16 --steady-cap 4096" "--batch 64 --steady-cap 4096" "--config c3" "--config c3 --mode reference" "--config c3 --lean" "--config c3i" "--config c4" "--config c5" "--layers 80 --seq-len 16384 --batch 4"; do
  timeout 900 $B $cfg >> "$OUT/${TAG}_configs.jsonl" 2>> "$OUT/configs.err"
done
timeout 600 python tools/bench_attention.py --json "$OUT/${TAG}_attention_bench.json" > /dev/null 2> "$OUT/attention.err"
timeout 600 python tools/bench_attention.py --block-layout slot_major --json "$OUT/${TAG}_attention_bench_slot_major.json" > /dev/null 2>> "$OUT/attention.err"
(python tools/cmp_attention_layouts.py "$OUT/${TAG}_attention_bench.json" "$OUT/${TAG}_attention_bench_slot_major.json" > "$OUT/${TAG}_attention_layouts.txt") 2>> "$OUT/attention.err"
# 5. config 3 as a whole decode step (S0 + S1 + S2 + S3), two sweeps of the store against harvest-ahead: kernel stats
#    and the comparison itself                                             -> <tag>_decode_step_c3.json, _kernel_stats.csv
"$REPO/tools/prof_decode_step.sh" "${TAG}_decode_step" > "$OUT/${TAG}_decode_step_stats.txt" 2>&1
cp "$REPO/gpurun_out/${TAG}_decode_step.json" "$OUT/${TAG}_decode_step_c3.json" 2>/dev/null
cp "$REPO/gpurun_out/${TAG}_decode_step_kernel_stats.csv" "$OUT/" 2>/dev/null
cd "$REPO"
"$REPO/tools/collect_decode_step_pmc.sh" "$TAG" > "$OUT/${TAG}_decode_step_pmc.txt" 2>&1
cp "$REPO/gpurun_out/${TAG}_decode_step_pmc.json" "$OUT/" 2>/dev/null
# 5a. the launches of one decode step in order (S1 on the lists the aggregation pass / the attention's epilogue made)  -> <tag>_s1_timeline.txt
(cd /tmp && rm -rf /tmp/tl_$TAG && timeout 600 rocprofv3 --kernel-trace -d /tmp/tl_$TAG --output-format csv -- python "$REPO/tools/decode_step.py" > /dev/null 2> "$OUT/timeline.err";
 cd "$REPO" && (echo "# fork flow: aggregate_decode() harvests for the next call, schedule_evictions on its lists"; python tools/s1_timeline.py /tmp/tl_$TAG aggregate_harvest_kernel;
                echo "# zero-sweep step: the last layer's attention launch with the harvesting epilogue, then the schedule"; python tools/s1_timeline.py /tmp/tl_$TAG paged_attention | tail -9)) > "$OUT/${TAG}_s1_timeline.txt" 2>> "$OUT/timeline.err"
cd "$REPO"
# 5b. the decode step in the fork's default mode (the reference's batch > 1 rule), with the oracle's two-stage verdict
timeout 600 python tools/decode_step.py --mode reference > "$OUT/${TAG}_decode_step_c3_reference_mode.json" 2> "$OUT/ds_ref.err"
# 5c. the zero-sweep decode step over 400 iterations of an evolving state: fused attention + epilogue harvest against the reference flow
timeout 600 python tools/soak_attention_harvest.py 400 8 4 512 10 > "$OUT/${TAG}_attention_harvest_soak.json" 2> "$OUT/soak_att.err"
# 6. harvest-ahead / pivot memory over 400 decode steps of an evolving on-device block state    -> <tag>_harvest_soak.txt
(timeout 500 python tools/soak_harvest.py 400 32 32 1024 uniform; timeout 500 python tools/soak_harvest.py 400 32 32 1024 peaky;
 timeout 300 python tools/soak_harvest.py 300 4 4 256 peaky) > "$OUT/${TAG}_harvest_soak.txt" 2> "$OUT/soak.err"
ls "$OUT"
