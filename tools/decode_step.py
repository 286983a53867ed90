#!/usr/bin/env python3
"""config 3 as a whole decode step (S0 + S1 + S2 + S3), two sweeps of the store against harvest-ahead:
bench.py's decode_step_compare on its own (bench.py reports it inside other_configs[c3]).
    python tools/decode_step.py [--batch 64] [bench.py flags]"""
import copy
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    args = bench.parse_args(["--config", "c3"] + sys.argv[1:])
    a2 = copy.copy(args)
    a2.config = "c3"
    res = bench.measure_workload(a2, 2000, "cuda:0", 10, 2, False)
    if res is None:
        raise SystemExit("does not fit")
    print(json.dumps({"stages_ms": res["stages_ms"], "S1_schedule": res["S1_schedule"], "candidate_slots": res["candidate_slots"],
                      "S1_sampled_pivots": res.get("S1_sampled_pivots"), "S1_schedule_reason": res.get("S1_schedule_reason"),
                      "S1_call_forms": res.get("S1_call_forms"), "decode_step": res.get("decode_step")}))


if __name__ == "__main__":
    main()
