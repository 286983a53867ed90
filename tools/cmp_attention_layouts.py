#!/usr/bin/env python3
"""Decode attention in the two in-block cache layouts side by side (two runs of tools/bench_attention.py on the same box):
    python tools/bench_attention.py --json ref.json; python tools/bench_attention.py --block-layout slot_major --json sm.json
    python tools/cmp_attention_layouts.py ref.json sm.json
(one argument: compared with gpurun_out/r6_attention_bench.json)"""
import json
import sys


def main():
    ref = sys.argv[1] if len(sys.argv) > 2 else "gpurun_out/r6_attention_bench.json"
    other = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    a, b = json.load(open(ref)), json.load(open(other))
    print("seqs ctx Hq bs cache record fused | ms reference | ms slot-major | ratio | frac of 8 TB/s (reference, slot-major)")
    for x, y in zip(a, b):
        print(x["num_seqs"], x["context_len"], x["num_heads"], x["block_size"], x["kv_cache_dtype"], x["record_kv_metrics"],
              x["fused_metric_aggregation"], round(x["ms_per_layer_step"], 4), round(y["ms_per_layer_step"], 4),
              round(y["ms_per_layer_step"] / x["ms_per_layer_step"], 3), round(x["frac_of_8TBps"], 3), round(y["frac_of_8TBps"], 3))


if __name__ == "__main__":
    main()
