import json,sys
a=json.load(open("gpurun_out/r6_attention_bench.json")); b=json.load(open(sys.argv[1]))
for x,y in zip(a,b):
    print(x["num_seqs"],x["context_len"],x["num_heads"],x["block_size"],x["kv_cache_dtype"],x["record_kv_metrics"],x["fused_metric_aggregation"], round(x["ms_per_layer_step"],4), round(y["ms_per_layer_step"],4), round(y["ms_per_layer_step"]/x["ms_per_layer_step"],3), round(x["frac_of_8TBps"],3), round(y["frac_of_8TBps"],3))
