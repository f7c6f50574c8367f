import json
d=json.loads(open("gpurun_out/r05_bench_B256.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["value"], r["kernel"], r["frac"], r["frac_f32_equiv"], r["kernel_avg_us"])
for k in r["by_kernel"]: print(k)
