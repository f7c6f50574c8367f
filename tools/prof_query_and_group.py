"""BASELINE's second metric under the counters: HBM bytes per launch of the materialising ball-query + group kernel
(`ball_query_cells_kernel`, pointnet2_utils.query_and_group at configs[3]) from two rocprofv3 PMC passes, merged into the round's
traffic file under the key "query_and_group" (bench.py reads roofline.sa_kernel_hbm.traffic from there).
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/qg_fetch -o r -- python tools/prof_query_and_group.py run
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/qg_write -o r -- python tools/prof_query_and_group.py run
    python tools/prof_query_and_group.py merge /tmp/qg_fetch /tmp/qg_write profiles/rNN_traffic.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    if sys.argv[1] == "run":
        import bench
        r = bench.sa_kernel_hbm(iters=10)
        print(json.dumps({k: r[k] for k in ("achieved", "frac", "launch_ms") if k in r}))
        return
    from tools.prof_traffic import per_kernel
    fetch_dir, write_dir, out = sys.argv[2:5]
    f, w = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    key = [k for k in f if "ball_query_cells" in k]
    assert key, sorted(f)
    k = key[0]
    res = json.load(open(out)) if os.path.exists(out) else {}
    res["query_and_group"] = {"kernel": k, "launches": f[k][0], "fetch_kb": round(f[k][1], 1), "write_kb": round(w[k][1], 1),
                              "bytes_per_launch": round((2 * f[k][1] + w[k][1]) * 1024.0, 0), "avg_us_under_pmc": round(f[k][2], 2),
                              "bytes_per_launch_fetch_as_counted": round((f[k][1] + w[k][1]) * 1024.0, 0),
                              "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of tools/prof_query_and_group.py run "
                                     "(configs[3]: B=128, N=4096, npoint 512, nsample 64, C=4); FETCH_SIZE doubled (gfx950)"}
    json.dump(res, open(out, "w"), indent=1)
    print("query_and_group (%s): fetch %.1f KB write %.1f KB -> %.2f MB / launch, %.1f us under PMC" % (
        k, f[k][1], w[k][1], res["query_and_group"]["bytes_per_launch"] / 1e6, f[k][2]))


if __name__ == "__main__":
    main()
