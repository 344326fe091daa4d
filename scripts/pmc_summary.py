"""per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; scripts/gpu_pmc.sh):
    python scripts/pmc_summary.py gpurun_out/pmc/fetch_results.db gpurun_out/pmc/write_results.db out.json
FETCH_SIZE / WRITE_SIZE are in KiB (MI355X_MICROARCH.md, HBM section: hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024); on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (128-byte requests tallied at 64 B), so the
read side is doubled, as that section prescribes. WRITE_SIZE is uncalibrated there and taken as it is."""
import json, sqlite3, sys
from collections import defaultdict

def per_kernel(db, counter):
    c = sqlite3.connect(db).cursor()
    acc = defaultdict(lambda: [0, 0.0])
    try:        # per launch shape for the mat-mul templates (one template serves several matrices)
        rows = list(c.execute("select kernel_name, value, grid_size_x, grid_size_y from counters_collection where counter_name = ?", (counter,)))
    except sqlite3.OperationalError:
        rows = [(a, v, None, None) for a, v in c.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,))]
    for name, val, gx, gy in rows:
        k = name.split("(")[0].replace("void ", "")
        if gx is not None and k.startswith("k_gemm_skinny_q4k"): k += " grid %sx%s" % (gx, gy)
        acc[k][0] += 1; acc[k][1] += val
    return {k: (n, s / n) for k, (n, s) in acc.items()}

fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
    n, f = fetch[k]; w = write.get(k, (0, 0.0))[1]
    out[k] = {"launches": n, "fetch_size_kib_avg": f, "write_size_kib_avg": w,
              "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0, "note": "2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE, KiB -> bytes"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in list(out.items())[:8]:
    print("%-60s n=%5d fetch %10.1f KiB write %9.1f KiB -> %.2f MB per launch" % (k[:60], v["launches"], v["fetch_size_kib_avg"], v["write_size_kib_avg"], v["hbm_bytes_per_launch"] / 1e6))
