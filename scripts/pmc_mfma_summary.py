"""matrix-pipe / VALU evidence per kernel from the third pass of scripts/gpu_pmc.sh:
    python scripts/pmc_mfma_summary.py gpurun_out/pmc/mfma_results.db out.json
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GUI_ACTIVE * 1024 SIMDs) as rocprofv3's derived metric defines it; GRBM_GUI_ACTIVE comes
back SUMMED over the 8 XCDs (1.72 M "cycles" for a 90 us kernel = 8 x 215 k), hence the / 8. VALU issue utilisation =
SQ_INSTS_VALU x 4 cycles (a wave64 instruction occupies its SIMD's issue for 4 cycles) over the same SIMD-cycles."""
import json, sqlite3, sys
from collections import defaultdict
c = sqlite3.connect(sys.argv[1]).cursor()
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
try:        # per launch shape when the view carries the grid (the 128- and 2048-token prompts launch the same GEMM templates)
    rows = list(c.execute("select kernel_name, counter_name, value, grid_size_x, grid_size_y from counters_collection"))
except sqlite3.OperationalError:
    try:
        rows = [(a, b, v, gx, 0) for a, b, v, gx in c.execute("select kernel_name, counter_name, value, grid_size from counters_collection")]
    except sqlite3.OperationalError:
        rows = [(a, b, v, None, None) for a, b, v in c.execute("select kernel_name, counter_name, value from counters_collection")]
for name, counter, val, gx, gy in rows:
    k = name.split("(")[0].replace("void ", "")
    if gx is not None and k.startswith(("k_gemm", "k_attention")): k += " grid %sx%s" % (gx, gy)
    acc[k][counter][0] += 1; acc[k][counter][1] += val
out = {}
for k, d in acc.items():
    avg = {cn: v[1] / v[0] for cn, v in d.items()}
    n = max(v[0] for v in d.values())
    gui = avg.get("GRBM_GUI_ACTIVE", 0.0)
    o = {"launches": n, **{cn: avg[cn] for cn in sorted(avg)}}
    if gui:
        simd_cycles = gui / 8.0 * 1024.0
        o["mfma_util_pct"] = 100.0 * avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles
        o["valu_active_pct_of_simd_cycles"] = 100.0 * avg.get("SQ_INSTS_VALU", 0.0) * 4 / simd_cycles
    out[k] = o
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU_MFMA_I8", 0) * kv[1]["launches"])[:6]:
    print("%-34s n=%4d  MFMA(i8) instr %10.0f  VALU instr %12.0f  MfmaUtil %5.2f %%  VALU-active %5.1f %% of SIMD cycles" % (
        k[:34], v["launches"], v.get("SQ_INSTS_VALU_MFMA_I8", 0), v.get("SQ_INSTS_VALU", 0), v.get("mfma_util_pct", 0), v.get("valu_active_pct_of_simd_cycles", 0)))
