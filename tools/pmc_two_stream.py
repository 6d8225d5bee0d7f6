#!/usr/bin/env python3
"""Chip-wide utilisation of the DEFAULT (two-stream) mode, profiles/rNN_prof/pmc_two_stream.md.

rocprofv3's dispatch counters serialise kernels, so the two-stream region cannot be counted directly.  The WORK of a
forward does not depend on how its kernels are interleaved, so: sum the per-dispatch counters of N serialised forwards
(tools/run_forward.py --n N --batch 8: SQ_ACTIVE_INST_VALU, SQ_VALU_MFMA_BUSY_CYCLES, SQ_LDS_IDX_ACTIVE, SQ_WAVE_CYCLES,
SQ_BUSY_CYCLES; HBM bytes from traffic.json) and divide by the wall time the same work takes in each mode -- the
one-stream and the two-stream bench (no profiler attached) -- times the shader clock rocm-smi reported DURING that
timed region.  usage: pmc_two_stream.py <sq1 dir> <sq2 dir> <traffic.json> <bench_two_stream.json> <bench_one_stream.json> <n forwards>"""
import collections, csv, glob, json, sys

sq1, sq2, traffic_f, b2_f, b1_f, nfwd = sys.argv[1:7]
nfwd = int(nfwd)
SIMDS, CUS, SES = 1024, 256, 32
tot = collections.Counter()
for d in (sq1, sq2):
    seen = set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("at::") or "elementwise" in r["Kernel_Name"]:
                continue
            key = (d, r["Counter_Name"])
            # SQ_BUSY_CYCLES is collected in both passes: count it from the first only
            if r["Counter_Name"] == "SQ_BUSY_CYCLES" and d == sq2:
                continue
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
per_fwd = {k: v / nfwd for k, v in tot.items()}
traffic = json.load(open(traffic_f))
hbm = traffic["total_hbm_bytes_per_forward"]
b2, b1 = json.load(open(b2_f)), json.load(open(b1_f))


def mode(b):
    B = b["config"]["slices_per_gpu_per_step"]
    steps = int(b["metric"].split(",")[1].split()[0])
    t_fwd8 = b["ms_per_step"] * 1e-3 / steps / (B / 8)          # seconds per batch-8 forward's worth of work
    box = b.get("box_during_untimed_replay") or b.get("box_during_timed_region") or {}
    clk = (box.get("sclk_mhz_mean") or 2400.0) * 1e6
    cyc = t_fwd8 * clk
    return {"slices_per_s": b["value"], "t_fwd8_ms": t_fwd8 * 1e3, "sclk_mhz": clk / 1e6, "power_w": box.get("socket_power_w_mean"),
            "valu": 4 * per_fwd.get("SQ_ACTIVE_INST_VALU", 0) / (cyc * SIMDS),
            "mfma": per_fwd.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * SIMDS),
            "lds": per_fwd.get("SQ_LDS_IDX_ACTIVE", 0) / (cyc * CUS),
            "occ": 4 * per_fwd.get("SQ_WAVE_CYCLES", 0) / (cyc * SIMDS),
            "hbm_tbs": hbm / t_fwd8 / 1e12}


m1, m2 = mode(b1), mode(b2)
ser_cyc = per_fwd.get("SQ_BUSY_CYCLES", 0) / SES
print("# Chip-wide utilisation of the benchmarked (two-stream) mode -- estimated from serialised counters\n")
print(__doc__.split("usage:")[0].strip().replace("\n", " ") + "\n")
print("| | one stream, batch 8 | two streams, 2 x batch 8 (default) |")
print("|---|---|---|")
print(f"| slices/s (bench, no profiler) | {m1['slices_per_s']:.2f} | {m2['slices_per_s']:.2f} |")
print(f"| wall time per batch-8 forward | {m1['t_fwd8_ms']:.2f} ms | {m2['t_fwd8_ms']:.2f} ms |")
print(f"| shader clock during the timed region (rocm-smi) | {m1['sclk_mhz']:.0f} MHz | {m2['sclk_mhz']:.0f} MHz |")
print(f"| socket power during the timed region | {m1['power_w']} W | {m2['power_w']} W |")
print(f"| VALU issue busy (SQ_ACTIVE_INST_VALU x 4 / SIMD cycles) | {m1['valu']:.0%} | {m2['valu']:.0%} |")
print(f"| MFMA busy (SQ_VALU_MFMA_BUSY_CYCLES / SIMD cycles) | {m1['mfma']:.0%} | {m2['mfma']:.0%} |")
print(f"| VALU + MFMA (they share a SIMD's issue port) | {m1['valu'] + m1['mfma']:.0%} | {m2['valu'] + m2['mfma']:.0%} |")
print(f"| LDS active (SQ_LDS_IDX_ACTIVE / CU cycles) | {m1['lds']:.0%} | {m2['lds']:.0%} |")
print(f"| resident waves per SIMD (SQ_WAVE_CYCLES x 4 / SIMD cycles) | {m1['occ']:.1f} | {m2['occ']:.1f} |")
print(f"| HBM traffic rate ({hbm / 1e9:.1f} GB per batch-8 forward, traffic.json) | {m1['hbm_tbs']:.2f} TB/s | {m2['hbm_tbs']:.2f} TB/s |")
print(f"\nSerialised under the profiler the same forward takes {ser_cyc / 1e6:.2f} M shader cycles (sum of SQ_BUSY_CYCLES / 32 shader engines).")
