#!/usr/bin/env python3
"""Occupancy of every SIMD over one k_msm_accumulate launch, from the per-wave records a -DCKZG_MSM_TRACE build
(tools/build_variant.sh trace -DCKZG_MSM_TRACE) dumps to $CKZG_HIP_MSM_TRACE_FILE: four u64 per wave
{s_memrealtime at entry, at exit, HW_REG_HW_ID, HW_REG_XCC_ID}.  Answers where the gap between the register
budget (2 waves per SIMD) and SQ_WAVE_CYCLES' 1.7 comes from: placement, ramp, rounds or tail.
usage: python tools/msm_trace.py trace.bin [ticks_per_us=100]"""
import collections
import json
import struct
import sys


def main():
    raw = open(sys.argv[1], "rb").read()
    tpu = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0   # s_memrealtime: 100 MHz
    n = len(raw) // 32
    rec = [struct.unpack_from("<4Q", raw, 32 * i) for i in range(n)]
    rec = [r for r in rec if r[0] and r[1]]
    t_min = min(r[0] for r in rec)
    t_max = max(r[1] for r in rec)
    span = (t_max - t_min) / tpu
    dur = sorted((r[1] - r[0]) / tpu for r in rec)
    simd = collections.defaultdict(list)
    cu = collections.defaultdict(int)
    for t0, t1, hw, xcc in rec:
        key = (xcc & 0xf, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf, (hw >> 4) & 3)
        simd[key].append((t0, t1))
        cu[key[:4]] += 1
    # time-weighted occupancy per SIMD
    occ_time = collections.defaultdict(float)   # waves resident -> SIMD-microseconds
    bins = 40
    timeline = [0.0] * bins
    for key, iv in simd.items():
        ev = sorted([(a, 1) for a, _ in iv] + [(b, -1) for _, b in iv])
        cur, last = 0, t_min
        for t, d in ev:
            occ_time[cur] += (t - last) / tpu
            last = t
            cur += d
        occ_time[0] += (t_max - last) / tpu
        for a, b in iv:
            for k in range(bins):
                lo = t_min + (t_max - t_min) * k / bins
                hi = t_min + (t_max - t_min) * (k + 1) / bins
                ov = max(0, min(b, hi) - max(a, lo))
                timeline[k] += ov / (hi - lo)
    nsimd = len(simd)
    total = sum(occ_time.values())
    mean_res = sum(k * v for k, v in occ_time.items()) / total
    starts = sorted((r[0] - t_min) / tpu for r in rec)
    out = {
        "waves": len(rec), "simds_seen": nsimd, "cus_seen": len(cu), "span_us": round(span, 1),
        "wave_us": {"min": round(dur[0], 1), "p10": round(dur[len(dur) // 10], 1), "median": round(dur[len(dur) // 2], 1),
                    "p90": round(dur[9 * len(dur) // 10], 1), "max": round(dur[-1], 1)},
        "mean_waves_resident_per_seen_simd": round(mean_res, 3),
        "mean_waves_resident_per_1024_simds": round(mean_res * nsimd / 1024, 3),
        "simd_time_share_by_resident_waves": {str(k): round(v / total, 4) for k, v in sorted(occ_time.items())},
        "waves_per_cu_histogram": dict(sorted(collections.Counter(cu.values()).items())),
        "waves_per_simd_histogram": dict(sorted(collections.Counter(len(v) for v in simd.values()).items())),
        "xcc_histogram": dict(sorted(collections.Counter(k[0] for k in simd).items())),
        "start_us_deciles": [round(starts[len(starts) * k // 10], 1) for k in range(10)] + [round(starts[-1], 1)],
        "resident_waves_per_simd_over_time_%d_bins" % bins: [round(x / max(nsimd, 1), 2) for x in timeline],
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
