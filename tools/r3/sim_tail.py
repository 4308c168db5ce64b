#!/usr/bin/env python3
"""Event-driven model of one k_msm_accumulate launch on 256 CUs x 2 workgroup slots, calibrated on the per-wave
trace of round 3 (profiles/r03_msm_trace_*.json): of two waves on a SIMD the OLDER one is served first and gets
~82 % of the issue capacity, the younger the rest; a lone wave gets 82 %.  Used to choose the size schedule of the
workgroups (whole blobs first, finer pieces last) before measuring it."""
import heapq
import sys

P_OLD, P_LONE = 0.82, 0.82
FOLD = 2.5 / 256      # a workgroup's closing fold, in units of one whole-blob workgroup's additions


def simulate(sizes, cus=256):
    """sizes: work of each workgroup in dispatch order (1.0 = 256 additions per lane).  Returns makespan in units
    of (whole-blob work / SIMD capacity)."""
    queue = list(sizes)[::-1]
    # per CU: list of [remaining, seq]
    res = [[] for _ in range(cus)]
    seq = 0
    t = 0.0
    for c in range(cus):
        for _ in range(2):
            if queue:
                res[c].append([queue.pop() + FOLD, seq])
                seq += 1
    while True:
        # next completion over all CUs
        best_dt, best = None, None
        for c in range(cus):
            r = res[c]
            if not r:
                continue
            if len(r) == 1:
                dt = r[0][0] / P_LONE
                cand = (dt, c, 0)
            else:
                o, y = (0, 1) if r[0][1] < r[1][1] else (1, 0)
                dto, dty = r[o][0] / P_OLD, r[y][0] / (1 - P_OLD)
                cand = (dto, c, o) if dto <= dty else (dty, c, y)
            if best_dt is None or cand[0] < best_dt:
                best_dt, best = cand[0], cand
        if best is None:
            return t
        dt = best_dt
        # advance everyone
        for c in range(cus):
            r = res[c]
            if len(r) == 1:
                r[0][0] -= dt * P_LONE
            elif len(r) == 2:
                o, y = (0, 1) if r[0][1] < r[1][1] else (1, 0)
                r[o][0] -= dt * P_OLD
                r[y][0] -= dt * (1 - P_OLD)
        t += dt
        for c in range(cus):
            r = res[c]
            keep = [x for x in r if x[0] > 1e-12]
            freed = len(r) - len(keep)
            res[c] = keep
            for _ in range(freed):
                if queue:
                    res[c].append([queue.pop() + FOLD, seq])
                    seq += 1


def shape(nvec, nfine, split):
    return [1.0] * (nvec - nfine) + [1.0 / split] * (nfine * split)


if __name__ == "__main__":
    nvec = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    ideal = nvec / 512.0 * 2 / 2   # total work / capacity: nvec blobs over 256 CUs x (capacity 1 per slot pair)
    ideal = nvec / 256.0
    print("nvec", nvec, "ideal", round(ideal, 3))
    for nfine, split in [(0, 1), (nvec, 2), (nvec, 4), (64, 8), (128, 8), (128, 4), (256, 4), (256, 8), (64, 16), (128, 16), (256, 2), (512, 2), (512, 4)]:
        if nfine > nvec:
            continue
        m = simulate(shape(nvec, nfine, split))
        print("fine %4d x %2d -> %d wgs  makespan %.3f  (%.1f %% over ideal)" % (nfine, split, nvec - nfine + nfine * split, m, (m / ideal - 1) * 100))
