"""Prints the kernel timeline of one frame from a rocprofv3 kernel trace (CSV): start offset, duration, gap to the previous kernel
on the same stream / queue.   usage: python tools/frame_timeline.py <dir with *_kernel_trace.csv> [frame_index]"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    which = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "avg" else -3
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    name = lambda r: r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
    # a frame starts at every hist_kernel
    starts = [i for i, r in enumerate(rows) if name(r) == "hist_kernel"]
    if len(sys.argv) > 2 and sys.argv[2] == "avg":
        # sequential schedule: average device-side spans over the frames (first / last few dropped)
        fe, up, tot = [], [], []
        for a, b in zip(starts[5:-3], starts[6:-2]):
            fr = rows[a:b]
            nm = [name(r) for r in fr]
            jn = "jacobian_nullspace_kernel" if "jacobian_nullspace_kernel" in nm else "jacobian_kernel"
            if jn not in nm or "ekf_commit_kernel" not in nm:
                continue
            j = nm.index(jn)
            c = len(nm) - 1 - nm[::-1].index("ekf_commit_kernel")
            r_end = max(i for i, x in enumerate(nm) if x == "ransac_select_kernel")
            t0 = int(fr[0]["Start_Timestamp"])
            fe.append((int(fr[r_end]["End_Timestamp"]) - t0) / 1e3)
            # the update chain: from the copy in front of the jacobian kernel to the end of the commit kernel
            js = int(fr[j]["Start_Timestamp"])
            busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in fr[j:c + 1]) / 1e3
            up.append(busy)   # device-busy time of the chain (Jacobians .. commit), host-side gaps excluded
            tot.append((int(fr[c]["End_Timestamp"]) - t0) / 1e3)
        import statistics
        print(f"frames {len(up)}: front-end span {statistics.mean(fe):.1f} us, update chain span {statistics.mean(up):.1f} us "
              f"(median {statistics.median(up):.1f}), hist..commit {statistics.mean(tot):.1f} us")
        return
    a = starts[which]
    b = starts[which + 1] if which + 1 < 0 or which + 1 < len(starts) else len(rows)
    t0 = int(rows[a]["Start_Timestamp"])
    last_end = {}
    print(f"{'kernel':28s} {'queue':>6s} {'start_us':>9s} {'dur_us':>8s} {'gap_us':>8s}")
    for r in rows[a:b]:
        q = r.get("Queue_Id", "0")
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
        print(f"{name(r):28s} {q:>6s} {(s - t0) / 1e3:9.2f} {(e - s) / 1e3:8.2f} {gap:8.2f}")
        last_end[q] = e
    print(f"frame span: {(max(int(r['End_Timestamp']) for r in rows[a:b]) - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
