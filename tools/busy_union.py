#!/usr/bin/env python3
"""Cross-check of bench.py's roofline against a rocprofv3 kernel trace of the same command.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o trace -- python3 bench.py --side-steps 0 --no-cpu-baseline
    python3 tools/busy_union.py DIR/trace_kernel_trace.csv [kernel-name-substring]

bench.py's `roofline.kernel_ms_per_step` is the time the device had at least one of the timed region's traces in
flight (HIP events per trace, merged by the library); two traces overlap on the device, so `rocprofv3 --stats`'s
average DURATION of a k_generation dispatch (a dispatch shares the device with another for most of its life) is about
twice the time the device spends per launch.  This script derives both from the per-dispatch timestamps:
  * dispatches that overlap another dispatch of the kernel (the overlapped regions): their count, the UNION of their
    intervals, union / count = device time per launch  -> compare with roofline.avg_launch_ms,
    and their average duration                          -> compare with roofline.busy.avg_launch_ms_on_its_stream;
  * dispatches that run alone (one-stream side region, first traces): average duration
                                                        -> compare with roofline.one_stream.avg_launch_ms."""
import csv
import sys


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "k_generation"
    spans = []
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            if want in row["Kernel_Name"]:
                spans.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
    spans.sort()
    if not spans:
        raise SystemExit(f"no dispatch of a kernel named like {want!r} in {path}")
    # which dispatches overlap a neighbour (sorted by start: compare with the furthest end seen so far and with the next start)
    overlapped = [False] * len(spans)
    furthest, owner = -1, -1
    for k, (start, end) in enumerate(spans):
        if start < furthest:
            overlapped[k] = True
            overlapped[owner] = True
        if end > furthest:
            furthest, owner = end, k
    # dispatches of no work (a launch that finds its generation empty exits in its prologue: a few microseconds)
    def summary(flags, label):
        chosen = [s for s, f in zip(spans, flags) if f]
        if not chosen:
            print(f"{label}: none")
            return
        total = sum(e - s for s, e in chosen)
        union, open_from, open_to = 0, chosen[0][0], chosen[0][1]
        for s, e in chosen[1:]:
            if s <= open_to:
                open_to = max(open_to, e)
            else:
                union += open_to - open_from
                open_from, open_to = s, e
        union += open_to - open_from
        n = len(chosen)
        print(f"{label}: {n} dispatches, average duration {total / n / 1e3:.2f} us, union of their intervals {union / 1e6:.3f} ms "
              f"= {union / n / 1e3:.2f} us of device time per launch")

    working = [(e - s) > 8000 for s, e in spans]
    print(f"{path}: {len(spans)} dispatches of *{want}*, {sum(working)} of them longer than 8 us (launches that found rays)")
    summary([o and w for o, w in zip(overlapped, working)], "overlapping another dispatch (traces in flight together)")
    summary([(not o) and w for o, w in zip(overlapped, working)], "alone on the device (one stream, synchronous, first traces)")


if __name__ == "__main__":
    main()
