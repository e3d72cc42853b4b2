"""Timeline of the kernels of a few steady-state steps out of a rocprofv3 --kernel-trace database (rocpd .db) of
tools/exchange_contention.py: when RCCL's kernels start and end relative to the product's kernels around them.

    python3 tools/exchange_timeline.py gpurun_out/.../x_results.db [--steps 2] [--skip 0.5]

Prints, for `--steps` consecutive steps taken after fraction `--skip` of the launches of pair_finish, one line per kernel
dispatch: start and end in us relative to the end of the step's first pair_finish, duration, workgroups, stream, short name; then
per RCCL kernel: the delay between the end of the kernel it had to wait for (the latest kernel that ended before it started)
and its own start."""
import argparse
import re
import sqlite3


def short(name):
    m = re.search(r"(pair_forces|pair_finish|pair_reduce|integrate_bodies_\w+|ncclDevKernel_\w+|__amd_rocclr_\w+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) [:60] if m else name[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--skip", type=float, default=0.5)
    ap.add_argument("--anchor", default="pair_finish")
    args = ap.parse_args()
    cur = sqlite3.connect(args.db).cursor()
    rows = cur.execute("select name, start, end, grid_x * grid_y * grid_z / (workgroup_x * workgroup_y * workgroup_z), stream, vgpr_count from kernels order by start").fetchall()
    anchors = [k for k, r in enumerate(rows) if args.anchor in r[0]]
    if len(anchors) < args.steps + 2:
        raise SystemExit(f"only {len(anchors)} launches of {args.anchor} in the trace")
    first = anchors[int(len(anchors) * args.skip)]
    last = anchors[int(len(anchors) * args.skip) + args.steps]
    t0 = rows[first][2]
    print(f"# {args.db}: dispatches {first}..{last} of {len(rows)}; times in us relative to the end of dispatch {first} ({short(rows[first][0])})")
    print(f"# {'start':>9} {'end':>9} {'dur':>8} {'wgs':>5} {'vgpr':>4}  stream      kernel")
    for name, start, end, wgs, stream, vgpr in rows[first:last + 1]:
        print(f"  {(start - t0) / 1e3:9.1f} {(end - t0) / 1e3:9.1f} {(end - start) / 1e3:8.1f} {wgs:5d} {vgpr:4d}  {stream:<10}  {short(name)}")
    print("# RCCL kernels: wait between the end of the latest kernel that ended before the start, and the start (what queueing for a CU costs)")
    waits = []
    for k, (name, start, end, wgs, stream, vgpr) in enumerate(rows):
        if "ncclDevKernel" not in name or k < first:
            continue
        before = max((r[2] for r in rows[max(0, k - 12):k] if r[2] <= start), default=None)
        running = [short(r[0]) for r in rows[max(0, k - 12):k + 12] if r[1] < start < r[2]]
        if before is not None:
            waits.append(((start - before) / 1e3, (end - start) / 1e3, wgs, ",".join(sorted(set(running))) or "-"))
    for w in waits[:12]:
        print(f"  started {w[0]:7.1f} us after the previous kernel end; ran {w[1]:7.1f} us; {w[2]} workgroups; meanwhile on the chip: {w[3]}")
    if waits:
        durs = sorted(w[1] for w in waits)
        print(f"# {len(waits)} RCCL launches: duration median {durs[len(durs) // 2]:.1f} us, max {durs[-1]:.1f} us")


if __name__ == "__main__":
    main()
