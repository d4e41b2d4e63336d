"""dev: kernel timeline (name, duration, gap to the previous kernel) of the LAST n launches of a rocprofv3 kernel trace:
   python3 tools/dev/timeline.py <dir with *kernel_trace.csv> [n] [--gaps us]   (--gaps: only launches behind a gap of at least that)"""
import sys, glob, csv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
gaps = float(sys.argv[sys.argv.index("--gaps") + 1]) if "--gaps" in sys.argv else None
if gaps is not None: args = [a for a in args if a != sys.argv[sys.argv.index("--gaps") + 1]]
rows = []
for f in glob.glob(args[0] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
n = int(args[1]) if len(args) > 1 else 40
sel = rows[-n:]
prev = None
busy = idle = 0.0
for s, e, k in sel:
    name = k.split('(')[0].replace('void tc::', '').replace('tc::', '')[:44]
    gap = (s - prev) / 1e3 if prev else 0.0
    busy += (e - s) / 1e3
    idle += max(gap, 0.0)
    if gaps is None or gap >= gaps:
        print(f"{name:46s} {(e - s) / 1e3:7.1f} gap {gap:6.1f}   t = {(s - sel[0][0]) / 1e3:9.1f}")
    prev = e
print(f"span {(sel[-1][1] - sel[0][0]) / 1e3:.1f} us, kernels {busy:.1f} us, gaps {idle:.1f} us")
