"""dev: print the kernel timeline (name, duration, gap to the previous kernel) of the LAST n launches of a rocprofv3 kernel trace:
   python3 tools/dev/timeline.py <dir with *kernel_trace.csv> [n]"""
import sys, glob, csv
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
prev = None
for s, e, k in rows[-n:]:
    name = k.split('(')[0].replace('void tc::', '').replace('tc::', '')[:44]
    print(f"{name:46s} {(e - s) / 1e3:7.1f} gap {((s - prev) / 1e3 if prev else 0):6.1f}")
    prev = e
