#!/usr/bin/env python3
"""Copy the summaries of the last tools/prof.sh r02 run (gpurun_out/r02/) into profiles/r02/ (what is committed)."""
import glob, os, shutil, subprocess, json, re
for d in glob.glob('gpurun_out/r02/*/vm'):
    files = glob.glob(d + '/*')
    newest = int(os.path.basename(max(files, key=os.path.getmtime)).split("_")[0])
    for f in files:
        if int(os.path.basename(f).split('_')[0]) != newest:
            os.remove(f)
shutil.copy(glob.glob('gpurun_out/r02/kt/vm/*kernel_stats.csv')[0], 'profiles/r02/kernel_stats.csv')
out = subprocess.run(['python3', 'tools/prof_summary.py', 'gpurun_out/r02', '65536'], capture_output=True, text=True).stdout
open('profiles/r02/rocprofv3_summary.txt', 'w').write(out)
shutil.copy('profiles/hbm_traffic.json', 'profiles/r02/hbm_traffic.json')
line = [l for l in open('gpurun_out/r02/bench_kt.log') if l.startswith('{"metric')][-1]
open('profiles/r02/bench_default.json', 'w').write(line)
shutil.copy('gpurun_out/r02/series.txt', 'profiles/r02/stream_series.txt')
old = open('profiles/r02/lean_trip_profile.txt').read()
hdr = "".join(l + "\n" for l in old.split("\n") if l.startswith("#"))
body = "\n".join(l for l in open('gpurun_out/r02/lean_profile.txt').read().split("\n") if "amdgpu.ids" not in l)
open('profiles/r02/lean_trip_profile.txt', 'w').write(hdr + body)
rows = []
for l in open('gpurun_out/r02/series.txt'):
    m = re.match(r"(\d+)\s+kernel (.+?) ([\d.]+) ms\s+step_ms ([\d.]+)\s+GB/s ([\d.]+)", l)
    if m:
        rows.append({"streams": int(m.group(1)), "kernel": m.group(2), "kernel_ms": float(m.group(3)), "step_ms": float(m.group(4)), "GBps": float(m.group(5))})
json.dump({"what": "single-GPU time of one batch over batch sizes (cfg2 streams, tools/series.sh): the cost of one shard of a strong-scaling run; a launch lasts as long as its longest stream, so 8,192 streams take about three quarters of the time of 65,536",
           "rows": rows}, open('profiles/r02/stream_series.json', 'w'), indent=1)
d = json.loads(line)
print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['frac_of_measured_copy'])
