#!/usr/bin/env python3
"""Copy the summaries of the last tools/prof.sh <tag> run (gpurun_out/<tag>/) into profiles/<tag>/ (what is committed).
usage: tools/save_profiles.py r03"""
import glob, os, shutil, subprocess, json, re, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
os.makedirs("profiles/" + TAG, exist_ok=True)
for d in glob.glob('gpurun_out/' + TAG + '/*/vm'):
    files = glob.glob(d + '/*')
    newest = int(os.path.basename(max(files, key=os.path.getmtime)).split("_")[0])
    for f in files:
        if int(os.path.basename(f).split('_')[0]) != newest:
            os.remove(f)
shutil.copy(glob.glob('gpurun_out/' + TAG + '/kt/vm/*kernel_stats.csv')[0], 'profiles/' + TAG + '/kernel_stats.csv')
out = subprocess.run(['python3', 'tools/prof_summary.py', 'gpurun_out/' + TAG + '', '65536'], capture_output=True, text=True).stdout
open('profiles/' + TAG + '/rocprofv3_summary.txt', 'w').write(out)
shutil.copy('profiles/hbm_traffic.json', 'profiles/' + TAG + '/hbm_traffic.json')
line = [l for l in open('gpurun_out/' + TAG + '/bench_kt.log') if l.startswith('{"metric')][-1]
open('profiles/' + TAG + '/bench_default.json', 'w').write(line)
shutil.copy('gpurun_out/' + TAG + '/series.txt', 'profiles/' + TAG + '/stream_series.txt')
old = open('profiles/' + TAG + '/lean_trip_profile.txt').read()
hdr = "".join(l + "\n" for l in old.split("\n") if l.startswith("#"))
body = "\n".join(l for l in open('gpurun_out/' + TAG + '/lean_profile.txt').read().split("\n") if "amdgpu.ids" not in l)
open('profiles/' + TAG + '/lean_trip_profile.txt', 'w').write(hdr + body)
rows = []
for l in open('gpurun_out/' + TAG + '/series.txt'):
    m = re.match(r"(\d+)\s+kernel (.+?) ([\d.]+) ms\s+step_ms ([\d.]+)\s+GB/s ([\d.]+)", l)
    if m:
        rows.append({"streams": int(m.group(1)), "kernel": m.group(2), "kernel_ms": float(m.group(3)), "step_ms": float(m.group(4)), "GBps": float(m.group(5))})
json.dump({"what": "single-GPU time of one batch over batch sizes (cfg2 streams, tools/series.sh): the cost of one shard of a strong-scaling run; a launch lasts as long as its longest stream, so 8,192 streams take about three quarters of the time of 65,536",
           "rows": rows}, open('profiles/' + TAG + '/stream_series.json', 'w'), indent=1)
d = json.loads(line)
print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['frac_of_measured_copy'])
