"""Print one step's kernel timeline (start offset, duration, gap) from a rocprofv3 kernel trace csv."""
import csv
import glob
import sys

tag = sys.argv[1]
f = glob.glob('gpurun_out/%s/*/*kernel_trace.csv' % tag)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last occurrence of the transform kernel starts the last step
idx = [i for i, r in enumerate(rows) if 'k_transform' in r['Kernel_Name']]
start = idx[-2]
end = idx[-1]
t0 = int(rows[start]['Start_Timestamp'])
prev_end = t0
for r in rows[start:end + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%8.1f us  dur %7.1f  gap %6.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r['Kernel_Name'][:60]))
    prev_end = e
