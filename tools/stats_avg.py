"""Average duration of the conv-engine launches in a rocprofv3 --stats CSV (all template instantiations pooled), to be read
next to bench.py's roofline.avg_launch_ms.  usage: stats_avg.py <kernel_stats.csv>"""
import csv, sys
n = tot = 0
for row in csv.DictReader(open(sys.argv[1])):
    if any(k in row['Name'] for k in ('conv_f16s_kernel', 'conv_f16x3_kernel', 'conv_mfma_kernel', 'conv_group_kernel', 'conv_chain_kernel')):
        n += int(row['Calls']); tot += float(row['TotalDurationNs'])
print('conv engine launches in the trace: %d, total %.3f ms, average %.2f us per launch' % (n, tot / 1e6, tot / max(n, 1) / 1e3))
