import csv, sys, collections
per = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_resize_stream' in r['Kernel_Name'] or 'k_blur_stream' in r['Kernel_Name'] or 'k_fast' in r['Kernel_Name']:
        key = (r['Kernel_Name'].split('(')[0][-20:], r['Grid_Size_X'], r['Grid_Size_Y'])
        per[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print(k, len(v), 'avg us %.1f' % (sum(v)/len(v)/1e3))
