import csv, collections, re, sys, glob
for d in sys.argv[1:]:
    f = (glob.glob(d + '/*/*counter_collection.csv') + glob.glob(d + '/*counter_collection.csv'))[0]
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list)); meta = {}
    for r in rows:
        m = re.search(r'(select_\w+|nms_\w+|score_hist\w*|activate\w+|assign_\w+|pack_\w+)', r['Kernel_Name'])
        if not m: continue
        short = m.group(1)
        agg[short][r['Counter_Name']].append(float(r['Counter_Value']))
        meta[short] = dict(vgpr=r['VGPR_Count'], sgpr=r['SGPR_Count'], grid=r['Grid_Size'], us=(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k, v in agg.items():
        if meta[k]['us'] < 15: continue
        print(k, meta[k], {c: round(sum(x) / len(x)) for c, x in v.items()})
