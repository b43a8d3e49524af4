import csv, collections, sys, glob
for d in sys.argv[1:]:
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in agg.items():
            if not any(t in k for t in ('tensor_', 'general', 'contact')): continue
            n = len(next(iter(v.values())))
            print(k, 'dispatches', n, 'VGPR', rows[0].get('VGPR_Count'), 'LDS', rows[0].get('LDS_Block_Size'))
            for c, vals in sorted(v.items()):
                print('   %-28s mean %.4g' % (c, sum(vals) / len(vals)))
