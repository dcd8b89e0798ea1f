import csv, collections, sys, glob
f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
rows = list(csv.DictReader(open(f)))
nblocks = (1 << int(sys.argv[2])) // 256
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name']
    if 'oroch' not in k or 'tile' in k or 'finish' in k: continue
    name = k.split('<')[0].split('::')[-1]
    agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name, cs in agg.items():
    print(name, {c: round(sum(v)/len(v)/nblocks,1) for c, v in sorted(cs.items())})
