"""Two-stream timeline of the last training step in a rocprofv3 --kernel-trace run (overlap on): per-queue busy time, union busy,
gaps on the main queue, kernel time by name and queue.  python tools/timeline_summary.py <rocprofv3 output dir>"""
import csv, glob, collections, sys
f=glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows:
    r['s']=int(r['Start_Timestamp']); r['e']=int(r['End_Timestamp']); r['q']=r['Queue_Id']+'/'+r['Stream_Id']
rows.sort(key=lambda r:r['s'])
adam=[i for i,r in enumerate(rows) if 'adam_step' in r['Kernel_Name']]
print(len(rows),'kernels; adam launches',len(adam))
# one step = from after adam[-3] to adam[-1] if two adam launches per step; detect
per=2 if len(adam)>=4 and (adam[-1]-adam[-2])<5 else 1
a_end=adam[-1]; a_start=adam[-1-per]
seg=rows[a_start+1:a_end+1]
t0=seg[0]['s']; t1=max(r['e'] for r in seg)
print('step span ms %.3f kernels %d'%((t1-t0)/1e6,len(seg)))
byq=collections.defaultdict(list)
for r in seg: byq[r['q']].append(r)
def union(l):
    ev=sorted((r['s'],r['e']) for r in l); u=0; cs,ce=ev[0]
    for s,e in ev[1:]:
        if s>ce: u+=ce-cs; cs,ce=s,e
        else: ce=max(ce,e)
    return u+ce-cs
for q,l in sorted(byq.items()):
    print('queue',q,'n',len(l),'sum ms %.3f union %.3f first %.3f last %.3f'%(sum(r['e']-r['s'] for r in l)/1e6,union(l)/1e6,(l[0]['s']-t0)/1e6,(max(r['e'] for r in l)-t0)/1e6))
print('all union busy %.3f idle %.3f'%(union(seg)/1e6,(t1-t0-union(seg))/1e6))
main=max(byq.items(), key=lambda kv: len(kv[1]))[1]
main.sort(key=lambda r:r['s'])
gaps=[]
for a,b in zip(main,main[1:]):
    g=b['s']-a['e']; gaps.append((g,a['Kernel_Name'][:45],b['Kernel_Name'][:45],(a['e']-t0)/1e6))
print('main gaps total %.3f ms; >3us: %.3f ms (%d)'%(sum(g for g,*_ in gaps if g>0)/1e6,sum(g for g,*_ in gaps if g>3000)/1e6,sum(1 for g,*_ in gaps if g>3000)))
for g in sorted(gaps,reverse=True)[:25]: print('%.1f us'%(g[0]/1e3), g[1],'->',g[2],'@%.2f'%g[3])
# kernel duration by name in this step (main vs side)
agg=collections.defaultdict(lambda:[0,0])
for r in seg:
    k=(r['q'],r['Kernel_Name'].split('(')[0][:60]); agg[k][0]+=1; agg[k][1]+=r['e']-r['s']
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:30]: print('%-8s %-62s n=%3d %.3f ms'%(k[0],k[1],v[0],v[1]/1e6))
