import sqlite3,collections,sys
def per_kernel(db):
    c=sqlite3.connect(db)
    tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    suf=[t for t in tabs if t.startswith('rocpd_pmc_event')][0][len('rocpd_pmc_event'):]
    q=f"""select d.id, ks.kernel_name, p.name, sum(e.value), d.end-d.start from rocpd_pmc_event{suf} e
      join rocpd_info_pmc{suf} p on e.pmc_id=p.id
      join rocpd_kernel_dispatch{suf} d on e.event_id=d.event_id
      join rocpd_info_kernel_symbol{suf} ks on d.kernel_id=ks.id group by 1,3 order by 1"""
    by=collections.defaultdict(lambda: collections.defaultdict(list))
    for did,kn,pn,v,dur in c.execute(q):
        k=[x for x in ('k_loopfilter_rows4','k_recon_inter','k_recon_intra4') if x in kn]
        if k: by[k[0]][pn].append((v,dur))
    return by
d=sys.argv[1]
for name in ('fetch','write','sq'):
    by=per_kernel('%s/%s_results.db'%(d,name))
    for k,v in sorted(by.items()):
        for pn,vals in sorted(v.items()):
            vs=[x[0] for x in vals]; ds=[x[1] for x in vals]
            print(name,k,pn,'n',len(vs),'avg %.4g'%(sum(vs)/len(vs)),'max %.4g'%max(vs), 'avg_dur_us %.1f'%(sum(ds)/len(ds)/1e3))
